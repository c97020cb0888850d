"""RecEVFlowNet — drop-in for the reference's ``models/model.py``.

``RecEVFlowNet(kwargs, num_bins=2, key="flow", min_size=16)``; ``forward(x) -> {key: [4 x [B, 2, H, W]]}``;
``states`` (get clones / set), ``detach_states()``, ``reset_states()``; identical ``state_dict`` keys
(arch.encoders.{0-3}.conv.conv2d / .recurrent_block.{reset,update,out}_gate, arch.resblocks.{0,1}.conv{1,2},
arch.decoders.{0-3}.conv2d, arch.preds.{0-3}.conv2d), 31 365 352 parameters with the default config.
"""

import torch

from .arch import *  # noqa: F401,F403
from .arch import MultiResUNetRecurrent
from .base import BaseModel
from .model_util import ImagePadder, copy_states
from .submodules import upsample_bilinear

__all__ = ["RecEVFlowNet", "MultiResUNetRecurrent"]


class RecEVFlowNet(BaseModel):
    """Recurrent version of the EV-FlowNet model (reference models/model.py:6-85)."""

    net_type = MultiResUNetRecurrent
    recurrent_block_type = "convgru"
    activations = ["relu", None]

    def __init__(self, kwargs, num_bins=2, key="flow", min_size=16):
        super().__init__()
        self.image_padder = ImagePadder(min_size=min_size)
        self.key = key
        arch_kwargs = {
            "num_bins": num_bins,
            "base_channels": 64,
            "num_encoders": 4,
            "num_residual_blocks": 2,
            "num_output_channels": 2,
            "skip_type": "sum",
            "norm": None,
            "use_upsample_conv": True,
            "kernel_size": 3,
            "encoder_stride": 2,
            "channel_multiplier": 2,
            "final_activation": "tanh",
            "activations": self.activations,
            "recurrent_block_type": self.recurrent_block_type,
        }
        arch_kwargs.update(kwargs)  # update params with config
        arch_kwargs.pop("name", None)
        self.arch = self.net_type(arch_kwargs)
        self.num_encoders = arch_kwargs["num_encoders"]

    @property
    def states(self):
        return copy_states(self.arch.states)

    @states.setter
    def states(self, states):
        self.arch.states = states

    def detach_states(self):
        """Truncated BPTT: keep the state, cut the graph (reference model.py:50-60)."""
        detached_states = []
        for state in self.arch.states:
            if type(state) is tuple:
                detached_states.append(tuple(hidden.detach() for hidden in state))
            else:
                detached_states.append(state.detach())
        self.arch.states = detached_states

    def reset_states(self):
        self.arch.states = [None] * self.arch.num_states

    def forward(self, x):
        # image padding (top / left, to a multiple of min_size)
        x = self.image_padder.pad(x).contiguous()
        multires_flow = self.arch.forward(x)
        # upsample flow estimates to the original input resolution (reference model.py:72-83)
        flow_list = []
        for i, flow in enumerate(multires_flow):
            scaling_h = x.shape[2] / flow.shape[2]
            scaling_w = x.shape[3] / flow.shape[3]
            scaling_flow = 2 ** (self.num_encoders - i - 1)
            upflow = upsample_bilinear(flow, scaling_h, scaling_w, mul=float(scaling_flow))
            flow_list.append(self.image_padder.unpad(upflow))
        return {self.key: flow_list}
