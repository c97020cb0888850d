"""RecEVFlowNet — drop-in for the reference's ``models/model.py`` (:6-85) on the MI355X pass engine.

``RecEVFlowNet(kwargs, num_bins=2, key="flow", min_size=16)``; ``forward(x) -> {key: [4 x [B, 2, H, W]]}`` (coarse to
fine, already scaled by 2^(3-k) and cropped to the input size); ``states`` (get copies / set), ``detach_states()``,
``reset_states()``; identical ``state_dict`` keys, 31 365 352 parameters with the default config.

One call = one recurrent pass = one autograd node (`engine._PassFn`): padding, the four up-samplings to the input size,
their scale factors and the crop are launches of that pass, not separate tensor operations.
"""

import numpy as np
import torch

from .arch import MultiResUNetRecurrent, NetPlan  # noqa: F401
from .model_util import ImagePadder, copy_states

__all__ = ["RecEVFlowNet", "MultiResUNetRecurrent"]

# the reference's architecture defaults (models/model.py:21-36); the config may override them
_DEFAULTS = dict(base_channels=64, num_encoders=4, num_residual_blocks=2, num_output_channels=2, skip_type="sum",
                 norm=None, use_upsample_conv=True, kernel_size=3, encoder_stride=2, channel_multiplier=2,
                 final_activation="tanh", activations=("relu", None), recurrent_block_type="convgru")


class RecEVFlowNet(torch.nn.Module):
    def __init__(self, kwargs, num_bins=2, key="flow", min_size=16):
        super().__init__()
        cfg = dict(_DEFAULTS, num_bins=num_bins, min_size=min_size)
        cfg.update({k: v for k, v in kwargs.items() if k != "name"})
        if "activations" in cfg:
            cfg["activations"] = tuple(cfg["activations"])
        self.arch = MultiResUNetRecurrent(cfg)
        self.key, self.num_encoders = key, self.arch.num_encoders
        self.image_padder = ImagePadder(min_size=min_size)      # geometry record; see model_util.py

    def __str__(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + f"\nTrainable parameters: {n}"

    # ---- recurrent state (reference :42-63): `states` reads copies and assigns a new list ------------------------------
    def _read_states(self):
        return copy_states(self.arch.states)

    def _assign_states(self, new_states):
        self.arch.states = list(new_states)

    states = property(_read_states, _assign_states)

    def detach_states(self):
        """Truncated BPTT: keep the values, cut the graph."""
        self.arch.states = [None if s is None else s.detach() for s in self.arch.states]

    def reset_states(self):
        self.arch.states = [None for _ in range(self.arch.num_states)]

    def forward(self, x):
        return {self.key: self.arch.step(x)}
