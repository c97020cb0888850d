"""CPU-side checks of the model mirror: constructor, parameter count and state_dict layout (no kernels run)."""
import torch


def test_state_dict_layout():
    from taming_event_flow_amd.models.model import RecEVFlowNet

    net = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2)
    sd = net.state_dict()
    assert sum(p.numel() for p in net.parameters()) == 31365352        # SURVEY.md §8a M7 [probe]
    assert len(sd) == 56
    expect = {
        "arch.encoders.0.conv.conv2d.weight": (64, 2, 3, 3),
        "arch.encoders.0.recurrent_block.reset_gate.weight": (64, 128, 3, 3),
        "arch.encoders.3.recurrent_block.out_gate.bias": (512,),
        "arch.resblocks.1.conv2.weight": (512, 512, 3, 3),
        "arch.decoders.0.conv2d.weight": (256, 512, 3, 3),
        "arch.decoders.1.conv2d.weight": (128, 258, 3, 3),
        "arch.decoders.3.conv2d.weight": (32, 66, 3, 3),
        "arch.preds.0.conv2d.weight": (2, 256, 1, 1),
        "arch.preds.3.conv2d.bias": (2,),
    }
    for k, shape in expect.items():
        assert tuple(sd[k].shape) == shape, k
    # final_w_scale reaches the prediction heads (reference arch.py:181-194)
    assert sd["arch.preds.0.conv2d.weight"].abs().max() <= 0.01
    assert net.arch.states == [None] * 4 and net.states == [None] * 4
    v = RecEVFlowNet({"name": "RecEVFlowNet"}, 5)
    assert v.state_dict()["arch.encoders.0.conv.conv2d.weight"].shape == (64, 5, 3, 3)
