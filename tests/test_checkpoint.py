"""Checkpoint compatibility (reference utils/utils.py:9-49,60-61): CPU only, no kernels run."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "taming_event_flow_amd")
REF = "/root/reference"


def _net(seed):
    from taming_event_flow_amd.models.model import RecEVFlowNet

    torch.manual_seed(seed)
    return RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2)


def test_save_load_round_trip(tmp_path):
    from taming_event_flow_amd.utils import checkpoint as utils

    a, b = _net(1), _net(2)
    assert not torch.equal(a.state_dict()["arch.resblocks.0.conv1.weight"], b.state_dict()["arch.resblocks.0.conv1.weight"])
    path = utils.save_model(a, str(tmp_path))
    assert path.endswith(os.path.join("model", "data", "model.pth"))
    b, epoch = utils.load_model(str(tmp_path), b, torch.device("cpu"))
    assert epoch == 0
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    # an unknown run leaves the model untouched (utils/utils.py:10-13)
    c = _net(3)
    ref = {k: v.clone() for k, v in c.state_dict().items()}
    c, epoch = utils.load_model(str(tmp_path / "missing"), c, torch.device("cpu"))
    assert epoch == 0 and all(torch.equal(v, c.state_dict()[k]) for k, v in ref.items())
    # plain state-dict files
    utils.save_state_dict(str(tmp_path), a.state_dict())
    sd = utils.load_state_dict(str(tmp_path))
    assert sd is not None and set(sd) == set(a.state_dict())
    assert utils.load_state_dict(str(tmp_path / "missing")) is None


def test_starting_epoch_from_loss_file(tmp_path):
    from taming_event_flow_amd.utils import checkpoint as utils

    art = tmp_path / "artifacts"
    utils.save_model(_net(1), str(art))
    os.makedirs(tmp_path / "metrics")
    with open(tmp_path / "metrics" / "loss", "w") as f:      # mlflow metric file: timestamp value step
        f.write("1700000000 2.5 0\n1700000100 2.1 1\n1700000200 1.9 7\n")
    _, epoch = utils.load_model(str(art), _net(2), torch.device("cpu"))
    assert epoch == 7


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (this container only)")
def test_reference_pickle_loads_into_hip_model(tmp_path):
    """A module pickled by the reference's own classes restores into this package's RecEVFlowNet when this package's
    directory shadows the reference's `models`, and the other way round."""
    ref_ckpt = tmp_path / "ref" / "model" / "data" / "model.pth"
    ours_ckpt = tmp_path / "ours"
    os.makedirs(ref_ckpt.parent)
    make_ref = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from models.model import RecEVFlowNet\n"
        "torch.manual_seed(11)\n"
        "m = RecEVFlowNet({'name': 'RecEVFlowNet', 'final_w_scale': 0.01}, 2)\n"
        "torch.save(m, %r); torch.save(m.state_dict(), %r)\n" % (REF, str(ref_ckpt), str(tmp_path / "ref_sd.pth")))
    subprocess.run([sys.executable, "-c", make_ref], check=True, cwd=str(tmp_path))
    load_ours = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from models.model import RecEVFlowNet\n"
        "from utils.checkpoint import load_model, save_model\n"
        "import models.model as mm; assert mm.__file__.startswith(%r), mm.__file__\n"
        "m = RecEVFlowNet({'name': 'RecEVFlowNet', 'final_w_scale': 0.01}, 2)\n"
        "m, _ = load_model(%r, m, torch.device('cpu'))\n"
        "sd = torch.load(%r)\n"
        "assert set(sd) == set(m.state_dict())\n"
        "assert all(torch.equal(v, m.state_dict()[k]) for k, v in sd.items())\n"
        "save_model(m, %r)\n" % (ROOT, PKG, PKG, str(tmp_path / "ref"), str(tmp_path / "ref_sd.pth"), str(ours_ckpt)))
    subprocess.run([sys.executable, "-c", load_ours], check=True, cwd=str(tmp_path))
    # the reference's restore path (utils/utils.py:20-31) on a state-dict saved from this package's model
    back = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from models.model import RecEVFlowNet\n"
        "m = RecEVFlowNet({'name': 'RecEVFlowNet', 'final_w_scale': 0.01}, 2)\n"
        "sd = torch.load(%r)\n"
        "new = m.state_dict(); new.update(sd); m.load_state_dict(new)\n"
        "assert all(torch.equal(v, m.state_dict()[k]) for k, v in sd.items())\n" % (REF, str(tmp_path / "ours_sd.pth")))
    sd_ours = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "m = torch.load(%r, weights_only=False)\n"
        "torch.save(m.state_dict(), %r)\n" % (ROOT, PKG, str(ours_ckpt / "model" / "data" / "model.pth"),
                                              str(tmp_path / "ours_sd.pth")))
    subprocess.run([sys.executable, "-c", sd_ours], check=True, cwd=str(tmp_path))
    subprocess.run([sys.executable, "-c", back], check=True, cwd=str(tmp_path))


def test_drop_in_import_mode(tmp_path):
    """with this package's directory first on sys.path, `from loss.flow import *` etc. resolve to the
    HIP-backed modules, and reference modules that have no counterpart here stay reachable behind them."""
    code = (
        "import sys, importlib.util as iu\n"
        "ref = %r\n"
        "import os\n"
        "if os.path.isdir(ref): sys.path.insert(0, ref)\n"
        "sys.path.insert(0, %r)\n"
        "from loss.flow import *\n"
        "from models.model import *\n"
        "assert Iterative.__module__ == 'loss.flow' and RecEVFlowNet.__module__ == 'models.model'\n"
        "from dataloader.encodings import events_to_channels, events_to_voxel, events_to_image\n"
        "from loss.flow_val import *\n"
        "from utils.iwe import compute_pol_iwe\n"
        "import loss.flow, models.model, dataloader.encodings, utils.iwe\n"
        "for m in (loss.flow, models.model, dataloader.encodings, utils.iwe):\n"
        "    assert m.__file__.startswith(%r), m.__file__\n"
        "assert Iterative.__module__ == 'loss.flow_val'\n"
        "if os.path.isdir(ref):\n"
        "    for name in ('utils.visualization', 'utils.utils', 'dataloader.h5'):\n"
        "        spec = iu.find_spec(name)\n"
        "        assert spec is not None and spec.origin.startswith(ref), (name, spec)\n" % (REF, PKG, PKG))
    subprocess.run([sys.executable, "-c", code], check=True, cwd=str(tmp_path))
