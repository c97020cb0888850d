"""Checkpoint compatibility with the reference's ``utils/utils.py`` (:9-49 ``load_model``, :60-61 ``save_model``,
:64-85 ``save_state_dict`` / ``load_state_dict``) without mlflow.  (Not named ``utils/utils.py``: in drop-in mode the
reference's own module of that name, with its logging helpers, must stay importable — and it works unchanged on this
package's model, whose parameter names are the reference's.)

The reference stores a whole pickled module at ``<mlflow artifact uri>/model/data/model.pth`` and restores it with
``torch.load(...).state_dict()`` merged over the fresh model's own ``state_dict()`` (:20-31).  The HIP-backed
``RecEVFlowNet`` has the reference's parameter names, so both directions work: a reference ``model.pth`` (pickled module
or plain state-dict) loads into this model, and what is saved here loads into the reference model.  mlflow is optional:
a run id is resolved through mlflow when it is importable, otherwise `prev_runid` is taken as a path (a ``.pth`` file,
an artifact directory or a run directory).
"""

import os

import numpy as np
import torch

_MODEL_REL = os.path.join("model", "data", "model.pth")


def _artifact_dir(runid):
    """Directory that holds ``model/data/model.pth`` for `runid` (mlflow run id or filesystem path), or None."""
    if runid is None or runid == "":
        return None
    cands = []
    if os.path.exists(runid):
        cands += [runid, os.path.join(runid, "artifacts")]
    else:
        try:
            import mlflow

            uri = mlflow.get_run(runid).info.artifact_uri
            cands.append(uri[7:] if uri[:7] == "file://" else uri)
        except Exception:
            return None
    for c in cands:
        if os.path.isfile(c) or os.path.isfile(os.path.join(c, _MODEL_REL)):
            return c
    return cands[0] if cands else None


def load_model(prev_runid, model, device, curr_run=None, tb_writer=None):
    """utils/utils.py:9-49: returns (model, starting_epoch); an unknown run leaves the model untouched (epoch 0)."""
    art = _artifact_dir(prev_runid)
    if art is None:
        return model, 0
    model_file = art if os.path.isfile(art) else os.path.join(art, _MODEL_REL)
    starting_epoch = 0
    if os.path.isfile(model_file):
        loaded = torch.load(model_file, map_location=device, weights_only=False)
        if isinstance(loaded, torch.nn.Module):       # the reference pickles the module itself (:60-61)
            loaded = loaded.state_dict()
        new_params = model.state_dict()
        new_params.update(loaded)                      # :29-31
        model.load_state_dict(new_params)
        loss_file = os.path.join(os.path.dirname(art.rstrip("/")), "metrics", "loss")      # :33-44
        if os.path.isfile(loss_file):
            loss = np.atleast_2d(np.genfromtxt(loss_file))
            if tb_writer is not None:
                for i in range(loss.shape[0]):
                    tb_writer.add_scalar("loss", loss[i, 1], int(loss[i, 2]))
            starting_epoch = int(loss[-1][-1])
        print("Model restored from " + str(prev_runid) + "\n")
    else:
        print("No model found at " + str(prev_runid) + "\n")
    return model, starting_epoch


def save_model(model, artifact_dir=None):
    """utils/utils.py:60-61: the whole module, pickled, at ``<artifact dir>/model/data/model.pth``."""
    if artifact_dir is None:
        import mlflow

        mlflow.pytorch.log_model(model, "model", conda_env={"dependencies": []})
        return None
    path = os.path.join(artifact_dir, _MODEL_REL)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(model, path)
    return path


def save_state_dict(runid, state_dict, dir="state_dict/", filename="state_dict.pth"):
    """utils/utils.py:64-67 (file based when `runid` is a directory)."""
    if os.path.isdir(runid):
        os.makedirs(os.path.join(runid, dir), exist_ok=True)
        torch.save(state_dict, os.path.join(runid, dir, filename))
        return
    import mlflow

    mlflow.start_run(runid)
    mlflow.pytorch.log_state_dict(state_dict, "state_dict")
    mlflow.end_run()


def load_state_dict(runid, dir="state_dict/", filename="state_dict.pth"):
    """utils/utils.py:70-83: the saved dict on the CPU, or None."""
    art = runid if os.path.isdir(runid) else _artifact_dir(runid)
    model_dict = None
    if art is not None and os.path.isfile(os.path.join(art, dir, filename)):
        model_dict = torch.load(os.path.join(art, dir, filename), map_location=torch.device("cpu"), weights_only=False)
        print("Model restored from " + str(runid))
    else:
        print("No model found at " + str(runid))
    return model_dict
