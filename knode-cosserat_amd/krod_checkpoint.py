"""Model interchange with the reference's checkpoints.

The reference stores a trained model as ``torch.save({'robot': robot, 'dtw': ..., 'loss': ...,
'optim': ...}, path)`` (physics_train.py:284-288, 388-392) - a pickle that names the class
``cosserat_ode_torch.CosseratRodTorch`` - and reads it back with
``torch.load(path)['robot'].nn_models`` (cosserat_ode.py:81-88, physics_train.py:187).  Because this
package provides a module and a class of the same names, such a file unpickles into OUR
``CosseratRodTorch`` (``__setstate__`` there adopts the reference's attribute dictionary), and a file
written by ``torch.save({'robot': our_robot})`` is readable by the reference.  No reference code is
needed or executed on either side.

A pickle is still a pickle: only load files you trust.  ``save_weights`` / ``load_weights`` are the
weights-only alternative (a plain ``.npz``: one array per ``state_dict`` key plus the layer strings
the NumPy class dispatches on, cosserat_ode.py:90-112).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def load_checkpoint(path, device="cpu"):
    """The reference's ``torch.load(MODEL_SAVE_PATH)`` with an explicit device.  Returns the stored dict;
    ``ckpt['robot']`` is a ``cosserat_ode_torch.CosseratRodTorch`` of this package."""
    import cosserat_ode_torch  # noqa: F401  (the module name the pickle refers to must resolve to ours)
    ckpt = torch.load(path, map_location=device, weights_only=False)
    rob = ckpt.get("robot") if isinstance(ckpt, dict) else None
    if rob is not None:
        rob.to(device)
    return ckpt


def save_checkpoint(path, robot, dtw=None, loss=None, optimizer=None):
    """physics_train.py:284-288."""
    torch.save({"robot": robot, "dtw": dtw if dtw is not None else [], "loss": loss if loss is not None else [],
                "optim": optimizer.state_dict() if optimizer is not None else {}}, path)


def layer_strings(nn_models):
    return [str(layer) for layer in nn_models]


def save_weights(path, nn_models):
    """Weights-only file: ``layers`` (the strings of every module, in order) and one array per
    ``state_dict`` key ('0.weight', '0.bias', '2.weight', ...)."""
    arrays = {"p_" + k: v.detach().cpu().numpy() for k, v in nn_models.state_dict().items()}
    np.savez(path, layers=np.array(layer_strings(nn_models)), **arrays)


_ACTS = {"Tanh": nn.Tanh, "Softplus": nn.Softplus, "ReLU": nn.ReLU, "ELU": nn.ELU, "Dropout": nn.Dropout}


def load_weights(path, device="cpu"):
    """-> (nn.ModuleList, param_ls): the two things the reference injects into a NumPy robot
    (physics_train.py:104-110: ``robot.nn_model``, ``robot.param_ls``) or assigns to
    ``CosseratRodTorch.nn_models``."""
    with np.load(path, allow_pickle=False) as f:
        layers = [str(s) for s in f["layers"]]
        params = {k[2:]: f[k] for k in f.files if k.startswith("p_")}
    mods = []
    for i, s in enumerate(layers):
        head = s.split("(")[0]
        if head == "Linear":
            w = params[f"{i}.weight"]
            lin = nn.Linear(w.shape[1], w.shape[0], bias=f"{i}.bias" in params)
            with torch.no_grad():
                lin.weight.copy_(torch.as_tensor(w))
                if lin.bias is not None:
                    lin.bias.copy_(torch.as_tensor(params[f"{i}.bias"]))
            mods.append(lin)
        elif head in _ACTS:
            mods.append(_ACTS[head]())
        else:
            raise ValueError(f"unknown layer {s!r}")
    ml = nn.ModuleList(mods).to(device)
    param_ls = [t.detach().cpu().numpy() for _, t in ml.state_dict().items()]
    return ml, param_ls
