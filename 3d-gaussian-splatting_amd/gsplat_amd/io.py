"""On-disk formats either side of the hot path, compatible with the reference in both
directions so that a scene trained on one side can be evaluated or exported by the other:

* checkpoints -- ``final_model.pth`` is the bare ``state_dict`` of the reference's ``GaussianModel`` (six tensors:
  xyz [N,3], features_dc [N,1,3], features_rest [N,15,3], scaling [N,3] (log), rotation [N,4] (w first), opacity [N,1]
  (logit)); ``checkpoint_<it>.pth`` wraps it as {iteration, model_state, optimizer_state, n_gaussians}
  (/root/reference/src/training/trainer.py:384-395,993).  Loading accepts both, like
  /root/reference/src/eval.py:24-38 and /root/reference/src/training/export.py:16-23.
* PLY -- binary little-endian, one ``vertex`` element, float32 properties in the order of
  /root/reference/src/training/export.py:62-72: x y z, nx ny nz (zeros), f_dc_0..2, opacity (logit), scale_0..2 (log),
  rot_0..3, f_rest_0..44 (``features_rest`` flattened row-major: index 3 * coefficient + channel).

Plain torch / numpy; no third-party PLY package is needed.
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .harness import FIELDS, FusedBucketAdam, SplatModel

PLY_PROPERTIES = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity",
                   "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
                  + [f"f_rest_{i}" for i in range(45)])


def model_state_dict(model: SplatModel) -> Dict[str, torch.Tensor]:
    """The reference model's ``state_dict()``: same six keys, shapes and dtypes, on the CPU."""
    return {name: getattr(model, name).detach().cpu().clone() for name, _ in FIELDS}


def _optimizer_state_dict(model: SplatModel, optimizer) -> Optional[dict]:
    if optimizer is None:
        return None
    if not isinstance(optimizer, FusedBucketAdam):
        return optimizer.state_dict()
    # the layout torch.optim.Adam.state_dict() produces for the reference's six groups (trainer.py:331-340)
    state, groups = {}, []
    for i, (g, p) in enumerate(zip(optimizer.param_groups, model.parameters())):
        st = optimizer.state[p]
        state[i] = {"step": torch.tensor(float(optimizer.step_count)), "exp_avg": st["exp_avg"].detach().cpu().clone(),
                    "exp_avg_sq": st["exp_avg_sq"].detach().cpu().clone()}
        groups.append({"lr": g["lr"], "name": g.get("name"), "betas": tuple(optimizer.betas), "eps": optimizer.eps,
                       "weight_decay": 0, "amsgrad": False, "params": [i]})
    return {"state": state, "param_groups": groups}


def save_checkpoint(path: str, model: SplatModel, optimizer=None, iteration: Optional[int] = None) -> None:
    """``iteration`` None: the bare state dict (``final_model.pth``); otherwise the wrapped periodic checkpoint."""
    sd = model_state_dict(model)
    if iteration is None:
        torch.save(sd, path)
    else:
        torch.save({"iteration": int(iteration), "model_state": sd,
                    "optimizer_state": _optimizer_state_dict(model, optimizer), "n_gaussians": model.n}, path)


def load_state(path: str, allow_pickle: bool = False) -> Tuple[Dict[str, torch.Tensor], dict]:
    """-> (model state dict, the wrapper's other entries); accepts bare and wrapped files (eval.py:24-38).
    Both formats hold only tensors, numbers and plain containers, so the file is read with ``weights_only=True``
    (no arbitrary unpickling); ``allow_pickle=True`` is the explicit opt-in for a trusted file that needs more."""
    ckpt = torch.load(path, map_location="cpu", weights_only=not allow_pickle)
    extra = {}
    if isinstance(ckpt, dict):
        for k in ("model_state", "state_dict"):
            if k in ckpt and isinstance(ckpt[k], dict):
                extra = {kk: v for kk, v in ckpt.items() if kk != k}
                ckpt = ckpt[k]
                break
    missing = [name for name, _ in FIELDS if name not in ckpt]
    if missing:
        raise ValueError(f"checkpoint {path} lacks {missing}")
    return ckpt, extra


def load_checkpoint(path: str, device="cuda") -> Tuple[SplatModel, dict]:
    sd, extra = load_state(path)
    model = SplatModel(int(sd["xyz"].shape[0]), device).load(**{name: sd[name].to(device) for name, _ in FIELDS})
    return model, extra


def export_ply(path: str, model_or_state) -> None:
    """Write the reference's PLY layout (export.py:53-121) from a ``SplatModel`` or a state dict."""
    sd = model_state_dict(model_or_state) if isinstance(model_or_state, SplatModel) else model_or_state
    n = int(sd["xyz"].shape[0])
    as_np = lambda t, w: np.asarray(t.detach().cpu() if torch.is_tensor(t) else t, dtype=np.float32).reshape(n, w)  # noqa: E731
    cols = [as_np(sd["xyz"], 3), np.zeros((n, 3), np.float32), as_np(sd["features_dc"], 3), as_np(sd["opacity"], 1),
            as_np(sd["scaling"], 3), as_np(sd["rotation"], 4), as_np(sd["features_rest"], 45)]
    mat = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property float {p}" for p in PLY_PROPERTIES] + ["end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(mat.tobytes())


def load_ply(path: str) -> Dict[str, torch.Tensor]:
    """Read a PLY of this layout (property order taken from its header) back into a reference-style state dict."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        props, n, fmt = [], None, None
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii").split()
            if tok[:1] == ["format"]:
                fmt = tok[1]
            elif tok[:2] == ["element", "vertex"]:
                n = int(tok[2])
            elif tok[:1] == ["property"]:
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"unsupported property type {tok[1]}")
                props.append(tok[2])
            elif tok[:1] == ["end_header"]:
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("expected a binary_little_endian PLY with a vertex element")
        mat = np.frombuffer(fh.read(n * len(props) * 4), dtype="<f4").reshape(n, len(props))
    col = {p: i for i, p in enumerate(props)}
    pick = lambda names: torch.from_numpy(np.stack([mat[:, col[p]] for p in names], axis=1).copy())  # noqa: E731
    rest = [f"f_rest_{i}" for i in range(45)]
    return {"xyz": pick(["x", "y", "z"]), "features_dc": pick(["f_dc_0", "f_dc_1", "f_dc_2"]).view(n, 1, 3),
            "features_rest": (pick(rest) if all(r in col for r in rest) else torch.zeros(n, 45)).view(n, 15, 3),
            "opacity": pick(["opacity"]), "scaling": pick(["scale_0", "scale_1", "scale_2"]),
            "rotation": pick(["rot_0", "rot_1", "rot_2", "rot_3"])}
