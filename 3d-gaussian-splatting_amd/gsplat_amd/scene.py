"""Deterministic synthetic scenes and cameras for tests and bench.

Portable RNG (``numpy.random.default_rng``), cast to fp32, draw order exactly:
xyz, log-scales, quaternions, opacity, SH-dc, SH-rest.  The values are the *activated* inputs of
the hot path (what /root/reference/src/training/trainer.py:399-406 hands to the backend).
"""

from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

SH_C0 = 0.28209479177387814


def make_scene(n: int, seed: int = 0, sh_degree: int = 3, s_min: float = 0.002, s_max: float = 0.02,
               device: str = "cpu") -> Dict[str, torch.Tensor]:
    """xyz~U(-1,1)^3, scales=exp(U(ln s_min, ln s_max)) per axis, unit quats (w first),
    opacity~U(0.05,0.95), SH-dc giving base colours in [0,1], SH-rest~N(0,0.05^2) (zeros at degree 0)."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, (n, 3))
    scales = np.exp(rng.uniform(math.log(s_min), math.log(s_max), (n, 3)))
    quats = rng.normal(0.0, 1.0, (n, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, (n,))
    sh_dc = rng.uniform(-1.0, 1.0, (n, 1, 3)) * (0.5 / SH_C0)
    sh_rest = rng.normal(0.0, 0.05, (n, 15, 3))
    if sh_degree == 0:
        sh_rest[:] = 0.0
    shs = np.concatenate([sh_dc, sh_rest], axis=1)

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)

    return {"means3D": t(xyz), "scales": t(scales), "rotations": t(quats), "opacities": t(opac)[:, None],
            "shs": t(shs)}


def make_intrinsics(width: int, height: int, device: str = "cpu") -> torch.Tensor:
    """Pinhole with the reference's focal heuristic fx=fy=1.2*max(W,H)
    (/root/reference/src/video/calibrate.py:350-363), principal point at the image centre."""
    f = 1.2 * max(width, height)
    return torch.tensor([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]],
                        dtype=torch.float32, device=device)


def make_camera(v: int, n_views: int, radius: float = 4.0, device: str = "cpu") -> torch.Tensor:
    """World->camera [4,4] (OpenCV axes: x right, y down, z forward) for view v of a ring of
    ``n_views`` cameras in the xz-plane looking at the origin; view 0 = identity rotation, t=(0,0,radius)."""
    th = 2.0 * math.pi * v / max(n_views, 1)
    fwd = np.array([math.sin(th), 0.0, math.cos(th)])
    down = np.array([0.0, 1.0, 0.0])
    right = np.cross(down, fwd)
    R = np.stack([right, down, fwd])
    centre = -radius * fwd
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = -R @ centre
    return torch.from_numpy(M.astype(np.float32)).to(device)


def make_v_render(width: int, height: int, seed: int = 2, device: str = "cpu") -> torch.Tensor:
    """Synthetic upstream gradient for fwd+bwd timing/parity: v_render ~ U(-1,1), seed 2 (§8d)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(-1.0, 1.0, (height, width, 3)).astype(np.float32)).to(device)
