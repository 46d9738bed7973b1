"""Checkpoint and PLY formats: files written by the reference load here, files written
here have the reference's structure (tests/golden/reference_*.pth were saved by the reference's own GaussianModel,
see tests/golden/make_densify_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "3d-gaussian-splatting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

NAMES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def gio():
    try:
        from gsplat_amd import io as gio
    except ImportError as e:   # the package refuses to import without its HIP library
        pytest.skip(str(e))
    return gio


def test_reference_checkpoints_load(gio):
    bare = torch.load(os.path.join(HERE, "golden", "reference_final_model.pth"), map_location="cpu")
    for fname in ("reference_final_model.pth", "reference_checkpoint_10000.pth"):
        model, extra = gio.load_checkpoint(os.path.join(HERE, "golden", fname), device="cpu")
        assert model.n == 6
        for k in NAMES:
            assert torch.equal(getattr(model, k).detach(), bare[k]), (fname, k)
    assert extra["iteration"] == 10000 and extra["n_gaussians"] == 6 and "optimizer_state" in extra


def test_written_checkpoints_have_the_reference_structure(gio, tmp_path):
    from gsplat_amd.harness import SplatModel
    ref_bare = torch.load(os.path.join(HERE, "golden", "reference_final_model.pth"), map_location="cpu")
    ref_wrapped = torch.load(os.path.join(HERE, "golden", "reference_checkpoint_10000.pth"), map_location="cpu",
                             weights_only=False)
    model = SplatModel(6, "cpu").load(**ref_bare)
    opt = torch.optim.Adam([{"params": [getattr(model, k)], "lr": 1e-3, "name": k} for k in NAMES], lr=0.0, eps=1e-15)
    for k in NAMES:
        getattr(model, k).grad.fill_(1e-3)
    gio.save_checkpoint(str(tmp_path / "final_model.pth"), model)
    gio.save_checkpoint(str(tmp_path / "checkpoint_10000.pth"), model, optimizer=opt, iteration=10000)
    mine = torch.load(str(tmp_path / "final_model.pth"), map_location="cpu")
    assert list(mine.keys()) == list(ref_bare.keys()) or sorted(mine.keys()) == sorted(ref_bare.keys())
    for k in NAMES:
        assert mine[k].shape == ref_bare[k].shape and mine[k].dtype == ref_bare[k].dtype and torch.equal(mine[k], ref_bare[k])
    wrapped = torch.load(str(tmp_path / "checkpoint_10000.pth"), map_location="cpu", weights_only=False)
    assert sorted(wrapped.keys()) == sorted(ref_wrapped.keys())
    assert sorted(wrapped["optimizer_state"].keys()) == sorted(ref_wrapped["optimizer_state"].keys())


def test_ply_layout_and_round_trip(gio, tmp_path):
    from gsplat_amd.harness import SplatModel
    g = torch.Generator().manual_seed(1)
    n = 17
    model = SplatModel(n, "cpu").load(xyz=torch.randn(n, 3, generator=g), features_dc=torch.randn(n, 1, 3, generator=g),
                                      features_rest=torch.randn(n, 15, 3, generator=g), opacity=torch.randn(n, 1, generator=g),
                                      scaling=torch.randn(n, 3, generator=g), rotation=torch.randn(n, 4, generator=g))
    path = str(tmp_path / "scene.ply")
    gio.export_ply(path, model)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    lines = header.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    # property order of the reference's exporter (src/training/export.py:62-72)
    assert [l.split()[2] for l in lines[3:]] == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity",
                                                  "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
                                                 + [f"f_rest_{i}" for i in range(45)])
    assert all(l.split()[:2] == ["property", "float"] for l in lines[3:]) and len(body) == n * 62 * 4
    row0 = np.frombuffer(body[:62 * 4], dtype="<f4")
    assert np.array_equal(row0[:3], model.xyz[0].detach().numpy()) and np.all(row0[3:6] == 0)
    assert row0[9] == float(model.opacity[0, 0]) and row0[17 + 3 * 2 + 1] == float(model.features_rest[0, 2, 1])
    back = gio.load_ply(path)
    for k in NAMES:
        assert torch.equal(back[k], getattr(model, k).detach()), k
