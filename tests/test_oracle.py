"""CPU tests of the oracle itself (no GPU, no HIP): known answers, sequential-loop equivalence,
autograd vs closed-form backward (Appendix A.6), fp64 gradcheck of the whole pipeline."""
import math

import numpy as np
import pytest
import torch

from oracle import constants as K_
from oracle import loop_reference as loops
from oracle import torch_oracle as orc
from gsplat_amd import scene


def _identity_cam():
    vm = torch.eye(4)
    vm[2, 3] = 4.0
    return vm


def _single(opacity=0.8, scale=0.05, W=64, H=64, mean=(0.0, 0.0, 0.0)):
    means = torch.tensor([mean], dtype=torch.float32)
    quats = torch.tensor([[1.0, 0, 0, 0]])
    scales = torch.full((1, 3), scale)
    opac = torch.tensor([opacity])
    colors = torch.tensor([[0.2, 0.5, 0.9]])
    return means, quats, scales, opac, colors, _identity_cam()[None], scene.make_intrinsics(W, H)[None], W, H


def test_single_gaussian_known_answer():
    """Isotropic gaussian on the optical axis: cov2d = (f s / z)^2 + 0.3 on the diagonal; the four
    pixels around the centre sit at d = (+-0.5, +-0.5)."""
    means, quats, scales, opac, colors, vm, Ks, W, H = _single()
    r, a, meta = orc.rasterization(means, quats, scales, opac, colors, vm, Ks, W, H, backgrounds=torch.zeros(1, 3))
    f = 1.2 * 64
    var = (f * 0.05 / 4.0) ** 2 + K_.EPS2D
    assert torch.allclose(meta["means2d"][0, 0], torch.tensor([32.0, 32.0]), atol=1e-5)
    assert torch.allclose(meta["conics"][0, 0], torch.tensor([1 / var, 0.0, 1 / var]), rtol=1e-5, atol=1e-7)
    assert abs(float(meta["depths"][0, 0]) - 4.0) < 1e-6
    ext = min(K_.RADIUS_EXTEND, math.sqrt(2 * math.log(0.8 * 255)))
    assert meta["radii"][0, 0].tolist() == [math.ceil(ext * math.sqrt(var))] * 2
    alpha = 0.8 * math.exp(-0.5 * (0.25 + 0.25) / var)
    assert abs(float(a[0, 32, 32, 0]) - alpha) < 1e-6
    assert torch.allclose(r[0, 31, 31], alpha * colors[0], atol=1e-6)


def test_background_and_alpha():
    means, quats, scales, opac, colors, vm, Ks, W, H = _single()
    bg = torch.tensor([[0.1, 0.2, 0.3]])
    r0, a0, _ = orc.rasterization(means, quats, scales, opac, colors, vm, Ks, W, H, backgrounds=torch.zeros(1, 3))
    r1, a1, _ = orc.rasterization(means, quats, scales, opac, colors, vm, Ks, W, H, backgrounds=bg)
    assert torch.allclose(a0, a1)
    assert torch.allclose(r1, r0 + (1 - a0) * bg[0], atol=1e-7)
    assert torch.allclose(r1[0, 0, 0], bg[0])  # far corner untouched


@pytest.mark.parametrize("case", ["behind", "low_opacity", "off_screen", "far_side"])
def test_culling(case):
    kw = {"behind": dict(mean=(0, 0, -4.5)), "low_opacity": dict(opacity=1.0 / 256),
          "off_screen": dict(mean=(5.0, 0, 0)), "far_side": dict(mean=(0, 0, -3.995))}[case]
    means, quats, scales, opac, colors, vm, Ks, W, H = _single(**kw)
    r, a, meta = orc.rasterization(means, quats, scales, opac, colors, vm, Ks, W, H, backgrounds=torch.zeros(1, 3))
    assert meta["radii"].abs().sum() == 0
    assert meta["isect_ids"].numel() == 0
    assert float(r.abs().max()) == 0 and float(a.abs().max()) == 0


def test_depth_ordering_and_tie_break():
    """Nearer splat is blended first; equal depth -> lower gaussian index first."""
    W = H = 32
    means = torch.tensor([[0, 0, 0.5], [0, 0, -0.5], [0, 0, -0.5]], dtype=torch.float32)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 3)
    scales = torch.full((3, 3), 0.2)
    opac = torch.tensor([0.5, 0.5, 0.5])
    colors = torch.eye(3)
    r, a, meta = orc.rasterization(means, quats, scales, opac, colors, _identity_cam()[None],
                                   scene.make_intrinsics(W, H)[None], W, H, backgrounds=torch.zeros(1, 3))
    assert meta["flatten_ids"][:3].tolist() == [1, 2, 0]  # one tile pair: depth 3.5 (ids 1,2) then 4.5
    px = r[0, 16, 16]
    assert px[1] > px[2] > px[0] > 0


def test_transmittance_stop_rule():
    """Opaque stack: alpha is capped at 0.999, so T goes 1 -> 1e-3 -> (1e-6 <= 1e-4: stop, not added)."""
    W = H = 16
    n = 4
    means = torch.tensor([[0, 0, -0.1 * i] for i in range(n)], dtype=torch.float32)
    quats = torch.tensor([[1.0, 0, 0, 0]] * n)
    scales = torch.full((n, 3), 5.0)
    opac = torch.ones(n)
    colors = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1.0, 1.0]])
    r, a, _ = orc.rasterization(means, quats, scales, opac, colors, _identity_cam()[None],
                                scene.make_intrinsics(W, H)[None], W, H, backgrounds=torch.ones(1, 3))
    # nearest is index n-1 (white): contributes 0.999; the next one would leave T = 1e-6 -> stop.
    assert torch.allclose(a[0, 8, 8, 0], torch.tensor(0.999), atol=1e-6)
    assert torch.allclose(r[0, 8, 8], torch.tensor([0.999 + 1e-3] * 3), atol=1e-5)


def _small_scene(n=60, W=40, H=24, seed=3, dtype=torch.float32, sh_degree=3):
    sc = scene.make_scene(n, seed=seed, sh_degree=sh_degree, s_min=0.02, s_max=0.3)
    args = dict(means=sc["means3D"], quats=sc["rotations"], scales=sc["scales"], opacities=sc["opacities"][:, 0],
                colors=sc["shs"])
    args = {k: v.to(dtype) for k, v in args.items()}
    vm = scene.make_camera(1, 5)[None].to(dtype)
    Ks = scene.make_intrinsics(W, H)[None].to(dtype)
    return args, vm, Ks, W, H


def test_vectorised_composite_equals_sequential_loops():
    """The batched cumprod formulation == the literal per-pixel loop, forward and backward."""
    args, vm, Ks, W, H = _small_scene(dtype=torch.float64)
    pr = orc.project(args["means"], args["quats"], args["scales"], args["opacities"], vm[0], Ks[0], W, H)
    rgb = orc.sh_colors(3, args["means"], args["colors"], vm[0], pr["visible"])
    tw, th = (W + 15) // 16, (H + 15) // 16
    cnt, keys, flat = orc.tile_intersect(pr["means2d"], pr["radii"], pr["depths"], 16, tw, th)
    order = torch.sort(keys, stable=True).indices
    keys, flat = keys[order], flat[order].long()
    offs = torch.searchsorted(keys >> 32, torch.arange(tw * th + 1))
    bg = torch.tensor([0.3, 0.1, 0.2], dtype=torch.float64)

    leaves = [t.clone().requires_grad_(True) for t in (pr["means2d"], pr["conics"], args["opacities"], rgb)]
    img, alpha = orc.composite(leaves[0], leaves[1], leaves[2], leaves[3], bg, W, H, 16, flat, offs, budget=20000)
    rng = np.random.default_rng(0)
    v_img = rng.uniform(-1, 1, (H, W, 3)); v_al = rng.uniform(-1, 1, (H, W))
    (img * torch.from_numpy(v_img)).sum().add((alpha[..., 0] * torch.from_numpy(v_al)).sum()).backward()

    np_ = [t.detach().numpy() for t in leaves]
    r_l, a_l, last = loops.composite_fwd_loops(*np_, bg.numpy(), W, H, 16, flat.numpy(), offs.numpy())
    assert np.abs(r_l - img.detach().numpy()).max() < 1e-12
    assert np.abs(a_l - alpha[..., 0].detach().numpy()).max() < 1e-12
    assert (last >= 0).sum() > 100
    g = loops.composite_bwd_loops(*np_, bg.numpy(), W, H, 16, flat.numpy(), offs.numpy(), a_l, last, v_img, v_al)
    for ref, leaf in zip(g, leaves):
        got = leaf.grad.numpy()
        assert np.abs(ref - got).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_full_pipeline_gradcheck_fp64():
    """autograd through project + SH + composite vs central finite differences (fp64)."""
    args, vm, Ks, W, H = _small_scene(n=12, W=24, H=16, seed=5, dtype=torch.float64)
    names = ["means", "quats", "scales", "opacities", "colors"]
    leaves = [args[k].clone().requires_grad_(True) for k in names]
    rng = np.random.default_rng(1)
    v = torch.from_numpy(rng.uniform(-1, 1, (1, H, W, 3)))
    va = torch.from_numpy(rng.uniform(-1, 1, (1, H, W, 1)))

    def f(*xs):
        kw = dict(zip(names, xs))
        r, a, _ = orc.rasterization(viewmats=vm, Ks=Ks, width=W, height=H, sh_degree=3,
                                    backgrounds=torch.full((1, 3), 0.25, dtype=torch.float64), **kw)
        return (r * v).sum() + (a * va).sum()

    assert torch.autograd.gradcheck(f, leaves, eps=1e-6, atol=1e-5, rtol=1e-4, nondet_tol=0.0)


def test_sh_rest_grads_zero_above_active_degree():
    """features_rest grads must be exactly zero above the active degree
    (/root/reference/src/training/trainer.py:78-83)."""
    args, vm, Ks, W, H = _small_scene()
    for d in (0, 1, 2):
        shs = args["colors"].clone().requires_grad_(True)
        r, _, _ = orc.rasterization(args["means"], args["quats"], args["scales"], args["opacities"], shs,
                                    vm, Ks, W, H, sh_degree=d, backgrounds=torch.zeros(1, 3))
        r.sum().backward()
        nb = (d + 1) ** 2
        assert float(shs.grad[:, nb:].abs().max()) == 0.0
        assert float(shs.grad[:, :nb].abs().max()) > 0.0


def test_c1_config_forward_plumbing():
    """BASELINE config C1: 10k gaussians, 256x256, SH degree 0, CPU forward through the backend surface."""
    W = H = 256
    sc = scene.make_scene(10_000, seed=0, sh_degree=0)
    be = orc.OracleBackend(scene.make_intrinsics(W, H))
    vp = {"world_view_transform": scene.make_camera(0, 1), "image_width": W, "image_height": H}
    out = be.render_with_depth(sc, vp, bg_color=torch.zeros(3), render_mode="RGB", device="cpu", sh_degree=0)
    assert out["render"].shape == (H, W, 3) and out["alpha"].shape == (H, W, 1)
    assert out["means2d"].shape == (10_000, 2) and out["radii"].shape == (10_000, 2)
    assert out["radii"].dtype == torch.int32
    assert out["visibility_filter"].shape == (10_000, 2)
    assert 0.05 < float(out["alpha"].mean()) < 0.6
    assert torch.isfinite(out["render"]).all()


def test_render_modes_depth():
    args, vm, Ks, W, H = _small_scene()
    kw = dict(viewmats=vm, Ks=Ks, width=W, height=H, sh_degree=3, backgrounds=torch.zeros(1, 3), **args)
    rgb, a, _ = orc.rasterization(render_mode="RGB", **kw)
    rgbd, a2, _ = orc.rasterization(render_mode="RGB+D", **kw)
    rgbed, _, _ = orc.rasterization(render_mode="RGB+ED", **kw)
    d, _, _ = orc.rasterization(render_mode="D", **kw)
    assert rgbd.shape[-1] == 4 and d.shape[-1] == 1
    assert torch.allclose(rgbd[..., :3], rgb) and torch.allclose(a, a2)
    assert torch.allclose(rgbd[..., 3:], d)
    assert torch.allclose(rgbed[..., 3:], d / a.clamp(min=1e-10))
