"""Mixed precision, the reference's DEFAULT live path (/root/reference/src/gaussians/training_config.py:63
``use_mixed_precision = True``): ``autocast`` around the render call (trainer.py:468-470) and a ``GradScaler``
(trainer.py:31,505-511).  The rasterizer's inputs are fp32, so it computes in fp32 either way; what changes is that
its backward pass sees an upstream gradient multiplied by the loss scale (65536 at the start), that the gradients are
unscaled before Adam, and that a step with a non-finite gradient is skipped.  The goldens were captured with AMP off
(CPU); with AMP on the same two steps must come out."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "harness_golden.npz")
LEAVES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def gold():
    return {k: v for k, v in np.load(GOLDEN).items()}


def _t(a, dev="cuda"):
    return torch.from_numpy(np.asarray(a)).to(dev)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


def _model_and_batch(gold, dev="cuda"):
    from gsplat_amd.harness import SplatModel
    n = gold["init_xyz"].shape[0]
    model = SplatModel(n, device=dev).load(**{k: _t(gold["init_" + k], dev) for k in LEAVES})
    H, W = gold["targets"].shape[1:3]
    batch = [{"image": _t(img, dev), "pose": _t(pose, dev), "width": W, "height": H}
             for img, pose in zip(gold["targets"], gold["poses"])]
    return model, batch


@pytest.mark.gpu
def test_scaled_upstream_gradient_through_the_hip_backward():
    """What trainer.py:468-511 does around the backend, spelled out: autocast + GradScaler(init_scale=65536) around
    ``HipBackend.render_with_depth``, backward of the scaled loss, ``scaler.step``.  The render is fp32 under
    autocast; after ``unscale_`` the five gradients equal those of the same call without AMP (a power-of-two scale
    is exact in fp32 short of overflow; what remains is the order of the float atomics)."""
    import gsplat_amd
    from gsplat_amd import scene
    W, H, N = 160, 96, 3000
    sc = scene.make_scene(N, seed=5, sh_degree=3, s_min=0.01, s_max=0.1)
    backend = gsplat_amd.HipBackend(scene.make_intrinsics(W, H).cuda())
    viewpoint = {"world_view_transform": scene.make_camera(1, 6).cuda(), "image_width": W, "image_height": H}
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).cuda()
    names = ("means3D", "scales", "rotations", "opacities", "shs")

    def run(amp):
        leaves = {k: sc[k].cuda().clone().requires_grad_(True) for k in names}
        opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0, enabled=amp)
        with torch.autocast(device_type="cuda", enabled=amp):
            out = backend.render_with_depth(leaves, viewpoint, bg_color=torch.zeros(3, device="cuda"), render_mode="RGB",
                                            device="cuda", sh_degree=3)
        assert out["render"].dtype == torch.float32
        loss = (out["render"] - target).abs().mean()
        scaler.scale(loss).backward()
        if amp:   # the backward pass really ran on the scaled gradient
            assert float(leaves["shs"].grad.abs().max()) > 1e3 * 1e-6
        scaler.unscale_(opt)
        grads = {k: v.grad.clone() for k, v in leaves.items()}
        scaler.step(opt)
        scaler.update()
        return grads, {k: v.detach().clone() for k, v in leaves.items()}, scaler.get_scale() if amp else None

    g0, p0, _ = run(False)
    g1, p1, s1 = run(True)
    assert s1 == 65536.0
    for k in names:
        assert _rel(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 2e-5, k
        assert _rel(p1[k].cpu().numpy(), p0[k].cpu().numpy()) < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_training_step_with_mixed_precision_matches_reference_trainer(gold, fused):
    """``StepConfig(use_mixed_precision=True)``: the reference trainer's two golden steps (captured with AMP off)
    within the bounds of the AMP-off HIP tests -- autograd path with torch's GradScaler, fused path with the scale
    folded into ``gsa_view_loss`` / ``gsa_activate_bwd`` and ``gsa_grad_unscale`` before the guarded Adam step."""
    import gsplat_amd
    from gsplat_amd.harness import StepConfig, ViewParallelTrainer
    model, batch = _model_and_batch(gold)
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(_t(gold["K"])), config=StepConfig(use_mixed_precision=True),
                             fused=fused)
    assert tr.amp and tr.loss_scale == 65536.0 and (tr.scaler is None) == fused
    tr.iteration = int(gold["step_iteration"])
    for step in (1, 2):
        metrics = tr.training_step(batch)
        got = np.asarray([metrics[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")])
        ref = gold[f"step{step}_metrics"]
        assert np.all(np.abs(got - ref) <= 2e-5 * np.maximum(1.0, np.abs(ref))), (step, got, ref)
        for k in LEAVES:   # .grad holds the UNSCALED gradient after the step, as with GradScaler.step
            assert _rel(getattr(model, k).grad.cpu().numpy(), gold[f"step{step}_grad_{k}"]) < 1e-4, (step, k)
            assert _rel(getattr(model, k).detach().cpu().numpy(), gold[f"step{step}_param_{k}"]) < 5e-5, (step, k)
    assert tr.skipped_steps == 0 and tr.loss_scale == 65536.0
    st = tr.optimizer.state[model.xyz]
    assert _rel(st["exp_avg"].cpu().numpy(), gold["step2_adam_exp_avg_xyz"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_non_finite_gradient_skips_the_step_and_halves_the_scale(gold, fused):
    """GradScaler's inf-skip path (trainer.py:507-508): a NaN in one target pixel makes the gradient non-finite; the
    step must leave parameters and Adam state untouched, halve the loss scale, and the next (clean) step must be the
    reference's FIRST golden step."""
    import gsplat_amd
    from gsplat_amd.harness import StepConfig, ViewParallelTrainer
    model, batch = _model_and_batch(gold)
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(_t(gold["K"])), config=StepConfig(use_mixed_precision=True),
                             fused=fused)
    tr.iteration = int(gold["step_iteration"])
    before = model.flat.detach().clone()
    poisoned = [dict(v) for v in batch]
    poisoned[1]["image"] = batch[1]["image"].clone()
    poisoned[1]["image"][7, 11, 2] = float("nan")
    tr.training_step(poisoned)
    assert torch.equal(model.flat.detach(), before), "a skipped step must not touch the parameters"
    assert tr.skipped_steps == 1 and tr.loss_scale == 32768.0
    if fused:
        assert tr.optimizer.step_count == 0 and float(tr.optimizer.exp_avg.abs().max()) == 0.0
    else:
        assert len(tr.optimizer.state) == 0 or all(float(s["exp_avg"].abs().max()) == 0 for s in tr.optimizer.state.values())
    metrics = tr.training_step(batch)                      # the scale is now 32768: still the reference's first step
    got = np.asarray([metrics[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")])
    assert np.all(np.abs(got - gold["step1_metrics"]) <= 2e-5 * np.maximum(1.0, np.abs(gold["step1_metrics"])))
    for k in LEAVES:
        assert _rel(getattr(model, k).detach().cpu().numpy(), gold[f"step1_param_{k}"]) < 5e-5, k
    assert tr.skipped_steps == 1 and tr.loss_scale == 32768.0


@pytest.mark.gpu
def test_grad_unscale_and_guarded_adam():
    """gsa_grad_unscale == GradScaler.unscale_ on a flat bucket (in-place scale, non-finite detection incl. a ragged
    tail); gsa_adam_step(skip_if_nonzero) leaves everything untouched when the flag is set."""
    from gsplat_amd import stages as S
    torch.manual_seed(0)
    for n in (4096, 4099, 3):
        g = torch.randn(n, device="cuda") * 65536.0
        ref = g / 65536.0
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        S.grad_unscale(g, 1.0 / 65536.0, flag)
        assert torch.equal(g, ref) and int(flag) == 0
        for bad in (float("inf"), float("nan")):
            g2 = torch.randn(n, device="cuda")
            g2[n - 1] = bad
            flag.zero_()
            S.grad_unscale(g2, 0.5, flag)
            assert int(flag) == 1
    n = 1000
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p0 = p.clone()
    S.adam_step(p, g, m, v, [n], [1e-2], 1, skip_if_nonzero=torch.ones(1, dtype=torch.int32, device="cuda"))
    assert torch.equal(p, p0) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0
    S.adam_step(p, g, m, v, [n], [1e-2], 1, skip_if_nonzero=torch.zeros(1, dtype=torch.int32, device="cuda"))
    assert not torch.equal(p, p0)
