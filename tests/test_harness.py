"""The training-step harness (gsplat_amd.harness) against golden vectors captured from the reference's own
trainer (tests/golden/make_harness_golden.py -> harness_golden.npz), and the view-parallel contract
"N-rank result == 1-rank result"  with world_size 2 over gloo on CPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "harness_golden.npz")
LEAVES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def gold():
    return {k: v for k, v in np.load(GOLDEN).items()}


def _t(a, dev="cpu"):
    return torch.from_numpy(np.asarray(a)).to(dev)


def _model_and_batch(gold, dev="cpu"):
    from gsplat_amd.harness import SplatModel
    n = gold["init_xyz"].shape[0]
    model = SplatModel(n, device=dev).load(**{k: _t(gold["init_" + k], dev) for k in LEAVES})
    H, W = gold["targets"].shape[1:3]
    batch = [{"image": _t(img, dev), "pose": _t(pose, dev), "width": W, "height": H}
             for img, pose in zip(gold["targets"], gold["poses"])]
    return model, batch


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


# ---------------------------------------------------------------------------------------------------------
def test_helper_functions_match_reference(gold):
    from gsplat_amd import harness as hz
    a, b = _t(gold["helper_img_a"]), _t(gold["helper_img_b"])
    ssim = hz.ssim_3x3(a.permute(2, 0, 1)[None], b.permute(2, 0, 1)[None])
    assert abs(float(ssim) - float(gold["ssim"])) < 1e-7
    assert abs(hz.psnr(a.clamp(0, 1), b) - float(gold["psnr"])) < 1e-9
    assert hz.psnr(b, b) == float(gold["psnr_identical"]) == 100.0
    x = a.clone().requires_grad_(True)
    y = hz.straight_through_clamp(x)
    (y * b).sum().backward()
    assert np.array_equal(y.detach().numpy(), gold["stclamp_fwd"])
    assert np.array_equal(x.grad.numpy(), gold["stclamp_grad"])


def test_schedules_match_reference(gold):
    from gsplat_amd.harness import SplatModel, ViewParallelTrainer
    tr = ViewParallelTrainer(SplatModel(4, "cpu"), backend=None)
    for it, degree, lr in gold["schedule"]:
        tr.iteration = int(it)
        assert tr.active_sh_degree() == int(degree)
        assert abs(tr.update_learning_rate() - lr) <= 1e-15 + 1e-12 * lr
        assert tr.optimizer.param_groups[0]["lr"] == tr.update_learning_rate()
    names = [g["name"] for g in tr.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert [g["lr"] for g in tr.optimizer.param_groups][1:] == [0.0025, 0.0025 / 20.0, 0.05, 0.005, 0.001]
    assert all(g["eps"] == 1e-15 for g in tr.optimizer.param_groups)


def test_flat_bucket_layout():
    from gsplat_amd.harness import FLOATS_PER_GAUSSIAN, SplatModel
    m = SplatModel(7, "cpu")
    assert FLOATS_PER_GAUSSIAN == 59 and m.flat.numel() == 59 * 7
    m.features_rest.grad.fill_(2.0)
    assert float(m.flat_grad.sum()) == 2.0 * 7 * 45
    # the SH coefficients are ONE segment of [N,16,3] rows at the front of the bucket (what the rasterizer reads and its
    # backward writes: no cat / split per step); features_dc / features_rest are strided views of those rows
    assert m.sh.data_ptr() == m.flat.data_ptr() == m.features_dc.data_ptr() and m.sh.shape == (7, 16, 3)
    assert m.features_rest.data_ptr() == m.flat.data_ptr() + 12 and m.features_rest.stride() == (48, 3, 1)
    assert m.xyz.data_ptr() == m.flat.data_ptr() + 4 * 48 * 7 and float(m.sh_grad[:, 1:].sum()) == 2.0 * 7 * 45
    m.check_bound()
    assert set(m.activated()) == {"means3D", "scales", "rotations", "opacities", "shs"}
    assert m.activated()["shs"].shape == (7, 16, 3) and m.activated()["opacities"].shape == (7, 1)


def _check_against_golden(gold, model, trainer, batch, tol_metric, tol_grad, tol_param):
    worst = {"grad": 0.0, "param": 0.0}
    for step in (1, 2):
        metrics = trainer.training_step(batch)
        got = np.asarray([metrics[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")])
        ref = gold[f"step{step}_metrics"]
        assert np.all(np.abs(got - ref) <= tol_metric * np.maximum(1.0, np.abs(ref))), (step, got, ref)
        for k in LEAVES:
            eg = _rel(getattr(model, k).grad.cpu().numpy(), gold[f"step{step}_grad_{k}"])
            ep = _rel(getattr(model, k).detach().cpu().numpy(), gold[f"step{step}_param_{k}"])
            worst["grad"], worst["param"] = max(worst["grad"], eg), max(worst["param"], ep)
            assert eg < tol_grad, (step, k, eg)
            assert ep < tol_param, (step, k, ep)
    print("harness vs reference goldens, worst relative error:", worst)
    st = trainer.optimizer.state[model.xyz]
    assert _rel(st["exp_avg"].cpu().numpy(), gold["step2_adam_exp_avg_xyz"]) < tol_grad
    assert _rel(st["exp_avg_sq"].cpu().numpy(), gold["step2_adam_exp_avg_sq_xyz"]) < 2 * tol_grad


def test_training_step_reproduces_reference_trainer(gold):
    """Two steps of this harness (oracle backend injected by the test) == two steps of the reference's
    GaussianTrainer._training_step on the same inputs: loss, L1, SSIM, PSNR, scale_reg, all six gradients,
    all six parameters after Adam, Adam moments."""
    from gsplat_amd.harness import ViewParallelTrainer
    from oracle import OracleBackend
    model, batch = _model_and_batch(gold)
    trainer = ViewParallelTrainer(model, OracleBackend(_t(gold["K"])))
    trainer.iteration = int(gold["step_iteration"])
    assert abs(float(gold["scale_reg"]) - gold["step1_metrics"][4]) < 1e-9
    _check_against_golden(gold, model, trainer, batch, tol_metric=2e-6, tol_grad=2e-5, tol_param=1e-6)


# ---------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_path):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-gaussian-splatting_amd"))
    from gsplat_amd.harness import ViewParallelTrainer
    from oracle import OracleBackend
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gold = {k: v for k, v in np.load(GOLDEN).items()}
    model, batch = _model_and_batch(gold)
    trainer = ViewParallelTrainer(model, OracleBackend(_t(gold["K"])), rank=rank, world_size=world)
    trainer.iteration = int(gold["step_iteration"])
    metrics = [trainer.training_step(batch) for _ in range(2)]
    if rank == 0:
        np.savez(out_path, flat=model.flat.detach().numpy(), grad=model.flat_grad.numpy(),
                 metrics=np.asarray([[m[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")] for m in metrics]))
    # every rank must end with identical parameters
    mine = model.flat.detach().clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other)
    dist.destroy_process_group()


def test_view_parallel_two_ranks_equal_one_rank(gold, tmp_path):
    """world_size 2 over gloo: batch of 3 views dealt (2 + 1), one all-reduce of the 59N bucket; parameters and
    metrics after two steps match the single-process run (and therefore the reference trainer)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    from gsplat_amd.harness import pack_bucket      # (the bucket's own order: coefficient rows first)
    n = gold["step2_param_xyz"].shape[0]
    ref_flat = pack_bucket(lambda k: torch.from_numpy(gold[f"step2_param_{k}"]), n).numpy()
    ref_grad = pack_bucket(lambda k: torch.from_numpy(gold[f"step2_grad_{k}"]), n).numpy()
    assert _rel(got["flat"], ref_flat) < 1e-6
    assert _rel(got["grad"], ref_grad) < 2e-5
    ref_m = np.stack([gold["step1_metrics"], gold["step2_metrics"]])
    assert np.all(np.abs(got["metrics"] - ref_m) <= 2e-6 * np.maximum(1.0, np.abs(ref_m)))


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_training_step_on_hip_backend_matches_reference_trainer(gold):
    """Same two steps with the HIP rasterizer underneath (fp32 kernels vs the torch oracle: measured 1.6e-5 on the gradients, 1.1e-5 on the parameters)."""
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    model, batch = _model_and_batch(gold, "cuda")
    trainer = ViewParallelTrainer(model, gsplat_amd.HipBackend(_t(gold["K"], "cuda")))
    trainer.iteration = int(gold["step_iteration"])
    _check_against_golden(gold, model, trainer, batch, tol_metric=2e-5, tol_grad=1e-4, tol_param=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,factored", [("alternate", True), (True, True), (False, True), (True, False), ("alternate", False)])
def test_fused_training_step_matches_reference_trainer(gold, overlap, factored):
    """The same two steps on HIP kernels end to end -- rasterizer stages called directly, activations
    (gsa_activate_fwd / _bwd), loss (gsa_view_loss), SH-gradient rebuild (gsa_sh_grad_expand_split) and Adam
    (gsa_adam_step): still the reference trainer's metrics, gradients, parameters and Adam moments.  With whole views on
    two alternating streams (the default), with only the next view's front end on a side stream, on one stream, and with
    the plain 48-float SH-gradient accumulation."""
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    model, batch = _model_and_batch(gold, "cuda")
    trainer = ViewParallelTrainer(model, gsplat_amd.HipBackend(_t(gold["K"], "cuda")), fused=True,
                                  overlap_views=overlap, factored_exchange=factored)
    trainer.iteration = int(gold["step_iteration"])
    _check_against_golden(gold, model, trainer, batch, tol_metric=2e-5, tol_grad=1e-4, tol_param=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 4099])
def test_activation_kernels_match_torch_autograd(n):
    """gsa_activate_fwd / gsa_activate_bwd vs the torch formulation of the reference's activations and scale-ratio
    regulariser (gaussian_model.py:86-108, trainer.py:434-445) differentiated by autograd in fp64: values, the three
    raw-parameter gradients with and without the regulariser's share, and its value.  Covers tied largest / smallest
    scales (gradient shared evenly), scales below the 1e-8 floor, ratios on both sides of the hinge, and a bucket
    whose rotation segment is not 16-byte aligned (odd n)."""
    import torch.nn.functional as F
    from gsplat_amd import stages as S
    from gsplat_amd.harness import SplatModel, scale_ratio_reg
    g = torch.Generator().manual_seed(n)
    model = SplatModel(n, device="cuda")
    with torch.no_grad():
        model.scaling.copy_((torch.rand(n, 3, generator=g) * 6 - 5).cuda())
        model.scaling[:50, 1] = model.scaling[:50, 0]                     # tied pairs
        model.scaling[50:80] = model.scaling[50:80, :1]                   # all three equal: ratio 1
        model.scaling[80:100, 2] = -21.0                                  # exp() below the 1e-8 floor
        model.rotation.copy_(torch.randn(n, 4, generator=g).cuda())
        model.opacity.copy_((torch.randn(n, 1, generator=g) * 3).cuda())
        model.features_dc.copy_(torch.randn(n, 1, 3, generator=g).cuda())
        model.features_rest.copy_(torch.randn(n, 15, 3, generator=g).cuda())
    scales, quats, opac, shs = S.activate_fwd(model.scaling, model.rotation, model.opacity, model.features_dc,
                                              model.features_rest)
    assert torch.allclose(scales, torch.exp(model.scaling.detach()), rtol=2e-6, atol=0)
    assert torch.allclose(quats, F.normalize(model.rotation.detach(), dim=-1), rtol=0, atol=2e-7)
    assert torch.allclose(opac, torch.sigmoid(model.opacity.detach())[:, 0], rtol=0, atol=2e-7)
    assert torch.equal(shs, torch.cat([model.features_dc, model.features_rest], dim=1).detach())
    assert torch.equal(shs, model.sh)                  # (the model's own rows: what the fused step hands to the rasterizer)
    # (both feature tensors are STRIDED views here, so the binding makes two contiguous temporaries for one launch: they
    # must both outlive the evaluation of the argument list -- r04 found the second taking the first's block)
    s2, q2, o2, none = S.activate_fwd(model.scaling, model.rotation, model.opacity)       # geometry only
    assert none is None and torch.equal(s2, scales) and torch.equal(q2, quats) and torch.equal(o2, opac)
    gs, gq, go = (torch.randn(n, k, generator=g).cuda() for k in (3, 4, 1))
    max_ratio, weight = 10.0, 0.1
    for with_reg in (True, False):
        raw = [t.detach().double().requires_grad_(True) for t in (model.scaling, model.rotation, model.opacity)]
        sc = torch.exp(raw[0])
        reg = scale_ratio_reg(sc, max_ratio, weight)
        total = (sc * gs).sum() + (F.normalize(raw[1], dim=-1) * gq).sum() + (torch.sigmoid(raw[2]) * go).sum()
        (total + reg if with_reg else total).backward()
        out = [torch.full_like(t, float("nan")) for t in (model.scaling, model.rotation, model.opacity)]
        val = S.activate_bwd(model.scaling, model.rotation, model.opacity, gs, gq, go[:, 0].contiguous(), out[0], out[1],
                             out[2], max_ratio, weight, add_reg_grad=with_reg)
        assert abs(float(val) - float(reg)) <= 1e-6 * max(1.0, abs(float(reg)))
        for name, a, b in zip(("scaling", "rotation", "opacity"), out, raw):
            assert _rel(a.double().cpu().numpy(), b.grad.cpu().numpy()) < 2e-6, (with_reg, name)


def _fused_rank_main(rank, world, port, factored):
    """One rank of the fused HIP trainer; the ranks share the box's one GPU and talk over gloo (the collectives'
    call pattern is the RCCL one; tests/ never launches more ranks on the card than this)."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-gaussian-splatting_amd"))
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gold = {k: v for k, v in np.load(GOLDEN).items()}
    model, batch = _model_and_batch(gold, "cuda")
    trainer = ViewParallelTrainer(model, gsplat_amd.HipBackend(_t(gold["K"], "cuda")), rank=rank, world_size=world,
                                  fused=True, factored_exchange=factored)
    assert trainer.factored == factored
    trainer.iteration = int(gold["step_iteration"])
    _check_against_golden(gold, model, trainer, batch, tol_metric=2e-5, tol_grad=1e-4, tol_param=5e-5)
    mine = model.flat.detach().clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    if factored:   # the rebuilt SH gradient is summed in batch order on every rank: parameters agree to the bit there
        nf = model.n
        assert torch.equal(mine[3 * nf:51 * nf], other[3 * nf:51 * nf])
    assert _rel(mine.cpu().numpy(), other.cpu().numpy()) < 1e-6
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("factored", [True, False])
def test_fused_view_parallel_two_ranks_match_reference_trainer(gold, factored):
    """world_size 2 with the fused HIP step (3 views dealt 2 + 1): the factored exchange (all-gather of d loss/d rgb,
    all-reduce of the 11 N geometry floats, SH gradient rebuilt locally) and the plain 59 N all-reduce both give the
    reference trainer's metrics, gradients and parameters after two steps."""
    import torch.multiprocessing as mp
    mp.spawn(_fused_rank_main, args=(2, _free_port(), factored), nprocs=2, join=True)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(32, 48), (67, 130), (1, 5), (95, 250)])
def test_view_loss_kernel_matches_torch_formulation(gold, shape):
    """gsa_view_loss vs autograd through the harness's torch formulation of the reference's loss (trainer.py:485-492):
    L1, SSIM, mse and d loss / d render; values outside [0, 1] exercise the straight-through clamp.  95 x 250 has strips
    that lie wholly inside the image (the kernel's border-free variant) next to border strips and a partial last band."""
    from gsplat_amd import stages as S
    from gsplat_amd.harness import view_loss
    h, w = shape
    g = torch.Generator().manual_seed(h * 1000 + w)
    render = (torch.rand(h, w, 3, generator=g) * 1.4 - 0.2).cuda()
    target = torch.rand(h, w, 3, generator=g).cuda()
    lam, scale = 0.2, 0.25
    sums, v = S.view_loss(render, target, lam, grad_scale=scale)
    r64 = render.double().requires_grad_(True)
    loss, l1, ssim = view_loss(r64, target.double(), lam)
    (loss * scale).backward()
    n = render.numel()
    assert abs(float(sums[0]) / n - float(l1.detach())) < 1e-6 and abs(float(sums[1]) / n - float(ssim.detach())) < 1e-5
    mse = float(((render.clamp(0, 1) - target) ** 2).double().mean())
    assert abs(float(sums[2]) / n - mse) < 1e-7
    assert _rel(v.cpu().numpy(), r64.grad.float().cpu().numpy()) < 2e-4
    # pinned transitively on the reference: view_loss / ssim_3x3 reproduce the reference's own SSIM, PSNR and
    # straight-through-clamp goldens in test_helper_functions_match_reference


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    """gsa_adam_step over a flat bucket with per-segment learning rates == torch.optim.Adam with the same groups,
    five steps, including a ragged tail (length not a multiple of 4) and zero gradients."""
    from gsplat_amd import stages as S
    torch.manual_seed(0)
    sizes, lrs = [37, 128, 1003, 2], [1.6e-4, 2.5e-3, 0.05, 1e-3]
    n = sum(sizes)
    flat = torch.randn(n, device="cuda")
    ref_params = [p.clone().requires_grad_(True) for p in flat.split(sizes)]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_params, lrs)], lr=0.0, eps=1e-15)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    ends = list(np.cumsum(sizes))
    for step in range(1, 6):
        g = torch.randn(n, device="cuda") * (10.0 ** torch.randint(-6, 2, (n,), device="cuda").float())
        g[::7] = 0.0
        for p, gp in zip(ref_params, g.split(sizes)):
            p.grad = gp.clone()
        opt.step()
        S.adam_step(flat, g, m, v, ends, lrs, step)
        ref = torch.cat([p.detach() for p in ref_params])
        assert float((flat - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    st = opt.state[ref_params[2]]
    # moments: same update up to how (g - m) * 0.1 is fused; an element's error is bounded relative to the tensor
    assert _rel(m.split(sizes)[2].cpu().numpy(), st["exp_avg"].cpu().numpy()) < 1e-6
    assert _rel(v.split(sizes)[2].cpu().numpy(), st["exp_avg_sq"].cpu().numpy()) < 1e-6


DENSIFY_GOLDEN = os.path.join(HERE, "golden", "densify_golden.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["free", "capped"])
@pytest.mark.parametrize("opt_kind", ["fused", "torch"])
def test_densify_and_prune_matches_reference_model(case, opt_kind):
    """SplatModel.densify_and_prune (one HIP gather pass over the 59 N bucket and both Adam moment buckets) ==
    the reference's GaussianModel.densify_and_prune with its optimizer-state splicing (gaussian_model.py:9-50,110-366)
    on the same inputs and the same split noise: statistics, all six parameter tensors, all twelve Adam moments.
    'capped' exercises the max_gaussians top-k path."""
    from gsplat_amd.harness import FIELDS, FusedBucketAdam, SplatModel, _adam_groups, StepConfig, pack_bucket
    g = {k: v for k, v in np.load(DENSIFY_GOLDEN).items()}
    names = [n for n, _ in FIELDS]
    n = g[f"{case}_in_xyz"].shape[0]
    model = SplatModel(n, "cuda").load(**{k: _t(g[f"{case}_in_{k}"], "cuda") for k in names})
    model.xyz_gradient_accum = _t(g[f"{case}_in_accum"], "cuda")
    model.xyz_gradient_count = _t(g[f"{case}_in_count"], "cuda")
    groups = _adam_groups(model, StepConfig())
    if opt_kind == "fused":
        opt = FusedBucketAdam(model, groups)
        opt.exp_avg.copy_(pack_bucket(lambda k: _t(g[f"{case}_in_m_{k}"], "cuda"), n))
        opt.exp_avg_sq.copy_(pack_bucket(lambda k: _t(g[f"{case}_in_v_{k}"], "cuda"), n))
    else:
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for k in names:
            opt.state[getattr(model, k)] = {"step": torch.tensor(2.0), "exp_avg": _t(g[f"{case}_in_m_{k}"], "cuda").clone(),
                                            "exp_avg_sq": _t(g[f"{case}_in_v_{k}"], "cuda").clone()}
    thr, min_op, extent, clone_r, prune_r, cap = g[f"{case}_args"]
    stats = model.densify_and_prune(float(thr), float(min_op), float(extent), optimizer=opt,
                                    clone_extent_ratio=float(clone_r), prune_extent_ratio=float(prune_r),
                                    max_gaussians=None if cap < 0 else int(cap), noise=_t(g[f"{case}_noise"], "cuda"))
    got = [stats[k] for k in ("cloned", "split", "pruned", "pruned_low_op", "pruned_too_big", "n_before", "n_after")]
    assert got + [int(stats["capped"])] == g[f"{case}_stats"].tolist()
    assert model.n == stats["n_after"] and float(model.xyz_gradient_accum.abs().max()) == 0
    for k in names:
        p = getattr(model, k)
        assert p.requires_grad and p.grad is not None and p.grad.shape == p.shape
        assert tuple(p.shape) == g[f"{case}_out_{k}"].shape, k
        assert _rel(p.detach().cpu().numpy(), g[f"{case}_out_{k}"]) < 1e-6, k
        st = opt.state[p]
        assert np.array_equal(st["exp_avg"].cpu().numpy(), g[f"{case}_out_m_{k}"]), k
        assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), g[f"{case}_out_v_{k}"]), k
    assert [grp["params"][0] is getattr(model, k) for grp, k in zip(opt.param_groups, names)] == [True] * 6
    opt.step() if opt_kind == "torch" else opt.step()   # the spliced optimizer still steps


@pytest.mark.gpu
def test_psnr_parity_of_the_optimisation_loop(gold):
    """BASELINE metric, second half ("PSNR parity vs ref"), at reduced size (>= 200 steps): 200
    optimisation steps of the harness with the HIP rasterizer (plain and fused paths, on the GPU) track the same 200
    steps with the CPU oracle underneath (which reproduces the reference trainer,
    test_training_step_reproduces_reference_trainer) while the PSNR climbs from 17 to 31 dB: within 0.02 dB at every
    one of the first 100 steps; over all 200 within 0.25 dB (the loop amplifies rounding, and the order of the float
    atomics differs from run to run: 0.003 ... 0.09 dB measured); final loss and positions agree."""
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    from oracle import OracleBackend
    steps = 200

    def run(dev, backend_cls, fused):
        model, batch = _model_and_batch(gold, dev)
        tr = ViewParallelTrainer(model, backend_cls(_t(gold["K"], dev)), fused=fused)
        tr.iteration = int(gold["step_iteration"])
        hist = []
        for _ in range(steps):
            tr.update_learning_rate()
            hist.append(tr.training_step(batch))
            tr.iteration += 1
        return model, hist

    torch.set_num_threads(8)
    m_ref, h_ref = run("cpu", OracleBackend, False)
    psnr_ref = np.asarray([h["psnr"] for h in h_ref])
    assert h_ref[-1]["loss"] < h_ref[0]["loss"] and psnr_ref[-1] > psnr_ref[0] + 10.0    # the loop does optimise
    for fused in (False, True):
        m, h = run("cuda", gsplat_amd.HipBackend, fused)
        drift = np.abs(np.asarray([x["psnr"] for x in h]) - psnr_ref)
        assert drift[:100].max() < 0.02, (fused, drift[:100].max())
        assert drift.max() < 0.25, (fused, drift.max())
        assert abs(h[-1]["loss"] - h_ref[-1]["loss"]) < 2e-2 * abs(h_ref[-1]["loss"])
        assert _rel(m.xyz.detach().cpu().numpy(), m_ref.xyz.detach().numpy()) < 1e-2
