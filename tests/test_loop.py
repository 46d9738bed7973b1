"""The optimisation LOOP around the training step (gsplat_amd.harness.ViewParallelTrainer.run) against golden vectors
captured by running the reference's own ``GaussianTrainer.train`` (tests/golden/make_loop_golden.py ->
loop_golden.npz; /root/reference/src/training/trainer.py:782-859): per-iteration population, loss / L1 / SSIM / PSNR /
scale_reg, and the six parameter tensors at the end -- step, xyz-gradient accumulation, densify cadence, opacity
reset, learning-rate update and scale clamp in the reference's order (BASELINE config C3 "full Adam loop")."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "loop_golden.npz")
LEAVES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def gold():
    return {k: v for k, v in np.load(GOLDEN).items()}


def _t(a, dev="cpu"):
    return torch.from_numpy(np.asarray(a)).to(dev)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


def _setup(gold, tag, dev):
    """-> model, loader (the reference's stub loader sequence minus the batch its iteration-0 debug render consumed,
    trainer.py:719-722), extent, StepConfig, LoopConfig, first iteration."""
    from gsplat_amd.harness import LoopConfig, SplatModel, StepConfig
    (first, iters, d_int, o_int, sh_int, thr, min_op, clone_r, prune_r, clamp_r, extent, cap) = gold[f"{tag}_config"]
    n = gold[f"{tag}_init_xyz"].shape[0]
    model = SplatModel(n, device=dev).load(**{k: _t(gold[f"{tag}_init_{k}"], dev) for k in LEAVES})
    H, W = gold[f"{tag}_images"].shape[1:3]
    views = [{"image": _t(img, dev), "pose": _t(pose, dev), "width": W, "height": H}
             for img, pose in zip(gold[f"{tag}_images"], gold[f"{tag}_poses"])]
    batches = [[views[j] for j in ids] for ids in gold[f"{tag}_batch_ids"]]

    def loader():
        i = 1
        while True:
            yield batches[i % len(batches)]
            i += 1

    step_cfg = StepConfig(sh_increment_interval=int(sh_int))
    loop_cfg = LoopConfig(iterations=int(iters), densify_interval=int(d_int), opacity_reset_interval=int(o_int),
                          densify_grads_threshold=float(thr), densify_min_opacity=float(min_op),
                          densify_clone_extent_ratio=float(clone_r), densify_prune_extent_ratio=float(prune_r),
                          scale_clamp_ratio=float(clamp_r), max_gaussians=int(cap))
    return model, loader(), float(extent), step_cfg, loop_cfg, int(first)


def _noise_from(gold, tag, dev):
    events = {int(it): i for i, it in enumerate(gold[f"{tag}_noise_iteration"])}

    def split_noise(iteration, n_split):
        noise = _t(gold[f"{tag}_noise_{events[iteration]}"], dev)
        assert noise.shape == (2 * n_split, 3), (iteration, n_split, tuple(noise.shape))
        return noise
    return split_noise


def _run_and_check(gold, tag, dev, backend, fused, tol_metric, tol_param, rank=0, world=1, split_noise=None):
    from gsplat_amd.harness import ViewParallelTrainer
    model, loader, extent, step_cfg, loop_cfg, first = _setup(gold, tag, dev)
    trainer = ViewParallelTrainer(model, backend, config=step_cfg, fused=fused, rank=rank, world_size=world)
    trainer.iteration = first
    trace = []
    final = trainer.run(loader, extent, loop_cfg, split_noise=split_noise,
                        on_step=lambda it, n, m, st: trace.append((it, n, [m[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")], st)))
    assert [t[0] for t in trace] == gold[f"{tag}_trace_iteration"].tolist()
    assert [t[1] for t in trace] == gold[f"{tag}_trace_n"].tolist()               # population before every step
    assert final.n == int(gold[f"{tag}_final_n"]) and trainer.iteration == loop_cfg.iterations
    got, ref = np.asarray([t[2] for t in trace]), gold[f"{tag}_trace_metrics"]
    assert np.all(np.abs(got - ref) <= tol_metric * np.maximum(1.0, np.abs(ref))), np.abs(got - ref).max(axis=0)
    for k in LEAVES:
        assert _rel(getattr(final, k).detach().cpu().numpy(), gold[f"{tag}_final_{k}"]) < tol_param, k
    final.check_bound()
    return trainer, trace


def test_loop_reproduces_reference_train_loop(gold):
    """CPU, oracle backend injected by the test: eight iterations of the reference's loop with the SH degree stepping
    0 -> 3 and two opacity resets (no densify event: the reference only densifies after iteration 5000)."""
    from oracle import OracleBackend
    tr, trace = _run_and_check(gold, "plain", "cpu", OracleBackend(_t(gold["plain_K"])), fused=False, tol_metric=2e-6,
                               tol_param=2e-6)
    assert all(t[3] is None for t in trace)


def test_in_place_reset_and_clamp_keep_the_bucket_views():
    """The reference's reset rebinds ``opacity.data`` (trainer.py:850), which would detach the parameter from the
    flat bucket the fused Adam step and the exchange work on: the counterpart writes in place, and a rebound
    parameter is caught at the start of the next iteration."""
    import math
    from gsplat_amd.harness import SplatModel
    m = SplatModel(5, "cpu")
    with torch.no_grad():
        m.opacity.copy_(torch.tensor([[-9.0], [-4.0], [0.0], [3.0], [40.0]]))
        m.scaling.copy_(torch.linspace(-4, 1, 15).view(5, 3))
    m.reset_opacity(0.01)
    m.clamp_scaling(extent=1.0, scale_clamp_ratio=0.2)
    m.check_bound()
    cur = torch.sigmoid(torch.tensor([[-9.0], [-4.0], [0.0], [3.0], [40.0]]))
    want = torch.logit(torch.minimum(cur, torch.full_like(cur, 0.01)).clamp(1e-6, 1 - 1e-6))
    assert torch.equal(m.opacity.detach(), want) and float(m.scaling.detach().max()) <= math.log(0.2 + 1e-9)
    assert float(m.flat.detach()[51 * 5:52 * 5].max()) == float(want.max())          # the bucket saw the write
    m.opacity.data = m.opacity.data.clone()                                 # what the reference's loop would do
    with pytest.raises(RuntimeError, match="flat parameter bucket"):
        m.check_bound()


def test_densify_accepts_the_reference_loops_keywords():
    """trainer.py:798-807 passes ``max_screen_size`` by keyword."""
    import inspect
    from gsplat_amd.harness import SplatModel
    params = list(inspect.signature(SplatModel.densify_and_prune).parameters)
    assert params[1:5] == ["grads_threshold", "min_opacity", "extent", "max_screen_size"]
    for k in ("optimizer", "clone_extent_ratio", "prune_extent_ratio", "max_gaussians"):
        assert k in params


# ---------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-gaussian-splatting_amd"))
    from oracle import OracleBackend
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gold = {k: v for k, v in np.load(GOLDEN).items()}
    tr, _ = _run_and_check(gold, "plain", "cpu", OracleBackend(_t(gold["plain_K"])), fused=False, tol_metric=2e-6,
                           tol_param=2e-6, rank=rank, world=world)
    mine = tr.model.flat.detach().clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other)
    dist.destroy_process_group()


def test_loop_two_ranks_equal_reference_train_loop():
    """world_size 2 over gloo on CPU: the loop with its batches of 2 views dealt one per rank still reproduces the
    reference's single-process loop, and both ranks hold identical parameters at the end."""
    import torch.multiprocessing as mp
    mp.spawn(_rank_main, args=(2, _free_port()), nprocs=2, join=True)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_loop_on_hip_matches_reference_train_loop(gold, fused):
    """The same eight iterations with the HIP rasterizer underneath (plain autograd path and the fused step)."""
    import gsplat_amd
    _run_and_check(gold, "plain", "cuda", gsplat_amd.HipBackend(_t(gold["plain_K"], "cuda")), fused=fused,
                   tol_metric=3e-5, tol_param=3e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_loop_with_densify_on_hip_matches_reference_train_loop(gold, fused):
    """Iterations 5000..5010 of the reference's loop: densify events at 5004 (clone + split + prune, with the
    reference's recorded split noise) and 5008 (population cap reached), opacity resets at 5000, 5005, 5010 -- the
    population before every step, the metrics of every step and the final parameters of the reference's run."""
    import gsplat_amd
    tr, trace = _run_and_check(gold, "densify", "cuda", gsplat_amd.HipBackend(_t(gold["densify_K"], "cuda")), fused=fused,
                               tol_metric=3e-5, tol_param=3e-4, split_noise=_noise_from(gold, "densify", "cuda"))
    events = [t for t in trace if t[3] is not None]
    assert [t[0] for t in events] == [5004, 5008]
    assert events[0][3]["n_after"] == 330 and events[1][3]["capped"]


@pytest.mark.gpu
def test_default_split_noise_is_identical_on_every_rank():
    """Without injected noise the draws come from a device generator seeded with the iteration: two trainers (two
    ranks) get the same children."""
    from gsplat_amd.harness import SplatModel, ViewParallelTrainer
    a = ViewParallelTrainer(SplatModel(4, "cuda"), backend=None)._default_split_noise(5100, 7)
    b = ViewParallelTrainer(SplatModel(4, "cuda"), backend=None, rank=1, world_size=1)._default_split_noise(5100, 7)
    c = ViewParallelTrainer(SplatModel(4, "cuda"), backend=None)._default_split_noise(5200, 7)
    assert a.shape == (14, 3) and torch.equal(a, b) and not torch.equal(a, c)
