"""BASELINE configs C4 and C5 at their real sizes, where the CPU oracle is out of reach: size-independent properties,
checksums against a second decomposition of the same work, and bit-exact data movement.

* C4 shape (1M gaussians, 1080p, 8-view batch, view-parallel): the gradient bucket and the parameters after ONE fused
  step over 8 ring views in one process == the same step on 2 ranks x 4 views (gloo, the ranks share the box's GPU),
  with the factored exchange and with the plain 59 N all-reduce.
* C5 shape (5M gaussians, 3840x2160, degree 3): the properties of test_full_size_properties_c3 (reproducible forward,
  linear backward, finite gradients, sorted bins); one densify / prune event at 5M against torch index operations
  (/root/reference/src/gaussians/gaussian_model.py:110-231: parameters and both Adam moment buckets, bit for bit
  where only data moves); the fused Adam step at 1M x 59 floats against torch.optim.Adam."""
import os
import socket
import sys

import numpy as np
import pytest
import math

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu
FIELDS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---------------------------------------------------------------------------------------------------------
# C4
# ---------------------------------------------------------------------------------------------------------
C4_N, C4_W, C4_H, C4_VIEWS = 1_000_000, 1920, 1080, 8


def _c4_model_and_batch(dev="cuda"):
    """The bench scene as raw (pre-activation) parameters of the reference model, and 8 ring views with seeded
    targets."""
    from gsplat_amd import scene
    from gsplat_amd.harness import SplatModel
    sc = scene.make_scene(C4_N, seed=0, sh_degree=3)
    model = SplatModel(C4_N, device=dev)
    op = sc["opacities"].clamp(1e-4, 1 - 1e-4)
    model.load(xyz=sc["means3D"].to(dev), scaling=torch.log(sc["scales"]).to(dev), rotation=sc["rotations"].to(dev),
               opacity=torch.log(op / (1 - op)).to(dev), features_dc=sc["shs"][:, :1].contiguous().to(dev),
               features_rest=sc["shs"][:, 1:].contiguous().to(dev))
    g = torch.Generator().manual_seed(77)
    batch = [{"image": torch.rand(C4_H, C4_W, 3, generator=g).to(dev), "pose": scene.make_camera(v, C4_VIEWS).to(dev),
              "width": C4_W, "height": C4_H} for v in range(C4_VIEWS)]
    return model, batch, scene.make_intrinsics(C4_W, C4_H).to(dev)


def _digest(t: torch.Tensor):
    """Checksums + a strided sample of a large tensor (what the ranks hand back)."""
    t = t.detach()
    return {"sum": float(t.double().sum()), "abs": float(t.double().abs().sum()), "sample": t[::9973].double().cpu().numpy()}


def _c4_rank_main(rank, world, port, factored, out_path):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-gaussian-splatting_amd"))
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    model, batch, K = _c4_model_and_batch()
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), rank=rank, world_size=world, fused=True,
                             factored_exchange=factored)
    tr.iteration = 3000
    metrics = tr.training_step(batch)
    mine = model.flat.detach().clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert _rel(mine, other) < 1e-6                      # identical parameters on every rank
    if factored:                                          # the rebuilt SH gradient is summed in batch order everywhere
        assert torch.equal(mine[3 * C4_N:51 * C4_N], other[3 * C4_N:51 * C4_N])
    if rank == 0:
        torch.save({"grad": _digest(model.flat_grad), "flat": _digest(model.flat), "metrics": metrics}, out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("factored", [True, False])
def test_c4_eight_views_one_process_equals_two_ranks_of_four(tmp_path, factored):
    import torch.multiprocessing as mp
    import gsplat_amd
    from gsplat_amd.harness import ViewParallelTrainer
    model, batch, K = _c4_model_and_batch()
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), fused=True, factored_exchange=factored)
    tr.iteration = 3000
    before = model.flat.detach().clone()
    metrics = tr.training_step(batch)                     # 8 backward passes accumulated into ONE bucket, then Adam
    one = {"grad": _digest(model.flat_grad), "flat": _digest(model.flat)}
    assert torch.isfinite(model.flat_grad).all() and not torch.equal(model.flat.detach(), before)
    assert 0.0 < metrics["loss"] < 1.0 and 0.0 < metrics["psnr"] < 30.0
    del tr, model, before
    torch.cuda.empty_cache()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_c4_rank_main, args=(2, _free_port(), factored, out), nprocs=2, join=True)
    two = torch.load(out, weights_only=False)
    for what, tol in (("grad", 2e-4), ("flat", 1e-6)):    # float atomics reorder the per-view sums
        a, b = one[what], two[what]
        scale = max(abs(b["abs"]), 1e-30)
        assert abs(a["sum"] - b["sum"]) <= tol * scale and abs(a["abs"] - b["abs"]) <= tol * scale, what
        assert np.abs(a["sample"] - b["sample"]).max() <= tol * max(np.abs(b["sample"]).max(), 1e-30), what
    for k in ("loss", "l1", "ssim", "psnr"):
        assert abs(metrics[k] - two["metrics"][k]) <= 1e-5 * max(1.0, abs(metrics[k])), k


# ---------------------------------------------------------------------------------------------------------
# C5
# ---------------------------------------------------------------------------------------------------------
def test_c5_full_size_properties_5m_4k():
    import gsplat_amd as gsa
    from gsplat_amd import scene
    N, W, H = 5_000_000, 3840, 2160
    sc = scene.make_scene(N, seed=0, sh_degree=3)
    vm, K = scene.make_camera(0, 1)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()
    v = scene.make_v_render(W, H, seed=2).cuda()[None]
    base = [sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda(), sc["opacities"][:, 0].cuda(), sc["shs"].cuda()]
    del sc

    def run(upstream):
        leaves = [t.clone().requires_grad_(True) for t in base]
        r, a, meta = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], vm, K, W, H, sh_degree=3,
                                       backgrounds=torch.zeros(1, 3, device="cuda"))
        r.backward(upstream)
        return r.detach(), a.detach(), meta, [t.grad for t in leaves]

    r0, a0, m0, g1 = run(v)
    r1, a1, m1, g2 = run(2.0 * v)                        # (steady state: remembered capacity and depth-key bits)
    assert torch.equal(r0, r1) and torch.equal(a0, a1)   # reproducible forward
    assert m0["n_isects"] == m1["n_isects"] > 8 * N
    for name, x, y in zip(("means", "quats", "scales", "opacities", "colors"), g1, g2):
        assert torch.isfinite(x).all() and torch.isfinite(y).all(), name
        assert _rel(2.0 * x, y) < 1e-4, name             # linear backward (atomics reorder the sums)
    assert float(a0.min()) >= 0.0 and float(a0.max()) <= 1.0
    del g1, g2, r1, a1
    # sorted bins: tile ids ascending, depth ascending inside a tile, offsets = first record of every tile
    off = m0["isect_offsets"].reshape(-1).long()
    gid = m0["flatten_ids"].long()
    depth = m0["depths"][0][gid]
    tile_of = torch.repeat_interleave(torch.arange(off.numel(), device="cuda"),
                                      torch.diff(torch.cat([off, off.new_tensor([gid.numel()])])))
    assert tile_of.numel() == gid.numel()
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((depth[1:][same_tile] >= depth[:-1][same_tile]).all())
    # every record's gaussian really covers its tile (tile rectangle of mean2d +- radius)
    tw = (W + 15) // 16
    sample = torch.randint(0, gid.numel(), (200_000,), device="cuda")
    g_s, t_s = gid[sample], tile_of[sample]
    m2d, rad = m0["means2d"][0][g_s], m0["radii"][0][g_s].float()
    tx, ty = (t_s % tw).float(), (t_s // tw).float()
    assert bool(((m2d[:, 0] + rad[:, 0]) / 16 > tx).all() and ((m2d[:, 0] - rad[:, 0]) / 16 < tx + 1).all())
    assert bool(((m2d[:, 1] + rad[:, 1]) / 16 > ty).all() and ((m2d[:, 1] - rad[:, 1]) / 16 < ty + 1).all())


def test_8k_frame_bins_32_equal_tiles_16():
    """7680x4320: 129600 tiles of 16 x 16 (beyond the fused binning's 16-bit ids: emit + 32-bit sort path) against
    32400 bins of 32 x 32 (fused path, what rasterization() picks) -- the same image, alpha and gradients."""
    import gsplat_amd as gsa
    from gsplat_amd import scene, stages as S
    N, W, H = 20_000, 7680, 4320
    sc = scene.make_scene(N, seed=4, sh_degree=3)
    vm, K = scene.make_camera(0, 1)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()
    base = [sc["means3D"].cuda(), sc["rotations"].cuda(), (sc["scales"] * 3.0).cuda(), sc["opacities"][:, 0].cuda(),
            sc["shs"].cuda()]
    v = scene.make_v_render(W, H, seed=2).cuda()[None]
    assert S.bin_tile_for(W, H, 0) == 32 and not gsa.lib.gsa_tile_bin16_supported(W, H)

    def run(bin_tile):
        was, S.DEFAULT_BIN_TILE = S.DEFAULT_BIN_TILE, bin_tile
        try:
            leaves = [t.clone().requires_grad_(True) for t in base]
            r, a, meta = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], vm, K, W, H, sh_degree=3)
            r.backward(v)
            return r.detach(), a.detach(), [t.grad for t in leaves], meta
        finally:
            S.DEFAULT_BIN_TILE = was

    r32, a32, g32, m32 = run(32)
    r16, a16, g16, m16 = run(16)
    assert m32["bin_tile"] == 32 and m16["bin_tile"] == 16 and m32["n_isects"] == m16["n_isects"] > 0
    assert torch.equal(r32, r16) and torch.equal(a32, a16)
    for name, x, y in zip(("means", "quats", "scales", "opacities", "colors"), g32, g16):
        assert _rel(x, y) < 5e-5, name      # (atomics reorder the sums; an 8K frame has gaussians on the hot-replica path: 1.3e-5 seen)
    assert torch.equal(m32["flatten_ids"], m16["flatten_ids"])


def test_c5_densify_and_prune_event_at_5m():
    """One densify / prune event at the C5 population: the HIP gather pass (parameters + both Adam moment buckets)
    against the same event written with torch index operations in the reference's order -- [rows that are not split]
    + [clones] + [first children] + [second children], then the prune mask (gaussian_model.py:110-231)."""
    from gsplat_amd.harness import FIELDS as HF, FusedBucketAdam, SplatModel, StepConfig, _adam_groups
    N = 5_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.rand(*s, generator=g, device="cuda")  # noqa: E731
    model = SplatModel(N, "cuda")
    model.load(xyz=rnd(N, 3) * 2 - 1, features_dc=rnd(N, 1, 3), features_rest=rnd(N, 15, 3) * 0.1,
               opacity=rnd(N, 1) * 10 - 7, scaling=torch.log(rnd(N, 3) * 0.25 + 1e-3), rotation=rnd(N, 4) - 0.5)
    model.xyz_gradient_accum = (rnd(N, 1) * 2e-3).expand(N, 3).contiguous()
    model.xyz_gradient_count = torch.ones(N, 1, device="cuda")
    opt = FusedBucketAdam(model, _adam_groups(model, StepConfig()))
    opt.exp_avg.copy_(rnd(59 * N) - 0.5)
    opt.exp_avg_sq.copy_(rnd(59 * N))
    old = {k: getattr(model, k).detach().clone() for k in FIELDS}
    old_m = {k: opt.state[getattr(model, k)]["exp_avg"].clone() for k in FIELDS}
    old_v = {k: opt.state[getattr(model, k)]["exp_avg_sq"].clone() for k in FIELDS}
    thr, min_op, extent = 1e-3, 0.005, 1.0
    grads = torch.norm(model.xyz_gradient_accum / (model.xyz_gradient_count + 1e-8), dim=-1)
    max_scale = torch.exp(old["scaling"]).amax(-1)
    opac = torch.sigmoid(old["opacity"]).squeeze(-1)
    hot = (grads >= thr) & (opac > min_op)
    clone, split = hot & (max_scale <= 0.1 * extent), hot & (max_scale > 0.1 * extent)
    n_split = int(split.sum())
    noise = torch.randn(2 * n_split, 3, generator=g, device="cuda")
    stats = model.densify_and_prune(thr, min_op, extent, max_screen_size=12.0, optimizer=opt, noise=noise)
    assert stats["cloned"] == int(clone.sum()) > 100_000 and stats["split"] == n_split > 100_000

    log_split = torch.log(torch.tensor(1.6)).cuda()      # fp32, computed on the host as the harness does

    def event(field, src, zero_new):
        keep, cl, sp = src[~split], src[clone], src[split]
        if zero_new:
            cl, sp = torch.zeros_like(cl), torch.zeros_like(sp)
        kids = torch.cat([sp, sp])
        if field == "scaling" and not zero_new:
            kids = kids - log_split
        if field == "xyz" and not zero_new:
            kid_scale = torch.exp(torch.cat([old["scaling"][split]] * 2) - log_split)
            kids = kids + noise * kid_scale
        return torch.cat([keep, cl, kids])

    pop_scale = event("scaling", old["scaling"], False)
    pop_op = torch.sigmoid(event("opacity", old["opacity"], False)).squeeze(-1)
    prune = (pop_op < min_op) | (torch.exp(pop_scale).amax(-1) > 2.0 * extent)
    assert stats["pruned"] == int(prune.sum()) > 10_000 and model.n == stats["n_after"] == int((~prune).sum())
    for k in FIELDS:
        want = event(k, old[k], False)[~prune]
        got = getattr(model, k).detach()
        if k == "xyz":    # the children's positions are computed (exp + fma) by the kernel
            assert _rel(got, want) < 1e-6, k
        else:
            assert torch.equal(got, want), k
        assert torch.equal(opt.state[getattr(model, k)]["exp_avg"], event(k, old_m[k], True)[~prune]), k
        assert torch.equal(opt.state[getattr(model, k)]["exp_avg_sq"], event(k, old_v[k], True)[~prune]), k
    model.check_bound()
    opt.step()                                            # the spliced optimizer still steps at the new size
    assert [name for name, _ in HF] == list(FIELDS)


def test_4k_frame_takes_64_pixel_bins_and_the_one_pass_placement():
    """3840x2160: the 8160 bins of 32 x 32 are beyond the one-pass placement (2047), the 2040 bins of 64 x 64 are not, and
    ``bin_tile_for`` picks those for frames that large (C5: binning 0.79 -> 0.54 ms; 2560x1440 measured no better at 64) -- the same image, alpha and record count as with 32 x 32 bins,
    gradients up to the order of the float atomics."""
    import gsplat_amd as gsa
    from gsplat_amd import scene, stages as S
    assert [S.bin_tile_for(w, h, 0) for w, h in ((1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320), (640, 480))] == \
        [32, 32, 64, 32, 32]
    assert S.bin_tile_for(3840, 2160, 1) == 16                       # (exact culling: the unfused path)
    N, W, H = 60_000, 3840, 2160
    assert gsa.lib.gsa_tile_bin16_direct(W, H, 64) == 1 and gsa.lib.gsa_tile_bin16_direct(W, H, 32) == 0
    sc = scene.make_scene(N, seed=6, sh_degree=3)
    vm, K = scene.make_camera(1, 3)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()
    base = [sc["means3D"].cuda(), sc["rotations"].cuda(), (sc["scales"] * 2.0).cuda(), sc["opacities"][:, 0].cuda(),
            sc["shs"].cuda()]
    v = scene.make_v_render(W, H, seed=2).cuda()[None]

    def run(force):
        was = S._BIN_TILE_ENV
        S._BIN_TILE_ENV = force            # (None: the automatic choice; "32": as if GSPLAT_AMD_BIN_TILE=32 were set)
        try:
            leaves = [t.clone().requires_grad_(True) for t in base]
            r, a, meta = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], vm, K, W, H, sh_degree=3)
            r.backward(v)
            return r.detach(), a.detach(), [t.grad for t in leaves], meta
        finally:
            S._BIN_TILE_ENV = was

    r64, a64, g64, m64 = run(None)
    r32, a32, g32, m32 = run("32")
    assert m64["bin_tile"] == 64 and m32["bin_tile"] == 32 and m64["n_isects"] == m32["n_isects"] > 0
    assert m64["n_binned"] < m32["n_binned"]
    assert torch.equal(r64, r32) and torch.equal(a64, a32)
    for x, y in zip(g64, g32):
        assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-30)


def _bench_bucket(tmp_path, tag, gpus, views_per_gpu, streams, exchange="factored", extra=()):
    """Runs bench.py (the N > 1 ranks over gloo, sharing this box's GPU) on the tiny workload; -> the exchanged gradient
    bucket of its last timed step and the printed line."""
    import json
    import subprocess
    out = os.path.join(str(tmp_path), f"{tag}.pt")
    env = dict(os.environ, GSPLAT_AMD_BENCH_BACKEND="gloo", GSPLAT_AMD_DEBUG_BINS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # (the ranks share this process's card: what the tests before left cached here -- and what only the cyclic collector
    # frees -- is theirs to use; a rank that cannot allocate dies inside a collective's worker thread, which aborts the process)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info()
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(gpus), "--workload", "T", "--steps", "3",
           "--warmup", "1", "--views-per-gpu", str(views_per_gpu), "--view-streams", str(streams), "--exchange", exchange,
           "--no-cpu-baseline", "--no-extras", "--views", "1", "--dump-grad", out, *extra]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:   # (the whole stderr of the ranks: the first rank to fail is rarely in the last lines)
        try:
            dump = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out")
            os.makedirs(dump, exist_ok=True)
            with open(os.path.join(dump, f"bench_{tag}.stderr"), "w") as fh:
                fh.write(f"[test] card memory before the launch: {free_b / 2**30:.1f} GiB free of {total_b / 2**30:.1f}\n")
                fh.write(res.stderr)
        except OSError:
            pass
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    return torch.load(out), line


def test_bench_multi_view_step_overlapped_equals_serial_and_one_process(tmp_path):
    """bench.py's step with several views per GPU: (a) two ranks x two views with the views alternating on two HIP streams
    (view j's all-gather under view j + 1's render) leave the bucket of the one-stream run up to the order of the float
    atomics inside each view's sweep (two RUNS are compared: within a run the SH part -- rebuilt from the gathered rgb
    gradients in a fixed view order -- is bit-identical on every rank, test_fused_view_parallel_two_ranks_match_reference_trainer);
    (b) the same four ring cameras in ONE process (4 views per step) give the same sums; (c) the plain all-reduce form
    agrees with the factored one."""
    n = 20_000
    serial, line_s = _bench_bucket(tmp_path, "serial", 2, 2, 1)
    overlap, line_o = _bench_bucket(tmp_path, "overlap", 2, 2, 2)
    assert line_o["config"]["view_streams"] == 2 and line_s["config"]["view_streams"] == 1
    assert line_o["n_gpus"] == 2 and line_o["config"]["views_per_gpu_per_step"] == 2 and line_o["exchange"]["world_size"] == 2
    scale = float(serial.abs().max())
    assert scale > 0 and float((overlap[:11 * n] - serial[:11 * n]).abs().max()) <= 1e-5 * scale
    sh_s, sh_o = serial[11 * n:], overlap[11 * n:]
    assert float((sh_s - sh_o).abs().max()) <= 1e-5 * float(sh_s.abs().max())
    one, line_1 = _bench_bucket(tmp_path, "one", 1, 4, 2)
    assert line_1["n_gpus"] == 1 and float((one - serial).abs().max()) <= 2e-5 * scale
    plain, _ = _bench_bucket(tmp_path, "plain", 2, 2, 2, exchange="allreduce")
    assert float((plain - serial).abs().max()) <= 2e-5 * scale
    # (d) the views' speculation checked once per step (the default with several views: gsplat_amd.pipeline.ViewPipeline) or
    # inside every call: the same sums
    assert line_1["config"]["speculation_check"].startswith("per step") and line_1["config"]["overlap"] == "alternate"
    other, line = _bench_bucket(tmp_path, "percall", 1, 4, 2, extra=("--no-defer-check",))
    assert float((other - serial).abs().max()) <= 2e-5 * scale
    assert line["config"]["speculation_check"] == "per view" and line["config"]["overlap"] == "alternate"


def test_bench_exchange_executes_over_rccl_with_one_rank(tmp_path):
    """The RCCL branch of bench.py's exchange on the device: ONE rank with a real ``nccl`` process group
    (GSPLAT_AMD_BENCH_FORCE_DIST=1) -- ``all_gather_into_tensor`` behind every view's backward, the asynchronous
    all-reduce of the geometry floats, the host-side gloo flag group, the local SH rebuild -- leaves the gradient bucket of
    the plain one-process run (which finishes its views' projection backward in one launch instead).  What this cannot
    show is a second rank; that a collective's arguments are what RCCL accepts, it does."""
    import json
    import subprocess
    n = 20_000

    def run(tag, force):
        out = os.path.join(str(tmp_path), f"{tag}.pt")
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GSPLAT_AMD_BENCH_BACKEND"):
            env.pop(k, None)
        if force:
            env["GSPLAT_AMD_BENCH_FORCE_DIST"] = "1"
        cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "1", "--workload", "T", "--steps", "3",
               "--warmup", "1", "--views-per-gpu", "4", "--no-cpu-baseline", "--no-extras", "--views", "1", "--dump-grad", out]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        return torch.load(out), json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])

    plain, line_p = run("plain", False)
    rccl, line_r = run("rccl", True)
    assert "exchange" not in line_p and line_p["config"]["projection"] == "per step: forward + backward"
    assert line_r["exchange"]["backend"] == "nccl" and line_r["exchange"]["world_size"] == 1
    scale = float(plain.abs().max())
    assert scale > 0 and float((rccl - plain).abs().max()) <= 2e-5 * scale


def test_c5_run_iteration_through_a_densify_event_then_renders_at_the_new_population():
    """BASELINE config C5 as the loop runs it: ``ViewParallelTrainer.run`` at 5M gaussians / 3840x2160 / SH 3 (fused step) for two
    iterations -- the first ends in a densify / prune event (clones, cut by the ``max_gaussians`` cap), the second
    renders at the NEW population.  The record capacity of the second step is predicted from the remembered ratio (no
    blocking read-back, no repair: the ``N > 1.4M`` rule keeps these sizes on the LSD depth ordering, whose key bits are
    remembered too), the step produces finite metrics and the model, the optimizer and the bucket stay bound."""
    import gsplat_amd
    from gsplat_amd import scene, stages as S
    from gsplat_amd.harness import LoopConfig, SplatModel, StepConfig, ViewParallelTrainer
    N, W, H = 5_000_000, 3840, 2160
    sc = scene.make_scene(N, seed=0, sh_degree=3)
    model = SplatModel(N, "cuda")
    model.load(xyz=sc["means3D"].cuda(), features_dc=sc["shs"][:, :1].cuda(), features_rest=sc["shs"][:, 1:].cuda(),
               opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)).cuda(), scaling=torch.log(sc["scales"]).cuda(),
               rotation=sc["rotations"].cuda())
    K = scene.make_intrinsics(W, H).cuda()
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), StepConfig(), fused=True)
    tr.iteration = 5100                                     # > densify_after and a multiple of densify_interval
    g = torch.Generator(device="cuda").manual_seed(1)
    target = torch.rand(H, W, 3, generator=g, device="cuda")
    views = [{"image": target, "pose": scene.make_camera(v, 8).cuda(), "width": W, "height": H} for v in (0, 1)]

    def loader():
        i = 0
        while True:
            yield [views[i % 2]]
            i += 1

    calls, orig = [], S.bin_tiles

    def spy(*a, **kw):
        b = orig(*a, **kw)
        calls.append((a[0].shape[0], bool(kw.get("exact", False)), b._m is None))
        return b

    S.bin_tiles = spy
    trace = []
    try:
        # warm the shape's memory (capacity ratio, key bits) the way the iterations before 5100 would have
        tr.training_step([views[1]])
        tr.training_step([views[0]])
        calls.clear()
        for k in S.REPAIR_STATS:
            S.REPAIR_STATS[k] = 0
        loop = LoopConfig(iterations=5102, densify_grads_threshold=1e-12, max_gaussians=5_100_000)
        tr.run(loader(), extent=1.0, loop=loop, on_step=lambda it, n0, met, st: trace.append((it, n0, met, st)))
    finally:
        S.bin_tiles = orig
    (it0, n0, met0, st0), (it1, n1, met1, st1) = trace
    assert (it0, n0, it1) == (5100, N, 5101) and st0 is not None and st1 is None
    # (one step's gradient reaches the ~150 k gaussians in front -- the rest of the 5M is occluded; more than the cap admits)
    assert st0["cloned"] + st0["split"] >= 50_000 and st0["capped"] and n1 == st0["n_after"] == model.n != N
    assert all(math.isfinite(float(v)) for v in met1.values()) and 0.0 < met1["l1"] < 1.0
    # iteration 5101 binned the new population ONCE, for a predicted capacity, without waiting for its record count
    second = [c for c in calls if c[0] == n1]
    assert second == [(n1, False, True)], calls
    assert S.REPAIR_STATS["capacity_misses"] == 0 and S.REPAIR_STATS["key_range_repairs"] == 0, S.REPAIR_STATS
    model.check_bound()
    assert tr.optimizer.exp_avg.numel() == 59 * n1 and torch.isfinite(model.flat).all()


def test_fused_adam_step_at_c3_size():
    """gsa_adam_step over the 59 x 1M bucket with the reference's six learning rates == torch.optim.Adam, three steps."""
    from gsplat_amd.harness import FusedBucketAdam, SplatModel, StepConfig, _adam_groups
    N = 1_000_000
    g = torch.Generator(device="cuda").manual_seed(11)
    model = SplatModel(N, "cuda")
    with torch.no_grad():
        model.flat.copy_(torch.randn(59 * N, generator=g, device="cuda"))
    ref = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    groups = _adam_groups(model, StepConfig())
    ref_opt = torch.optim.Adam([{"params": [p], "lr": grp["lr"]} for p, grp in zip(ref, groups)], lr=0.0, eps=1e-15)
    opt = FusedBucketAdam(model, groups)
    for _ in range(3):
        grad = torch.randn(59 * N, generator=g, device="cuda") * 1e-3
        grad[::5] = 0.0
        model.flat_grad.copy_(grad)
        for p, q in zip(ref, model.parameters()):     # (the same gradient, field by field: the bucket's order is its own)
            p.grad = q.grad.detach().clone()
        ref_opt.step()
        opt.step()
    for p, q in zip(ref, model.parameters()):
        assert _rel(q.detach(), p.detach()) < 2e-6
    st = ref_opt.state[ref[2]]
    assert _rel(opt.state[model.features_rest]["exp_avg"], st["exp_avg"]) < 1e-6
    assert _rel(opt.state[model.features_rest]["exp_avg_sq"], st["exp_avg_sq"]) < 1e-6
