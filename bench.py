#!/usr/bin/env python3
"""bench.py -- Mpix/s of rasterizer forward+backward on the synthetic 1M-gaussian / 1080p scene
(BASELINE.json metric; SURVEY.md section 8d workload "C3"), 1..8 MI355X, one process per GPU.

A step = one pass of the hot path over one batch of views: every rank renders `--views-per-gpu` views (BASELINE
config C3: B = 4 views per optimisation step) -- rasterization forward (project, bin, composite) + the synthetic
upstream gradient v_render + full backward -- and the views' gradients are summed into ONE 59*N-float bucket; with
N > 1 ranks the view-parallel exchange (RCCL over xGMI) is inside the step.  `value` = all views of all ranks x W x H /
wall time.  The line also carries `single_view`: one view per step, one stream, a fixed camera -- the serial latency
of a view, which is what the per-stage `stage_ms` / `roofline` figures are taken on.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no launcher environment starts the N ranks itself (a child
``torch.distributed.run`` on 127.0.0.1, before this process touches the GPU) and relays rank 0's JSON line.
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "3d-gaussian-splatting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
DEFAULT_VIEWS_PER_GPU = 4   # BASELINE config C3: B = 4 views per optimisation step
DEFAULT_VIEW_STREAMS = 2
WORKLOADS = {
    # name: (gaussians, width, height, sh_degree)
    "C3": (1_000_000, 1920, 1080, 3),
    "C2": (100_000, 1920, 1080, 3),
    "C5": (5_000_000, 3840, 2160, 3),
    "T": (20_000, 320, 192, 3),   # (tests/test_scale.py rehearses the multi-rank / multi-view step on this one)
}


def algorithmic_bytes(n: int, m: int, p: int, sh_degree: int):
    """compulsory traffic per stage for one fwd+bwd view (bytes)."""
    ka = (sh_degree + 1) ** 2
    return {
        "gsa_project_fwd": n * (44 + 12 * ka + 48),
        "binning": m * (12 + 24),
        "gsa_composite_fwd": m * (4 + 36) + p * 20,
        "gsa_composite_bwd": m * (4 + 36 + 36) + p * 24,
        "gsa_project_bwd": n * (44 + 20 + 36 + 44 + 12 * ka + 192),
    }


def measured_copy_peak(dev) -> float:
    """Practical HBM ceiling of THIS box (SURVEY.md section 8d asks for it beside the nominal 8 TB/s): a 1 GiB
    device-to-device copy, bytes read + written over the best of 5 HIP-event timings, in GB/s."""
    import torch
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    best = float("inf")
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return 2.0 * src.numel() * 4 / (best * 1e-3) / 1e9


def host_cores() -> int:
    """CPU threads this process may use (the GPU box gives one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(n: int, width: int, height: int, sh_degree: int, crop=(1280, 720)):
    """The CPU oracle ("port": the restatement of the gsplat contract the reference lacks) timed on this box's
    host cores, forward+backward, on a bounded sample of the same workload: the central crop of the same
    1M-gaussian 1080p view (all N gaussians are projected; 4/9 of the pixels are composited; tile batches are
    recomputed in the backward pass -- ``checkpoint=True`` -- to bound host memory)."""
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization

    cores = host_cores()
    torch.set_num_threads(cores)
    cw, ch = crop
    sc = scene.make_scene(n, seed=0, sh_degree=sh_degree)
    K = scene.make_intrinsics(width, height).clone()
    K[0, 2] -= (width - cw) / 2.0
    K[1, 2] -= (height - ch) / 2.0
    vm = scene.make_camera(0, 1)
    v = scene.make_v_render(cw, ch)
    leaves = [sc[k].clone().requires_grad_(True) for k in ("means3D", "rotations", "scales", "opacities", "shs")]
    t0 = time.perf_counter()
    r, _, meta = oracle_rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm[None], K[None],
                                      cw, ch, sh_degree=sh_degree, backgrounds=torch.zeros(1, 3), checkpoint=True)
    (r[0] * v).sum().backward()
    dt = time.perf_counter() - t0
    return {"value": round(cw * ch / dt / 1e6, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"central {cw}x{ch} crop of the {n}-gaussian {width}x{height} view 0, fwd+bwd, torch fp32 "
                      f"oracle (tile batches checkpointed), 1 run of {dt:.1f} s, M={meta['flatten_ids'].numel()}"}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_protocol():
    """BASELINE.md section 3 / SURVEY.md section 8d: the CPU oracle at C1 and C2 sizes, forward-only and
    forward+backward, 1 warm-up + median of 5, Mpix/s, with the core count and CPU model."""
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization

    cores = host_cores()
    torch.set_num_threads(cores)
    rows = {}
    for name, (n, w, h, deg) in (("C1", (10_000, 256, 256, 0)), ("C2", WORKLOADS["C2"])):
        sc = scene.make_scene(n, seed=0, sh_degree=deg)
        K, vm, v = scene.make_intrinsics(w, h), scene.make_camera(0, 1), scene.make_v_render(w, h)
        for mode in ("fwd", "fwd+bwd"):
            times = []
            for _ in range(6):
                grad = mode == "fwd+bwd"
                leaves = [sc[k].clone().requires_grad_(grad) for k in ("means3D", "rotations", "scales", "opacities", "shs")]
                t0 = time.perf_counter()
                with torch.set_grad_enabled(grad):
                    r, _, meta = oracle_rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm[None],
                                                      K[None], w, h, sh_degree=deg, backgrounds=torch.zeros(1, 3),
                                                      checkpoint=grad)
                    if grad:
                        (r[0] * v).sum().backward()
                times.append(time.perf_counter() - t0)
            med = sorted(times[1:])[2]
            rows[f"{name}_{mode}"] = {"Mpix/s": round(w * h / med / 1e6, 4), "median_s": round(med, 3),
                                      "M": int(meta["flatten_ids"].numel())}
    return {"what": "CPU restatement of the reference's gsplat contract (oracle/, torch fp32) -- a reported baseline, "
                    "not the target; 1 warm-up + median of 5", "cores": cores, "cpu_model": cpu_model(), "runs": rows}


def train_step_leg(sc, n, width, height, deg, views, dev, steps):
    """BASELINE config C3's "full Adam loop" step on the product's fused harness (gsplat_amd.harness.ViewParallelTrainer,
    fused=True: activations, B x (render, loss, backward), SH rebuild, Adam; metrics read back one step late): ms per
    step and Mpix/s.  Reported beside `value`, never part of it."""
    import gsplat_amd
    from gsplat_amd import scene
    from gsplat_amd.harness import SplatModel, ViewParallelTrainer
    K = scene.make_intrinsics(width, height).to(dev)
    batch = [{"image": torch.rand(height, width, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + v)),
              "pose": scene.make_camera(v, views).to(dev), "width": width, "height": height} for v in range(views)]
    model = SplatModel(n, dev).load(xyz=sc["means3D"].to(dev), features_dc=sc["shs"][:, :1].to(dev),
                                    features_rest=sc["shs"][:, 1:].to(dev),
                                    opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)).to(dev),
                                    scaling=torch.log(sc["scales"]).to(dev), rotation=sc["rotations"].to(dev))
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), fused=True)
    tr.async_metrics = True
    tr.iteration = 1000 * deg     # the SH degree of the workload (trainer.py:77-87)
    for _ in range(4 + steps):   # (an un-timed window of the same length directly in front of the fence: timed() says why)
        tr.training_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.training_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"what": "fused harness step: activations, %d x (render, loss, backward), SH rebuild, Adam; metrics one step late" % views,
            "views": views, "steps": steps, "ms_per_step": round(dt * 1e3, 4),
            "value": round(views * width * height / dt / 1e6, 3), "unit": "Mpix/s", "loss": round(float(out["loss"]), 5)}


def self_launch(args) -> int:
    """``--gpus N`` without a launcher environment: start N ranks with torch.distributed.run as a CHILD process
    (nothing in this process has touched the GPU yet), relay its output, return its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # The pool's host driver only supports dmabuf IPC: with the legacy IPC mode RCCL's buffer registration between the
    # ranks of one node fails with `hipIpcGetMemHandle: invalid argument`.  The image exports this already; it is
    # repeated here so that a shell which lost it still gets working collectives (never overrides a caller's value).
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def step_cameras(i: int, views: int, world: int, rank: int, ring_size: int):
    """Ring cameras of THIS rank's views of step i: the step renders the `views x world` consecutive cameras starting at
    (i x views x world) mod ring_size, dealt round-robin -- view j of rank r is camera base + j x world + r (rank r
    renders views {v : v mod R = r} of the batch)."""
    base = (i * views * world) % ring_size
    return [(base + j * world + rank) % ring_size for j in range(views)]


def step_window(i: int, views: int, world: int, ring_size: int):
    """All cameras of step i in batch order: entry j x world + r is rank r's view j (what the exchange's local SH rebuild sums over)."""
    base = (i * views * world) % ring_size
    return [(base + k) % ring_size for k in range(views * world)]


def percentiles(xs):
    """``max_after_fill``: the largest step behind the pass's first one -- that pass starts on an idle device, so its step 0
    is the pipeline filling (2.5 ms where the others take 1.9) and is always the plain ``max``."""
    rest = xs[1:] or xs
    xs = sorted(xs)
    pick = lambda q: xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]  # noqa: E731
    return {"mean": round(sum(xs) / len(xs), 4), "median": round(pick(0.5), 4), "p10": round(pick(0.1), 4),
            "p90": round(pick(0.9), 4), "max": round(xs[-1], 4), "max_after_fill": round(max(rest), 4)}


class Exchange:
    """The view-parallel exchange of one step: sums the 59 N-float gaussian gradient over the ranks.

    ``factored``: every view leaves d loss / d rgb (3 N floats) instead of the 48 N-float
    SH-coefficient gradient; the ranks all-GATHER that per view -- right behind the view's backward pass, so that with
    several views per GPU the gather of view j travels while view j + 1 renders -- all-REDUCE the 11 N geometry floats
    once per step, and rebuild sum_views basis (x) v_rgb locally (``expand``: stages.sh_grad_expand; summed in a fixed
    view order, so the result is bit-identical on every rank).  Otherwise: one all-reduce of the whole bucket.

    ``dist`` is torch.distributed or any object with its collectives (tests/test_bench_exchange.py drives this class with
    a recording stand-in on the CPU, so that a shape or argument mistake in the RCCL branch does not wait for an 8-GPU
    node to be found); ``dist = None``: one process, the factored form still saves the 48 N-float read-modify-write of
    every view but the first."""

    def __init__(self, dist, world: int, rank: int, n: int, views_per_gpu: int, factored: bool, device, expand=None):
        self.dist, self.world, self.rank, self.n, self.B, self.factored = dist, world, rank, n, views_per_gpu, factored
        self.expand = expand
        self.backend = dist.get_backend() if dist is not None else None
        # [view j of the step][rank][gaussian][rgb]: slot (j, rank) is what this rank's view j writes; gather j fills row j
        self.v_rgb = torch.zeros(views_per_gpu, world, n, 3, device=device) if factored else None
        self.pending = []
        self.bytes_sent = 0   # by this rank, over all steps so far (ring / direct algorithms: (R - 1) / R of a buffer per phase)
        self.steps = 0

    def view_target(self, j: int):
        """Where view j of this rank's step writes its rgb gradient (None: plain form)."""
        return self.v_rgb[j, self.rank] if self.factored else None

    def after_view(self, j: int) -> None:
        """Queued right behind view j's backward pass, on its stream."""
        if self.dist is None or not self.factored:
            return
        mine = self.v_rgb[j, self.rank]
        if self.backend == "nccl":
            work = self.dist.all_gather_into_tensor(self.v_rgb[j].view(self.world * self.n, 3), mine, async_op=True)
        else:   # (gloo rehearsal: list form; the input must not alias an output)
            work = self.dist.all_gather(list(self.v_rgb[j].unbind(0)), mine.clone(), async_op=True)
        self.pending.append(work)
        self.bytes_sent += mine.numel() * 4 * (self.world - 1)

    def finish(self, flat_grad, means, deg: int, viewmats, Kmat, v_colors) -> None:
        """After the last view's backward: the geometry all-reduce, the gathers' completion, the local rebuild.
        viewmats: [B * world, 4, 4], entry j * world + r = the camera of rank r's view j."""
        n = self.n
        self.steps += 1
        if self.factored:
            summed = None
            if self.dist is not None:
                summed = self.dist.all_reduce(flat_grad[:11 * n], async_op=True)
                self.bytes_sent += 2 * 11 * n * 4 * (self.world - 1) // self.world   # (ring / direct: 2 (R-1)/R of the buffer)
            for work in self.pending:
                work.wait()
            self.pending = []
            self.expand(means, deg, viewmats, Kmat, self.v_rgb.view(self.B * self.world, n, 3), v_colors)
            if summed is not None:
                summed.wait()
        elif self.dist is not None:
            self.dist.all_reduce(flat_grad)
            self.bytes_sent += 2 * flat_grad.numel() * 4 * (self.world - 1) // self.world


def rccl_version() -> str:
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as exc:   # noqa: BLE001
        return f"unknown ({type(exc).__name__})"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # x 4 views: 0.25 s of device time
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--cull-mode", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the (untimed-by-value) training-step leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the legs `value` and the multi-rank figures come from (tests): no fixed-camera / rank-step / "
                         "forward-only / rotating / wide-depth / training-step legs")
    ap.add_argument("--exchange", default="factored", choices=("factored", "allreduce"),
                    help="how the gaussian gradients of the views are summed (see DESIGN.md section 6)")
    ap.add_argument("--views-per-gpu", type=int, default=DEFAULT_VIEWS_PER_GPU,
                    help="views every rank renders per step (BASELINE config C3: B = 4 views per step); the cameras of a "
                         "step are `views-per-gpu x gpus` consecutive cameras of the ring")
    ap.add_argument("--view-streams", type=int, default=DEFAULT_VIEW_STREAMS,
                    help="HIP streams the views of a step alternate on (2: view j + 1's HBM- / latency-bound front end "
                         "runs beside view j's VALU-bound sweeps; 1: strictly one view after the other)")
    ap.add_argument("--no-defer-check", action="store_true",
                    help="multi-view steps: wait for every view's control words inside its rasterization() call (the r03 "
                         "behaviour) instead of checking the step's views once, behind the last one (defer_check=True)")
    ap.add_argument("--views", type=int, default=8,
                    help="ring size of the ROTATING-view pass (the reference samples random views every step, "
                         "trainer.py:315-317): one view per step, cycling through this many cameras")
    ap.add_argument("--ring", type=int, default=0,
                    help="cameras on the ring the HEADLINE step's views are taken from, step after step (0: max(8, 2 x gpus x "
                         "views-per-gpu), so that consecutive steps render different cameras, as a trainer's do: "
                         "trainer.py:315-317 samples random views); `value_fixed_cameras` is the same step on gpus x views-per-gpu "
                         "cameras, i.e. the same cameras every step")
    ap.add_argument("--no-batch-projection", action="store_true",
                    help="project the gaussians once per view instead of once per step for all its views "
                         "(gsplat_amd.project_cameras: geometry and SH rows read once)")
    ap.add_argument("--no-batch-backward", action="store_true",
                    help="one projection backward per view (+ the SH expansion) instead of one launch for all views of the step "
                         "(gsplat_amd.project_cameras_backward; single process only)")
    ap.add_argument("--rank-batch", action="store_true",
                    help="multi-rank / rank-form steps: colour gradients out of the backward sweeps + ONE geometry-only projection "
                         "backward per step (gsa_composite_bwd_lists_rgb / gsa_project_bwd_batch_geometry) instead of one factored "
                         "projection backward per view -- measured slower (profiles/r05_rank_step.md), kept for A/B")
    ap.add_argument("--dump-grad", default=None,
                    help="(tests) rank 0 saves the gradient bucket of the last step of the timed region to this .pt file")
    ap.add_argument("--profile-views", type=int, default=1,
                    help="with --profile-run: views per step (default 1: single-view steps)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3: only `warmup` + `steps` single-view steps (one view per step, one stream, fixed "
                         "camera), no settling, no extra passes; prints a minimal line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    # "nccl" is RCCL; GSPLAT_AMD_BENCH_BACKEND=gloo only exists to rehearse the N > 1 code path on a one-GPU box
    # (the ranks then share the card)
    backend = os.environ.get("GSPLAT_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # GSPLAT_AMD_BENCH_FORCE_DIST=1: a process group even for ONE rank -- the exchange's RCCL branch (all-gather per view,
    # all-reduce per step, the host-side flag group) then executes on a one-GPU box, with nothing to exchange (tests)
    if world > 1 or os.environ.get("GSPLAT_AMD_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    # the ranks agree on "did every view's speculation hold" through a HOST-side group: a collective on the device would
    # sit behind the step's kernels in stream order, and the host is a step ahead of those
    flag_group, flag_error = None, None
    if dist is not None:
        try:
            flag_group = dist.new_group(backend="gloo")
        except Exception as exc:   # noqa: BLE001  (no usable host interface for gloo)
            flag_error = f"{type(exc).__name__}: {exc}"
        # every rank must take the SAME decision (a rank that defers alone would wait in an all-reduce the others never
        # enter): one collective on the default group says whether all of them got the host-side group
        have = torch.tensor([0 if flag_group is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) == 0:
            flag_group = None
            args.no_defer_check = True
            print(f"[bench] rank {rank}: no host-side group for the per-step check on every rank ({flag_error}); "
                  "checking per view", file=sys.stderr)
    agree_stats = [0, 0.0]   # agreements, seconds spent in them (host side)

    def agree(ok: bool) -> bool:
        if dist is None:
            return ok
        t0 = time.perf_counter()
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=flag_group)
        agree_stats[0] += 1
        agree_stats[1] += time.perf_counter() - t0
        return bool(t.item())

    import gsplat_amd
    from gsplat_amd import scene, stages
    from gsplat_amd.pipeline import ViewPipeline, ViewState

    B = max(1, args.views_per_gpu)
    n, width, height, deg = WORKLOADS[args.workload]
    sc = scene.make_scene(n, seed=0, sh_degree=deg, device="cpu")
    # the five rasterizer inputs (59 floats per gaussian) are views of one flat buffer and their gradients are
    # views of one flat bucket, so the view-parallel exchange is ONE all-reduce of 59*N floats (harness.SplatModel
    # lays the six raw parameter tensors out the same way)
    shapes = (("means3D", (n, 3)), ("rotations", (n, 4)), ("scales", (n, 3)), ("opacities", (n, 1)), ("shs", (n, 16, 3)))
    flat = torch.empty(59 * n, dtype=torch.float32, device=dev)
    flat_grad = torch.zeros_like(flat)
    leaves, grad_views, off = [], [], 0
    for name, shape in shapes:
        cnt = 1
        for d in shape:
            cnt *= d
        leaf = flat[off:off + cnt].view(shape)
        leaf.copy_(sc[name].to(dev))
        leaf.requires_grad_(True)
        leaves.append(leaf)
        grad_views.append(flat_grad[off:off + cnt].view(shape if name != "opacities" else (n,)))
        off += cnt
    opac = leaves[3]
    K = scene.make_intrinsics(width, height).to(dev)[None]
    v_render = scene.make_v_render(width, height, seed=2).to(dev)[None]
    bg = torch.zeros(1, 3, device=dev)
    info = {}
    rings = {}

    def ring(v):   # the v cameras of a ring, on the device
        if v not in rings:
            rings[v] = torch.stack([scene.make_camera(i, v) for i in range(v)]).to(dev)
        return rings[v]

    def make_step(form: str, views: int, n_streams: int, ring_size: int, cams_override=None, rank_form: bool = False):
        """One step = `views` views on this rank (cameras j * world + rank of the step's window of the ring), forward +
        backward each, their gradients summed into the bucket, then the exchange.  form: "factored" | "allreduce".
        The views are queued by the product's own orchestrator (gsplat_amd.pipeline.ViewPipeline): alternating streams,
        bucket summed in view order, the speculation of all views checked once behind the last one."""
        # the projection of the gaussians for all views of the step in ONE launch (their geometry and coefficient rows
        # read once), queued in front of the first view; --no-batch-projection: one projection launch per view
        batch_proj = views > 1 and not args.no_batch_projection
        # ... and, in one process, their projection BACKWARD in one launch behind the last sweep (geometry read once, every
        # gradient written once, the SH-coefficient gradient formed in registers: no per-view rgb gradients, no expansion
        # kernel).  With several ranks the per-view factored form stays: its all-gathers travel under the next view.
        # rank_form: the step a rank of a multi-GPU run executes (per-view factored backward, local SH rebuild), in ONE process
        batch_bwd = batch_proj and dist is None and deg >= 1 and views <= 8 and not args.no_batch_backward and not rank_form   # (the kernel's camera limit)
        # one process, one view: nothing to sum, the backward writes the whole gradient itself
        factored = form == "factored" and (world > 1 or views > 1) and not batch_bwd
        # ... and for the RANKS of a view-parallel run (r05): every view's backward SWEEP leaves its masked colour gradient in
        # the exchange's [N,3] slot (the all-gather of view j is queued behind view j's sweep, under view j + 1), and the
        # geometry gradients of the rank's views are summed by ONE launch behind the last sweep (project_cameras_backward on
        # projections that carry v_rgb) -- instead of one factored projection backward per view
        # MEASURED SLOWER and off (profiles/r05_rank_step.md; --rank-batch switches it on): the compact [N,3] atomics cost the
        # sweep 0.13 ms per view where the factored projection backward they replace takes 0.037
        rank_batch = args.rank_batch and factored and batch_proj and deg >= 1 and views <= 8 and not args.no_batch_backward
        ex = Exchange(dist, world, rank, n, views, factored, dev, expand=stages.sh_grad_expand)
        pipe = ViewPipeline(dev, n_streams)
        cams, windows = (ring(ring_size) if cams_override is None else cams_override), {}
        defer = views > 1 and not args.no_defer_check   # the views' speculation is checked once per step, behind the last view

        shared, step_cams = {}, {}

        def step(i: int = 0):
            for t in leaves:
                t.grad = None
            base = (i * views * world) % ring_size
            marks, turns = {}, {}

            def prepare():
                if base not in step_cams:   # this rank's cameras of the step, one [views, 4, 4] tensor per window of the ring
                    idx = torch.tensor(step_cameras(i, views, world, rank, ring_size), device=dev)
                    step_cams[base] = cams[idx].contiguous()
                shared["proj"] = gsplat_amd.project_cameras(leaves[0], leaves[1], leaves[2], opac.squeeze(-1), leaves[4],
                                                            step_cams[base], K[0], width, height, sh_degree=deg,
                                                            cull_mode=args.cull_mode, defer_backward=batch_bwd or rank_batch)
                return gsplat_amd.projection_buffers(shared["proj"])

            def forward(j: int, deferred: bool) -> ViewState:
                # (with a shared projection: the camera tensor it was made for -- rasterization(projection=...) checks that)
                vm = shared["proj"][j].viewmat[None] if batch_proj else cams[step_cameras(i, views, world, rank, ring_size)[j]][None]
                # the backward pass writes every gradient element straight into the bucket (no zero-fill, no copy);
                # views after the first ADD to it.  (opacities [N,1] -> [N] as the reference backend does it,
                # gsplat_backend.py:157-158)
                r, a, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], opac.squeeze(-1), leaves[4], vm, K,
                                                      width, height, sh_degree=deg, backgrounds=bg,
                                                      cull_mode=args.cull_mode, grad_out=grad_views,
                                                      grad_accumulate=j > 0, sh_grad_factored=ex.view_target(j),
                                                      defer_check=deferred, speculation=pipe.speculation,
                                                      projection=shared["proj"][j] if batch_proj else None,
                                                      grad_ready=(lambda j=j: turns.get(j)))
                if "M" not in info:   # (once: with 32 x 32 bins in the hot path the 16 x 16 record count is a reduction + read-back)
                    info["M"] = meta["n_isects"]
                    info["binned"] = meta["n_binned"]
                return ViewState(r, (lambda: bool(meta["speculation_ok"])) if deferred else (lambda: True))

            def backward(j: int, state: ViewState, before_bucket) -> None:
                # (with the shared projection backward a view's own backward pass stops behind its sweep and never touches
                # the bucket: nothing to wait for)
                # (a view that finishes its own projection backward: THAT waits for the bucket's turn -- rasterization(grad_ready=)
                # -- not the sweep in front of it, which touches nothing of the bucket)
                turns[j] = before_bucket if not (batch_bwd or rank_batch) else None
                state.payload.backward(v_render)

            def after_view(j: int) -> None:
                if dist is not None and stages.TIMER.enabled and j == views - 1:   # instrumented pass: the exchange's tail
                    marks["x0"] = torch.cuda.Event(enable_timing=True)
                    marks["x0"].record()
                ex.after_view(j)

            def finish():   # the projection backward of all views of the step: one launch, straight into the bucket
                gsplat_amd.project_cameras_backward(shared["proj"], leaves[0], leaves[1], leaves[2], opac.squeeze(-1), leaves[4],
                                                    grad_views)

            pipe.run(views, forward, backward, after_view=after_view, agree=agree if dist is not None else None, defer=defer,
                     prepare=prepare if batch_proj else None, finish=finish if (batch_bwd or rank_batch) else None)
            if factored or dist is not None:
                if base not in windows:   # the cameras of the step's views, entry j * world + r = rank r's view j
                    windows[base] = cams[torch.tensor(step_window(i, views, world, ring_size), device=dev)].contiguous()
                ex.finish(flat_grad, leaves[0], deg, windows[base], K, grad_views[4])
            if dist is not None and stages.TIMER.enabled:
                x1 = torch.cuda.Event(enable_timing=True)
                x1.record()
                stages.TIMER.records.append(("exchange_tail", marks["x0"], x1))
            pipe.mark_bucket_used()

        step.exchange = ex
        step.pipe = pipe
        step.batch_proj, step.batch_bwd, step.rank_batch = batch_proj, batch_bwd, rank_batch
        return step

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def settle(step, limit=400, chunk=10, tol=0.01):
        """Un-timed steps until three consecutive chunks of `chunk` steps take the same time within `tol` (bounded): the
        first calls of a shape take the exact-size / LSD paths, the clocks ramp, and a short timed window right behind a
        short warm-up measures that instead of the steady state.  -> steps run."""
        times, done = [], 0
        while done < limit:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(chunk):
                step(done + k)
            b.record()
            b.synchronize()
            done += chunk
            ms = a.elapsed_time(b)
            if dist is not None:   # the steps contain collectives: every rank must take the same decision, on the same number
                t = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
            times.append(ms)
            if len(times) >= 3 and max(times[-3:]) <= (1.0 + tol) * min(times[-3:]):
                break
        return done

    def timed(step, steps, warmup, per_step=False, converge=False):
        """`warmup` untimed steps (+ `settle` with converge), then exactly `steps` steps between two fences.  per_step:
        one event per step boundary on the launch stream gives the per-step device times (median / p10 / p90) -- in a
        pass of its own: an event record leaves a ~6 us bubble on the stream (rocprof kernel trace)."""
        for i in range(warmup):
            step(i)
        extra = settle(step) if converge else 0
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step else None
        # The host queues a step in ~1.3 ms of Python against ~1.9 ms of device time; what eats its lead on a slower host is
        # the cyclic collector walking the long-lived objects of the set-up (scene tensors, modules, the interpreter's own)
        # every few hundred allocations: 0.2 ms per step (tools/bench_nogc.py).  Freezing what exists now -- the usual
        # measure for a long-running loop after its initialisation -- leaves the collector on for what the steps create.
        gc.collect()
        gc.freeze()
        # LAST in front of the window, because the device must not sit idle there: after ~10 ms without work the shader clock
        # needs ~20 ms of load to come back (tools/ramp_stages.py: the two sweeps of a view 0.628 + 0.902 ms behind a 34-ms
        # stall, 0.552 + 0.783 ten steps later; projection and binning do not move) -- the collection above is such a gap,
        # and a 20-step window behind it is all ramp.  So: un-timed windows of the same length, the fence, the window.
        # With `converge` they also run until the caching allocator has stopped growing (in a window without a host wait
        # the host runs further ahead than in settle()'s chunks, more buffers are alive at once, and new segments mean
        # hipMalloc calls inside the window), at most five.
        for k in range(5 if converge else 1):
            grown = torch.cuda.memory_stats().get("segment.all.allocated", 0)
            for i in range(steps):
                step(i)
            extra += steps
            if torch.cuda.memory_stats().get("segment.all.allocated", 0) == grown:
                break
            fence()
        fence()
        stages.WAIT_STATS[:] = [0, 0.0]
        agree_stats[:] = [0, 0.0]
        pipe_ = getattr(step, "pipe", None)
        if pipe_ is not None:   # what the window's views do to the step's own speculation state
            for k in pipe_.speculation.repairs:
                pipe_.speculation.repairs[k] = 0
            repaired0 = pipe_.repairs
        dbg = os.environ.get("GSPLAT_AMD_BENCH_DEBUG") == "1"
        if dbg:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if dbg:
            e0.record()
        if per_step:
            marks[0].record()
        host_t = []
        for i in range(steps):
            step(i)
            if per_step:
                marks[i + 1].record()
            if dbg:
                host_t.append((time.perf_counter() - t0, stages.WAIT_STATS[1]))
        if dbg:
            prev_t, prev_w, rows = 0.0, 0.0, []
            for t_, w_ in host_t[:20]:
                rows.append(f"{(t_ - prev_t) * 1e3:.2f}({(w_ - prev_w) * 1e3:.2f})")
                prev_t, prev_w = t_, w_
            print("[bench debug] host ms per step (of which waiting for control words): " + " ".join(rows), file=sys.stderr)
        wait_ms = stages.WAIT_STATS[1] / max(steps, 1) * 1e3   # (before the closing fence)
        if dbg:
            e1.record()
            tq = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if dbg:
            ms_ = torch.cuda.memory_stats()
            print(f"[bench debug] segments allocated so far {ms_.get('segment.all.allocated')}, freed {ms_.get('segment.all.freed')}, "
                  f"reserved {ms_.get('reserved_bytes.all.current', 0) / 2**30:.2f} GiB", file=sys.stderr)
            print(f"[bench debug] steps {steps} per_step {per_step}: wall {dt * 1e3:.3f} ms, host done queueing at {tq * 1e3:.3f} ms, "
                  f"device span (main stream events) {e0.elapsed_time(e1):.3f} ms", file=sys.stderr, flush=True)
        if dist is not None:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return {"s": dt, "warmup_effective": warmup + extra, "host_wait_ms": wait_ms,
                "agree_ms": agree_stats[1] / max(steps, 1) * 1e3,
                "stats": dict(pipe_.speculation.repairs) if pipe_ is not None else None,
                "repaired_steps": (pipe_.repairs - repaired0) if pipe_ is not None else None,
                "per_step_ms": [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)] if per_step else None}

    STAGES = ("gsa_project_fwd", "binning", "gsa_composite_fwd", "gsa_composite_bwd", "gsa_project_bwd")

    def per_stage(summary, steps):
        """StageTimer summary -> ({stage: ms per step}, {stage: C-ABI calls per step}); binning = depth ordering + record
        generation + tile passes (everything that is not one of the four single-launch stages or named below)."""
        per, launches = {}, {}
        renames = {"gsa_sh_grad_expand": "sh_grad_rebuild", "gsa_composite_bwd_prepare": "bwd_prepare", "exchange_tail": "exchange_tail"}
        for name, (calls, total) in summary.items():
            key = name if name in STAGES else renames.get(name, "binning")
            per[key] = per.get(key, 0.0) + total / steps
            launches[key] = launches.get(key, 0) + calls / steps
        return per, launches

    def instrumented(step, steps):
        """`steps` more steps with HIP events around every C-ABI stage (the event records cost a few percent: never the timed pass)."""
        stages.TIMER.reset(enabled=True)
        for i in range(steps):
            step(i)
        fence()
        out = dict(stages.TIMER.summary())
        stages.TIMER.reset(enabled=False)
        return out

    def count_records(cams_list):
        """Mean 16 x 16-rule intersections (SURVEY's M) and binned records over the given cameras: one forward-only call each on
        a throw-away speculation state (a reduction and a read-back per camera; outside every timed window)."""
        spec = stages.Speculation()
        ms, bs = [], []
        with torch.no_grad():
            for vm in cams_list:
                _, _, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], opac.squeeze(-1), leaves[4], vm[None], K, width,
                                                      height, sh_degree=deg, backgrounds=bg, cull_mode=args.cull_mode, speculation=spec)
                ms.append(int(meta["n_isects"]))
                bs.append(int(meta["n_binned"]))
        return sum(ms) / len(ms), sum(bs) / len(bs)

    p = width * height
    n_streams = max(1, args.view_streams) if B > 1 else 1
    if args.profile_run:
        # (--profile-views V: V-view steps on ONE stream -- the kernels of the multi-view step, the per-step projection
        # launches among them, one after the other: what the PMC passes of tools/collect_pmc.sh read their bytes from)
        pv = max(1, args.profile_views)
        prof_step = make_step(args.exchange, pv, 1, world * pv)
        res = timed(prof_step, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({"profile_run": True, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_view": round(res["s"] / args.steps * 1e3, 4), "intersections_M": info["M"],
                              "binned_records": info.get("binned")}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- the timed region `value` comes from: B views per rank and step, the default exchange form
    # (on a ring of more cameras than a step has views: consecutive steps render DIFFERENT cameras, so what the binning
    # remembers -- record capacities, key ranges -- has to hold across cameras, as in a trainer's loop)
    head_ring = args.ring if args.ring > 0 else max(8, 2 * world * B)
    head_step = make_step(args.exchange, B, n_streams, head_ring)
    head = timed(head_step, args.steps, args.warmup, converge=True)
    grad_l1_head = float(flat_grad.double().abs().sum())   # of the last step's exchanged bucket (same on every rank)
    if args.dump_grad and rank == 0:
        torch.save(flat_grad.detach().cpu(), args.dump_grad)
    dist_ms = timed(head_step, args.steps, 0, per_step=True)["per_step_ms"]   # the same steps, one event per step
    # the step form the ranks of a multi-GPU run execute (shared projection forward; per-view factored backward + local SH
    # rebuild, no shared projection backward), timed in this one process: what the N > 1 values are like-for-like with
    rank_step = timed(make_step(args.exchange, B, n_streams, head_ring, rank_form=True), args.steps, 2, converge=True) \
        if (dist is None and B > 1 and not args.no_extras) else None
    # the same step on the SAME cameras every step (r04's headline protocol), beside it
    fixed = timed(make_step(args.exchange, B, n_streams, world * B), args.steps, 2, converge=True) \
        if (head_ring != world * B and not args.no_extras) else head
    c4 = None
    if dist is not None and B > 1:   # BASELINE config C4 as written: ONE view per GPU per step (a world-view batch)
        c4_step = make_step(args.exchange, 1, 1, world)
        c4 = timed(c4_step, args.steps, 2, converge=True)
        stages.TIMER.reset(enabled=True)
        for i in range(min(args.steps, 20)):
            c4_step(i)
        fence()
        tail1 = stages.TIMER.summary().get("exchange_tail", (0, 0.0))
        stages.TIMER.reset(enabled=False)
        c4["exchange_tail_ms"] = tail1[1] / max(tail1[0], 1)
        # ... and the B-view step's tail, the same way
        stages.TIMER.reset(enabled=True)
        for i in range(min(args.steps, 20)):
            head_step(i)
        fence()
        tailb = stages.TIMER.summary().get("exchange_tail", (0, 0.0))
        stages.TIMER.reset(enabled=False)
        c4["exchange_tail_ms_B"] = tailb[1] / max(tailb[0], 1)
    other = None
    if dist is not None:   # the same steps with the other exchange form, for the comparison figure only
        other_form = "allreduce" if args.exchange == "factored" else "factored"
        other = timed(make_step(other_form, B, n_streams, head_ring), args.steps, 2, converge=True)
    # ---- ONE view per step, one stream, a fixed camera: the serial latency of a view (rounds 1 and 2's headline), then
    # the same steps with HIP events around every C-ABI stage for the roofline object (the event records cost a few
    # percent, which is why the instrumented pass is not the timed one)
    solo_step = make_step(args.exchange, 1, 1, world)
    # (warm-up: the remembered depth-key range is the union over the four ring cameras of the pass above; it shrinks to this
    # one camera after 32 consecutive views of it, and the bucket plan with it -- inside the timed window otherwise, which
    # made this figure read 0.62 or 0.70 ms depending on where the switch fell)
    solo = timed(solo_step, args.steps, max(2, stages.SPECULATION_SETTLE_VIEWS), converge=True)
    stage_ms = instrumented(solo_step, args.steps)
    grad_l1 = float(flat_grad.double().abs().sum())   # of the last single-view step's exchanged gradient (same on every rank)
    # ---- rotating views: one view per step cycling a ring of cameras -- how often the speculation behind the
    # asynchronous binning (record capacity, depth-key range, bucket capacity) has to be repaired
    rot = None
    if world == 1 and args.views > 1 and not args.no_extras:
        rot_step = make_step(args.exchange, 1, 1, args.views)
        for i in range(args.views):   # every camera once: the first call of a shape is exact by construction
            rot_step(i)
        rot = timed(rot_step, args.steps, 0)
        rot["stage_ms"] = per_stage(instrumented(rot_step, args.steps), args.steps)[0]
        rot["M"], rot["binned"] = count_records(list(ring(args.views)))
    # ---- wide depth: the same scene seen from INSIDE the cloud (ring of radius 0.35: visible depths from the near plane to
    # ~2.3, i.e. 8 octaves of depth-key span where the ring of radius 4 shows 0.7) -- which depth ordering ran, how often
    wide = None
    if world == 1 and args.views > 1 and not args.no_extras:
        wide_cams = torch.stack([scene.make_camera(i, args.views, radius=0.35) for i in range(args.views)]).to(dev)
        wide_step = make_step(args.exchange, 1, 1, args.views, cams_override=wide_cams)
        for i in range(2 * args.views):
            wide_step(i)
        for k in stages.PATH_STATS:
            stages.PATH_STATS[k] = 0
        wide = timed(wide_step, args.steps, 0)
        wide["paths"] = dict(stages.PATH_STATS)
        wide["stage_ms"] = per_stage(instrumented(wide_step, args.steps), args.steps)[0]
        wide["M"], wide["binned"] = count_records(list(wide_cams))
        wide["key_span_octaves"] = stages.last_key_span_octaves(dev, width, height, speculation=wide_step.pipe.speculation)
    # ---- forward only: what the reference's no_grad callers get (validation renders trainer.py:417-424, eval.py:171-183):
    # backend.render_with_depth under no_grad, one view per call, a fixed camera, the call checking its own speculation
    fwd_only = None
    if world == 1 and not args.no_extras:
        backend = gsplat_amd.HipBackend(K[0])
        params = {"means3D": leaves[0], "rotations": leaves[1], "scales": leaves[2], "opacities": opac, "shs": leaves[4]}
        viewpoint = {"world_view_transform": ring(max(args.views, 1))[0], "image_width": width, "image_height": height}

        def fwd_step(i=0):
            with torch.no_grad():
                backend.render_with_depth(params, viewpoint, bg_color=bg[0], render_mode="RGB", device=str(dev), sh_degree=deg)

        fwd_only = timed(fwd_step, args.steps, 3)
    # mean record counts over the cameras of the headline ring (its whole-step roofline is priced with THEIR intersections)
    ring_M, ring_binned = count_records(list(ring(head_ring))) if rank == 0 else (None, None)
    fixed_M = (count_records(list(ring(world * B)))[0] if (head_ring != world * B and not args.no_extras) else ring_M) if rank == 0 else None
    # ---- the optimisation step around the rasterizer (BASELINE config C3 "full Adam loop"): the fused harness step -- loss,
    # backward, SH rebuild, Adam -- on the same scene and views; NOT part of `value`
    train = None
    if world == 1 and args.workload in ("C3", "T") and not args.no_train_step and not args.no_extras:
        # (no reset of what the passes above remembered: every pipeline / backend speculates on its own memory of the frame size)
        train = train_step_leg(sc, n, width, height, deg, B, dev, max(5, min(args.steps, 20)))
    copy_peak = measured_copy_peak(dev) if rank == 0 else 0.0   # after the timed regions

    ex = head_step.exchange
    if dist is not None:   # per rank, to stderr: what a sweep over NCCL_ALGO / NCCL_PROTO needs to be read without code changes
        tail = stage_ms.get("exchange_tail", (0, 0.0))
        print(f"[bench] rank {rank}/{dist.get_world_size()} backend={dist.get_backend()} rccl={rccl_version()} "
              f"form={args.exchange} views_per_gpu={B} bytes_sent_per_step={ex.bytes_sent // max(1, ex.steps)} "
              f"exchange_tail_ms(1 view/step)={tail[1] / max(tail[0], 1):.4f} "
              f"NCCL_ALGO={os.environ.get('NCCL_ALGO', '(unset)')} NCCL_PROTO={os.environ.get('NCCL_PROTO', '(unset)')} "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '(unset)')}",
              file=sys.stderr, flush=True)
    if rank == 0:
        m = info["M"]
        ms_per_step = head["s"] / args.steps * 1e3
        value = world * B * p * args.steps / head["s"] / 1e6
        ms_per_view = ms_per_step / B
        alg = algorithmic_bytes(n, m, p, deg)
        # per C-ABI stage: average device time per launch (binning = depth ordering + record generation + tile passes)
        per, launches = per_stage(stage_ms, args.steps)
        # the dominant KERNEL: "binning" is a stage of several short launches, the other four stages are one launch each
        dominant = max((k for k in per if k in alg and k != "binning"), key=lambda k: per[k])
        ach = alg[dominant] / (per[dominant] * 1e-3) / 1e9
        total_bytes = sum(alg.values())                                    # of camera 0's view (the single-view protocol)
        ring_bytes = sum(algorithmic_bytes(n, ring_M, p, deg).values())    # mean over the headline ring's cameras
        # measured per launch, from separate rocprofv3 --pmc passes of the single-view step (profiles/*_traffic.json,
        # tools/make_traffic_json.py): HBM bytes of every kernel, VALU utilisation, float-atomic requests, and the
        # bound the counters point at.  NOT measured in this run: `counters_from` names the file and the sources' revision.
        traffic, counters, counters_from, measured_bytes, step_bytes, valu_roofline = None, {}, None, None, None, None
        for fname in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as fh:
                    doc = json.load(fh)
                if args.workload == "C3":
                    counters = doc["kernels"][dominant]
                    traffic = counters["hbm_bytes_per_launch"]
                    measured_bytes = sum(k.get("hbm_bytes_per_launch", 0) * k.get("launches_per_view", 1)
                                         for grp in ("kernels", "binning_kernels") for k in doc.get(grp, {}).values())
                    # the kernels of the MULTI-VIEW step (per-step projection launches: 1 / views of a launch per view)
                    step_bytes = sum(k.get("hbm_bytes_per_launch", 0) * k.get("launches_per_view", 1)
                                     for k in doc.get("step_kernels", {}).values()) or None
                    counters_from = f"profiles/{fname} @ csrc {doc.get('csrc_rev', 'of round ' + fname[1:3])}"
                    # the VALU roofline of the two sweeps (the counters' file; tools/r05_valu_pmc.sh): wave-instructions per
                    # launch, the launch in SIMD-cycles, busy cycles of the vector ALU, EXEC-active lanes per instruction
                    keep = ("wave_insts", "simd_cycles", "cycles_per_inst", "busy_cycles_per_inst", "lane_utilisation",
                            "issue_frac", "issue_frac_at_2_cycles_per_inst", "lds_array_busy", "clock_GHz")
                    valu_roofline = {k: {f: v["valu_roofline"].get(f) for f in keep} for k, v in doc["kernels"].items()
                                     if isinstance(v, dict) and v.get("valu_roofline")} or None
                    break
            except (OSError, KeyError, ValueError, TypeError):
                continue
        stage_gbs = {k: round(alg[k] / (per[k] * 1e-3) / 1e9, 1) for k in sorted(per) if k in alg}
        solo_ms = solo["s"] / args.steps * 1e3
        line = {
            "metric": "Mpix/s fwd+bwd @1M gaussians/1080p", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} gaussians, {width}x{height}, SH degree {deg}, "
                                   f"{B} view{'s' if B > 1 else ''} per GPU per step ({world * B} consecutive cameras of a ring of "
                                   f"{head_ring} per step, the window moving on every step), "
                                   f"forward+backward into one 59N-float gradient bucket"
                                   + ("; speculation checked per view (no host group)" if flag_error and world > 1 else "")
                                   + ((", RCCL all-gather of 3N rgb grads per view + one all-reduce of 11N geometry grads per "
                                       "step, SH gradient rebuilt locally" if args.exchange == "factored" else
                                       ", RCCL all-reduce of 59N-float grads") if world > 1 else ""),
                       "views_per_gpu_per_step": B, "view_streams": n_streams,
                       "overlap": ("alternate" if n_streams > 1 else "none"),
                       # what the views of a step share: one projection launch (and, in one process, one projection-backward
                       # launch) for all of them -- gsplat_amd.project_cameras / project_cameras_backward
                       "projection": ("per step: %s" % ("forward + backward" if head_step.batch_bwd else
                                                         "forward + geometry backward (colour gradients out of the sweeps)"
                                                         if head_step.rank_batch else "forward"))
                                     if head_step.batch_proj else "per view",
                       "speculation_check": ("per step, repairs: %d" % head_step.pipe.repairs) if B > 1 and not args.no_defer_check else "per view",
                       "ms_per_view": round(ms_per_view, 4), "ring_cameras": head_ring,
                       "intersections_M_ring_mean": round(ring_M, 1), "binned_records_ring_mean": round(ring_binned, 1),
                       "intersections_M": m, "bin_tile": stages.bin_tile_for(width, height, stages.DEFAULT_CULL_MODE if args.cull_mode is None else args.cull_mode), "binned_records": info.get("binned"),
                       "grad_l1": round(grad_l1_head, 3), "grad_l1_single_view": round(grad_l1, 3),
                       "cull_mode": stages.DEFAULT_CULL_MODE if args.cull_mode is None else args.cull_mode,
                       # replica gradient rows for gaussians that cover >= 24 tile columns (ABI 14; none in this scene from outside)
                       "hot_splats": bool(stages._HOT_SPLATS)},
            # the same step on the SAME gpus x views cameras every step (r04's protocol: remembered capacities / key ranges
            # never move), and what the moving window cost the timed region in speculation
            "value_fixed_cameras": round(world * B * p * args.steps / fixed["s"] / 1e6, 3),
            "ms_per_view_fixed_cameras": round(fixed["s"] / args.steps * 1e3 / B, 4),
            "speculation": {"repaired_steps": head["repaired_steps"], "views": head["stats"]["views"],
                            "capacity_misses_per_100": round(100.0 * head["stats"]["capacity_misses"] / max(head["stats"]["views"], 1), 2),
                            "key_range_repairs_per_100": round(100.0 * head["stats"]["key_range_repairs"] / max(head["stats"]["views"], 1), 2)},
            "warmup_effective": head["warmup_effective"],   # steps actually run before the timed window (see settle())
            "step_ms": percentiles(dist_ms),
            # (where the slow steps of the per-step pass sit: their indices -- a burst, or spread out?)
            "slow_steps": [i for i, x in enumerate(dist_ms) if x > 1.25 * sorted(dist_ms)[len(dist_ms) // 2]][:16],
            "step_ms_first_20": [round(x, 3) for x in dist_ms[:20]],
            # how long per step the host sat waiting for the device's control words (it runs that far ahead)
            "host_wait_ms_per_step": round(head["host_wait_ms"], 4),
            # ONE view per step on one stream with a fixed camera -- the serial latency of a view, what rounds 1 and 2
            # reported as `value`; every per-stage figure below is of THIS step
            "single_view": {"ms": round(solo_ms, 4), "value": round(world * p / solo_ms / 1e3, 3),
                            "warmup_effective": solo["warmup_effective"]},
            # `bound`: the roofline `achieved` / `peak` are priced against is HBM (no MFMA on this path);
            # `limiter`: what the PMC counters say the dominant kernel actually waits for
            "roofline": {"protocol": "single_view", "bound": "hbm", "kernel": dominant, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "limiter": counters.get("limiter"), "valu_util": counters.get("valu_util"),
                         "atomic_requests": counters.get("atomic_requests"), "counters_from": counters_from,
                         "valu_roofline": valu_roofline,
                         "measured_copy_peak": round(copy_peak, 1),
                         "algorithmic_bytes_per_launch": alg[dominant], "avg_launch_ms": round(per[dominant], 4)},
            # SURVEY 8d's compulsory bytes of one view over the time of one view of the timed region
            # (the headline's views are the ring's cameras: priced with THEIR mean intersections)
            "roofline_whole_step": {"algorithmic_bytes_per_view": int(ring_bytes),
                                    "achieved_GBs": round(ring_bytes / (ms_per_view * 1e-3) / 1e9, 2),
                                    "frac": round(ring_bytes / (ms_per_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "frac_fixed_cameras": round(sum(algorithmic_bytes(n, fixed_M, p, deg).values())
                                                                / (fixed["s"] / args.steps / B) / 1e9 / HBM_PEAK_GBS, 5),
                                    "frac_single_view": round(total_bytes / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms_protocol": "single_view",
            "stage_ms": {k: round(v, 4) for k, v in sorted(per.items())},
            "stage_abi_calls": {k: round(v, 2) for k, v in sorted(launches.items())},   # (C-ABI calls per view; binning: 3 calls, 12 kernels)
            "stage_algorithmic_GBs": stage_gbs,
            # a stage priced above the HBM peak is not moving SURVEY's bytes (it avoids them: digest instead of SH rows,
            # bins instead of tiles, early-out) -- the contractual fraction then flatters; see roofline_measured
            "over_peak": sorted(k for k, v in stage_gbs.items() if v > HBM_PEAK_GBS)
                         + (["whole_step"] if ring_bytes / (ms_per_view * 1e-3) / 1e9 > HBM_PEAK_GBS else []),
        }
        if measured_bytes:
            # counter bytes of the SINGLE-VIEW step's kernels (profiles/*_traffic.json), over the time of a view of the
            # pipelined step (`frac_of_peak`: the bytes are the same, the kernels overlap) and of a view alone
            # (the multi-view step's own kernels where the counter file has them: its per-step projection launches move
            # fewer bytes per view than the single-view step's)
            in_step = step_bytes if (step_bytes and head_step.batch_bwd) else measured_bytes
            gbs = in_step / (ms_per_view * 1e-3) / 1e9
            gbs1 = measured_bytes / (solo_ms * 1e-3) / 1e9
            line["roofline_measured"] = {"hbm_bytes_per_view": int(in_step), "hbm_bytes_per_view_single_view": int(measured_bytes),
                                         "achieved_GBs": round(gbs, 1),
                                         "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                                         "frac_of_copy_ceiling": round(gbs / max(copy_peak, 1.0), 4),
                                         "achieved_GBs_single_view": round(gbs1, 1),
                                         "frac_of_peak_single_view": round(gbs1 / HBM_PEAK_GBS, 4),
                                         "counters_from": counters_from}
            line["roofline_whole_step"]["frac_measured"] = round(gbs / HBM_PEAK_GBS, 4)
        if rot is not None:
            st = rot["stats"]
            per100 = 100.0 / max(st["views"], 1)
            line["rotating"] = {"views": args.views, "views_per_step": 1,
                                "value_rotating": round(p * args.steps / rot["s"] / 1e6, 3),
                                "ms_per_view": round(rot["s"] / args.steps * 1e3, 4),
                                "stage_ms": {k: round(v, 4) for k, v in sorted(rot["stage_ms"].items())},
                                "intersections_M": round(rot["M"], 1), "binned_records": round(rot["binned"], 1),
                                "capacity_misses_per_100": round(st["capacity_misses"] * per100, 2),
                                "key_range_repairs_per_100": round(st["key_range_repairs"] * per100, 2)}
        if wide is not None:
            st = wide["stats"]
            per100 = 100.0 / max(st["views"], 1)
            line["wide_depth"] = {"what": f"one view per step cycling {args.views} cameras on a ring of radius 0.35 INSIDE the cloud",
                                  "key_span_octaves": wide["key_span_octaves"],
                                  "ms_per_view": round(wide["s"] / args.steps * 1e3, 4),
                                  "value": round(p * args.steps / wide["s"] / 1e6, 3),
                                  "depth_ordering_paths": wide["paths"],
                                  "stage_ms": {k: round(v, 4) for k, v in sorted(wide["stage_ms"].items())},
                                  "intersections_M": round(wide["M"], 1), "binned_records": round(wide["binned"], 1),
                                  "capacity_misses_per_100": round(st["capacity_misses"] * per100, 2),
                                  "key_range_repairs_per_100": round(st["key_range_repairs"] * per100, 2)}
        if rank_step is not None:
            rs_ms = rank_step["s"] / args.steps * 1e3 / B
            line["rank_step"] = {"what": "the step a rank of an N > 1 run executes (shared projection forward, per-view factored "
                                         "projection backward, local SH rebuild; no collectives in this process)",
                                 "ms_per_view": round(rs_ms, 4), "value": round(p / rs_ms / 1e3, 3),
                                 "frac": round(ring_bytes / (rs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        if fwd_only is not None:
            fo_ms = fwd_only["s"] / args.steps * 1e3
            line["forward_only"] = {"what": "HipBackend.render_with_depth under no_grad (the reference's validation / eval callers, "
                                            "trainer.py:417-424, eval.py:171-183), one view per call, fixed camera",
                                    "ms": round(fo_ms, 4), "Mpix/s": round(p / fo_ms / 1e3, 3)}
        if train is not None:
            line["train_step"] = train
        if dist is not None:
            line["agree_ms_per_step"] = round(head["agree_ms"], 4)
            if c4 is not None:
                line["value_c4"] = round(world * p * args.steps / c4["s"] / 1e6, 3)
                line["c4"] = {"what": "BASELINE config C4 as written: ONE view per GPU per step (a %d-view batch)" % world,
                              "ms_per_step": round(c4["s"] / args.steps * 1e3, 4),
                              "exchange_tail_ms": round(c4["exchange_tail_ms"], 4),
                              "exchange_tail_ms_at_%d_views_per_gpu" % B: round(c4["exchange_tail_ms_B"], 4)}
            other_value = world * B * p * args.steps / other["s"] / 1e6
            fact = args.exchange == "factored"
            tail = stage_ms.get("exchange_tail", (0, 0.0))
            line["exchange"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl": rccl_version(),
                                "form": args.exchange,
                                "exchange_tail_ms_single_view": round(tail[1] / max(tail[0], 1), 4),
                                "value_factored": round(value if fact else other_value, 3),
                                "value_allreduce": round(other_value if fact else value, 3),
                                "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO")}
        if world == 1 and not args.no_cpu_baseline and args.workload == "C3":   # (the oracle: bench.py's cpu_baseline leg only)
            print(f"[bench] GPU leg done ({ms_per_view:.3f} ms/view); timing the CPU oracle on a bounded sample ...",
                  file=sys.stderr, flush=True)
            line["cpu_baseline"] = cpu_baseline(n, width, height, deg)
            line["cpu_baseline"]["cpu_model"] = cpu_model()
            print("[bench] CPU oracle at C1 / C2 sizes, fwd and fwd+bwd, median of 5 ...", file=sys.stderr, flush=True)
            line["cpu_baseline"]["protocol"] = cpu_baseline_protocol()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
