#!/usr/bin/env python3
"""bench.py -- Mpix/s of rasterizer forward+backward on the synthetic 1M-gaussian / 1080p scene
(BASELINE.json metric; SURVEY.md section 8d workload "C3"), 1..8 MI355X, one process per GPU.

A step = one pass of the hot path over one view per rank: rasterization forward (project, bin, composite)
+ the synthetic upstream gradient v_render + full backward to the five input gradients; with N > 1 ranks the
view-parallel exchange (RCCL all-reduce of the 59*N-float gaussian gradients) is inside the step.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no launcher environment starts the N ranks itself (a child
``torch.distributed.run`` on 127.0.0.1, before this process touches the GPU) and relays rank 0's JSON line.
"""

from __future__ import annotations

import argparse
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
WORKLOADS = {
    # name: (gaussians, width, height, sh_degree)
    "C3": (1_000_000, 1920, 1080, 3),
    "C2": (100_000, 1920, 1080, 3),
    "C5": (5_000_000, 3840, 2160, 3),
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


def self_launch(args) -> int:
    """``--gpus N`` without a launcher environment: start N ranks with torch.distributed.run as a CHILD process
    (nothing in this process has touched the GPU yet), relay its output, return its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def percentiles(xs):
    xs = sorted(xs)
    pick = lambda q: xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]  # noqa: E731
    return {"mean": round(sum(xs) / len(xs), 4), "median": round(pick(0.5), 4), "p10": round(pick(0.1), 4),
            "p90": round(pick(0.9), 4)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.14 s of device time: a 50-step window moved by +-1.5 %
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--cull-mode", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="factored", choices=("factored", "allreduce"),
                    help="N > 1: how ranks sum the gaussian gradients (see DESIGN.md section 6)")
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import gsplat_amd
    from gsplat_amd import scene, stages

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
    # weak scaling: rank r renders view r of a ring of `world` cameras (view 0 = the single-GPU view)
    vm = scene.make_camera(rank, world).to(dev)[None]
    v_render = scene.make_v_render(width, height, seed=2).to(dev)[None]
    bg = torch.zeros(1, 3, device=dev)
    info = {}
    # N > 1, factored exchange: 48 of the 59 gradient floats per gaussian are the SH-coefficient gradient, which is
    # the outer product basis(view direction) x (d loss / d rgb); every rank knows every rank's camera, so ranks
    # all-reduce the 11 N geometry floats, all-gather 3 N rgb-gradient floats each and rebuild the coefficient
    # gradient locally (summed in rank order: bit-identical on every rank).
    factored = dist is not None and args.exchange == "factored"
    if dist is not None:
        vm_all = torch.stack([scene.make_camera(r, world) for r in range(world)]).to(dev)
        vr_all = torch.zeros(world, n, 3, device=dev)
        vr_mine = vr_all[rank]

    def step(factored=factored):
        for t in leaves:
            t.grad = None
        # the backward pass writes every gradient element straight into the bucket (no zero-fill, no copy)
        # (opacities [N,1] -> [N] as the reference backend does it, gsplat_backend.py:157-158)
        r, a, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], opac.squeeze(-1), leaves[4], vm, K, width,
                                              height, sh_degree=deg, backgrounds=bg, cull_mode=args.cull_mode,
                                              grad_out=grad_views, sh_grad_factored=vr_mine if factored else None)
        r.backward(v_render)
        if dist is not None and stages.TIMER.enabled:   # instrumented pass: time the whole exchange on the stream
            x0 = torch.cuda.Event(enable_timing=True)
            x0.record()
        if factored:
            # gather first, so the local rebuild of the SH gradient runs beside the geometry all-reduce
            if dist.get_backend() == "nccl":
                gathered = dist.all_gather_into_tensor(vr_all, vr_mine, async_op=True)
            else:
                gathered = dist.all_gather(list(vr_all.unbind(0)), vr_mine.clone(), async_op=True)
            summed = dist.all_reduce(flat_grad[:11 * n], async_op=True)
            gathered.wait()
            stages.sh_grad_expand(leaves[0], deg, vm_all, K, vr_all, grad_views[4])
            summed.wait()
        elif dist is not None:  # one all-reduce of the whole 59 N-float bucket (RCCL over xGMI)
            dist.all_reduce(flat_grad)
        if dist is not None and stages.TIMER.enabled:
            x1 = torch.cuda.Event(enable_timing=True)
            x1.record()
            stages.TIMER.records.append(("exchange", x0, x1))
        if "M" not in info:   # (once: with 32 x 32 bins in the hot path the 16 x 16 record count is a reduction + read-back)
            info["M"] = meta["n_isects"]
            info["binned"] = meta["n_binned"]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(form, warmup, per_step=False):
        """`warmup` untimed steps, then exactly `steps` steps between two fences.  per_step: one event per step
        boundary on the launch stream gives the per-step device times (median / p10 / p90) -- in a pass of its own:
        an event record leaves a ~6 us bubble on the stream (rocprof kernel trace), 0.8 % of a step."""
        for _ in range(warmup):
            step(form)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if per_step else None
        fence()
        stages.WAIT_STATS[:] = [0, 0.0]
        t0 = time.perf_counter()
        if per_step:
            marks[0].record()
        for i in range(args.steps):
            step(form)
            if per_step:
                marks[i + 1].record()
        if not per_step and form == factored:   # (the headline pass; before the closing fence)
            info["host_wait_ms"] = stages.WAIT_STATS[1] / max(args.steps, 1) * 1e3
        fence()
        dt = time.perf_counter() - t0
        return dt, ([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)] if per_step else None)

    # the timed region `value` comes from: the default exchange form of this run
    elapsed, _ = timed(factored, args.warmup)
    _, per_step_ms = timed(factored, 0, per_step=True)   # the same steps again, one event per step: the distribution
    other_elapsed = None
    if dist is not None:   # the same steps with the other exchange form, for the comparison figure only
        other_elapsed, _ = timed(not factored, args.warmup)
        for _ in range(2):
            step()
    # a second, instrumented pass of the same steps gives the per-stage device times for the roofline object
    # (HIP events around every C-ABI stage on the launch stream; the event records themselves cost a few percent,
    # which is why this pass is not the one `value` is computed from)
    stages.TIMER.reset(enabled=True)
    for _ in range(args.steps):
        step()
    fence()
    stage_ms = {k: (c, t) for k, (c, t) in stages.TIMER.summary().items()}
    stages.TIMER.reset(enabled=False)

    if dist is not None:
        tmax = torch.tensor([elapsed, other_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, other_elapsed = (float(x) for x in tmax.tolist())

    grad_l1 = float(flat_grad.double().abs().sum())   # of the last step's exchanged gradient (same on every rank)
    copy_peak = measured_copy_peak(dev) if rank == 0 else 0.0   # after the timed regions
    if rank == 0:
        m, p = info["M"], width * height
        ms_per_step = elapsed / args.steps * 1e3
        value = world * p * args.steps / elapsed / 1e6
        alg = algorithmic_bytes(n, m, p, deg)
        # per C-ABI stage: average device time per step (binning = sorts + scan + emit + offsets)
        per = {}
        for name, (calls, total) in stage_ms.items():
            other = {"gsa_sh_grad_expand": "sh_grad_rebuild", "gsa_composite_bwd_prepare": "bwd_prepare"}
            key = name if name in alg or name == "exchange" else other.get(name, "binning")   # (exchange incl. rebuild)
            per[key] = per.get(key, 0.0) + total / args.steps
        # the dominant KERNEL: "binning" is a stage of ~24 short launches (sorts, scan, emit, offsets), the other
        # four stages are one kernel launch each
        dominant = max((k for k in per if k in alg and k != "binning"), key=lambda k: per[k])
        ach = alg[dominant] / (per[dominant] * 1e-3) / 1e9
        total_bytes = sum(alg.values())
        # measured per launch of the dominant kernel, from separate rocprofv3 --pmc passes of this command
        # (profiles/*_traffic.json, tools/make_traffic_json.py): HBM bytes, VALU utilisation, float-atomic requests,
        # and the bound the counters point at
        traffic, counters = None, {}
        for fname in ("r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as fh:
                    if args.workload == "C3":
                        counters = json.load(fh)["kernels"][dominant]
                        traffic = counters["hbm_bytes_per_launch"]
                        break
            except (OSError, KeyError, ValueError):
                continue
        line = {
            "metric": "Mpix/s fwd+bwd @1M gaussians/1080p", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} gaussians, {width}x{height}, SH degree {deg}, "
                                   f"1 view per GPU per step, forward+backward"
                                   + ((", RCCL all-reduce of 11N geometry grads + all-gather of 3N rgb grads per rank,"
                                       " SH gradient rebuilt locally" if factored else
                                       ", RCCL all-reduce of 59N-float grads") if world > 1 else ""),
                       "intersections_M": m, "bin_tile": stages.DEFAULT_BIN_TILE, "binned_records": info.get("binned"),
                       "grad_l1": round(grad_l1, 3), "cull_mode": stages.DEFAULT_CULL_MODE if args.cull_mode is None else args.cull_mode},
            # `bound`: the roofline `achieved` / `peak` are priced against is HBM (no MFMA on this path);
            # `limiter`: what the PMC counters say the dominant kernel actually waits for
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "limiter": counters.get("limiter"), "valu_util": counters.get("valu_util"),
                         "atomic_requests": counters.get("atomic_requests"),
                         "measured_copy_peak": round(copy_peak, 1),
                         "algorithmic_bytes_per_launch": alg[dominant], "avg_launch_ms": round(per[dominant], 4)},
            "step_ms": percentiles(per_step_ms),
            # how long per step the host sat waiting for the device's control words (it runs that far ahead)
            "host_wait_ms_per_step": round(info.get("host_wait_ms", 0.0), 4),
            "roofline_whole_step": {"algorithmic_bytes": total_bytes,
                                    "achieved_GBs": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                                    "frac": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms": {k: round(v, 4) for k, v in sorted(per.items())},
            "stage_algorithmic_GBs": {k: round(alg[k] / (per[k] * 1e-3) / 1e9, 1) for k in sorted(per) if k in alg},
        }
        if dist is not None:
            other_value = world * p * args.steps / other_elapsed / 1e6
            line["exchange"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                "form": "factored" if factored else "allreduce",
                                "exchange_ms": round(per.get("exchange", 0.0), 4),
                                "value_factored": round(value if factored else other_value, 3),
                                "value_allreduce": round(other_value if factored else value, 3)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "C3":
            print(f"[bench] GPU leg done ({ms_per_step:.3f} ms/step); timing the CPU oracle on a bounded sample ...",
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
