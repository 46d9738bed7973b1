#!/usr/bin/env python3
"""Stream schedules for the B views of a step, measured on the stage API (no autograd): what does the order in which the
views' front ends (projection + binning), sweeps and projection backwards are queued buy on one MI355X?

  alt       whole views alternate on S streams (pipeline.ViewPipeline's scheme)
  phased    all front ends first, spread over F streams (latency-bound binning chains of different views beside each
            other and beside the other views' HBM-bound projection); then the sweeps of all views in view order on ONE
            stream; every view's projection backward on a side stream (under the next view's sweeps)
  serial    one stream

usage: python tools/sched_experiment.py [--views 4] [--steps 30] [--n 1000000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-gaussian-splatting_amd"))
import torch  # noqa: E402

from gsplat_amd import scene, stages as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--only", default=None)
    ap.add_argument("--sleep", type=float, default=0, help="device cycles to sleep in front of the fronts-only variants")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, W, H, deg, B = args.n, args.width, args.height, 3, args.views
    sc = scene.make_scene(n, seed=0, sh_degree=deg, device="cpu")
    means, quats, scales = sc["means3D"].to(dev), sc["rotations"].to(dev), sc["scales"].to(dev)
    opac, shs = sc["opacities"].to(dev).squeeze(-1).contiguous(), sc["shs"].to(dev)
    K = scene.make_intrinsics(W, H).to(dev).contiguous()
    cams = [scene.make_camera(i, B).to(dev).contiguous() for i in range(B)]
    v_render = scene.make_v_render(W, H, seed=2).to(dev).contiguous()
    bg = torch.zeros(3, device=dev)
    grad_out = [torch.zeros_like(means), torch.zeros_like(quats), torch.zeros_like(scales), torch.zeros_like(opac), torch.zeros_like(shs)]
    vr = torch.zeros(B, n, 3, device=dev)
    cull = S.DEFAULT_CULL_MODE
    bt = S.bin_tile_for(W, H, cull)

    emu = {"on": False, "cache": {}}

    def front(j, exact=False):
        if emu["on"] and j >= 2:   # EMULATION of a batched projection's cost: views 2, 3 reuse the records of views 0, 1
            splat, radii, tiles, keys, rects, aux, kr, qb = emu["cache"][j - 2]
            bins = S.bin_tiles(splat, radii, tiles, keys, W, H, cull, tile_rects=rects, exact=exact, bin_tile=bt,
                               key_range=kr, qbox=qb)
            order = S.tile_schedule(bins.tile_offsets, W, H, bt)
            return splat, radii, bins, order, aux
        kr, qb = S.key_range_buffer(n, dev), S.qbox_buffer(n, dev)
        splat, radii, tiles, keys, rects, aux = S.project_fwd(means, quats, scales, opac, shs, deg, cams[j], K, W, H, cull,
                                                              return_rects=True, return_aux=True, bin_tile=bt,
                                                              key_range=kr, qbox=qb)
        if emu["on"]:
            emu["cache"][j] = (splat, radii, tiles, keys, rects, aux, kr, qb)
        bins = S.bin_tiles(splat, radii, tiles, keys, W, H, cull, tile_rects=rects, exact=exact, bin_tile=bt,
                           key_range=kr, qbox=qb)
        order = S.tile_schedule(bins.tile_offsets, W, H, bt)
        return splat, radii, bins, order, aux

    def sweep_fwd(f):
        splat, radii, bins, order, aux = f
        return S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, W, H, 3, order, want_lists=True,
                               bin_tile=bt, gid_boxes=bins.boxed)

    def sweep_bwd(f, o):
        splat, radii, bins, order, aux = f
        render, _, final_T, last_id, lists = o
        return S.composite_bwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, final_T, last_id, v_render, None, W, H,
                               3, order, quad_lists=lists, bin_tile=bt, gid_boxes=bins.boxed)

    skip = {"back": False}

    def back(j, f, v_splat):
        if skip["back"]:
            return
        splat, radii, bins, order, aux = f
        S.project_bwd(means, quats, scales, opac, shs, deg, cams[j], K, W, H, splat, radii, v_splat, out=grad_out,
                      accumulate=j > 0, v_rgb_out=vr[j], sh_aux=aux)

    pool = [torch.cuda.Stream(device=dev) for _ in range(8)]

    def ev(stream=None):
        e = torch.cuda.Event()
        e.record(stream if stream is not None else torch.cuda.current_stream())
        return e

    keep = []   # tensors of the step in flight stay referenced until the step is queued (stream-ordered allocator reuse)

    def step_serial():
        fr = []
        for j in range(B):
            f = front(j)
            o = sweep_fwd(f)
            back(j, f, sweep_bwd(f, o))
            fr.append(f)
        return fr

    def step_alt(n_streams):
        main_s = torch.cuda.current_stream()
        start = ev()
        prev, fr, done = None, [], []
        for j in range(B):
            st = pool[j % n_streams]
            st.wait_event(start)
            with torch.cuda.stream(st):
                f = front(j)
                o = sweep_fwd(f)
                vs = sweep_bwd(f, o)
                if prev is not None:
                    st.wait_event(prev)
                back(j, f, vs)
                prev = ev(st)
                done.append(prev)
                keep.append((f, o, vs))
            fr.append(f)
        for e in done:
            main_s.wait_event(e)
        return fr

    def step_phased(n_front, n_back, lookahead=None):
        """fronts on pool[0:n_front]; sweeps on the current stream; project_bwd on pool[4:4+n_back]"""
        main_s = torch.cuda.current_stream()
        start = ev()
        fr, ready = [], []
        for j in range(B):
            st = pool[j % n_front]
            st.wait_event(start)
            with torch.cuda.stream(st):
                f = front(j)
                ready.append(ev(st))
            fr.append(f)
        prev, done = None, []
        for j in range(B):
            main_s.wait_event(ready[j])
            o = sweep_fwd(fr[j])
            vs = sweep_bwd(fr[j], o)
            swept = ev()
            keep.append((o, vs))
            if n_back == 0:
                back(j, fr[j], vs)
            else:
                st = pool[4 + j % n_back]
                st.wait_event(swept)
                if prev is not None:
                    st.wait_event(prev)
                with torch.cuda.stream(st):
                    back(j, fr[j], vs)
                    prev = ev(st)
                    done.append(prev)
        for e in done:
            main_s.wait_event(e)
        return fr

    def step_phased_la(n_front, n_back, ahead):
        """like phased, but only `ahead` fronts run before the first sweep; front j + ahead is queued behind sweep j's start"""
        main_s = torch.cuda.current_stream()
        start = ev()
        fr, ready = [None] * B, [None] * B

        def queue_front(j, after=None):
            st = pool[j % n_front]
            st.wait_event(start)
            if after is not None:
                st.wait_event(after)
            with torch.cuda.stream(st):
                fr[j] = front(j)
                ready[j] = ev(st)

        for j in range(min(ahead, B)):
            queue_front(j)
        prev, done = None, []
        for j in range(B):
            main_s.wait_event(ready[j])
            begun = ev()
            if j + ahead < B:
                queue_front(j + ahead, begun)
            o = sweep_fwd(fr[j])
            vs = sweep_bwd(fr[j], o)
            swept = ev()
            keep.append((o, vs))
            st = pool[4 + j % max(n_back, 1)]
            st.wait_event(swept)
            if prev is not None:
                st.wait_event(prev)
            with torch.cuda.stream(st):
                back(j, fr[j], vs)
                prev = ev(st)
                done.append(prev)
        for e in done:
            main_s.wait_event(e)
        return fr

    def step_fronts(n_front, what="front"):
        """only the front ends of the B views (what: front | proj | bin), on n_front streams"""
        main_s = torch.cuda.current_stream()
        if args.sleep:   # the host queues all chains while the device sleeps: what the DEVICE does with chains that are all there
            torch.cuda._sleep(int(args.sleep))
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        spans.append([start, None])
        fr, done = [], []
        for j in range(B):
            st = pool[j % n_front] if n_front > 0 else main_s
            st.wait_event(start)
            with torch.cuda.stream(st):
                if what == "front":
                    f = front(j)
                elif what == "proj":
                    kr, qb = S.key_range_buffer(n, dev), S.qbox_buffer(n, dev)
                    f = S.project_fwd(means, quats, scales, opac, shs, deg, cams[j], K, W, H, cull, return_rects=True,
                                      return_aux=True, bin_tile=bt, key_range=kr, qbox=qb)
                    keep.append((kr, qb, f))
                    f = None
                else:
                    splat, radii, tiles, keys, rects, kr, qb = projected[j]
                    bins = S.bin_tiles(splat, radii, tiles, keys, W, H, cull, tile_rects=rects, bin_tile=bt, key_range=kr, qbox=qb)
                    order = S.tile_schedule(bins.tile_offsets, W, H, bt)
                    f = (splat, radii, bins, order, None)
                done.append(ev(st))
            if f is not None:
                fr.append(f)
        for e in done:
            main_s.wait_event(e)
        spans[-1][1] = torch.cuda.Event(enable_timing=True)
        spans[-1][1].record()
        return fr

    spans = []
    projected = []
    for j in range(B):
        kr, qb = S.key_range_buffer(n, dev), S.qbox_buffer(n, dev)
        pr = S.project_fwd(means, quats, scales, opac, shs, deg, cams[j], K, W, H, cull, return_rects=True, return_aux=True,
                           bin_tile=bt, key_range=kr, qbox=qb)
        projected.append((pr[0], pr[1], pr[2], pr[3], pr[4], kr, qb))

    def emulated(fn):
        def run():
            emu["on"] = True
            try:
                return fn()
            finally:
                emu["on"] = False
        return run

    def noback(fn):
        def run():
            skip["back"] = True
            try:
                return fn()
            finally:
                skip["back"] = False
        return run

    variants = {
        "noP2 emuP1 alt2": noback(emulated(lambda: step_alt(2))),
        "emuP1 phased f2 b1": emulated(lambda: step_phased(2, 1)),
        "emuP1 alt2": emulated(lambda: step_alt(2)),
        "fronts s1": lambda: step_fronts(1),
        "fronts s2": lambda: step_fronts(2),
        "fronts s4": lambda: step_fronts(4),
        "proj s1": lambda: step_fronts(1, "proj"),
        "proj s4": lambda: step_fronts(4, "proj"),
        "bin s1": lambda: step_fronts(1, "bin"),
        "bin s2": lambda: step_fronts(2, "bin"),
        "bin s4": lambda: step_fronts(4, "bin"),
        "serial": step_serial,
        "alt2": lambda: step_alt(2),
        "alt3": lambda: step_alt(3),
        "phased f4 b1": lambda: step_phased(4, 1),
        "phased f2 b1": lambda: step_phased(2, 1),
        "phased f4 b2": lambda: step_phased(4, 2),
        "phased f4 b0": lambda: step_phased(4, 0),
        "phased f1 b1": lambda: step_phased(1, 1),
        "la2 f2 b1": lambda: step_phased_la(2, 1, 2),
        "la1 f1 b1": lambda: step_phased_la(1, 1, 1),
    }
    # first calls of the shape (exact sizes, learn the capacity and the key range), one view at a time
    for _ in range(3):
        for j in range(B):
            f = front(j)
            f[2].finalize()
    torch.cuda.synchronize()
    ref = None
    for name, fn in variants.items():
        if args.only and args.only not in name:
            continue

        def one():
            fr = fn()
            ok = all(f[2].finalize()[1] for f in fr)   # the step's speculation, checked behind the last view
            keep.clear()
            return ok

        for _ in range(args.warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bad = 0
        for _ in range(args.steps):
            bad += 0 if one() else 1
        torch.cuda.synchronize()
        if 'spans' in dir() and spans:
            dev_ms = sorted(a.elapsed_time(b) for a, b in spans[-args.steps:])
            print(f"   device span per step: median {dev_ms[len(dev_ms) // 2]:.4f} ms, min {dev_ms[0]:.4f}")
            spans.clear()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        g = float(grad_out[0].double().abs().sum() + grad_out[1].double().abs().sum() + vr.double().abs().sum())
        ref = g if ref is None else ref
        print(f"{name:14s} {dt:7.4f} ms per step  {dt / B:7.4f} ms per view  misses {bad}  grad_l1 {g:.6e} "
              f"(rel {abs(g - ref) / max(ref, 1e-30):.1e})", flush=True)


if __name__ == "__main__":
    main()
