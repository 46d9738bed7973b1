"""Developer aid: per-wave timeline of the two compositor kernels (needs a library built with -DGSA_TRACE:
tools/build_variant.sh trace -DGSA_TRACE; GSPLAT_AMD_LIB=.../ab_trace.so python tools/trace_tiles.py)."""
import ctypes, os, sys
import numpy as np
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
import gsplat_amd
from gsplat_amd import scene, _lib

n, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device("cuda", 0)
sc = scene.make_scene(n, seed=0, sh_degree=deg)
leaves = [sc[k].to(dev).requires_grad_(True) for k in ("means3D", "rotations", "scales", "opacities", "shs")]
K = scene.make_intrinsics(W, H).to(dev)[None]; vm = scene.make_camera(0, 1).to(dev)[None]
v = scene.make_v_render(W, H, seed=2).to(dev)[None]; bg = torch.zeros(1, 3, device=dev)
buf = torch.zeros(2 * 262144 * 16, dtype=torch.int64, device=dev)
def step():
    for t in leaves: t.grad = None
    r, a, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm, K, W, H,
                                          sh_degree=deg, backgrounds=bg)
    r.backward(v)
for _ in range(3): step()
fn = _lib.lib.gsa_debug_set_trace; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf.data_ptr()) == 0
torch.cuda.synchronize(); step(); torch.cuda.synchronize()
tr = buf.cpu().numpy().reshape(2, 262144, 16)
for k, name in enumerate(("fwd", "bwd")):
    d = tr[k]; d = d[d[:, 1] > 0]
    if not len(d):   # (the two-phase backward kernel carries no trace marks; GSPLAT_AMD_BWD_MODE=0 / 1 do)
        print(f"== {name}: no trace records")
        continue
    t0, t1, hw, work = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64), d[:, 2], d[:, 3]
    base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us (100 MHz counter)
    dur = t1 - t0
    xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"== {name}: {len(d)} waves, span {t1.max():.1f} us, sum of WG time {dur.sum():.0f} us, "
          f"mean {dur.mean():.2f} us, p50 {np.percentile(dur,50):.2f}, p90 {np.percentile(dur,90):.2f}, "
          f"p99 {np.percentile(dur,99):.2f}, max {dur.max():.2f}")
    print(f"   distinct CUs seen {len(np.unique(cuid))}, avg concurrent WGs {dur.sum()/t1.max():.1f} "
          f"(= {dur.sum()/t1.max()/max(1,len(np.unique(cuid))):.2f} per CU)")
    # concurrency over time in 10 slices
    edges = np.linspace(0, t1.max(), 11)
    conc = [(np.minimum(t1, b) - np.maximum(t0, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   concurrency per tenth of the span:", " ".join(f"{c:.0f}" for c in conc))
    started = [int(((t0 >= a) & (t0 < b)).sum()) for a, b in zip(edges[:-1], edges[1:])]
    print("   WGs started per tenth:", started)
    # duration vs work
    order = np.argsort(work)
    for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 0.95), (0.95, 1.0)):
        sel = order[int(lo * len(d)):int(hi * len(d))]
        print(f"   work quantile {lo:.2f}-{hi:.2f}: list len {work[sel].mean():.0f}, WG time {dur[sel].mean():.2f} us")
    acc = d[:, 4:12] / 100.0; nb, nv = d[:, 12], d[:, 13]
    names = ["prologue", "stage (wait rows, LDS, test, list)", "-", "-", "walk", "-", "epilogue", "-"]
    print("   wave 0 phases, us per WG: " + ", ".join(f"{n} {acc[:, k].mean():.2f}" for k, n in enumerate(names[:7]))
          + f", unaccounted {(dur - acc.sum(1)).mean():.2f}")
    print(f"   batches/WG {nb.mean():.2f}, visits/WG (wave 0) {nv.mean():.1f}, walk per visit {1e3*acc[:,4].sum()/max(1,nv.sum()):.0f} ns")
    percu = np.bincount(cuid, weights=dur, minlength=cuid.max() + 1); percu = percu[percu > 0]
    print(f"   per-CU busy WG-time: min {percu.min():.0f} mean {percu.mean():.0f} max {percu.max():.0f} us")
