"""Developer aid: per-workgroup phase times of the binning kernels (library built with -DGSA_TRACE:
tools/build_variant.sh trace -DGSA_TRACE; GSPLAT_AMD_LIB=.../ab_trace.so python tools/trace_sort.py)."""
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
leaves = [sc[k].to(dev) for k in ("means3D", "rotations", "scales", "opacities", "shs")]
K = scene.make_intrinsics(W, H).to(dev)[None]; vm = scene.make_camera(0, 1).to(dev)[None]
bg = torch.zeros(1, 3, device=dev)
NK = 4
buf = torch.zeros(NK * 32768 * 16, dtype=torch.int64, device=dev)
def step():
    with torch.no_grad():
        gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm, K, W, H, sh_degree=deg, backgrounds=bg)
for _ in range(3): step()
fn = ctypes.CDLL(_lib.LIB_PATH).gsa_debug_set_sort_trace; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf.data_ptr()) == 0
torch.cuda.synchronize(); step(); torch.cuda.synchronize()
tr = buf.cpu().numpy().reshape(NK, 32768, 16)
names = {0: ("radix_scatter<u32> (last call)", ["prologue", "ranking", "reorder", "stores"]),
         1: ("radix_scatter<u16>", ["prologue", "ranking", "reorder", "stores"]),
         2: (("emit_place", ["loads + owners", "generation", "bases + ranking", "wave ranks", "stores"])
             if os.environ.get("GSPLAT_AMD_BIN_DIRECT", "1") != "0" else
             ("emit_scatter", ["hist columns", "owners", "generation", "ranking", "reorder", "stores"])),
         3: ("depth_bucket_sort", ["bounds + loads", "local passes", "gathers", "count scan", "wait in front", "offsets"])}
for k, (name, phases) in names.items():
    d = tr[k]; d = d[d[:, 1] > 0]
    if not len(d): continue
    t0, t1 = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
    span = (t1.max() - t0.min()) / 100.0
    dur = (t1 - t0) / 100.0
    acc = d[:, 2:2 + len(phases)] / 100.0
    print(f"== {name}: {len(d)} workgroups, span {span:.1f} us, mean WG time {dur.mean():.2f} us (p90 {np.percentile(dur, 90):.2f}), "
          f"avg concurrent {dur.sum() / span:.0f}")
    st = (t0 - t0.min()) / 100.0
    print(f"   WG start after the first, us: p10 {np.percentile(st, 10):.1f} p50 {np.percentile(st, 50):.1f} p90 {np.percentile(st, 90):.1f} "
          f"max {st.max():.1f}; duration of the earliest / latest tenth: {dur[st <= np.percentile(st, 10)].mean():.1f} / "
          f"{dur[st >= np.percentile(st, 90)].mean():.1f}")
    print("   phases, us per WG: " + ", ".join(f"{p} {acc[:, i].mean():.2f}" for i, p in enumerate(phases))
          + f"; tail (stores drain) {(dur - acc.sum(1)).mean():.2f}")
