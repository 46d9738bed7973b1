"""Developer aid: what the HOST spends per view queuing the bench's step (C3, forward + backward through
gsplat_amd.rasterization): wall time of the enqueue loop with the device far behind or ahead, and a cProfile of it.
usage (GPU box): python tools/host_profile.py [views]"""
import cProfile, os, pstats, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
import gsplat_amd
from gsplat_amd import scene

views = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000      # (a tiny scene shows the host's own cost: same calls, no device time)
W, H, deg = (int(sys.argv[3]), int(sys.argv[4]), 3) if len(sys.argv) > 4 else (1920, 1080, 3)
dev = torch.device("cuda", 0)
sc = scene.make_scene(n, seed=0, sh_degree=deg)
names = ("means3D", "rotations", "scales", "opacities", "shs")
leaves = [sc[k].to(dev).requires_grad_(True) for k in names]
grads = [torch.zeros_like(t) if k != "opacities" else torch.zeros(n, device=dev) for k, t in zip(names, leaves)]
K = scene.make_intrinsics(W, H).to(dev)[None]
cams = torch.stack([scene.make_camera(i, 4) for i in range(4)]).to(dev)
bg = torch.zeros(1, 3, device=dev)
v_render = scene.make_v_render(W, H, seed=2).to(dev)[None]

def view(j):
    r, a, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], leaves[3].squeeze(-1), leaves[4], cams[j % 4][None], K,
                                          W, H, sh_degree=deg, backgrounds=bg, grad_out=grads, grad_accumulate=False)
    r.backward(v_render)

from gsplat_amd import stages as S
acc = {}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = acc.setdefault(label or name, [0, 0.0]); d[0] += 1; d[1] += time.perf_counter() - t
    setattr(obj, name, w)
if os.environ.get("HOST_PROFILE_SECTIONS", "1") != "0":
    for nm in ("project_fwd", "bin_tiles", "tile_schedule", "composite_fwd", "composite_bwd", "project_bwd", "key_range_buffer",
               "_wait_slot", "_pinned_scalar", "f32c", "_call"):
        wrap(S, nm)
    wrap(S.ViewChain, "forward", "ViewChain.forward")
    wrap(S.ViewChain, "backward", "ViewChain.backward")
    wrap(S, "_view_layout")
    wrap(torch, "empty", "torch.empty")
    wrap(gsplat_amd, "rasterization", "rasterization()")
    wrap(torch.Tensor, "backward", "Tensor.backward")
for j in range(20): view(j)
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for j in range(views): view(j)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / views:.4f} ms per view; with the final wait {1e3 * (t2 - t0) / views:.4f} ms per view", flush=True)
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:22s} {c / views:5.1f} calls / view  {1e6 * t / views:8.1f} us / view")
pr = cProfile.Profile()
pr.enable()
for j in range(views): view(j)
pr.disable()
torch.cuda.synchronize()
if os.environ.get("HOST_PROFILE_CPROFILE", "0") != "0":
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
