"""Depth-bucket occupancy of the bucketed depth ordering (255 equal-width buckets over the view's own key range + 1/16
margin) for the cameras of a ring: the fullest bucket against the 8192-key LDS capacity.  python tools/bucket_stats.py [N] [V]"""
import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
from gsplat_amd import scene, stages as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
sc = scene.make_scene(n, seed=0, sh_degree=3)
d = {k: v.cuda().contiguous() for k, v in sc.items()}
K = scene.make_intrinsics(W, H).cuda()
for v in range(V):
    vm = scene.make_camera(v, V).cuda()
    splat, radii, tiles, keys = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"], 3, vm, K, W, H, 0)
    k = keys.long() & 0xFFFFFFFF
    k = k[k != 0xFFFFFFFF]
    lo, hi = int(k.min()), int(k.max())
    a, scale = S._msd_plan(lo, hi) or (lo, 1 << 16)
    b = ((k - a) * scale) >> 32
    h = torch.bincount(b.clamp(max=254), minlength=255)
    # leveled: equal-count partition of 1024 cells
    cells = torch.bincount((((k - a) * ((1023 << 32) // max(hi - lo + 2 * ((hi - lo) >> 4) + 3, 1024))) >> 32).clamp(max=1023), minlength=1024)
    print(f"view {v}: visible {k.numel()}, key span {hi - lo} ({(hi - lo) / 65536 / 255:.2f} of the 16-bit limit), fullest bucket {int(h.max())}, "
          f"mean {k.numel() / 255:.0f}, ratio {float(h.max()) * 255 / k.numel():.2f}; fullest 1/1024 cell {int(cells.max())}")
