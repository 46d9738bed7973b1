"""Developer aid (GPU box): how often the forward walk's CAREFUL path runs on the C3 bench scene.  For a sample of 16 x 16
tiles, per 8 x 8 quadrant: the records that pass the alpha test at one pixel of the quadrant at least, in depth order (what
the wave walks, up to the reach test's false positives), walked until all 64 pixels have stopped; the walk is cut into groups
of G records and a group is "careful" when a pixel stops inside it (composite.hip walk_fwd re-evaluates such a group record by
record).  python tools/analyze_careful.py [tiles] [G]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch  # noqa: E402
from gsplat_amd import scene, stages as S  # noqa: E402

N, W, H, deg = 1_000_000, 1920, 1080, 3
n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = {k: v.cuda() for k, v in scene.make_scene(N, seed=0, sh_degree=deg).items()}
vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
splat, radii, tiles, keys = S.project_fwd(sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"],
                                          deg, vm, K, W, H, 0)
st, sg, to = S.bin_tiles(splat, radii, tiles, keys, W, H, 0)
tw, th = S.tile_grid(W, H)
to_c = to.cpu().tolist()
g = torch.Generator().manual_seed(0)
pick = torch.randperm(tw * th, generator=g)[:n_tiles].tolist()
ar = torch.arange(8, device="cuda")
tot = {"waves": 0, "walked": 0, "groups": 0, "careful": 0, "stops": 0, "useful_lanes": 0.0, "live_lanes": 0.0}
for t in pick:
    s, e = to_c[t], to_c[t + 1]
    if e == s:
        continue
    rec = splat[sg[s:e].long()]
    tx, ty = t % tw, t // tw
    for q in range(4):
        x0, y0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        px = (x0 + ar + 0.5)[None, :].expand(8, 8).reshape(-1)
        py = (y0 + ar + 0.5)[:, None].expand(8, 8).reshape(-1)
        dx = rec[None, :, 0] - px[:, None]
        dy = rec[None, :, 1] - py[:, None]
        sig = 0.5 * (rec[None, :, 2] * dx * dx + rec[None, :, 4] * dy * dy) + rec[None, :, 3] * dx * dy
        al = torch.clamp(rec[None, :, 5] * torch.exp(-sig), max=0.999)
        valid = (sig >= 0) & (al >= 1 / 255)
        reach = valid.any(0)                       # records the wave walks (exact reach)
        al, valid = al[:, reach], valid[:, reach]
        L = int(reach.sum())
        if L == 0:
            continue
        a = torch.where(valid, al, torch.zeros_like(al))
        nT = torch.cumprod(1 - a, dim=1)
        stop_here = valid & (nT <= 1e-4)
        has = stop_here.any(1)
        first = torch.where(has, stop_here.float().argmax(1), torch.full_like(has, L, dtype=torch.long))   # index of the record at which the pixel stops
        walked = int(first.max()) + 1 if bool(has.all()) else L
        walked = min(walked, L)
        n_groups = (walked + G - 1) // G
        stops = first[first < walked]
        careful = int(torch.unique(stops // G).numel())
        live = (torch.arange(walked, device="cuda")[None, :] < first[:, None])       # pixel still accumulating at record j
        tot["waves"] += 1
        tot["walked"] += walked
        tot["groups"] += n_groups
        tot["careful"] += careful
        tot["stops"] += int(stops.numel())
        tot["useful_lanes"] += float((valid[:, :walked] & live).sum())
        tot["live_lanes"] += float(live.sum())
w = max(tot["waves"], 1)
print(f"sampled quadrant-waves {tot['waves']}: walked records per wave {tot['walked'] / w:.1f}, groups of {G} per wave {tot['groups'] / w:.1f}, "
      f"CAREFUL groups per wave {tot['careful'] / w:.1f} ({100.0 * tot['careful'] / max(tot['groups'], 1):.1f} % of the groups), "
      f"pixel stops per wave {tot['stops'] / w:.1f}")
print(f"lanes: live (pixel still accumulating) {100 * tot['live_lanes'] / (64.0 * max(tot['walked'], 1)):.1f} % of the walk's lane-slots, "
      f"useful (live AND alpha >= 1/255) {100 * tot['useful_lanes'] / (64.0 * max(tot['walked'], 1)):.1f} %")
