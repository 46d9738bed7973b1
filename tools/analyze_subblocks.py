"""Developer aid: would finer-than-quadrant work units pay?  For the bench scene (C3) take every (tile, record) the
backward sweep stages, apply the compositor's reach test to the four 8x8 quadrants and to the eight 8x4 half-quadrants,
and compare the walk trips of (a) one wave per quadrant (today) with (b) two half-waves per wave, each walking its own
list in lock-step (trips = max of the two lists, per 64-record batch, four entries per trip)."""
import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
import gsplat_amd
from gsplat_amd import scene

n, W, H = 1_000_000, 1920, 1080
dev = "cuda"
sc = scene.make_scene(n, seed=0, sh_degree=3)
leaves = [sc[k].to(dev) for k in ("means3D", "rotations", "scales", "opacities", "shs")]
K = scene.make_intrinsics(W, H).to(dev)[None]; vm = scene.make_camera(0, 1).to(dev)[None]
with torch.no_grad():
    r, a, meta = gsplat_amd.rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm, K, W, H, sh_degree=3)
flat = meta["flatten_ids"].long(); offs = meta["isect_offsets"].reshape(-1).long()
means2d, conics = meta["means2d"][0], meta["conics"][0]
opac = leaves[3][:, 0]
tile_w = (W + 15) // 16
M = flat.numel()
# last list position any pixel of a tile saw (stand-in: the forward's last_id is not exposed; use alpha saturation proxy:
# walk every record of the first 384 of each tile's list, which is where the C3 walks end)
tile_of = torch.searchsorted(offs, torch.arange(M, device=dev), right=True) - 1
pos_in_tile = torch.arange(M, device=dev) - offs[tile_of]
sel = pos_in_tile < 384
g = flat[sel]; t = tile_of[sel]; p = pos_in_tile[sel]
mx, my = means2d[g, 0], means2d[g, 1]
ca, cb, cc = conics[g, 0], conics[g, 1], conics[g, 2]
thr = torch.log(255.0 * opac[g]) * 1.0001 + 1e-4
tx, ty = (t % tile_w).float() * 16, (t // tile_w).float() * 16

def reach(x0, y0, w, h):
    x1, y1 = x0 + (w - 1), y0 + (h - 1)
    inside = (mx >= x0) & (mx <= x1) & (my >= y0) & (my <= y1)
    best = torch.full_like(mx, 3e38)
    for dx in (mx - x0, mx - x1):
        dy = torch.minimum(my - y0, torch.maximum(my - y1, -cb * dx / cc))
        best = torch.minimum(best, 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy)
    for dy in (my - y0, my - y1):
        dx = torch.minimum(mx - x0, torch.maximum(mx - x1, -cb * dy / ca))
        best = torch.minimum(best, 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy)
    return inside | (best <= thr)

quad = torch.stack([reach(tx + (q & 1) * 8 + 0.5, ty + (q >> 1) * 8 + 0.5, 8, 8) for q in range(4)])          # [4, S]
half_h = torch.stack([reach(tx + (q & 1) * 8 + 0.5, ty + (q >> 1) * 8 + hh * 4 + 0.5, 8, 4) for q in range(4) for hh in range(2)])
half_v = torch.stack([reach(tx + (q & 1) * 8 + hh * 4 + 0.5, ty + (q >> 1) * 8 + 0.5, 4, 8) for q in range(4) for hh in range(2)])
print(f"records analysed {g.numel()} ; quadrant visits {int(quad.sum())} ({float(quad.float().mean()):.3f} of pairs)")
batch = (t * 6 + p // 64)                                  # (tile, 64-record batch) id
nb = int(batch.max()) + 1
def per_batch(mask):
    return torch.zeros(nb, device=dev).index_add_(0, batch, mask.float())
trips_now = sum(torch.ceil(per_batch(quad[q]) / 4).sum() for q in range(4))
for name, hm in (("8x4 (top/bottom)", half_h), ("4x8 (left/right)", half_v)):
    visits = int(hm.sum())
    trips = sum(torch.ceil(torch.maximum(per_batch(hm[2 * q]), per_batch(hm[2 * q + 1])) / 4).sum() for q in range(4))
    print(f"{name}: half visits {visits} = {visits / int(quad.sum()):.2f} per quadrant visit; walk trips {float(trips):.0f} vs "
          f"{float(trips_now):.0f} today = {float(trips) / float(trips_now):.3f}")
