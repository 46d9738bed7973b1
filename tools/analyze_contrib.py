"""Developer aid (GPU box): of the (8x8 quadrant, record) pairs the forward walk visits in the C3 scene, how many have
at least one pixel that actually blends the record (alpha >= 1/255, pixel still live)?  The reach test keeps ~45 % of
the pairs; pairs it keeps but where no pixel blends are what a per-batch "contributed" bitmask from the forward pass
could spare the backward pass.  Samples every STRIDE-th tile."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch
from gsplat_amd import scene, stages as S

N, W, H, deg = 1_000_000, 1920, 1080, 3
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = {k: v.cuda() for k, v in scene.make_scene(N, seed=0, sh_degree=deg).items()}
vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
splat, radii, tiles, keys = S.project_fwd(sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"], deg, vm, K, W, H, 0)
st, sg, to = S.bin_tiles(splat, radii, tiles, keys, W, H, 0)
tw, th = S.tile_grid(W, H)
to_c = to.cpu().tolist()
ar = torch.arange(16, device="cuda")
pairs = contrib = lanes = walked_pairs = 0
hist = torch.zeros(65, dtype=torch.long)
for t in range(0, tw * th, stride):
    s, e = to_c[t], to_c[t + 1]
    L = e - s
    if L == 0: continue
    rec = splat[sg[s:e].long()]
    tx, ty = t % tw, t // tw
    px = (tx * 16 + ar + 0.5)[None, :].expand(16, 16).reshape(-1); py = (ty * 16 + ar + 0.5)[:, None].expand(16, 16).reshape(-1)
    dx = rec[None, :, 0] - px[:, None]; dy = rec[None, :, 1] - py[:, None]
    sig = 0.5 * (rec[None, :, 2] * dx * dx + rec[None, :, 4] * dy * dy) + rec[None, :, 3] * dx * dy
    al = torch.clamp(rec[None, :, 5] * torch.exp(-sig), max=0.999)
    valid = (sig >= 0) & (al >= 1 / 255) & ((px < W) & (py < H))[:, None]
    Tn = torch.cumprod(1 - torch.where(valid, al, torch.zeros_like(al)), dim=1)
    Tprev = torch.cat([torch.ones_like(Tn[:, :1]), Tn[:, :-1]], dim=1)
    dead = torch.cummax((valid & (Tn <= 1e-4)).int(), dim=1).values.bool()      # stopped AT or before this record
    blends = valid & ~dead                                                     # the stopping record itself is not blended
    quad = ((torch.arange(256, device="cuda") // 16) // 8) * 2 + ((torch.arange(256, device="cuda") % 16) // 8)
    for q in range(4):
        b = blends[quad == q]                       # [64, L]
        per_rec = b.sum(0)                          # blending pixels per record
        live_any = (~dead[quad == q]).any(0)        # quadrant still has live pixels at this record
        walked = int(live_any.sum())                # records the quadrant's wave walks at all
        pairs += L; walked_pairs += walked
        contrib += int((per_rec > 0).sum()); lanes += int(per_rec.sum())
        hist += torch.bincount(per_rec[live_any].cpu(), minlength=65)[:65]
print(f"sampled (quadrant, record) pairs: {pairs}; while the quadrant is live: {walked_pairs} ({walked_pairs/pairs:.3f})")
print(f"pairs with >= 1 blending pixel: {contrib} ({contrib/pairs:.3f} of all, {contrib/walked_pairs:.3f} of live)")
print(f"blending pixels per contributing pair: {lanes/max(contrib,1):.1f} of 64")
c = hist.float() / hist.sum()
print("share of live pairs by blending-pixel count: 0:", round(float(c[0]), 3), " 1-8:", round(float(c[1:9].sum()), 3),
      " 9-32:", round(float(c[9:33].sum()), 3), " 33-64:", round(float(c[33:].sum()), 3))
