"""Developer aid (GPU box): statistics of the C3 workload that drive kernel design decisions."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch
from gsplat_amd import scene, stages as S

N, W, H, deg = 1_000_000, 1920, 1080, 3
cull = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = {k: v.cuda() for k, v in scene.make_scene(N, seed=0, sh_degree=deg).items()}
vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
splat, radii, tiles, keys = S.project_fwd(sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"], deg, vm, K, W, H, cull)
st, sg, to = S.bin_tiles(splat, radii, tiles, keys, W, H, cull)
r, a, fT, last = S.composite_fwd(splat, sg, to, None, W, H, 3)
tw, th = S.tile_grid(W, H)
M = sg.numel()
L = (to[1:] - to[:-1]).float()
print(f"cull={cull} M={M} visible={(radii[:,0]>0).sum().item()} tiles={tw*th} L mean={L.mean():.0f} max={L.max():.0f}")
lastp = torch.full((th * 16, tw * 16), -1, device="cuda", dtype=torch.int32); lastp[:H, :W] = last
lt = lastp.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th * tw, 256)
start = to[:-1][:, None]
proc = torch.where(lt >= 0, lt - start + 1, torch.zeros_like(lt)).float()  # records walked per pixel (fwd, approx)
print(f"per-pixel processed (to last contributor): mean={proc.mean():.0f}; per-tile max-last depth mean={proc.max(1).values.mean():.0f};"
      f" sum over tiles of max depth={proc.max(1).values.sum():.3e} (bwd staged records)")
print(f"final T: mean={fT.mean():.4f} frac(T<1e-3)={(fT<1e-3).float().mean():.3f} alpha mean={a.mean():.3f}")
# sample tiles: pair statistics
g = torch.Generator(device="cpu").manual_seed(0)
sel = torch.randperm(tw * th, generator=g)[:300].cuda()
tot_pairs = tot_valid = 0
strip_any = blk_any = strips = 0
tile_any = 0; recs = 0
quad_any = 0; quads = 0
for t in sel.tolist():
    s, e = int(to[t]), int(to[t + 1])
    depth = int(proc[t].max().item())
    if depth == 0: continue
    ids = sg[s:s + depth].long()
    rec = splat[ids]
    tx, ty = t % tw, t // tw
    px = (tx * 16 + torch.arange(16, device="cuda") + 0.5)[None, :].expand(16, 16).reshape(-1)
    py = (ty * 16 + torch.arange(16, device="cuda") + 0.5)[:, None].expand(16, 16).reshape(-1)
    dx = rec[None, :, 0] - px[:, None]; dy = rec[None, :, 1] - py[:, None]
    sig = 0.5 * (rec[None, :, 2] * dx * dx + rec[None, :, 4] * dy * dy) + rec[None, :, 3] * dx * dy
    al = torch.clamp(rec[None, :, 5] * torch.exp(-sig), max=0.999)
    live = torch.arange(depth, device="cuda")[None, :] < proc[t][:, None]          # pair walked by its pixel
    valid = (sig >= 0) & (al >= 1 / 255) & live
    tot_pairs += int(live.sum()); tot_valid += int(valid.sum())
    v = valid.view(16, 16, depth)
    strip_any += int(v.view(4, 4, 16, depth).any(1).any(1).sum()); strips += 4 * depth
    blk_any += int(v.view(2, 8, 2, 8, depth).any(3).any(1).sum())
    quad_any += int(v.view(4, 4, 4, 4, depth).any(3).any(1).sum()); quads += 16 * depth
    tile_any += int(v.any(0).any(0).sum()); recs += depth
print(f"sampled tiles: walked pairs={tot_pairs:.3e} valid frac={tot_valid/tot_pairs:.3f}")
print(f"(wave,gaussian) any-valid: 16x4 strips {strip_any/strips:.3f}  8x8 blocks {blk_any/strips:.3f}; 4x4 quads {quad_any/quads:.3f}; tile-level any {tile_any/recs:.3f}")
