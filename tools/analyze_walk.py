"""Developer aid (GPU box): per-tile forward walk length (records visited until every pixel of the tile has
stopped, or the whole list) for the C3 scene -- the critical-path length of a wave-per-tile compositor."""
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
tw, th = S.tile_grid(W, H)
to_c = to.cpu().tolist()
walk = torch.zeros(tw * th); Ls = torch.zeros(tw * th); full = 0
ar = torch.arange(16, device="cuda")
for t in range(tw * th):
    s, e = to_c[t], to_c[t + 1]
    L = e - s; Ls[t] = L
    if L == 0: continue
    rec = splat[sg[s:e].long()]
    tx, ty = t % tw, t // tw
    px = (tx * 16 + ar + 0.5)[None, :].expand(16, 16).reshape(-1); py = (ty * 16 + ar + 0.5)[:, None].expand(16, 16).reshape(-1)
    inside = (px < W) & (py < H)
    dx = rec[None, :, 0] - px[:, None]; dy = rec[None, :, 1] - py[:, None]
    sig = 0.5 * (rec[None, :, 2] * dx * dx + rec[None, :, 4] * dy * dy) + rec[None, :, 3] * dx * dy
    al = torch.clamp(rec[None, :, 5] * torch.exp(-sig), max=0.999)
    valid = (sig >= 0) & (al >= 1 / 255)
    nT = torch.cumprod(1 - torch.where(valid, al, torch.zeros_like(al)), dim=1)
    stopped = valid & (nT <= 1e-4)
    has = stopped.any(1)
    first = torch.where(has, stopped.float().argmax(1), torch.full_like(has, L, dtype=torch.long))
    first = first[inside]
    wk = int(first.max()) + 1 if bool((first < L).all()) else L
    if wk >= L: full += 1
    walk[t] = min(wk, L)
print(f"cull={cull} tiles={tw*th} nonempty={(Ls>0).sum().item()} L mean(nonempty)={Ls[Ls>0].mean():.0f} max={Ls.max():.0f}")
print(f"walk: sum={walk.sum():.3e} mean(nonempty)={walk[Ls>0].mean():.0f} max={walk.max():.0f} tiles walking whole list={full}")
q = torch.quantile(walk[Ls > 0], torch.tensor([0.5, 0.9, 0.99, 0.999]))
print("walk quantiles 50/90/99/99.9:", q.tolist())
