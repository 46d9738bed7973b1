"""Experiment (GPU box): backward compositor over the forward pass's per-quadrant lists of blended records vs the
plain sweep: kernel times of both passes and agreement of the results (C3 scene)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch
from gsplat_amd import scene, stages as S

N, W, H, deg = 1_000_000, 1920, 1080, 3
sc = {k: v.cuda() for k, v in scene.make_scene(N, seed=0, sh_degree=deg).items()}
vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
splat, radii, tiles, keys, rects = S.project_fwd(sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(),
                                                 sc["shs"], deg, vm, K, W, H, 0, return_rects=True)
bins = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, tile_rects=rects)
M = bins.finalize()[0]
order = S.tile_schedule(bins.tile_offsets, W, H)
bg = torch.zeros(3, device="cuda")
v_render = scene.make_v_render(W, H, seed=2).cuda()
gid, offs = bins.sorted_gid_buffer, bins.tile_offsets

def timed(fn, name, key, reps=20):
    for _ in range(3): out = fn()
    S.TIMER.reset(enabled=True)
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    c, t = S.TIMER.summary()[key]
    S.TIMER.reset(enabled=False)
    print(f"{name}: {t / c * 1e3:.1f} us")
    return out

r0, a0, T0, l0 = timed(lambda: S.composite_fwd(splat, gid, offs, bg, W, H, 3, order), "fwd plain      ", "gsa_composite_fwd")
r1, a1, T1, l1, lists = timed(lambda: S.composite_fwd(splat, gid, offs, bg, W, H, 3, order, want_lists=True), "fwd with lists ", "gsa_composite_fwd")
print("fwd identical:", torch.equal(r0, r1), torch.equal(T0, T1), torch.equal(l0, l1))
qc = lists[1]
print(f"records: M = {M}; blended (quadrant, record) pairs = {int(qc.sum())} ({int(qc.sum()) / (4 * M):.3f} of 4M); max per quadrant {int(qc.max())}")
g0 = timed(lambda: S.composite_bwd(splat, gid, offs, bg, T0, l0, v_render, None, W, H, 3, order), "bwd plain      ", "gsa_composite_bwd")
g1 = timed(lambda: S.composite_bwd(splat, gid, offs, bg, T0, l0, v_render, None, W, H, 3, order, quad_lists=lists), "bwd over lists ", "gsa_composite_bwd")
d = (g0 - g1).abs().max(0).values / g0.abs().max(0).values.clamp_min(1e-30)
print("bwd max rel diff per field:", [f"{float(x):.1e}" for x in d[:10]])
