"""Behind a device synchronisation the 4-view step needs ~15 steps to reach its steady time (bench.py's step_ms_first_20).
Which stage's kernels get faster, or is it the gaps between them?  Stage API, ONE stream, 4 views per step, an event pair
around every C-ABI stage (stages.TIMER) and around every step; repeated for three windows of 20 steps, each behind a sync.
python tools/ramp_stages.py"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-gaussian-splatting_amd"))
import torch
from gsplat_amd import scene, stages as S

dev = torch.device("cuda", 0)
n, W, H, deg, B = 1_000_000, 1920, 1080, 3, 4
sc = scene.make_scene(n, seed=0, sh_degree=deg, device="cpu")
means, quats, scales = sc["means3D"].to(dev), sc["rotations"].to(dev), sc["scales"].to(dev)
opac, shs = sc["opacities"].to(dev).squeeze(-1).contiguous(), sc["shs"].to(dev)
K = scene.make_intrinsics(W, H).to(dev).contiguous()
cams = torch.stack([scene.make_camera(i, B) for i in range(B)]).to(dev).contiguous()
v_render = scene.make_v_render(W, H, seed=2).to(dev).contiguous()
bg = torch.zeros(3, device=dev)
out = [torch.zeros_like(means), torch.zeros_like(quats), torch.zeros_like(scales), torch.zeros_like(opac), torch.zeros_like(shs)]
cull = S.DEFAULT_CULL_MODE
bt = S.bin_tile_for(W, H, cull)


def step():
    views = S.project_views(means, quats, scales, opac, shs, deg, cams, K, W, H, cull, return_aux=True)
    vs, bins_all = [], []
    for v in views:
        bins = S.bin_tiles(v.splat, v.radii, v.tiles_touched, v.depth_keys, W, H, cull, tile_rects=v.tile_rects, bin_tile=bt,
                           key_range=v.key_range, qbox=v.qbox)
        order = S.tile_schedule(bins, W, H, bt)
        o = S.composite_fwd(v.splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, W, H, 3, order, want_lists=True, bin_tile=bt,
                            gid_boxes=bins.boxed)
        vs.append(S.composite_bwd(v.splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, o[2], o[3], v_render, None, W, H, 3,
                                  order, quad_lists=o[4], bin_tile=bt, gid_boxes=bins.boxed))
        bins_all.append(bins)
    S.project_views_bwd(means, quats, scales, opac, shs, deg, views, vs, W, H, out)
    for b in bins_all:
        b.finalize()


for _ in range(8):
    step()
torch.cuda.synchronize()
for window in range(3):
    S.TIMER.reset(enabled=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    cuts = [0]
    marks[0].record()
    for i in range(20):
        step()
        marks[i + 1].record()
        cuts.append(len(S.TIMER.records))
    torch.cuda.synchronize()
    recs = list(S.TIMER.records)
    S.TIMER.reset(enabled=False)
    print(f"window {window}: step  span_ms  sum_stages_ms  gaps_ms | per stage (ms)")
    for i in range(20):
        per = collections.OrderedDict()
        for name, a, b in recs[cuts[i]:cuts[i + 1]]:
            per[name] = per.get(name, 0.0) + a.elapsed_time(b)
        span = marks[i].elapsed_time(marks[i + 1])
        tot = sum(per.values())
        print(f"   {i:2d}  {span:7.3f}  {tot:7.3f}  {span - tot:6.3f} | " + "  ".join(f"{k.replace('gsa_', '')}={v:.3f}" for k, v in per.items()))
    torch.cuda.synchronize()
