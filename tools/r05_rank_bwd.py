"""Developer aid (GPU box): the pieces of a view-parallel rank's backward on the C3 scene, stage API, HIP events:
backward sweep with the colour moments in the gradient record vs in a compact [N,3] array (gsa_composite_bwd_lists_rgb), the
per-view factored projection backward, the per-step geometry-only one, the SH rebuild.  python tools/r05_rank_bwd.py [views]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch  # noqa: E402
from gsplat_amd import scene, stages as S  # noqa: E402

N, W, H, deg = 1_000_000, 1920, 1080, 3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
sc = {k: v.to(dev) for k, v in scene.make_scene(N, seed=0, sh_degree=deg).items()}
means, quats, scales, opac, colors = sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"]
vms = torch.stack([scene.make_camera(i, 8) for i in range(B)]).to(dev)
K = scene.make_intrinsics(W, H).to(dev)
v_render = scene.make_v_render(W, H, seed=2).to(dev)
bt = S.bin_tile_for(W, H, 0)
views = S.project_views(means, quats, scales, opac, colors, deg, vms, K, W, H, 0, return_aux=True)
fw = []
for pv in views:
    for _ in range(2):   # (second pass: remembered capacity, bucketed ordering)
        bins = S.bin_tiles(pv.splat, pv.radii, pv.tiles_touched, pv.depth_keys, W, H, 0, tile_rects=pv.tile_rects, bin_tile=bt,
                           key_range=pv.key_range, qbox=pv.qbox)
        bins.finalize()
    order = S.tile_schedule(bins, W, H, bt)
    fw.append((pv, bins, order))


def fwd(pv, bins, order):
    return S.composite_fwd(pv.splat, bins.sorted_gid_buffer, bins.tile_offsets, None, W, H, 3, order, want_lists=True, bin_tile=bt,
                           gid_boxes=bins.boxed)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rgb = torch.zeros(B, N, 3, device=dev)
out = [torch.empty_like(t) for t in (means, quats, scales, opac)] + [torch.empty_like(colors)]
state = {}


def sweep(with_rgb):
    pv, bins, order = fw[0]
    _, _, final_T, last_id, lists = fwd(pv, bins, order)
    state["v"] = S.composite_bwd(pv.splat, bins.sorted_gid_buffer, bins.tile_offsets, None, final_T, last_id, v_render, None, W, H, 3,
                                 order, quad_lists=lists, bin_tile=bt, gid_boxes=bins.boxed, v_rgb_out=rgb[0] if with_rgb else None)


t_fwd = timeit(lambda: fwd(*fw[0]))
t_plain = timeit(lambda: sweep(False)) - t_fwd
t_rgb = timeit(lambda: sweep(True)) - t_fwd
print(f"forward sweep {t_fwd:.4f} ms; backward sweep (incl. prepare): colour moments in the record {t_plain:.4f} ms, in a compact [N,3] array "
      f"(+ its zero fill) {t_rgb:.4f} ms")
vs = []
for c, (pv, bins, order) in enumerate(fw):
    _, _, final_T, last_id, lists = fwd(pv, bins, order)
    vs.append(S.composite_bwd(pv.splat, bins.sorted_gid_buffer, bins.tile_offsets, None, final_T, last_id, v_render, None, W, H, 3, order,
                              quad_lists=lists, bin_tile=bt, gid_boxes=bins.boxed, v_rgb_out=rgb[c]))
t_fact = timeit(lambda: [S.project_bwd(means, quats, scales, opac, colors, deg, pv.viewmat, K, W, H, pv.splat, pv.radii, vs[c], out=out,
                                       accumulate=c > 0, v_rgb_out=rgb[c], sh_aux=pv.sh_aux) for c, (pv, _, _) in enumerate(fw)])
t_geom = timeit(lambda: S.project_views_bwd(means, quats, scales, opac, colors, deg, views, vs, W, H, out, v_rgbs=list(rgb.unbind(0))))
t_full = timeit(lambda: S.project_views_bwd(means, quats, scales, opac, colors, deg, views, vs, W, H, out))
t_exp = timeit(lambda: S.sh_grad_expand(means, deg, vms, K[None], rgb, out[4]))
print(f"{B} views: per-view factored projection backward {t_fact:.4f} ms per step; geometry-only batch {t_geom:.4f}; full batch {t_full:.4f}; "
      f"SH rebuild over {B} views {t_exp:.4f}")
