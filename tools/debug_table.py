import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
from gsplat_amd import scene, stages as S
from gsplat_amd._lib import check, lib
N, W, H = 1_000_000, 1920, 1080
sc = scene.make_scene(N, seed=0, sh_degree=3)
d = {k: v.cuda().contiguous() for k, v in sc.items()}
K = scene.make_intrinsics(W, H).cuda()
for view in (1, 0):
    vm = scene.make_camera(view, 8).cuda()
    kr = S.key_range_buffer(N, d["means3D"].device)
    splat, radii, tiles, keys, rects = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"], 3, vm, K, W, H, 0, return_rects=True, bin_tile=32, key_range=kr)
    bufs = [torch.empty(N, dtype=torch.int32, device="cuda") for _ in range(3)]
    owner = torch.empty(2000, dtype=torch.int32, device="cuda"); ctl = torch.empty(4, dtype=torch.int32, device="cuda")
    ws = torch.zeros(lib.gsa_depth_order_msd_workspace_bytes(N), dtype=torch.uint8, device="cuda")
    check(lib.gsa_depth_order_msd(N, keys.data_ptr(), rects.data_ptr(), 0, 0, kr.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), owner.data_ptr(), 1999, ctl.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), S._stream()), "msd")
    torch.cuda.synchronize()
    off = lib.gsa_depth_order_msd_table_offset(N)
    tab = ws[off:off + 4 * lib.gsa_depth_order_msd_table_words()].view(torch.int32).cpu()
    cells, shift, g0, flags = int(tab[1286]), int(tab[1285]), int(tab[1284]) & 0xFFFFFFFF, int(tab[1287])
    lut = tab[:1024].contiguous().view(torch.uint8)[:cells].long(); cs = tab[1024:1024 + 257].long()
    k = (keys.long() & 0xFFFFFFFF).cpu(); k = k[k != 0xFFFFFFFF]
    cell = (k >> shift) - g0
    per_cell = torch.bincount(cell, minlength=cells)
    per_b = torch.bincount(lut[cell], minlength=255)
    krc = kr.long().cpu() & 0xFFFFFFFF
    samp = torch.cat([krc[:, 2], krc[:, 3]]); samp = samp[samp != 0xFFFFFFFF]
    per_cell_s = torch.bincount((samp >> shift) - g0, minlength=cells)
    top = torch.argsort(per_b, descending=True)[:5]
    print(f"view {view}: status {int(ctl[3])} cells {cells} shift {shift} flags {flags} buckets {int(lut.max())+1} fullest cell {int(per_cell.max())} sample {samp.numel()} fullest sample cell {int(per_cell_s.max())}")
    for b in top.tolist():
        c0, c1 = int(cs[b]), int(cs[b + 1])
        print(f"   bucket {b}: {int(per_b[b])} keys, cells [{c0},{c1}) sample there {int(per_cell_s[c0:c1].sum())} (share target {samp.numel() / 240:.0f})")
