"""Microbenchmark of the depth ordering alone (no consumers, so experimental builds that leave garbage are safe here):
gsa_depth_order_msd vs gsa_depth_order + gsa_scan_tiles_sorted on 1 M bench-like keys.  usage: python tools/bench_depth_order.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
from gsplat_amd import stages as S  # noqa: E402
from gsplat_amd._lib import check, lib  # noqa: E402

n = 1_000_000
g = torch.Generator().manual_seed(0)
keys = (torch.rand(n, generator=g) * 2 + 3).to(torch.float32).view(torch.int32).long()
culled = torch.rand(n, generator=g) < 0.08
keys[culled] = 0xFFFFFFFF
counts = torch.randint(0, 17, (n,), generator=g)
counts[culled] = 0
pairs = torch.stack([torch.randint(1, 2 ** 30, (n,), generator=g), counts], dim=1).to(torch.int32).contiguous().cuda()
k32 = torch.where(keys >= 2 ** 31, keys - 2 ** 32, keys).to(torch.int32).cuda()
vis = keys[~culled]
plan = S._msd_plan(int(vis.min()), int(vis.max()))
n_blocks = int(counts.sum()) // 4096 + 2
bufs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(3)]
owner = torch.empty(n_blocks + 1, dtype=torch.int32, device="cuda")
ctl = torch.empty(4, dtype=torch.int32, device="cuda")
ws_m = torch.empty(lib.gsa_depth_order_msd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
ws_l = torch.empty(lib.gsa_depth_order_workspace_bytes(n), dtype=torch.uint8, device="cuda")
ws_s = torch.empty(lib.gsa_sort_workspace_bytes(n), dtype=torch.uint8, device="cuda")
st = S._stream()


def msd():
    check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), plan[0], plan[1], None, bufs[0].data_ptr(), bufs[1].data_ptr(),
                                  bufs[2].data_ptr(), owner.data_ptr(), n_blocks, ctl.data_ptr(), None, 0, ws_m.data_ptr(), ws_m.numel(), st), "msd")


def lsd():
    check(lib.gsa_depth_order(n, k32.data_ptr(), pairs.data_ptr(), 24, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                              ctl[1:3].data_ptr(), ws_l.data_ptr(), ws_l.numel(), st), "lsd")
    check(lib.gsa_scan_tiles_sorted(n, bufs[2].data_ptr(), owner.data_ptr(), n_blocks, ctl[0:1].data_ptr(), ws_s.data_ptr(),
                                    ws_s.numel(), st), "scan")


for name, fn in (("msd", msd), ("lsd+scan", lsd)):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call; ctl = {[int(x) & 0xFFFFFFFF for x in ctl]}", flush=True)
