"""Microbenchmark of the record generation + tile sort alone (gsa_tile_bin16 on bench-like rectangles; nothing consumes
the outputs, so experimental builds with early exits are safe here).  usage: python tools/bench_tile_bin.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
from gsplat_amd import stages as S  # noqa: E402
from gsplat_amd._lib import check, lib  # noqa: E402

W, H, n = 1920, 1080, 1_000_000
tw, th = S.tile_grid(W, H)
g = torch.Generator().manual_seed(0)
w = torch.randint(1, 5, (n,), generator=g)
h = torch.randint(1, 5, (n,), generator=g)
x0 = (torch.rand(n, generator=g) * (tw - 4)).long()
y0 = (torch.rand(n, generator=g) * (th - 4)).long()
culled = torch.rand(n, generator=g) < 0.08
counts = (w * h)
counts[culled] = 0
rect = (x0 | (y0 << 10) | (w << 20))
pairs = torch.stack([rect, counts], dim=1).to(torch.int32).contiguous().cuda()
keys = (torch.rand(n, generator=g) * 2 + 3).to(torch.float32).view(torch.int32).long()
keys[culled] = 0xFFFFFFFF
k32 = torch.where(keys >= 2 ** 31, keys - 2 ** 32, keys).to(torch.int32).cuda()
m = int(counts.sum())
cap = m + 4096
n_blocks = (cap + S.BIN_BLOCK - 1) // S.BIN_BLOCK
order, rects, offsets = (torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(3))
owner = torch.empty(n_blocks + 1, dtype=torch.int32, device="cuda")
ctl = torch.empty(4, dtype=torch.int32, device="cuda")
st = S._stream()
vis = keys[~culled]
plan = S._msd_plan(int(vis.min()), int(vis.max()))
ws = torch.empty(lib.gsa_depth_order_msd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), plan[0], plan[1], None, order.data_ptr(), rects.data_ptr(),
                              offsets.data_ptr(), owner.data_ptr(), n_blocks, ctl.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st), "msd")
torch.cuda.synchronize()
assert (int(ctl[0]), int(ctl[3])) == (m, 0)
sorted_tile = torch.empty(cap, dtype=torch.int16, device="cuda")
sorted_gid = torch.empty(cap, dtype=torch.int32, device="cuda")
k_tmp, v_tmp = torch.empty_like(sorted_tile), torch.empty_like(sorted_gid)
tile_offsets = torch.empty(tw * th + 1, dtype=torch.int32, device="cuda")
ws2 = torch.empty(lib.gsa_sort_workspace_bytes(cap), dtype=torch.uint8, device="cuda")


def run():
    check(lib.gsa_tile_bin16(n, order.data_ptr(), offsets.data_ptr(), rects.data_ptr(), owner.data_ptr(), W, H, 16, cap,
                             ctl[0:1].data_ptr(), k_tmp.data_ptr(), v_tmp.data_ptr(), sorted_tile.data_ptr(),
                             sorted_gid.data_ptr(), tile_offsets.data_ptr(), ws2.data_ptr(), ws2.numel(), st), "bin16")


for _ in range(5):
    run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    run()
b.record()
torch.cuda.synchronize()
print(f"gsa_tile_bin16: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call, M = {m}", flush=True)
