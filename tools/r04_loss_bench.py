"""gsa_view_loss alone at 1080p: device time per call (HIP events), bytes moved, rate."""
import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
from gsplat_amd import stages as S
H, W = 1080, 1920
r = torch.rand(H, W, 3, device="cuda"); t = torch.rand(H, W, 3, device="cuda")
for _ in range(20):
    S.view_loss(r, t, 0.2)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
a.record()
for _ in range(n):
    S.view_loss(r, t, 0.2)
b.record(); b.synchronize()
ms = a.elapsed_time(b) / n
print(f"view_loss (loss + gradient, 2 launches): {ms*1e3:.1f} us per call; {3*H*W*4*3/1e6:.1f} MB -> {3*H*W*4*3/ms/1e6:.0f} GB/s")
