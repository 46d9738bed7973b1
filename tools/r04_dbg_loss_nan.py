import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
from gsplat_amd import stages as S
torch.manual_seed(0)
H, W = 32, 48
r = torch.rand(H, W, 3, device="cuda"); t = torch.rand(H, W, 3, device="cuda")
s0, v0 = S.view_loss(r, t, 0.2)
t2 = t.clone(); t2[7, 11, 2] = float("nan")
s1, v1 = S.view_loss(r, t2, 0.2)
torch.cuda.synchronize()
print("sums", s0.tolist(), s1.tolist())
nan = torch.isnan(v1)
print("nan count", int(nan.sum()), "where", nan.nonzero()[:30].tolist())
print("finite v0", bool(torch.isfinite(v0).all()))
