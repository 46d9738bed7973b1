"""Developer aid: per-tensor error report HIP vs oracle for one scene (GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
import torch, numpy as np
import gsplat_amd
from gsplat_amd import scene
from oracle import rasterization as orc

W, H, N = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (160, 96, 2000))]
deg = 3
sc = scene.make_scene(N, seed=0, sh_degree=3, s_min=0.01, s_max=0.1)
K = scene.make_intrinsics(W, H); vm = scene.make_camera(1, 8); v = scene.make_v_render(W, H)
names = ["means3D", "rotations", "scales", "opacities", "shs"]

def run(fn, dev, **kw):
    leaves = [sc[k].to(dev).clone().requires_grad_(True) for k in names]
    r, a, m = fn(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm[None].to(dev), K[None].to(dev), W, H,
                 sh_degree=deg, backgrounds=torch.full((1, 3), 0.1, device=dev), **kw)
    (r[0] * v.to(dev)).sum().backward()
    return r.detach().cpu(), a.detach().cpu(), m, [t.grad.cpu() for t in leaves]

r1, a1, m1, g1 = run(gsplat_amd.rasterization, "cuda")
r0, a0, m0, g0 = run(orc, "cpu")
d = (r1 - r0).abs()[0]
print("render max err", d.max().item(), "n>1e-4:", int((d.max(-1).values > 1e-4).sum()), "of", H * W)
idx = torch.nonzero(d.max(-1).values > 1e-4)[:10]
for y, x in idx.tolist():
    print("  px", x, y, "hip", r1[0, y, x].tolist(), "ref", r0[0, y, x].tolist(), "alpha", a1[0, y, x, 0].item(), a0[0, y, x, 0].item())
print("alpha max err", (a1 - a0).abs().max().item())
print("M hip", m1["flatten_ids"].numel(), "ref", m0["flatten_ids"].numel(), "flat equal:",
      m1["flatten_ids"].numel() == m0["flatten_ids"].numel() and bool((m1["flatten_ids"].cpu() == m0["flatten_ids"]).all()))
print("radii equal", bool((m1["radii"].cpu() == m0["radii"]).all()))
for k in ("means2d", "conics", "depths"):
    print(k, (m1[k].cpu() - m0[k]).abs().max().item(), m0[k].abs().max().item())
for n, x, y in zip(names, g1, g0):
    e = (x - y).abs()
    print("grad", n, "max err", e.max().item(), "max ref", y.abs().max().item(), "rel", e.max().item() / y.abs().max().item())
