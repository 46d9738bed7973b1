import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3d-gaussian-splatting_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gsplat_amd as gsa
from gsplat_amd import scene
W, H = 320, 200
sc = scene.make_scene(20000, seed=3, sh_degree=3, s_min=0.005, s_max=0.08)
args = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], sc["shs"]]
args = [t.cuda() for t in args] + [scene.make_camera(1, 5)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()]
with torch.no_grad():
    r0, a0, m0 = gsa.rasterization(*args, W, H, sh_degree=3, cull_mode=0)
    r0b, a0b, _ = gsa.rasterization(*args, W, H, sh_degree=3, cull_mode=0)
    r1, a1, m1 = gsa.rasterization(*args, W, H, sh_degree=3, cull_mode=1)
print("rerun equal:", torch.equal(r0, r0b), torch.equal(a0, a0b))
d = (r0 - r1).abs().amax(-1)[0]
print("differing pixels:", int((d > 0).sum()), "max", float(d.max()), "alpha diff px", int(((a0 - a1).abs() > 0).sum()))
ys, xs = torch.nonzero(d > 0, as_tuple=True)
for y, x in list(zip(ys.tolist(), xs.tolist()))[:10]:
    print(y, x, r0[0, y, x].tolist(), r1[0, y, x].tolist(), float(a0[0, y, x, 0]), float(a1[0, y, x, 0]))
