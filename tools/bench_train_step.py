"""Config C3: full optimisation steps (B views -> loss -> backward -> Adam) at 1M gaussians / 1080p on one GPU,
torch loss + torch.optim.Adam vs the fused HIP loss / Adam kernels.  python tools/bench_train_step.py [N] [B]"""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
import gsplat_amd
from gsplat_amd import scene
from gsplat_amd.harness import SplatModel, ViewParallelTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 1920, 1080
dev = torch.device("cuda", 0)
sc = scene.make_scene(n, seed=0, sh_degree=3)
K = scene.make_intrinsics(W, H).to(dev)
batch = [{"image": torch.rand(H, W, 3, device=dev), "pose": scene.make_camera(v, B).to(dev), "width": W, "height": H}
         for v in range(B)]
for fused in ((True,) if os.environ.get("FUSED_ONLY") else (False, True)):
    model = SplatModel(n, dev).load(xyz=sc["means3D"].to(dev), features_dc=sc["shs"][:, :1].to(dev),
                                    features_rest=sc["shs"][:, 1:].to(dev),
                                    opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)).to(dev),
                                    scaling=torch.log(sc["scales"]).to(dev), rotation=sc["rotations"].to(dev))
    tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), fused=fused, overlap_views=os.environ.get("OVERLAP", "1") != "0")
    tr.async_metrics = fused and os.environ.get("ASYNC_METRICS", "1") != "0"   # (the loop's way: metrics one step late)
    tr.iteration = 5000   # SH degree 3
    steps = 10
    for _ in range(3 + steps):   # (an un-timed window of the same length right in front of the fence: bench.py's timed() says why)
        tr.training_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.training_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"fused={fused}: {dt*1e3:.2f} ms per step of {B} views ({B*W*H/dt/1e6:.0f} Mpix/s), loss {out['loss']:.5f} psnr {out['psnr']:.3f}")
