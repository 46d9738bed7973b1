"""Where a fused training step's time goes: host enqueue time per step (no device wait inside the loop besides the
one-step-late metrics) against the device's step time, and the per-kernel totals of a rocprofv3 run of this script.
python tools/train_step_host.py [steps]"""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "3d-gaussian-splatting_amd")); sys.path.insert(0, R)
import gsplat_amd
from gsplat_amd import scene, stages
from gsplat_amd.harness import SplatModel, ViewParallelTrainer

n, B, W, H = 1_000_000, 4, 1920, 1080
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
sc = scene.make_scene(n, seed=0, sh_degree=3)
K = scene.make_intrinsics(W, H).to(dev)
batch = [{"image": torch.rand(H, W, 3, device=dev), "pose": scene.make_camera(v, B).to(dev), "width": W, "height": H}
         for v in range(B)]
model = SplatModel(n, dev).load(xyz=sc["means3D"].to(dev), features_dc=sc["shs"][:, :1].to(dev),
                                features_rest=sc["shs"][:, 1:].to(dev),
                                opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)).to(dev),
                                scaling=torch.log(sc["scales"]).to(dev), rotation=sc["rotations"].to(dev))
tr = ViewParallelTrainer(model, gsplat_amd.HipBackend(K), fused=True)
tr.async_metrics = True
tr.iteration = 5000
for _ in range(5):
    tr.training_step(batch)
torch.cuda.synchronize()
host = []
stages.WAIT_STATS[:] = [0, 0.0]
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    tr.training_step(batch)
    host.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
host.sort()
print(f"device-paced step {dt * 1e3:.3f} ms; host time in training_step: median {host[len(host) // 2] * 1e3:.3f} ms, "
      f"min {host[0] * 1e3:.3f}; of which waiting for control words {stages.WAIT_STATS[1] / steps * 1e3:.3f} ms per step")
