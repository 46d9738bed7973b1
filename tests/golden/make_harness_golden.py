"""Capture golden vectors of the reference's training-step harness.

RUNS IN THE AUTHORING CONTAINER ONLY: it imports the reference's own Python from /root/reference (which never
travels) and writes ``tests/golden/harness_golden.npz`` -- inputs and expected outputs, data only.

How: the modules the reference imports but this image lacks (cv2, plyfile, wandb, kornia) are replaced by
MagicMock stubs, and the module name ``gsplat`` is bound to this repo's CPU oracle, so that
``GaussianRasterizer(backend="gsplat")`` (/root/reference/src/gaussians/gaussian_rasterizer.py:97-118)
builds the reference's real ``GSplatBackend`` on top of it.  Then the reference's real
``GaussianTrainer`` (/root/reference/src/training/trainer.py) is driven on a small seeded scene:
``_compute_ssim``, ``_psnr``, ``_stclamp``, ``_scale_ratio_reg``, ``_active_sh_degree``, ``_update_learning_rate`` and
two consecutive ``_training_step`` calls (AMP off, CPU) with ``_setup_optimizer``'s Adam.

    python tests/golden/make_harness_golden.py
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))
sys.path.insert(0, REFERENCE)

for name in ("cv2", "plyfile", "wandb", "kornia", "kornia.feature"):
    sys.modules.setdefault(name, MagicMock())
import oracle  # noqa: E402

shim = types.ModuleType("gsplat")
shim.rasterization = oracle.rasterization
sys.modules["gsplat"] = shim

from src.gaussians.gaussian_model import GaussianModel  # noqa: E402
from src.gaussians.gaussian_rasterizer import GaussianRasterizer  # noqa: E402
from src.gaussians.training_config import TrainingConfig  # noqa: E402
from src.training.trainer import GaussianTrainer  # noqa: E402

N, W, H, B = 300, 48, 32, 3


def scene_leaves(seed):
    """Raw (pre-activation) parameters of a small visible scene, deterministic."""
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    return {
        "xyz": t(rng.uniform(-1.0, 1.0, (N, 3))),
        "features_dc": t(rng.uniform(-1.0, 1.0, (N, 1, 3)) * 1.5),
        "features_rest": t(rng.normal(0.0, 0.05, (N, 15, 3))),
        "opacity": t(rng.uniform(-2.0, 2.0, (N, 1))),                      # logits
        "scaling": t(np.log(rng.uniform(0.02, 0.4, (N, 3)))),              # log scale (max/min ratio up to 20)
        "rotation": t(rng.normal(0.0, 1.0, (N, 4))),
    }


def cameras():
    from gsplat_amd import scene
    K = scene.make_intrinsics(W, H)
    return K, [scene.make_camera(v, 5) for v in range(B)]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    out = {}
    cfg = TrainingConfig()
    cfg.use_mixed_precision = False
    trainer = GaussianTrainer(cfg, torch.device("cpu"), wandb_run=None)

    # ---- pure helper functions
    rng = np.random.default_rng(7)
    img_a = torch.from_numpy(rng.uniform(-0.2, 1.2, (H, W, 3)).astype(np.float32))
    img_b = torch.from_numpy(rng.uniform(0.0, 1.0, (H, W, 3)).astype(np.float32))
    out["helper_img_a"], out["helper_img_b"] = img_a.numpy(), img_b.numpy()
    out["ssim"] = trainer._compute_ssim(img_a.permute(2, 0, 1)[None], img_b.permute(2, 0, 1)[None]).numpy()
    out["psnr"] = np.float64(trainer._psnr(img_a.clamp(0, 1), img_b))
    out["psnr_identical"] = np.float64(trainer._psnr(img_b, img_b))
    x = img_a.clone().requires_grad_(True)
    y = trainer._stclamp(x)
    (y * img_b).sum().backward()
    out["stclamp_fwd"], out["stclamp_grad"] = y.detach().numpy(), x.grad.numpy()
    sched = []
    for it in (0, 999, 1000, 2500, 3000, 50_000, 300_000, 400_000):
        trainer.iteration = it
        opt = trainer._setup_optimizer(GaussianModel(4, "cpu"))
        sched.append([it, trainer._active_sh_degree(), trainer._update_learning_rate(opt)])
    out["schedule"] = np.asarray(sched, dtype=np.float64)

    # ---- two training steps through the reference's own backend plumbing
    leaves = scene_leaves(seed=11)
    for k, v in leaves.items():
        out["init_" + k] = v.numpy()
    K, poses = cameras()
    target_leaves = scene_leaves(seed=12)
    model_t = GaussianModel(N, "cpu")
    with torch.no_grad():
        for k, v in target_leaves.items():
            getattr(model_t, k).copy_(v)
    rasterizer = GaussianRasterizer(K=K, device="cpu", enable_caching=True, backend="gsplat")
    assert type(rasterizer.backend).__name__ == "GSplatBackend"
    batch = []
    with torch.no_grad():
        for pose in poses:  # targets: renders of a second scene, clamped to [0,1]
            vt = {"pose": pose, "K": K, "width": W, "height": H}
            trainer.iteration = 3000
            img = trainer._render_view(model_t, rasterizer, vt)["render"].clamp(0, 1)
            batch.append({"image": img, "pose": pose, "K": K, "width": W, "height": H})
    out["K"] = K.numpy()
    out["poses"] = torch.stack(poses).numpy()
    out["targets"] = torch.stack([b["image"] for b in batch]).numpy()

    model = GaussianModel(N, "cpu")
    with torch.no_grad():
        for k, v in leaves.items():
            getattr(model, k).copy_(v)
    trainer.iteration = 2500  # SH degree 2 is active
    out["step_iteration"] = np.int64(trainer.iteration)
    optimizer = trainer._setup_optimizer(model)
    out["scale_reg"] = trainer._scale_ratio_reg(model).detach().numpy()
    for step in (1, 2):
        metrics = trainer._training_step(model, rasterizer, batch, optimizer)
        out[f"step{step}_metrics"] = np.asarray([metrics[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")],
                                                dtype=np.float64)
        for k in leaves:
            out[f"step{step}_grad_{k}"] = getattr(model, k).grad.detach().numpy().copy()
            out[f"step{step}_param_{k}"] = getattr(model, k).detach().numpy().copy()
    st = optimizer.state[model.xyz]
    out["step2_adam_exp_avg_xyz"] = st["exp_avg"].numpy().copy()
    out["step2_adam_exp_avg_sq_xyz"] = st["exp_avg_sq"].numpy().copy()

    path = os.path.join(HERE, "harness_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB;", "step metrics:", out["step1_metrics"],
          out["step2_metrics"])


if __name__ == "__main__":
    main()
