"""Capture golden vectors of the reference's OPTIMISATION LOOP (BASELINE config C3 "full
src/train.py Adam loop"): /root/reference/src/training/trainer.py:782-859 -- step, xyz-gradient accumulation
(:786-791), densify cadence (:794-828), opacity reset (:841-850), learning-rate update (:852), scale clamp (:857-859),
in that order, over several iterations.

RUNS IN THE AUTHORING CONTAINER ONLY: it imports the reference's own Python from /root/reference (which never
travels) and writes ``tests/golden/loop_golden.npz`` -- inputs and expected outputs, data only.

How: the reference's real ``GaussianTrainer.train`` is run.  Only its DATA PLUMBING is replaced, on a subclass: scene
geometry (a fixed extent), the initial model (a seeded one), video loaders / view lists (a fixed cyclic sequence of
view batches), validation, checkpoints and plots (no-ops).  The loop body, ``_training_step``, the Adam optimizer, the
model's ``densify_and_prune`` with its Adam-state splicing, the opacity reset, the schedule and the clamp are the
reference's own code, running on top of the reference's ``GSplatBackend`` with this repo's CPU oracle bound as the
module ``gsplat`` (AMP off: CPU).  The split noise of every densify event (``torch.randn_like``,
gaussian_model.py:302) is recorded so that the counterpart can be driven with the same draws.

Two runs:
  * ``plain``   iterations 0..7, SH degree stepping every 2 iterations, opacity reset every 3 -- no densify event
                (the reference only densifies after iteration 5000): runs on CPU in the counterpart's tests too;
  * ``densify`` iterations 5000..5010 with densify every 4 and opacity reset every 5 iterations: two densify events,
                three resets, population changes.

    python tests/golden/make_loop_golden.py
"""
import os
import sys
import tempfile
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

FIELDS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
N, W, H, B, V = 300, 48, 32, 2, 5
EXTENT = 1.0


def scene_leaves(seed, n=N):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    return {
        "xyz": t(rng.uniform(-1.0, 1.0, (n, 3))),
        "features_dc": t(rng.uniform(-1.0, 1.0, (n, 1, 3)) * 1.5),
        "features_rest": t(rng.normal(0.0, 0.05, (n, 15, 3))),
        "opacity": t(rng.uniform(-2.0, 2.0, (n, 1))),                      # logits
        "scaling": t(np.log(rng.uniform(0.02, 0.4, (n, 3)))),              # some above the clamp ceiling ln(0.2)
        "rotation": t(rng.normal(0.0, 1.0, (n, 4))),
    }


class DrivenTrainer(GaussianTrainer):
    """The reference trainer with its data plumbing replaced (see the module docstring)."""

    def __init__(self, cfg, model, batches):
        super().__init__(cfg, torch.device("cpu"), wandb_run=None)
        self._model, self._batches = model, batches
        self.trace = []          # per iteration: (iteration, n before the step, metrics)
        self.noise = []          # split noise per densify event (may be empty)

    def _compute_scene_geometry(self, merged_data):
        return EXTENT, np.ones(3), 0, 4.0, np.ones(len(merged_data["points_3d"]), dtype=bool)

    def _initialize_gaussians(self, merged_data):
        return self._model

    def _get_loader(self, video_path):
        loader = MagicMock()
        loader.get_frame.return_value = np.zeros((H, W, 3), dtype=np.uint8)
        return loader

    def debug_reprojection(self, *a, **k):
        return np.zeros((H, W, 3), dtype=np.uint8)

    def _build_views(self, merged_data):
        return [{"dummy": i} for i in range(V)]

    def _split_views(self, all_views):
        return all_views, []

    def _prepare_video_loaders(self, all_views):
        pass

    def _create_train_loader(self, train_views):
        def gen():
            i = 0
            while True:
                yield self._batches[i % len(self._batches)]
                i += 1
        return gen()

    def _validate(self, *a, **k):
        return None

    def _save_checkpoint(self, *a, **k):
        pass

    def draw_graphs(self):
        pass

    def _training_step(self, gaussians, rasterizer, batch, optimizer):
        n = int(gaussians.xyz.shape[0])
        m = super()._training_step(gaussians, rasterizer, batch, optimizer)
        self.trace.append((self.iteration, n, [m[k] for k in ("loss", "l1", "ssim", "psnr", "scale_reg")]))
        return m


def run(tag, out, cfg, first_iteration, leaves_seed, target_seed):
    from gsplat_amd import scene
    torch.manual_seed(0)
    np.random.seed(0)
    K = scene.make_intrinsics(W, H)
    poses = [scene.make_camera(v, V) for v in range(V)]
    # targets: renders of a second scene through the reference's own backend, clamped to [0,1]
    target = GaussianModel(N, "cpu")
    with torch.no_grad():
        for k, v in scene_leaves(target_seed).items():
            getattr(target, k).copy_(v)
    rasterizer = GaussianRasterizer(K=K, device="cpu", enable_caching=True, backend="gsplat")
    probe = GaussianTrainer(cfg, torch.device("cpu"), wandb_run=None)
    probe.iteration = 3 * cfg.sh_increment_interval
    images = []
    with torch.no_grad():
        for pose in poses:
            images.append(probe._render_view(target, rasterizer, {"pose": pose, "K": K, "width": W, "height": H})
                          ["render"].clamp(0, 1))
    views = [{"image": img, "pose": pose, "K": K, "width": W, "height": H} for img, pose in zip(images, poses)]
    # the cyclic batch sequence the stub loader yields (the loop's first batch is consumed by the reference's
    # iteration-0 debug render, trainer.py:719-722)
    batch_ids = [[(B * i + j) % V for j in range(B)] for i in range(V)]
    batches = [[views[j] for j in ids] for ids in batch_ids]
    leaves = scene_leaves(leaves_seed)
    model = GaussianModel(N, "cpu")
    with torch.no_grad():
        for k, v in leaves.items():
            getattr(model, k).copy_(v)
    trainer = DrivenTrainer(cfg, model, batches)
    trainer.iteration = first_iteration
    merged = {"points_3d": np.zeros((8, 3)), "colors": np.zeros((8, 3)), "all_poses": [np.stack([p.numpy() for p in poses])],
              "all_intrinsics": [K.numpy()], "video_info": [{"path": "stub.mp4"}], "frame_indices": [list(range(V))]}

    real = torch.randn_like

    def recording(x, *a, **k):
        r = real(x, *a, **k)
        trainer.noise.append((trainer.iteration, r.clone()))
        return r

    torch.randn_like = recording
    try:
        with tempfile.TemporaryDirectory() as tmp:
            final = trainer.train(merged, tmp)
    finally:
        torch.randn_like = real

    out[f"{tag}_K"] = K.numpy()
    out[f"{tag}_poses"] = torch.stack(poses).numpy()
    out[f"{tag}_images"] = torch.stack(images).numpy()
    out[f"{tag}_batch_ids"] = np.asarray(batch_ids, dtype=np.int64)
    for k, v in leaves.items():
        out[f"{tag}_init_{k}"] = v.numpy()
    out[f"{tag}_config"] = np.asarray([first_iteration, cfg.iterations_per_video, cfg.densify_interval,
                                       cfg.opacity_reset_interval, cfg.sh_increment_interval, cfg.densify_grads_threshold,
                                       cfg.densify_min_opacity, cfg.densify_clone_extent_ratio,
                                       cfg.densify_prune_extent_ratio, cfg.scale_clamp_ratio, EXTENT,
                                       int(cfg.max_gaussians)], dtype=np.float64)
    out[f"{tag}_trace_iteration"] = np.asarray([t[0] for t in trainer.trace], dtype=np.int64)
    out[f"{tag}_trace_n"] = np.asarray([t[1] for t in trainer.trace], dtype=np.int64)
    out[f"{tag}_trace_metrics"] = np.asarray([t[2] for t in trainer.trace], dtype=np.float64)
    out[f"{tag}_noise_iteration"] = np.asarray([it for it, _ in trainer.noise], dtype=np.int64)
    for i, (_, r) in enumerate(trainer.noise):
        out[f"{tag}_noise_{i}"] = r.numpy()
    for k in FIELDS:
        out[f"{tag}_final_{k}"] = getattr(final, k).detach().numpy().copy()
    out[f"{tag}_final_n"] = np.int64(final.xyz.shape[0])
    print(tag, "iterations", out[f"{tag}_trace_iteration"].tolist(), "n", out[f"{tag}_trace_n"].tolist(),
          "final n", int(final.xyz.shape[0]), "noise events", [int(i) for i, _ in trainer.noise])
    print(tag, "loss", [round(t[2][0], 5) for t in trainer.trace])


def main():
    torch.set_num_threads(4)
    out = {}
    cfg = TrainingConfig()
    cfg.use_mixed_precision = False
    cfg.iterations_per_video = 8
    cfg.sh_increment_interval = 2
    cfg.opacity_reset_interval = 3
    run("plain", out, cfg, first_iteration=0, leaves_seed=11, target_seed=12)

    cfg = TrainingConfig()
    cfg.use_mixed_precision = False
    cfg.iterations_per_video = 5011
    cfg.densify_interval = 4
    cfg.opacity_reset_interval = 5
    cfg.densify_grads_threshold = 2e-4
    cfg.max_gaussians = 330          # the second event runs into the population cap
    run("densify", out, cfg, first_iteration=5000, leaves_seed=13, target_seed=14)

    path = os.path.join(HERE, "loop_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
