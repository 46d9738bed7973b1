"""Training-step harness around the hot path, single-GPU and view-parallel (SURVEY.md section 8 rows a4 and e).

Counterpart of the reference's per-step logic -- ``GaussianTrainer._training_step``
(/root/reference/src/training/trainer.py:447-519) with its helpers ``_gaussian_params`` (:399-406),
``_stclamp`` (:102-107), ``_compute_ssim`` (:354-371), ``_psnr`` (:70-75), ``_scale_ratio_reg`` (:434-445),
``_setup_optimizer`` (:331-340), ``_update_learning_rate`` (:373-382), ``_active_sh_degree`` (:77-87) -- and of the
optimisation loop around it (``GaussianTrainer.train``, trainer.py:782-859: ``ViewParallelTrainer.run``) -- written
for one-process-per-GPU execution:

* the six parameter tensors of the scene model (/root/reference/src/gaussians/gaussian_model.py:59-88) are views
  into ONE flat fp32 buffer of 59*N floats, and their ``.grad`` tensors are views into one flat gradient bucket;
* with ``world_size`` > 1 the B views of a batch are dealt round-robin to the ranks, every rank renders its own
  views with the HIP rasterizer, and the gradient bucket is summed with ONE all-reduce (RCCL over xGMI) before
  an identical Adam step on every rank.  The reference has no multi-GPU code; the contract here
  is "N-rank result == 1-rank result on the same batch".

The rendering backend is injected (anything with ``render_with_depth``): ``HipBackend`` in production.
Nothing here imports the CPU oracle.
"""

from __future__ import annotations

import contextlib
import math
import warnings
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F



@contextlib.contextmanager
def _quiet_grad_layout():
    """features_dc / features_rest are strided views of the [N,16,3] coefficient rows, and so are their .grad tensors: autograd
    (the non-fused path) accumulates into those views in place, which is the point -- torch notes that such a layout "may
    impair performance"; the copy it would prefer is what the layout saves.  The note is silenced only around THIS package's
    own gradient binding and backward() calls: other models of the host process keep it."""
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="grad and param do not obey the gradient layout contract")
        yield


# (name, per-gaussian shape) in the order of the reference's optimizer groups (trainer.py:331-340)
FIELDS = (("xyz", (3,)), ("features_dc", (1, 3)), ("features_rest", (15, 3)), ("opacity", (1,)),
          ("scaling", (3,)), ("rotation", (4,)))
FLOATS_PER_GAUSSIAN = sum(math.prod(s) for _, s in FIELDS)  # 59
# How the flat buckets (parameters, gradients, Adam moments) are laid out: segment-major, and the SH coefficients as ONE
# segment of [N,16,3] rows -- exactly what the rasterizer reads and its backward writes, so a step neither assembles
# cat(features_dc, features_rest) (192 B per gaussian read + written) nor splits the gradient again.  ``features_dc`` /
# ``features_rest`` are the strided views [:, :1] / [:, 1:] of those rows.  The rows come first: their 16-byte alignment
# (the kernels move them as float4) then never depends on N.
STORAGE = (("sh", 48), ("xyz", 3), ("opacity", 1), ("scaling", 3), ("rotation", 4))
_ADAM_GROUP_FIELD = {"xyz": "xyz", "f_dc": "features_dc", "f_rest": "features_rest", "opacity": "opacity",
                     "scaling": "scaling", "rotation": "rotation"}


def bucket_views(flat: torch.Tensor, n: int) -> Dict[str, torch.Tensor]:
    """The named views of a flat 59 n bucket: the six fields of the reference model + ``sh`` (the [n,16,3] rows)."""
    out, off = {}, 0
    for seg, width in STORAGE:
        part = flat[off:off + n * width]
        if seg == "sh":
            rows = part.view(n, 16, 3)
            out["sh"], out["features_dc"], out["features_rest"] = rows, rows[:, :1], rows[:, 1:]
        else:
            out[seg] = part.view(n, width)
        off += n * width
    return out


def pack_bucket(get, n: int) -> torch.Tensor:
    """A flat bucket in STORAGE order from per-field tensors (``get(name)``)."""
    parts = []
    for seg, width in STORAGE:
        if seg == "sh":
            parts.append(torch.cat([get("features_dc").reshape(n, 1, 3), get("features_rest").reshape(n, 15, 3)], dim=1).reshape(-1))
        else:
            parts.append(get(seg).reshape(-1))
    return torch.cat(parts)


@dataclass
class StepConfig:
    """The subset of the reference's ``TrainingConfig`` that shapes a step
    (/root/reference/src/gaussians/training_config.py:43-73); same defaults."""
    lambda_dssim: float = 0.2
    scale_reg_max_ratio: float = 10.0
    scale_reg_weight: float = 0.1
    sh_degree_max: int = 3
    sh_increment_interval: int = 1000
    position_lr_init: float = 1.6e-4
    position_lr_final: float = 1.6e-7
    position_lr_max_steps: float = 3e5
    # The reference's default is True (training_config.py:63: autocast around the render call + GradScaler,
    # trainer.py:31,468-470,505-511).  The rasterizer computes in fp32 either way (its inputs are fp32), so what
    # mixed precision changes is that the backward pass sees an upstream gradient multiplied by the loss scale
    # (65536 at the start) and that a step with a non-finite gradient is skipped.  Off by default HERE because the
    # reference's goldens can only be captured with it off (CPU); with it on, the step reproduces them too
    # (tests/test_amp.py).  Ignored on a CPU device.
    use_mixed_precision: bool = False


@dataclass
class LoopConfig:
    """The fields of the reference's ``TrainingConfig`` that shape the loop around the step
    (/root/reference/src/gaussians/training_config.py:21-39,54); same defaults.  ``densify_after`` is the constant
    the reference hard-codes (``self.iteration > 5000``, trainer.py:795)."""
    iterations: int = 300_000
    densify_after: int = 5000
    densify_interval: int = 100
    opacity_reset_interval: int = 5000
    opacity_reset_ceiling: float = 0.01
    densify_grads_threshold: float = 5e-4
    densify_min_opacity: float = 0.005
    densify_max_screen_size: float = 12.0
    densify_clone_extent_ratio: float = 0.1
    densify_prune_extent_ratio: float = 2.0
    scale_clamp_ratio: float = 0.2
    max_gaussians: int = 10_000_000


class SplatModel:
    """Scene parameters with the reference model's names, shapes and activations
    (gaussian_model.py:59-108), stored as views of one flat buffer (+ one flat gradient bucket)."""

    def __init__(self, n: int, device="cuda"):
        self._bind(int(n), torch.zeros(FLOATS_PER_GAUSSIAN * int(n), dtype=torch.float32, device=device))
        with torch.no_grad():  # the reference's initial values (gaussian_model.py:66-84)
            self.scaling.fill_(-3.0)
            self.rotation[:, 0] = 1.0

    def _bind(self, n: int, flat: torch.Tensor) -> None:
        """(Re)create the six parameter views, their gradient views and the densification buffers over a flat
        bucket of 59 * n floats (gaussian_model.py:66-80)."""
        self.n = n
        self.flat = flat
        self.flat_grad = torch.zeros_like(flat)
        views, grads = bucket_views(self.flat, n), bucket_views(self.flat_grad, n)
        for name, shape in FIELDS:
            p = views[name]
            assert tuple(p.shape) == (n, *shape)
            p.requires_grad_(True)
            with _quiet_grad_layout():
                p.grad = grads[name]
            setattr(self, name, p)
        self.sh, self.sh_grad = views["sh"], grads["sh"]   # the coefficient rows as the rasterizer takes them (no copy)
        dev = flat.device
        self.xyz_gradient_accum = torch.zeros(n, 3, device=dev)
        self.xyz_gradient_count = torch.zeros(n, 1, device=dev)
        self.max_radii_2D = torch.zeros(n, device=dev)

    @torch.no_grad()
    def accumulate_xyz_grad(self) -> None:
        """What the reference's loop does after every step (trainer.py:786-791)."""
        self.xyz_gradient_accum += torch.norm(self.xyz.grad, dim=-1, keepdim=True)
        self.xyz_gradient_count += 1

    @torch.no_grad()
    def densify_and_prune(self, grads_threshold: float, min_opacity: float, extent: float,
                          max_screen_size: Optional[float] = None, optimizer=None,
                          clone_extent_ratio: float = 0.1, prune_extent_ratio: float = 2.0,
                          max_gaussians: Optional[int] = None, noise: Optional[torch.Tensor] = None) -> Dict:
        """Counterpart of ``GaussianModel.densify_and_prune`` (gaussian_model.py:110-231) with the same selection
        rules, row order and Adam-state splicing, but ONE data pass: the masks give a source-row map for the final
        population ([survivors] + [clones] + [first children] + [second children], minus the pruned rows), and
        ``gsa_densify_gather`` builds the new 59 N' bucket and both Adam moment buckets from it.
        ``noise``: [2 * n_split, 3] standard-normal draws for the split offsets (gaussian_model.py:302; default:
        drawn on the device).  ``optimizer``: ``FusedBucketAdam`` or ``torch.optim.Adam`` over this model, or None.
        ``max_screen_size`` is accepted in the reference's position (its loop passes it by keyword, trainer.py:798-807)
        and, as in the reference's body, not used."""
        from . import stages as S
        if not self.flat.is_cuda:
            raise RuntimeError("SplatModel.densify_and_prune runs on the HIP device only (no CPU path)")
        n, dev = self.n, self.flat.device
        stats = {"cloned": 0, "split": 0, "pruned": 0, "n_before": n, "capped": False}
        grads_norm = torch.norm(self.xyz_gradient_accum / (self.xyz_gradient_count + 1e-8), dim=-1)
        max_scale = torch.exp(self.scaling).amax(dim=-1)
        opacity = torch.sigmoid(self.opacity).squeeze(-1)
        hot = (grads_norm >= grads_threshold) & (opacity > min_opacity)
        clone_mask = hot & (max_scale <= extent * clone_extent_ratio)
        split_mask = hot & (max_scale > extent * clone_extent_ratio)
        if max_gaussians is not None and n >= int(max_gaussians):       # population cap (:167-180)
            clone_mask, split_mask = torch.zeros_like(clone_mask), torch.zeros_like(split_mask)
            stats["capped"] = True
        elif max_gaussians is not None:
            allowed = int(max_gaussians) - n
            if int(clone_mask.sum()) + int(split_mask.sum()) > allowed:
                stats["capped"] = True
                combined = clone_mask | split_mask
                keep = torch.zeros_like(combined)
                if allowed > 0:
                    idx = combined.nonzero().view(-1)
                    keep[idx[torch.topk(grads_norm[idx], k=allowed).indices]] = True
                clone_mask, split_mask = clone_mask & keep, split_mask & keep
        idx_keep = (~split_mask).nonzero().view(-1)
        idx_clone = clone_mask.nonzero().view(-1)
        idx_split = split_mask.nonzero().view(-1)
        n_clone, n_split = int(idx_clone.numel()), int(idx_split.numel())
        stats["cloned"], stats["split"] = n_clone, n_split
        src = torch.cat([idx_keep, idx_clone, idx_split, idx_split])
        kind = torch.cat([torch.zeros_like(idx_keep), torch.ones_like(idx_clone),
                          torch.full((2 * n_split,), 2, dtype=idx_split.dtype, device=dev)])
        if isinstance(noise, _LazyNoise):   # (the reference draws nothing when no gaussian is split, :288-289)
            noise = noise.draw(n_split) if n_split else torch.zeros(0, 3, device=dev)
        if noise is None:
            noise = torch.randn(2 * n_split, 3, device=dev)
        noise = noise.to(dev, torch.float32).reshape(2 * n_split, 3)
        log_split = torch.log(torch.tensor(1.6))                         # fp32, as the reference computes it (:298)
        # prune rule on the population after clone + split (:213-215): low opacity OR too large
        scale_s = self.scaling[src] - (kind == 2).float()[:, None] * log_split.to(dev)
        low_op = opacity[src] < min_opacity
        too_big = torch.exp(scale_s).amax(dim=-1) > extent * prune_extent_ratio
        prune = low_op | too_big
        stats["pruned"], stats["pruned_low_op"], stats["pruned_too_big"] = int(prune.sum()), int(low_op.sum()), int(too_big.sum())
        final = ~prune
        noise_full = torch.zeros(src.numel(), 3, device=dev)
        noise_full[src.numel() - 2 * n_split:] = noise
        src_f, kind_f, noise_f = src[final], kind[final], noise_full[final]
        old_m = old_v = None
        if isinstance(optimizer, FusedBucketAdam):
            old_m, old_v = optimizer.exp_avg, optimizer.exp_avg_sq
        elif optimizer is not None:                                      # torch.optim.Adam: gather its per-parameter state
            def moment(key):
                def get(name):
                    p = getattr(self, name)
                    return optimizer.state[p][key] if p in optimizer.state and key in optimizer.state[p] else torch.zeros_like(p)
                return pack_bucket(get, n)
            old_m, old_v = moment("exp_avg"), moment("exp_avg_sq")
        widths = [width for _, width in STORAGE]
        names = [name for name, _ in STORAGE]
        new_p, new_m, new_v = S.densify_gather(n, widths, names.index("xyz"), names.index("scaling"), float(log_split),
                                               src_f, kind_f, noise_f, self.flat.detach(), old_m, old_v)
        old_params = self.parameters()
        radii = torch.where(kind_f == 0, self.max_radii_2D[src_f], torch.zeros((), device=dev))
        self._bind(int(src_f.numel()), new_p)            # also resets the gradient accumulators (:227-228)
        self.max_radii_2D = radii
        if isinstance(optimizer, FusedBucketAdam):
            optimizer.rebind(self, new_m, new_v)
        elif optimizer is not None:
            m_views, v_views = bucket_views(new_m, self.n), bucket_views(new_v, self.n)
            for group, old, (name, _) in zip(optimizer.param_groups, old_params, FIELDS):
                new = getattr(self, name)
                state = optimizer.state.pop(old, None)
                group["params"] = [new]
                if state is not None:
                    state = dict(state)
                    state["exp_avg"], state["exp_avg_sq"] = m_views[name], v_views[name]
                    optimizer.state[new] = state
        stats["n_after"] = self.n
        return stats

    def parameters(self) -> List[torch.Tensor]:
        return [getattr(self, name) for name, _ in FIELDS]

    def check_bound(self) -> None:
        """Every parameter (and its gradient) must still be a view of the flat buckets: the fused Adam step, the
        exchange and the densify gather work on ``flat`` / ``flat_grad``.  Rebinding ``p.data`` -- what the reference's
        opacity reset does (trainer.py:850) -- detaches a parameter silently; use ``reset_opacity`` /
        ``clamp_scaling`` (in place) instead."""
        views, grads = bucket_views(self.flat, self.n), bucket_views(self.flat_grad, self.n)
        for name, _ in FIELDS:
            p, want = getattr(self, name), views[name]
            if p.data_ptr() != want.data_ptr() or p.shape != want.shape or p.stride() != want.stride():
                raise RuntimeError(f"SplatModel.{name} no longer lives in the flat parameter bucket (was .data rebound?)")
            if p.grad is None or p.grad.data_ptr() != grads[name].data_ptr() or p.grad.stride() != grads[name].stride():
                raise RuntimeError(f"SplatModel.{name}.grad no longer lives in the flat gradient bucket")

    @torch.no_grad()
    def reset_opacity(self, ceiling: float = 0.01) -> None:
        """The reference's periodic opacity reset (trainer.py:841-850): every opacity becomes min(current, ceiling),
        clamped to [1e-6, 1 - 1e-6] before the inverse sigmoid -- written IN PLACE into the bucket view."""
        current = torch.sigmoid(self.opacity)
        target = torch.minimum(current, torch.full_like(current, ceiling)).clamp(min=1e-6, max=1.0 - 1e-6)
        self.opacity.copy_(torch.logit(target))

    @torch.no_grad()
    def clamp_scaling(self, extent: float, scale_clamp_ratio: float) -> None:
        """The reference's per-step ceiling on the log-scale (trainer.py:857-859), in place."""
        self.scaling.clamp_(max=math.log(extent * scale_clamp_ratio + 1e-9))

    @torch.no_grad()
    def load(self, **tensors: torch.Tensor) -> "SplatModel":
        for name, value in tensors.items():
            getattr(self, name).copy_(value.reshape(getattr(self, name).shape))
        return self

    def activated(self) -> Dict[str, torch.Tensor]:
        """What the reference hands to the backend (trainer.py:399-406)."""
        return {
            "means3D": self.xyz,
            "scales": torch.exp(self.scaling),
            "rotations": F.normalize(self.rotation, dim=-1),
            "opacities": torch.sigmoid(self.opacity),
            "shs": torch.cat([self.features_dc, self.features_rest], dim=1),
        }

    def zero_grad(self) -> None:
        self.flat_grad.zero_()


# ---- the reference's per-view loss pieces ------------------------------------------------------------------
def straight_through_clamp(x: torch.Tensor) -> torch.Tensor:
    """Forward clamp to [0,1], identity backward (trainer.py:102-107)."""
    return x + (x.clamp(0.0, 1.0) - x).detach()


def ssim_3x3(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Mean SSIM with a 3x3 box window, zero padding counted in the mean (trainer.py:354-371). [1,3,H,W]."""
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    box = lambda t: F.avg_pool2d(t, 3, 1, padding=1)  # noqa: E731
    mu_a, mu_b = box(a), box(b)
    var_a = box(a * a) - mu_a * mu_a
    var_b = box(b * b) - mu_b * mu_b
    cov = box(a * b) - mu_a * mu_b
    num = (2 * mu_a * mu_b + c1) * (2 * cov + c2)
    den = (mu_a * mu_a + mu_b * mu_b + c1) * (var_a + var_b + c2)
    return (num / den).mean()


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """-10 log10(mse); 100 below 1e-10 (trainer.py:70-75)."""
    mse = float(F.mse_loss(pred, target))
    return 100.0 if mse < 1e-10 else -10.0 * math.log10(mse)


def scale_ratio_reg(scales: torch.Tensor, max_ratio: float, weight: float) -> torch.Tensor:
    """Hinge on max/min activated scale per gaussian (trainer.py:434-445)."""
    ratio = scales.amax(dim=-1) / scales.amin(dim=-1).clamp(min=1e-8)
    return weight * (ratio.clamp(min=max_ratio) - max_ratio).mean()


def view_loss(rendered: torch.Tensor, target: torch.Tensor, lambda_dssim: float):
    """(1 - lambda) L1 + lambda (1 - SSIM) on the straight-through-clamped render (trainer.py:485-492)."""
    img = straight_through_clamp(rendered)
    l1 = F.l1_loss(img, target)
    # .contiguous(): PyTorch 2.10+rocm7.0 returns a WRONG avg_pool2d gradient on the GPU for the permuted
    # (channels-last-strided) [1,3,H,W] view that trainer.py:487-490 builds (measured on MI355X: SSIM gradient
    # off by more than its own magnitude; exact once the view is made contiguous).  Same values either way.
    ssim = ssim_3x3(img.permute(2, 0, 1)[None].contiguous(), target.permute(2, 0, 1)[None].contiguous())
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim), l1, ssim


class _LazyNoise:
    """Split noise that is drawn once the number of split gaussians is known (``ViewParallelTrainer.run``)."""

    def __init__(self, fn, iteration: int):
        self.fn, self.iteration = fn, iteration

    def draw(self, n_split: int) -> torch.Tensor:
        return self.fn(self.iteration, n_split)


def _adam_groups(model: SplatModel, c: StepConfig):
    """The reference's six optimizer groups (trainer.py:331-340)."""
    return [
        {"params": [model.xyz], "lr": c.position_lr_init, "name": "xyz"},
        {"params": [model.features_dc], "lr": 0.0025, "name": "f_dc"},
        {"params": [model.features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
        {"params": [model.opacity], "lr": 0.05, "name": "opacity"},
        {"params": [model.scaling], "lr": 0.005, "name": "scaling"},
        {"params": [model.rotation], "lr": 0.001, "name": "rotation"},
    ]


class FusedBucketAdam:
    """torch.optim.Adam (betas 0.9 / 0.999, eps 1e-15, no weight decay) over the model's ONE flat 59 N bucket as a
    single HIP kernel (``gsa_adam_step``: 4 arrays read, 3 written, once) instead of torch's multi-pass foreach
    update.  Exposes ``param_groups`` (name, lr) and ``state[param]`` (views of the flat moments) like the torch
    optimizer it replaces, so schedules and checkpoints keep working."""

    def __init__(self, model: SplatModel, groups, betas=(0.9, 0.999), eps: float = 1e-15):
        self.model, self.betas, self.eps = model, betas, eps
        self.param_groups = groups
        self.step_count = 0
        self.rebind(model, torch.zeros_like(model.flat), torch.zeros_like(model.flat))

    def rebind(self, model: SplatModel, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> None:
        """Adopt a model and its moment buckets (construction; after densify / prune: the spliced buckets)."""
        assert exp_avg.numel() == exp_avg_sq.numel() == model.flat.numel()
        self.model, self.exp_avg, self.exp_avg_sq = model, exp_avg, exp_avg_sq
        m_views, v_views = bucket_views(exp_avg, model.n), bucket_views(exp_avg_sq, model.n)
        self.state = {}
        for g in self.param_groups:   # the reference's six groups, by name (trainer.py:331-340)
            field = _ADAM_GROUP_FIELD[g["name"]]
            p = getattr(model, field)
            g["params"] = [p]
            self.state[p] = {"exp_avg": m_views[field], "exp_avg_sq": v_views[field]}
        self.segment_end, off = [], 0
        for _, width in STORAGE:
            off += model.n * width
            self.segment_end.append(off)

    def step(self, skip_if_nonzero: Optional[torch.Tensor] = None) -> None:
        """``skip_if_nonzero`` (int32 [1] on the device): the update is skipped when it is set -- the caller then takes
        ``step_count`` back by one once it has read the flag (GradScaler.step does not call the optimizer at all)."""
        from . import stages as S
        self.step_count += 1
        lr = {g["name"]: float(g["lr"]) for g in self.param_groups}
        # one learning rate per segment; the coefficient rows are striped: 3 floats of f_dc, 45 of f_rest, per gaussian
        lrs = [lr["f_rest"] if seg == "sh" else lr[seg] for seg, _ in STORAGE]
        stripes = [(48, 3, lr["f_dc"]) if seg == "sh" else None for seg, _ in STORAGE]
        with torch.no_grad():
            S.adam_step(self.model.flat, self.model.flat_grad, self.exp_avg, self.exp_avg_sq, self.segment_end, lrs,
                        self.step_count, self.betas[0], self.betas[1], self.eps, skip_if_nonzero=skip_if_nonzero,
                        stripes=stripes)


class ViewParallelTrainer:
    """One optimisation step over a batch of views, sharded by view across ``world_size`` ranks.

    ``fused=True`` (HIP device only): the per-view loss and its gradient come from ``gsa_view_loss`` (two streaming
    passes instead of ~20 torch passes per view) and Adam is one ``gsa_adam_step`` over the flat bucket."""

    def __init__(self, model: SplatModel, backend, config: Optional[StepConfig] = None,
                 process_group=None, rank: int = 0, world_size: int = 1, fused: bool = False,
                 factored_exchange: Optional[bool] = None, overlap_views=True):
        self.model, self.backend = model, backend
        self.cfg = config or StepConfig()
        self.group, self.rank, self.world = process_group, rank, world_size
        self.iteration = 0
        self.fused = bool(fused)
        # fused step: truthy (default) -- whole views on two alternating HIP streams, their speculation checked once in front
        # of the optimizer (gsplat_amd.pipeline.ViewPipeline, the orchestrator bench.py times); False -- one stream
        self.overlap_views = bool(overlap_views)
        # fused step: ``async_metrics = True`` returns the metrics of the PREVIOUS step (the first step returns its own):
        # the step then ends without a host read-back and the host queues the next step while this one runs -- what the
        # reference's loop needs them for (a progress line, trainer.py:513-519) does not care which step they are of.
        # Default False: a step returns its own metrics (the golden tests compare step by step).
        self.async_metrics = False
        self._pending_report = None
        self._pipe = None
        if self.fused and not model.flat.is_cuda:
            raise RuntimeError("ViewParallelTrainer(fused=True) needs the model on a HIP device (no CPU path)")
        # fused: the SH-coefficient gradient is kept (and, for world > 1, exchanged) in factored form
        # (see _training_step_fused)
        self.factored = self.fused if factored_exchange is None else bool(factored_exchange)
        if self.factored and not self.fused:
            raise ValueError("factored_exchange needs fused=True (it is a property of the HIP backward pass)")
        groups = _adam_groups(model, self.cfg)
        if self.fused:
            self.optimizer = FusedBucketAdam(model, groups)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)  # trainer.py:331-340
        # mixed precision (trainer.py:31): torch's GradScaler on the autograd path; the fused path keeps the same
        # state itself (init 65536, x2 after 2000 good steps, x0.5 after a non-finite gradient: GradScaler's defaults)
        self.amp = bool(self.cfg.use_mixed_precision) and model.flat.is_cuda
        self.scaler = torch.amp.GradScaler("cuda") if self.amp and not self.fused else None
        self.loss_scale, self._good_steps, self.skipped_steps = 65536.0, 0, 0

    def active_sh_degree(self) -> int:
        return min(int(self.iteration // self.cfg.sh_increment_interval), int(self.cfg.sh_degree_max))

    def update_learning_rate(self) -> float:
        c = self.cfg
        progress = min(self.iteration / c.position_lr_max_steps, 1.0)
        lr = c.position_lr_init * (c.position_lr_final / c.position_lr_init) ** progress
        for group in self.optimizer.param_groups:
            if group.get("name") == "xyz":
                group["lr"] = lr
        return lr

    def run(self, loader, extent: float, loop: Optional[LoopConfig] = None, split_noise=None, on_step=None) -> SplatModel:
        """The reference's optimisation loop (``GaussianTrainer.train``, trainer.py:782-859 and :989) from
        ``self.iteration`` up to ``loop.iterations``, in the reference's order per iteration:

          1. ``batch = next(loader)``; ``training_step(batch)``                                   (:783-784)
          2. xyz-gradient accumulation from the EXCHANGED gradient                                (:786-791)
          3. densify / prune when ``iteration > densify_after and iteration % densify_interval == 0`` (:794-807)
          4. opacity reset when ``iteration % opacity_reset_interval == 0 and iteration > 0``     (:841-850)
          5. learning-rate update                                                                 (:852)
          6. scale clamp                                                                          (:857-859)
          7. ``iteration += 1``                                                                   (:989)

        ``loader``: an iterator of view batches -- the SAME sequence on every rank.  ``extent``: the scene extent the
        thresholds scale with (trainer.py:659-661).  ``split_noise(iteration, n_split) -> [2 n_split, 3]``: the
        standard-normal draws of a densify event; by default drawn on the device from a generator seeded with the
        iteration, hence identical on every rank (the reference draws from the global generator,
        gaussian_model.py:302).  ``on_step(iteration, n_before_step, metrics, densify_stats or None)`` is called after
        every iteration (logging / tests).  Steady state has no host wait besides the step's metric read-back;
        a densify event reads back the population counts, as the reference does."""
        loop = loop or LoopConfig()
        m = self.model
        while self.iteration < loop.iterations:
            m.check_bound()
            batch = next(loader)
            n_before = m.n
            metrics = self.training_step(batch)
            m.accumulate_xyz_grad()
            stats = None
            if self.iteration > loop.densify_after and self.iteration % loop.densify_interval == 0:
                noise_fn = split_noise if split_noise is not None else self._default_split_noise
                stats = m.densify_and_prune(
                    grads_threshold=loop.densify_grads_threshold, min_opacity=loop.densify_min_opacity, extent=extent,
                    max_screen_size=loop.densify_max_screen_size, optimizer=self.optimizer,
                    clone_extent_ratio=loop.densify_clone_extent_ratio,
                    prune_extent_ratio=loop.densify_prune_extent_ratio, max_gaussians=int(loop.max_gaussians),
                    noise=_LazyNoise(noise_fn, self.iteration))
            if self.iteration % loop.opacity_reset_interval == 0 and self.iteration > 0:
                m.reset_opacity(loop.opacity_reset_ceiling)
            self.update_learning_rate()
            m.clamp_scaling(extent, loop.scale_clamp_ratio)
            if on_step is not None:
                on_step(self.iteration, n_before, metrics, stats)
            self.iteration += 1
        return m

    def _default_split_noise(self, iteration: int, n_split: int) -> torch.Tensor:
        dev = self.model.flat.device
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x5EED0000 + int(iteration))
        return torch.randn(2 * n_split, 3, device=dev, generator=gen)

    def _all_reduce(self, t: torch.Tensor, async_op: bool = False):
        if self.world > 1:
            import torch.distributed as dist
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return None

    def _all_gather_slabs(self, slabs: torch.Tensor):
        """In-place all-gather (asynchronous: returns the work handle): rank r owns ``slabs[r]``; afterwards every rank
        holds every slab."""
        import torch.distributed as dist
        if dist.get_backend(self.group) == "nccl":
            return dist.all_gather_into_tensor(slabs, slabs[self.rank], group=self.group, async_op=True)
        return dist.all_gather(list(slabs.unbind(0)), slabs[self.rank].clone(), group=self.group, async_op=True)

    def training_step(self, batch: Sequence[Dict]) -> Dict[str, float]:
        """``batch``: B dicts {image [H,W,3], pose [4,4] world->camera, width, height} -- the SAME list on every
        rank (trainer.py:270-281).  Rank r renders views r, r + world, ...; the summed gradient equals the
        single-process one: each view's loss is divided by B, the scale regulariser is added on rank 0 only."""
        if self.fused:
            return self._training_step_fused(batch)
        m, c = self.model, self.cfg
        m.zero_grad()
        params = m.activated()
        degree = self.active_sh_degree()
        dev = m.flat.device
        n_views = len(batch)
        total = torch.zeros((), device=dev)
        stats = torch.zeros(3, device=dev, dtype=torch.float64)  # l1, ssim, psnr sums over the local views
        for view in batch[self.rank::self.world]:
            viewpoint = {"world_view_transform": view["pose"], "image_width": view["width"],
                         "image_height": view["height"]}
            with torch.autocast(device_type=dev.type, enabled=self.amp):   # trainer.py:468-470
                out = self.backend.render_with_depth(params, viewpoint, bg_color=torch.zeros(3, device=dev),
                                                     render_mode="RGB", device=str(dev), sh_degree=degree)
            rendered, target = out["render"], view["image"]
            loss, l1, ssim = view_loss(rendered, target, c.lambda_dssim)
            total = total + loss
            stats += torch.tensor([float(l1.detach()), float(ssim.detach()), psnr(rendered.detach().clamp(0, 1), target)],
                                  dtype=torch.float64, device=dev)
        total = total / n_views
        reg = scale_ratio_reg(params["scales"], c.scale_reg_max_ratio, c.scale_reg_weight)
        if self.rank == 0:
            total = total + reg
        with _quiet_grad_layout():
            if self.scaler is not None:      # trainer.py:505-508; the exchange sums the SCALED gradients, so every rank
                self.scaler.scale(total).backward()   # unscales the same numbers and takes the same skip decision
            else:
                total.backward()
        if self.scaler is not None:
            self._all_reduce(m.flat_grad)
            before = self.scaler.get_scale()
            self.scaler.step(self.optimizer)      # unscale_ (in place, on the bucket views) + inf check + Adam
            self.scaler.update()
            self.loss_scale = self.scaler.get_scale()
            self.skipped_steps += int(self.loss_scale < before)
        else:
            self._all_reduce(m.flat_grad)  # the one exchange step: 59*N floats, SUM
            self.optimizer.step()

        report = torch.cat([stats, total.detach().double()[None]])
        self._all_reduce(report)
        return {"loss": float(report[3]), "l1": float(report[0]) / n_views, "ssim": float(report[1]) / n_views,
                "psnr": float(report[2]) / n_views, "scale_reg": float(reg.detach())}

    def _training_step_fused(self, batch: Sequence[Dict], exact: bool = False) -> Dict[str, float]:
        """Same step on the HIP kernels end to end.  Besides the loss and Adam kernels, the gradients never pass
        through autograd's accumulation: the activations (trainer.py:399-406) are evaluated once without a graph,
        every view's backward ADDS its activated-parameter gradients into one bucket (``grad_accumulate``), and the
        activation derivatives are applied once, writing the raw-parameter gradients into ``flat_grad``.
        One host read-back, at the end (the metrics) -- none with ``async_metrics`` (they then arrive one step late).

        ``factored`` (default): 48 of the 59 gradient floats per gaussian are the SH-coefficient gradient, and per view
        that is the outer product SH-basis(view direction) x d loss / d rgb.  The backward passes therefore leave only
        those 3 floats per gaussian and view, and ONE kernel rebuilds the coefficient gradient of the whole batch
        (``gsa_sh_grad_expand``) -- no 48-float read-modify-write per view.  world > 1: every rank holds the whole
        batch's poses, so the ranks all-gather the 3 N floats per view, all-reduce only the 11 N geometry floats, and
        each rebuilds the coefficient gradient itself (summed in batch order, so bit-identical on every rank) --
        instead of all-reducing 59 N floats."""
        from . import stages as S
        m, c = self.model, self.cfg
        degree = self.active_sh_degree()
        dev = m.flat.device
        n, n_views = m.n, len(batch)
        K = self.backend.K.to(dev)[None]
        bg = torch.zeros(1, 3, device=dev)
        # activations (trainer.py:399-406), once, without a graph: one kernel
        # (geometry only: the coefficient rows of the model ARE what the rasterizer reads)
        scales, quats, opac, _ = S.activate_fwd(m.scaling, m.rotation, m.opacity)
        shs = m.sh.detach()
        # gradient bucket of the ACTIVATED parameters; the xyz and coefficient parts ARE the raw gradients (identity)
        if getattr(self, "_act_grad", None) is None or self._act_grad[1].shape[0] != n:
            self._act_grad = (None, torch.empty(n, 4, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
        g_means, g_shs = m.xyz.grad, m.sh_grad
        _, g_quats, g_scales, g_opac = self._act_grad
        grad_out = [g_means, g_quats, g_scales, g_opac, g_shs]
        leaves = [m.xyz.detach(), quats, scales, opac, shs]
        mine = list(batch[self.rank::self.world])
        # The views of the step share their gaussians: ONE projection launch for all of them reads every gaussian's geometry
        # and coefficient row once (stages.project_views); views of different frame sizes are projected one by one.  In one
        # process their projection BACKWARD is one launch as well, behind the last sweep (stages.project_views_bwd: geometry
        # read once, every gradient written once, the coefficient gradient formed in registers -- no per-view rgb gradients
        # and no expansion kernel).  With several ranks the per-view factored form stays: its gathers overlap the next view.
        same_frame = len(mine) > 1 and len({(v["width"], v["height"]) for v in mine}) == 1
        batch_bwd = same_frame and self.world == 1 and degree >= 1 and len(mine) <= 8 and m.sh.shape[1] == 16
        factored = self.factored and not batch_bwd
        # ... with several ranks (r05, ``self.rank_batch``; OFF: measured slower, profiles/r05_rank_step.md): every view's
        # backward SWEEP leaves its masked colour gradient in the exchange's [N,3] slot, and the geometry gradients of the
        # rank's views are summed by ONE launch behind the last sweep (stages.project_views_bwd(v_rgbs=...)) -- instead of one
        # factored projection backward per view
        rank_batch = getattr(self, "rank_batch", False) and factored and same_frame and degree >= 1 and len(mine) <= 8
        if factored:  # slot (r, j) = the j-th view of rank r; ranks with fewer views leave zeros (skipped by the expand)
            slots = -(-n_views // self.world)
            vr_all = torch.empty(self.world, slots, n, 3, device=dev)   # (every slot of a view is written whole by its backward)
            if len(mine) < slots:
                vr_all[self.rank, len(mine):].zero_()
            eye = None
            rows = []
            for r_ in range(self.world):
                theirs = batch[r_::self.world]
                for j in range(slots):
                    if j < len(theirs):
                        rows.append(theirs[j]["pose"].to(dev, torch.float32))
                    else:
                        eye = torch.eye(4, device=dev) if eye is None else eye
                        rows.append(eye)
            poses_all = torch.stack(rows).view(self.world, slots, 4, 4)   # one kernel
        # The views of the batch are independent until their gradients meet in the bucket, and a view's front end
        # (projection + binning: HBM- and latency-bound) uses what its sweeps (compositor forward, loss, compositor backward:
        # VALU-bound) leave idle: whole views alternate on two streams, nobody waits for a view's control words until all
        # are queued, a step whose speculation failed is rendered again with exact sizes (nothing was applied yet) -- all
        # of it the product's one orchestrator, pipeline.ViewPipeline.  No autograd here: the stages are called directly.
        main = torch.cuda.current_stream()
        if self._pipe is None or bool(self._pipe.streams) != self.overlap_views:
            from .pipeline import ViewPipeline
            # (the step speculates on what the trainer's own backend remembers about its frames: one memory per trainer)
            self._pipe = ViewPipeline(dev, 2 if self.overlap_views else 1, speculation=getattr(self.backend, "speculation", None))
        cull = S.DEFAULT_CULL_MODE
        Kc, bgc = K[0].contiguous(), bg[0].contiguous()
        scale = self.loss_scale if self.amp else 1.0   # loss scale: the backward passes see v_render times it
        # the views' loss sums (l1, ssim map, squared error), one row per local view: the step's metrics come out of ONE
        # kernel behind the optimizer (gsa_step_report) instead of a few dozen tiny torch launches
        view_sums = torch.zeros(max(len(mine), 1), 3, dtype=torch.float64, device=dev)
        counts = [3 * view["width"] * view["height"] for view in mine]
        params_ready = torch.cuda.Event()              # (behind everything the views read that this stream prepared)
        params_ready.record(main)

        shared = {}

        def prepare():
            poses = torch.stack([v["pose"].to(dev, torch.float32) for v in mine]).contiguous()
            shared["views"] = S.project_views(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], degree, poses, Kc,
                                              mine[0]["width"], mine[0]["height"], cull, return_aux=True)
            from .rasterization import projection_buffers
            return projection_buffers(shared["views"])

        def project(j):
            """view j's projection: its slice of the step's shared launch, or a launch of its own (split: the records part on
            the projection side stream) -> (pose, splat, radii, tiles, keys, rects, aux, key range, boxes, records event)"""
            view = mine[j]
            W, H = view["width"], view["height"]
            bt = S.bin_tile_for(W, H, cull)   # 32 x 32 bins where the fused binning applies
            if "views" in shared:
                pv = shared["views"][j]
                return (pv.viewmat, pv.splat, pv.radii, pv.tiles_touched, pv.depth_keys, pv.tile_rects, pv.sh_aux, pv.key_range,
                        pv.qbox, pv.records_ready, bt)
            pose = view["pose"].to(torch.float32).contiguous()
            kr = S.key_range_buffer(leaves[0].shape[0], dev)   # the view's depth-key range: the depth ordering plans its buckets from it
            qb = S.qbox_buffer(leaves[0].shape[0], dev)        # the gaussians' quadrant boxes: the forward sweep's masks
            forked = [] if S._SPLIT_PROJECTION else None   # (records part beside the binning chain: stages.project_fwd)
            splat, radii, tiles, keys, rects, aux = S.project_fwd(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4],
                                                                  degree, pose, Kc, W, H, cull, return_rects=True,
                                                                  return_aux=True, bin_tile=bt, key_range=kr, qbox=qb,
                                                                  records_ready=forked)
            return pose, splat, radii, tiles, keys, rects, aux, kr, qb, (forked[0] if forked else None), bt

        def forward(j, deferred):
            """front end, compositor forward and the loss of view j on the current stream"""
            from .pipeline import ViewState
            view = mine[j]
            W, H = view["width"], view["height"]
            pose, splat, radii, tiles, keys, rects, aux, kr, qb, ready, bt = project(j)
            spec = self._pipe.speculation
            # the steady state: the whole chain behind the projection in ONE C-ABI call (stages.ViewChain: gsa_view_fwd)
            chain = S.ViewChain.forward(splat, keys, rects, kr, qb, W, H, 3, cull, bt, bgc, True, speculation=spec,
                                        records_ready=ready) if deferred else None
            if chain is not None:
                _, v = S.view_loss(chain.render, view["image"], c.lambda_dssim, grad_scale=scale / n_views, sums_out=view_sums[j])
                return ViewState((view, pose, splat, radii, chain, aux, v), lambda: chain.bins.finalize()[1])
            # first call of a frame size, repair, what the fused call does not cover: stage by stage
            bins = S.bin_tiles(splat, radii, tiles, keys, W, H, cull, tile_rects=rects, exact=False, bin_tile=bt,
                               key_range=kr, qbox=qb, speculation=spec)
            if not deferred and not bins.finalize()[1]:   # (a remembered capacity / key range did not hold: exact sizes)
                bins = S.bin_tiles(splat, radii, tiles, keys, W, H, cull, tile_rects=rects, exact=True, bin_tile=bt,
                                   key_range=kr, qbox=qb, speculation=spec)
                bins.finalize()
            order = S.tile_schedule(bins, W, H, bt)
            if ready is not None:   # the splat records (written on the projection side stream) are read from here on
                torch.cuda.current_stream().wait_event(ready)
            render, _, final_T, last_id, lists = S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bgc, W,
                                                                 H, 3, order, want_lists=True, bin_tile=bt,
                                                                 gid_boxes=bins.boxed)
            if not deferred:
                view_sums[j].zero_()               # (a step that is rendered again counts every view once)
            _, v = S.view_loss(render, view["image"], c.lambda_dssim, grad_scale=scale / n_views, sums_out=view_sums[j])
            return ViewState((view, pose, splat, radii, (bins, order, final_T, last_id, lists), aux, v),
                             (lambda: bins.finalize()[1]) if deferred else (lambda: True))

        def backward(j, state, before_bucket):
            view, pose, splat, radii, fwd, aux, v = state.payload
            W, H = view["width"], view["height"]
            if isinstance(fwd, S.ViewChain):   # reverse sweep (+ this view's projection backward) in ONE call: gsa_view_bwd
                if batch_bwd or rank_batch:
                    fwd.backward(v, sweep_rgb=vr_all[self.rank, j] if rank_batch else None)
                    shared.setdefault("v_splat", {})[j] = fwd.view("v_splat")
                else:
                    fwd.backward(v, project={"means": leaves[0], "quats": leaves[1], "scales": leaves[2], "opacities": leaves[3],
                                             "colors": leaves[4], "sh_degree": degree, "viewmat": pose, "Kmat": Kc, "radii": radii,
                                             "sh_aux": aux, "out": grad_out, "accumulate": j > 0,
                                             "v_rgb_out": vr_all[self.rank, j] if factored else None},
                                 wait_bucket=before_bucket)
                return
            bins, order, final_T, last_id, lists = fwd
            v_splat = S.composite_bwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bgc, final_T, last_id, v, None,
                                      W, H, 3, order, quad_lists=lists, bin_tile=bins.bin_tile, gid_boxes=bins.boxed,
                                      v_rgb_out=vr_all[self.rank, j] if rank_batch else None, fold=False)
            if batch_bwd or rank_batch:   # (finished for all views at once, below)
                shared.setdefault("v_splat", {})[j] = v_splat
                return
            if before_bucket is not None:   # (the bucket is summed into in view order)
                torch.cuda.current_stream().wait_event(before_bucket)
            S.project_bwd(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], degree, pose, Kc, W, H, splat, radii,
                          v_splat, out=grad_out, accumulate=j > 0,
                          v_rgb_out=vr_all[self.rank, j] if factored else None, sh_aux=aux)

        def finish():
            ok = S.project_views_bwd(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], degree, shared["views"],
                                     [shared["v_splat"][j] for j in range(len(mine))], mine[0]["width"], mine[0]["height"],
                                     grad_out, v_rgbs=[vr_all[self.rank, j] for j in range(len(mine))] if rank_batch else None)
            assert ok, "project_views_bwd refused a configuration batch_bwd admitted"
            shared.pop("v_splat")

        self._pipe.run(len(mine), forward, backward, inputs_ready=params_ready, defer=not exact,
                       prepare=prepare if same_frame else None, finish=finish if (batch_bwd or rank_batch) else None)
        first = not mine
        if first:                                          # this rank rendered nothing
            m.flat_grad.zero_()
            for g in (g_quats, g_scales, g_opac):
                g.zero_()
        # chain rule of the activations + the scale regulariser (value everywhere, gradient on rank 0), one kernel,
        # written straight into the raw-parameter gradient views of flat_grad
        reg_partials, reg_factor = S.activate_bwd(m.scaling, m.rotation, m.opacity, g_scales, g_quats, g_opac, m.scaling.grad,
                                                  m.rotation.grad, m.opacity.grad, c.scale_reg_max_ratio,
                                                  c.scale_reg_weight * scale, add_reg_grad=self.rank == 0,
                                                  return_partials=True)
        if factored:
            # the SH-coefficient gradient of the whole batch, rebuilt from the 3 floats per gaussian and view the
            # backward passes left (summed in batch order), straight into the gradient bucket's coefficient rows.
            # world > 1: those 3 N floats per view are all-gathered and only the 11 N geometry floats are summed (the
            # bucket's tail); the rebuild runs beside the all-reduce.
            summed = None
            if self.world > 1:
                gathered = self._all_gather_slabs(vr_all)
                summed = self._all_reduce(m.flat_grad[48 * n:], async_op=True)
                gathered.wait()
            S.sh_grad_expand(leaves[0].detach(), degree, poses_all.view(-1, 4, 4), K, vr_all.view(-1, n, 3), m.sh_grad)
            if summed is not None:
                summed.wait()
        else:
            self._all_reduce(m.flat_grad)  # the one exchange step: 59*N floats, SUM (the views wrote the rows in place)
        found_inf = None
        if self.amp:   # GradScaler.unscale_ + the skip decision, on the exchanged bucket (identical on every rank)
            found_inf = torch.zeros(1, dtype=torch.int32, device=dev)
            S.grad_unscale(m.flat_grad, 1.0 / scale, found_inf)
        self.optimizer.step(skip_if_nonzero=found_inf)

        # l1, ssim, psnr sums over the local views; their loss share + the regulariser (the same on every rank: counted on
        # rank 0); the regulariser -- one kernel
        report = S.step_report(view_sums if mine else None, counts, c.lambda_dssim, n_views, reg_partials, reg_factor / scale,
                               self.rank == 0, dev)
        self._all_reduce(report)
        if found_inf is not None:
            report = torch.cat([report, found_inf.double()])
        self._pipe.mark_bucket_used()
        if self.async_metrics and found_inf is None:   # (mixed precision: the loss-scale update wants THIS step's flag)
            # no read-back in the step: the report goes to pinned memory behind the step's kernels, and what is returned is
            # the report the PREVIOUS step left there (its copy finished long ago: the host is about one step ahead)
            host = torch.empty(report.shape, dtype=report.dtype, pin_memory=True)
            host.copy_(report, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            prev, self._pending_report = self._pending_report, (host, ev)
            host, ev = prev if prev is not None else self._pending_report
            ev.synchronize()
            r = host.tolist()
        else:
            r = report.tolist()                                    # the step's one host read-back
        if found_inf is not None:                                  # GradScaler.update()
            if r[5] != 0.0:
                self.optimizer.step_count -= 1                     # the update was skipped on the device
                self.loss_scale, self._good_steps = self.loss_scale * 0.5, 0
                self.skipped_steps += 1
            else:
                self._good_steps += 1
                if self._good_steps >= 2000:
                    self.loss_scale, self._good_steps = self.loss_scale * 2.0, 0
        return {"loss": r[3], "l1": r[0] / n_views, "ssim": r[1] / n_views, "psnr": r[2] / n_views, "scale_reg": r[4]}
