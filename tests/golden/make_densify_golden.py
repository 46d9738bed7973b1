"""Capture golden vectors of the reference's densify / clone / split / prune step with Adam-state splicing
(/root/reference/src/gaussians/gaussian_model.py:9-50,110-366).

RUNS IN THE AUTHORING CONTAINER ONLY: imports the reference's ``GaussianModel`` from /root/reference (which never
travels) and writes ``tests/golden/densify_golden.npz`` -- inputs and expected outputs, data only.  The split noise
(``torch.randn_like``, gaussian_model.py:302) is recorded so that the counterpart can be driven with the same draws.

    python tests/golden/make_densify_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from src.gaussians.gaussian_model import GaussianModel  # noqa: E402

FIELDS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
N = 400


def build(seed):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    model = GaussianModel(N, "cpu")
    with torch.no_grad():
        model.xyz.copy_(t(rng.uniform(-1, 1, (N, 3))))
        model.features_dc.copy_(t(rng.uniform(-1, 1, (N, 1, 3))))
        model.features_rest.copy_(t(rng.normal(0, 0.05, (N, 15, 3))))
        model.opacity.copy_(t(rng.uniform(-7.0, 3.0, (N, 1))))                 # logits: some below min_opacity
        base = np.exp(rng.uniform(np.log(0.005), np.log(4.0), (N, 1)))          # clone (<= 0.1), split, too big (> 2)
        model.scaling.copy_(t(np.log(base * rng.uniform(0.5, 1.0, (N, 3)))))
        model.rotation.copy_(t(rng.normal(0, 1, (N, 4))))
    model.xyz_gradient_accum = t(rng.uniform(0, 4e-3, (N, 1))).expand(N, 3).clone() * 5.0
    model.xyz_gradient_count = torch.full((N, 1), 5.0)
    opt = torch.optim.Adam([{"params": [getattr(model, f)], "lr": 1e-3, "name": f} for f in FIELDS], lr=0.0, eps=1e-15)
    for _ in range(2):  # give every parameter non-trivial Adam moments
        for f in FIELDS:
            p = getattr(model, f)
            p.grad = t(rng.normal(0, 1e-2, tuple(p.shape)))
        opt.step()
    return model, opt


def main():
    out = {}
    for name, cap in (("free", None), ("capped", N + 40)):
        model, opt = build(seed=21)
        for f in FIELDS:
            p = getattr(model, f)
            out[f"{name}_in_{f}"] = p.detach().numpy().copy()
            out[f"{name}_in_m_{f}"] = opt.state[p]["exp_avg"].numpy().copy()
            out[f"{name}_in_v_{f}"] = opt.state[p]["exp_avg_sq"].numpy().copy()
        out[f"{name}_in_accum"] = model.xyz_gradient_accum.numpy().copy()
        out[f"{name}_in_count"] = model.xyz_gradient_count.numpy().copy()
        noise = []
        real = torch.randn_like

        def recording(x, *a, **k):
            r = real(x, *a, **k)
            noise.append(r.clone())
            return r

        torch.randn_like = recording
        torch.manual_seed(5)
        args = dict(grads_threshold=2e-3, min_opacity=0.005, extent=1.0, max_screen_size=20.0, optimizer=opt,
                    clone_extent_ratio=0.1, prune_extent_ratio=2.0, max_gaussians=cap)
        stats = model.densify_and_prune(**args)
        torch.randn_like = real
        out[f"{name}_args"] = np.asarray([args["grads_threshold"], args["min_opacity"], args["extent"],
                                          args["clone_extent_ratio"], args["prune_extent_ratio"],
                                          -1 if cap is None else cap], dtype=np.float64)
        out[f"{name}_noise"] = (noise[0] if noise else torch.zeros(0, 3)).numpy()
        out[f"{name}_stats"] = np.asarray([stats[k] for k in ("cloned", "split", "pruned", "pruned_low_op",
                                                              "pruned_too_big", "n_before", "n_after")] +
                                          [int(stats["capped"])], dtype=np.int64)
        for f in FIELDS:
            p = getattr(model, f)
            out[f"{name}_out_{f}"] = p.detach().numpy().copy()
            out[f"{name}_out_m_{f}"] = opt.state[p]["exp_avg"].numpy().copy()
            out[f"{name}_out_v_{f}"] = opt.state[p]["exp_avg_sq"].numpy().copy()
        assert float(model.xyz_gradient_accum.abs().max()) == 0 and model.xyz_gradient_accum.shape[0] == stats["n_after"]
        print(name, stats)
    # ---- checkpoint files as the reference writes them (trainer.py:384-395,993): bare state_dict and wrapped
    model, opt = build(seed=33)
    small = GaussianModel(6, "cpu")
    with torch.no_grad():
        for f in FIELDS:
            getattr(small, f).copy_(getattr(model, f)[:6])
    opt_small = torch.optim.Adam([{"params": [getattr(small, f)], "lr": 1e-3, "name": f} for f in FIELDS], lr=0.0, eps=1e-15)
    for f in FIELDS:
        getattr(small, f).grad = torch.ones_like(getattr(small, f)) * 1e-3
    opt_small.step()
    torch.save(small.state_dict(), os.path.join(HERE, "reference_final_model.pth"))
    torch.save({"iteration": 10000, "model_state": small.state_dict(), "optimizer_state": opt_small.state_dict(),
                "n_gaussians": 6}, os.path.join(HERE, "reference_checkpoint_10000.pth"))
    path = os.path.join(HERE, "densify_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
