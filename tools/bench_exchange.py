"""Device-side cost of the factored gradient exchange at the bench workload (C3: 1M gaussians, 1080p), one GPU:
project_bwd in plain vs factored form, and gsa_sh_grad_expand over V views.  (The collectives themselves need the
8-GPU node; this measures what the factored form adds and saves on each rank.)
    python tools/bench_exchange.py [--views 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--n", type=int, default=1_000_000)
    args = ap.parse_args()
    import gsplat_amd
    from gsplat_amd import scene, stages
    n, W, H, V = args.n, 1920, 1080, args.views
    dev = torch.device("cuda")
    sc = {k: v.to(dev) for k, v in scene.make_scene(n, seed=0, sh_degree=3, device="cpu").items()}
    K = scene.make_intrinsics(W, H).to(dev)[None]
    vms = torch.stack([scene.make_camera(r, V) for r in range(V)]).to(dev)
    v_render = scene.make_v_render(W, H, seed=2).to(dev)[None]
    bucket = [torch.zeros(n, 3, device=dev), torch.zeros(n, 4, device=dev), torch.zeros(n, 3, device=dev),
              torch.zeros(n, device=dev), torch.zeros(n, 16, 3, device=dev)]
    vr_all = torch.zeros(V, n, 3, device=dev)
    leaves = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"]]
    leaves = [t.clone().requires_grad_(True) for t in leaves]

    def run(factored):
        for v in range(V):
            r, _, _ = gsplat_amd.rasterization(*leaves, vms[v:v + 1], K, W, H, sh_degree=3, grad_out=bucket,
                                               sh_grad_factored=vr_all[v] if factored else None)
            r.backward(v_render)

    for factored in (False, True):
        run(factored)
        stages.TIMER.reset(enabled=True)
        for _ in range(5):
            run(factored)
        torch.cuda.synchronize()
        s = stages.TIMER.summary()
        stages.TIMER.reset(enabled=False)
        c, t = s["gsa_project_bwd"]
        print(f"project_bwd {'factored' if factored else 'plain   '}: {t / c * 1e3:7.1f} us per view")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for nv in sorted({1, 2, 4, V}):
        stages.sh_grad_expand(leaves[0].detach(), 3, vms[:nv], K, vr_all[:nv].contiguous(), bucket[4])
        ev[0].record()
        for _ in range(20):
            stages.sh_grad_expand(leaves[0].detach(), 3, vms[:nv], K, vr_all[:nv].contiguous(), bucket[4])
        ev[1].record()
        torch.cuda.synchronize()
        print(f"sh_grad_expand over {nv} views: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
