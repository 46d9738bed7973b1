"""CPU dry run of bench.py's exchange (the branch that first executes for real on an 8-GPU node over RCCL): the
``Exchange`` class is driven with a stand-in for ``torch.distributed`` that checks every collective's arguments the way
RCCL would (contiguity, sizes, no aliasing) and applies the collective's arithmetic for a simulated second rank, and
with a torch formulation of ``stages.sh_grad_expand``.  A typo or a shape mistake in the ``nccl`` branch fails here."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Work:
    def __init__(self):
        self.waited = False

    def wait(self):
        self.waited = True


class FakeDist:
    """world_size ranks simulated in one process: this object plays rank `rank`; what the OTHER ranks contribute is
    given up front (``others_rgb[j][r]``, ``others_geom``)."""

    def __init__(self, backend, world, rank, others_rgb, others_geom, others_full):
        self.backend, self.world, self.rank = backend, world, rank
        self.others_rgb, self.others_geom, self.others_full = others_rgb, others_geom, others_full
        self.calls, self.works, self.gathers = [], [], 0

    def get_backend(self):
        return self.backend

    def get_world_size(self):
        return self.world

    def all_gather_into_tensor(self, out, inp, async_op=False):
        assert self.backend == "nccl"
        assert out.is_contiguous() and inp.is_contiguous(), "RCCL needs contiguous buffers"
        assert out.numel() == self.world * inp.numel() and out.dtype == inp.dtype
        self.calls.append("all_gather_into_tensor")
        j = self.gathers
        self.gathers += 1
        chunks = out.view(self.world, -1)
        mine = inp.reshape(-1).clone()
        for r in range(self.world):
            chunks[r].copy_(mine if r == self.rank else self.others_rgb[j][r].reshape(-1))
        w = _Work()
        self.works.append(w)
        return w if async_op else None

    def all_gather(self, outs, inp, async_op=False):
        assert self.backend == "gloo"
        assert len(outs) == self.world and all(o.shape == inp.shape and o.is_contiguous() for o in outs)
        assert all(o.data_ptr() != inp.data_ptr() for o in outs), "input aliases an output"
        self.calls.append("all_gather")
        j = self.gathers
        self.gathers += 1
        for r, o in enumerate(outs):
            o.copy_(inp if r == self.rank else self.others_rgb[j][r])
        w = _Work()
        self.works.append(w)
        return w if async_op else None

    def all_reduce(self, t, async_op=False):
        assert t.is_contiguous()
        self.calls.append(("all_reduce", t.numel()))
        t.add_(self.others_geom if t.numel() == self.others_geom.numel() else self.others_full)
        w = _Work()
        self.works.append(w)
        return w if async_op else None


def _expand_ref(means, deg, viewmats, Kmat, v_rgb_all, v_colors):
    """Stand-in for stages.sh_grad_expand: any fixed linear map of (view, v_rgb) will do for the plumbing check."""
    assert viewmats.shape == (v_rgb_all.shape[0], 4, 4) and v_rgb_all.is_contiguous()
    w = viewmats[:, 0, 3].view(-1, 1, 1, 1)
    k = torch.arange(1, v_colors.shape[1] + 1, dtype=torch.float32).view(1, 1, -1, 1)
    v_colors.copy_((w * k * v_rgb_all[:, :, None, :]).sum(0))


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
@pytest.mark.parametrize("B", [1, 3])
def test_factored_exchange_plumbing(backend, B):
    bench = _bench()
    world, rank, n = 2, 1, 37
    g = torch.Generator().manual_seed(B)
    others_rgb = [[torch.randn(n, 3, generator=g) for _ in range(world)] for _ in range(B)]
    others_geom = torch.randn(11 * n, generator=g)
    fd = FakeDist(backend, world, rank, others_rgb, others_geom, None)
    ex = bench.Exchange(fd, world, rank, n, B, True, "cpu", expand=_expand_ref)
    flat_grad = torch.randn(59 * n, generator=g)
    geom_before = flat_grad[:11 * n].clone()
    mine = [torch.randn(n, 3, generator=g) for _ in range(B)]
    for j in range(B):
        tgt = ex.view_target(j)
        assert tgt.shape == (n, 3) and tgt.is_contiguous()
        tgt.copy_(mine[j])                      # what gsa_project_bwd_factored leaves
        ex.after_view(j)
    viewmats = torch.zeros(B * world, 4, 4)
    viewmats[:, 0, 3] = torch.arange(1, B * world + 1, dtype=torch.float32)
    v_colors = flat_grad[11 * n:].view(n, 16, 3)
    ex.finish(flat_grad, torch.zeros(n, 3), 3, viewmats, torch.eye(3)[None], v_colors)
    assert all(w.waited for w in fd.works) and fd.gathers == B
    assert fd.calls.count(("all_reduce", 11 * n)) == 1
    assert torch.equal(flat_grad[:11 * n], geom_before + others_geom)
    # view j of rank r sits at row j * world + r of what the rebuild reads
    rows = torch.stack([mine[j] if r == rank else others_rgb[j][r] for j in range(B) for r in range(world)])
    want = torch.empty(n, 16, 3)
    _expand_ref(None, 3, viewmats, None, rows.contiguous(), want)
    assert torch.equal(v_colors, want)
    assert ex.bytes_sent == B * n * 3 * 4 * (world - 1) + 2 * 11 * n * 4 * (world - 1) // world and ex.steps == 1


def test_plain_exchange_is_one_all_reduce_of_the_bucket():
    bench = _bench()
    world, n = 4, 11
    full = torch.ones(59 * n)
    fd = FakeDist("nccl", world, 0, None, torch.zeros(1), full)
    ex = bench.Exchange(fd, world, 0, n, 2, False, "cpu")
    assert ex.view_target(0) is None
    ex.after_view(0)
    ex.after_view(1)
    flat_grad = torch.zeros(59 * n)
    ex.finish(flat_grad, None, 3, None, None, None)
    assert fd.calls == [("all_reduce", 59 * n)] and torch.equal(flat_grad, full)


def test_single_process_factored_needs_no_collective():
    bench = _bench()
    n, B = 9, 4
    ex = bench.Exchange(None, 1, 0, n, B, True, "cpu", expand=_expand_ref)
    for j in range(B):
        ex.view_target(j).fill_(float(j + 1))
        ex.after_view(j)
    viewmats = torch.zeros(B, 4, 4)
    viewmats[:, 0, 3] = 1.0
    v_colors = torch.empty(n, 16, 3)
    ex.finish(torch.zeros(59 * n), torch.zeros(n, 3), 3, viewmats, torch.eye(3)[None], v_colors)
    assert torch.allclose(v_colors[:, 0, :], torch.full((n, 3), 10.0)) and ex.bytes_sent == 0


@pytest.mark.parametrize("world,B,ring", [(8, 1, 16), (4, 2, 16), (8, 1, 8), (2, 4, 24), (8, 4, 64), (1, 4, 8), (3, 2, 7)])
def test_step_camera_indexing_deals_the_window_round_robin(world, B, ring):
    """bench.py's camera indexing for the shapes an 8-GPU node runs (BASELINE C4: 8 ranks x 1 view; 4 x 2; the default 8 x 4)
    with a ring LARGER than a step's window: per step the ranks' views are disjoint, together they are the step's window of
    `world x B` consecutive ring cameras, entry j x world + r of the window is rank r's view j (the order the local SH rebuild
    sums in), and the window moves on from step to step."""
    bench = _bench()
    for i in range(5):
        window = bench.step_window(i, B, world, ring)
        assert window == [((i * B * world) + k) % ring for k in range(B * world)]
        per_rank = [bench.step_cameras(i, B, world, r, ring) for r in range(world)]
        assert all(len(c) == B for c in per_rank)
        for j in range(B):
            for r in range(world):
                assert per_rank[r][j] == window[j * world + r]
        if ring >= B * world:
            assert sorted(c for cams in per_rank for c in cams) == sorted(window) and len(set(window)) == B * world
    if ring > B * world:
        assert bench.step_window(0, B, world, ring) != bench.step_window(1, B, world, ring)


def _fake_view(cam: int, n: int):
    """What a view of ring camera `cam` contributes: (11 n geometry floats, [n,3] rgb gradient) -- any function of the camera."""
    g = torch.Generator().manual_seed(1000 + cam)
    return torch.randn(11 * n, generator=g), torch.randn(n, 3, generator=g)


def _exchange_rank(rank, world, B, ring, port, form):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        bench = _bench()
        n = 53
        factored = form == "factored"
        ex = bench.Exchange(dist, world, rank, n, B, factored, "cpu", expand=_expand_ref)
        for i in (0, 1, 3):
            flat_grad = torch.zeros(59 * n)
            for j, cam in enumerate(bench.step_cameras(i, B, world, rank, ring)):
                geom, rgb = _fake_view(cam, n)
                flat_grad[:11 * n] += geom
                if factored:
                    ex.view_target(j).copy_(rgb)
                else:   # (the plain form: every view adds its whole coefficient gradient)
                    one = torch.empty(n, 16, 3)
                    vm = torch.zeros(1, 4, 4)
                    vm[0, 0, 3] = cam + 1.0
                    _expand_ref(None, 3, vm, None, rgb[None].contiguous(), one)
                    flat_grad[11 * n:] += one.reshape(-1)
                ex.after_view(j)
            window = bench.step_window(i, B, world, ring)
            viewmats = torch.zeros(B * world, 4, 4)
            viewmats[:, 0, 3] = torch.tensor([c + 1.0 for c in window])
            ex.finish(flat_grad, torch.zeros(n, 3), 3, viewmats, torch.eye(3)[None], flat_grad[11 * n:].view(n, 16, 3))
            want_geom = sum(_fake_view(c, n)[0] for c in window)
            want_sh = torch.empty(n, 16, 3)
            _expand_ref(None, 3, viewmats, None, torch.stack([_fake_view(c, n)[1] for c in window]).contiguous(), want_sh)
            assert torch.allclose(flat_grad[:11 * n], want_geom, rtol=1e-5, atol=1e-5), (rank, i, "geometry")
            assert torch.allclose(flat_grad[11 * n:].view(n, 16, 3), want_sh, rtol=1e-5, atol=1e-4), (rank, i, "sh")
        assert ex.steps == 3 and not ex.pending
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,form", [(8, 1, "factored"), (4, 2, "factored"), (8, 1, "allreduce")])
def test_exchange_over_gloo_at_the_8_gpu_shapes(world, B, form):
    """The exchange with the world sizes the driver's scaling run uses, over REAL collectives (gloo, CPU tensors; a GPU box
    admits at most 6 processes on its card, so 8 ranks cannot share it): BASELINE C4 as written (8 ranks x 1 view) and 4 x 2,
    ring of 16 cameras so that the window moves: on every rank the exchanged bucket is the sum over the step's 8 cameras --
    geometry all-reduced, rgb gradients all-gathered per view and rebuilt locally in batch order -- for three steps."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_exchange_rank, args=(world, B, 16, port, form), nprocs=world, join=True)
