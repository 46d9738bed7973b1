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
