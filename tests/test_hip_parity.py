"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Tolerance (BASELINE.json north_star): rendered RGB and all gradients within 1e-4
relative fp32 -- measured as max|a-b| / max|b| per tensor.  Integer outputs (radii, tile counts, sorted
ids, offsets) are compared exactly, up to the handful of gaussians whose ceil() lands on the other side
of an fp32 rounding."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
NAMES = ["means3D", "rotations", "scales", "opacities", "shs"]


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _run(fn, dev, sc, vm, K, W, H, deg, bg, v_r, v_a=None, colors_key="shs", **kw):
    leaves = [sc[k].to(dev).clone().requires_grad_(True) for k in NAMES]
    colors = leaves[4] if deg is not None else leaves[4][:, 0, :] * 0.28 + 0.5
    r, a, meta = fn(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], colors, vm[None].to(dev), K[None].to(dev),
                    W, H, sh_degree=deg, backgrounds=None if bg is None else bg[None].to(dev), **kw)
    loss = (r[0] * v_r.to(dev)).sum()
    if v_a is not None:
        loss = loss + (a[0, ..., 0] * v_a.to(dev)).sum()
    loss.backward()
    return r, a, meta, [t.grad for t in leaves]


def _compare(sc, vm, K, W, H, deg, bg, seed=2, use_v_alpha=True, tol=REL_TOL, **kw):
    import gsplat_amd
    from oracle import rasterization as oracle_rasterization

    rng = np.random.default_rng(seed)
    v_r = torch.from_numpy(rng.uniform(-1, 1, (H, W, 3)).astype(np.float32))
    v_a = torch.from_numpy(rng.uniform(-1, 1, (H, W)).astype(np.float32)) if use_v_alpha else None
    r1, a1, m1, g1 = _run(gsplat_amd.rasterization, "cuda", sc, vm, K, W, H, deg, bg, v_r, v_a, **kw)
    r0, a0, m0, g0 = _run(oracle_rasterization, "cpu", sc, vm, K, W, H, deg, bg, v_r, v_a)
    errs = {"render": _rel(r1, r0), "alpha": _rel(a1, a0)}
    for n, x, y in zip(NAMES, g1, g0):
        errs["grad_" + n] = _rel(x, y)
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (bad, errs)
    return m1, m0, errs


@pytest.fixture(scope="module")
def gsa():
    import gsplat_amd
    return gsplat_amd


def _scene(n, seed=0, deg=3, s_min=0.01, s_max=0.1):
    from gsplat_amd import scene
    return scene.make_scene(n, seed=seed, sh_degree=deg, s_min=s_min, s_max=s_max)


# ---------------------------------------------------------------------------------------------------
# stage-level checks
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,begin,end", [(1, 0, 32), (255, 0, 8), (4096, 0, 32), (100_037, 0, 32), (100_037, 0, 13),
                                         (300_001, 3, 19)])
def test_sort_pairs_is_a_stable_radix_sort(gsa, n, begin, end):
    from gsplat_amd.rasterization import _sort_pairs
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64)
    keys = (keys * 2 + torch.randint(0, 2, (n,), generator=g)).to(torch.int64)  # use all 32 bits
    if end - begin < 20:
        keys = keys & (((1 << (end - begin)) - 1) << begin) | (keys & ((1 << begin) - 1))
    k32 = (keys & 0xFFFFFFFF).to(torch.int64)
    dev_keys = torch.where(k32 >= 2 ** 31, k32 - 2 ** 32, k32).to(torch.int32).cuda()
    vals = torch.arange(n, dtype=torch.int32).flip(0).contiguous().cuda()
    ks, vs = _sort_pairs(n, dev_keys, vals, begin, end)
    digit = (k32 >> begin) & ((1 << (end - begin)) - 1)
    order = torch.sort(digit, stable=True).indices
    assert torch.equal(ks.cpu().long() & 0xFFFFFFFF, k32[order])
    assert torch.equal(vs.cpu(), vals.cpu()[order])
    # vals_in = NULL means iota
    ks2, vs2 = _sort_pairs(n, dev_keys, None, begin, end)
    assert torch.equal(vs2.cpu().long(), order)


def test_scan_and_offsets(gsa):
    from gsplat_amd._lib import check, lib
    from gsplat_amd.rasterization import _ptr, _stream
    n = 70_001
    g = torch.Generator().manual_seed(1)
    tiles = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32)
    order = torch.randperm(n, generator=g).to(torch.int32)
    d_t, d_o = tiles.cuda(), order.cuda()
    offs = torch.empty(n, dtype=torch.int32, device="cuda")
    total = torch.empty(1, dtype=torch.int32, device="cuda")
    wsb = lib.gsa_sort_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.gsa_scan_tiles(n, _ptr(d_t), _ptr(d_o), _ptr(offs), _ptr(total), _ptr(ws), wsb, _stream()), "scan")
    ref = torch.cumsum(tiles[order.long()].long(), 0)
    assert int(total.item()) == int(ref[-1])
    assert torch.equal(offs.cpu().long(), ref - tiles[order.long()].long())
    # tile offsets incl. empty tiles at both ends and sentinel records (tile id == n_tiles)
    n_tiles = 50
    st = torch.sort(torch.cat([torch.randint(3, 45, (5000,), generator=g), torch.full((7,), n_tiles)])).values
    d_st = st.to(torch.int32).cuda()
    to = torch.empty(n_tiles + 1, dtype=torch.int32, device="cuda")
    check(lib.gsa_tile_offsets(st.numel(), _ptr(d_st), n_tiles, _ptr(to), _stream()), "offsets")
    assert torch.equal(to.cpu().long(), torch.searchsorted(st, torch.arange(n_tiles + 1)))
    check(lib.gsa_tile_offsets(0, None, n_tiles, _ptr(to), _stream()), "offsets-empty")
    assert int(to.abs().sum()) == 0


@pytest.mark.parametrize("deg", [None, 0, 1, 2, 3])
def test_projection_and_binning_match_oracle(gsa, deg):
    """Camera inside the cloud: exercises near-plane, off-screen and the x/z clamp branches."""
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization
    W, H, N = 200, 120, 6000
    sc = _scene(N, seed=4, s_min=0.005, s_max=0.2)
    vm, K = scene.make_camera(1, 7, radius=0.8), scene.make_intrinsics(W, H)
    colors = sc["shs"] if deg is not None else sc["shs"][:, 0, :] * 0.28 + 0.5
    args = (sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], colors)
    with torch.no_grad():
        r1, a1, m1 = gsa.rasterization(*[t.cuda() for t in args], vm[None].cuda(), K[None].cuda(), W, H, sh_degree=deg)
        r0, a0, m0 = oracle_rasterization(*args, vm[None], K[None], W, H, sh_degree=deg)
    rad1, rad0 = m1["radii"].cpu(), m0["radii"]
    same = (rad1 == rad0).all(-1)[0]
    assert int((~same).sum()) <= 3, "radii differ beyond fp32 ceil() noise"
    assert 0.2 < float((rad0 > 0).all(-1).float().mean()) < 0.9  # a real mix of culled and visible
    for key in ("means2d", "conics", "depths"):
        x, y = m1[key].cpu()[0][same], m0[key][0][same]
        assert _rel(x, y) < 1e-5, key
    if bool(same.all()):
        assert torch.equal(m1["tiles_per_gauss"].cpu(), m0["tiles_per_gauss"])
        assert torch.equal(m1["flatten_ids"].cpu(), m0["flatten_ids"])
        assert torch.equal(m1["isect_offsets"].cpu(), m0["isect_offsets"])
        assert torch.equal(m1["isect_ids"].cpu(), m0["isect_ids"])
    assert _rel(r1, r0) < REL_TOL and _rel(a1, a0) < REL_TOL


# ---------------------------------------------------------------------------------------------------
# end-to-end forward + backward
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("deg", [None, 0, 1, 2, 3])
@pytest.mark.parametrize("cull_mode", [0, 1])
def test_forward_backward_small(gsa, deg, cull_mode):
    from gsplat_amd import scene
    W, H = 150, 90  # ragged right/bottom tiles
    sc = _scene(2500, seed=1)
    _compare(sc, scene.make_camera(2, 9), scene.make_intrinsics(W, H), W, H, deg,
             bg=torch.tensor([0.2, 0.4, 0.1]), cull_mode=cull_mode)


def test_forward_backward_dense_opaque_scene(gsa):
    """Big, opaque splats: long per-tile lists, the T <= 1e-4 stop rule and the 0.999 alpha cap are hit."""
    from gsplat_amd import scene
    W, H = 128, 128
    sc = _scene(3000, seed=7, s_min=0.05, s_max=0.6)
    sc["opacities"] = (sc["opacities"] * 0 + 1.0) * torch.linspace(0.6, 1.0, 3000)[:, None]
    m1, m0, errs = _compare(sc, scene.make_camera(0, 1), scene.make_intrinsics(W, H), W, H, 3, bg=None)
    assert m1["flatten_ids"].numel() > 50_000


def test_camera_inside_cloud_gradients(gsa):
    from gsplat_amd import scene
    W, H = 176, 112
    sc = _scene(4000, seed=4, s_min=0.005, s_max=0.2)
    _compare(sc, scene.make_camera(1, 7, radius=0.8), scene.make_intrinsics(W, H), W, H, 3,
             bg=torch.tensor([0.0, 0.0, 0.0]))


def test_cull_modes_give_identical_images(gsa):
    """Exact ellipse culling only removes (tile, splat) pairs whose every pixel fails alpha >= 1/255."""
    from gsplat_amd import scene
    W, H = 320, 200
    sc = _scene(20_000, seed=3, s_min=0.005, s_max=0.08)
    args = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], sc["shs"]]
    args = [t.cuda() for t in args] + [scene.make_camera(1, 5)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()]
    with torch.no_grad():
        r0, a0, m0 = gsa.rasterization(*args, W, H, sh_degree=3, cull_mode=0)
        r1, a1, m1 = gsa.rasterization(*args, W, H, sh_degree=3, cull_mode=1)
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    assert m1["flatten_ids"].numel() < m0["flatten_ids"].numel()


@pytest.mark.parametrize("mode", ["D", "ED", "RGB+D", "RGB+ED"])
def test_depth_render_modes(gsa, mode):
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization
    W, H = 96, 64
    sc = _scene(1500, seed=5)
    vm, K = scene.make_camera(1, 6), scene.make_intrinsics(W, H)
    outs = []
    for fn, dev in ((gsa.rasterization, "cuda"), (oracle_rasterization, "cpu")):
        leaves = [sc[k].to(dev).clone().requires_grad_(True) for k in NAMES]
        r, a, _ = fn(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm[None].to(dev), K[None].to(dev),
                     W, H, sh_degree=2, render_mode=mode, backgrounds=torch.zeros(1, 3, device=dev))
        w = torch.linspace(0.5, 1.5, r.shape[-1], device=dev)
        (r * w).sum().backward()
        outs.append((r, a, [t.grad for t in leaves]))
    (r1, a1, g1), (r0, a0, g0) = outs
    assert r1.shape == r0.shape
    assert _rel(r1, r0) < REL_TOL and _rel(a1, a0) < REL_TOL
    for n, x, y in zip(NAMES, g1, g0):
        assert _rel(x, y) < REL_TOL, n


def test_degenerate_inputs(gsa):
    from gsplat_amd import scene
    W, H = 64, 48
    K, vm = scene.make_intrinsics(W, H)[None].cuda(), scene.make_camera(0, 1)[None].cuda()
    sc = _scene(10, seed=0)
    base = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], sc["shs"]]
    # everything behind the camera -> background only, zero gradients
    leaves = [t.cuda().clone().requires_grad_(True) for t in base]
    with torch.no_grad():
        leaves[0][:, 2] = -10.0
    bg = torch.tensor([[0.3, 0.2, 0.1]], device="cuda")
    r, a, meta = gsa.rasterization(*leaves, vm, K, W, H, sh_degree=3, backgrounds=bg)
    assert meta["flatten_ids"].numel() == 0 and int(meta["radii"].abs().sum()) == 0
    assert torch.allclose(r[0], bg[0].expand(H, W, 3)) and float(a.abs().max()) == 0
    r.sum().backward()
    assert all(float(t.grad.abs().max()) == 0 for t in leaves)
    # a single gaussian
    one = [t[:1].cuda() for t in base]
    r, a, meta = gsa.rasterization(*one, vm, K, W, H, sh_degree=3)
    assert r.shape == (1, H, W, 3) and torch.isfinite(r).all()
    # wrong shapes / devices raise, they do not crash
    with pytest.raises(ValueError):
        gsa.rasterization(one[0], one[1], one[2], one[3][:, None], one[4], vm, K, W, H, sh_degree=3)
    with pytest.raises(RuntimeError):
        gsa.rasterization(*[t.cpu() for t in one], vm.cpu(), K.cpu(), W, H, sh_degree=3)


def test_backend_plugin_surface(gsa):
    """HipBackend.render_with_depth against the oracle backend, through the reference's dict-in/dict-out
    contract (/root/reference/src/backends/gsplat_backend.py:129-210)."""
    from gsplat_amd import scene
    from oracle import OracleBackend
    W, H = 112, 80
    sc = _scene(1200, seed=9)
    K, vm = scene.make_intrinsics(W, H), scene.make_camera(3, 8)
    vp = {"world_view_transform": vm, "image_width": W, "image_height": H}
    ref = OracleBackend(K).render_with_depth(sc, vp, bg_color=torch.zeros(3), render_mode="RGB", device="cpu",
                                            sh_degree=2)
    sc_d = {k: v.cuda() for k, v in sc.items()}
    vp_d = {"world_view_transform": vm.cuda(), "image_width": W, "image_height": H}
    be = gsa.HipBackend(K.cuda())
    out = be.render_with_depth(sc_d, vp_d, bg_color=torch.zeros(3, device="cuda"), render_mode="RGB", device="cuda",
                               sh_degree=2)
    assert set(out) == set(ref)
    assert out["render"].shape == (H, W, 3) and out["alpha"].shape == (H, W, 1)
    assert out["radii"].dtype == torch.int32 and out["visibility_filter"].dtype == torch.bool
    assert _rel(out["render"], ref["render"]) < REL_TOL
    assert torch.equal(out["visibility_filter"].cpu(), ref["visibility_filter"])
    full = be.render_with_depth(sc_d, vp_d)  # defaults: RGB+ED, degree 3, zero background
    assert set(full) >= {"render", "expected_depth", "alpha", "means2d", "radii", "visibility_filter"}
    single = be.render_single(sc_d, vp_d, torch.zeros(3, device="cuda"))
    assert torch.allclose(single["render"], full["render"]) and single["depth"].shape == (H, W)
    batch = be.render_batch(sc_d, vm.cuda()[None].repeat(2, 1, 1), torch.eye(4, device="cuda")[None].repeat(2, 1, 1),
                            W, H, torch.zeros(3, device="cuda"), "cuda")
    assert batch["images"].shape == (2, 3, H, W) and batch["alphas"].shape == (2, 1, H, W)


def test_config_c2_100k_1080p_deg3(gsa):
    """BASELINE config C2: 100k gaussians, 1920x1080, SH degree 3, forward+backward, vs the CPU oracle."""
    from gsplat_amd import scene
    W, H = 1920, 1080
    sc = scene.make_scene(100_000, seed=0, sh_degree=3)
    m1, m0, errs = _compare(sc, scene.make_camera(0, 1), scene.make_intrinsics(W, H), W, H, 3,
                            bg=torch.zeros(3), use_v_alpha=False)
    print("C2 errors:", {k: f"{v:.2e}" for k, v in errs.items()}, "M =", m1["flatten_ids"].numel())
    assert abs(m1["flatten_ids"].numel() - m0["flatten_ids"].numel()) <= 64
