"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): rendered RGB and all gradients within 1e-4 relative fp32, measured
as max|a-b| / max|b| per tensor (``REL_TOL``).  Two properties of the function itself shape how that is applied:

* It is DISCONTINUOUS at its thresholds (alpha = 1/255 skip, T(1-alpha) = 1e-4 stop, alpha cap 0.999 for the
  gradient).  Where the oracle sees a comparison within a small relative window of a threshold, two correct
  fp32 implementations may decide differently, and one flipped decision moves a pixel by ~4e-3.  Those pixels
  (a bounded, reported fraction) are excluded from the image comparison and their upstream gradient is zeroed
  on BOTH sides; everything else is compared.
* End to end it is ILL-CONDITIONED in fp32: mean2d is an fp32 pixel coordinate (ulp 1.2e-4 at x > 1024) and
  d alpha / alpha = conic * d * (error of mean2d).  The end-to-end tests therefore measure BOTH the HIP path and the
  fp32 oracle against the fp64 oracle.  The bound on the HIP path is the plain 1e-4 on every tensor, outright -- it
  holds everywhere (measured: <= 3e-5 at C2, profiles/r02_parity_c2.json; <= 4e-6 on the small scenes); the fp32
  oracle's own distance from the fp64 one is reported beside it for scale.  ``_compare(relaxed=...)`` can name
  tensors that get max(1e-4, 2 x fp32-oracle error) instead; no test needs it.
  The stage-level tests feed the oracle the same intermediates the kernels consume and hold 1e-4 or tighter.

Integer outputs (radii, tile counts, sorted ids, offsets) are compared exactly, up to the handful of gaussians
whose ceil() lands on the other side of an fp32 rounding."""
import json
import math
import contextlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
NAMES = ["means3D", "rotations", "scales", "opacities", "shs"]
AMB_TOL_STAGE = 1e-4     # identical inputs on both sides: alpha differs by ~1e-6
MAX_AMB_FRACTION = 2e-2


# Stage-level tensors that may take max(1e-4, 2 x the fp32 oracle's own distance from the fp64 oracle) instead of the
# outright 1e-4 -- each with the measured value that put it here.  EMPTY: every stage tensor holds 1e-4 outright
# (measured values: profiles/r04_parity_stage.json, written by the tests when GSPLAT_AMD_EVIDENCE is set).
STAGE_RELAXED = set()


def _evidence(name, payload):
    """With GSPLAT_AMD_EVIDENCE=<dir> in the environment the parity tests leave their measured errors there (one JSON
    file per test case) -- what profiles/r04_parity_*.json are assembled from."""
    d = os.environ.get("GSPLAT_AMD_EVIDENCE")
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)


def _amb_tol_e2e(W, H):
    """Relative window for end-to-end runs: d alpha/alpha <= conic(<=3.3) * d(~3 px) * ulp(mean2d)."""
    return max(1e-4, 10.0 * float(np.spacing(np.float32(max(W, H)))))


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _run(fn, dev, dtype, sc, vm, K, W, H, deg, bg, v_r, v_a, keep=None, **kw):
    """Forward + backward with the upstream gradient masked by ``keep`` ([H,W] 0/1, from the fp64 oracle)."""
    leaves = [sc[k].to(dev, dtype).clone().requires_grad_(True) for k in NAMES]
    colors = leaves[4] if deg is not None else leaves[4][:, 0, :] * 0.28 + 0.5
    r, a, meta = fn(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], colors, vm[None].to(dev, dtype),
                    K[None].to(dev, dtype), W, H, sh_degree=deg,
                    backgrounds=None if bg is None else bg[None].to(dev, dtype), **kw)
    if keep is None:
        keep = (~meta["ambiguous"][0]).double()
    loss = (r[0] * (v_r * keep[..., None]).to(dev, dtype)).sum()
    if v_a is not None:
        loss = loss + (a[0, ..., 0] * (v_a * keep).to(dev, dtype)).sum()
    loss.backward()
    k3 = keep[None, ..., None]
    outs = {"render": r.detach().cpu().double() * k3, "alpha": a.detach().cpu().double() * k3}
    for n, t in zip(NAMES, leaves):
        outs["grad_" + n] = t.grad.detach().cpu().double()
    outs["_render_unmasked"] = r.detach().cpu().double()     # (the masked pixels too: what `_masked_pixels` looks at)
    return outs, meta, keep


def _masked_pixels(hip, o32, o64, keep, flip_bound):
    """What happens at the pixels the comparison masks (threshold-ambiguous in the fp64 oracle).  Nothing is compared
    there tensor by tensor -- one flipped skip / stop decision moves such a pixel by ~4e-3 -- but a correct implementation can
    only differ from the fp64 oracle THROUGH those decisions: every masked pixel must lie within the oracle's own bound
    for that (``meta['flip_bound']``: the weights alpha T of the pixel's near-threshold records, see
    ``oracle.torch_oracle.composite``) plus the continuous tolerance.  Returned for the evidence files: how many pixels
    HIP and the fp32 oracle actually decide differently from the fp64 oracle / from each other (image further apart than
    ten continuous tolerances), and the largest excess over the bound (must be <= 0)."""
    masked = keep[0] == 0 if keep.dim() == 3 else keep == 0
    scale = float(o64.abs().max())
    cont = REL_TOL * scale
    d_h = (hip - o64)[0].abs().amax(-1)
    d_o = (o32 - o64)[0].abs().amax(-1)
    d_ho = (hip - o32)[0].abs().amax(-1)
    excess = (d_h - flip_bound - cont)[masked]
    rep = {"masked_pixels": int(masked.sum()),
           "hip_decides_differently_from_fp64_oracle": int((d_h[masked] > 10 * cont).sum()),
           "fp32_oracle_decides_differently_from_fp64_oracle": int((d_o[masked] > 10 * cont).sum()),
           "hip_decides_differently_from_fp32_oracle": int((d_ho[masked] > 10 * cont).sum()),
           "decisions_differ_outside_the_mask": int((d_ho[~masked] > 10 * cont).sum()),
           "max_masked_pixel_difference": float(d_h[masked].max()) if masked.any() else 0.0,
           "max_flip_bound": float(flip_bound[masked].max()) if masked.any() else 0.0,
           "max_excess_over_flip_bound": float(excess.max()) if masked.any() else 0.0}
    assert rep["decisions_differ_outside_the_mask"] == 0, rep
    assert rep["max_excess_over_flip_bound"] <= 0.0, f"a masked pixel differs by more than its flipped decisions allow: {rep}"
    return rep


def _compare(sc, vm, K, W, H, deg, bg, seed=2, use_v_alpha=True, checkpoint=False, relaxed=(), **kw):
    """HIP and the fp32 oracle, both against the fp64 oracle; see the module docstring.  Every tensor must be within
    REL_TOL of the fp64 oracle OUTRIGHT, except those named in ``relaxed``: for them (ill-conditioned in fp32 on that
    scene: the fp32 oracle itself is further than 1e-4 from the fp64 one) the bound is max(1e-4, 2 x fp32-oracle error)."""
    import gsplat_amd
    from oracle import rasterization as oracle_rasterization

    rng = np.random.default_rng(seed)
    v_r = torch.from_numpy(rng.uniform(-1, 1, (H, W, 3)))
    v_a = torch.from_numpy(rng.uniform(-1, 1, (H, W))) if use_v_alpha else None
    o64, m64, keep = _run(oracle_rasterization, "cpu", torch.float64, sc, vm, K, W, H, deg, bg, v_r, v_a,
                          ambiguity_tol=_amb_tol_e2e(W, H), checkpoint=checkpoint)
    amb_frac = 1.0 - float(keep.mean())
    assert amb_frac <= MAX_AMB_FRACTION, f"too many threshold-ambiguous pixels: {amb_frac:.2e}"
    o32, m32, _ = _run(oracle_rasterization, "cpu", torch.float32, sc, vm, K, W, H, deg, bg, v_r, v_a, keep=keep,
                       checkpoint=checkpoint)
    hip, m1, _ = _run(gsplat_amd.rasterization, "cuda", torch.float32, sc, vm, K, W, H, deg, bg, v_r, v_a,
                      keep=keep, **kw)
    report, bad = {"ambiguous_fraction": amb_frac}, {}
    report["masked_pixels"] = _masked_pixels(hip.pop("_render_unmasked"), o32.pop("_render_unmasked"),
                                             o64.pop("_render_unmasked"), keep, m64["flip_bound"][0].double())
    for k in o64:
        e32, eh = _rel(o32[k], o64[k]), _rel(hip[k], o64[k])
        report[k] = (eh, e32)
        bound = max(REL_TOL, 2.0 * e32) if k in relaxed else REL_TOL
        if not eh <= bound:
            bad[k] = (eh, e32)
    print("parity report:", report)
    assert not bad, f"HIP error vs fp64 oracle exceeds its bound (1e-4; `relaxed` tensors max(1e-4, 2 x fp32-oracle error)): {bad}; all: {report}"
    return m1, m32, report


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
    from gsplat_amd.stages import sort_pairs as _sort_pairs
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
    from gsplat_amd.stages import _ptr, _stream
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
        r1, a1, m1 = gsa.rasterization(*[t.cuda() for t in args], vm[None].cuda(), K[None].cuda(), W, H, sh_degree=deg,
                                       cull_mode=0)  # the published tile rectangle: binning comparable 1:1
        r0, a0, m0 = oracle_rasterization(*args, vm[None], K[None], W, H, sh_degree=deg,
                                          ambiguity_tol=_amb_tol_e2e(W, H))
    rad1, rad0 = m1["radii"].cpu(), m0["radii"]
    same = (rad1 == rad0).all(-1)[0]
    assert int((~same).sum()) <= 3, "radii differ beyond fp32 ceil() noise"
    assert 0.2 < float((rad0 > 0).all(-1).float().mean()) < 0.9  # a real mix of culled and visible
    for key in ("means2d", "conics", "depths"):
        x, y = m1[key].cpu()[0][same], m0[key][0][same]
        assert _rel(x, y) < 5e-5, key
    # The integer outputs are compared on the gaussians whose radii agree: the (<= 3) gaussians whose ceil() landed on the
    # other side of an fp32 rounding are dropped from BOTH sides' record lists -- the block always runs (r03 skipped it
    # whenever one radius differed).
    ok_g = same
    assert torch.equal(m1["tiles_per_gauss"].cpu()[0][ok_g], m0["tiles_per_gauss"][0][ok_g])
    f1, f0 = m1["flatten_ids"].cpu(), m0["flatten_ids"]
    ids1, ids0 = m1["isect_ids"].cpu(), m0["isect_ids"]
    k1, k0 = ok_g[f1 % N], ok_g[f0 % N]
    assert int((~k1).sum()) <= 3 * 64 and int((~k0).sum()) <= 3 * 64   # (a handful of records)
    assert torch.equal(f1[k1], f0[k0]), "sorted gaussian ids (tile, depth, index order)"
    # 64-bit keys: the tile half must agree exactly -- with it the per-tile record counts, i.e. isect_offsets; the depth
    # half holds fp32 bits of camera-space z, whose last bits depend on how t + R.m is fused (cancellation when the camera
    # sits inside the cloud) -- compare as floats
    assert torch.equal((ids1 >> 32)[k1], (ids0 >> 32)[k0])
    if bool(same.all()):
        assert torch.equal(m1["isect_offsets"].cpu(), m0["isect_offsets"])
    n_t = m0["isect_offsets"].numel()
    assert torch.equal(torch.bincount((ids1 >> 32)[k1], minlength=n_t), torch.bincount((ids0 >> 32)[k0], minlength=n_t))
    z1 = (ids1 & 0xFFFFFFFF).to(torch.int32).view(torch.float32)[k1]
    z0 = (ids0 & 0xFFFFFFFF).to(torch.int32).view(torch.float32)[k0]
    assert float((z1 - z0).abs().max()) <= 5e-5 * float(z0.abs().max())
    k3 = (~m0["ambiguous"]).float()[..., None]
    assert float(1 - k3.mean()) <= MAX_AMB_FRACTION
    assert _rel(r1.cpu() * k3, r0 * k3) < REL_TOL and _rel(a1.cpu() * k3, a0 * k3) < REL_TOL


def _hip_front_end(sc, vm, K, W, H, deg, cull_mode=0):
    from gsplat_amd import stages as S
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    args = (d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"])
    splat, radii, tiles, keys = S.project_fwd(*args, deg, vm.cuda(), K.cuda(), W, H, cull_mode)
    sorted_tile, sorted_gid, tile_offsets = S.bin_tiles(splat, radii, tiles, keys, W, H, cull_mode)
    return args, splat, radii, sorted_gid, tile_offsets


@pytest.mark.parametrize("W,H,N,cull_mode,with_bg", [(640, 360, 20_000, 0, True), (1920, 1080, 20_000, 1, False),
                                                      (250, 130, 3000, 0, True)])
def test_stage_composite_fwd_bwd(gsa, W, H, N, cull_mode, with_bg):
    """Compositor alone: HIP vs oracle.composite on the SAME splat records, sorted
    ids and tile offsets; fp64-calibrated (the record gradients are cancelling sums over thousands of pixels)."""
    from gsplat_amd import scene, stages as S
    from oracle import torch_oracle as orc
    big = N == 3000
    sc = _scene(N, seed=11, s_min=0.05 if big else 0.004, s_max=0.5 if big else 0.05)
    vm, K = scene.make_camera(1, 6), scene.make_intrinsics(W, H)
    _, splat, radii, sorted_gid, tile_offsets = _hip_front_end(sc, vm, K, W, H, 3, cull_mode)
    bg = torch.tensor([0.3, 0.6, 0.1]) if with_bg else None
    rng = np.random.default_rng(5)
    v_r = torch.from_numpy(rng.uniform(-1, 1, (H, W, 3)).astype(np.float32))
    v_a = torch.from_numpy(rng.uniform(-1, 1, (H, W)).astype(np.float32))

    sp = splat.cpu()

    def oracle(dtype, keep=None):
        leaves = [sp[:, 0:2].to(dtype).requires_grad_(True), sp[:, 2:5].to(dtype).requires_grad_(True),
                  sp[:, 5].to(dtype).requires_grad_(True), sp[:, 6:9].to(dtype).requires_grad_(True)]
        img, al, amb, _ = orc.composite(*leaves, None if bg is None else bg.to(dtype), W, H, 16,
                                     sorted_gid.cpu().long(), tile_offsets.cpu().long(),
                                     ambiguity_tol=AMB_TOL_STAGE)
        if keep is None:
            keep = (~amb).float()
        kd = keep.to(dtype)
        ((img * v_r.to(dtype) * kd[..., None]).sum() + (al[..., 0] * v_a.to(dtype) * kd).sum()).backward()
        return img.detach(), al.detach(), [t.grad for t in leaves], keep

    img0, al0, g0, keep = oracle(torch.float64)
    assert 1 - float(keep.mean()) <= MAX_AMB_FRACTION
    img32, al32, g32, _ = oracle(torch.float32, keep)

    bg_d = None if bg is None else bg.cuda()
    img1, al1, fin_T, last = S.composite_fwd(splat, sorted_gid, tile_offsets, bg_d, W, H, 3)
    v_splat = S.composite_bwd(splat, sorted_gid, tile_offsets, bg_d, fin_T, last, (v_r * keep[..., None]).cuda(),
                              (v_a * keep).cuda().contiguous(), W, H, 3)
    v_splat = S.moments_to_record_grads(splat.double(), v_splat.double()).float().cpu()  # raw moments -> gradients
    k3 = keep[..., None]
    checks = [("render", img1.cpu() * k3, img32 * k3, img0 * k3), ("alpha", al1.cpu() * k3, al32 * k3, al0 * k3),
              ("v_means2d", v_splat[:, 0:2], g32[0], g0[0]), ("v_conics", v_splat[:, 2:5], g32[1], g0[1]),
              ("v_opacity", v_splat[:, 5], g32[2], g0[2]), ("v_rgb", v_splat[:, 6:9], g32[3], g0[3])]
    report = {}
    for name, got, o32, o64 in checks:
        eh, e32 = _rel(got, o64), _rel(o32, o64)
        report[name] = (eh, e32)
        # 1e-4 outright (r03 allowed max(1e-4, 2 x the fp32 oracle's own error)); STAGE_RELAXED names what may not hold it
        bound = max(REL_TOL, 2 * e32) if (name, W, N) in STAGE_RELAXED else REL_TOL
        assert eh <= bound, (name, eh, e32, report)
    _evidence(f"stage_composite_{W}x{H}_{N}", {k: {"hip_vs_fp64": v[0], "fp32_oracle_vs_fp64": v[1]} for k, v in report.items()})
    assert float(v_splat[:, 9:].abs().max()) == 0
    if big:
        assert int((last >= 0).sum()) > 0.9 * W * H


@pytest.mark.parametrize("aux", [False, True])
@pytest.mark.parametrize("deg", [None, 0, 1, 2, 3])
def test_stage_project_bwd(gsa, deg, aux):
    """Projection + SH backward alone (row a3.6): HIP vs autograd through oracle.project / sh_colors for the
    same record gradient, fp64-calibrated (the conic VJP shares the determinant cancellation of the forward).
    aux: with the forward's 48-B digest of the coefficient rows (the rows themselves are then not read; handed NaN
    rows here to prove it)."""
    from gsplat_amd import scene, stages as S
    from oracle import torch_oracle as orc
    W, H, N = 320, 200, 8000
    sc = _scene(N, seed=13, s_min=0.005, s_max=0.2)
    vm, K = scene.make_camera(2, 7, radius=1.2), scene.make_intrinsics(W, H)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    colors_d = d["shs"] if deg is not None else (d["shs"][:, 0, :] * 0.28 + 0.5).contiguous()
    hdeg = -1 if deg is None else deg
    args = (d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), colors_d)
    was = S._SH_AUX
    S._SH_AUX = True   # (GSPLAT_AMD_SH_AUX=0 in the environment must not switch this test's subject off)
    try:
        splat, radii, _, _, sh_aux = S.project_fwd(*args, hdeg, vm.cuda(), K.cuda(), W, H, 0, return_aux=True)
    finally:
        S._SH_AUX = was
    assert (sh_aux is None) == (hdeg < 1)
    vis = (radii > 0).all(-1).cpu()
    g = torch.Generator().manual_seed(3)
    v = torch.randn(N, 16, generator=g)
    v[:, 10:] = 0
    v[~vis] = 0
    # project_bwd consumes the sweep's raw moments: hand it the moments whose implied record gradient is v
    mom = S.record_grads_to_moments(splat.double().cpu(), v.double()).float()
    mom[~vis] = 0
    if aux and sh_aux is not None:
        poisoned = args[:4] + (torch.full_like(colors_d, float("nan")),)
        got = S.project_bwd(*poisoned, hdeg, vm.cuda(), K.cuda(), W, H, splat, radii, mom.cuda(), sh_aux=sh_aux)
        v_rgb = torch.empty(N, 3, device="cuda")
        fact = S.project_bwd(*poisoned, hdeg, vm.cuda(), K.cuda(), W, H, splat, radii, mom.cuda(), v_rgb_out=v_rgb,
                             sh_aux=sh_aux)
        v_rgb0 = torch.empty(N, 3, device="cuda")
        fact0 = S.project_bwd(*args, hdeg, vm.cuda(), K.cuda(), W, H, splat, radii, mom.cuda(), v_rgb_out=v_rgb0)
        assert torch.equal(v_rgb, v_rgb0)                              # same clamp flags, same d loss / d rgb
        for a, b, c in zip(fact[:4], fact0[:4], got[:4]):
            assert _rel(a, c) < 1e-6 and _rel(a, b) < 1e-5                # factored == full; digest == rows to rounding
    else:
        got = S.project_bwd(*args, hdeg, vm.cuda(), K.cuda(), W, H, splat, radii, mom.cuda())

    def oracle(dtype):
        leaves = [sc[k].to(dtype).clone().requires_grad_(True) for k in NAMES]
        opac = leaves[3][:, 0]
        pr = orc.project(leaves[0], leaves[1], leaves[2], opac, vm.to(dtype), K.to(dtype), W, H)
        assert torch.equal(pr["visible"], vis)
        if deg is None:
            col_leaf = (leaves[4][:, 0, :] * 0.28 + 0.5).detach().requires_grad_(True)
            rgb = col_leaf * vis.to(dtype)[:, None]
        else:
            col_leaf = leaves[4]
            rgb = orc.sh_colors(deg, leaves[0], leaves[4], vm.to(dtype), pr["visible"])
        vd = v.to(dtype)
        loss = (vd[:, 0:2] * pr["means2d"]).sum() + (vd[:, 2:5] * pr["conics"]).sum() + (vd[:, 5] * opac).sum() \
            + (vd[:, 6:9] * rgb).sum() + (vd[:, 9] * pr["depths"]).sum()
        loss.backward()
        return [leaves[0].grad, leaves[1].grad, leaves[2].grad, leaves[3].grad[:, 0], col_leaf.grad]

    ref64, ref32 = oracle(torch.float64), oracle(torch.float32)
    report = {}
    for name, x, y64, y32 in zip(["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"], got, ref64, ref32):
        eh, e32 = _rel(x, y64), _rel(y32, y64)
        report[name] = (eh, e32)
        bound = max(REL_TOL, 2 * e32) if (name, "project_bwd", deg) in STAGE_RELAXED else REL_TOL
        assert eh <= bound, (name, eh, e32, report)
    _evidence(f"stage_project_bwd_deg{deg}_aux{int(aux)}", {k: {"hip_vs_fp64": v[0], "fp32_oracle_vs_fp64": v[1]} for k, v in report.items()})
    if deg is not None and deg < 3:
        assert float(got[4][:, (deg + 1) ** 2:].abs().max()) == 0  # zeros above the active degree


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


def test_project_fwd_leaves_the_views_depth_key_range(gsa):
    """``project_fwd(key_range=...)``: per wave of 64 gaussians the extreme depth keys of its VISIBLE members (culled ones
    carry 0xFFFFFFFF and do not count) and the keys of its members 0 and 32 (the sample the depth table is built from)."""
    from gsplat_amd import scene, stages as S
    W, H, N = 320, 200, 50_037
    sc = _scene(N, seed=21)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    d["means3D"][:500, 2] = -30.0                     # behind the camera: culled (the first seven waves entirely)
    vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
    kr = S.key_range_buffer(N, d["means3D"].device)
    kr.fill_(7)                                       # (every pair is written)
    splat, radii, tiles, keys = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(),
                                              d["shs"], 3, vm, K, W, H, 0, key_range=kr)
    k = keys.long() & 0xFFFFFFFF
    pad = torch.full((kr.shape[0] * 64 - N,), 0xFFFFFFFF, dtype=torch.long, device="cuda")
    kw = torch.cat([k, pad]).view(-1, 64)
    vis = kw != 0xFFFFFFFF
    want_lo = torch.where(vis, kw, torch.full_like(kw, 0xFFFFFFFF)).amin(1)
    want_hi = torch.where(vis, kw, torch.zeros_like(kw)).amax(1)
    got = kr.long() & 0xFFFFFFFF
    assert 0 < int(vis.sum()) < N and int((~vis.any(1)).sum()) >= 7
    assert torch.equal(got[:, 0], want_lo) and torch.equal(got[:, 1], want_hi)
    assert torch.equal(got[:, 2], kw[:, 0]) and torch.equal(got[:, 3], kw[:, 32])      # the sample: lanes 0 and 32


@pytest.mark.parametrize("rem", [1, 63, 64, 65, 129, 192, 193, 255, 0])
def test_key_range_store_stays_inside_an_exact_size_buffer(gsa, rem):
    """The key-range buffer is 4 * ceil(n / 64) words, as include/gsplat_amd.h says, and not one more: a canary right
    behind an exact-size buffer must survive ``project_fwd`` whatever n % 256 is (the kernel's blocks are 256 threads, so
    up to three waves of the last block lie wholly beyond n; under torch's allocator their stores used to land in the
    rounding slack -- for any other C-ABI caller a heap overflow)."""
    from gsplat_amd import scene, stages as S
    W, H = 160, 96
    N = 3 * 256 + rem
    sc = _scene(N, seed=3)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    vm, K = scene.make_camera(0, 1).cuda(), scene.make_intrinsics(W, H).cuda()
    rows = (N + 63) // 64
    arena = torch.full((rows + 16, 4), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    kr = arena[:rows]
    S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"], 3, vm, K, W, H, 0,
                  key_range=kr)
    torch.cuda.synchronize()
    assert bool((arena[rows:] == 0x5A5A5A5A).all()), "project_fwd wrote past ceil(n / 64) key-range entries"
    assert bool((kr[:, 0] != 0x5A5A5A5A).all())      # ... and every entry inside was written


@pytest.mark.parametrize("n,deg,shared_k", [(5000, 3, True), (777, 3, False), (4097, 1, True), (3000, 0, False), (1234, -1, True)])
def test_multi_camera_projection_equals_single_camera_launches(gsa, n, deg, shared_k):
    """``stages.project_views`` (gsa_project_fwd_batch: the views of a step projected in one launch, geometry and
    coefficient rows read once) leaves, per camera, the same BITS as ``project_fwd`` with that camera -- records, radii,
    counts, depth keys, bin rectangles, coefficient digests, per-wave key ranges, quadrant boxes -- for populations that are
    no multiple of the wave or the block, every SH degree, shared and per-camera intrinsics."""
    from gsplat_amd import scene, stages as S
    W, H, C = 320, 200, 3
    sc = _scene(n, seed=5, deg=max(deg, 0))
    means, quats, scales = sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda()
    opac = sc["opacities"][:, 0].contiguous().cuda()
    colors = sc["shs"].cuda() if deg >= 0 else torch.rand(n, 3, device="cuda")
    vms = torch.stack([scene.make_camera(k, 7) for k in (0, 2, 5)]).cuda()
    K1 = scene.make_intrinsics(W, H).cuda()
    Ks = K1 if shared_k else torch.stack([K1, K1 * torch.tensor([[1.1], [0.9], [1.0]], device="cuda"), K1]).contiguous()
    bt = S.bin_tile_for(W, H, 0)
    views = S.project_views(means, quats, scales, opac, colors, deg, vms, Ks, W, H, 0, return_aux=True, bin_tile=bt)
    assert len(views) == C
    for c, v in enumerate(views):
        kr, qb = S.key_range_buffer(n, means.device), S.qbox_buffer(n, means.device)
        one = S.project_fwd(means, quats, scales, opac, colors, deg, vms[c].contiguous(), (Ks if shared_k else Ks[c]).contiguous(),
                            W, H, 0, return_rects=True, return_aux=True, bin_tile=bt, key_range=kr, qbox=qb)
        got = (v.splat, v.radii, v.tiles_touched, v.depth_keys, v.tile_rects, v.sh_aux)
        for name, a, b in zip(("splat", "radii", "tiles_touched", "depth_keys", "tile_rects", "sh_aux"), got, one):
            assert (a is None) == (b is None), name
            if a is not None:
                assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                   b.view(torch.int32) if b.dtype == torch.float32 else b), (name, c)
        assert torch.equal(v.key_range, kr), c
        vis = one[2] > 0    # (quadrant boxes: unspecified for culled gaussians)
        assert (qb is None) == (v.qbox is None)
        if qb is not None:
            assert torch.equal(v.qbox[vis], qb[vis]), c


@pytest.mark.parametrize("n,deg,cams", [(5000, 3, 1), (777, 2, 1), (4097, 1, 3), (3000, 0, 2), (1234, -1, 1), (70_001, 3, 4)])
def test_split_projection_parts_equal_the_whole_stage(gsa, n, deg, cams):
    """``gsa_project_fwd_parts`` (ABI 13): the GEOMETRY part (what the binning reads) and the RECORDS part (splat records +
    coefficient digests, launched on the projection side stream beside the binning) together write the bits the one-launch
    stage writes -- single camera (``project_fwd(records_ready=...)``) and the per-step launch (``project_views(split=True)``)."""
    from gsplat_amd import scene, stages as S
    W, H = 640, 360
    sc = _scene(n, seed=n % 97)
    means, quats, scales = sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda()
    opac = sc["opacities"][:, 0].contiguous().cuda()
    colors = sc["shs"].cuda() if deg >= 0 else (sc["shs"][:, 0, :] * 0.28 + 0.5).contiguous().cuda()
    vms = torch.stack([scene.make_camera(k, 7) for k in range(cams)]).cuda()
    K = scene.make_intrinsics(W, H).cuda()
    bt = S.bin_tile_for(W, H, 0)
    if cams == 1:
        def run(split):
            kr, qb = S.key_range_buffer(n, means.device), S.qbox_buffer(n, means.device)
            forked = [] if split else None
            out = S.project_fwd(means, quats, scales, opac, colors, deg, vms[0], K, W, H, 0, return_rects=True, return_aux=True,
                                bin_tile=bt, key_range=kr, qbox=qb, records_ready=forked)
            if split:
                assert len(forked) == 1
                torch.cuda.current_stream().wait_event(forked[0])
            return [t for t in out if t is not None] + [kr, qb]
        whole, parts = run(False), run(True)
    else:
        def run(split):
            views = S.project_views(means, quats, scales, opac, colors, deg, vms, K, W, H, 0, return_aux=True, split=split)
            assert (views[0].records_ready is not None) == split
            if split:
                torch.cuda.current_stream().wait_event(views[0].records_ready)
            return [getattr(v, f) for v in views for f in ("splat", "radii", "tiles_touched", "depth_keys", "tile_rects", "sh_aux",
                                                           "key_range", "qbox") if getattr(v, f) is not None]
        whole, parts = run(False), run(True)
    assert len(whole) == len(parts)
    for a, b in zip(whole, parts):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)


@pytest.mark.parametrize("n,deg,accumulate,shared_k", [(5000, 3, False, True), (777, 2, True, False), (4097, 1, False, True),
                                                       (100_003, 3, True, True)])
def test_multi_camera_projection_backward_equals_the_accumulating_launches(gsa, n, deg, accumulate, shared_k):
    """``stages.project_views_bwd`` (gsa_project_bwd_batch: the projection backward of all cameras of a step in one launch)
    against the chain it replaces -- one ``project_bwd(accumulate=...)`` per camera: the geometry gradients are the same
    sums in the same order, the SH-coefficient gradient -- formed in registers as sum_c basis(dir_c) x d loss / d rgb_c --
    the sum the per-camera launches leave; both to rounding (1e-5 of the largest entry: the two kernels contract their
    multiply-adds differently; measured 1.4e-6 on random gradient records).  Arbitrary gradient records,
    populations that are no multiple of the wave / block, culled gaussians, shared and per-camera intrinsics."""
    from gsplat_amd import scene, stages as S
    W, H, C = 320, 200, 4
    sc = _scene(n, seed=9, deg=deg)
    means, quats, scales = sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda()
    opac, colors = sc["opacities"][:, 0].contiguous().cuda(), sc["shs"].cuda()
    vms = torch.stack([scene.make_camera(k, 9) for k in (0, 2, 5, 7)]).cuda()
    K1 = scene.make_intrinsics(W, H).cuda()
    Ks = K1 if shared_k else torch.stack([K1, K1 * torch.tensor([[1.1], [0.9], [1.0]], device="cuda"), K1, K1]).contiguous()
    views = S.project_views(means, quats, scales, opac, colors, deg, vms, Ks, W, H, 0, return_aux=True)
    gen = torch.Generator(device="cuda").manual_seed(4)
    v_splats = [torch.randn(n, 16, device="cuda", generator=gen) for _ in range(C)]
    init = [torch.randn_like(t) for t in (means, quats, scales, opac, colors)]
    ref = [t.clone() for t in init]
    for c, v in enumerate(views):
        S.project_bwd(means, quats, scales, opac, colors, deg, v.viewmat, v.Kmat, W, H, v.splat, v.radii, v_splats[c],
                      out=ref, accumulate=accumulate or c > 0, sh_aux=v.sh_aux)
    got = [t.clone() for t in init]
    assert S.project_views_bwd(means, quats, scales, opac, colors, deg, views, v_splats, W, H, got, accumulate=accumulate)
    assert sum(int((v.radii > 0).all(-1).sum()) for v in views) > n   # (the cameras see most of the scene)
    for name, a, b in zip(("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"), got, ref):
        assert torch.isfinite(a).all(), name
        assert _rel(a, b) <= 1e-5, (name, _rel(a, b))
    # without digests (or without view-dependent colour) the batched kernel does not apply: the caller is told
    plain = S.project_views(means, quats, scales, opac, colors, deg, vms, Ks, W, H, 0, return_aux=False)
    assert S.project_views_bwd(means, quats, scales, opac, colors, deg, plain, v_splats, W, H, got) is False


@pytest.mark.parametrize("deg", [3, 1])
def test_rank_step_colour_gradients_out_of_the_sweep_and_one_geometry_backward(gsa, deg):
    """The step of a view-parallel rank (r05): ``rasterization(projection=deferred, sh_grad_factored=slot)`` makes every view's
    backward SWEEP leave its masked d loss / d rgb in the exchange's [N,3] slot (``gsa_composite_bwd_lists_rgb``), and
    ``project_cameras_backward`` then sums the cameras' GEOMETRY gradients in one launch (``gsa_project_bwd_batch_geometry``).
    Against the chain it replaces -- one factored projection backward per view, accumulating: same colour slabs, same geometry
    sums (float atomics reorder the sweeps' sums between two runs: 2e-5), and the coefficient gradient rebuilt from the slabs
    equals the per-view autograd one."""
    from gsplat_amd import scene, stages as S
    W, H, N, C = 320, 192, 9000, 3
    sc = _scene(N, seed=77 + deg)
    vms = torch.stack([scene.make_camera(k, 7) for k in (1, 4, 6)]).cuda()
    K = scene.make_intrinsics(W, H).cuda()
    v = torch.from_numpy(np.random.default_rng(deg).uniform(-1, 1, (C, H, W, 3)).astype(np.float32)).cuda()
    bg = torch.full((1, 3), 0.2, device="cuda")
    # (colours pushed towards the clamp so that the mask matters)
    base = [sc[k].cuda() for k in NAMES]
    base[4] = base[4] - 0.9
    for attempt in range(2):   # (first pass: the first calls of the frame size, stage by stage; second: the fused per-view calls)
        # reference: the per-view chain with the factored projection backward, accumulating
        lv = [t.clone().requires_grad_(True) for t in base]
        args = (lv[0], lv[1], lv[2], lv[3][:, 0], lv[4])
        ref_out = [torch.zeros_like(t.detach()) for t in args]
        ref_rgb = torch.zeros(C, N, 3, device="cuda")
        for c in range(C):
            r, _, _ = gsa.rasterization(*args, vms[c:c + 1], K[None], W, H, sh_degree=deg, backgrounds=bg, grad_out=ref_out,
                                        grad_accumulate=c > 0, sh_grad_factored=ref_rgb[c])
            (r[0] * v[c]).sum().backward()
        # the rank step
        proj = gsa.project_cameras(*args, vms, K, W, H, sh_degree=deg, defer_backward=True)
        assert proj.deferred
        rgb = torch.full((C, N, 3), float("nan"), device="cuda")
        for c in range(C):
            r, _, _ = gsa.rasterization(*args, vms[c:c + 1], K[None], W, H, sh_degree=deg, backgrounds=bg, projection=proj[c],
                                        sh_grad_factored=rgb[c])
            (r[0] * v[c]).sum().backward()
        out = [torch.full_like(t.detach(), float("nan")) for t in args]
        gsa.project_cameras_backward(proj, *args, out)
        assert torch.isnan(out[4]).all()                       # the coefficient gradient is the caller's to rebuild
        scale_rgb = float(ref_rgb.abs().max())
        assert scale_rgb > 0 and float((rgb - ref_rgb).abs().max()) <= 2e-5 * scale_rgb
        assert float((rgb == 0).float().mean()) > 0.05         # (clamped channels and culled gaussians: exact zeros)
        for name, a, b in zip(NAMES[:4], out[:4], ref_out[:4]):
            assert _rel(a, b) <= 2e-5, (name, _rel(a, b))
        S.sh_grad_expand(lv[0].detach(), deg, vms, K[None], rgb, out[4])
        S.sh_grad_expand(lv[0].detach(), deg, vms, K[None], ref_rgb, ref_out[4])
        assert _rel(out[4], ref_out[4]) <= 2e-5


def test_deferred_projection_backward_of_a_step_equals_per_camera_autograd(gsa):
    """The product path of a multi-view step: ``project_cameras(defer_backward=True)`` -> one ``rasterization(projection=...)``
    per camera -> ``backward()`` per camera (stops behind its sweep) -> ``project_cameras_backward`` once.  The gradient bucket
    equals the sum of the per-camera autograd gradients of plain single-camera calls."""
    from gsplat_amd import scene
    W, H, N, C = 256, 160, 6000, 3
    sc = _scene(N, seed=21)
    vms = torch.stack([scene.make_camera(k, 7) for k in (0, 3, 5)]).cuda()
    K = scene.make_intrinsics(W, H).cuda()[None]
    v = torch.from_numpy(np.random.default_rng(8).uniform(-1, 1, (C, H, W, 3)).astype(np.float32)).cuda()
    bg = torch.zeros(1, 3, device="cuda")

    def leaves():
        return [sc[k].cuda().clone().requires_grad_(True) for k in NAMES]

    ref = None
    for _ in range(2):   # (first pass: first calls of the shape)
        lv = leaves()
        for c in range(C):
            r, _, _ = gsa.rasterization(lv[0], lv[1], lv[2], lv[3][:, 0], lv[4], vms[c:c + 1], K, W, H, sh_degree=3, backgrounds=bg)
            (r * v[c]).sum().backward()
        ref = [t.grad.clone() for t in lv]
    lv = leaves()
    proj = gsa.project_cameras(lv[0], lv[1], lv[2], lv[3][:, 0], lv[4], vms, K[0], W, H, sh_degree=3, defer_backward=True)
    assert all(p.defer_backward for p in proj)
    renders = []
    for c in range(C):
        r, _, _ = gsa.rasterization(lv[0], lv[1], lv[2], lv[3][:, 0], lv[4], vms[c:c + 1], K, W, H, sh_degree=3, backgrounds=bg,
                                    projection=proj[c])
        renders.append(r)
    with pytest.raises(RuntimeError):   # nothing backpropagated yet
        gsa.project_cameras_backward(proj, lv[0], lv[1], lv[2], lv[3][:, 0], lv[4], [torch.empty(1)] * 5)
    for c in range(C):
        (renders[c] * v[c]).sum().backward()
    assert all(t.grad is None for t in lv)   # the cameras' own backward passes stopped behind their sweeps
    out = [torch.full_like(t.detach(), float("nan")) for t in (lv[0], lv[1], lv[2], lv[3][:, 0].contiguous(), lv[4])]
    gsa.project_cameras_backward(proj, lv[0], lv[1], lv[2], lv[3][:, 0], lv[4], out)
    for name, a, b in zip(NAMES, out, ref):
        assert _rel(a.reshape(b.shape), b) <= 1e-5, (name, _rel(a.reshape(b.shape), b))


@pytest.mark.parametrize("W,H,N,s_max", [(1920, 1080, 60_000, 0.05), (1000, 600, 20_000, 0.1), (3840, 2160, 50_000, 0.05)])
def test_launch_order_from_the_placement_launch_equals_tile_schedule(gsa, W, H, N, s_max):
    """``bin_tiles`` on the one-pass placement leaves the compositor's launch order itself (gsa_tile_bin16_scheduled: eight
    extra workgroups of the placement launch): per XCD band the same tiles as ``gsa_tile_schedule`` writes, in the same
    bucket order (list length / 64, longest first; the order inside a bucket is free: LDS atomics), the tiles of one bin on
    consecutive places, the same padding -- and the image rendered with it is the image."""
    from gsplat_amd import scene, stages as S
    sc = _scene(N, seed=13, s_max=s_max)
    means, quats, scales = sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda()
    opac, colors = sc["opacities"][:, 0].contiguous().cuda(), sc["shs"].cuda()
    vm, K = scene.make_camera(1, 6).cuda().contiguous(), scene.make_intrinsics(W, H).cuda()
    bt = S.bin_tile_for(W, H, 0)
    assert bt in (32, 64)

    def run():
        kr, qb = S.key_range_buffer(N, means.device), S.qbox_buffer(N, means.device)
        splat, radii, tiles, keys, rects = S.project_fwd(means, quats, scales, opac, colors, 3, vm, K, W, H, 0, return_rects=True,
                                                         bin_tile=bt, key_range=kr, qbox=qb)
        return splat, S.bin_tiles(splat, radii, tiles, keys, W, H, 0, tile_rects=rects, bin_tile=bt, key_range=kr, qbox=qb)

    run()[1].finalize()                      # (first call of the shape: exact sizes)
    splat, bins = run()
    assert bins.finalize()[1] and bins.tile_order is not None and S.tile_schedule(bins, W, H, bt) is bins.tile_order
    ref = S.tile_schedule(bins.tile_offsets, W, H, bt)          # a launch of its own
    got = bins.tile_order
    assert got.shape == ref.shape
    tw, th = S.tile_grid(W, H)
    n_tiles, chunk = tw * th, (tw * th + 7) // 8
    side = bt // 16
    bw = (tw + side - 1) // side
    offs = bins.tile_offsets.cpu().long()

    def bucket(tile):
        ty, tx = divmod(int(tile), tw)
        idx = (ty // side) * bw + tx // side
        return 63 - min(63, int(offs[idx + 1] - offs[idx]) >> 6)

    g, r = got.cpu().tolist(), ref.cpu().tolist()
    for band in range(8):
        lo, hi = band * chunk, min(n_tiles, (band + 1) * chunk)
        gb, rb = g[lo:lo + chunk], r[lo:lo + chunk]
        live = max(hi - lo, 0)
        assert sorted(gb[:live]) == list(range(lo, hi)) == sorted(rb[:live]) and all(t == -1 for t in gb[live:])
        keys = [bucket(t) for t in gb[:live]]
        assert keys == sorted(keys) and keys == [bucket(t) for t in rb[:live]]
    bg = torch.zeros(3, device="cuda")
    a = S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, W, H, 3, got, bin_tile=bt, gid_boxes=bins.boxed)
    b = S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg, W, H, 3, ref, bin_tile=bt, gid_boxes=bins.boxed)
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])


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


def test_record_capacity_miss_is_repaired(gsa):
    """The forward pass is queued for a remembered record capacity without waiting for the live count; when the
    scene outgrows it the pass must notice and repeat with exact sizes -- same image, same record list."""
    from gsplat_amd import scene, stages as S
    W, H, N = 320, 200, 20_000
    sc = _scene(N, seed=3, s_min=0.005, s_max=0.08)
    args = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], sc["shs"]]
    args = [t.cuda() for t in args] + [scene.make_camera(1, 5)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()]
    with torch.no_grad():
        r0, a0, m0 = gsa.rasterization(*args, W, H, sh_degree=3)          # first call: exact sizes
        ids0 = m0["flatten_ids"].clone()
        r1, a1, m1 = gsa.rasterization(*args, W, H, sh_degree=3)          # steady state: capacity + device count
        key = ("cuda", torch.cuda.current_device(), W, H, S.DEFAULT_CULL_MODE, S.bin_tile_for(W, H, S.DEFAULT_CULL_MODE))
        assert S._capacity_for(S.DEFAULT_SPECULATION, key, N) >= m0["n_binned"]                   # (records of the bins the hot path sorts)
        S.DEFAULT_SPECULATION.capacity[key][0] = m0["n_binned"] / N / 3                       # pretend the scene grew threefold
        r2, a2, m2 = gsa.rasterization(*args, W, H, sh_degree=3)
    for r, a, m in ((r1, a1, m1), (r2, a2, m2)):
        assert torch.equal(r, r0) and torch.equal(a, a0)
        assert m["n_isects"] == m0["n_isects"] and torch.equal(m["flatten_ids"], ids0)
    assert S._capacity_for(S.DEFAULT_SPECULATION, key, N) >= m0["n_binned"]                       # the miss re-learned the capacity


def test_deferred_check_reports_a_miss_and_the_next_call_is_right(gsa):
    """``defer_check=True``: the call does not wait for the view's control words (the host queues views far ahead of the
    device); ``meta['speculation_ok']`` says afterwards whether the remembered capacity / key range held.  In steady state
    it is True and image, gradients and meta entries equal the checked call's; after a simulated threefold growth it is
    False, and rendering the view again gives the checked call's result.  Also with the whole call on a stream of its own
    (how ViewPipeline runs the views of a step)."""
    from gsplat_amd import scene, stages as S
    W, H, N = 320, 200, 20_000
    sc = _scene(N, seed=3, s_min=0.005, s_max=0.08)
    base = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0], sc["shs"]]
    cam = [scene.make_camera(1, 5)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()]
    w = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(9)).cuda()
    side = torch.cuda.Stream()

    def run(**kw):
        leaves = [t.cuda().clone().requires_grad_(True) for t in base]
        r, a, m = gsa.rasterization(*leaves, *cam, W, H, sh_degree=3, **kw)
        (r * w).sum().backward()
        torch.cuda.synchronize()
        return r.detach(), a.detach(), m, [t.grad for t in leaves]

    run()                                                                   # first call of the shape: exact sizes
    r0, a0, m0, g0 = run()
    key = ("cuda", torch.cuda.current_device(), W, H, S.DEFAULT_CULL_MODE, S.bin_tile_for(W, H, S.DEFAULT_CULL_MODE))
    for kw, st in (({"defer_check": True}, None), ({"defer_check": True}, side), ({}, side)):
        if st is not None:
            st.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            r1, a1, m1, g1 = run(**kw)
        assert m1["speculation_ok"] is True
        assert torch.equal(r1, r0) and torch.equal(a1, a0)
        assert all(_rel(x.reshape(-1).cpu(), y.reshape(-1).cpu()) < 1e-5 for x, y in zip(g1, g0)), kw   # (float atomics: not bit-equal run to run)
        assert m1["n_binned"] == m0["n_binned"] and m1["n_isects"] == m0["n_isects"]
        assert torch.equal(m1["flatten_ids"], m0["flatten_ids"]) and torch.equal(m1["isect_ids"], m0["isect_ids"])
    S.DEFAULT_SPECULATION.capacity[key][0] = m0["n_binned"] / N / 3                            # pretend the scene grew threefold
    r2, a2, m2, g2 = run(defer_check=True)
    assert m2["speculation_ok"] is False                                    # ... reported, not repaired
    assert m2["n_binned"] < m0["n_binned"] <= S._capacity_for(S.DEFAULT_SPECULATION, key, N)       # (clamped to the capacity it ran with; the miss re-learned it)
    r3, a3, m3, g3 = run(defer_check=True)
    assert m3["speculation_ok"] is True and torch.equal(r3, r0)
    assert all(_rel(x.reshape(-1).cpu(), y.reshape(-1).cpu()) < 1e-5 for x, y in zip(g3, g0))


def test_population_change_keeps_the_pipeline_asynchronous(gsa):
    """Densify / prune changes N every 100 iterations (/root/reference/src/training/trainer.py:794-797): the record
    capacity is remembered per scene SHAPE as records per gaussian, so a population 5 % larger or smaller is binned
    without a blocking read-back (no exact-size pass, no second scan), and the cache stays bounded."""
    from gsplat_amd import scene, stages as S
    W, H, N = 320, 200, 20_000
    sc = _scene(N + N // 20, seed=3, s_min=0.005, s_max=0.08)
    K, vm = scene.make_intrinsics(W, H).cuda(), scene.make_camera(1, 5).cuda()
    key = ("cuda", torch.cuda.current_device(), W, H, 0, 16)
    S.DEFAULT_SPECULATION.capacity.pop(key, None)

    def bins(n):
        t = [sc[k][:n].cuda().contiguous() for k in ("means3D", "rotations", "scales", "opacities", "shs")]
        splat, radii, tiles, keys, rects = S.project_fwd(t[0], t[1], t[2], t[3][:, 0].contiguous(), t[4], 3, vm, K, W, H, 0,
                                                         return_rects=True)
        return S.bin_tiles(splat, radii, tiles, keys, W, H, 0, tile_rects=rects), int(tiles.sum())

    first, m = bins(N)
    assert first._m == m                        # first call of a shape: the one blocking read-back (exact sizes)
    for n in (N + N // 20, N - N // 20, N):
        b, m = bins(n)
        assert b._m is None, "a changed population must not cost a blocking read-back"
        assert b.capacity >= m and b.finalize() == (m, True)
    assert len(S.DEFAULT_SPECULATION.capacity) <= S._CAPACITY_MAX_KEYS
    # two bins left un-finalized have their own read-back slots
    b1, m1 = bins(N)
    b2, m2 = bins(N - N // 20)
    assert m1 != m2 and b2.finalize()[0] == m2 and b1.finalize()[0] == m1
    # ... and ANY number of bins may wait un-finalized (a deferred-check step over many views, a 40-camera call): the
    # pool of read-back slots grows on demand (r03: a ring of 16 that raised beyond it), every one still reads its own count
    old, m_old = bins(N - N // 20)
    later = [bins(N) for _ in range(3 * S._PINNED_BLOCK)]
    assert old.finalize() == (m_old, True)
    assert all(b.finalize() == (m, True) for b, m in later)
    free = S._PINNED[("cuda", torch.cuda.current_device())]
    assert len(free) >= 3 * S._PINNED_BLOCK                # ... and the slots are back in the pool


def test_two_cameras_in_one_call_equal_two_calls(gsa):
    """C = 2 cameras in one rasterization() call (the reference's dead ``render_batch`` shape, gsplat_backend.py:12-55):
    images, alphas and gradients are those of two single-camera calls, and the 16 x 16 tile lists of ``meta`` (built on
    demand when the hot path bins by 32 x 32) are the per-camera lists with the camera offset the published layout has."""
    from gsplat_amd import scene
    W, H, N = 200, 120, 4000
    sc = _scene(N, seed=11)
    vms = torch.stack([scene.make_camera(k, 5) for k in (0, 2)]).cuda()
    Ks = torch.stack([scene.make_intrinsics(W, H)] * 2).cuda()
    v = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (2, H, W, 3)).astype(np.float32)).cuda()

    def run(cams, **kw):
        leaves = [sc[k].cuda().clone().requires_grad_(True) for k in NAMES]
        r, a, meta = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vms[cams], Ks[cams], W, H,
                                       sh_degree=3, backgrounds=torch.zeros(len(cams), 3, device="cuda"), **kw)
        (r * v[cams]).sum().backward()
        return r.detach(), a.detach(), meta, [t.grad for t in leaves]

    run([0, 1])                                        # (first call of the shape: exact sizes)
    r2, a2, m2, g2 = run([0, 1])
    # the same call without waiting for either camera's control words: same results, and the meta entries that depend on
    # the record counts wait for them on access
    rd, ad, md, gd = run([0, 1], defer_check=True)
    assert md["speculation_ok"] is True and torch.equal(rd, r2) and torch.equal(ad, a2)
    assert md["n_isects"] == m2["n_isects"] and torch.equal(md["flatten_ids"], m2["flatten_ids"])
    assert torch.equal(md["isect_ids"], m2["isect_ids"]) and torch.equal(md["isect_offsets"], m2["isect_offsets"])
    for name, x, y in zip(NAMES, gd, g2):
        assert _rel(x, y) < 1e-5, name
    singles = [run([k]) for k in (0, 1)]
    assert r2.shape == (2, H, W, 3) and m2["n_cameras"] == 2
    for k, (r1, a1, m1, _) in enumerate(singles):
        assert torch.equal(r2[k], r1[0]) and torch.equal(a2[k], a1[0])
        assert torch.equal(m2["radii"][k], m1["radii"][0]) and torch.equal(m2["means2d"][k], m1["means2d"][0])
    assert m2["n_isects"] == sum(s[2]["n_isects"] for s in singles)
    ids = torch.cat([singles[0][2]["flatten_ids"], singles[1][2]["flatten_ids"] + N])
    assert torch.equal(m2["flatten_ids"], ids)
    assert m2["isect_offsets"].shape == (2,) + tuple(singles[0][2]["isect_offsets"].shape[1:])
    for name, x, y0, y1 in zip(NAMES, g2, singles[0][3], singles[1][3]):
        assert _rel(x, y0 + y1) < 1e-5, name


@pytest.mark.parametrize("mode", ["D", "ED", "RGB+D", "RGB+ED"])
def test_depth_render_modes(gsa, mode):
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization
    W, H = 96, 64
    sc = _scene(1500, seed=5)
    vm, K = scene.make_camera(1, 6), scene.make_intrinsics(W, H)
    outs, keep = [], None
    for fn, dev, kw in ((oracle_rasterization, "cpu", {"ambiguity_tol": _amb_tol_e2e(W, H)}), (gsa.rasterization, "cuda", {})):
        leaves = [sc[k].to(dev).clone().requires_grad_(True) for k in NAMES]
        r, a, meta = fn(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm[None].to(dev),
                        K[None].to(dev), W, H, sh_degree=2, render_mode=mode,
                        backgrounds=torch.zeros(1, 3, device=dev), **kw)
        if keep is None:
            keep = (~meta["ambiguous"]).float()[..., None]
        w = torch.linspace(0.5, 1.5, r.shape[-1], device=dev)
        (r * w * keep.to(dev)).sum().backward()
        outs.append((r.detach().cpu() * keep, a.detach().cpu() * keep,
                     [torch.zeros_like(t) if t.grad is None else t.grad for t in leaves]))
    (r0, a0, g0), (r1, a1, g1) = outs
    assert r1.shape == r0.shape and float(1 - keep.mean()) <= MAX_AMB_FRACTION
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
    # the shared (multi-camera) projection on the same degenerate scenes: everything culled for both cameras -> zero
    # gradients out of the one projection-backward launch; an EMPTY scene for two cameras
    vm2, K2 = torch.cat([vm, vm]), torch.cat([K, K])
    leaves = [t.cuda().clone().requires_grad_(True) for t in base]
    with torch.no_grad():
        leaves[0][:, 2] = -10.0
    proj = gsa.project_cameras(*leaves, vm2, K2, W, H, sh_degree=3, defer_backward=True)
    for c in range(2):
        r, a, meta = gsa.rasterization(*leaves, vm2[c:c + 1], K2[c:c + 1], W, H, sh_degree=3, backgrounds=bg, projection=proj[c])
        assert torch.allclose(r[0], bg[0].expand(H, W, 3)) and float(a.abs().max()) == 0
        r.sum().backward()
    out = [torch.full_like(t.detach(), float("nan")) for t in leaves]
    gsa.project_cameras_backward(proj, *leaves, out)
    assert all(float(t.abs().max()) == 0 for t in out)
    none = [t[:0].cuda() for t in base]
    r, a, meta = gsa.rasterization(*none, vm2, K2, W, H, sh_degree=3)
    assert r.shape == (2, H, W, 3) and float(r.abs().max()) == 0 and float(a.abs().max()) == 0
    # wrong shapes / devices raise, they do not crash
    with pytest.raises(ValueError):
        gsa.rasterization(one[0], one[1], one[2], one[3][:, None], one[4], vm, K, W, H, sh_degree=3)
    with pytest.raises(ValueError):   # a projection made for another frame
        gsa.rasterization(*one, vm, K, W, H, sh_degree=3,
                          projection=gsa.project_cameras(*one, vm, K, 2 * W, H, sh_degree=3)[0])
    with pytest.raises(RuntimeError):
        gsa.rasterization(*[t.cpu() for t in one], vm.cpu(), K.cpu(), W, H, sh_degree=3)


def test_views_free_their_memory_without_the_cyclic_collector(gsa):
    """A view's buffers (the fused chain's 0.4-GB workspace at bench size) must go back to the allocator by reference counting
    alone: bench.py and any long-running trainer freeze or disable the cyclic collector, and a reference cycle through the
    autograd node (r05: node -> chain -> output tensor -> grad_fn = node) leaked a workspace per view until the card was
    full.  With the collector off, 40 forward + backward views -- through ``rasterization`` and through the deferred-step
    form -- leave the allocated bytes where they were after the first few."""
    import gc
    from gsplat_amd import scene
    W, H, N = 640, 352, 40_000
    sc = _scene(N, seed=5)
    leaves = [sc[k].cuda().clone().requires_grad_(True) for k in NAMES]
    args = (leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4])
    vms = torch.stack([scene.make_camera(k, 5) for k in range(3)]).cuda()
    K = scene.make_intrinsics(W, H).cuda()
    v = torch.rand(H, W, 3, device="cuda")
    out = [torch.empty_like(t.detach()) for t in args]

    def single(i):
        for t in leaves:
            t.grad = None
        r, _, meta = gsa.rasterization(*args, vms[i % 3:i % 3 + 1], K[None], W, H, sh_degree=3)
        (r[0] * v).sum().backward()

    def step(i):
        proj = gsa.project_cameras(*args, vms, K, W, H, sh_degree=3, defer_backward=True)
        for c in range(3):
            r, _, meta = gsa.rasterization(*args, vms[c:c + 1], K[None], W, H, sh_degree=3, projection=proj[c], defer_check=True)
            (r[0] * v).sum().backward()
            assert meta["speculation_ok"] in (True, False)
        gsa.project_cameras_backward(proj, *args, out)

    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        for fn in (single, step):
            for i in range(6):
                fn(i)
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
            for i in range(40):
                fn(i)
            torch.cuda.synchronize()
            grown = torch.cuda.memory_allocated() - base
            assert grown <= 8 << 20, f"{fn.__name__}: {grown / 2**20:.1f} MiB allocated and not freed over 40 iterations"
    finally:
        if was:
            gc.enable()


def test_projection_is_checked_against_the_camera_and_the_parameters(gsa):
    """``rasterization(projection=...)`` refuses a projection made for another camera, or from parameters that were written
    since: the forward would blend the projection's splats while the backward differentiates the arguments.
    Host-side compares of pointers and version counters; the legitimate uses stay silent."""
    from gsplat_amd import scene
    W, H, N = 160, 96, 1500
    sc = _scene(N, seed=4)
    leaves = [sc[k].cuda().clone().requires_grad_(True) for k in NAMES]
    args = (leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4])
    vms = torch.stack([scene.make_camera(k, 5) for k in (0, 2)]).cuda()
    K = scene.make_intrinsics(W, H).cuda()
    proj = gsa.project_cameras(*args, vms, K, W, H, sh_degree=3)
    assert proj.deferred is False   # (not asked for)
    gsa.rasterization(*args, vms[1:2], K[None], W, H, sh_degree=3, projection=proj[1])
    gsa.rasterization(*args, proj[0].viewmat[None], proj[0].Kmat[None], W, H, sh_degree=3, projection=proj[0])
    with pytest.raises(ValueError, match="camera"):   # camera 0's tensors with camera 1's projection
        gsa.rasterization(*args, vms[0:1], K[None], W, H, sh_degree=3, projection=proj[1])
    with pytest.raises(ValueError, match="camera"):   # an equal-valued COPY is not the tensor either (no device compare)
        gsa.rasterization(*args, vms[1:2].clone(), K[None], W, H, sh_degree=3, projection=proj[1])
    with torch.no_grad():
        leaves[0].add_(0.01)                           # an optimizer step
    with pytest.raises(ValueError, match="written since"):
        gsa.rasterization(*args, vms[1:2], K[None], W, H, sh_degree=3, projection=proj[1])
    # a step that asks for the shared backward where it does not apply (colours without SH) is told so
    flat = [leaves[0], leaves[1], leaves[2], leaves[3][:, 0], (leaves[4][:, 0, :] * 0.28 + 0.5).detach().requires_grad_(True)]
    p2 = gsa.project_cameras(*flat, vms, K, W, H, sh_degree=None, defer_backward=True)
    assert p2.deferred is False
    with pytest.raises(ValueError, match="not made with a deferred backward"):
        gsa.project_cameras_backward(p2, *flat, [torch.empty(1)] * 5)


def test_two_backends_of_equal_frame_size_do_not_share_speculation(gsa):
    """what the asynchronous binning remembers (record capacity, depth-key range, bucket plan) lives in
    an object owned by the backend / pipeline, not in module globals keyed by frame size: a trainer's backend and an
    evaluator's backend that render frames of the same size from DIFFERENT scenes and cameras interleave without a single
    cross-induced repair, each remembers its own key range, and the process-wide default state never hears of the frame."""
    from gsplat_amd import scene, stages as S
    W, H, N = 640, 352, 40_000
    K = scene.make_intrinsics(W, H).cuda()
    a, b = gsa.HipBackend(K), gsa.HipBackend(K)
    assert a.speculation is not b.speculation and a.speculation is not S.DEFAULT_SPECULATION
    far = _scene(N, seed=31, s_max=0.03)                      # seen from the ring of radius 4: a narrow depth range, few records
    near = _scene(N, seed=32, s_min=0.02, s_max=0.2)          # big splats seen from INSIDE the cloud: wide range, many records

    def params(sc):
        return {"means3D": sc["means3D"].cuda(), "rotations": sc["rotations"].cuda(), "scales": sc["scales"].cuda(),
                "opacities": sc["opacities"].cuda(), "shs": sc["shs"].cuda()}

    pa, pb = params(far), params(near)
    cams_a = [scene.make_camera(k, 6).cuda() for k in range(6)]
    cams_b = [scene.make_camera(k, 6, radius=0.35).cuda() for k in range(6)]
    key = ("cuda", torch.cuda.current_device(), W, H, 0, S.bin_tile_for(W, H, 0))
    S.DEFAULT_SPECULATION.capacity.pop(key, None)
    S.DEFAULT_SPECULATION.key_range.pop(key, None)
    bg = torch.zeros(3, device="cuda")

    def view(backend, p, vm):
        return backend.render_with_depth(p, {"world_view_transform": vm, "image_width": W, "image_height": H}, bg, "RGB", "cuda", 3)

    with torch.no_grad():
        alone = [view(gsa.HipBackend(K), pa, cams_a[0])["render"].clone()]   # (a backend of its own: the reference image)
        for vm_a, vm_b in zip(cams_a, cams_b):                  # every camera once: first calls, growth of the remembered ranges
            view(a, pa, vm_a), view(b, pb, vm_b)
        for st in (a.speculation.repairs, b.speculation.repairs):
            for k in st:
                st[k] = 0
        for _ in range(3):
            for vm_a, vm_b in zip(cams_a, cams_b):
                view(a, pa, vm_a)
                view(b, pb, vm_b)
    assert a.speculation.repairs["views"] == 18 and b.speculation.repairs["views"] == 18
    assert a.speculation.repairs["capacity_misses"] == 0 and a.speculation.repairs["key_range_repairs"] == 0, a.speculation.repairs
    assert b.speculation.repairs["capacity_misses"] == 0, b.speculation.repairs
    ka, kb = a.speculation.key_range[key], b.speculation.key_range[key]
    assert ka[:2] != kb[:2] and (kb[1] - kb[0]) > 4 * (ka[1] - ka[0])       # each remembers ITS cameras' depth range
    assert a.speculation.capacity[key][0] < 0.5 * b.speculation.capacity[key][0]   # ... and ITS records per gaussian
    assert key not in S.DEFAULT_SPECULATION.capacity and key not in S.DEFAULT_SPECULATION.key_range
    with torch.no_grad():
        again = view(a, pa, cams_a[0])["render"]
    assert torch.equal(again, alone[0])


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
    # every pixel: within 1e-4 of the fp64 oracle outright where its threshold decisions are unambiguous, within what its
    # ambiguous decisions can move it where they are not (the mask logic of _compare / _masked_pixels)
    from oracle import rasterization as oracle_rasterization
    r64, _, m64 = oracle_rasterization(sc["means3D"].double(), sc["rotations"].double(), sc["scales"].double(),
                                       sc["opacities"][:, 0].double(), sc["shs"].double(), vm[None].double(), K[None].double(),
                                       W, H, sh_degree=2, backgrounds=torch.zeros(1, 3, dtype=torch.float64),
                                       ambiguity_tol=_amb_tol_e2e(W, H))
    masked, bound = m64["ambiguous"][0], m64["flip_bound"][0]
    assert float(masked.float().mean()) <= MAX_AMB_FRACTION
    cont = REL_TOL * float(r64.abs().max())
    diff = (out["render"].cpu().double() - r64[0]).abs().amax(-1)
    assert float(diff[~masked].max()) <= cont, float(diff[~masked].max())
    assert float((diff - bound - cont)[masked].max() if masked.any() else 0.0) <= 0.0
    assert float((ref["render"].double() - r64[0]).abs().amax(-1)[~masked].max()) <= cont   # (the fp32 oracle backend too)
    assert torch.equal(out["visibility_filter"].cpu(), ref["visibility_filter"])
    full = be.render_with_depth(sc_d, vp_d)  # defaults: RGB+ED, degree 3, zero background
    assert set(full) >= {"render", "expected_depth", "alpha", "means2d", "radii", "visibility_filter"}
    single = be.render_single(sc_d, vp_d, torch.zeros(3, device="cuda"))
    assert torch.allclose(single["render"], full["render"]) and single["depth"].shape == (H, W)
    batch = be.render_batch(sc_d, vm.cuda()[None].repeat(2, 1, 1), torch.eye(4, device="cuda")[None].repeat(2, 1, 1),
                            W, H, torch.zeros(3, device="cuda"), "cuda")
    assert batch["images"].shape == (2, 3, H, W) and batch["alphas"].shape == (2, 1, H, W)
    # (one multi-camera operator call -- a shared projection launch -- gives every camera the single-view image)
    for c in range(2):
        assert torch.equal(batch["images"][c], single["render"].permute(2, 0, 1))
        assert torch.equal(batch["alphas"][c], single["alpha"].permute(2, 0, 1))


def _parity_record(name, record):
    """Evidence file of a config-sized parity test: where the GPU run's artefacts are collected (committed copies:
    profiles/r04_parity_*.json)."""
    try:
        out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                               "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, name + ".json"), "w") as fh:
            json.dump(record, fh, indent=1)
    except OSError:
        pass


_ERR_KEYS = ("ambiguous_fraction", "masked_pixels")


@pytest.mark.parametrize("cam", [(0, 1), (3, 8)])
def test_config_c2_100k_1080p_deg3(gsa, cam):
    """BASELINE config C2: 100k gaussians, 1920x1080, SH degree 3, forward+backward, vs the CPU oracle -- on view 0
    (identity rotation) and on ring camera 3 of 8 (rotated by 135 degrees: every entry of R is in play)."""
    from gsplat_amd import scene
    W, H = 1920, 1080
    sc = scene.make_scene(100_000, seed=0, sh_degree=3)
    m1, m0, errs = _compare(sc, scene.make_camera(*cam), scene.make_intrinsics(W, H), W, H, 3,
                            bg=torch.zeros(3), use_v_alpha=False, checkpoint=True)
    print("C2 (err HIP vs fp64, err fp32-oracle vs fp64):", errs, "M =", m1["flatten_ids"].numel())
    # evidence: per tensor (err of HIP vs the fp64 oracle, err of the fp32 oracle vs the fp64 oracle), the masked pixels
    _parity_record(f"parity_c2_cam{cam[0]}of{cam[1]}", {
        "config": f"C2: 100000 gaussians, 1920x1080, SH degree 3, forward+backward, ring camera {cam[0]} of {cam[1]}",
        "metric": "max|a - b| / max|b| per tensor, b = fp64 oracle; pixels whose threshold comparisons are "
                  "ambiguous in the fp64 oracle are masked on both sides and checked against their flip bound instead",
        "masked_pixel_fraction": errs["ambiguous_fraction"], "intersections_M": int(m1["flatten_ids"].numel()),
        "masked_pixels": errs["masked_pixels"],
        "tensors": {k: {"err_hip": v[0], "err_fp32_oracle": v[1]} for k, v in errs.items() if k not in _ERR_KEYS}})
    assert 0 < m1["flatten_ids"].numel() <= m0["flatten_ids"].numel() + 64  # default cull_mode drops unreachable tiles


def test_config_c2_step_of_4_cameras(gsa):
    """The step form bench.py's `value` is quoted on, at BASELINE config C2's size, DIRECTLY against the oracle:
    ``project_cameras(defer_backward=True)`` (one projection launch for the step's four cameras) -> 4 x
    ``rasterization(projection=...)`` + ``backward()`` (each stops behind its sweep) -> ``project_cameras_backward`` (one
    launch) -- the gradient bucket against the SUM of the fp64 oracle's per-camera gradients, every image against the fp64
    oracle's, 1e-4 outright.  (Each camera's upstream gradient is masked at the pixels the fp64 oracle finds
    threshold-ambiguous for that camera, on both sides, as in ``_compare``.)"""
    from gsplat_amd import scene
    from oracle import rasterization as oracle_rasterization
    W, H, C = 1920, 1080, 4
    sc = scene.make_scene(100_000, seed=0, sh_degree=3)
    K = scene.make_intrinsics(W, H)
    cams = [scene.make_camera(c, C) for c in range(C)]
    bg = torch.zeros(3)
    rng = np.random.default_rng(12)
    v = [torch.from_numpy(rng.uniform(-1, 1, (H, W, 3))) for _ in range(C)]
    want = {n: 0.0 for n in NAMES}
    keeps, renders64, amb = [], [], []
    for c in range(C):
        o64, m64, keep = _run(oracle_rasterization, "cpu", torch.float64, sc, cams[c], K, W, H, 3, bg, v[c], None,
                              ambiguity_tol=_amb_tol_e2e(W, H), checkpoint=True)
        assert 1.0 - float(keep.mean()) <= MAX_AMB_FRACTION
        amb.append(1.0 - float(keep.mean()))
        keeps.append(keep)
        renders64.append(o64["render"])
        for n in NAMES:
            want[n] = want[n] + o64["grad_" + n]
    leaves = [sc[k].cuda().clone().requires_grad_(True) for k in NAMES]
    args = (leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4])
    vms, Kd = torch.stack(cams).cuda(), K.cuda()
    errs = {}
    for attempt in range(2):   # (the first pass meets the first calls of the frame size: exact sizes, LSD ordering; the second the steady state)
        proj = gsa.project_cameras(*args, vms, Kd, W, H, sh_degree=3, defer_backward=True)
        assert proj.deferred
        for c in range(C):
            r, a, meta = gsa.rasterization(*args, vms[c:c + 1], Kd[None], W, H, sh_degree=3, backgrounds=bg[None].cuda(),
                                           projection=proj[c])
            errs[f"render_cam{c}"] = _rel(r.detach().cpu().double() * keeps[c][None, ..., None], renders64[c])
            (r[0] * (v[c] * keeps[c][..., None]).float().cuda()).sum().backward()
        assert all(t.grad is None for t in leaves)
        bucket = [torch.full_like(t.detach(), float("nan")) for t in args]
        gsa.project_cameras_backward(proj, *args, bucket)
        for n, g in zip(NAMES, bucket):
            errs["grad_" + n] = _rel(g.reshape(want[n].shape), want[n])
        print(f"C2 step of 4 cameras, pass {attempt}:", errs)
        assert all(e <= REL_TOL for e in errs.values()), errs
    _parity_record("parity_c2_step", {
        "config": "C2: 100000 gaussians, 1920x1080, SH degree 3; ONE step of the 4 ring cameras through project_cameras("
                  "defer_backward=True) -> 4 x rasterization(projection=...) + backward -> project_cameras_backward",
        "metric": "max|a - b| / max|b| per tensor; b = the fp64 oracle's image per camera / the SUM of its per-camera gradients",
        "masked_pixel_fraction_per_camera": amb, "tensors": errs})


@pytest.mark.parametrize("cam", [(0, 1), (3, 8)])
def test_config_c3_density_crop_1m(gsa, cam):
    """BASELINE config C3's DENSITY against the oracle: a central crop (shifted principal point -- the construction of
    bench.py's cpu_baseline) of the 1M-gaussian 1080p view, SH degree 3, forward+backward: ~1500 records per tile,
    where the T <= 1e-4 stop, the early-out and the blended-record lists decide (C2 has ~100).  Image and all five
    gradients against the fp64 oracle, 1e-4 outright; the masked pixels within their flip bound.  View 0 and ring camera
    3 of 8 (rotated).  The crop is 384x216 (24 x 14 tiles): the fp64 oracle with its ambiguity bookkeeping needs minutes
    for the 1280x720 crop bench.py times in fp32, and what this test is after -- the list length per tile -- does not
    depend on the crop's size."""
    from gsplat_amd import scene
    W, H, cw, ch = 1920, 1080, 384, 216
    sc = scene.make_scene(1_000_000, seed=0, sh_degree=3)
    K = scene.make_intrinsics(W, H).clone()
    K[0, 2] -= (W - cw) / 2.0
    K[1, 2] -= (H - ch) / 2.0
    m1, m0, errs = _compare(sc, scene.make_camera(*cam), K, cw, ch, 3, bg=torch.zeros(3), use_v_alpha=False, checkpoint=True)
    print("C3 crop (err HIP vs fp64, err fp32-oracle vs fp64):", errs, "M =", m1["flatten_ids"].numel())
    _parity_record(f"parity_c3crop_cam{cam[0]}of{cam[1]}", {
        "config": f"C3 density: central {cw}x{ch} crop (principal point shifted) of the 1000000-gaussian 1920x1080 view, "
                  f"ring camera {cam[0]} of {cam[1]}, SH degree 3, forward+backward",
        "metric": "max|a - b| / max|b| per tensor, b = fp64 oracle; pixels whose threshold comparisons are ambiguous "
                  "in the fp64 oracle are masked on both sides and checked against their flip bound instead",
        "masked_pixel_fraction": errs["ambiguous_fraction"], "masked_pixels": errs["masked_pixels"],
        "intersections_M": int(m1["flatten_ids"].numel()),
        "records_per_tile": int(m1["flatten_ids"].numel()) / (((cw + 15) // 16) * ((ch + 15) // 16)),
        "tensors": {k: {"err_hip": v[0], "err_fp32_oracle": v[1]} for k, v in errs.items() if k not in _ERR_KEYS}})


def test_config_c3_density_large_coordinate_window_1m(gsa):
    """C3's density where the pixel coordinates are LARGE: mean2d is an fp32 pixel coordinate, its ulp is 1.2e-4 beyond
    x = 1024, and the shifted principal point of the crop test above moves every coordinate back to the small ones.  Here
    the frame stays 1920 x 1080 with the true camera and the SCENE is cut instead: the gaussians whose tile rectangle
    touches the window x in [1280, 1664), y in [432, 648) (24 x 14 tiles in the right third of the frame; the cube's
    projection thins out beyond x = 1700) -- the 1M-gaussian scene's density inside the window, an empty frame around it,
    so the fp64 oracle finishes -- forward + backward, image and all five gradients against the fp64 oracle, 1e-4 outright,
    masked pixels within their flip bounds."""
    from gsplat_amd import scene, stages as S
    W, H = 1920, 1080
    x0, x1, y0, y1 = 1280, 1664, 432, 648
    sc = scene.make_scene(1_000_000, seed=0, sh_degree=3)
    vm, K = scene.make_camera(0, 1), scene.make_intrinsics(W, H)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    splat, radii, _, _ = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"],
                                       3, vm.cuda(), K.cuda(), W, H, 0)
    mx, my, rx, ry = splat[:, 0], splat[:, 1], radii[:, 0].float(), radii[:, 1].float()
    # one more tile of margin than the rectangles need: every gaussian that can reach a window pixel is in
    touch = ((radii > 0).all(-1) & (mx + rx > x0 - 16) & (mx - rx < x1 + 16) & (my + ry > y0 - 16) & (my - ry < y1 + 16)).cpu()
    sub = {k: v[touch].contiguous() for k, v in sc.items()}
    n_sub = int(touch.sum())
    assert 5_000 < n_sub < 300_000, n_sub
    del d, splat, radii
    m1, m0, errs = _compare(sub, vm, K, W, H, 3, bg=torch.zeros(3), use_v_alpha=False, checkpoint=True)
    offs = m1["isect_offsets"][0].cpu()
    tw = (W + 15) // 16
    per_tile = torch.diff(torch.cat([offs.flatten(), torch.tensor([int(m1["flatten_ids"].numel())])])).view(-1, tw)
    inside = per_tile[y0 // 16:y1 // 16, x0 // 16:x1 // 16].float().mean()
    print("C3 right-edge window (err HIP vs fp64, err fp32-oracle vs fp64):", errs, "records per window tile =", float(inside))
    assert float(inside) > 500       # C3's density regime (1 673 per tile at the frame's centre, thinner towards the sides)
    _parity_record("parity_c3_large_coordinates", {
        "config": f"C3 density at large pixel coordinates: the {n_sub} of 1000000 gaussians that touch the window "
                  f"x in [{x0}, {x1}), y in [{y0}, {y1}) of the 1920x1080 view 0 (true camera, full frame), SH degree 3, "
                  "forward+backward",
        "metric": "max|a - b| / max|b| per tensor, b = fp64 oracle; masked pixels checked against their flip bound",
        "masked_pixel_fraction_of_the_frame": errs["ambiguous_fraction"], "masked_pixels": errs["masked_pixels"],
        "records_per_window_tile": float(inside), "intersections_M": int(m1["flatten_ids"].numel()),
        "tensors": {k: {"err_hip": v[0], "err_fp32_oracle": v[1]} for k, v in errs.items() if k not in _ERR_KEYS}})


def test_grad_out_writes_into_caller_bucket(gsa):
    """``grad_out``: the backward pass writes straight into caller-provided buffers (one flat all-reduce bucket)
    and autograd hands out those very tensors; values equal the default path up to float-atomic ordering."""
    from gsplat_amd import scene
    W, H, n = 160, 96, 3000
    sc = {k: v.cuda() for k, v in _scene(n, seed=0).items()}
    K, vm = scene.make_intrinsics(W, H).cuda()[None], scene.make_camera(1, 8).cuda()[None]

    def run(bucket):
        leaves = [sc[k].clone().requires_grad_(True) for k in NAMES]
        outs = None
        if bucket is not None:
            outs, off = [], 0
            for t in leaves:
                shape = t.shape if t.shape != (n, 1) else (n,)
                outs.append(bucket[off:off + t.numel()].view(shape))
                off += t.numel()
        r, a, _ = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3][:, 0], leaves[4], vm, K, W, H,
                                    sh_degree=3, grad_out=outs)
        (r * torch.linspace(0.5, 1.5, 3, device="cuda")).sum().backward()
        return [t.grad for t in leaves]

    bucket = torch.full((59 * n,), float("nan"), device="cuda")
    g0, g1 = run(None), run(bucket)
    assert g1[0].data_ptr() == bucket.data_ptr() and not torch.isnan(bucket).any()  # every element written
    for a, b in zip(g0, g1):
        assert _rel(b.reshape(-1), a.reshape(-1)) < 1e-5
    with pytest.raises(ValueError):
        gsa.rasterization(*[sc[k] if k != "opacities" else sc[k][:, 0] for k in NAMES], vm.repeat(2, 1, 1),
                          K.repeat(2, 1, 1), W, H, sh_degree=3, grad_out=[bucket] * 5)


@pytest.mark.parametrize("deg", [None, 1, 3])
def test_grad_accumulate_sums_views_into_one_bucket(gsa, deg):
    """``grad_accumulate``: the backward passes of several views ADD into the caller's buffers (first view overwrites)
    and autograd receives nothing; the bucket equals the sum of the per-view gradients of the default path."""
    from gsplat_amd import scene
    W, H, n = 160, 96, 3000
    sc = {k: v.cuda() for k, v in _scene(n, seed=2).items()}
    colors = sc["shs"] if deg is not None else (sc["shs"][:, 0, :] * 0.28 + 0.5).contiguous()
    K = scene.make_intrinsics(W, H).cuda()[None]
    views = [scene.make_camera(v, 7).cuda()[None] for v in (1, 3, 4)]
    w = torch.linspace(0.5, 1.5, 3, device="cuda")
    base = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), colors]
    ref = [torch.zeros_like(t) for t in base]
    for vm in views:
        leaves = [t.clone().requires_grad_(True) for t in base]
        r, _, _ = gsa.rasterization(*leaves, vm, K, W, H, sh_degree=deg)
        (r * w).sum().backward()
        for acc, t in zip(ref, leaves):
            acc += t.grad
    bucket = [torch.full_like(t, float("nan")) for t in base]
    leaves = [t.clone().requires_grad_(True) for t in base]
    for i, vm in enumerate(views):
        r, _, _ = gsa.rasterization(*leaves, vm, K, W, H, sh_degree=deg, grad_out=bucket, grad_accumulate=i > 0)
        (r * w).sum().backward()
    # the overwriting first view hands autograd the bucket tensors themselves; the accumulating ones hand it nothing
    assert all(t.grad is None or t.grad.data_ptr() == b.data_ptr() for t, b in zip(leaves, bucket))
    for name, a, b in zip(NAMES, bucket, ref):
        assert not torch.isnan(a).any(), name
        assert _rel(a.reshape(-1).cpu(), b.reshape(-1).cpu()) < 1e-5, name
    with pytest.raises(ValueError):
        gsa.rasterization(*base, views[0], K, W, H, sh_degree=deg, grad_accumulate=True)


def test_random_shapes_fused_paths_agree_with_plain_ones(gsa):
    """Twelve seeded random (image size, gaussian count, scale range, camera) combinations -- odd sizes, images smaller
    than a tile, partial border tiles, one- and two-digit tile grids: the fused binning equals emit + sort bit for
    bit, the forward image does not depend on the list option, and the backward sweep over the lists equals the plain
    one up to the order of the float atomics."""
    from gsplat_amd import scene, stages as S
    rng = np.random.default_rng(2024)
    for trial in range(12):
        W, H = int(rng.integers(9, 700)), int(rng.integers(5, 400))
        N = int(rng.integers(50, 20_000))
        s_max = float(10 ** rng.uniform(-2.0, -0.3))
        sc = _scene(N, seed=100 + trial, s_min=s_max * 0.05, s_max=s_max)
        vm, K = scene.make_camera(int(rng.integers(0, 9)), 9).cuda(), scene.make_intrinsics(W, H).cuda()
        d = {k: v.cuda().contiguous() for k, v in sc.items()}
        qb = torch.empty(N, dtype=torch.int32, device="cuda")
        splat, radii, tiles, keys, rects = S.project_fwd(d["means3D"], d["rotations"], d["scales"],
                                                         d["opacities"][:, 0].contiguous(), d["shs"], 3, vm, K, W, H, 0,
                                                         return_rects=True, qbox=qb)
        a = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, fused=False)
        b = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, fused=True, tile_rects=rects, qbox=qb)
        tag = (trial, W, H, N, s_max)
        assert a.finalize()[0] == b.finalize()[0] == int(tiles.sum()), tag
        assert torch.equal(b.sorted_gid, a.sorted_gid) and torch.equal(b.tile_offsets, a.tile_offsets), tag
        assert torch.equal(S.tile_ids(b.sorted_tile), S.tile_ids(a.sorted_tile)), tag
        bg = torch.tensor([0.2, 0.4, 0.6]).cuda()
        gid, offs = b.sorted_gid_buffer, b.tile_offsets
        # (the reference image: the plain ids of the unfused path, every record a candidate for every quadrant)
        r0, a0, T0, l0 = S.composite_fwd(splat, a.sorted_gid_buffer, a.tile_offsets, bg, W, H, 3)
        # with the quadrant boxes the one-pass placement packs into the ids (grids of 257 .. 2047 tiles) a wave gathers
        # only the records whose box covers its quadrant: the same image, bit for bit
        assert b.boxed == bool(gsa.lib.gsa_tile_bin16_boxes(W, H, 16, N)) and not a.boxed, tag
        r1, a1, T1, l1, lists = S.composite_fwd(splat, gid, offs, bg, W, H, 3, S.tile_schedule(offs, W, H), want_lists=True,
                                                gid_boxes=b.boxed)
        assert torch.equal(r0, r1) and torch.equal(T0, T1) and torch.equal(a0, a1), tag
        v_r = torch.from_numpy(rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)).cuda()
        g0 = S.composite_bwd(splat, a.sorted_gid_buffer, a.tile_offsets, bg, T0, l0, v_r, None, W, H, 3)
        g1 = S.composite_bwd(splat, gid, offs, bg, T1, l1, v_r, None, W, H, 3, quad_lists=lists, gid_boxes=b.boxed)
        scale = g0.abs().amax(0).clamp_min(1e-30)
        assert float(((g0 - g1).abs().amax(0) / scale).max()) < 5e-5, tag
        # the same view through 32 x 32 (and 64 x 64) BINS: fewer records sorted, the tiles of a bin share its list -- the
        # image bit for bit, the sweeps (all three list kernels and the plain one) up to the order of the float atomics
        for bt in (32, 64):
            rects_b = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"],
                                    3, vm, K, W, H, 0, return_rects=True, bin_tile=bt)[4]
            bb = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, tile_rects=rects_b, bin_tile=bt, qbox=qb)
            assert bb.bin_tile == bt and bb.finalize()[0] <= int(tiles.sum()), tag
            assert bb.boxed == (bt == 32 and bool(gsa.lib.gsa_tile_bin16_boxes(W, H, bt, N))), (tag, bt)   # (64: 8 x 8 quadrants a bin)
            gb, ob = bb.sorted_gid_buffer, bb.tile_offsets
            rb, ab, Tb, lb, lists_b = S.composite_fwd(splat, gb, ob, bg, W, H, 3, S.tile_schedule(ob, W, H, bt),
                                                      want_lists=True, bin_tile=bt, gid_boxes=bb.boxed)
            assert torch.equal(rb, r0) and torch.equal(Tb, T0) and torch.equal(ab, a0), (tag, bt)
            # the plain sweep (ids masked), the sweep over the forward's lists, and -- boxed -- the same bins WITHOUT the boxes
            sweeps = [S.composite_bwd(splat, gb, ob, bg, Tb, lb, v_r, None, W, H, 3, bin_tile=bt, gid_boxes=bb.boxed),
                      S.composite_bwd(splat, gb, ob, bg, Tb, lb, v_r, None, W, H, 3, quad_lists=lists_b, bin_tile=bt,
                                      gid_boxes=bb.boxed)]
            if bb.boxed:
                plain = bb.sorted_gid.contiguous()
                rp, ap, Tp, lp = S.composite_fwd(splat, plain, ob, bg, W, H, 3, bin_tile=bt)
                assert torch.equal(rp, r0) and torch.equal(Tp, T0) and torch.equal(ap, a0), (tag, bt)
            for gsw in sweeps:
                assert float(((g0 - gsw).abs().amax(0) / scale).max()) < 5e-5, (tag, bt)


@pytest.mark.parametrize("bt", [16, 32, 64])
@pytest.mark.parametrize("N,s_min,s_max", [(6000, 0.004, 0.05), (1500, 0.01, 0.6), (4000, 0.0005, 0.3)])
def test_quadrant_boxes_cover_the_per_pixel_rule(gsa, bt, N, s_min, s_max):
    """The quadrant box the placement pass packs into a record's id (from project_fwd's qbox: the bounding box of the
    gaussian's alpha >= 1/255 ellipse) must cover every quadrant of the record's bin in which ONE pixel centre passes the
    per-pixel alpha test (sigma <= ln(255 o)) -- checked against a brute-force evaluation of sigma at every pixel of every
    record's bin (fp64) -- and the ids must be the plain binning's.  Also reports how tight the boxes are (quadrants
    needed / quadrants covered)."""
    from gsplat_amd import scene, stages as S
    W, H = (653, 401) if bt < 64 else (1301, 1103)   # partial border bins; more than 256 bins (the one-pass placement's grids)
    sc = _scene(N, seed=5, s_min=s_min, s_max=s_max)
    if s_max > 0.5:          # needle-like splats
        sc["scales"][:, 0] *= 0.02
    vm, K = scene.make_camera(1, 5).cuda(), scene.make_intrinsics(W, H).cuda()
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    qb = torch.full((N,), 0x12345678, dtype=torch.int32, device="cuda")
    splat, radii, tiles, keys, rects = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(),
                                                     d["shs"], 3, vm, K, W, H, 0, return_rects=True, bin_tile=bt, qbox=qb)
    assert gsa.lib.gsa_tile_bin16_boxes(W, H, bt, N) == 1      # (a grid the one-pass placement takes)
    bins = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, tile_rects=rects, bin_tile=bt, qbox=qb)
    plain = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, tile_rects=rects, bin_tile=bt)
    m = bins.finalize()[0]
    assert bins.boxed and not plain.boxed and m > 1000
    assert torch.equal(bins.sorted_gid, plain.sorted_gid) and torch.equal(bins.tile_offsets, plain.tile_offsets)
    gid = bins.sorted_gid.long()
    bin_id = S.tile_ids(bins.sorted_tile)
    box = bins.sorted_boxes.long()
    bw = (W + bt - 1) // bt
    sp = splat.double()[gid]                                           # [m,16]
    q = bt // 8                                                        # quadrants per bin side
    ys, xs = torch.meshgrid(torch.arange(bt, device="cuda"), torch.arange(bt, device="cuda"), indexing="ij")
    need = torch.zeros(m, q, q, dtype=torch.bool, device="cuda")
    for lo in range(0, m, 4096):
        hi = min(m, lo + 4096)
        r = sp[lo:hi]
        X0 = (bin_id[lo:hi] % bw * bt).double()[:, None, None]
        Y0 = (bin_id[lo:hi] // bw * bt).double()[:, None, None]
        dx = r[:, 0, None, None] - (X0 + xs[None] + 0.5)
        dy = r[:, 1, None, None] - (Y0 + ys[None] + 0.5)
        sig = 0.5 * (r[:, 2, None, None] * dx * dx + r[:, 4, None, None] * dy * dy) + r[:, 3, None, None] * dx * dy
        ok = (sig >= 0) & (sig <= r[:, 10, None, None])                 # alpha = o exp(-sigma) >= 1/255
        ok &= ((X0 + xs[None]) < W) & ((Y0 + ys[None]) < H)             # (pixels of a border bin beyond the image do not exist)
        need[lo:hi] = ok.view(-1, q, 8, q, 8).any(4).any(2)             # [m', qy, qx]
    cols, rows = box & 15, (box >> 4) & 15       # masks of the column / row units the box covers
    sh = 1 if bt == 64 else 0        # (a 64-pixel bin has 8 x 8 quadrants: its masks count PAIRS of quadrants)
    qi = torch.arange(q, device="cuda") >> sh
    cover = (((cols[:, None, None] >> qi[None, None, :]) & 1) & ((rows[:, None, None] >> qi[None, :, None]) & 1)).bool()   # [m, qy, qx]
    units = q >> sh
    assert int(max(int(cols.max()), int(rows.max()))) < (1 << units)
    for mask in (cols, rows):        # a box: contiguous bits (x & (x + lowest bit) == 0), or none at all (both masks zero)
        low = mask & -mask
        assert bool(((mask & (mask + low)) == 0).all())
    assert bool(((cols == 0) == (rows == 0)).all())
    missing = need & ~cover
    assert int(missing.sum()) == 0, f"{int(missing.any(-1).any(-1).sum())} records miss a quadrant they reach"
    print(f"box tightness (bin {bt}): {int(cover.sum())} quadrants covered for {int(need.sum())} needed, of {m * q * q} possible")
    if s_max <= 0.05 and bt < 64:        # splats of the bench scene's kind: the box is a good stand-in for the ellipse (large slanted
        assert int(cover.sum()) <= 1.6 * int(need.sum()) + 0.03 * m * q * q   # ones: 2 x, reported above, never wrong)


@pytest.mark.parametrize("W,H,N,cull_mode,channels", [(640, 360, 20_000, 0, 3), (250, 130, 3000, 1, 4),
                                                       (1920, 1080, 60_000, 0, 3)])
def test_backward_sweep_over_blended_record_lists(gsa, W, H, N, cull_mode, channels):
    """The forward sweep can note, per 8x8 quadrant, which records it blended (gsa_composite_fwd's quad_list /
    quad_count); gsa_composite_bwd_lists sweeps exactly those.  The forward outputs do not depend on the option (bit
    for bit); the lists hold ascending positions inside the tile's slice, none beyond what a pixel of the quadrant
    saw; the backward result equals the plain sweep's up to the order of the float atomics."""
    from gsplat_amd import scene, stages as S
    big = N == 3000
    sc = _scene(N, seed=31, s_min=0.05 if big else 0.004, s_max=0.5 if big else 0.05)
    vm, K = scene.make_camera(3, 7), scene.make_intrinsics(W, H)
    _, splat, radii, sorted_gid, tile_offsets = _hip_front_end(sc, vm, K, W, H, 3, cull_mode)
    if channels == 4:   # the depth channel rides in the fourth float4 of the staged record
        splat = splat.clone()
    bg = torch.tensor([0.3, 0.6, 0.1, 0.0][:channels]).cuda()
    rng = np.random.default_rng(7)
    v_r = torch.from_numpy(rng.uniform(-1, 1, (H, W, channels)).astype(np.float32)).cuda()
    v_a = torch.from_numpy(rng.uniform(-1, 1, (H, W)).astype(np.float32)).cuda()
    r0, a0, T0, l0 = S.composite_fwd(splat, sorted_gid, tile_offsets, bg, W, H, channels)
    r1, a1, T1, l1, (qlist, qcount) = S.composite_fwd(splat, sorted_gid, tile_offsets, bg, W, H, channels, want_lists=True)
    assert torch.equal(r0, r1) and torch.equal(a0, a1) and torch.equal(T0, T1) and torch.equal(l0, l1)
    tw, th = S.tile_grid(W, H)
    offs = tile_offsets.cpu().numpy()
    qc, ql, last = qcount.cpu().numpy(), qlist.cpu().numpy(), l0.cpu().numpy()
    assert qc.min() >= 0 and 0 < qc.sum() < 4 * sorted_gid.numel()
    for t in np.random.default_rng(1).choice(tw * th, size=40, replace=False):
        s0, e0 = int(offs[t]), int(offs[t + 1])
        ty, tx = divmod(int(t), tw)
        for q in range(4):
            n = int(qc[t, q])
            assert n <= e0 - s0
            lst = ql[4 * s0 + q * (e0 - s0): 4 * s0 + q * (e0 - s0) + n]
            y0, x0 = ty * 16 + (q >> 1) * 8, tx * 16 + (q & 1) * 8
            seen = last[y0:y0 + 8, x0:x0 + 8]
            if n:
                assert np.all(np.diff(lst.astype(np.int64)) > 0) and lst[0] >= s0 and lst[-1] < e0
                assert seen.size and int(lst[-1]) <= int(seen.max())
    g0 = S.composite_bwd(splat, sorted_gid, tile_offsets, bg, T0, l0, v_r, v_a, W, H, channels)
    scale = g0.abs().amax(0).clamp_min(1e-30)
    # the sweep over the lists (the two-phase kernel) against the sweep over the tiles' whole slices
    g1 = S.composite_bwd(splat, sorted_gid, tile_offsets, bg, T0, l0, v_r, v_a, W, H, channels, quad_lists=(qlist, qcount))
    assert float(((g0 - g1).abs().amax(0) / scale).max()) < 2e-5


def test_hot_gaussians_spread_their_atomics_over_replicas(gsa):
    """ABI 14: a camera inside the scene sees splats that cover most of the frame; every (quadrant, record) visit of the reverse
    sweep is a 64-byte atomic on the gaussian's gradient record, and thousands of them on one line serialise.  A gaussian whose
    tile rectangle is >= GSA_HOT_MIN_WIDTH tiles wide claims an entry of the hot region behind the N records and the waves add
    into its replicas (gsa_composite_bwd_lists_hot); the claiming wave marks the gaussian's own record and the projection
    backward -- single-camera and per-step kernels -- adds the replicas when it reads a marked record.  Checked: wide gaussians
    exist and get marked; record + replicas equal the plain sweep's record; both projection backward kernels give the plain
    path's gradients (to the order of the float additions); the whole operator holds the oracle's tolerance on this scene."""
    from gsplat_amd import scene, stages as S
    W, H, N, deg = 800, 592, 4000, 3     # 50 x 37 tiles
    sc = _scene(N, seed=12, s_min=0.02, s_max=0.9)
    K = scene.make_intrinsics(W, H)
    vms = [scene.make_camera(1, 7, radius=0.6), scene.make_camera(4, 7, radius=0.6)]
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    args = (d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"])
    views = S.project_views(*args, deg, torch.stack(vms).cuda(), K.cuda(), W, H, 0, return_aux=True)
    bg = torch.tensor([0.3, 0.6, 0.1]).cuda()
    rng = np.random.default_rng(3)
    hot_rows, entries, replicas = S.HOT_ROWS, 4096, 16
    assert hot_rows == entries * replicas + entries // 16
    records = {True: [], False: []}
    try:
        for hot in (True, False):
            S._HOT_SPLATS = hot
            for c, v in enumerate(views):
                st, sg, to = S.bin_tiles(v.splat, v.radii, v.tiles_touched, v.depth_keys, W, H, 0)
                v_r = torch.from_numpy(np.random.default_rng(10 + c).uniform(-1, 1, (H, W, 3)).astype(np.float32)).cuda()
                v_a = torch.from_numpy(np.random.default_rng(20 + c).uniform(-1, 1, (H, W)).astype(np.float32)).cuda()
                r, a, T, last, lists = S.composite_fwd(v.splat, sg, to, bg, W, H, 3, want_lists=True)
                assert lists.zeroed.shape[0] == N + (hot_rows if hot else 0)
                g = S.composite_bwd(v.splat, sg, to, bg, T, last, v_r, v_a, W, H, 3, quad_lists=lists, fold=False)
                assert g.shape == (N, 16)
                records[hot].append(g)
    finally:
        S._HOT_SPLATS = True
    for c, v in enumerate(views):
        g_hot, g_plain = records[True][c], records[False][c]
        width = (v.splat[:, 11].contiguous().view(torch.int32) >> 20) & 0x3FF
        wide = (width >= 24) & (v.radii > 0).all(-1)
        marks = g_hot[:, 11].contiguous().view(torch.int32)
        assert int(wide.sum()) >= 20, int(wide.sum())
        assert int((marks != 0).sum()) >= 20 and not bool((marks != 0)[~wide].any())   # only wide gaussians claim entries
        assert int(marks.max()) <= entries and float(g_plain[:, 10:].abs().max()) == 0.0
        # record + replicas == the plain sweep's record
        whole = g_hot._base if g_hot._base is not None else g_hot
        assert whole.shape[0] == N + hot_rows
        folded = g_hot.clone()
        idx = torch.nonzero(marks != 0)[:, 0]
        e = (marks[idx] - 1).long()
        rep = whole[N:N + entries * replicas].view(replicas, entries, 16)[:, e].sum(0)   # (replica-major)
        folded[idx] += rep
        tags = whole[N + entries * replicas:].contiguous().view(torch.int32).view(-1)[:entries]
        assert torch.equal(tags[e].long(), idx + 1)                                     # the entries are tagged with their owners
        scale = g_plain[:, :10].abs().amax(0).clamp_min(1e-30)
        assert float(((folded[:, :10] - g_plain[:, :10]).abs().amax(0) / scale).max()) < 2e-5
        # a marked gaussian's own record is NOT the whole sum any more: the replicas carry most of it
        assert float((g_hot[idx, 5] - g_plain[idx, 5]).abs().max()) > 0
        # the stage function's default hands complete records back (what a caller that reads them expects), marks cleared
        done = S.fold_hot_replicas(whole.clone(), N)
        assert float(((done[:, :10] - g_plain[:, :10]).abs().amax(0) / scale).max()) < 2e-5 and float(done[:, 10:].abs().max()) == 0
    # the projection backward folds the replicas: the single-camera kernel ...
    outs = {}
    for hot in (True, False):
        got = None
        for c, v in enumerate(views):
            got = S.project_bwd(*args, deg, v.viewmat, v.Kmat, W, H, v.splat, v.radii, records[hot][c], out=got,
                                accumulate=c > 0, sh_aux=v.sh_aux)
        outs[hot] = got
    # (both paths sum thousands of fp32 atomics per wide gaussian in an order of their own, and the projection backward
    # amplifies that through cancelling terms: two runs of ONE path differ by a few 1e-5 here; the bar against the fp64
    # oracle is at the end)
    for name, a_, b_ in zip(("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"), outs[True], outs[False]):
        assert torch.isfinite(a_).all() and _rel(a_, b_) <= 2e-4, (name, _rel(a_, b_))
    # ... and the per-step kernel
    step = [torch.empty_like(t) for t in outs[False]]
    assert S.project_views_bwd(*args, deg, views, records[True], W, H, step)
    for name, a_, b_ in zip(("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"), step, outs[True]):
        assert _rel(a_, b_) <= 1e-5, (name, _rel(a_, b_))   # (the same records through the other kernel)
    # the operator against the oracle with hot gaussians in the frame (the default path: hot region on) -- a smaller frame and
    # population: the CPU oracle takes minutes on the scene above
    W2, H2 = 448, 272     # 28 x 17 tiles
    sc2 = _scene(1200, seed=13, s_min=0.02, s_max=0.9)
    vm2, K2 = scene.make_camera(2, 7, radius=0.6), scene.make_intrinsics(W2, H2)
    _, splat2, radii2, _, _ = _hip_front_end(sc2, vm2, K2, W2, H2, deg)
    wide2 = (((splat2[:, 11].contiguous().view(torch.int32) >> 20) & 0x3FF) >= 24) & (radii2 > 0).all(-1)
    assert int(wide2.sum()) >= 3, int(wide2.sum())
    _compare(sc2, vm2, K2, W2, H2, deg, bg=torch.tensor([0.3, 0.6, 0.1]))


@pytest.mark.parametrize("W,H,N,s_max,giants", [(640, 360, 30_000, 0.05, 0), (160, 96, 4000, 0.3, 0),
                                                 (1920, 1080, 50_000, 0.02, 5), (40, 24, 500, 0.5, 0),
                                                 (800, 592, 60_000, 0.03, 4)])
def test_fused_binning_equals_emit_and_sort(gsa, W, H, N, s_max, giants):
    """gsa_scan_tiles_rect + gsa_tile_bin16 (records generated inside the first tile-sort pass, 16-bit tile ids) ==
    gsa_tile_emit + gsa_sort_pairs_u32 + gsa_tile_offsets, bit for bit: tile ids, gaussian ids, tile offsets.
    Covers one- and two-digit tile grids and the grids in between that take the ONE-PASS placement (gsa_tile_bin16_direct:
    920 and 1850 tiles here), gaussians without records between ones that have some (the off-screen
    5 %), and a remembered capacity smaller than the record count (reported by finalize(); the caller re-runs)."""
    from gsplat_amd import scene, stages as S
    sc = _scene(N, seed=21, s_min=0.002, s_max=s_max)
    if giants:   # splats whose rectangle is the whole tile grid: one record list spans several 4096-slot blocks
        sc["scales"][7:7 + giants] = 4.0
        sc["means3D"][7:7 + giants] *= 0.2
    vm, K = scene.make_camera(2, 5).cuda(), scene.make_intrinsics(W, H).cuda()
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    splat, radii, tiles, keys = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(),
                                              d["shs"], 3, vm, K, W, H, 0)
    assert int((tiles == 0).sum()) > 0 and int((tiles > 0).sum()) > 0
    grid = S.tile_grid(W, H)[0] * S.tile_grid(W, H)[1]
    direct = bool(gsa.lib.gsa_tile_bin16_direct(W, H, 16))
    assert direct == (256 < grid < 2048)
    if giants:
        assert int(tiles.max()) == grid and (direct or grid > S.BIN_BLOCK)   # the whole grid
    a = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, fused=False)
    b = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, fused=True)
    assert (b._tile is None) == direct                 # (the one-pass placement stores no ids: the offsets imply them)
    assert b.sorted_tile.dtype == (torch.int32 if direct else torch.int16) and a.sorted_tile.dtype == torch.int32
    m = a.finalize()[0]
    assert m == b.finalize()[0] == int(tiles.sum()) and m > 0
    assert torch.equal(S.tile_ids(b.sorted_tile), S.tile_ids(a.sorted_tile))
    assert torch.equal(b.sorted_gid, a.sorted_gid)
    assert torch.equal(b.tile_offsets, a.tile_offsets)
    # steady state with a remembered capacity (device-side count) gives the same bins
    c = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, fused=True)
    assert c.finalize() == (m, True)
    assert torch.equal(c.sorted_gid, a.sorted_gid) and torch.equal(c.tile_offsets, a.tile_offsets)
    real = S._capacity_for
    S._capacity_for = lambda spec, key, n: max(1, m // 3)        # too small: kernels clamp to it, finalize() says so
    try:
        t = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, fused=True)
    finally:
        S._capacity_for = real
    assert t.finalize() == (m, False) and int(t.tile_offsets.max()) <= t.capacity
    e = S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=True, fused=True)
    assert torch.equal(e.sorted_gid, a.sorted_gid) and torch.equal(e.tile_offsets, a.tile_offsets)


@pytest.mark.parametrize("N,W,H,bt", [(200_000, 1920, 1080, 32), (1_250_000, 1920, 1080, 32), (30_000, 640, 480, 16),
                                      (60_000, 1000, 700, 32), (20_000, 500, 300, 16)])
def test_one_pass_placement_equals_the_two_pass_tile_sort(gsa, N, W, H, bt):
    """gsa_tile_bin16 without second-pass buffers (per-bin histogram rows, column scan, every id stored at its final
    place) == the two-pass tile sort + boundary search: sorted_gid and tile_offsets bit for bit, at the bench's bin grid
    (2040 bins; 170 and ~1100 record blocks: one and several rows per column-scan segment, beyond the rows a thread keeps
    in registers), on 16 x 16 tiles, and on frames that are no multiple of the bin (704 and 608 bins); with a remembered
    capacity above the count."""
    from gsplat_amd import scene, stages as S
    sc = scene.make_scene(N, seed=5, sh_degree=0)
    vm, K = scene.make_camera(1, 4).cuda(), scene.make_intrinsics(W, H).cuda()
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    out = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"], 0, vm, K,
                        W, H, 0, return_rects=True, bin_tile=bt)
    splat, radii, tiles, keys, rects = out[:5]
    assert gsa.lib.gsa_tile_bin16_direct(W, H, bt) == 1
    res = {}
    for direct in (False, True):
        real = S._BIN_DIRECT
        S._BIN_DIRECT = direct
        try:
            # (depth_msd=False: the LSD ordering -- 1.25 M gaussians overflow the bucketed one's LDS, which is another test's subject)
            res[direct] = [S.bin_tiles(splat, radii, tiles, keys, W, H, 0, exact=e, fused=True, tile_rects=rects, bin_tile=bt,
                                       depth_msd=False)
                           for e in (True, False)]
        finally:
            S._BIN_DIRECT = real
    m = res[False][0].finalize()[0]
    assert m > (4_300_000 if N > 1_000_000 else 30_000)
    assert res[True][0]._tile is None and res[True][1]._tile is None   # (the one-pass placement stores no tile ids)
    for a, b in zip(res[False], res[True]):
        assert b.finalize() == (m, True) == a.finalize()
        assert torch.equal(b.tile_offsets, a.tile_offsets) and int(b.tile_offsets[-1]) == m
        assert torch.equal(b.sorted_gid, a.sorted_gid)
        assert torch.equal(S.tile_ids(b.sorted_tile), S.tile_ids(a.sorted_tile))
    # a view without a single record, queued for the remembered capacity: every list is empty
    none = rects.clone()
    none[:, 1] = 0
    e = S.bin_tiles(splat, radii, torch.zeros_like(tiles), keys, W, H, 0, fused=True, tile_rects=none, bin_tile=bt,
                    depth_msd=False)
    assert e.capacity >= m and e.finalize() == (0, True) and int(e.tile_offsets.abs().max()) == 0


@pytest.mark.parametrize("deg", [3, 1])
def test_factored_sh_gradient_rebuilds_the_coefficient_gradient(gsa, deg):
    """``sh_grad_factored``: the backward pass leaves d loss / d rgb [N,3] per view; ``stages.sh_grad_expand`` over the
    stacked views rebuilds the sum of the per-view [N,K,3] coefficient gradients of the default path (the form the
    view-parallel ranks exchange: all-gather of 3 N floats instead of all-reducing 48 N).  The geometry gradients
    (incl. the view-direction term of the mean) are those of the default path."""
    from gsplat_amd import scene, stages
    W, H, n = 160, 96, 3000
    sc = {k: v.cuda() for k, v in _scene(n, seed=4).items()}
    K = scene.make_intrinsics(W, H).cuda()[None]
    views = [scene.make_camera(v, 7).cuda()[None] for v in (0, 2, 5)]
    w = torch.linspace(0.5, 1.5, 3, device="cuda")
    base = [sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"][:, 0].contiguous(), sc["shs"]]
    ref = [torch.zeros_like(t) for t in base]
    for vm in views:
        leaves = [t.clone().requires_grad_(True) for t in base]
        r, _, _ = gsa.rasterization(*leaves, vm, K, W, H, sh_degree=deg)
        (r * w).sum().backward()
        for acc, t in zip(ref, leaves):
            acc += t.grad
    bucket = [torch.full_like(t, float("nan")) for t in base]
    vr_all = torch.full((3, n, 3), float("nan"), device="cuda")
    leaves = [t.clone().requires_grad_(True) for t in base]
    for i, vm in enumerate(views):
        r, _, _ = gsa.rasterization(*leaves, vm, K, W, H, sh_degree=deg, grad_out=bucket, grad_accumulate=i > 0,
                                    sh_grad_factored=vr_all[i])
        (r * w).sum().backward()
    assert leaves[4].grad is None and torch.isnan(bucket[4]).all()        # the coefficient gradient is not written
    assert not torch.isnan(vr_all).any()
    stages.sh_grad_expand(leaves[0].detach(), deg, torch.cat(views), K, vr_all, bucket[4])
    for name, a, b in zip(NAMES, bucket, ref):
        assert not torch.isnan(a).any(), name
        assert _rel(a.reshape(-1).cpu(), b.reshape(-1).cpu()) < 1e-5, name
    # accumulate=True adds on top; coefficients above the active degree stay zero
    twice = bucket[4].clone()
    stages.sh_grad_expand(leaves[0].detach(), deg, torch.cat(views), K, vr_all, twice, accumulate=True)
    assert _rel(twice.cpu(), 2.0 * ref[4].cpu()) < 1e-5
    assert float(bucket[4][:, (deg + 1) ** 2:, :].abs().max()) == 0.0 if deg < 3 else True
    with pytest.raises(ValueError):
        gsa.rasterization(*base, views[0], K, W, H, sh_degree=deg, sh_grad_factored=vr_all[0])   # needs grad_out


def test_full_size_properties_c3_1m_1080p(gsa):
    """BASELINE config C3 size (1M gaussians, 1920x1080, degree 3), where the CPU oracle is out of reach: properties
    that hold at any size.  (a) the forward image is bit-reproducible; (b) the backward pass is linear in the
    upstream gradient; (c) gaussians the contract culls (opacity below 1/255, behind the camera) change nothing,
    wherever they sit in the arrays; (d) alpha = 1 - T stays in [0, 1] and every gradient is finite; (e) the record
    list is sorted by (tile, depth) and every tile's slice matches its offsets."""
    from gsplat_amd import scene
    N, W, H = 1_000_000, 1920, 1080
    sc = scene.make_scene(N, seed=0, sh_degree=3)      # the bench scene, view 0
    vm, K = scene.make_camera(0, 1)[None].cuda(), scene.make_intrinsics(W, H)[None].cuda()
    v = scene.make_v_render(W, H, seed=2).cuda()[None]

    def run(leaves_in, upstream=None):
        leaves = [t.clone().requires_grad_(upstream is not None) for t in leaves_in]
        r, a, meta = gsa.rasterization(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], vm, K, W, H, sh_degree=3,
                                       backgrounds=torch.zeros(1, 3, device="cuda"))
        if upstream is not None:
            r.backward(upstream)
        return r.detach(), a.detach(), meta, [t.grad for t in leaves]

    base = [sc["means3D"].cuda(), sc["rotations"].cuda(), sc["scales"].cuda(), sc["opacities"][:, 0].cuda(), sc["shs"].cuda()]
    r0, a0, m0, g1 = run(base, v)
    r1, a1, _, g2 = run(base, 2.0 * v)
    assert torch.equal(r0, r1) and torch.equal(a0, a1)                                   # (a)
    for name, x, y in zip(NAMES, g1, g2):                                                # (b) atomics reorder sums
        assert torch.isfinite(x).all() and torch.isfinite(y).all(), name                 # (d)
        assert _rel(2.0 * x.cpu(), y.cpu()) < 1e-4, name
    assert float(a0.min()) >= 0.0 and float(a0.max()) <= 1.0                             # (d)
    # (c) 1000 culled gaussians spliced in at random positions (half invisible by opacity, half behind the camera)
    g = torch.Generator().manual_seed(9)
    extra = 1000
    pos = torch.randperm(N + extra, generator=g)[:extra].sort().values
    keep = torch.ones(N + extra, dtype=torch.bool)
    keep[pos] = False
    spliced = []
    for t in base:
        big = torch.empty(N + extra, *t.shape[1:], device="cuda")
        big[keep.cuda()] = t
        big[pos.cuda()] = t[:extra]
        spliced.append(big)
    spliced[3][pos[:extra // 2].cuda()] = 1.0 / 512.0                                    # alpha can never reach 1/255
    spliced[0][pos[extra // 2:].cuda(), 2] = -10.0                                       # camera-space z = -6: behind it
    r2, a2, m2, _ = run(spliced)
    assert torch.equal(r2, r0) and torch.equal(a2, a0)
    assert m2["n_isects"] == m0["n_isects"]
    # (e) sortedness of the binned records
    ids = m0["isect_ids"]
    assert bool((ids[1:] >= ids[:-1]).all())
    offs = m0["isect_offsets"].reshape(-1).long()
    tiles = (ids >> 32)
    first = torch.searchsorted(tiles, torch.arange(offs.numel(), device=tiles.device))
    assert torch.equal(first, offs)


@pytest.mark.parametrize("n,dist", [(200_000, "uniform"), (1_000_003, "bench"), (5000, "uniform"), (70_000, "ties"),
                                    (300, "uniform"), (50_000, "all_culled"), (400_000, "wide"), (1_200_000, "wide")])
def test_bucketed_depth_order_equals_lsd_sort(gsa, n, dist):
    """gsa_depth_order_msd (one global bucket pass + a local LDS sort per bucket fused with the scan) == gsa_depth_order
    + gsa_scan_tiles_sorted (LSD passes, separate scan), bit for bit, on every entry that belongs to a visible gaussian:
    order, rectangles, offsets, block owners, record count, key range; ties keep index order."""
    from gsplat_amd import stages as S
    from gsplat_amd._lib import check, lib
    g = torch.Generator().manual_seed(n)
    if dist == "bench":      # float bits of depths in [3, 5]: twice as dense above 4.0
        keys = (torch.rand(n, generator=g) * 2 + 3).to(torch.float32).view(torch.int32).long()
    elif dist == "wide":     # depths log-uniform over 0.05 .. 500 (13 octaves, 112 M keys: buckets ~440 k keys wide -> the
        z = torch.exp(torch.rand(n, generator=g, dtype=torch.float64) * math.log(1e4) + math.log(0.05))   # 24-bit local sorts)
        keys = z.to(torch.float32).view(torch.int32).long()
    elif dist == "ties":     # many equal keys: stability
        keys = torch.randint(0x40400000, 0x40400000 + 3000, (n,), generator=g)
    else:
        keys = torch.randint(0x40000000, 0x40C00000, (n,), generator=g)
    culled = torch.rand(n, generator=g) < (1.0 if dist == "all_culled" else 0.08)
    keys[culled] = 0xFFFFFFFF
    counts = torch.randint(0, 20, (n,), generator=g)
    counts[culled] = 0
    counts[torch.rand(n, generator=g) < 0.05] = 0          # visible gaussians without records
    if n > 100_000:
        counts[12345] = 9000                               # one record list spanning several 4096-slot blocks
    rect_words = torch.randint(1, 2 ** 30, (n,), generator=g)
    pairs = torch.stack([rect_words, counts], dim=1).to(torch.int32).contiguous().cuda()
    k32 = torch.where(keys >= 2 ** 31, keys - 2 ** 32, keys).to(torch.int32).cuda()
    m_true = int(counts.sum())
    n_blocks = (m_true + S.BIN_BLOCK - 1) // S.BIN_BLOCK + 3
    st = S._stream()

    def buffers():
        return (torch.full((n,), -1, dtype=torch.int32, device="cuda"), torch.full((n,), -1, dtype=torch.int32, device="cuda"),
                torch.full((n,), -1, dtype=torch.int32, device="cuda"), torch.full((n_blocks + 1,), -7, dtype=torch.int32, device="cuda"),
                torch.full((4,), -1, dtype=torch.int32, device="cuda"))

    # reference: LSD over all 32 bits + scan
    o0, r0, f0, b0, c0 = buffers()
    ws = torch.empty(lib.gsa_depth_order_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    check(lib.gsa_depth_order(n, k32.data_ptr(), pairs.data_ptr(), 32, o0.data_ptr(), r0.data_ptr(), f0.data_ptr(),
                              c0[1:3].data_ptr(), ws.data_ptr(), ws.numel(), st), "gsa_depth_order")
    ws2 = torch.empty(lib.gsa_sort_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    check(lib.gsa_scan_tiles_sorted(n, f0.data_ptr(), b0.data_ptr(), n_blocks, c0[0:1].data_ptr(), ws2.data_ptr(), ws2.numel(), st),
          "gsa_scan_tiles_sorted")
    n_vis = int((~culled).sum())
    vis_keys = keys[~culled]
    lo, hi = (int(vis_keys.min()), int(vis_keys.max())) if n_vis else (0xFFFFFFFF, 0)
    assert (int(c0[0]) & 0xFFFFFFFF, int(c0[1]) & 0xFFFFFFFF, int(c0[2]) & 0xFFFFFFFF) == (m_true, lo, hi)
    ref_order = torch.sort(keys, stable=True).indices
    assert torch.equal(o0.cpu().long()[:n_vis], ref_order[:n_vis])
    # bucketed ordering for the range a caller would have remembered
    plan = S._msd_plan(lo, hi) if n_vis else (0x3F000000, 1 << 16)
    assert plan is not None
    o1, r1, f1, b1, c1 = buffers()
    ws = torch.empty(lib.gsa_depth_order_msd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    # (the control words also go to a pinned host slot, written by the kernel itself and tagged with the caller's token)
    host = torch.zeros(4, dtype=torch.int32, pin_memory=True)
    token = 0xABCDE
    check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), plan[0], plan[1], None, o1.data_ptr(), r1.data_ptr(),
                                  f1.data_ptr(), b1.data_ptr(), n_blocks, c1.data_ptr(), host.data_ptr(), token, ws.data_ptr(),
                                  ws.numel(), st),
          "gsa_depth_order_msd")
    S._wait_slot((host, None, token))                     # polls word 3 for the token: no event, no copy
    assert [int(x) & 0xFFFFFFFF for x in host] == [m_true, lo, hi, token << 8]
    torch.cuda.synchronize()
    assert [int(x) & 0xFFFFFFFF for x in c1] == [m_true, lo, hi, 0]
    assert torch.equal(o1[:n_vis], o0[:n_vis]) and torch.equal(r1[:n_vis], r0[:n_vis]) and torch.equal(f1[:n_vis], f0[:n_vis])
    live_blocks = (m_true + S.BIN_BLOCK - 1) // S.BIN_BLOCK
    assert torch.equal(b1[:live_blocks], b0[:live_blocks])
    if m_true:
        assert int(b1[n_blocks]) == int(b0[n_blocks])
    # the same through the view's own DEPTH TABLE, built on the device from what gsa_project_fwd leaves per wave of 64
    # gaussians -- (min, max) of the visible keys and two of the keys (a sample): the order must not depend on the table
    def per_wave(kk):
        pad = torch.full(((-n) % 64,), 0xFFFFFFFF, dtype=torch.long)
        kw = torch.cat([kk, pad]).view(-1, 64)
        vis_w = kw != 0xFFFFFFFF
        wlo = torch.where(vis_w, kw, torch.full_like(kw, 0xFFFFFFFF)).amin(1)
        whi = torch.where(vis_w, kw, torch.zeros_like(kw)).amax(1)
        out = torch.stack([wlo, whi, kw[:, 0], kw[:, 32]], dim=1)
        return torch.where(out >= 2 ** 31, out - 2 ** 32, out).to(torch.int32).contiguous().cuda()

    kr = per_wave(keys)
    o2, r2, f2, b2, c2 = buffers()
    check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), 0, 0, kr.data_ptr(), o2.data_ptr(), r2.data_ptr(),
                                  f2.data_ptr(), b2.data_ptr(), n_blocks, c2.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st),
          "gsa_depth_order_msd(key_range_dev)")
    torch.cuda.synchronize()
    assert [int(x) & 0xFFFFFFFF for x in c2] == [m_true, lo, hi, 0]
    assert torch.equal(o2[:n_vis], o0[:n_vis]) and torch.equal(r2[:n_vis], r0[:n_vis]) and torch.equal(f2[:n_vis], f0[:n_vis])
    assert torch.equal(b2[:live_blocks], b0[:live_blocks])
    if m_true:
        assert int(b2[n_blocks]) == int(b0[n_blocks])


@pytest.mark.parametrize("n", [5_000_000, 10_000_000])
def test_depth_order_at_the_reference_population(gsa, n):
    """The depth ordering at the population the reference trains at ("~5.5M gaussians", /root/reference/src/training/trainer.py:92-96;
    cap 1e7, training_config.py:22): beyond ~4 M gaussians the LSD passes scatter through the 22-KB instance
    (``radix_scatter_compact_kernel``: more blocks than the 38-KB instance has slots on the chip) and the scan reports the
    control words into the caller's pinned slot itself (``gsa_scan_tiles_sorted_report``).  Order == a stable sort of the
    keys (ties keep index order), rectangles and counts gathered in that order, offsets == their exclusive sum, count and key
    range reported on both paths, bit for bit; 24 key bits (what these depths differ in) == 32."""
    from gsplat_amd import stages as S
    from gsplat_amd._lib import check, lib
    g = torch.Generator().manual_seed(n % 1013)
    keys = (torch.rand(n, generator=g) * 2 + 3).to(torch.float32).view(torch.int32).long()   # depths in [3, 5]
    culled = torch.rand(n, generator=g) < 0.08
    keys[culled] = 0xFFFFFFFF
    counts = torch.randint(0, 6, (n,), generator=g)
    counts[culled] = 0
    counts[777_777] = 9000
    rect_words = torch.randint(1, 2 ** 30, (n,), generator=g)
    pairs = torch.stack([rect_words, counts], dim=1).to(torch.int32).contiguous().cuda()
    k32 = torch.where(keys >= 2 ** 31, keys - 2 ** 32, keys).to(torch.int32).cuda()
    m_true, n_vis = int(counts.sum()), int((~culled).sum())
    lo, hi = int(keys[~culled].min()), int(keys[~culled].max())
    assert S._key_bits_needed(lo, hi) == 24
    n_blocks = (m_true + S.BIN_BLOCK - 1) // S.BIN_BLOCK
    st = S._stream()
    ref = torch.sort(keys.cuda(), stable=True).indices[:n_vis]
    ref_counts = counts.cuda()[ref]
    ref_offsets = torch.cumsum(ref_counts, 0) - ref_counts
    ws = torch.empty(lib.gsa_depth_order_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ws2 = torch.empty(lib.gsa_sort_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    for bits in (24, 32):
        order, rects, offs = (torch.full((n,), -1, dtype=torch.int32, device="cuda") for _ in range(3))
        owner = torch.full((n_blocks + 1,), -7, dtype=torch.int32, device="cuda")
        ctl = torch.full((4,), -1, dtype=torch.int32, device="cuda")
        host = torch.zeros(4, dtype=torch.int32, pin_memory=True)
        token = 0x5A5A5 + bits
        check(lib.gsa_depth_order(n, k32.data_ptr(), pairs.data_ptr(), bits, order.data_ptr(), rects.data_ptr(), offs.data_ptr(),
                                  ctl[1:3].data_ptr(), ws.data_ptr(), ws.numel(), st), "gsa_depth_order")
        check(lib.gsa_scan_tiles_sorted_report(n, offs.data_ptr(), owner.data_ptr(), n_blocks, ctl.data_ptr(), host.data_ptr(), token,
                                               ws2.data_ptr(), ws2.numel(), st), "gsa_scan_tiles_sorted_report")
        S._wait_slot((host, None, token))
        assert [int(x) & 0xFFFFFFFF for x in host] == [m_true, lo, hi, token << 8]
        torch.cuda.synchronize()
        assert [int(x) & 0xFFFFFFFF for x in ctl[:3]] == [m_true, lo, hi]
        assert torch.equal(order[:n_vis].long(), ref)
        assert torch.equal(rects[:n_vis], pairs[:, 0][ref]) and torch.equal(offs[:n_vis].long(), ref_offsets)
        # the owner of every 4096-slot record block: the last gaussian (in depth order) whose list starts at or before the block
        starts = torch.arange(n_blocks, device="cuda") * S.BIN_BLOCK
        has = ref_counts > 0
        idx = torch.nonzero(has).squeeze(1)
        want_owner = idx[torch.searchsorted(ref_offsets[has].contiguous(), starts, right=True) - 1]
        assert torch.equal(owner[:n_blocks].long(), want_owner) and int(owner[n_blocks]) == int(idx[-1])


def test_depth_table_levels_the_buckets(gsa):
    """The table the bucketed ordering builds for a view (cells -> buckets, from the per-wave sample of keys): valid --
    monotone, every bucket at most 2^24 keys wide -- and LEVEL: on keys as unevenly distributed as the diagonal cameras of
    the bench ring see them (3.5 x the mean in the fullest of 255 equal buckets, more than the local sorts hold) the fullest
    bucket stays well inside the 8192-key capacity, and the ordering reports no overflow."""
    from gsplat_amd import scene, stages as S
    from gsplat_amd._lib import check, lib
    N, W, H = 1_000_000, 1920, 1080
    sc = scene.make_scene(N, seed=0, sh_degree=3)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    K = scene.make_intrinsics(W, H).cuda()
    for view in (1, 0):        # 45 degrees: the cube seen along a diagonal; 0: along an axis
        vm = scene.make_camera(view, 8).cuda()
        kr = S.key_range_buffer(N, d["means3D"].device)
        splat, radii, tiles, keys, rects = S.project_fwd(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(),
                                                         d["shs"], 3, vm, K, W, H, 0, return_rects=True, bin_tile=32, key_range=kr)
        st = S._stream()
        bufs = [torch.empty(N, dtype=torch.int32, device="cuda") for _ in range(3)]
        owner = torch.empty(2000, dtype=torch.int32, device="cuda")
        ctl = torch.empty(4, dtype=torch.int32, device="cuda")
        ws = torch.zeros(lib.gsa_depth_order_msd_workspace_bytes(N), dtype=torch.uint8, device="cuda")
        check(lib.gsa_depth_order_msd(N, keys.data_ptr(), rects.data_ptr(), 0, 0, kr.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(),
                                      bufs[2].data_ptr(), owner.data_ptr(), 1999, ctl.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st),
              "gsa_depth_order_msd")
        torch.cuda.synchronize()
        assert int(ctl[3]) == 0, f"view {view}: status {int(ctl[3])}"
        k = keys.long() & 0xFFFFFFFF
        k = k[k != 0xFFFFFFFF]
        lo, hi = int(k.min()), int(k.max())
        assert (int(ctl[1]) & 0xFFFFFFFF, int(ctl[2]) & 0xFFFFFFFF) == (lo, hi)
        # the order is the stable order by key
        n_vis = k.numel()
        ref = torch.sort(keys.long() & 0xFFFFFFFF, stable=True).indices
        assert torch.equal(bufs[0][:n_vis].long(), ref[:n_vis])
        # equal buckets for comparison (the host-side plan of r02)
        a, scale = S._msd_plan(lo, hi)
        fullest_equal = int(torch.bincount((((k - a) * scale) >> 32).clamp(max=254), minlength=255).max())
        # the table the first launch built and stored inside the workspace
        off = lib.gsa_depth_order_msd_table_offset(N)
        tab = ws[off:off + 4 * lib.gsa_depth_order_msd_table_words()].view(torch.int32).cpu()
        cells, shift, g0 = int(tab[1286]), int(tab[1285]), int(tab[1284]) & 0xFFFFFFFF
        lut = tab[:1024].contiguous().view(torch.uint8)[:cells].long()
        k = k.cpu()
        cs = tab[1024:1024 + 257].long()
        assert 0 < cells <= 4096 and (hi >> shift) - g0 + 1 == cells and lo >> shift == g0
        assert int(lut[0]) == 0 and bool((lut[1:] >= lut[:-1]).all()) and int(lut.max()) <= 254
        for v in range(int(lut.max()) + 1):
            assert bool((lut[int(cs[v]):int(cs[v + 1])] == v).all()) and (int(cs[v + 1]) - int(cs[v])) << shift <= lib.gsa_depth_order_msd_max_width(), v
        assert int(cs[int(lut.max()) + 1]) == cells
        fullest = int(torch.bincount(lut[(k >> shift) - g0], minlength=255).max())
        print(f"view {view}: {n_vis} visible keys, {cells} cells of 2^{shift} keys, {int(lut.max()) + 1} buckets; fullest bucket "
              f"{fullest} (255 equal buckets: {fullest_equal}), mean {n_vis / 254:.0f}")
        assert fullest < 0.8 * 8192 and fullest < 0.85 * fullest_equal


def test_bucketed_depth_order_reports_what_it_cannot_do(gsa):
    """A bucket beyond the LDS capacity (a depth cluster) and keys outside the assumed range are REPORTED through the
    control words -- the caller (stages.bin_tiles) then repeats with the LSD path -- and nothing hangs or faults."""
    from gsplat_amd import stages as S
    from gsplat_amd._lib import check, lib
    n = 60_000
    g = torch.Generator().manual_seed(1)
    keys = torch.randint(0x40000000, 0x40800000, (n,), generator=g)
    keys[:20_000] = 0x40400000 + torch.randint(0, 50, (20_000,), generator=g)     # 20 k keys in one bucket
    pairs = torch.stack([torch.ones(n, dtype=torch.int64), torch.ones(n, dtype=torch.int64)], dim=1).to(torch.int32).cuda()
    k32 = keys.to(torch.int32).cuda()
    st = S._stream()
    out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(3)]
    owner = torch.empty(64, dtype=torch.int32, device="cuda")
    ctl = torch.empty(4, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.gsa_depth_order_msd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    plan = S._msd_plan(int(keys.min()), int(keys.max()))
    check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), plan[0], plan[1], None, out[0].data_ptr(), out[1].data_ptr(),
                                  out[2].data_ptr(), owner.data_ptr(), 63, ctl.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st), "msd")
    assert int(ctl[3]) & 1, "bucket overflow must be reported"
    assert not S._note_keys(S.DEFAULT_SPECULATION, ("t", 0), int(ctl[1]), int(ctl[2]), plan, int(ctl[3]))
    # keys below the assumed minimum: the measured range tells
    narrow = (0x40400000, 1 << 24)   # (buckets 256 keys wide from there)
    check(lib.gsa_depth_order_msd(n, k32.data_ptr(), pairs.data_ptr(), narrow[0], narrow[1], None, out[0].data_ptr(), out[1].data_ptr(),
                                  out[2].data_ptr(), owner.data_ptr(), 63, ctl.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st), "msd")
    assert not S._note_keys(S.DEFAULT_SPECULATION, ("t", 1), int(ctl[1]), int(ctl[2]), narrow, int(ctl[3]))
    # (r03 also reported ranges beyond two octaves of depth -- status bit 2; with 24-bit local sorts every range of positive
    # float depths fits, see the "wide" case of test_bucketed_depth_order_equals_lsd_sort)
    S.DEFAULT_SPECULATION.msd_backoff.pop(("t", 2), None)
    S.DEFAULT_SPECULATION.key_range.pop(("t", 2), None)
    S.DEFAULT_SPECULATION.key_range.pop(("t", 0), None)
    S.DEFAULT_SPECULATION.key_range.pop(("t", 1), None)


def test_views_equal_buckets_do_not_fit_move_the_shape_to_depth_tables(gsa):
    """The bench ring at 1 M gaussians: the axis-aligned cameras fit 255 equal depth buckets (planned on the host from the
    remembered key range: the fast path), the diagonal ones put 12.6 k keys into one bucket.  The first such view is
    repaired once; from then on the shape's views run on their own device-built depth tables -- no repair, no LSD
    fallback -- and the images are those of the exact (LSD) path."""
    from gsplat_amd import scene, stages as S
    N, W, H = 1_000_000, 1920, 1080
    sc = scene.make_scene(N, seed=0, sh_degree=3)
    d = {k: v.cuda().contiguous() for k, v in sc.items()}
    K = scene.make_intrinsics(W, H).cuda()[None]
    key = ("cuda", torch.cuda.current_device(), W, H, 0, S.bin_tile_for(W, H, 0))
    for t in (S.DEFAULT_SPECULATION.capacity, S.DEFAULT_SPECULATION.key_range, S.DEFAULT_SPECULATION.msd_backoff, S.DEFAULT_SPECULATION.table_mode, S.DEFAULT_SPECULATION.msd_streak, S.DEFAULT_SPECULATION.host_plan_backoff):
        t.pop(key, None)
    for k in S.REPAIR_STATS:
        S.REPAIR_STATS[k] = 0

    def render(v):
        with torch.no_grad():
            r, a, meta = gsa.rasterization(d["means3D"], d["rotations"], d["scales"], d["opacities"][:, 0].contiguous(), d["shs"],
                                           scene.make_camera(v, 8).cuda()[None], K, W, H, sh_degree=3)
        return r

    first = [render(v) for v in (0, 0, 2, 1)]        # first call (exact), host-planned, host-planned, the diagonal view: repaired
    assert S.REPAIR_STATS["key_range_repairs"] == 1 and S.DEFAULT_SPECULATION.table_mode.get(key, 0) > 0, (S.REPAIR_STATS, S.DEFAULT_SPECULATION.table_mode)
    again = [render(v) for v in (1, 0, 3, 2, 5, 4, 7, 6)]
    assert S.REPAIR_STATS["key_range_repairs"] == 1 and S.REPAIR_STATS["capacity_misses"] <= 1, S.REPAIR_STATS
    assert torch.equal(again[0], first[3]) and torch.equal(again[1], first[0]) and torch.equal(again[3], first[2])
    for t in (S.DEFAULT_SPECULATION.capacity, S.DEFAULT_SPECULATION.key_range, S.DEFAULT_SPECULATION.msd_backoff, S.DEFAULT_SPECULATION.table_mode, S.DEFAULT_SPECULATION.msd_streak, S.DEFAULT_SPECULATION.host_plan_backoff):
        t.pop(key, None)


def test_depth_cluster_falls_back_to_the_lsd_path(gsa):
    """``stages.bin_tiles`` end to end on a population with a depth cluster (20 k of 100 k keys in one bucket): the
    bucketed ordering reports the overflow, ``finalize()`` says so, the exact repeat gives the first call's bins, and
    the shape then stays on the LSD path for a while instead of paying a repair per call."""
    from gsplat_amd import stages as S
    W, H, N = 640, 368, 100_000
    tw, th = S.tile_grid(W, H)
    g = torch.Generator().manual_seed(4)
    keys = torch.randint(0x40000000, 0x40800000, (N,), generator=g)
    keys[:20_000] = 0x40400000 + torch.randint(0, 64, (20_000,), generator=g)
    x0 = torch.randint(0, tw - 2, (N,), generator=g)
    y0 = torch.randint(0, th - 2, (N,), generator=g)
    rect = (x0 | (y0 << 10) | (2 << 20)).to(torch.int32)                     # 2 x 2 tiles each
    counts = torch.full((N,), 4, dtype=torch.int32)
    pairs = torch.stack([rect, counts], dim=1).contiguous().cuda()
    splat = torch.zeros(N, 16, device="cuda")
    radii = torch.ones(N, 2, dtype=torch.int32, device="cuda")
    k32 = keys.to(torch.int32).cuda()
    key = ("cuda", torch.cuda.current_device(), W, H, 0, 16)
    for d in (S.DEFAULT_SPECULATION.capacity, S.DEFAULT_SPECULATION.key_range, S.DEFAULT_SPECULATION.msd_backoff, S.DEFAULT_SPECULATION.table_mode, S.DEFAULT_SPECULATION.msd_streak, S.DEFAULT_SPECULATION.host_plan_backoff):
        d.pop(key, None)

    def bins(**kw):
        return S.bin_tiles(splat, radii, counts.cuda(), k32, W, H, 0, tile_rects=pairs, **kw)

    first = bins()                                     # first call of the shape: LSD, exact sizes, learns the range
    m, ok = first.finalize()
    assert ok and m == 4 * N and first.key_bits == 32
    ref_gid, ref_off = first.sorted_gid.clone(), first.tile_offsets.clone()
    # steady state: bucketed ordering for the remembered range.  The buffers it fills come out of the caching allocator
    # un-initialised; here the blocks it is about to recycle hold 0xFF bytes, and -- as rasterization() and the harness
    # do -- the compositor is queued on the bins BEFORE the host has looked at the report: every entry the overflowing
    # bucket could not sort must still be defined (no gaussian id beyond N, no record list outside the buffers).
    junk = [torch.full((sz,), -1, dtype=torch.int32, device="cuda") for sz in (N, N, N, 4 * N, 4 * N, 2 * N, 2 * N, N // 2)]
    del junk
    second = bins()
    assert isinstance(second.key_bits, tuple)
    order = S.tile_schedule(second.tile_offsets, W, H)
    out = S.composite_fwd(splat, second.sorted_gid_buffer, second.tile_offsets, None, W, H, 3, order, want_lists=True)
    torch.cuda.synchronize()                           # (a fault would surface here)
    assert float(out[0].abs().max()) == 0.0
    m2, ok2 = second.finalize()
    assert ok2 is False, "the overflowing bucket must surface through finalize()"   # (its M counts the sorted part only)
    live = second.sorted_gid_buffer[:min(m2, second.capacity)].long() & 0xFFFFFFFF
    assert m2 < m and int(live.max()) < N
    repair = bins(exact=True)
    assert repair.finalize() == (m, True)
    assert torch.equal(repair.sorted_gid, ref_gid) and torch.equal(repair.tile_offsets, ref_off)
    third = bins()                                     # backed off: LSD on the bits the keys differ in
    assert not isinstance(third.key_bits, tuple) and third.finalize() == (m, True)
    assert torch.equal(third.sorted_gid, ref_gid) and torch.equal(third.tile_offsets, ref_off)
    for d in (S.DEFAULT_SPECULATION.capacity, S.DEFAULT_SPECULATION.key_range, S.DEFAULT_SPECULATION.msd_backoff, S.DEFAULT_SPECULATION.table_mode, S.DEFAULT_SPECULATION.msd_streak, S.DEFAULT_SPECULATION.host_plan_backoff):
        d.pop(key, None)
