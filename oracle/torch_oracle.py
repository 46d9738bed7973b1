"""Pure-PyTorch CPU restatement of the rasterizer the reference calls.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py (incl. the "parity unpinned" statement).

What it restates
----------------
The one call the reference makes on its hot path,
``gsplat.rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
packed=False, render_mode=..., sh_degree=..., backgrounds=...)``
(/root/reference/src/backends/gsplat_backend.py:83-97 and :168-182), with every argument the
reference leaves unset at the gsplat 1.5.3 default.  gsplat itself is an un-vendored pip
dependency (/root/reference/requirements.txt:9), so the arithmetic below follows the published
algorithm as restated in SURVEY.md Appendix A (sections A.1-A.7), cited per function.

Everything differentiable is written with plain torch ops so that ``torch.autograd`` is the
gradient oracle (A.6/A.7 are not hand-derived here).  Decisions that the published algorithm
treats as non-differentiable (culling, radii, the alpha < 1/255 skip, the T <= 1e-4 stop) are
taken under ``torch.no_grad`` and applied as masks.  Works in float32 (the parity dtype) and in
float64 (for ``torch.autograd.gradcheck``).
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.utils.checkpoint

from . import constants as K_

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


# --------------------------------------------------------------------------------------
# A.2  projection + EWA
# --------------------------------------------------------------------------------------
def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """(w, x, y, z) -> rotation matrix, normalising inside (Appendix A.1; quaternion order per
    /root/reference/src/gaussians/gaussian_model.py:82-84)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    rows = [
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ]
    return torch.stack(rows, dim=-1).reshape(quats.shape[:-1] + (3, 3))


def project(
    means: torch.Tensor,      # [N,3]
    quats: torch.Tensor,      # [N,4]
    scales: torch.Tensor,     # [N,3]
    opacities: torch.Tensor,  # [N]
    viewmat: torch.Tensor,    # [4,4] world->camera
    Kmat: torch.Tensor,       # [3,3]
    width: int,
    height: int,
    near_plane: float = K_.NEAR_PLANE,
    far_plane: float = K_.FAR_PLANE,
    radius_clip: float = K_.RADIUS_CLIP,
    eps2d: float = K_.EPS2D,
) -> Dict[str, torch.Tensor]:
    """World -> image projection with EWA 2D covariance (Appendix A.2).

    Returns radii [N,2] int32 (0 = culled), means2d [N,2], depths [N], conics [N,3] (a, b, c).
    Culled gaussians get zeros in the float outputs.
    """
    dt = means.dtype
    R = viewmat[:3, :3].to(dt)
    t = viewmat[:3, 3].to(dt)
    fx, fy, cx, cy = (float(Kmat[0, 0]), float(Kmat[1, 1]), float(Kmat[0, 2]), float(Kmat[1, 2]))

    p = means @ R.T + t
    x, y, z = p.unbind(-1)
    in_z = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(in_z, z, torch.ones_like(z))  # keep the culled lanes finite

    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    cov = M @ M.transpose(-1, -2)
    cov_c = R @ cov @ R.T

    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    lim_xp = (width - cx) / fx + K_.FOV_MARGIN * tanx
    lim_xn = cx / fx + K_.FOV_MARGIN * tanx
    lim_yp = (height - cy) / fy + K_.FOV_MARGIN * tany
    lim_yn = cy / fy + K_.FOV_MARGIN * tany

    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.clamp(x * rz, min=-lim_xn, max=lim_xp)
    ty = zs * torch.clamp(y * rz, min=-lim_yn, max=lim_yp)

    zero = torch.zeros_like(rz)
    J = torch.stack(
        [fx * rz, zero, -fx * tx * rz2,
         zero, fy * rz, -fy * ty * rz2], dim=-1
    ).reshape(-1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(-1, -2)
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    c00 = cov2[:, 0, 0] + eps2d
    c01 = cov2[:, 0, 1]
    c11 = cov2[:, 1, 1] + eps2d
    det = c00 * c11 - c01 * c01
    det_ok = det > 0
    dets = torch.where(det_ok, det, torch.ones_like(det))
    conics = torch.stack([c11 / dets, -c01 / dets, c00 / dets], dim=-1)

    with torch.no_grad():
        o = opacities
        op_ok = o >= K_.ALPHA_THRESHOLD
        o_s = torch.where(op_ok, o, torch.full_like(o, K_.ALPHA_THRESHOLD))
        ext = torch.clamp(torch.sqrt(2.0 * torch.log(o_s / K_.ALPHA_THRESHOLD)), max=K_.RADIUS_EXTEND)
        rx = torch.ceil(ext * torch.sqrt(c00.clamp(min=0)))
        ry = torch.ceil(ext * torch.sqrt(c11.clamp(min=0)))
        rad_ok = ~((rx <= radius_clip) & (ry <= radius_clip))
        mx, my = means2d[:, 0], means2d[:, 1]
        in_img = ~((mx + rx <= 0) | (mx - rx >= width) | (my + ry <= 0) | (my - ry >= height))
        visible = in_z & det_ok & op_ok & rad_ok & in_img
        radii = torch.stack([rx, ry], dim=-1)
        radii = torch.where(visible[:, None], radii, torch.zeros_like(radii)).to(torch.int32)

    vis_f = visible.to(dt)
    return {
        "radii": radii,
        "visible": visible,
        "means2d": means2d * vis_f[:, None],
        "depths": z * vis_f,
        "conics": conics * vis_f[:, None],
    }


# --------------------------------------------------------------------------------------
# A.3  spherical-harmonics colour
# --------------------------------------------------------------------------------------
def sh_basis(degree: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis [N, (degree+1)^2] on normalised directions (Appendix A.3)."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, K_.SH_C0)]
    if degree >= 1:
        out += [-K_.SH_C1 * y, K_.SH_C1 * z, -K_.SH_C1 * x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        out += [
            K_.SH_C2[0] * x * y,
            K_.SH_C2[1] * y * z,
            K_.SH_C2[2] * (2 * zz - xx - yy),
            K_.SH_C2[3] * x * z,
            K_.SH_C2[4] * (xx - yy),
        ]
    if degree >= 3:
        out += [
            K_.SH_C3[0] * y * (3 * xx - yy),
            K_.SH_C3[1] * x * y * z,
            K_.SH_C3[2] * y * (4 * zz - xx - yy),
            K_.SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
            K_.SH_C3[4] * x * (4 * zz - xx - yy),
            K_.SH_C3[5] * z * (xx - yy),
            K_.SH_C3[6] * x * (xx - 3 * yy),
        ]
    return torch.stack(out, dim=-1)


def sh_colors(
    degree: int, means: torch.Tensor, shs: torch.Tensor, viewmat: torch.Tensor, visible: torch.Tensor
) -> torch.Tensor:
    """rgb = max(SH(dir) + 0.5, 0), evaluated where the gaussian is visible (Appendix A.3)."""
    dt = means.dtype
    cam_pos = torch.linalg.inv(viewmat.to(dt))[:3, 3]
    dirs = means - cam_pos
    safe = torch.where(visible[:, None], dirs, torch.ones_like(dirs))
    nb = (degree + 1) ** 2
    basis = sh_basis(degree, safe)                       # [N, nb]
    rgb = (basis[:, :, None] * shs[:, :nb, :]).sum(dim=1)  # [N, 3]
    rgb = torch.clamp_min(rgb + 0.5, 0.0)
    return rgb * visible.to(dt)[:, None]


# --------------------------------------------------------------------------------------
# A.4  tile binning
# --------------------------------------------------------------------------------------
@torch.no_grad()
def tile_intersect(
    means2d: torch.Tensor, radii: torch.Tensor, depths: torch.Tensor,
    tile_size: int, tile_w: int, tile_h: int, cam_id: int = 0, n_gauss: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Per-gaussian tile rectangle -> (tiles_per_gauss [N], isect_ids [M] int64, flatten_ids [M] int32),
    UNSORTED, one record per covered tile (Appendix A.4)."""
    N = means2d.shape[0]
    n_gauss = N if n_gauss is None else n_gauss
    vis = (radii > 0).all(dim=-1)
    r = radii.to(means2d.dtype) / tile_size
    c = means2d / tile_size
    lo = torch.floor(c - r)
    hi = torch.ceil(c + r)
    x0 = lo[:, 0].clamp(0, tile_w).long()
    x1 = hi[:, 0].clamp(0, tile_w).long()
    y0 = lo[:, 1].clamp(0, tile_h).long()
    y1 = hi[:, 1].clamp(0, tile_h).long()
    nx = torch.where(vis, x1 - x0, torch.zeros_like(x0))
    ny = torch.where(vis, y1 - y0, torch.zeros_like(y0))
    cnt = nx * ny
    M = int(cnt.sum())
    gid = torch.repeat_interleave(torch.arange(N), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    j = torch.arange(M) - start[gid]
    nxg = nx[gid].clamp(min=1)
    ty = y0[gid] + j // nxg
    tx = x0[gid] + j % nxg
    tile_id = ty * tile_w + tx
    tile_bits = max(1, math.ceil(math.log2(max(tile_w * tile_h, 2))))
    depth_bits = depths.to(torch.float32).contiguous().view(torch.int32).long() & 0xFFFFFFFF
    keys = (cam_id << (32 + tile_bits)) | (tile_id << 32) | depth_bits[gid]
    flat = (cam_id * n_gauss + gid).to(torch.int32)
    return cnt.to(torch.int32), keys, flat


# --------------------------------------------------------------------------------------
# A.5  front-to-back compositing (autograd gives A.6)
# --------------------------------------------------------------------------------------
def composite(
    means2d: torch.Tensor,   # [N,2]
    conics: torch.Tensor,    # [N,3]
    opacities: torch.Tensor, # [N]
    feats: torch.Tensor,     # [N,D]
    background: Optional[torch.Tensor],  # [D] or None
    width: int, height: int, tile_size: int,
    sorted_gids: torch.Tensor,   # [M] gaussian index per sorted record (this camera)
    tile_offsets: torch.Tensor,  # [T+1] start of each tile's slice (last = M)
    budget: int = 6_000_000,
    ambiguity_tol: Optional[float] = None,
    checkpoint: bool = False,
):
    """Per pixel, front to back over its tile's sorted list (Appendix A.5):
    sigma = 1/2 (a dx^2 + c dy^2) + b dx dy;  alpha = min(0.999, o exp(-sigma));
    skip if sigma < 0 or alpha < 1/255;  stop (splat not added) when T (1 - alpha) <= 1e-4.
    Returns render [H,W,D] (incl. T*background) and alpha [H,W,1] = 1 - T.

    The function is discontinuous at its three thresholds (alpha = 1/255, T(1-alpha) = 1e-4, and for the
    gradient o*exp(-sigma) = 0.999).  With ``ambiguity_tol`` a third value is returned: a bool [H,W] mask of
    the pixels where some comparison the pixel actually reaches sits within that RELATIVE distance of its
    threshold -- there two correct fp32 implementations may legitimately decide differently -- and a fourth: a float
    [H,W] BOUND on how far such an implementation's pixel may lie from this one's because of those decisions alone
    (per colour channel; 0 where nothing is ambiguous).  One flipped SKIP decision at record e moves the pixel by
    alpha_e T_e (c_e - colour behind e), at most alpha_e T_e (|c_e| + Cmax) with Cmax the largest colour in the tile's
    list; a flipped STOP decision at e adds or removes alpha_e T_e c_e and whatever the transmittance T_e (1 - alpha_e)
    that is left lets through, at most alpha_e T_e |c_e| + T_e (1 - alpha_e) Cmax.  The bound sums the first over the
    pixel's near-threshold skip decisions and adds the largest of the second over the stop decisions inside a 2 %
    window of 1e-4 (a flipped skip upstream scales T by 1 - 1/255, which can carry a stop across its threshold from
    further away than ``ambiguity_tol``)."""
    dt = means2d.dtype
    N, D = feats.shape
    tw = (width + tile_size - 1) // tile_size
    th = (height + tile_size - 1) // tile_size
    T_ = tw * th
    npx = tile_size * tile_size
    counts = (tile_offsets[1:] - tile_offsets[:-1]).long()

    # one dummy splat (index N) with zero opacity pads ragged tile lists
    m2 = torch.cat([means2d, means2d.new_zeros(1, 2)])
    cn = torch.cat([conics, conics.new_tensor([[1.0, 0.0, 1.0]])])
    op = torch.cat([opacities, opacities.new_zeros(1)])
    ft = torch.cat([feats, feats.new_zeros(1, D)])

    lin = torch.arange(npx)
    px_in = (lin % tile_size).to(dt) + 0.5
    py_in = (lin // tile_size).to(dt) + 0.5

    def blend(ids, px, py, m2_, cn_, op_, ft_):
        """One batch of tiles: ids [nb,L] (padded with N), pixel centres [nb,npx]."""
        g_m = m2_[ids]                                                  # [nb, L, 2]
        g_c = cn_[ids]
        g_o = op_[ids]
        g_f = ft_[ids]                                                  # [nb, L, D]
        dx = g_m[:, None, :, 0] - px[:, :, None]                        # [nb, npx, L]
        dy = g_m[:, None, :, 1] - py[:, :, None]
        sigma = 0.5 * (g_c[:, None, :, 0] * dx * dx + g_c[:, None, :, 2] * dy * dy) \
            + g_c[:, None, :, 1] * dx * dy
        alpha = torch.clamp(g_o[:, None, :] * torch.exp(-sigma), max=K_.MAX_ALPHA)
        with torch.no_grad():
            valid = (sigma >= 0) & (alpha >= K_.ALPHA_THRESHOLD)
            a_v = torch.where(valid, alpha, torch.zeros_like(alpha))
            next_T = torch.cumprod(1.0 - a_v, dim=-1)
            contrib = valid & (next_T > K_.TRANSMITTANCE_EPS)
            amb = torch.zeros(ids.shape[0], px.shape[1], dtype=torch.bool)
            bound = torch.zeros(ids.shape[0], px.shape[1], dtype=dt)
            if ambiguity_tol is not None:
                tol = ambiguity_tol
                raw = g_o[:, None, :] * torch.exp(-sigma)
                prev_T = torch.cat([torch.ones_like(next_T[..., :1]), next_T[..., :-1]], -1)
                live = prev_T > K_.TRANSMITTANCE_EPS
                live[..., 0] = True
                near_a = (alpha * 255.0 - 1.0).abs() < tol
                near_t = valid & ((next_T / K_.TRANSMITTANCE_EPS - 1.0).abs() < tol)
                near_c = valid & ((raw / K_.MAX_ALPHA - 1.0).abs() < tol)
                reach = live & (sigma >= 0)
                amb = ((near_a | near_t | near_c) & reach).any(dim=-1)
                cnorm = g_f.abs().amax(-1)[:, None, :]                   # [nb, 1, L]
                cmax = cnorm.amax(-1, keepdim=True)
                wgt = alpha * prev_T
                skip_term = torch.where(near_a & reach, wgt * (cnorm + cmax), torch.zeros_like(wgt)).sum(-1)
                wide_t = valid & reach & ((next_T / K_.TRANSMITTANCE_EPS - 1.0).abs() < 0.02)
                stop_term = torch.where(wide_t, wgt * cnorm + next_T * cmax, torch.zeros_like(wgt)).amax(-1)
                bound = torch.where(amb, skip_term + stop_term, torch.zeros_like(skip_term))
        a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1.0 - a_eff, dim=-1)
        T_excl = torch.cat([torch.ones_like(T_incl[..., :1]), T_incl[..., :-1]], dim=-1)
        w = a_eff * T_excl
        return torch.einsum("tpl,tld->tpd", w, g_f), T_incl[..., -1], amb, bound

    order = torch.argsort(counts, stable=True)
    order = order[counts[order] > 0]
    lens = counts[order].tolist()
    ar_max = torch.arange(max(lens) if lens else 0)
    tile_idx_out, color_out, trans_out, amb_out, bound_out = [], [], [], [], []
    i, n_t = 0, len(lens)
    while i < n_t:
        # grow a batch of tiles of similar list length (ascending) under the element budget
        j = i + 1
        while j < n_t and (j + 1 - i) * npx * lens[j] <= budget:
            j += 1
        tiles = order[i:j]
        L = lens[j - 1]
        i = j
        pos = tile_offsets[tiles].long()[:, None] + ar_max[None, :L]
        ok = ar_max[None, :L] < counts[tiles][:, None]
        ids = torch.where(ok, sorted_gids.long()[pos.clamp(max=max(sorted_gids.numel() - 1, 0))],
                          torch.full_like(pos, N))                      # [nb, L]
        px = ((tiles % tw) * tile_size).to(dt)[:, None] + px_in[None, :]   # [nb, npx]
        py = ((tiles // tw) * tile_size).to(dt)[:, None] + py_in[None, :]
        if checkpoint and torch.is_grad_enabled():
            color, t_fin, amb, bnd = torch.utils.checkpoint.checkpoint(blend, ids, px, py, m2, cn, op, ft,
                                                                       use_reentrant=False)
        else:
            color, t_fin, amb, bnd = blend(ids, px, py, m2, cn, op, ft)
        tile_idx_out.append(tiles)
        color_out.append(color)
        trans_out.append(t_fin)
        amb_out.append(amb)
        bound_out.append(bnd.detach())

    canvas_c = means2d.new_zeros(T_, npx, D)
    canvas_t = means2d.new_ones(T_, npx)
    canvas_a = torch.zeros(T_, npx, dtype=torch.bool)
    canvas_b = means2d.new_zeros(T_, npx).detach()
    if tile_idx_out:
        tiles_all = torch.cat(tile_idx_out)
        canvas_c = canvas_c.index_copy(0, tiles_all, torch.cat(color_out))
        canvas_t = canvas_t.index_copy(0, tiles_all, torch.cat(trans_out))
        if ambiguity_tol is not None:
            canvas_a = canvas_a.index_copy(0, tiles_all, torch.cat(amb_out))
            canvas_b = canvas_b.index_copy(0, tiles_all, torch.cat(bound_out))
    if background is not None:
        canvas_c = canvas_c + canvas_t[..., None] * background.to(dt)

    def untile(v: torch.Tensor) -> torch.Tensor:  # [T, npx, k] -> [H, W, k]
        k = v.shape[-1]
        v = v.reshape(th, tw, tile_size, tile_size, k).permute(0, 2, 1, 3, 4)
        return v.reshape(th * tile_size, tw * tile_size, k)[:height, :width]

    if ambiguity_tol is not None:
        return (untile(canvas_c), untile((1.0 - canvas_t)[..., None]), untile(canvas_a[..., None])[..., 0],
                untile(canvas_b[..., None])[..., 0])
    return untile(canvas_c), untile((1.0 - canvas_t)[..., None])


# --------------------------------------------------------------------------------------
# the operator surface: gsplat.rasterization as the reference calls it
# --------------------------------------------------------------------------------------
def rasterization(
    means: torch.Tensor,
    quats: torch.Tensor,
    scales: torch.Tensor,
    opacities: torch.Tensor,
    colors: torch.Tensor,
    viewmats: torch.Tensor,
    Ks: torch.Tensor,
    width: int,
    height: int,
    near_plane: float = K_.NEAR_PLANE,
    far_plane: float = K_.FAR_PLANE,
    radius_clip: float = K_.RADIUS_CLIP,
    eps2d: float = K_.EPS2D,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = K_.TILE_SIZE,
    backgrounds: Optional[torch.Tensor] = None,
    render_mode: str = "RGB",
    ambiguity_tol: Optional[float] = None,
    checkpoint: bool = False,
    **unsupported,
) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """``ambiguity_tol`` (oracle-only extension): adds ``meta['ambiguous']`` [C,H,W] bool and ``meta['flip_bound']``
    [C,H,W] (how far a pixel may move through its ambiguous threshold decisions), see ``composite``.

    Same keyword surface and return contract as the call at
    /root/reference/src/backends/gsplat_backend.py:168-182:
    ``(renders [C,H,W,D], alphas [C,H,W,1], meta)`` with ``meta['means2d']`` [C,N,2] and
    ``meta['radii']`` [C,N,2] int32 (read at gsplat_backend.py:204-208)."""
    for k, v in unsupported.items():
        if k in ("sparse_grad", "absgrad", "distributed") and not v:
            continue
        if k == "rasterize_mode" and v == "classic":
            continue
        if k == "camera_model" and v == "pinhole":
            continue
        raise NotImplementedError(f"oracle: argument {k}={v!r} is outside the reference's call")
    if packed:
        raise NotImplementedError("oracle: packed=True is outside the reference's call")
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"unknown render_mode {render_mode!r}")
    N = means.shape[0]
    C = viewmats.shape[0]
    if opacities.shape != (N,):
        raise ValueError(f"opacities must be [N], got {tuple(opacities.shape)}")
    width, height = int(width), int(height)
    tw = (width + tile_size - 1) // tile_size
    th = (height + tile_size - 1) // tile_size

    out_r, out_a = [], []
    m_means2d, m_radii, m_depths, m_conics, m_tiles = [], [], [], [], []
    keys_all, flat_all, m_amb, m_bound = [], [], [], []
    for c in range(C):
        pr = project(means, quats, scales, opacities, viewmats[c], Ks[c], width, height,
                     near_plane, far_plane, radius_clip, eps2d)
        vis = pr["visible"]
        if render_mode in ("D", "ED"):
            feats = pr["depths"][:, None]
        else:
            if sh_degree is None:
                rgb = colors[c] if colors.dim() == 3 else colors
                rgb = rgb * vis.to(rgb.dtype)[:, None]
            else:
                shs = colors[c] if colors.dim() == 4 else colors
                if shs.shape[-2] < (sh_degree + 1) ** 2:
                    raise ValueError("colors holds fewer SH coefficients than sh_degree needs")
                rgb = sh_colors(sh_degree, means, shs, viewmats[c], vis)
            feats = rgb
            if render_mode in ("RGB+D", "RGB+ED"):
                feats = torch.cat([rgb, pr["depths"][:, None]], dim=-1)
        bg = None
        if backgrounds is not None:
            bg = backgrounds[c]
            if render_mode in ("RGB+D", "RGB+ED"):
                bg = torch.cat([bg, bg.new_zeros(1)])
            elif render_mode in ("D", "ED"):
                bg = bg.new_zeros(1)

        cnt, keys, flat = tile_intersect(pr["means2d"].detach(), pr["radii"], pr["depths"].detach(),
                                         tile_size, tw, th, cam_id=c, n_gauss=N)
        order = torch.sort(keys, stable=True).indices
        keys, flat = keys[order], flat[order]
        tile_bits = max(1, math.ceil(math.log2(max(tw * th, 2))))
        tile_of = (keys >> 32) & ((1 << tile_bits) - 1)
        offs = torch.searchsorted(tile_of, torch.arange(tw * th + 1))
        comp = composite(pr["means2d"], pr["conics"], opacities, feats, bg,
                         width, height, tile_size, flat.long() - c * N, offs, ambiguity_tol=ambiguity_tol,
                         checkpoint=checkpoint)
        img, alpha = comp[0], comp[1]
        if ambiguity_tol is not None:
            m_amb.append(comp[2])
            m_bound.append(comp[3])
        if render_mode in ("ED", "RGB+ED"):
            img = torch.cat([img[..., :-1], img[..., -1:] / alpha.clamp(min=K_.DEPTH_ALPHA_EPS)], dim=-1)
        out_r.append(img)
        out_a.append(alpha)
        m_means2d.append(pr["means2d"]); m_radii.append(pr["radii"]); m_depths.append(pr["depths"])
        m_conics.append(pr["conics"]); m_tiles.append(cnt)
        keys_all.append(keys); flat_all.append(flat)
        # offsets are global positions in the concatenated record list
        if c == 0:
            offsets_all = [offs[:-1]]
            base = keys.numel()
        else:
            offsets_all.append(offs[:-1] + base)
            base += keys.numel()

    meta = {
        "camera_ids": None, "gaussian_ids": None,
        "means2d": torch.stack(m_means2d), "radii": torch.stack(m_radii),
        "depths": torch.stack(m_depths), "conics": torch.stack(m_conics),
        "opacities": opacities[None].expand(C, N),
        "tiles_per_gauss": torch.stack(m_tiles),
        "isect_ids": torch.cat(keys_all), "flatten_ids": torch.cat(flat_all),
        "isect_offsets": torch.stack(offsets_all).reshape(C, th, tw).to(torch.int32),
        "tile_width": tw, "tile_height": th, "tile_size": tile_size,
        "width": width, "height": height, "n_cameras": C,
    }
    if ambiguity_tol is not None:
        meta["ambiguous"] = torch.stack(m_amb)
        meta["flip_bound"] = torch.stack(m_bound)   # [C,H,W]: see composite()
    return torch.stack(out_r), torch.stack(out_a), meta


# --------------------------------------------------------------------------------------
# the plugin surface: what src/backends exposes (fills the PyTorchBackend hole)
# --------------------------------------------------------------------------------------
class OracleBackend:
    """CPU backend with the surface of the reference's ``GSplatBackend``
    (/root/reference/src/backends/gsplat_backend.py:7-210) -- the role the reference's
    ``PyTorchBackend`` stub leaves empty (/root/reference/src/backends/pytorch_backend.py:1-5)."""

    _MODE_KEYS = {
        "RGB": (("render", slice(0, 3)),),
        "D": (("accumulated_depth", 0),),
        "ED": (("expected_depth", 0),),
        "RGB+D": (("render", slice(0, 3)), ("accumulated_depth", 3)),
        "RGB+ED": (("render", slice(0, 3)), ("expected_depth", 3)),
    }

    def __init__(self, K: torch.Tensor):
        self.rasterization = rasterization
        self.K = K

    def _call(self, params: Dict, viewpoint: Dict, bg: torch.Tensor, mode: str, sh_degree, squeeze_opacity=True):
        opac = params["opacities"]
        if squeeze_opacity and opac.dim() == 2:
            opac = opac.squeeze(-1)
        colors = params.get("colors", params.get("shs"))
        return self.rasterization(
            means=params["means3D"], quats=params["rotations"], scales=params["scales"],
            opacities=opac, colors=colors,
            viewmats=viewpoint["world_view_transform"][None], Ks=self.K[None],
            width=viewpoint["image_width"], height=viewpoint["image_height"],
            packed=False, render_mode=mode, sh_degree=sh_degree,
            backgrounds=bg[None] if bg.dim() == 1 else bg,
        )

    def render_with_depth(self, gaussian_params: Dict, viewpoint: Dict, bg_color: torch.Tensor = None,
                          render_mode: str = "RGB+ED", device: str = "cpu", sh_degree: int = None) -> Dict:
        """gsplat_backend.py:129-210."""
        if bg_color is None:
            bg_color = torch.zeros(3, device=device)
        if sh_degree is None:
            sh_degree = 3 if "shs" in gaussian_params else None
        renders, alphas, meta = self._call(gaussian_params, viewpoint, bg_color, render_mode, sh_degree)
        out, result = renders[0], {"alpha": alphas[0]}
        for key, sel in self._MODE_KEYS[render_mode]:
            result[key] = out[..., sel]
        if "means2d" in meta:
            result["means2d"] = meta["means2d"][0]
        if "radii" in meta:
            result["radii"] = meta["radii"][0]
            result["visibility_filter"] = result["radii"] > 0
        return result

    def render_single(self, gaussian_params: Dict, viewpoint: Dict, bg_color: torch.Tensor) -> Dict:
        """gsplat_backend.py:57-127 (always RGB+ED, SH degree 3)."""
        r = self.render_with_depth(gaussian_params, viewpoint, bg_color, "RGB+ED", str(bg_color.device), 3)
        return {"render": r["render"], "alpha": r["alpha"], "depth": r["expected_depth"],
                "means2d": r["means2d"], "radii": r["radii"], "visibility_filter": r["visibility_filter"]}

    def render_batch(self, gaussian_params: Dict, view_matrices: torch.Tensor, proj_matrices: torch.Tensor,
                     width: int, height: int, bg_color: torch.Tensor, device: str) -> Dict:
        """gsplat_backend.py:12-55 (a loop of single views; images come back channel-first)."""
        B = view_matrices.shape[0]
        images = torch.zeros(B, 3, height, width, device=device)
        alphas = torch.zeros(B, 1, height, width, device=device)
        for i in range(B):
            vp = {"world_view_transform": view_matrices[i], "projection_matrix": proj_matrices[i],
                  "image_width": width, "image_height": height}
            r = self.render_single(gaussian_params, vp, bg_color)
            images[i] = r["render"].permute(2, 0, 1)
            alphas[i] = r["alpha"].permute(2, 0, 1)
        return {"images": images, "alphas": alphas}
