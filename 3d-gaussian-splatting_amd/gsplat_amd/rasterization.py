"""Operator surface: ``rasterization(...)`` with the keyword surface and return contract of the call
the reference makes at /root/reference/src/backends/gsplat_backend.py:168-182
(``gsplat.rasterization``), executed by the HIP library through its C ABI.

One ``torch.autograd.Function`` per camera wraps the five C-ABI stages; torch only owns memory
(caching allocator), the stream and autograd bookkeeping.
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from ._lib import SPLAT_STRIDE, check, lib

TILE_SIZE = 16
_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")
_DEFAULT_CULL_MODE = 0


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"gsplat_amd.rasterization: `{name}` must live on a HIP device (got {t.device}); "
                           "there is no CPU path in the product")
    return t.detach().to(torch.float32).contiguous()


def _sort_pairs(n: int, keys: torch.Tensor, vals: Optional[torch.Tensor], begin_bit: int, end_bit: int):
    """Stable radix sort of int32-typed storage holding uint32 keys; returns (sorted_keys, sorted_vals)."""
    dev = keys.device
    k_out = torch.empty(n, dtype=torch.int32, device=dev)
    v_out = torch.empty(n, dtype=torch.int32, device=dev)
    k_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    v_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsa_sort_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.gsa_sort_pairs_u32(n, _ptr(keys), _ptr(vals), _ptr(k_out), _ptr(v_out), _ptr(k_tmp), _ptr(v_tmp),
                                 begin_bit, end_bit, _ptr(ws), ws_bytes, _stream()), "gsa_sort_pairs_u32")
    return k_out, v_out


class _RasterizeOneCamera(torch.autograd.Function):
    """Forward: project -> depth sort -> scan -> emit -> tile sort -> offsets -> composite.
    Backward: composite_bwd -> project_bwd.  Differentiable w.r.t. means, quats, scales, opacities, colors
    (/root/reference/src/training/trainer.py:889-898 reads those grads); none to camera or background."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, Kmat, background,
                width: int, height: int, sh_degree: int, channels: int, cull_mode: int):
        dev = means.device
        N = means.shape[0]
        means_c, quats_c, scales_c = _f32c(means, "means"), _f32c(quats, "quats"), _f32c(scales, "scales")
        opac_c, colors_c = _f32c(opacities, "opacities"), _f32c(colors, "colors")
        view_c, K_c = _f32c(viewmat, "viewmats"), _f32c(Kmat, "Ks")
        bg_c = None if background is None else _f32c(background, "backgrounds")
        sh_coeffs = colors_c.shape[1] if sh_degree >= 0 else 0
        tile_w = (width + TILE_SIZE - 1) // TILE_SIZE
        tile_h = (height + TILE_SIZE - 1) // TILE_SIZE
        n_tiles = tile_w * tile_h
        st = _stream()

        splat = torch.empty(N, SPLAT_STRIDE, dtype=torch.float32, device=dev)
        radii = torch.empty(N, 2, dtype=torch.int32, device=dev)
        tiles_touched = torch.empty(N, dtype=torch.int32, device=dev)
        depth_keys = torch.empty(N, dtype=torch.int32, device=dev)
        check(lib.gsa_project_fwd(N, _ptr(means_c), _ptr(quats_c), _ptr(scales_c), _ptr(opac_c), _ptr(colors_c),
                                  sh_degree, sh_coeffs, _ptr(view_c), _ptr(K_c), width, height, cull_mode,
                                  _ptr(splat), _ptr(radii), _ptr(tiles_touched), _ptr(depth_keys), st),
              "gsa_project_fwd")

        # depth order of the gaussians (ties keep index order; culled ones sort last)
        _, order = _sort_pairs(N, depth_keys, None, 0, 32)
        offsets = torch.empty(N, dtype=torch.int32, device=dev)
        total = torch.empty(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.gsa_sort_workspace_bytes(N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.gsa_scan_tiles(N, _ptr(tiles_touched), _ptr(order), _ptr(offsets), _ptr(total), _ptr(ws),
                                 ws_bytes, st), "gsa_scan_tiles")
        M = int(total.item())  # the one host read-back of the path: sizes the record buffers

        isect_tile = torch.empty(M, dtype=torch.int32, device=dev)
        isect_gid = torch.empty(M, dtype=torch.int32, device=dev)
        check(lib.gsa_tile_emit(N, _ptr(order), _ptr(offsets), _ptr(splat), _ptr(radii), width, height, cull_mode,
                                M, _ptr(isect_tile), _ptr(isect_gid), st), "gsa_tile_emit")
        tile_bits = max(1, math.ceil(math.log2(n_tiles + 1)))
        if M > 0:
            sorted_tile, sorted_gid = _sort_pairs(M, isect_tile, isect_gid, 0, tile_bits)
        else:
            sorted_tile, sorted_gid = isect_tile, isect_gid
        tile_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        check(lib.gsa_tile_offsets(M, _ptr(sorted_tile), n_tiles, _ptr(tile_offsets), st), "gsa_tile_offsets")

        render = torch.empty(height, width, channels, dtype=torch.float32, device=dev)
        alpha = torch.empty(height, width, 1, dtype=torch.float32, device=dev)
        last_id = torch.empty(height, width, dtype=torch.int32, device=dev)
        check(lib.gsa_composite_fwd(width, height, channels, _ptr(splat), _ptr(sorted_gid), _ptr(tile_offsets),
                                    _ptr(bg_c), _ptr(render), _ptr(alpha), _ptr(last_id), st), "gsa_composite_fwd")

        ctx.save_for_backward(means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii,
                              sorted_gid, tile_offsets, alpha, last_id)
        ctx.dims = (N, width, height, sh_degree, sh_coeffs, channels)
        ctx.in_shapes = (means.shape, quats.shape, scales.shape, opacities.shape, colors.shape)
        extras = (splat, radii, tiles_touched, sorted_tile, sorted_gid, tile_offsets, last_id)
        ctx.mark_non_differentiable(*extras)
        return (render, alpha) + extras

    @staticmethod
    def backward(ctx, v_render, v_alpha, *unused):
        (means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii, sorted_gid, tile_offsets,
         alpha, last_id) = ctx.saved_tensors
        N, width, height, sh_degree, sh_coeffs, channels = ctx.dims
        dev = means_c.device
        st = _stream()
        if v_render is None:
            v_render = torch.zeros(height, width, channels, dtype=torch.float32, device=dev)
        v_render = v_render.to(torch.float32).contiguous()
        v_alpha_c = None if v_alpha is None else v_alpha.to(torch.float32).contiguous()

        v_splat = torch.zeros(N, SPLAT_STRIDE, dtype=torch.float32, device=dev)
        check(lib.gsa_composite_bwd(width, height, channels, _ptr(splat), _ptr(sorted_gid), _ptr(tile_offsets),
                                    _ptr(bg_c), _ptr(alpha), _ptr(last_id), _ptr(v_render), _ptr(v_alpha_c),
                                    _ptr(v_splat), st), "gsa_composite_bwd")
        v_means = torch.empty_like(means_c)
        v_quats = torch.empty_like(quats_c)
        v_scales = torch.empty_like(scales_c)
        v_opac = torch.empty_like(opac_c)
        v_colors = torch.empty_like(colors_c)
        check(lib.gsa_project_bwd(N, _ptr(means_c), _ptr(quats_c), _ptr(scales_c), _ptr(opac_c), _ptr(colors_c),
                                  sh_degree, sh_coeffs, _ptr(view_c), _ptr(K_c), width, height, _ptr(splat),
                                  _ptr(radii), _ptr(v_splat), _ptr(v_means), _ptr(v_quats), _ptr(v_scales),
                                  _ptr(v_opac), _ptr(v_colors), st), "gsa_project_bwd")
        shp = ctx.in_shapes
        need = ctx.needs_input_grad
        grads = (v_means.view(shp[0]) if need[0] else None, v_quats.view(shp[1]) if need[1] else None,
                 v_scales.view(shp[2]) if need[2] else None, v_opac.view(shp[3]) if need[3] else None,
                 v_colors.view(shp[4]) if need[4] else None)
        return grads + (None,) * 8


def rasterization(
    means: torch.Tensor,       # [N,3]
    quats: torch.Tensor,       # [N,4] (w first)
    scales: torch.Tensor,      # [N,3]
    opacities: torch.Tensor,   # [N]
    colors: torch.Tensor,      # [N,K,3] SH coefficients (sh_degree given) or [N,3] colours (sh_degree None)
    viewmats: torch.Tensor,    # [C,4,4] world->camera
    Ks: torch.Tensor,          # [C,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = TILE_SIZE,
    backgrounds: Optional[torch.Tensor] = None,
    render_mode: str = "RGB",
    cull_mode: Optional[int] = None,
    **unsupported,
) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """Returns ``(renders [C,H,W,D], alphas [C,H,W,1], meta)``; ``meta['means2d']`` [C,N,2] and
    ``meta['radii']`` [C,N,2] int32 are what the reference reads (gsplat_backend.py:204-208).

    ``cull_mode`` (extension, default 0): 0 = tile rectangle of mean2d +- radius (the published rule);
    1 = also skip tiles the alpha >= 1/255 ellipse cannot reach (same image and gradients, fewer records).
    """
    for k, v in unsupported.items():
        ok = (k in ("sparse_grad", "absgrad", "distributed") and not v) or \
             (k == "rasterize_mode" and v == "classic") or (k == "camera_model" and v == "pinhole")
        if not ok:
            raise NotImplementedError(f"gsplat_amd.rasterization: argument {k}={v!r} is outside the reference's call")
    if packed:
        raise NotImplementedError("gsplat_amd.rasterization: packed=True is outside the reference's call")
    if (near_plane, far_plane, radius_clip, eps2d, tile_size) != (0.01, 1e10, 0.0, 0.3, TILE_SIZE):
        raise NotImplementedError("gsplat_amd.rasterization: near/far/radius_clip/eps2d/tile_size are fixed at the "
                                  "defaults the reference uses (include/gsplat_amd_constants.h)")
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"unknown render_mode {render_mode!r}")
    N = means.shape[0]
    C = viewmats.shape[0]
    if means.shape != (N, 3) or quats.shape != (N, 4) or scales.shape != (N, 3):
        raise ValueError("means/quats/scales must be [N,3]/[N,4]/[N,3]")
    if opacities.shape != (N,):
        raise ValueError(f"opacities must be [N], got {tuple(opacities.shape)}")
    if viewmats.shape != (C, 4, 4) or Ks.shape != (C, 3, 3):
        raise ValueError("viewmats/Ks must be [C,4,4]/[C,3,3]")
    if sh_degree is None:
        if colors.shape != (N, 3):
            raise ValueError("with sh_degree=None colors must be [N,3]")
        deg = -1
    else:
        deg = int(sh_degree)
        if not (0 <= deg <= 3):
            raise ValueError("sh_degree must be in 0..3")
        if colors.dim() != 3 or colors.shape[0] != N or colors.shape[2] != 3 or colors.shape[1] < (deg + 1) ** 2:
            raise ValueError("colors must be [N,K,3] with K >= (sh_degree+1)^2")
    if backgrounds is not None and backgrounds.shape != (C, 3):
        raise ValueError("backgrounds must be [C,3]")
    width, height = int(width), int(height)
    cull = _DEFAULT_CULL_MODE if cull_mode is None else int(cull_mode)
    want_depth = render_mode != "RGB"
    channels = 4 if want_depth else 3

    renders, alphas = [], []
    m_splat, m_radii, m_tiles, m_keys_tile, m_flat, m_offs = [], [], [], [], [], []
    base = 0
    for c in range(C):
        bg = None
        if backgrounds is not None:
            bg = backgrounds[c]
            if want_depth:
                bg = torch.cat([bg, bg.new_zeros(1)])
        out = _RasterizeOneCamera.apply(means, quats, scales, opacities, colors, viewmats[c], Ks[c], bg,
                                        width, height, deg, channels, cull)
        render, alpha, splat, radii, tiles_touched, sorted_tile, sorted_gid, tile_offsets, _last = out
        if render_mode in ("ED", "RGB+ED"):
            render = torch.cat([render[..., :3], render[..., 3:] / alpha.clamp(min=1e-10)], dim=-1)
        if render_mode in ("D", "ED"):
            render = render[..., 3:]
        renders.append(render)
        alphas.append(alpha)
        m_splat.append(splat); m_radii.append(radii); m_tiles.append(tiles_touched)
        m_keys_tile.append(sorted_tile); m_flat.append(sorted_gid + c * N)
        m_offs.append(tile_offsets[:-1] + base)
        base += sorted_gid.numel()

    splat_all = torch.stack(m_splat)  # [C,N,16]
    tile_w = (width + TILE_SIZE - 1) // TILE_SIZE
    tile_h = (height + TILE_SIZE - 1) // TILE_SIZE
    meta = _Meta({
        "camera_ids": None, "gaussian_ids": None,
        "means2d": splat_all[..., 0:2], "conics": splat_all[..., 2:5], "depths": splat_all[..., 9],
        "radii": torch.stack(m_radii), "opacities": opacities[None].expand(C, N),
        "tiles_per_gauss": torch.stack(m_tiles), "flatten_ids": torch.cat(m_flat),
        "isect_offsets": torch.stack(m_offs).reshape(C, tile_h, tile_w),
        "tile_width": tile_w, "tile_height": tile_h, "tile_size": TILE_SIZE,
        "width": width, "height": height, "n_cameras": C, "cull_mode": cull,
    })
    meta._lazy_isect = (m_keys_tile, m_flat, splat_all, N, n_tiles_bits(tile_w * tile_h))
    return torch.stack(renders), torch.stack(alphas), meta


def n_tiles_bits(n_tiles: int) -> int:
    return max(1, math.ceil(math.log2(max(n_tiles, 2))))


class _Meta(dict):
    """``meta`` dict; the 64-bit ``isect_ids`` of the published layout (camera | tile | depth bits) are
    not needed by the kernels (the sort works on split keys) and are only materialised when asked for."""

    _lazy_isect = None

    def __missing__(self, key):
        if key == "isect_ids" and self._lazy_isect is not None:
            tiles, flats, splat_all, N, tile_bits = self._lazy_isect
            parts = []
            for c, (tl, fl) in enumerate(zip(tiles, flats)):
                depth_bits = splat_all[c, :, 9].contiguous().view(torch.int32).long() & 0xFFFFFFFF
                gid = fl.long() - c * N
                parts.append((c << (32 + tile_bits)) | (tl.long() << 32) | depth_bits[gid])
            val = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64)
            self[key] = val
            return val
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == "isect_ids" and self._lazy_isect is not None)
