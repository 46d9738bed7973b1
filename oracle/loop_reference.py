"""Literal per-pixel loops for compositing forward (A.5) and backward (A.6).  Small cases only.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

``torch_oracle.composite`` is vectorised and leaves the backward to autograd.  This file restates
the published per-pixel algorithm step by step in float64 numpy -- the front-to-back loop with
its skip and stop rules, and the reverse sweep from ``last_id`` with the closed-form gradient
 -- so tests can check that the vectorised formulation and its
autograd gradients are the same function as the sequential one.
"""

from __future__ import annotations

import numpy as np

from . import constants as K_


def composite_fwd_loops(means2d, conics, opac, rgb, bg, width, height, tile_size, sorted_gids, tile_offsets):
    """Returns render [H,W,D], alpha [H,W], last_id [H,W] (index into the sorted list, -1 = none)."""
    D = rgb.shape[1]
    tw = (width + tile_size - 1) // tile_size
    render = np.zeros((height, width, D))
    alpha_out = np.zeros((height, width))
    last = -np.ones((height, width), dtype=np.int64)
    for i in range(height):
        for j in range(width):
            tile = (i // tile_size) * tw + (j // tile_size)
            px, py = j + 0.5, i + 0.5
            T = 1.0
            acc = np.zeros(D)
            for m in range(int(tile_offsets[tile]), int(tile_offsets[tile + 1])):
                g = int(sorted_gids[m])
                dx, dy = means2d[g, 0] - px, means2d[g, 1] - py
                a, b, c = conics[g]
                sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
                alpha = min(K_.MAX_ALPHA, opac[g] * np.exp(-sigma))
                if sigma < 0 or alpha < K_.ALPHA_THRESHOLD:
                    continue
                nT = T * (1 - alpha)
                if nT <= K_.TRANSMITTANCE_EPS:
                    break
                acc += rgb[g] * alpha * T
                T = nT
                last[i, j] = m
            render[i, j] = acc + T * bg
            alpha_out[i, j] = 1 - T
    return render, alpha_out, last


def composite_bwd_loops(means2d, conics, opac, rgb, bg, width, height, tile_size, sorted_gids, tile_offsets,
                        alpha_img, last, v_render, v_alpha):
    """Reverse sweep (A.6).  Returns v_means2d [N,2], v_conics [N,3], v_opac [N], v_rgb [N,D]."""
    N, D = rgb.shape
    tw = (width + tile_size - 1) // tile_size
    v_m = np.zeros((N, 2)); v_c = np.zeros((N, 3)); v_o = np.zeros(N); v_f = np.zeros((N, D))
    for i in range(height):
        for j in range(width):
            if last[i, j] < 0:
                continue
            tile = (i // tile_size) * tw + (j // tile_size)
            px, py = j + 0.5, i + 0.5
            T_f = 1 - alpha_img[i, j]
            T = T_f
            buf = np.zeros(D)
            vo = v_render[i, j]
            for m in range(int(last[i, j]), int(tile_offsets[tile]) - 1, -1):
                g = int(sorted_gids[m])
                dx, dy = means2d[g, 0] - px, means2d[g, 1] - py
                a, b, c = conics[g]
                sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
                vis = np.exp(-sigma)
                alpha = min(K_.MAX_ALPHA, opac[g] * vis)
                if sigma < 0 or alpha < K_.ALPHA_THRESHOLD:
                    continue
                ra = 1.0 / (1.0 - alpha)
                T *= ra
                fac = alpha * T
                v_f[g] += fac * vo
                v_al = np.dot(rgb[g] * T - buf * ra, vo) + T_f * ra * v_alpha[i, j] - T_f * ra * np.dot(bg, vo)
                if opac[g] * vis <= K_.MAX_ALPHA:
                    v_sigma = -opac[g] * vis * v_al
                    v_c[g] += (0.5 * v_sigma * dx * dx, v_sigma * dx * dy, 0.5 * v_sigma * dy * dy)
                    v_m[g] += (v_sigma * (a * dx + b * dy), v_sigma * (b * dx + c * dy))
                    v_o[g] += vis * v_al
                buf += rgb[g] * fac
    return v_m, v_c, v_o, v_f
