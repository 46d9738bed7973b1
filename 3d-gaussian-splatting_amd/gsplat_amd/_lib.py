"""ctypes binding of ``lib/libgsplat_amd.so`` -- the C ABI declared in ``include/gsplat_amd.h``.

Loading fails loudly (ImportError) when the library has not been built: there is no CPU or
PyTorch fallback on the product path.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint32, c_void_p

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# GSPLAT_AMD_LIB: developer override to A/B another build of the SAME C ABI (still a HIP library; no fallback)
LIB_PATH = os.environ.get("GSPLAT_AMD_LIB") or os.path.join(_PKG_ROOT, "lib", "libgsplat_amd.so")
ABI_VERSION = 14
SPLAT_STRIDE = 16

P = c_void_p  # every device pointer and the stream cross the boundary as void*


class ViewArgs(ctypes.Structure):
    """``gsa_view_args`` of include/gsplat_amd.h (ABI 13), field for field: the arguments of one view's fused forward /
    backward chain (``gsa_view_fwd`` / ``gsa_view_bwd``)."""
    _fields_ = [
        ("struct_bytes", c_uint32),
        ("n", c_int), ("width", c_int), ("height", c_int), ("channels", c_int), ("bin_tile", c_int), ("gid_boxes", c_int),
        ("n_blocks", c_int),
        ("m_capacity", c_uint32), ("key_min", c_uint32), ("scale", c_uint32), ("token", c_uint32),
        ("splat", P), ("depth_keys", P), ("tile_rects", P), ("key_range", P), ("qbox", P), ("background", P),
        ("order", P), ("rects", P), ("offsets", P), ("block_owner", P), ("ctl", P), ("ctl_host", P),
        ("depth_ws", P), ("depth_ws_bytes", c_size_t),
        ("sorted_gid", P), ("tile_offsets", P), ("tile_order", P), ("bin_ws", P), ("bin_ws_bytes", c_size_t),
        ("wait_event", P),
        ("render", P), ("alpha", P), ("final_T", P), ("last_id", P), ("quad_list", P), ("quad_count", P), ("v_splat", P),
        ("v_render", P), ("v_alpha", P), ("bwd_tile_order", P),
        ("project_mode", c_int), ("accumulate", c_int), ("sh_degree", c_int), ("sh_coeffs", c_int),
        ("wait_bucket", P),
        ("means", P), ("quats", P), ("scales", P), ("opacities", P), ("colors", P), ("viewmat", P), ("intrin", P), ("sh_aux", P),
        ("radii", P),
        ("v_means", P), ("v_quats", P), ("v_scales", P), ("v_opacities", P), ("v_colors", P), ("v_rgb", P),
        ("sweep_rgb", P), ("hot", c_int),
    ]


# name -> (restype, argtypes); mirrors include/gsplat_amd.h one to one
SIGNATURES = {
    "gsa_abi_version": (c_int, []),
    "gsa_error_string": (c_char_p, [c_int]),
    "gsa_project_fwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P, P, P]),
    "gsa_project_fwd_batch": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                      c_int, P, P, P]),
    "gsa_project_fwd_parts": (c_int, [c_int, c_int, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, P, P, P, P,
                                      P, P, c_int, P, P, P]),
    "gsa_sort_workspace_bytes": (c_size_t, [c_int]),
    "gsa_sort_pairs_u32": (c_int, [c_int, P, P, P, P, P, P, c_int, c_int, P, c_size_t, P]),
    "gsa_scan_tiles": (c_int, [c_int, P, P, P, P, P, c_size_t, P]),
    "gsa_tile_bin16_supported": (c_int, [c_int, c_int]),
    "gsa_tile_bin16_direct": (c_int, [c_int, c_int, c_int]),
    "gsa_tile_bin16_boxes": (c_int, [c_int, c_int, c_int, c_int]),
    "gsa_tile_bin16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_uint32]),
    "gsa_tile_bins_supported": (c_int, [c_int, c_int, c_int]),
    "gsa_scan_tiles_rect": (c_int, [c_int, P, P, P, P, P, c_int, P, P, c_size_t, P]),
    "gsa_depth_order_workspace_bytes": (c_size_t, [c_int]),
    "gsa_depth_order": (c_int, [c_int, P, P, c_int, P, P, P, P, P, c_size_t, P]),
    "gsa_depth_order_msd_workspace_bytes": (c_size_t, [c_int]),
    "gsa_depth_order_msd_max_width": (c_int, []),
    "gsa_depth_order_msd_table_offset": (c_size_t, [c_int]),
    "gsa_depth_order_msd_table_words": (c_int, []),
    "gsa_depth_order_msd": (c_int, [c_int, P, P, c_uint32, c_uint32, P, P, P, P, P, c_int, P, P, c_uint32, P, c_size_t, P]),
    "gsa_scan_tiles_sorted": (c_int, [c_int, P, P, c_int, P, P, c_size_t, P]),
    "gsa_scan_tiles_sorted_report": (c_int, [c_int, P, P, c_int, P, P, c_uint32, P, c_size_t, P]),
    "gsa_tile_bin16": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_uint32, P, P, P, P, P, P, P, P, c_size_t, P]),
    "gsa_tile_bin16_scheduled": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_uint32, P, P, P, P, P, P, P, c_size_t, P]),
    "gsa_tile_emit": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_uint32, P, P, P]),
    "gsa_tile_offsets": (c_int, [c_uint32, P, c_int, P, P]),
    "gsa_sort_pairs_u32_dev": (c_int, [c_int, P, P, P, P, P, P, P, c_int, c_int, P, c_size_t, P]),
    "gsa_tile_emit_dev": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_uint32, P, P, P, P]),
    "gsa_tile_offsets_dev": (c_int, [c_uint32, P, P, c_int, P, P]),
    "gsa_composite_fwd": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_longlong, c_int, P]),
    "gsa_composite_bwd_prepare": (c_int, [c_longlong, P, c_int, P, P, P]),
    "gsa_composite_bwd_lists": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P]),
    "gsa_composite_bwd_lists_hot": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, c_longlong, P, c_int, P]),
    "gsa_hot_rows": (c_longlong, []),
    "gsa_composite_bwd_lists_rgb": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P]),
    "gsa_tile_schedule": (c_int, [c_int, P, P, c_int, c_int, P]),
    "gsa_composite_bwd": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P, P, c_int, P]),
    "gsa_densify_gather": (c_int, [c_longlong, c_longlong, c_int, P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P, P]),
    "gsa_view_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gsa_view_loss": (c_int, [c_int, c_int, P, P, c_float, c_float, P, c_size_t, P, P, P]),
    "gsa_adam_step": (c_int, [c_longlong, P, P, P, P, c_int, P, P, c_double, c_double, c_double, c_int, P, P]),
    "gsa_adam_step_striped": (c_int, [c_longlong, P, P, P, P, c_int, P, P, P, P, P, c_double, c_double, c_double, c_int, P, P]),
    "gsa_grad_unscale": (c_int, [c_longlong, P, c_float, P, P]),
    "gsa_step_report": (c_int, [c_int, P, P, c_double, c_int, P, c_longlong, c_double, c_int, P, P]),
    "gsa_view_args_size": (c_size_t, []),
    "gsa_view_fwd": (c_int, [ctypes.POINTER(ViewArgs), P]),
    "gsa_view_bwd": (c_int, [ctypes.POINTER(ViewArgs), P]),
    "gsa_project_bwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, P, P, P, P, P, P, P, P, P, P]),
    "gsa_project_bwd_factored": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P, P]),
    "gsa_project_bwd_batch": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                                      c_int, P]),
    "gsa_project_bwd_batch_geometry": (c_int, [c_int, P, P, P, P, c_int, c_int, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                                               c_int, P]),
    "gsa_sh_grad_expand_split": (c_int, [c_int, P, c_int, c_int, c_int, P, P, P, P, P, P]),
    "gsa_activate_fwd": (c_int, [c_longlong, P, P, P, P, P, P, P, P, P, P]),
    "gsa_activate_partials": (c_size_t, [c_longlong]),
    "gsa_activate_bwd": (c_int, [c_longlong, P, P, P, P, P, P, c_float, c_float, c_int, P, P, P, P, P]),
    "gsa_sh_grad_expand": (c_int, [c_int, P, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "gsa_project_bwd_acc": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P, P]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"gsplat_amd: HIP library not built ({LIB_PATH} missing). Run `python __graft_entry__.py` or "
            "`python 3d-gaussian-splatting_amd/gsplat_amd/build.py` (needs hipcc, gfx950). There is no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    got = lib.gsa_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"gsplat_amd: library ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
    if lib.gsa_view_args_size() != ctypes.sizeof(ViewArgs):
        raise ImportError(f"gsplat_amd: gsa_view_args is {lib.gsa_view_args_size()} bytes in the library, "
                          f"{ctypes.sizeof(ViewArgs)} in the binding; rebuild the library")
    return lib


lib = _load()


class GsaError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.gsa_error_string(code)
        raise GsaError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")
