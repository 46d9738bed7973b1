"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/gsplat_amd.h declares, the ctypes binding covers exactly those, and the argument checks
answer with error codes (no compute call is made without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsplat_amd.h")


def _declared():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(gsa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import gsplat_amd
    from gsplat_amd import _lib

    names = _declared()
    assert len(names) >= 11
    raw = ctypes.CDLL(gsplat_amd.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in gsplat_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert gsplat_amd.lib.gsa_abi_version() == gsplat_amd.ABI_VERSION == 14


def test_header_cites_reference_boundary():
    text = open(HEADER).read()
    assert "src/backends/gsplat_backend.py:168-182" in text


def test_error_codes_and_strings():
    from gsplat_amd._lib import GsaError, check, lib

    assert lib.gsa_error_string(0) == b"ok"
    assert b"invalid" in lib.gsa_error_string(-1)
    assert b"workspace" in lib.gsa_error_string(-2)
    # argument validation happens on the host before any launch
    assert lib.gsa_project_fwd(-1, None, None, None, None, None, 3, 16, None, None, 64, 64, 0, None, None, None, None, None, None, 0, None, None, None) == -1
    assert lib.gsa_project_fwd(10, None, None, None, None, None, 3, 16, None, None, 64, 64, 0, None, None, None, None, None, None, 0, None, None, None) == -1
    assert lib.gsa_project_fwd(10, None, None, None, None, None, 7, 16, None, None, 64, 64, 0, None, None, None, None, None, None, 0, None, None, None) == -1
    assert lib.gsa_composite_fwd(64, 64, 5, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, 0, None) == -1
    assert lib.gsa_composite_fwd(64, 64, 3, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, 24, None) == -1   # bin_tile
    assert lib.gsa_composite_fwd(64, 64, 3, None, None, 1, None, None, None, None, None, None, None, None, None, None, 0, 64, None) == -1   # boxes: bins of 16 or 32 only
    assert lib.gsa_tile_schedule(0, None, None, 0, 0, None) == -1
    assert lib.gsa_sort_pairs_u32(16, None, None, None, None, None, None, 0, 40, None, 0, None) == -1
    assert lib.gsa_sort_workspace_bytes(1_000_000) > 0
    # an 8K frame has too many 16 x 16 tiles for the 16-bit tile ids of the fused binning, but not 32 x 32 bins
    assert lib.gsa_tile_bin16_supported(1920, 1080) == 1 and lib.gsa_tile_bin16_supported(7680, 4320) == 0
    assert lib.gsa_tile_bins_supported(7680, 4320, 32) == 1 and lib.gsa_tile_bins_supported(7680, 4320, 24) == 0
    # the loss kernel addresses image rows with 32-bit byte offsets: 12 H W >= 4 GiB is refused (fake non-null pointers:
    # nothing is dereferenced before the checks)
    fake, big = ctypes.c_void_p(4096), 20000
    ws = lib.gsa_view_loss_workspace_bytes(big, big)
    assert lib.gsa_view_loss(big, big, fake, fake, 0.2, 1.0, fake, ws, fake, fake, None) == -1
    assert lib.gsa_view_loss(64, 64, fake, fake, 0.2, 1.0, fake, 8, fake, fake, None) == -2   # workspace too small
    with pytest.raises(GsaError):
        check(-1, "x")


def test_view_calls_guard_their_argument_struct():
    """ABI 13: gsa_view_fwd / gsa_view_bwd take ONE struct; a host built against another layout must be refused before
    anything is queued (struct_bytes), and so must a struct without the buffers the chain needs."""
    from gsplat_amd._lib import ViewArgs, lib

    assert lib.gsa_view_args_size() == ctypes.sizeof(ViewArgs)
    assert lib.gsa_view_fwd(None, None) == -1 and lib.gsa_view_bwd(None, None) == -1
    a = ViewArgs()
    a.struct_bytes = ctypes.sizeof(ViewArgs) - 8        # another layout
    a.n, a.m_capacity = 100, 4096
    assert lib.gsa_view_fwd(ctypes.byref(a), None) == -1 and lib.gsa_view_bwd(ctypes.byref(a), None) == -1
    a.struct_bytes = ctypes.sizeof(ViewArgs)
    a.n = 0                                             # nothing to order
    assert lib.gsa_view_fwd(ctypes.byref(a), None) == -1
    a.n, a.m_capacity = 100, 0                          # no record capacity: the caller has to run the stages
    assert lib.gsa_view_fwd(ctypes.byref(a), None) == -1
    a.m_capacity = 4096                                 # no launch-order buffer (tile_order): not the scheduled placement
    assert lib.gsa_view_fwd(ctypes.byref(a), None) == -1
    assert lib.gsa_view_bwd(ctypes.byref(a), None) == -1   # no lists / gradient buffers


def test_product_does_not_import_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "3d-gaussian-splatting_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree only exists in the authoring container")
def test_reference_backend_binds_to_this_package_by_module_name():
    """INTEGRATION.md level 1: with ``install_as_gsplat()`` the reference's own ``GSplatBackend``
    (src/backends/gsplat_backend.py:4,9) picks up this package's ``rasterization`` -- and a CPU call fails loudly
    instead of falling back."""
    import sys

    import torch

    import gsplat_amd

    saved = {k: sys.modules.get(k) for k in ("gsplat", "src", "src.backends", "src.backends.gsplat_backend")}
    sys.path.insert(0, "/root/reference")
    try:
        gsplat_amd.install_as_gsplat()
        for k in ("src", "src.backends", "src.backends.gsplat_backend"):
            sys.modules.pop(k, None)
        from src.backends.gsplat_backend import GSplatBackend

        be = GSplatBackend(torch.eye(3))
        assert be.rasterization is gsplat_amd.rasterization
        params = {"means3D": torch.zeros(4, 3), "scales": torch.ones(4, 3), "rotations": torch.ones(4, 4),
                  "opacities": torch.ones(4, 1), "shs": torch.zeros(4, 16, 3)}
        vp = {"world_view_transform": torch.eye(4), "image_width": 32, "image_height": 32}
        with pytest.raises(RuntimeError, match="no CPU path"):
            be.render_with_depth(params, vp, bg_color=torch.zeros(3), render_mode="RGB", device="cpu", sh_degree=0)
    finally:
        sys.path.remove("/root/reference")
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
