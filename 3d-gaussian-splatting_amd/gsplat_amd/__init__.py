"""gsplat_amd -- host side of the MI355X-native differentiable 3D-Gaussian rasterizer.

Two drop-in surfaces over one C-ABI HIP library (``include/gsplat_amd.h``):

* operator: ``rasterization(...)`` -- the call at /root/reference/src/backends/gsplat_backend.py:168-182;
  ``install_as_gsplat()`` registers this package under the module name ``gsplat`` so that the
  reference's ``from gsplat import rasterization`` (gsplat_backend.py:4,
  /root/reference/src/gaussians/gaussian_rasterizer.py:106) binds to it with no source change;
* plugin: ``HipBackend(K).render_with_depth(...)`` -- the ``src/backends`` class surface.

Importing this package loads the HIP library and raises ImportError if it is not built.
"""

import sys as _sys

from . import harness, scene, stages  # noqa: F401
from ._lib import ABI_VERSION, LIB_PATH, GsaError, lib  # noqa: F401
from .backend import HipBackend  # noqa: F401
from .rasterization import project_cameras, project_cameras_backward, projection_buffers, rasterization  # noqa: F401

__all__ = ["rasterization", "project_cameras", "project_cameras_backward", "projection_buffers", "HipBackend", "install_as_gsplat", "scene", "stages", "harness", "lib", "LIB_PATH", "ABI_VERSION"]


def install_as_gsplat() -> None:
    """Make ``import gsplat`` / ``from gsplat import rasterization`` resolve to this package."""
    _sys.modules["gsplat"] = _sys.modules[__name__]
