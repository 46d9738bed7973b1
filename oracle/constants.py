"""Constants of the rasterizer contract -- oracle copy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Mirrors include/gsplat_amd_constants.h;
tests/test_constants.py checks that the two agree.  Source of the values: the
`gsplat.rasterization` defaults behind the reference's call at
/root/reference/src/backends/gsplat_backend.py:168-182.
"""

TILE_SIZE = 16
NEAR_PLANE = 0.01
FAR_PLANE = 1e10
EPS2D = 0.3
RADIUS_CLIP = 0.0
ALPHA_THRESHOLD = 1.0 / 255.0
MAX_ALPHA = 0.999
TRANSMITTANCE_EPS = 1e-4
RADIUS_EXTEND = 3.33
FOV_MARGIN = 0.3
DEPTH_ALPHA_EPS = 1e-10

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (
    1.0925484305920792,
    -1.0925484305920792,
    0.31539156525252005,
    -1.0925484305920792,
    0.5462742152960396,
)
SH_C3 = (
    -0.5900435899266435,
    2.890611442640554,
    -0.4570457994644658,
    0.3731763325901154,
    -0.4570457994644658,
    1.445305721320277,
    -0.5900435899266435,
)
