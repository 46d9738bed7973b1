/*
 * gsplat_amd_constants.h -- every numeric constant of the rasterizer contract, in one place.
 *
 * These are the constants of the `gsplat.rasterization(...)` contract as the reference calls it
 * (reference: src/backends/gsplat_backend.py:168-182; all unlisted arguments at gsplat 1.5.3 defaults,
 * SURVEY.md Appendix A).  The CPU oracle (oracle/constants.py) carries the same values and
 * tests/test_constants.py checks that the two files agree, so oracle and kernels cannot drift.
 */
#ifndef GSPLAT_AMD_CONSTANTS_H
#define GSPLAT_AMD_CONSTANTS_H

#define GSA_TILE_SIZE 16               /* tile edge in pixels (gsplat default tile_size=16)            */
#define GSA_NEAR_PLANE 0.01f           /* cull if camera-space z < near                                */
#define GSA_FAR_PLANE 1e10f            /* cull if camera-space z > far                                 */
#define GSA_EPS2D 0.3f                 /* blur added to the diagonal of the 2D covariance              */
#define GSA_RADIUS_CLIP 0.0f           /* cull if both radii <= radius_clip                            */
#define GSA_ALPHA_THRESHOLD (1.0f / 255.0f) /* skip a splat at a pixel if alpha < 1/255               */
#define GSA_MAX_ALPHA 0.999f           /* alpha = min(0.999, opacity * exp(-sigma))                    */
#define GSA_TRANSMITTANCE_EPS 1e-4f    /* stop a pixel when T * (1 - alpha) <= 1e-4                    */
#define GSA_RADIUS_EXTEND 3.33f        /* radius = ceil(min(3.33, sqrt(2 ln(255 o))) * sqrt(cov_ii))   */
#define GSA_FOV_MARGIN 0.3f            /* x/z clamp limit = (W - cx)/fx + 0.3 * (0.5 W / fx)           */
#define GSA_DEPTH_ALPHA_EPS 1e-10f     /* expected depth = D / max(alpha, 1e-10)                       */

/* Real spherical-harmonics basis constants (Inria / 3DGS sign convention). */
#define GSA_SH_C0 0.28209479177387814f
#define GSA_SH_C1 0.4886025119029199f
#define GSA_SH_C2_0 1.0925484305920792f
#define GSA_SH_C2_1 -1.0925484305920792f
#define GSA_SH_C2_2 0.31539156525252005f
#define GSA_SH_C2_3 -1.0925484305920792f
#define GSA_SH_C2_4 0.5462742152960396f
#define GSA_SH_C3_0 -0.5900435899266435f
#define GSA_SH_C3_1 2.890611442640554f
#define GSA_SH_C3_2 -0.4570457994644658f
#define GSA_SH_C3_3 0.3731763325901154f
#define GSA_SH_C3_4 -0.4570457994644658f
#define GSA_SH_C3_5 1.445305721320277f
#define GSA_SH_C3_6 -0.5900435899266435f

#endif /* GSPLAT_AMD_CONSTANTS_H */
