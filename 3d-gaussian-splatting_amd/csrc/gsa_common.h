// gsa_common.h -- shared host/device helpers for the gfx950 rasterizer kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsplat_amd.h"

#define GSA_WAVE 64

// Launch-error plumbing: every ABI function returns int and never throws.
#define GSA_RETURN_IF(cond, code) \
    do {                          \
        if (cond) return (code);  \
    } while (0)

#define GSA_LAUNCH_CHECK()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

static inline int gsa_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Camera constants, loaded once per thread from device memory (uniform addresses -> scalar loads).
struct GsaCamera {
    float R[9];      // world->camera rotation, row-major
    float t[3];      // translation
    float fx, fy, cx, cy;
    float campos[3]; // camera centre in world space: -R^-1 t (general 3x3 inverse, like torch.inverse)
};

__device__ __forceinline__ GsaCamera gsa_load_camera(const float *__restrict__ viewmat,
                                                     const float *__restrict__ intrin) {
    GsaCamera c;
    c.R[0] = viewmat[0]; c.R[1] = viewmat[1]; c.R[2] = viewmat[2];  c.t[0] = viewmat[3];
    c.R[3] = viewmat[4]; c.R[4] = viewmat[5]; c.R[5] = viewmat[6];  c.t[1] = viewmat[7];
    c.R[6] = viewmat[8]; c.R[7] = viewmat[9]; c.R[8] = viewmat[10]; c.t[2] = viewmat[11];
    c.fx = intrin[0]; c.cx = intrin[2]; c.fy = intrin[4]; c.cy = intrin[5];
    // inverse of R by cofactors
    const float *R = c.R;
    float c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
    float det = R[0] * c00 + R[1] * c01 + R[2] * c02;
    float id = 1.0f / det;
    float i00 = c00 * id, i01 = (R[2] * R[7] - R[1] * R[8]) * id, i02 = (R[1] * R[5] - R[2] * R[4]) * id;
    float i10 = c01 * id, i11 = (R[0] * R[8] - R[2] * R[6]) * id, i12 = (R[2] * R[3] - R[0] * R[5]) * id;
    float i20 = c02 * id, i21 = (R[1] * R[6] - R[0] * R[7]) * id, i22 = (R[0] * R[4] - R[1] * R[3]) * id;
    c.campos[0] = -(i00 * c.t[0] + i01 * c.t[1] + i02 * c.t[2]);
    c.campos[1] = -(i10 * c.t[0] + i11 * c.t[1] + i12 * c.t[2]);
    c.campos[2] = -(i20 * c.t[0] + i21 * c.t[1] + i22 * c.t[2]);
    return c;
}
