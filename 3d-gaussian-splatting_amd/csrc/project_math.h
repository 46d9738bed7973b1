// project_math.h -- per-gaussian projection / EWA / SH arithmetic shared by the forward and
// backward projection kernels.  Device-only, fp32.
#pragma once

#include "gsa_common.h"

struct GsaProj {
    // camera-space position and clamp state
    float x, y, z, rz, rz2, tx, ty;
    bool x_in, y_in;
    // rotation from the (normalised) quaternion and its inputs
    float qw, qx, qy, qz, qinv;  // normalised components, 1/|q|
    float Rq[9];
    float sx, sy, sz;
    // camera-space covariance (symmetric): c00 c01 c02 c11 c12 c22
    float S00, S01, S02, S11, S12, S22;
    // Jacobian entries
    float J00, J02, J11, J12;
    // blurred 2D covariance, conic
    float c00, c01, c11, det;
    float qa, qb, qc;
    float mx, my;
    bool ok;  // passed the z and determinant tests
};

__device__ __forceinline__ void gsa_quat_to_rotmat(float w, float x, float y, float z, float *R) {
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Forward projection of one gaussian (A.2).  Leaves radius / culling against the image to the caller.
__device__ __forceinline__ GsaProj gsa_project_one(const GsaCamera &cam, const float *__restrict__ mean,
                                                   const float *__restrict__ quat,
                                                   const float *__restrict__ scale, int width, int height) {
    GsaProj p;
    const float *R = cam.R;
    float m0 = mean[0], m1 = mean[1], m2 = mean[2];
    p.x = R[0] * m0 + R[1] * m1 + R[2] * m2 + cam.t[0];
    p.y = R[3] * m0 + R[4] * m1 + R[5] * m2 + cam.t[1];
    p.z = R[6] * m0 + R[7] * m1 + R[8] * m2 + cam.t[2];
    p.ok = (p.z >= GSA_NEAR_PLANE) && (p.z <= GSA_FAR_PLANE);
    float zs = p.ok ? p.z : 1.f;

    float w = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
    p.qinv = rsqrtf(w * w + qx * qx + qy * qy + qz * qz);
    p.qw = w * p.qinv; p.qx = qx * p.qinv; p.qy = qy * p.qinv; p.qz = qz * p.qinv;
    gsa_quat_to_rotmat(p.qw, p.qx, p.qy, p.qz, p.Rq);
    p.sx = scale[0]; p.sy = scale[1]; p.sz = scale[2];

    // M = Rq diag(s); Sigma = M M^T
    float M[9];
    M[0] = p.Rq[0] * p.sx; M[1] = p.Rq[1] * p.sy; M[2] = p.Rq[2] * p.sz;
    M[3] = p.Rq[3] * p.sx; M[4] = p.Rq[4] * p.sy; M[5] = p.Rq[5] * p.sz;
    M[6] = p.Rq[6] * p.sx; M[7] = p.Rq[7] * p.sy; M[8] = p.Rq[8] * p.sz;
    float V00 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    float V01 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    float V02 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    float V11 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    float V12 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    float V22 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    // Sigma_c = R Sigma R^T ; A = R Sigma first
    float A0 = R[0] * V00 + R[1] * V01 + R[2] * V02, A1 = R[0] * V01 + R[1] * V11 + R[2] * V12,
          A2 = R[0] * V02 + R[1] * V12 + R[2] * V22;
    float B0 = R[3] * V00 + R[4] * V01 + R[5] * V02, B1 = R[3] * V01 + R[4] * V11 + R[5] * V12,
          B2 = R[3] * V02 + R[4] * V12 + R[5] * V22;
    float C0 = R[6] * V00 + R[7] * V01 + R[8] * V02, C1 = R[6] * V01 + R[7] * V11 + R[8] * V12,
          C2 = R[6] * V02 + R[7] * V12 + R[8] * V22;
    p.S00 = A0 * R[0] + A1 * R[1] + A2 * R[2];
    p.S01 = A0 * R[3] + A1 * R[4] + A2 * R[5];
    p.S02 = A0 * R[6] + A1 * R[7] + A2 * R[8];
    p.S11 = B0 * R[3] + B1 * R[4] + B2 * R[5];
    p.S12 = B0 * R[6] + B1 * R[7] + B2 * R[8];
    p.S22 = C0 * R[6] + C1 * R[7] + C2 * R[8];

    float tanx = 0.5f * width / cam.fx, tany = 0.5f * height / cam.fy;
    float lim_xp = (width - cam.cx) / cam.fx + GSA_FOV_MARGIN * tanx;
    float lim_xn = cam.cx / cam.fx + GSA_FOV_MARGIN * tanx;
    float lim_yp = (height - cam.cy) / cam.fy + GSA_FOV_MARGIN * tany;
    float lim_yn = cam.cy / cam.fy + GSA_FOV_MARGIN * tany;
    p.rz = 1.f / zs;
    p.rz2 = p.rz * p.rz;
    float xr = p.x * p.rz, yr = p.y * p.rz;
    p.x_in = (xr <= lim_xp) && (xr >= -lim_xn);
    p.y_in = (yr <= lim_yp) && (yr >= -lim_yn);
    p.tx = zs * fminf(lim_xp, fmaxf(-lim_xn, xr));
    p.ty = zs * fminf(lim_yp, fmaxf(-lim_yn, yr));
    p.J00 = cam.fx * p.rz;
    p.J02 = -cam.fx * p.tx * p.rz2;
    p.J11 = cam.fy * p.rz;
    p.J12 = -cam.fy * p.ty * p.rz2;

    // cov2d = J Sigma_c J^T  (J = [[J00, 0, J02], [0, J11, J12]])
    float c00 = p.J00 * p.J00 * p.S00 + 2.f * p.J00 * p.J02 * p.S02 + p.J02 * p.J02 * p.S22;
    float c01 = p.J00 * p.J11 * p.S01 + p.J00 * p.J12 * p.S02 + p.J02 * p.J11 * p.S12 + p.J02 * p.J12 * p.S22;
    float c11 = p.J11 * p.J11 * p.S11 + 2.f * p.J11 * p.J12 * p.S12 + p.J12 * p.J12 * p.S22;
    p.c00 = c00 + GSA_EPS2D;
    p.c01 = c01;
    p.c11 = c11 + GSA_EPS2D;
    p.det = p.c00 * p.c11 - p.c01 * p.c01;
    p.ok = p.ok && (p.det > 0.f);
    float idet = 1.f / (p.ok ? p.det : 1.f);
    p.qa = p.c11 * idet;
    p.qb = -p.c01 * idet;
    p.qc = p.c00 * idet;
    p.mx = cam.fx * p.x * p.rz + cam.cx;
    p.my = cam.fy * p.y * p.rz + cam.cy;
    return p;
}

// Real SH basis on a unit direction (A.3).  B must hold (deg+1)^2 floats.
template <int DEG>
__device__ __forceinline__ void gsa_sh_basis(float x, float y, float z, float *B) {
    B[0] = GSA_SH_C0;
    if constexpr (DEG >= 1) {
        B[1] = -GSA_SH_C1 * y; B[2] = GSA_SH_C1 * z; B[3] = -GSA_SH_C1 * x;
    }
    if constexpr (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z;
        B[4] = GSA_SH_C2_0 * x * y;
        B[5] = GSA_SH_C2_1 * y * z;
        B[6] = GSA_SH_C2_2 * (2.f * zz - xx - yy);
        B[7] = GSA_SH_C2_3 * x * z;
        B[8] = GSA_SH_C2_4 * (xx - yy);
        if constexpr (DEG >= 3) {
            B[9] = GSA_SH_C3_0 * y * (3.f * xx - yy);
            B[10] = GSA_SH_C3_1 * x * y * z;
            B[11] = GSA_SH_C3_2 * y * (4.f * zz - xx - yy);
            B[12] = GSA_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
            B[13] = GSA_SH_C3_4 * x * (4.f * zz - xx - yy);
            B[14] = GSA_SH_C3_5 * z * (xx - yy);
            B[15] = GSA_SH_C3_6 * x * (xx - 3.f * yy);
        }
    }
}

// d(basis_j)/d(x,y,z) contracted with vB: returns sum_j vB[j] * grad B_j.
template <int DEG>
__device__ __forceinline__ void gsa_sh_basis_vjp(float x, float y, float z, const float *vB, float &gx,
                                                 float &gy, float &gz) {
    gx = gy = gz = 0.f;
    if constexpr (DEG >= 1) {
        gy += -GSA_SH_C1 * vB[1]; gz += GSA_SH_C1 * vB[2]; gx += -GSA_SH_C1 * vB[3];
    }
    if constexpr (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z;
        gx += GSA_SH_C2_0 * y * vB[4];           gy += GSA_SH_C2_0 * x * vB[4];
        gy += GSA_SH_C2_1 * z * vB[5];           gz += GSA_SH_C2_1 * y * vB[5];
        gx += GSA_SH_C2_2 * -2.f * x * vB[6];    gy += GSA_SH_C2_2 * -2.f * y * vB[6];  gz += GSA_SH_C2_2 * 4.f * z * vB[6];
        gx += GSA_SH_C2_3 * z * vB[7];           gz += GSA_SH_C2_3 * x * vB[7];
        gx += GSA_SH_C2_4 * 2.f * x * vB[8];     gy += GSA_SH_C2_4 * -2.f * y * vB[8];
        if constexpr (DEG >= 3) {
            gx += GSA_SH_C3_0 * 6.f * x * y * vB[9];                 gy += GSA_SH_C3_0 * (3.f * xx - 3.f * yy) * vB[9];
            gx += GSA_SH_C3_1 * y * z * vB[10];  gy += GSA_SH_C3_1 * x * z * vB[10];  gz += GSA_SH_C3_1 * x * y * vB[10];
            gx += GSA_SH_C3_2 * -2.f * x * y * vB[11];
            gy += GSA_SH_C3_2 * (4.f * zz - xx - 3.f * yy) * vB[11];
            gz += GSA_SH_C3_2 * 8.f * y * z * vB[11];
            gx += GSA_SH_C3_3 * -6.f * x * z * vB[12];
            gy += GSA_SH_C3_3 * -6.f * y * z * vB[12];
            gz += GSA_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * vB[12];
            gx += GSA_SH_C3_4 * (4.f * zz - 3.f * xx - yy) * vB[13];
            gy += GSA_SH_C3_4 * -2.f * x * y * vB[13];
            gz += GSA_SH_C3_4 * 8.f * x * z * vB[13];
            gx += GSA_SH_C3_5 * 2.f * x * z * vB[14];
            gy += GSA_SH_C3_5 * -2.f * y * z * vB[14];
            gz += GSA_SH_C3_5 * (xx - yy) * vB[14];
            gx += GSA_SH_C3_6 * (3.f * xx - 3.f * yy) * vB[15];
            gy += GSA_SH_C3_6 * -6.f * x * y * vB[15];
        }
    }
}

// ---- tile rectangle and the optional exact ellipse-vs-tile test --------------------------------
struct GsaTileRect { int x0, y0, x1, y1; };

// (tile_size: GSA_TILE_SIZE for the published 16 x 16 tiles; the fused binning may use coarser BIN tiles, see
// gsa_project_fwd's bin_tile)
__device__ __forceinline__ GsaTileRect gsa_tile_rect(float mx, float my, int rx, int ry, int tile_w, int tile_h,
                                                     int tile_size = GSA_TILE_SIZE) {
    const float inv = 1.0f / (float)tile_size;
    float cx = mx * inv, cy = my * inv, trx = rx * inv, try_ = ry * inv;
    GsaTileRect r;
    r.x0 = (int)fminf(fmaxf(floorf(cx - trx), 0.f), (float)tile_w);
    r.x1 = (int)fminf(fmaxf(ceilf(cx + trx), 0.f), (float)tile_w);
    r.y0 = (int)fminf(fmaxf(floorf(cy - try_), 0.f), (float)tile_h);
    r.y1 = (int)fminf(fmaxf(ceilf(cy + try_), 0.f), (float)tile_h);
    return r;
}

// Smallest sigma = 1/2 d^T Q d over the rectangle of pixel CENTRES [x0,x1] x [y0,y1] (d = mean - pixel).
__device__ __forceinline__ float gsa_min_sigma_rect(float mx, float my, float a, float b, float c, float x0,
                                                    float y0, float x1, float y1) {
// no FMA contraction here: the tile-count and tile-emit kernels must take bit-identical decisions
#pragma clang fp contract(off)
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return 0.f;
    float best = 3.0e38f;
    // vertical edges: dx fixed, minimise over dy in [my - y1, my - y0]
    float dlo = my - y1, dhi = my - y0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float dx = mx - (e == 0 ? x0 : x1);
        float dy = fminf(dhi, fmaxf(dlo, -b * dx / c));
        best = fminf(best, 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy);
    }
    float elo = mx - x1, ehi = mx - x0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float dy = my - (e == 0 ? y0 : y1);
        float dx = fminf(ehi, fmaxf(elo, -b * dy / a));
        best = fminf(best, 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy);
    }
    return best;
}

// Which 8x8 quadrants of tile (tx, ty) can the alpha >= 1/255 region of this splat reach?  Bit g (gx = g & 1,
// gy = g >> 1) is set if some pixel CENTRE of the quadrant may pass the per-pixel alpha test.  Conservative:
// a small margin on the threshold keeps every pixel the per-pixel test would accept.  sig_max = ln(255 o).
__device__ __forceinline__ uint32_t gsa_quadrant_mask(float mx, float my, float a, float b, float c, float sig_max,
                                                      int tx, int ty) {
    const float thr = sig_max * 1.0001f + 1e-4f;
    uint32_t m = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float x0 = tx * GSA_TILE_SIZE + (g & 1) * 8 + 0.5f, y0 = ty * GSA_TILE_SIZE + (g >> 1) * 8 + 0.5f;
        float s = gsa_min_sigma_rect(mx, my, a, b, c, x0, y0, x0 + 7.f, y0 + 7.f);
        m |= (s <= thr) ? (1u << g) : 0u;
    }
    return m;
}
