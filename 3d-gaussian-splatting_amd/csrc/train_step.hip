// train_step.hip -- the two stages that follow the rasterizer in a training step:
//
//   gsa_view_loss   the reference's per-view photometric loss AND its gradient with respect to the render, in one
//                   streaming pass:  loss = (1 - lambda) L1 + lambda (1 - SSIM_3x3)  on the straight-through-clamped
//                   render (/root/reference/src/training/trainer.py:485-492 with _stclamp :102-107 and _compute_ssim
//                   :354-371; PSNR's mse :70-75 comes out of the same pass).  The torch formulation makes ~10
//                   full-image passes forward and as many backward; here render and target are read once.
//   gsa_adam_step   one fused Adam update over the flat 59 N parameter bucket with per-segment learning rates
//                   (torch.optim.Adam as configured at trainer.py:331-340: betas 0.9 / 0.999, eps 1e-15, no weight
//                   decay, no amsgrad): 4 arrays read, 3 written, once.
//
//   gsa_densify_gather   rank 2: clone / split / prune of the parameter bucket AND its Adam moments as one gather pass.
//
// All are HBM-streaming kernels (fp32, no MFMA).

#include "gsa_common.h"

namespace {

constexpr int kThreads = 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// block sum of three doubles -> block_sums[3 * blockIdx.x ..].  (One atomicAdd per block on the three result words was
// the whole cost of the kernel: 8100 blocks hammering three addresses serialise at the memory side at ~90 adds/us.)
__device__ __forceinline__ void block_add3(double a, double b, double c, double *__restrict__ block_sums) {
    __shared__ double red[3][kThreads / GSA_WAVE];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d); b += __shfl_xor(b, d); c += __shfl_xor(c, d);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int w = 0; w < kThreads / GSA_WAVE; ++w) s += red[threadIdx.x][w];
        block_sums[3 * (long)blockIdx.x + threadIdx.x] = s;
    }
}

// sums[k] += sum over blocks of block_sums[3 * b + k]   (one workgroup)
__global__ __launch_bounds__(kThreads) void loss_finalize_kernel(int n_blocks, const double *__restrict__ block_sums,
                                                                 double *__restrict__ sums) {
    __shared__ double red[3][kThreads / GSA_WAVE];
    double acc[3] = {0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += kThreads)
        for (int k = 0; k < 3; ++k) acc[k] += block_sums[3 * (long)b + k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        for (int k = 0; k < 3; ++k) acc[k] += __shfl_xor(acc[k], d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) red[k][wave] = acc[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int w = 0; w < kThreads / GSA_WAVE; ++w) s += red[threadIdx.x][w];
        sums[threadIdx.x] += s;
    }
}

// Both passes work on 32x8-pixel tiles: the tile plus its one-pixel halo is staged through LDS with coalesced
// row reads (each input element is read from HBM/L2 about 1.4 times instead of 9), the 3x3 window then reads LDS.
constexpr int kTileW = 32, kTileH = 8, kHaloW = kTileW + 2, kHaloH = kTileH + 2;

// Stage `nch` floats per pixel of the tile at (x0, y0) with its halo into lds[(yy * kHaloW + xx) * nch + c];
// outside the image: zeros (the zero padding of avg_pool2d).  CLAMP applies clamp(., 0, 1) on the way in.
template <bool CLAMP>
__device__ __forceinline__ void stage_tile(float *__restrict__ lds, const float *__restrict__ src, int nch, int x0, int y0,
                                           int width, int height) {
    const int row_floats = kHaloW * nch;
    for (int i = threadIdx.x; i < kHaloH * row_floats; i += kThreads) {
        const int yy = i / row_floats, rest = i - yy * row_floats;
        const int y = y0 - 1 + yy, xf = (x0 - 1) * nch + rest;       // float offset inside image row y
        float v = 0.f;
        if (y >= 0 && y < height && xf >= 0 && xf < width * nch) {
            v = src[(long)y * width * nch + xf];
            if (CLAMP) v = clamp01(v);
        }
        lds[i] = v;
    }
}

// One pass: a block stages its 32x8 tile of a = clamp(render) and b = target with a TWO-pixel halo, computes the 3x3
// box statistics (zero padding, divisor 9 -- avg_pool2d(3, 1, padding=1) with count_include_pad) and from them the SSIM
// value and its partial derivatives with respect to mu_a, E[a^2], E[ab] for the tile plus its ONE-pixel halo into LDS,
// then every tile pixel applies the adjoint of the box filter to those partials:
//   v_render = grad_scale * [ (1 - lambda) sign(a - b) - lambda * d SSIM ] / (3 H W)     (straight-through clamp).
// The partials never travel through HBM (as a two-kernel version they were 36 B per pixel out and back).
constexpr int kHalo2W = kTileW + 4, kHalo2H = kTileH + 4;

template <bool CLAMP>
__device__ __forceinline__ void stage_tile2(float *__restrict__ lds, const float *__restrict__ src, int x0, int y0, int width,
                                            int height) {
    const int row_floats = kHalo2W * 3;
    for (int i = threadIdx.x; i < kHalo2H * row_floats; i += kThreads) {
        const int yy = i / row_floats, rest = i - yy * row_floats;
        const int y = y0 - 2 + yy, xf = (x0 - 2) * 3 + rest;
        float v = 0.f;
        if (y >= 0 && y < height && xf >= 0 && xf < width * 3) {
            v = src[(long)y * width * 3 + xf];
            if (CLAMP) v = clamp01(v);
        }
        lds[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void loss_fused_kernel(int width, int height, const float *__restrict__ render,
                                                              const float *__restrict__ target, float lambda_dssim,
                                                              float grad_scale, float *__restrict__ v_render,
                                                              double *__restrict__ block_sums) {
    __shared__ float sa[kHalo2H * kHalo2W * 3], sb[kHalo2H * kHalo2W * 3];
    __shared__ float sp[kHaloH * kHaloW * 9];
    const int tiles_x = (width + kTileW - 1) / kTileW;
    const int x0 = (blockIdx.x % tiles_x) * kTileW, y0 = (blockIdx.x / tiles_x) * kTileH;
    stage_tile2<true>(sa, render, x0, y0, width, height);
    stage_tile2<false>(sb, target, x0, y0, width, height);
    __syncthreads();
    double l1 = 0, ssim = 0, mse = 0;
    // partials of the tile and its one-pixel halo (positions outside the image have no window: zeros)
    for (int i = threadIdx.x; i < kHaloH * kHaloW; i += kThreads) {
        const int hy = i / kHaloW, hx = i - hy * kHaloW;
        const int x = x0 - 1 + hx, y = y0 - 1 + hy;
        float *out = sp + i * 9;
        if (x < 0 || x >= width || y < 0 || y >= height) {
#pragma unroll
            for (int k = 0; k < 9; ++k) out[k] = 0.f;
            continue;
        }
        const bool interior = hx >= 1 && hx <= kTileW && hy >= 1 && hy <= kTileH;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s_a = 0.f, s_b = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int q = ((hy + dy) * kHalo2W + hx + dx) * 3 + c;
                    const float a = sa[q], b = sb[q];
                    s_a += a; s_b += b; saa += a * a; sbb += b * b; sab += a * b;
                }
            const float inv9 = 1.f / 9.f;
            const float mu_a = s_a * inv9, mu_b = s_b * inv9;
            const float var_a = saa * inv9 - mu_a * mu_a, var_b = sbb * inv9 - mu_b * mu_b;
            const float cov = sab * inv9 - mu_a * mu_b;
            const float A1 = 2.f * mu_a * mu_b + kC1, A2 = 2.f * cov + kC2;
            const float B1 = mu_a * mu_a + mu_b * mu_b + kC1, B2 = var_a + var_b + kC2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            // d S / d mu_a with E[a^2], E[ab] held fixed: var_a and cov depend on mu_a too
            const float dA1 = 2.f * mu_b, dA2 = -2.f * mu_b, dB1 = 2.f * mu_a, dB2 = -2.f * mu_a;
            out[c * 3 + 0] = ((dA1 * A2 + A1 * dA2) - S * (dB1 * B2 + B1 * dB2)) * inv;
            out[c * 3 + 1] = -S / B2;            // B2 contains + E[a^2]
            out[c * 3 + 2] = 2.f * A1 * inv;     // A2 contains + 2 E[ab]
            if (interior) {
                const int q0 = ((hy + 1) * kHalo2W + hx + 1) * 3 + c;
                const float d0 = sa[q0] - sb[q0];
                l1 += fabsf(d0); mse += (double)d0 * d0; ssim += S;
            }
        }
    }
    block_add3(l1, ssim, mse, block_sums);   // (contains the barrier that publishes sp)
    if (!v_render) return;
    const int lx = threadIdx.x % kTileW, ly = threadIdx.x / kTileW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= width || y >= height) return;
    const long p = (long)y * width + x;
    const float inv_count = grad_scale / (3.f * (float)width * (float)height);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s_mu = 0.f, s_aa = 0.f, s_ab = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float *in = sp + (((ly + dy) * kHaloW + lx + dx) * 3 + c) * 3;
                s_mu += in[0]; s_aa += in[1]; s_ab += in[2];
            }
        const int q0 = ((ly + 2) * kHalo2W + lx + 2) * 3 + c;
        const float a = sa[q0], b = sb[q0];
        const float d_ssim = (s_mu + 2.f * a * s_aa + b * s_ab) * (1.f / 9.f);
        const float d = a - b;
        const float sign = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        v_render[p * 3 + c] = inv_count * ((1.f - lambda_dssim) * sign - lambda_dssim * d_ssim);
    }
}

// One Adam step over a flat bucket.  seg_end[k] = one past the last element of segment k (ascending), lr[k] its
// learning rate.  Mirrors torch.optim.Adam's single-tensor arithmetic:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
struct AdamSegments {
    long end[8];
    float lr[8];
    int count;
};

__global__ __launch_bounds__(kThreads) void adam_kernel(long n, float *__restrict__ param, const float *__restrict__ grad,
                                                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                        AdamSegments seg, float w1, float beta2, float w2, float eps,
                                                        float inv_bc1, float inv_sqrt_bc2,   // w = 1 - beta
                                                        const uint32_t *__restrict__ skip_if_nonzero) {
    if (skip_if_nonzero && *skip_if_nonzero) return;   // a non-finite gradient was found: the step is skipped whole
    const long i0 = ((long)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (i0 + 4 <= n) {
        const float4 g4 = *reinterpret_cast<const float4 *>(grad + i0);
        float4 p4 = *reinterpret_cast<float4 *>(param + i0);
        float4 m4 = *reinterpret_cast<float4 *>(exp_avg + i0);
        float4 v4 = *reinterpret_cast<float4 *>(exp_avg_sq + i0);
        float *pp = &p4.x, *mm = &m4.x, *vv = &v4.x;
        const float *gg = &g4.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lr = 0.f;
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if (k < seg.count && i0 + j < seg.end[k]) lr = seg.lr[k];
            const float g = gg[j];
            const float m = mm[j] + (g - mm[j]) * w1;
            const float v = vv[j] * beta2 + w2 * g * g;
            const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
            pp[j] = pp[j] - (lr * inv_bc1) * (m / denom);
            mm[j] = m; vv[j] = v;
        }
        *reinterpret_cast<float4 *>(param + i0) = p4;
        *reinterpret_cast<float4 *>(exp_avg + i0) = m4;
        *reinterpret_cast<float4 *>(exp_avg_sq + i0) = v4;
    } else {
        for (long i = i0; i < n; ++i) {
            float lr = 0.f;
            for (int k = seg.count - 1; k >= 0; --k)
                if (i < seg.end[k]) lr = seg.lr[k];
            const float g = grad[i];
            const float m = exp_avg[i] + (g - exp_avg[i]) * w1;
            const float v = exp_avg_sq[i] * beta2 + w2 * g * g;
            const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
            param[i] = param[i] - (lr * inv_bc1) * (m / denom);
            exp_avg[i] = m; exp_avg_sq[i] = v;
        }
    }
}

// Densify / prune as ONE gather pass (gaussian_model.py:110-366 does it as ~20 index / cat operations per
// event): output row j of every field copies row src_row[j] of the old bucket.  kind[j]: 0 = survivor (Adam
// moments travel with it), 1 = clone (moments zero), 2 = split child (moments zero, log-scale lowered by
// log(1.6), position moved by noise * new scale -- gaussian_model.py:296-303).  Buckets are field-major:
// field f occupies [off_f * n, (off_f + w_f) * n) with off_f = sum of the widths before it.
struct FieldLayout {
    int width[8], offset[8];   // floats per gaussian of field f; sum of the widths before it
    int count, total, xyz, scaling;
};

__global__ __launch_bounds__(kThreads) void densify_gather_kernel(
    long n_old, long n_new, FieldLayout lay, float log_split, const int32_t *__restrict__ src_row,
    const uint8_t *__restrict__ kind, const float *__restrict__ noise, const float *__restrict__ old_p,
    const float *__restrict__ old_m, const float *__restrict__ old_v, float *__restrict__ new_p,
    float *__restrict__ new_m, float *__restrict__ new_v) {
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;
    if (e >= n_new * lay.total) return;
    // e enumerates the NEW bucket in memory order: field-major, then row, then component
    int f = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < lay.count && e >= (long)lay.offset[k] * n_new) f = k;
    const long within = e - (long)lay.offset[f] * n_new;
    const int w = lay.width[f];
    const long j = within / w;
    const int c = (int)(within - j * w);
    const long src = src_row[j];
    const int kd = kind[j];
    const long so = (long)lay.offset[f] * n_old + src * w + c;
    float val = old_p[so];
    if (kd == 2) {
        if (f == lay.scaling) val -= log_split;
        if (f == lay.xyz) {
            const float sc = old_p[(long)lay.offset[lay.scaling] * n_old + src * lay.width[lay.scaling] + c] - log_split;
            val += noise[j * 3 + c] * expf(sc);
        }
    }
    new_p[e] = val;
    if (new_m) {
        new_m[e] = kd == 0 ? old_m[so] : 0.f;
        new_v[e] = kd == 0 ? old_v[so] : 0.f;
    }
}

// ---- activations of the raw parameters and their derivatives -------------------
// The reference activates per render (gaussian_model.py:86-108 via trainer.py:399-406): scales = exp(scaling),
// quats = normalize(rotation), opacities = sigmoid(opacity), shs = cat(features_dc, features_rest).  In torch that is
// five kernels forward and a dozen backward per step; here one each way.

// thread q < 12 n: one float4 of the [N,16,3] SH table (48 floats per gaussian, so a float4 never straddles two
// gaussians); thread i < n additionally the geometry of gaussian i
__global__ __launch_bounds__(kThreads) void activate_fwd_kernel(long n, const float *__restrict__ scaling,
                                                                const float *__restrict__ rotation,
                                                                const float *__restrict__ opacity,
                                                                const float *__restrict__ f_dc,
                                                                const float *__restrict__ f_rest,
                                                                float *__restrict__ scales, float *__restrict__ quats,
                                                                float *__restrict__ opac, float *__restrict__ shs) {
    const long q = (long)blockIdx.x * kThreads + threadIdx.x;
    if (q < 12 * n) {
        const long g = q / 12;
        const int j0 = (int)(q - g * 12) * 4;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            v[u] = j < 3 ? f_dc[g * 3 + j] : f_rest[g * 45 + (j - 3)];
        }
        reinterpret_cast<float4 *>(shs)[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (q < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) scales[q * 3 + k] = expf(scaling[q * 3 + k]);
        // (scalar loads: the rotation segment of the flat bucket starts at float 55 n -- 16-B aligned only if 4 | n)
        const float4 r = make_float4(rotation[q * 4], rotation[q * 4 + 1], rotation[q * 4 + 2], rotation[q * 4 + 3]);
        const float inv = 1.0f / fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);   // F.normalize's eps
        *reinterpret_cast<float4 *>(quats + q * 4) = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
        opac[q] = 1.0f / (1.0f + expf(-opacity[q]));
    }
}

// raw-parameter gradients from the activated ones, plus the scale-ratio regulariser (trainer.py:434-445):
//   reg = weight * mean(max(ratio, max_ratio) - max_ratio),  ratio = max(scales) / max(min(scales), 1e-8)
// Its hinge sum goes to reg_partials[block] (double); its gradient is added when add_reg_grad (rank 0 only in the
// view-parallel step).  Ties of the max / min share the gradient evenly, as torch's amax / amin do.
__global__ __launch_bounds__(kThreads) void activate_bwd_kernel(long n, const float *__restrict__ scaling,
                                                                const float *__restrict__ rotation,
                                                                const float *__restrict__ opacity,
                                                                const float *__restrict__ v_scales,
                                                                const float *__restrict__ v_quats,
                                                                const float *__restrict__ v_opac, float max_ratio,
                                                                float reg_scale, int add_reg_grad,
                                                                float *__restrict__ v_scaling, float *__restrict__ v_rotation,
                                                                float *__restrict__ v_opacity, double *__restrict__ reg_partials) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    double hinge = 0.0;
    if (i < n) {
        float s[3], gs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { s[k] = expf(scaling[i * 3 + k]); gs[k] = v_scales[i * 3 + k]; }
        const float smax = fmaxf(s[0], fmaxf(s[1], s[2])), smin = fminf(s[0], fminf(s[1], s[2]));
        const float smin_c = fmaxf(smin, 1e-8f);
        const float ratio = smax / smin_c;
        hinge = (double)(fmaxf(ratio, max_ratio) - max_ratio);
        if (add_reg_grad && ratio >= max_ratio) {   // clamp(min=) passes the gradient at equality, like torch
            const float n_max = (float)((s[0] == smax) + (s[1] == smax) + (s[2] == smax));
            const float n_min = (float)((s[0] == smin) + (s[1] == smin) + (s[2] == smin));
            const float d_max = reg_scale / smin_c / n_max;
            const float d_min = smin >= 1e-8f ? -reg_scale * smax / (smin_c * smin_c) / n_min : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) gs[k] += (s[k] == smax ? d_max : 0.f) + (s[k] == smin ? d_min : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) v_scaling[i * 3 + k] = gs[k] * s[k];
        const float4 r = make_float4(rotation[i * 4], rotation[i * 4 + 1], rotation[i * 4 + 2], rotation[i * 4 + 3]);
        const float4 g = *reinterpret_cast<const float4 *>(v_quats + i * 4);
        const float inv = 1.0f / fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
        const float qx = r.x * inv, qy = r.y * inv, qz = r.z * inv, qw = r.w * inv;
        const float dot = g.x * qx + g.y * qy + g.z * qz + g.w * qw;
        v_rotation[i * 4] = (g.x - dot * qx) * inv; v_rotation[i * 4 + 1] = (g.y - dot * qy) * inv;
        v_rotation[i * 4 + 2] = (g.z - dot * qz) * inv; v_rotation[i * 4 + 3] = (g.w - dot * qw) * inv;
        const float o = 1.0f / (1.0f + expf(-opacity[i]));
        v_opacity[i] = v_opac[i] * (o * (1.0f - o));
    }
    block_add3(hinge, 0.0, 0.0, reg_partials);   // -> reg_partials[3 * block] (the other two words stay zero)
}

}  // namespace

extern "C" int gsa_densify_gather(long long n_old, long long n_new, int n_fields, const int *field_width, int xyz_field,
                                  int scaling_field, float log_split, const int32_t *src_row, const uint8_t *kind,
                                  const float *noise, const float *old_param, const float *old_exp_avg,
                                  const float *old_exp_avg_sq, float *new_param, float *new_exp_avg,
                                  float *new_exp_avg_sq, gsa_stream_t stream) {
    GSA_RETURN_IF(n_old < 0 || n_new < 0 || n_fields < 1 || n_fields > 8 || !field_width, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(xyz_field < 0 || xyz_field >= n_fields || scaling_field < 0 || scaling_field >= n_fields,
                  GSA_ERR_INVALID_ARGUMENT);
    if (n_new == 0) return GSA_OK;
    GSA_RETURN_IF(!src_row || !kind || !noise || !old_param || !new_param, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((old_exp_avg == nullptr) != (new_exp_avg == nullptr) || (old_exp_avg == nullptr) != (old_exp_avg_sq == nullptr) ||
                      (new_exp_avg == nullptr) != (new_exp_avg_sq == nullptr), GSA_ERR_INVALID_ARGUMENT);
    FieldLayout lay;
    lay.count = n_fields; lay.xyz = xyz_field; lay.scaling = scaling_field;
    int run = 0;
    for (int k = 0; k < 8; ++k) {
        lay.width[k] = k < n_fields ? field_width[k] : 1;
        lay.offset[k] = run;
        if (k < n_fields) {
            GSA_RETURN_IF(field_width[k] <= 0, GSA_ERR_INVALID_ARGUMENT);
            run += field_width[k];
        }
    }
    lay.total = run;
    GSA_RETURN_IF(field_width[xyz_field] != 3 || field_width[scaling_field] != 3, GSA_ERR_INVALID_ARGUMENT);
    const int grid = gsa_div_up(n_new * run, kThreads);
    densify_gather_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        (long)n_old, (long)n_new, lay, log_split, src_row, kind, noise, old_param, old_exp_avg, old_exp_avg_sq, new_param,
        new_exp_avg, new_exp_avg_sq);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" size_t gsa_view_loss_workspace_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const size_t blocks = (size_t)gsa_div_up(width, kTileW) * gsa_div_up(height, kTileH);
    return blocks * 3 * sizeof(double);   // the per-block sums
}

extern "C" int gsa_view_loss(int width, int height, const float *render, const float *target, float lambda_dssim,
                             float grad_scale, void *workspace, size_t workspace_bytes, float *v_render, double *sums,
                             gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || !render || !target || !workspace || !sums, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_view_loss_workspace_bytes(width, height), GSA_ERR_WORKSPACE_TOO_SMALL);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(workspace) & 7) != 0, GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = gsa_div_up(width, kTileW) * gsa_div_up(height, kTileH);
    double *block_sums = static_cast<double *>(workspace);
    loss_fused_kernel<<<grid, kThreads, 0, s>>>(width, height, render, target, lambda_dssim, grad_scale, v_render,
                                               block_sums);
    GSA_LAUNCH_CHECK();
    loss_finalize_kernel<<<1, kThreads, 0, s>>>(grid, block_sums, sums);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_adam_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                             int n_segments, const long long *segment_end, const float *segment_lr, double beta1,
                             double beta2, double eps, int step, const uint32_t *skip_if_nonzero, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || n_segments < 1 || n_segments > 8 || step < 1, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!param || !grad || !exp_avg || !exp_avg_sq || !segment_end || !segment_lr, GSA_ERR_INVALID_ARGUMENT);
    AdamSegments seg;
    seg.count = n_segments;
    long long prev = 0;
    for (int k = 0; k < 8; ++k) {
        seg.end[k] = k < n_segments ? (long)segment_end[k] : (long)n;
        seg.lr[k] = k < n_segments ? segment_lr[k] : 0.f;
        if (k < n_segments) {
            GSA_RETURN_IF(segment_end[k] < prev, GSA_ERR_INVALID_ARGUMENT);
            prev = segment_end[k];
        }
    }
    GSA_RETURN_IF(prev != n, GSA_ERR_INVALID_ARGUMENT);
    // 1 - beta and the bias corrections in double on the host, as torch computes them in Python floats
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const int grid = gsa_div_up((n + 3) / 4, kThreads);
    adam_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>((long)n, param, grad, exp_avg, exp_avg_sq, seg,
                                                                        (float)(1.0 - beta1), (float)beta2,
                                                                        (float)(1.0 - beta2), (float)eps,
                                                                        (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)),
                                                                        skip_if_nonzero);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

// Loss-scale bookkeeping of a mixed-precision step (torch's GradScaler.unscale_, trainer.py:505-511): grad *= inv_scale
// in place; *found_inf (accumulated: zero it first) becomes 1 if any element is not finite.
__global__ __launch_bounds__(kThreads) void unscale_kernel(long n, float *__restrict__ grad, float inv_scale,
                                                           uint32_t *__restrict__ found_inf) {
    const long i0 = ((long)blockIdx.x * kThreads + threadIdx.x) * 4;
    bool bad = false;
    if (i0 + 4 <= n) {
        float4 g = *reinterpret_cast<float4 *>(grad + i0);
        bad = !(isfinite(g.x) && isfinite(g.y) && isfinite(g.z) && isfinite(g.w));
        g.x *= inv_scale; g.y *= inv_scale; g.z *= inv_scale; g.w *= inv_scale;
        *reinterpret_cast<float4 *>(grad + i0) = g;
    } else {
        for (long i = i0; i < n; ++i) {
            const float g = grad[i];
            bad = bad || !isfinite(g);
            grad[i] = g * inv_scale;
        }
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1u);
}

extern "C" int gsa_grad_unscale(long long n, float *grad, float inv_scale, uint32_t *found_inf, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || !found_inf, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!grad || (reinterpret_cast<uintptr_t>(grad) & 15) != 0, GSA_ERR_INVALID_ARGUMENT);
    const int grid = gsa_div_up((n + 3) / 4, kThreads);
    unscale_kernel<<<grid, kThreads, 0, static_cast<hipStream_t>(stream)>>>((long)n, grad, inv_scale, found_inf);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" size_t gsa_activate_partials(long long n) {   // doubles gsa_activate_bwd's reg_partials must hold
    return n <= 0 ? 0 : 3 * (size_t)gsa_div_up(n, kThreads);
}

extern "C" int gsa_activate_fwd(long long n, const float *scaling, const float *rotation, const float *opacity,
                                const float *features_dc, const float *features_rest, float *scales, float *quats,
                                float *opacities, float *shs, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!scaling || !rotation || !opacity || !features_dc || !features_rest || !scales || !quats ||
                      !opacities || !shs,
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(((reinterpret_cast<uintptr_t>(quats) | reinterpret_cast<uintptr_t>(shs)) & 15) != 0,
                  GSA_ERR_INVALID_ARGUMENT);
    activate_fwd_kernel<<<gsa_div_up(12 * n, kThreads), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        (long)n, scaling, rotation, opacity, features_dc, features_rest, scales, quats, opacities, shs);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_activate_bwd(long long n, const float *scaling, const float *rotation, const float *opacity,
                                const float *v_scales, const float *v_quats, const float *v_opacities,
                                float reg_max_ratio, float reg_weight, int add_reg_grad, float *v_scaling,
                                float *v_rotation, float *v_opacity, double *reg_partials, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!scaling || !rotation || !opacity || !v_scales || !v_quats || !v_opacities || !v_scaling ||
                      !v_rotation || !v_opacity || !reg_partials,
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(v_quats) & 15) != 0, GSA_ERR_INVALID_ARGUMENT);
    // d reg / d ratio of one gaussian: weight / n  (the mean)
    activate_bwd_kernel<<<gsa_div_up(n, kThreads), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        (long)n, scaling, rotation, opacity, v_scales, v_quats, v_opacities, reg_max_ratio,
        (float)((double)reg_weight / (double)n), add_reg_grad ? 1 : 0, v_scaling, v_rotation, v_opacity, reg_partials);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}
