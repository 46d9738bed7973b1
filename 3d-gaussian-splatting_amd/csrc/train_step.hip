// train_step.hip -- the two stages that follow the rasterizer in a training step:
//
//   gsa_view_loss   the reference's per-view photometric loss AND its gradient with respect to the render, in one
//                   streaming pass:  loss = (1 - lambda) L1 + lambda (1 - SSIM_3x3)  on the straight-through-clamped
//                   render (/root/reference/src/training/trainer.py:485-492 with _stclamp :102-107 and _compute_ssim
//                   :354-371; PSNR's mse :70-75 comes out of the same pass).  The torch formulation makes ~10
//                   full-image passes forward and as many backward; here render and target are read once (1.2 times,
//                   counting the strips' halos) and nothing passes through LDS.
//   gsa_adam_step   one fused Adam update over the flat 59 N parameter bucket with per-segment learning rates
//                   (torch.optim.Adam as configured at trainer.py:331-340: betas 0.9 / 0.999, eps 1e-15, no weight
//                   decay, no amsgrad): 4 arrays read, 3 written, once.
//
//   gsa_densify_gather   rank 2: clone / split / prune of the parameter bucket AND its Adam moments as one gather pass.
//
// All are HBM-streaming kernels (fp32, no MFMA).

#include <type_traits>

#include "gsa_common.h"

namespace {

constexpr int kThreads = 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// block sum of three doubles -> block_sums[3 * blockIdx.x ..]  (gsa_activate_bwd's regulariser value; one atomicAdd per
// block on a few result words serialises at the memory side at ~90 adds/us)
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


// sums[k] += sum over blocks of block_sums[3 * b + k]   (one workgroup of 1024; a thread's loads are issued together: the
// 256-thread loop of dependent loads this replaces took 7 us for a 1080p view's 3744 entries)
constexpr int kFinalizeThreads = 1024, kFinalizeBatch = 4;
__global__ __launch_bounds__(kFinalizeThreads) void loss_finalize_kernel(int n_blocks, const double *__restrict__ block_sums,
                                                                         double *__restrict__ sums) {
    __shared__ double red[3][kFinalizeThreads / GSA_WAVE];
    double acc[3] = {0, 0, 0};
    for (int b0 = threadIdx.x; b0 < n_blocks; b0 += kFinalizeBatch * kFinalizeThreads) {
        double v[kFinalizeBatch][3];
#pragma unroll
        for (int j = 0; j < kFinalizeBatch; ++j) {
            const int b = b0 + j * kFinalizeThreads;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[j][k] = b < n_blocks ? block_sums[3 * (long)b + k] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kFinalizeBatch; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] += v[j][k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        for (int k = 0; k < 3; ++k) acc[k] += __shfl_xor(acc[k], d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) red[k][wave] = acc[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int w = 0; w < kFinalizeThreads / GSA_WAVE; ++w) s += red[threadIdx.x][w];
        sums[threadIdx.x] += s;
    }
}

// ---- the per-view loss and its gradient: one wave per (image strip, colour channel), sliding windows in registers ----
// r01-r03 staged 32 x 8 tiles with halos through LDS and read every 3 x 3 window out of it: ~150 LDS reads per pixel, 75 us for
// the 75 MB a 1080p view moves (1.0 TB/s; profiles/r03_train_step_summary.md).  Both box filters of the loss are separable
// -- the 3 x 3 statistics behind SSIM, and the adjoint of that filter applied to SSIM's partial derivatives -- so a lane
// that walks DOWN one image column needs, per row, only the row's horizontal 3-sums (its two neighbours' values: two
// whole-wave DPP shifts, no LDS) and keeps the last two rows' sums in registers:
//   lane l of a wave = column x0 + l - 2 of the strip (x0 = 60 strip): 64 columns loaded, 62 with statistics, 60 written;
//   the three waves of a workgroup = the three colour channels of the same strip (their strided loads and stores meet
//   in the CU's L1 / the XCD's L2 as whole lines);
//   rows y0 - 2 .. y0 + R + 1 are read (R = kStripRows), statistics exist for y0 - 1 .. y0 + R, gradients for y0 .. y0 + R - 1:
//   row i gives h(i) = horizontal sums of (a, b, a^2, b^2, a b); h(i - 2) + h(i - 1) + h(i) = the window sums centred on row
//   i - 1 -> SSIM and its partials P(i - 1) w.r.t. (mu_a, E[a^2], E[a b]) -> their horizontal sums hP(i - 1);
//   hP(i - 3) + hP(i - 2) + hP(i - 1) = the adjoint box filter centred on row i - 2 -> the gradient of that row.
// No LDS, no barrier; rows are fetched four at a time, the next four in flight under the arithmetic of the current ones.
//   v_render = grad_scale * [ (1 - lambda) sign(a - b) - lambda * d SSIM ] / (3 H W)     (straight-through clamp)
// with zero padding and divisor 9 (avg_pool2d(3, 1, padding=1), count_include_pad), partials zero outside the image.
#ifndef GSA_LOSS_ROWS
#define GSA_LOSS_ROWS 28
#endif
#ifndef GSA_LOSS_CHUNK
#define GSA_LOSS_CHUNK 4
#endif
constexpr int kStripCols = 60, kStripRows = GSA_LOSS_ROWS, kStripChunk = GSA_LOSS_CHUNK;
static_assert((kStripRows + 4) % kStripChunk == 0, "whole chunks of input rows");

// the value of the lane below / above (0 at the wave's ends): DPP wave shifts, folded into the adds that use them
__device__ __forceinline__ float lane_below(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_above(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float hsum3(float v) {
#pragma clang fp contract(off)   // v is often a product: fused into the first add it could not take the shifted operand (+1 move)
    return (lane_below(v) + v) + lane_above(v);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// A whole image as a buffer resource (base pointer from the kernel arguments, so in scalar registers): a row's byte offset
// rides in the scalar offset of a buffer load / store and the lane's column in the 32-bit vector offset -- no 64-bit
// address arithmetic per lane and row.  (The host side refuses images of 4 GiB and more.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_resource(const float *image, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(image), 0, bytes, 0x00020000);
}

// One strip of one band.  EDGE: the strip touches the image border (zero padding, clamped fetches, partial bands);
// interior strips (~90 % at 1080p) run the same arithmetic with every border test compiled out.  The rows of a strip
// are walked as a first chunk (no statistics before input row 2, no gradient before input row 4), whole middle chunks
// and a last chunk, so what a row has to do is known at compile time.
template <bool WANT_GRAD, bool EDGE>
__device__ __forceinline__ void loss_strip_rows(int width, int height, int lane, int c, int x, int y0,
                                                const float *__restrict__ render, const float *__restrict__ target,
                                                float w_l1, float w_ss, float *__restrict__ v_render, float &l1_out,
                                                float &ssim_out, float &mse_out) {
    static_assert(kStripChunk >= 4 && kStripRows + 4 >= 2 * kStripChunk, "first / middle / last chunks");
    const bool col_in = !EDGE || (x >= 0 && x < width);
    const bool col_out = col_in && lane >= 2 && lane < 2 + kStripCols;   // the columns this wave writes and counts
    // byte offsets of the lane's column within a row (32-bit: a row base in scalar registers + this, no 64-bit lane math)
    const unsigned col_bytes = (unsigned)((EDGE ? min(max(x, 0), width - 1) : x) * 3 + c) * 4u;
    const unsigned row_bytes = (unsigned)width * 12u, image_bytes = (unsigned)height * row_bytes;
    const __amdgpu_buffer_rsrc_t render_rows = image_resource(render, image_bytes), target_rows = image_resource(target, image_bytes),
                                 grad_rows = image_resource(WANT_GRAD ? v_render : render, image_bytes);

    // input rows i = 0 .. kStripRows + 3 are image rows y0 - 2 + i; loads are clamped, not predicated (in flight together)
    auto fetch = [&](int i, float &a, float &b) {
        const int y = y0 - 2 + i;
        const unsigned row = (unsigned)(EDGE ? min(max(y, 0), height - 1) : y) * row_bytes;   // (uniform)
        a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(render_rows, col_bytes, row, 0));
        b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(target_rows, col_bytes, row, 0));
    };
    float na[kStripChunk], nb[kStripChunk];
#pragma unroll
    for (int k = 0; k < kStripChunk; ++k) fetch(k, na[k], nb[k]);

    // the statistics of the render and of the target go through the same arithmetic: kept as (a, b) pairs, which the packed
    // fp32 instructions add / multiply two at a time
    f32x2 h1_m = {0, 0}, h2_m = {0, 0}, h1_q = {0, 0}, h2_q = {0, 0};   // horizontal sums of (a, b), (a a, b b): rows i - 1, i - 2
    float h1_x = 0.f, h2_x = 0.f;                                       // ... of a b
    f32x2 p1_m = {0, 0}, p2_m = {0, 0};                                 // horizontal sums of the partials (mu, aa): rows i - 2, i - 3
    float p1_x = 0.f, p2_x = 0.f;                                       // ... of the partial ab
    float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;                       // a, b of rows i - 1, i - 2
    float l1 = 0.f, ssim = 0.f, mse = 0.f;

    // phase 0: the first chunk, 1: a middle chunk, 2: the last chunk
    auto chunk = [&](auto phase, const int i0) {
        constexpr int PHASE = decltype(phase)::value;
        float ca[kStripChunk], cb[kStripChunk];
#pragma unroll
        for (int k = 0; k < kStripChunk; ++k) { ca[k] = na[k]; cb[k] = nb[k]; }
        if (PHASE != 2) {
#pragma unroll
            for (int k = 0; k < kStripChunk; ++k) fetch(i0 + kStripChunk + k, na[k], nb[k]);
        }
#pragma unroll
        for (int k = 0; k < kStripChunk; ++k) {
            const int i = i0 + k, y = y0 - 2 + i;
            const bool in = !EDGE || (col_in && y >= 0 && y < height);
            const float a = in ? clamp01(ca[k]) : 0.f, b = in ? cb[k] : 0.f;
            const f32x2 ab = {a, b}, sq = ab * ab;
            const f32x2 h_m = {hsum3(a), hsum3(b)}, h_q = {hsum3(sq.x), hsum3(sq.y)};
            const float h_x = hsum3(a * b);
            // ---- statistics centred on row y - 1 (needs rows y - 2 .. y)
            f32x2 P_m = {0.f, 0.f};
            float P_x = 0.f;
            const int yc = y - 1;
            if (PHASE != 0 || k >= 2) {
                const float inv9 = 1.f / 9.f;
                const f32x2 mu = ((h2_m + h1_m) + h_m) * inv9, mu_sq = mu * mu;
                const f32x2 var = ((h2_q + h1_q) + h_q) * inv9 - mu_sq;
                const float mu_ab = mu.x * mu.y, cov = ((h2_x + h1_x) + h_x) * inv9 - mu_ab;
                const float A1 = 2.f * mu_ab + kC1, A2 = 2.f * cov + kC2;
                const float B1 = mu_sq.x + mu_sq.y + kC1, B2 = var.x + var.y + kC2;
                const float den = B1 * B2;
                float inv = __builtin_amdgcn_rcpf(den);
                inv = inv * (2.f - den * inv);            // one Newton step: as good as the division
                const float S = A1 * A2 * inv;
                // (outside the image there is no window: its partials are zeros)
                const float inv_p = (!EDGE || (col_in && yc >= 0 && yc < height)) ? inv : 0.f;
                // d S / d mu_a with E[a^2], E[ab] held fixed (var_a and cov depend on mu_a too); d S / d E[a^2]; d S / d E[ab]
                P_m.x = (2.f * mu.y * (A2 - A1) - S * 2.f * mu.x * (B2 - B1)) * inv_p;
                P_m.y = -S * B1 * inv_p;
                P_x = 2.f * A1 * inv_p;
                // the band's own rows: 3 <= i < kStripRows + 3 (lanes outside the written columns are dropped at the end)
                const bool own = (PHASE == 0 ? k >= 3 : true) && (PHASE == 2 ? k < kStripChunk - 1 : true);
                if (own) {
                    const float d = a1 - b1;
                    if (!EDGE || yc < height) {   // (uniform; rows above the image do not occur: y0 >= 0)
                        l1 += fabsf(d); mse = __builtin_fmaf(d, d, mse); ssim += S;
                    }
                }
            }
            const f32x2 hp_m = {hsum3(P_m.x), hsum3(P_m.y)};
            const float hp_x = hsum3(P_x);
            // ---- the gradient of row y - 2 (needs the partials of rows y - 3 .. y - 1)
            if (WANT_GRAD && (PHASE != 0 || k >= 4)) {
                const int yo = y - 2;
                if (col_out && (!EDGE || yo < height)) {
                    const f32x2 s_m = (p2_m + p1_m) + hp_m;   // (mu, aa)
                    const float s_ab = (p2_x + p1_x) + hp_x;
                    const float d_ssim = s_m.x + 2.f * a2 * s_m.y + b2 * s_ab;
                    const float d = a2 - b2;
                    const float sign = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, w_l1 * sign - w_ss * d_ssim), grad_rows,
                                                          col_bytes, (unsigned)yo * row_bytes, 0);
                }
            }
            h2_m = h1_m; h1_m = h_m; h2_q = h1_q; h1_q = h_q; h2_x = h1_x; h1_x = h_x;
            p2_m = p1_m; p1_m = hp_m; p2_x = p1_x; p1_x = hp_x;
            a2 = a1; b2 = b1; a1 = a; b1 = b;
        }
    };
    chunk(std::integral_constant<int, 0>{}, 0);
    for (int i0 = kStripChunk; i0 < kStripRows + 4 - kStripChunk; i0 += kStripChunk) chunk(std::integral_constant<int, 1>{}, i0);
    chunk(std::integral_constant<int, 2>{}, kStripRows + 4 - kStripChunk);
    l1_out = col_out ? l1 : 0.f; ssim_out = col_out ? ssim : 0.f; mse_out = col_out ? mse : 0.f;
}

template <bool WANT_GRAD>
__global__ __launch_bounds__(3 * GSA_WAVE) void loss_strip_kernel(int width, int height, int strips,
                                                                    const float *__restrict__ render,
                                                                    const float *__restrict__ target, float lambda_dssim,
                                                                    float grad_scale, float *__restrict__ v_render,
                                                                    double *__restrict__ wave_sums) {
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;      // the wave's colour channel
    const int strip = blockIdx.x % strips, band = blockIdx.x / strips;
    const int x = strip * kStripCols + lane - 2, y0 = band * kStripRows;
    const float inv_count = grad_scale / (3.f * (float)width * (float)height);
    const float w_l1 = (1.f - lambda_dssim) * inv_count, w_ss = lambda_dssim * inv_count * (1.f / 9.f);
    // interior: all 64 columns and all kStripRows + 4 rows of the strip lie inside the image
    const bool interior = strip > 0 && (strip + 1) * kStripCols + 2 <= width && band > 0 &&
                          (band + 1) * kStripRows + 2 <= height;
    float l1, ssim, mse;
    if (interior)
        loss_strip_rows<WANT_GRAD, false>(width, height, lane, c, x, y0, render, target, w_l1, w_ss, v_render, l1, ssim, mse);
    else
        loss_strip_rows<WANT_GRAD, true>(width, height, lane, c, x, y0, render, target, w_l1, w_ss, v_render, l1, ssim, mse);
    double s0 = l1, s1 = ssim, s2 = mse;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d);
    }
    if (lane == 0) {
        double *out = wave_sums + 3 * ((long)blockIdx.x * 3 + c);
        out[0] = s0; out[1] = s1; out[2] = s2;
    }
}

// One Adam step over a flat bucket.  seg_end[k] = one past the last element of segment k (ascending), lr[k] its
// learning rate.  Mirrors torch.optim.Adam's single-tensor arithmetic:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// A segment may be STRIPED: its elements repeat with `period`, and the first `head` of every period take lr_head instead
// of lr (the [N,16,3] SH table kept as ONE segment: the 3 floats of coefficient 0 are the reference's "f_dc" group, the
// other 45 its "f_rest" group at a twentieth of the rate, trainer.py:333-334).  period 0: a plain segment.
struct AdamSegments {
    long end[8];
    float lr[8], lr_head[8];
    int period[8], head[8];
    int count;
};

__device__ __forceinline__ float adam_lr(const AdamSegments &seg, long i) {
    float lr = 0.f;
    long begin = 0;
    int k_hit = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k)
        if (k < seg.count && i < seg.end[k]) { lr = seg.lr[k]; k_hit = k; }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (k < k_hit) begin = seg.end[k];
    const int period = seg.period[k_hit];
    if (period > 0 && (int)((unsigned long)(i - begin) % (unsigned)period) < seg.head[k_hit]) lr = seg.lr_head[k_hit];
    return lr;
}

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
            const float lr = adam_lr(seg, i0 + j);
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
            const float lr = adam_lr(seg, i);
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
    if (shs && q < 12 * n) {   // (no table to build when the model keeps its coefficients as [N,16,3] rows already)
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

static long view_loss_waves(int width, int height) {
    return 3L * gsa_div_up(width, kStripCols) * gsa_div_up(height, kStripRows);
}

extern "C" size_t gsa_view_loss_workspace_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    return (size_t)view_loss_waves(width, height) * 3 * sizeof(double);   // the per-wave sums
}

extern "C" int gsa_view_loss(int width, int height, const float *render, const float *target, float lambda_dssim,
                             float grad_scale, void *workspace, size_t workspace_bytes, float *v_render, double *sums,
                             gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || !render || !target || !workspace || !sums, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_view_loss_workspace_bytes(width, height), GSA_ERR_WORKSPACE_TOO_SMALL);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(workspace) & 7) != 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((unsigned long long)width * height * 12ull >= (1ull << 32), GSA_ERR_INVALID_ARGUMENT);   // 32-bit byte offsets
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int strips = gsa_div_up(width, kStripCols);
    const int grid = strips * gsa_div_up(height, kStripRows);
    double *wave_sums = static_cast<double *>(workspace);
    if (v_render)
        loss_strip_kernel<true><<<grid, 3 * GSA_WAVE, 0, s>>>(width, height, strips, render, target, lambda_dssim, grad_scale,
                                                             v_render, wave_sums);
    else
        loss_strip_kernel<false><<<grid, 3 * GSA_WAVE, 0, s>>>(width, height, strips, render, target, lambda_dssim, grad_scale,
                                                              nullptr, wave_sums);
    GSA_LAUNCH_CHECK();
    loss_finalize_kernel<<<1, kFinalizeThreads, 0, s>>>((int)view_loss_waves(width, height), wave_sums, sums);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_adam_step_striped(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                     int n_segments, const long long *segment_end, const float *segment_lr,
                                     const int *segment_period, const int *segment_head, const float *segment_head_lr,
                                     double beta1, double beta2, double eps, int step, const uint32_t *skip_if_nonzero,
                                     gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || n_segments < 1 || n_segments > 8 || step < 1, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!param || !grad || !exp_avg || !exp_avg_sq || !segment_end || !segment_lr, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(segment_period && (!segment_head || !segment_head_lr), GSA_ERR_INVALID_ARGUMENT);
    AdamSegments seg;
    seg.count = n_segments;
    long long prev = 0;
    for (int k = 0; k < 8; ++k) {
        seg.end[k] = k < n_segments ? (long)segment_end[k] : (long)n;
        seg.lr[k] = k < n_segments ? segment_lr[k] : 0.f;
        seg.period[k] = (k < n_segments && segment_period) ? segment_period[k] : 0;
        seg.head[k] = seg.period[k] > 0 ? segment_head[k] : 0;
        seg.lr_head[k] = seg.period[k] > 0 ? segment_head_lr[k] : 0.f;
        if (k < n_segments) {
            GSA_RETURN_IF(segment_end[k] < prev || seg.period[k] < 0 || seg.head[k] < 0 || seg.head[k] > seg.period[k],
                          GSA_ERR_INVALID_ARGUMENT);
            GSA_RETURN_IF(seg.period[k] > 0 && (segment_end[k] - prev) % seg.period[k] != 0, GSA_ERR_INVALID_ARGUMENT);
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

extern "C" int gsa_adam_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                             int n_segments, const long long *segment_end, const float *segment_lr, double beta1,
                             double beta2, double eps, int step, const uint32_t *skip_if_nonzero, gsa_stream_t stream) {
    return gsa_adam_step_striped(n, param, grad, exp_avg, exp_avg_sq, n_segments, segment_end, segment_lr, nullptr, nullptr,
                                 nullptr, beta1, beta2, eps, step, skip_if_nonzero, stream);
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

// The step's metrics in ONE launch (the trainer's bookkeeping around the loss, src/training/trainer.py:485-519, otherwise
// ~25 five-microsecond torch kernels between the Adam step and the next step's first kernel): per local view
// l1 / ssim / mse = sums / count, psnr = mse < 1e-10 ? 100 : -10 log10(mse) (:70-75), loss = (1 - lambda) l1 + lambda
// (1 - ssim); the regulariser = reg_factor * sum(reg_partials).  report = { sum l1, sum ssim, sum psnr,
// sum loss / n_views_total + reg (if add_reg), reg (if add_reg) } -- what the ranks then sum.
struct ReportViews {
    double inv_count[16];
    int count;
};

__global__ __launch_bounds__(kThreads) void step_report_kernel(ReportViews views, const double *__restrict__ view_sums,
                                                               double lambda, double inv_views_total,
                                                               const double *__restrict__ reg_partials, long n_partials,
                                                               double reg_factor, int add_reg, double *__restrict__ report) {
    __shared__ double red[kThreads / GSA_WAVE];
    double acc = 0;
    for (long b = threadIdx.x; b < n_partials; b += kThreads) acc += reg_partials[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double reg = 0;
        for (int w = 0; w < kThreads / GSA_WAVE; ++w) reg += red[w];
        reg *= reg_factor;
        double l1 = 0, ssim = 0, psnr = 0, loss = 0;
        for (int j = 0; j < views.count; ++j) {
            const double a = view_sums[3 * j] * views.inv_count[j], b = view_sums[3 * j + 1] * views.inv_count[j],
                         mse = view_sums[3 * j + 2] * views.inv_count[j];
            l1 += a; ssim += b;
            psnr += mse < 1e-10 ? 100.0 : -10.0 * log10(fmax(mse, 1e-300));
            loss += (1.0 - lambda) * a + lambda * (1.0 - b);
        }
        report[0] = l1; report[1] = ssim; report[2] = psnr;
        report[3] = loss * inv_views_total + (add_reg ? reg : 0.0);
        report[4] = add_reg ? reg : 0.0;
    }
}

extern "C" int gsa_step_report(int n_local_views, const double *view_sums, const double *view_inv_count, double lambda_dssim,
                               int n_views_total, const double *reg_partials, long long n_partials, double reg_factor,
                               int add_reg, double *report, gsa_stream_t stream) {
    GSA_RETURN_IF(n_local_views < 0 || n_local_views > 16 || n_views_total < 1 || n_partials < 0 || !report,
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((n_local_views > 0 && (!view_sums || !view_inv_count)) || (n_partials > 0 && !reg_partials),
                  GSA_ERR_INVALID_ARGUMENT);
    ReportViews views;
    views.count = n_local_views;
    for (int j = 0; j < 16; ++j) views.inv_count[j] = j < n_local_views ? view_inv_count[j] : 0.0;   // (a HOST array)
    step_report_kernel<<<1, kThreads, 0, static_cast<hipStream_t>(stream)>>>(views, view_sums, lambda_dssim,
                                                                             1.0 / (double)n_views_total, reg_partials,
                                                                             (long)n_partials, reg_factor, add_reg ? 1 : 0,
                                                                             report);
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
    GSA_RETURN_IF(!scaling || !rotation || !opacity || !scales || !quats || !opacities, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(shs && (!features_dc || !features_rest), GSA_ERR_INVALID_ARGUMENT);   // shs NULL: geometry only
    GSA_RETURN_IF(((reinterpret_cast<uintptr_t>(quats) | reinterpret_cast<uintptr_t>(shs)) & 15) != 0,
                  GSA_ERR_INVALID_ARGUMENT);
    activate_fwd_kernel<<<gsa_div_up(shs ? 12 * n : n, kThreads), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
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
