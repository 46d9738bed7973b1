// composite.hip -- stage 3 (front-to-back alpha compositing) and stage 4 (reverse gradient sweep).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile, one lane per pixel; wave w owns the 16x4 pixel
// strip of rows 4w..4w+3 so image rows are written in 64-B runs.  The tile's depth-sorted splat list is
// staged through LDS in batches of 256 records (each lane gathers one 64-B splat record by its sorted
// index -- one cache line, served from L2/Infinity Cache since the whole record table is 64 MB at 1M
// gaussians); the blend loop then reads each record as an LDS broadcast.  Early-out is wave-level
// (a wave whose 64 pixels are all done leaves the blend loop through its exec mask) and
// workgroup-level (all 256 pixels done -> stop staging).
//
// Restates SURVEY.md Appendix A.5 (forward) and A.6 (backward) -- gsplat's rasterize_to_pixels
// behind /root/reference/src/backends/gsplat_backend.py:168-182.

#include "gsa_common.h"

namespace {

constexpr int kTile = GSA_TILE_SIZE;
constexpr int kBlock = kTile * kTile;  // 256
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp_neg(float sigma) {  // exp(-sigma) via v_exp_f32
    return __builtin_amdgcn_exp2f(-sigma * kLog2e);
}

template <int CH>
struct Pix {
    float v[CH];
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(kBlock) void composite_fwd_kernel(
    int width, int height, int tile_w, const float *__restrict__ splat, const uint32_t *__restrict__ sorted_gid,
    const int32_t *__restrict__ tile_offsets, const float *__restrict__ background, float *__restrict__ render,
    float *__restrict__ alpha_out, float *__restrict__ final_T, int32_t *__restrict__ last_id) {
    __shared__ float4 s_q0[kBlock];  // mean.x mean.y conic.a conic.b
    __shared__ float4 s_q1[kBlock];  // conic.c opacity r g
    __shared__ float2 s_q2[kBlock];  // b depth

    const int t = threadIdx.x;
    const int tile = blockIdx.x;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px = tx * kTile + (t & 15), py = ty * kTile + (t >> 4);
    const bool inside = px < width && py < height;
    const float pxf = px + 0.5f, pyf = py + 0.5f;
    const int start = tile_offsets[tile], end = tile_offsets[tile + 1];

    float T = 1.f;
    float acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = 0.f;
    int last = -1;
    bool done = !inside;

    for (int base = start; base < end; base += kBlock) {
        // also orders the previous batch's LDS reads before this batch's LDS writes
        if (__syncthreads_count(done) == kBlock) break;
        int idx = base + t;
        if (idx < end) {
            const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)sorted_gid[idx] * GSA_SPLAT_STRIDE);
            float4 q0 = row[0], q1 = row[1], q2 = row[2];
            s_q0[t] = q0; s_q1[t] = q1; s_q2[t] = make_float2(q2.x, q2.y);
        }
        __syncthreads();
        const int cnt = min(kBlock, end - base);
        for (int j = 0; j < cnt && !done; ++j) {
            float4 q0 = s_q0[j], q1 = s_q1[j];
            float dx = q0.x - pxf, dy = q0.y - pyf;
            float sigma = 0.5f * (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;
            float a = fminf(GSA_MAX_ALPHA, q1.y * fast_exp_neg(sigma));
            if (sigma < 0.f || a < GSA_ALPHA_THRESHOLD) continue;
            float nT = T * (1.f - a);
            if (nT <= GSA_TRANSMITTANCE_EPS) { done = true; break; }
            float w = a * T;
            float2 q2 = s_q2[j];
            acc[0] += q1.z * w; acc[1] += q1.w * w; acc[2] += q2.x * w;
            if (CH == 4) acc[CH - 1] += q2.y * w;
            T = nT;
            last = base + j;
        }
    }

    if (inside) {
        size_t pix = (size_t)py * width + px;
#pragma unroll
        for (int k = 0; k < CH; ++k) render[pix * CH + k] = acc[k] + (background ? T * background[k] : 0.f);
        alpha_out[pix] = 1.f - T;
        final_T[pix] = T;  // kept beside alpha: 1 - (1 - T) loses the low bits of a small T
        last_id[pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(r);
}

// sum over the 64 lanes of a wave; result is wave-uniform
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);  // row_mirror        -> every lane holds its row-of-16 sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int CH>
__global__ __launch_bounds__(kBlock) void composite_bwd_kernel(
    int width, int height, int tile_w, const float *__restrict__ splat, const uint32_t *__restrict__ sorted_gid,
    const int32_t *__restrict__ tile_offsets, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ last_id, const float *__restrict__ v_render,
    const float *__restrict__ v_alpha, float *__restrict__ v_splat) {
    __shared__ float4 s_q0[kBlock];
    __shared__ float4 s_q1[kBlock];
    __shared__ float2 s_q2[kBlock];
    __shared__ uint32_t s_gid[kBlock];
    __shared__ int s_max_last;

    const int t = threadIdx.x, lane = t & 63;
    const int tile = blockIdx.x;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px = tx * kTile + (t & 15), py = ty * kTile + (t >> 4);
    const bool inside = px < width && py < height;
    const float pxf = px + 0.5f, pyf = py + 0.5f;
    const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
    if (end <= start) return;

    const size_t pix = inside ? (size_t)py * width + px : 0;
    const int bin_final = inside ? last_id[pix] : -1;
    const float T_final = inside ? final_T[pix] : 1.f;
    float T = T_final;
    float vout[CH], buf[CH];
    float bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vout[k] = inside ? v_render[pix * CH + k] : 0.f;
        buf[k] = 0.f;
        if (background) bgdot += background[k] * vout[k];
    }
    const float va = (inside && v_alpha) ? v_alpha[pix] : 0.f;

    if (t == 0) s_max_last = -1;
    __syncthreads();
    if (bin_final >= 0) atomicMax(&s_max_last, bin_final);
    __syncthreads();
    const int top0 = s_max_last + 1;  // records at or beyond top0 touched no pixel of this tile

    for (int top = top0; top > start; top -= kBlock) {
        const int cnt = min(kBlock, top - start);
        __syncthreads();
        if (t < cnt) {
            uint32_t g = sorted_gid[top - 1 - t];
            const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)g * GSA_SPLAT_STRIDE);
            float4 q0 = row[0], q1 = row[1], q2 = row[2];
            s_q0[t] = q0; s_q1[t] = q1; s_q2[t] = make_float2(q2.x, q2.y);
            s_gid[t] = g;
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const int m = top - 1 - j;
            float4 q0 = s_q0[j], q1 = s_q1[j];
            float dx = q0.x - pxf, dy = q0.y - pyf;
            float sigma = 0.5f * (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;
            float vis = fast_exp_neg(sigma);
            float a = fminf(GSA_MAX_ALPHA, q1.y * vis);
            bool valid = (m <= bin_final) && !(sigma < 0.f || a < GSA_ALPHA_THRESHOLD);
            if (__ballot(valid) == 0ull) continue;

            float g_mx = 0.f, g_my = 0.f, g_a = 0.f, g_b = 0.f, g_c = 0.f, g_o = 0.f;
            float g_f[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) g_f[k] = 0.f;
            if (valid) {
                float2 q2 = s_q2[j];
                float f[CH];
                f[0] = q1.z; f[1] = q1.w; f[2] = q2.x;
                if (CH == 4) f[CH - 1] = q2.y;
                float ra = 1.f / (1.f - a);
                T *= ra;
                float fac = a * T;
                float v_a = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    g_f[k] = fac * vout[k];
                    v_a += (f[k] * T - buf[k] * ra) * vout[k];
                    buf[k] += f[k] * fac;
                }
                v_a += T_final * ra * va;
                v_a -= T_final * ra * bgdot;
                if (q1.y * vis <= GSA_MAX_ALPHA) {
                    float v_sigma = -q1.y * vis * v_a;
                    g_a = 0.5f * v_sigma * dx * dx;
                    g_b = v_sigma * dx * dy;
                    g_c = 0.5f * v_sigma * dy * dy;
                    g_mx = v_sigma * (q0.z * dx + q0.w * dy);
                    g_my = v_sigma * (q0.w * dx + q1.x * dy);
                    g_o = vis * v_a;
                }
            }
            // wave reduction, then lanes 0..5+CH add one record field each (one 64-B atomic request)
            float s0 = wave_sum(g_mx), s1 = wave_sum(g_my), s2 = wave_sum(g_a), s3 = wave_sum(g_b),
                  s4 = wave_sum(g_c), s5 = wave_sum(g_o);
            float sf[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) sf[k] = wave_sum(g_f[k]);
            float val = lane == 0 ? s0 : lane == 1 ? s1 : lane == 2 ? s2 : lane == 3 ? s3 : lane == 4 ? s4 : s5;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (lane == 6 + k) val = sf[k];
            if (lane < 6 + CH) atomicAdd(v_splat + (size_t)s_gid[j] * GSA_SPLAT_STRIDE + lane, val);
        }
    }
}

}  // namespace

extern "C" int gsa_composite_fwd(int width, int height, int channels, const float *splat,
                                 const uint32_t *sorted_gid, const int32_t *tile_offsets, const float *background,
                                 float *render, float *alpha, float *final_T, int32_t *last_id,
                                 gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !render || !alpha || !final_T || !last_id, GSA_ERR_INVALID_ARGUMENT);
    int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3)
        composite_fwd_kernel<3><<<tile_w * tile_h, kBlock, 0, s>>>(width, height, tile_w, splat, sorted_gid,
                                                                  tile_offsets, background, render, alpha, final_T, last_id);
    else
        composite_fwd_kernel<4><<<tile_w * tile_h, kBlock, 0, s>>>(width, height, tile_w, splat, sorted_gid,
                                                                  tile_offsets, background, render, alpha, final_T, last_id);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_composite_bwd(int width, int height, int channels, const float *splat,
                                 const uint32_t *sorted_gid, const int32_t *tile_offsets, const float *background,
                                 const float *final_T, const int32_t *last_id, const float *v_render,
                                 const float *v_alpha, float *v_splat, gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !final_T || !last_id || !v_render || !v_splat, GSA_ERR_INVALID_ARGUMENT);
    int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3)
        composite_bwd_kernel<3><<<tile_w * tile_h, kBlock, 0, s>>>(width, height, tile_w, splat, sorted_gid,
                                                                  tile_offsets, background, final_T, last_id, v_render,
                                                                  v_alpha, v_splat);
    else
        composite_bwd_kernel<4><<<tile_w * tile_h, kBlock, 0, s>>>(width, height, tile_w, splat, sorted_gid,
                                                                  tile_offsets, background, final_T, last_id, v_render,
                                                                  v_alpha, v_splat);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}
