// composite.hip -- stage 3 (front-to-back alpha compositing) and stage 4 (reverse gradient sweep).
//
// Mapping (gfx950, wave64): one 256-thread workgroup per 16x16 tile, ONE WAVE PER 8x8 QUADRANT, one lane per
// pixel (lane l -> (x = l & 7, y = l >> 3) inside the quadrant).
//
//  * Staging: the workgroup gathers 256 sorted splat records at a time (one 64-B record per thread, one cache
//    line, served from L2 / Infinity Cache) into LDS.  The staging thread also computes for ITS record the
//    4-bit mask of the quadrants whose pixel centres the alpha >= 1/255 ellipse can reach (exact
//    ellipse-vs-rectangle test, conservative margin).
//  * Ballot compaction: each wave turns the mask bits of its own quadrant into a 64-bit SGPR mask per 64
//    staged records (__ballot) and walks only the set bits (s_ff1) -- a record that cannot touch the quadrant
//    costs the wave nothing, and there is no per-record test at all.  About 45 % of (quadrant, record)
//    pairs survive.  Results are identical to the plain per-pixel loop: a skipped pair fails the alpha test
//    at every pixel of the quadrant anyway.
//  * Blend: the surviving record is read as an LDS broadcast; per-pixel termination is carried in the sign
//    of T (negative = finished), so the body is branch-free.  A wave leaves the loop when its 64 pixels are
//    done (ballot), the workgroup stops staging when all 256 are (__syncthreads_count).
//  * Backward: per (quadrant, record) the 64 per-pixel contributions are reduced across the wave -- the eight
//    leading fields by a transpose-reduce built on the gfx950 v_permlane32_swap / v_permlane16_swap
//    instructions (24 VALU ops instead of 54) -- and added with one atomic instruction whose active lanes
//    cover one 64-B gradient record (a single memory-side request).
//
// Why one wave per quadrant and not per tile: a lone wave issues a dependent instruction stream at a small
// fraction of the SIMD rate, so the longest tile bounds the kernel.  A wave-per-tile version (4 pixels per
// lane, same masks) had a 2.6x longer critical path and ran 0.25 / 0.49 ms where this runs faster; a
// persistent-wave variant pulling tiles from per-XCD atomic queues was slower still.
//
// Restates SURVEY.md Appendix A.5 (forward) and A.6 (backward) -- gsplat's rasterize_to_pixels behind
// /root/reference/src/backends/gsplat_backend.py:168-182.

#include "project_math.h"

namespace {

constexpr int kTile = GSA_TILE_SIZE;
constexpr int kBlock = 256;   // 4 waves = 4 quadrants
constexpr int kBatch = 256;   // records staged per workgroup and step
constexpr float kLog2e = 1.4426950408889634f;

// Workgroup -> tile.  Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness):
// give every XCD one contiguous band of tile rows so neighbouring tiles (which share splats) share an L2.
// Injective from [0, 8 * chunk) onto itself; callers drop ids >= n_tiles.
__device__ __forceinline__ int xcd_band_tile(int b, int n_tiles) {
    const int chunk = (n_tiles + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// 64-bit mask of the staged records [c*64, c*64+64) that can touch quadrant q (wave-uniform result)
__device__ __forceinline__ unsigned long long relevant_records(const uint32_t *s_mask, int c, int q, int lane, int cnt) {
    const int j = c * 64 + lane;
    const bool hit = j < cnt && ((s_mask[j] >> q) & 1u);
    return __ballot(hit);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(kBlock) void composite_fwd_kernel(
    int width, int height, int tile_w, int n_tiles, const float *__restrict__ splat,
    const uint32_t *__restrict__ sorted_gid, const int32_t *__restrict__ tile_offsets,
    const float *__restrict__ background, float *__restrict__ render, float *__restrict__ alpha_out,
    float *__restrict__ final_T, int32_t *__restrict__ last_id) {
    // conic pre-scaled for exp2:  s = log2(e) * sigma = dx (ha dx + hb dy) + hc dy^2
    __shared__ float4 s_q0[kBatch];    // mean.x mean.y ha hb
    __shared__ float4 s_q1[kBatch];    // hc opacity r g
    __shared__ float2 s_q2[kBatch];    // b depth
    __shared__ uint32_t s_mask[kBatch];

    const int tile = xcd_band_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);  // this wave's quadrant (wave-uniform -> scalar control flow)
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px = tx * kTile + (q & 1) * 8 + (lane & 7), py = ty * kTile + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < width && py < height;
    const float pxf = px + 0.5f, pyf = py + 0.5f;
    const int start = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int end = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);

    // T > 0: running transmittance; T < 0: pixel finished, |T| is its final transmittance
    float T = inside ? 1.f : -1.f;
    float acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = 0.f;
    int last = -1;
    bool wave_live = __ballot(T > 0.f) != 0ull;

    for (int base = start; base < end; base += kBatch) {
        // also orders the previous batch's LDS reads before this batch's LDS writes
        if (__syncthreads_count(T > 0.f) == 0) break;
        const int idx = base + t;
        if (idx < end) {
            const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)sorted_gid[idx] * GSA_SPLAT_STRIDE);
            const float4 r0 = row[0], r1 = row[1], r2 = row[2];
            s_q0[t] = make_float4(r0.x, r0.y, 0.5f * kLog2e * r0.z, kLog2e * r0.w);
            s_q1[t] = make_float4(0.5f * kLog2e * r1.x, r1.y, r1.z, r1.w);
            s_q2[t] = make_float2(r2.x, r2.y);
            s_mask[t] = gsa_quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r2.z, tx, ty);
        }
        __syncthreads();
        const int cnt = min(kBatch, end - base);
        for (int c = 0; c * 64 < cnt && wave_live; ++c) {
            unsigned long long bits = relevant_records(s_mask, c, q, lane, cnt);
            int seen = 0;
            while (bits) {
                const int j = c * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                const float4 q0 = s_q0[j], q1 = s_q1[j];
                const float2 q2 = s_q2[j];
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                const float s = dx * (q0.z * dx + q0.w * dy) + (q1.x * dy) * dy;
                const float a = fminf(GSA_MAX_ALPHA, q1.y * __builtin_amdgcn_exp2f(-s));
                const bool valid = (s >= 0.f) && (a >= GSA_ALPHA_THRESHOLD);
                const float nT = T * (1.f - a);
                const bool contrib = valid && (nT > GSA_TRANSMITTANCE_EPS);  // false once T < 0
                const float wgt = contrib ? a * T : 0.f;
                acc[0] += q1.z * wgt; acc[1] += q1.w * wgt; acc[2] += q2.x * wgt;
                if (CH == 4) acc[CH - 1] += q2.y * wgt;
                last = contrib ? base + j : last;
                // valid but not blended: the stop rule fired (or the pixel was finished already)
                T = contrib ? nT : (valid ? -fabsf(T) : T);
                if ((++seen & 7) == 0 && __ballot(T > 0.f) == 0ull) {  // all 64 pixels finished
                    wave_live = false;
                    break;
                }
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)py * width + px;
        const float Tf = fabsf(T);
#pragma unroll
        for (int k = 0; k < CH; ++k) render[pix * CH + k] = acc[k] + (background ? Tf * background[k] : 0.f);
        alpha_out[pix] = 1.f - Tf;
        final_T[pix] = Tf;  // kept beside alpha: 1 - (1 - T) loses the low bits of a small T
        last_id[pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

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

// lanes [0,32): sum over both halves of x;  lanes [32,64): the same for y   (v_permlane32_swap)
__device__ __forceinline__ float fold32(float x, float y) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// even rows: x summed over its row pair; odd rows: the same for y   (v_permlane16_swap)
__device__ __forceinline__ float fold16(float x, float y) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// Transpose-reduce of eight per-lane values: on return lane 8k (and its 7 neighbours) holds the wave-wide
// sum of f[k].  24 VALU ops; the plain butterfly needs 6 per value.
__device__ __forceinline__ float reduce8(const float *f, int lane) {
    // level 32: (f0,f4) (f2,f6) (f1,f5) (f3,f7)
    float X = fold32(f[0], f[4]), Y = fold32(f[2], f[6]), Z = fold32(f[1], f[5]), W = fold32(f[3], f[7]);
    // level 16: rows become [f0, f2, f4, f6] and [f1, f3, f5, f7]
    float P = fold16(X, Y), Q = fold16(Z, W);
    // level 8: low half-row keeps P, high half-row keeps Q; partner is 8 lanes away in the same row
    const bool hi = (lane & 8) != 0;
    float keep = hi ? Q : P, send = hi ? P : Q;
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    float v = keep + __int_as_float(r);
    // levels 4, 2, 1 inside each group of 8 lanes
    v = dpp_add<0xB1, 0xF>(v);
    v = dpp_add<0x4E, 0xF>(v);
    v = dpp_add<0x141, 0xF>(v);
    return v;
}

template <int CH>
__global__ __launch_bounds__(kBlock) void composite_bwd_kernel(
    int width, int height, int tile_w, int n_tiles, const float *__restrict__ splat,
    const uint32_t *__restrict__ sorted_gid, const int32_t *__restrict__ tile_offsets,
    const float *__restrict__ background, const float *__restrict__ final_T, const int32_t *__restrict__ last_id,
    const float *__restrict__ v_render, const float *__restrict__ v_alpha, float *__restrict__ v_splat) {
    __shared__ float4 s_q0[kBatch];    // mean.x mean.y a/2 b
    __shared__ float4 s_q1[kBatch];    // c/2 opacity r g
    __shared__ float2 s_q2[kBatch];    // b depth
    __shared__ uint32_t s_mask[kBatch];
    __shared__ uint32_t s_gid[kBatch];
    __shared__ int s_max_last;

    const int tile = xcd_band_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int start = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int end = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (end <= start) return;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px = tx * kTile + (q & 1) * 8 + (lane & 7), py = ty * kTile + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < width && py < height;
    const float pxf = px + 0.5f, pyf = py + 0.5f;

    // per pixel: T = transmittance behind the current record; Wd = T_final (v_alpha - bg.v_out) - sum of
    // (colour . v_out) * weight over the splats behind -- what d(out)/d(alpha_i) multiplies by 1/(1-alpha_i)
    const size_t pix = inside ? (size_t)py * width + px : 0;
    const int binf = inside ? last_id[pix] : -1;
    const float Tf = inside ? final_T[pix] : 1.f;
    float T = Tf;
    float vout[CH], bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vout[k] = inside ? v_render[pix * CH + k] : 0.f;
        if (background) bgdot += background[k] * vout[k];
    }
    float Wd = Tf * (((inside && v_alpha) ? v_alpha[pix] : 0.f) - bgdot);

    int mx = binf;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
    const int qmax = __builtin_amdgcn_readfirstlane(mx);  // last record any pixel of this quadrant blended
    if (t == 0) s_max_last = -1;
    __syncthreads();
    if (lane == 0 && qmax >= 0) atomicMax(&s_max_last, qmax);
    __syncthreads();
    const int top0 = s_max_last + 1;  // records at or beyond top0 touched no pixel of this tile

    for (int top = top0; top > start; top -= kBatch) {
        const int cnt = min(kBatch, top - start);
        __syncthreads();
        if (t < cnt) {  // slot t holds record top-1-t: the loop below walks slots upwards = records backwards
            const uint32_t gi = sorted_gid[top - 1 - t];
            const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)gi * GSA_SPLAT_STRIDE);
            const float4 r0 = row[0], r1 = row[1], r2 = row[2];
            s_q0[t] = make_float4(r0.x, r0.y, 0.5f * r0.z, r0.w);
            s_q1[t] = make_float4(0.5f * r1.x, r1.y, r1.z, r1.w);
            s_q2[t] = make_float2(r2.x, r2.y);
            s_mask[t] = gsa_quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r2.z, tx, ty);
            s_gid[t] = gi;
        }
        __syncthreads();
        const int first = max(0, top - 1 - qmax);  // slots below `first` hold records behind this quadrant's last one
        for (int c = first >> 6; c * 64 < cnt; ++c) {
            unsigned long long bits = relevant_records(s_mask, c, q, lane, cnt);
            if (c == (first >> 6)) bits &= ~0ull << (first & 63);
            while (bits) {
                const int j = c * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                const int rec = top - 1 - j;
                const float4 q0 = s_q0[j], q1 = s_q1[j];
                const float2 q2 = s_q2[j];
                const float ha = q0.z, cb = q0.w, hc = q1.x, op = q1.y;
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                const float hadx = ha * dx, hcdy = hc * dy;
                const float sx = hadx + cb * dy;              // a/2 dx + b dy
                const float sigma = dx * sx + hcdy * dy;
                const float vis = __builtin_amdgcn_exp2f(-kLog2e * sigma);
                const float raw = op * vis;
                const float a = fminf(GSA_MAX_ALPHA, raw);
                const bool valid = (rec <= binf) && (sigma >= 0.f) && (a >= GSA_ALPHA_THRESHOLD);
                if (__ballot(valid) == 0ull) continue;
                const float ra = __builtin_amdgcn_rcpf(1.f - a);
                const float Tn = T * ra;             // transmittance in front of this splat
                const float fac = valid ? a * Tn : 0.f;
                float cv = q1.z * vout[0] + q1.w * vout[1] + q2.x * vout[2];
                if (CH == 4) cv += q2.y * vout[CH - 1];
                const float v_a = Tn * cv + ra * Wd;
                Wd -= fac * cv;
                const bool gate = valid && (raw <= GSA_MAX_ALPHA);
                const float u = gate ? -raw * v_a : 0.f;    // d loss / d sigma
                const float hu = 0.5f * u;
                float f[8];                                  // v_mx v_my v_a v_b v_c v_o v_r v_g
                f[0] = u * (sx + hadx);                      // a dx + b dy
                f[1] = u * (cb * dx + 2.f * hcdy);           // b dx + c dy
                f[2] = (hu * dx) * dx;
                f[3] = (u * dx) * dy;
                f[4] = (hu * dy) * dy;
                f[5] = gate ? vis * v_a : 0.f;
                f[6] = fac * vout[0];
                f[7] = fac * vout[1];
                T = valid ? Tn : T;
                // one reduction and one atomic instruction (one 64-B request) per (quadrant, splat)
                const float r8 = reduce8(f, lane);
                const float t8 = wave_sum(fac * vout[2]);
                const float t9 = (CH == 4) ? wave_sum(fac * vout[CH - 1]) : 0.f;
                float *dst = v_splat + (size_t)s_gid[j] * GSA_SPLAT_STRIDE;
                const bool lead = (lane & 7) == 0;
                const int off = lead ? (lane >> 3) : (lane == 1 ? 8 : 9);
                const float val = lead ? r8 : (lane == 1 ? t8 : t9);
                if (lead || lane == 1 || (CH == 4 && lane == 2)) atomicAdd(dst + off, val);
            }
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
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    const int grid = 8 * gsa_div_up(n_tiles, 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3)
        composite_fwd_kernel<3><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid, tile_offsets,
                                                       background, render, alpha, final_T, last_id);
    else
        composite_fwd_kernel<4><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid, tile_offsets,
                                                       background, render, alpha, final_T, last_id);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_composite_bwd(int width, int height, int channels, const float *splat,
                                 const uint32_t *sorted_gid, const int32_t *tile_offsets, const float *background,
                                 const float *final_T, const int32_t *last_id, const float *v_render,
                                 const float *v_alpha, float *v_splat, gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !final_T || !last_id || !v_render || !v_splat, GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    const int grid = 8 * gsa_div_up(n_tiles, 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3)
        composite_bwd_kernel<3><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid, tile_offsets,
                                                       background, final_T, last_id, v_render, v_alpha, v_splat);
    else
        composite_bwd_kernel<4><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid, tile_offsets,
                                                       background, final_T, last_id, v_render, v_alpha, v_splat);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}
