// composite.hip -- stage 3 (front-to-back alpha compositing) and stage 4 (reverse gradient sweep).
//
// Mapping (gfx950, wave64): ONE WAVE PER 16x16 TILE.  A lane owns four pixels, one in each 8x8 quadrant of
// the tile: lane l -> (x = l & 7, y = l >> 3) inside the quadrant.  A 256-thread workgroup is four
// independent waves (four consecutive tiles); there is no workgroup barrier anywhere.
//
//  * Placement: workgroups are dealt round-robin over the 8 XCDs, so workgroup b is mapped to tile group
//    (b % 8) * chunk + b / 8: every XCD gets one contiguous band of tile rows and neighbouring tiles --
//    which share splats -- share an L2.  All 8160 waves of a 1080p frame are resident at once (8 per SIMD).
//    (A persistent-wave variant pulling tiles from per-XCD atomic queues was measured 35-220 % slower on
//    the forward pass at every grid size: the kernels are VALU-issue-bound, not imbalance-bound.)
//  * Staging: the wave gathers 64 sorted splat records at a time (one 64-B record per lane, one cache line,
//    served from L2 / Infinity Cache) into its private LDS slice.  Each sorted record value carries, above
//    the gaussian index, the 4-bit mask of the quadrants whose pixel centres the alpha >= 1/255 ellipse can
//    reach (exact ellipse-vs-rectangle test, conservative margin).  The forward pass computes it while
//    staging -- only ~20 % of the binned records are ever visited -- and writes it back into the record
//    value, where the backward pass finds it.
//  * Blend loop: every lane reads the same record (LDS broadcast, 3 reads per record for 256 pixels, issued
//    one record ahead), the mask is moved to an SGPR and the four quadrants are evaluated under scalar
//    branches -- a quadrant the splat cannot touch, or whose 64 pixels are all finished, costs nothing.
//    Per-pixel termination is carried in the sign of T (negative = finished), so the body is branch-free
//    vector code.
//  * Backward: per record the four quadrants accumulate into per-lane partial sums first, so there is
//    ONE cross-lane reduction and ONE atomic (a single 64-B request) per (tile, splat) instead of one per
//    wave.  The reduction of the eight leading fields is a transpose-reduce built on the gfx950
//    v_permlane32_swap / v_permlane16_swap instructions (24 VALU ops instead of 54).
//
// Results are identical to the plain per-pixel loop: the quadrant mask only removes (pixel, splat) pairs
// that fail the alpha >= 1/255 test anyway.
//
// Restates SURVEY.md Appendix A.5 (forward) and A.6 (backward) -- gsplat's rasterize_to_pixels behind
// /root/reference/src/backends/gsplat_backend.py:168-182.

#include "project_math.h"

namespace {

#ifndef GSA_FWD_WAVES_PER_SIMD
#define GSA_FWD_WAVES_PER_SIMD 6  // register budget of the forward compositor (512 / waves VGPRs per lane)
#endif
#ifndef GSA_BWD_WAVES_PER_SIMD
#define GSA_BWD_WAVES_PER_SIMD 5
#endif

constexpr int kTile = GSA_TILE_SIZE;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kBatch = 64;  // records staged per wave and step
constexpr float kLog2e = 1.4426950408889634f;
constexpr uint32_t kGidMask = (1u << GSA_GID_BITS) - 1u;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup -> tile group.  Injective from [0, 8 * chunk) onto itself; callers drop groups >= n_groups.
__device__ __forceinline__ int xcd_band_remap(int b, int chunk) { return (b & 7) * chunk + (b >> 3); }

// This wave's tile, or -1 (wave-uniform).
__device__ __forceinline__ int wave_tile(int wave, int n_tiles) {
    const int n_groups = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const int grp = xcd_band_remap(blockIdx.x, (n_groups + 7) >> 3);
    const int tile = grp * kWavesPerBlock + wave;
    return (grp < n_groups && tile < n_tiles) ? tile : -1;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(kBlock, GSA_FWD_WAVES_PER_SIMD) void composite_fwd_kernel(
    int width, int height, int tile_w, int n_tiles, int fill_masks,
    const float *__restrict__ splat, uint32_t *__restrict__ sorted_gid,
    const int32_t *__restrict__ tile_offsets, const float *__restrict__ background, float *__restrict__ render,
    float *__restrict__ alpha_out, float *__restrict__ final_T, int32_t *__restrict__ last_id) {
    // conic pre-scaled for exp2:  s = log2(e) * sigma = dx (ha dx + hb dy) + hc dy^2
    __shared__ float4 s_q0[kWavesPerBlock][kBatch + 1];  // mean.x mean.y ha hb
    __shared__ float4 s_q1[kWavesPerBlock][kBatch + 1];  // hc opacity r g
    __shared__ float4 s_q2[kWavesPerBlock][kBatch + 1];  // b depth quadrant-mask(bits) -

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> scalar control flow
    const int lx = lane & 7, ly = lane >> 3;
    const int tile = wave_tile(w, n_tiles);
    if (tile < 0) return;
    {
        const int tx = tile % tile_w, ty = tile / tile_w;
        const float px0 = tx * kTile + lx + 0.5f, py0 = ty * kTile + ly + 0.5f;
        const int start = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
        const int end = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);

        // T > 0: running transmittance; T < 0: pixel finished, |T| is its final transmittance
        float T[4], acc[4][CH];
        int last[4];
        bool inside[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            inside[g] = (tx * kTile + (g & 1) * 8 + lx < width) && (ty * kTile + (g >> 1) * 8 + ly < height);
            T[g] = inside[g] ? 1.f : -1.f;
            last[g] = -1;
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[g][k] = 0.f;
        }
        uint32_t alive = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) alive |= (__ballot(T[g] > 0.f) != 0ull) ? (1u << g) : 0u;
        alive = __builtin_amdgcn_readfirstlane(alive);

        for (int base = start; base < end && alive != 0; base += kBatch) {
            const int idx = base + lane;
            if (idx < end) {
                uint32_t val = sorted_gid[idx];
                const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)(val & kGidMask) * GSA_SPLAT_STRIDE);
                float4 r0 = row[0], r1 = row[1], r2 = row[2];
                if (fill_masks) {  // quadrant mask of this record, kept in the record value for the backward pass
                    val |= gsa_quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r2.z, tx, ty) << GSA_GID_BITS;
                    sorted_gid[idx] = val;
                }
                s_q0[w][lane] = make_float4(r0.x, r0.y, 0.5f * kLog2e * r0.z, kLog2e * r0.w);
                s_q1[w][lane] = make_float4(0.5f * kLog2e * r1.x, r1.y, r1.z, r1.w);
                s_q2[w][lane] = make_float4(r2.x, r2.y, __uint_as_float(val >> GSA_GID_BITS), 0.f);
            }
            wave_lds_fence();
            const int cnt = min(kBatch, end - base);
            float4 n0 = s_q0[w][0], n1 = s_q1[w][0], n2 = s_q2[w][0];
            for (int j = 0; j < cnt; ++j) {
                const float4 q0 = n0, q1 = n1, q2 = n2;
                n0 = s_q0[w][j + 1]; n1 = s_q1[w][j + 1]; n2 = s_q2[w][j + 1];  // next record (slot 64 is padding)
                const uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z) & alive);
                if (m != 0) {
                    const float dx0 = q0.x - px0, dy0 = q0.y - py0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (m & (1u << g)) {
                            const float dx = dx0 - (float)((g & 1) * 8), dy = dy0 - (float)((g >> 1) * 8);
                            const float s = dx * (q0.z * dx + q0.w * dy) + (q1.x * dy) * dy;
                            const float a = fminf(GSA_MAX_ALPHA, q1.y * __builtin_amdgcn_exp2f(-s));
                            const bool valid = (s >= 0.f) && (a >= GSA_ALPHA_THRESHOLD);
                            const float Tg = T[g];
                            const float nT = Tg * (1.f - a);
                            const bool contrib = valid && (nT > GSA_TRANSMITTANCE_EPS);  // false once T < 0
                            const float wgt = contrib ? a * Tg : 0.f;
                            acc[g][0] += q1.z * wgt; acc[g][1] += q1.w * wgt; acc[g][2] += q2.x * wgt;
                            if (CH == 4) acc[g][CH - 1] += q2.y * wgt;
                            last[g] = contrib ? base + j : last[g];
                            // valid but not blended: the stop rule fired (or the pixel was finished already)
                            T[g] = contrib ? nT : (valid ? -fabsf(Tg) : Tg);
                        }
                    }
                }
                if ((j & 7) == 7) {  // refresh which quadrants still have unfinished pixels
                    alive = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) alive |= (__ballot(T[g] > 0.f) != 0ull) ? (1u << g) : 0u;
                    alive = __builtin_amdgcn_readfirstlane(alive);
                    if (alive == 0) break;
                }
            }
            wave_lds_fence();
        }

#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (inside[g]) {
                const int px = tx * kTile + (g & 1) * 8 + lx, py = ty * kTile + (g >> 1) * 8 + ly;
                const size_t pix = (size_t)py * width + px;
                const float Tf = fabsf(T[g]);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    render[pix * CH + k] = acc[g][k] + (background ? Tf * background[k] : 0.f);
                alpha_out[pix] = 1.f - Tf;
                final_T[pix] = Tf;  // kept beside alpha: 1 - (1 - T) loses the low bits of a small T
                last_id[pix] = last[g];
            }
        }
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
__global__ __launch_bounds__(kBlock, GSA_BWD_WAVES_PER_SIMD) void composite_bwd_kernel(
    int width, int height, int tile_w, int n_tiles,
    const float *__restrict__ splat, const uint32_t *__restrict__ sorted_gid,
    const int32_t *__restrict__ tile_offsets, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ last_id, const float *__restrict__ v_render,
    const float *__restrict__ v_alpha, float *__restrict__ v_splat) {
    __shared__ float4 s_q0[kWavesPerBlock][kBatch + 1];  // mean.x mean.y a/2 b
    __shared__ float4 s_q1[kWavesPerBlock][kBatch + 1];  // c/2 opacity r g
    __shared__ float4 s_q2[kWavesPerBlock][kBatch + 1];  // b depth quadrant-mask(bits) gaussian-index(bits)

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lx = lane & 7, ly = lane >> 3;
    const int tile = wave_tile(w, n_tiles);
    if (tile < 0) return;
    {
        const int start = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
        const int end = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
        if (end <= start) return;
        const int tx = tile % tile_w, ty = tile / tile_w;
        const float px0 = tx * kTile + lx + 0.5f, py0 = ty * kTile + ly + 0.5f;

        // per pixel: T = transmittance behind the current record; Wd = T_final (v_alpha - bg.v_out) - sum of
        // (colour . v_out) * weight over the splats behind -- what d(out)/d(alpha_i) multiplies by 1/(1-alpha_i)
        float T[4], Wd[4], vout[4][CH];
        int binf[4];
        int gmax[4];  // wave-uniform: last record any pixel of the quadrant blended
        int max_last = -1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int px = tx * kTile + (g & 1) * 8 + lx, py = ty * kTile + (g >> 1) * 8 + ly;
            const bool inside = px < width && py < height;
            const size_t pix = inside ? (size_t)py * width + px : 0;
            binf[g] = inside ? last_id[pix] : -1;
            const float Tf = inside ? final_T[pix] : 1.f;
            T[g] = Tf;
            float bgdot = 0.f;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                vout[g][k] = inside ? v_render[pix * CH + k] : 0.f;
                if (background) bgdot += background[k] * vout[g][k];
            }
            const float va = (inside && v_alpha) ? v_alpha[pix] : 0.f;
            Wd[g] = Tf * (va - bgdot);
            int mx = binf[g];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
            gmax[g] = __builtin_amdgcn_readfirstlane(mx);
            max_last = max(max_last, gmax[g]);
        }

        for (int top = max_last + 1; top > start; top -= kBatch) {
            const int cnt = min(kBatch, top - start);
            if (lane < cnt) {
                const uint32_t val = sorted_gid[top - 1 - lane];
                const uint32_t gi = val & kGidMask;
                const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)gi * GSA_SPLAT_STRIDE);
                float4 r0 = row[0], r1 = row[1], r2 = row[2];
                s_q0[w][lane] = make_float4(r0.x, r0.y, 0.5f * r0.z, r0.w);
                s_q1[w][lane] = make_float4(0.5f * r1.x, r1.y, r1.z, r1.w);
                s_q2[w][lane] = make_float4(r2.x, r2.y, __uint_as_float(val >> GSA_GID_BITS), __uint_as_float(gi));
            }
            wave_lds_fence();
            float4 n0 = s_q0[w][0], n1 = s_q1[w][0], n2 = s_q2[w][0];
            for (int j = 0; j < cnt; ++j) {
                const int rec = top - 1 - j;
                const float4 q0 = n0, q1 = n1, q2 = n2;
                n0 = s_q0[w][j + 1]; n1 = s_q1[w][j + 1]; n2 = s_q2[w][j + 1];  // next record (slot 64 is padding)
                uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z));
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (rec > gmax[g]) m &= ~(1u << g);
                m = __builtin_amdgcn_readfirstlane(m);
                if (m == 0) continue;

                float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // v_mx v_my v_a v_b v_c v_o v_r v_g
                float f8 = 0.f, f9 = 0.f;                               // v_b(lue), v_depth
                bool any = false;
                const float dx0 = q0.x - px0, dy0 = q0.y - py0;
                const float ha = q0.z, cb = q0.w, hc = q1.x, op = q1.y;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (m & (1u << g)) {
                        const float dx = dx0 - (float)((g & 1) * 8), dy = dy0 - (float)((g >> 1) * 8);
                        const float hadx = ha * dx, hcdy = hc * dy;
                        const float sx = hadx + cb * dy;              // a/2 dx + b dy
                        const float sigma = dx * sx + hcdy * dy;
                        const float vis = __builtin_amdgcn_exp2f(-kLog2e * sigma);
                        const float raw = op * vis;
                        const float a = fminf(GSA_MAX_ALPHA, raw);
                        const bool valid = (rec <= binf[g]) && (sigma >= 0.f) && (a >= GSA_ALPHA_THRESHOLD);
                        any = any || valid;
                        const float ra = __builtin_amdgcn_rcpf(1.f - a);
                        const float Tn = T[g] * ra;             // transmittance in front of this splat
                        const float fac = valid ? a * Tn : 0.f;
                        float cv = q1.z * vout[g][0] + q1.w * vout[g][1] + q2.x * vout[g][2];
                        if (CH == 4) cv += q2.y * vout[g][CH - 1];
                        const float v_a = Tn * cv + ra * Wd[g];
                        Wd[g] -= fac * cv;
                        f[6] += fac * vout[g][0]; f[7] += fac * vout[g][1]; f8 += fac * vout[g][2];
                        if (CH == 4) f9 += fac * vout[g][CH - 1];
                        const bool gate = valid && (raw <= GSA_MAX_ALPHA);
                        const float u = gate ? -raw * v_a : 0.f;    // d loss / d sigma
                        const float hu = 0.5f * u;
                        f[0] += u * (sx + hadx);                    // a dx + b dy
                        f[1] += u * (cb * dx + 2.f * hcdy);         // b dx + c dy
                        f[2] += (hu * dx) * dx;
                        f[3] += (u * dx) * dy;
                        f[4] += (hu * dy) * dy;
                        f[5] += gate ? vis * v_a : 0.f;
                        T[g] = valid ? Tn : T[g];
                    }
                }
                if (__ballot(any) == 0ull) continue;
                // one reduction and one 64-B atomic request per (tile, splat)
                const float r8 = reduce8(f, lane);
                const float t8 = wave_sum(f8);
                const float t9 = (CH == 4) ? wave_sum(f9) : 0.f;
                float *dst = v_splat + (size_t)__builtin_amdgcn_readfirstlane(__float_as_uint(q2.w)) * GSA_SPLAT_STRIDE;
                const bool lead = (lane & 7) == 0;
                const int off = lead ? (lane >> 3) : (lane == 1 ? 8 : 9);
                const float val = lead ? r8 : (lane == 1 ? t8 : t9);
                if (lead || lane == 1 || (CH == 4 && lane == 2)) atomicAdd(dst + off, val);
            }
            wave_lds_fence();
        }
    }
}

int tile_grid_blocks(int n_tiles) { return 8 * gsa_div_up(gsa_div_up(n_tiles, kWavesPerBlock), 8); }

}  // namespace

extern "C" int gsa_composite_fwd(int width, int height, int channels, int fill_masks, const float *splat,
                                 uint32_t *sorted_gid, const int32_t *tile_offsets, const float *background,
                                 float *render, float *alpha, float *final_T, int32_t *last_id,
                                 gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !render || !alpha || !final_T || !last_id, GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = tile_grid_blocks(n_tiles);
    if (channels == 3)
        composite_fwd_kernel<3><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, fill_masks, splat, sorted_gid,
                                                       tile_offsets, background, render, alpha, final_T, last_id);
    else
        composite_fwd_kernel<4><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, fill_masks, splat, sorted_gid,
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
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = tile_grid_blocks(n_tiles);
    if (channels == 3)
        composite_bwd_kernel<3><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid,
                                                       tile_offsets, background, final_T, last_id, v_render, v_alpha,
                                                       v_splat);
    else
        composite_bwd_kernel<4><<<grid, kBlock, 0, s>>>(width, height, tile_w, n_tiles, splat, sorted_gid,
                                                       tile_offsets, background, final_T, last_id, v_render, v_alpha,
                                                       v_splat);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}
