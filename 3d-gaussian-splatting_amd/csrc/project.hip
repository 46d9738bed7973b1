// project.hip -- stage 1 (projection + EWA + SH colour + tile count) and stage 5 (its backward).
//
// One thread per gaussian; 256-thread blocks (4 wave64).  The kernels are HBM-streaming: per gaussian
// the forward reads 44 B of geometry + up to 192 B of SH coefficients and writes one 64-B splat
// record + 16 B of integers.  The [N,16,3] SH rows (192-B stride) are staged through LDS by whole
// waves so the HBM side sees contiguous 16-B-per-lane accesses in both directions
// (a lane-per-row access pattern would touch 64 cache lines per instruction).
//
// Restates SURVEY.md Appendix A.2 / A.3 (forward) and A.7 (backward) -- the gsplat kernels behind
// /root/reference/src/backends/gsplat_backend.py:168-182.

#include "project_math.h"

namespace {

#ifndef GSA_PROJ_BLOCK
#define GSA_PROJ_BLOCK 128   // (256: the same 4-view step, project_bwd of a single view 0.072 -> 0.067 ms at 128; 512: slower)
#endif
constexpr int kBlock = GSA_PROJ_BLOCK;
#ifndef GSA_EXP_AUX
#define GSA_EXP_AUX 0
#endif
#ifndef GSA_PROJ_FWD_REVERSE
#define GSA_PROJ_FWD_REVERSE 1
#endif
#ifndef GSA_PROJ_BWD_WAVES
#define GSA_PROJ_BWD_WAVES 4   // the backward kernel's register budget: 128 VGPRs, four waves a SIMD (LDS allows five)
#endif
#ifndef GSA_PROJ_FWD_WAVES
#define GSA_PROJ_FWD_WAVES 1   // minimum waves per SIMD the forward kernel is compiled for (register budget)
#endif
constexpr int kRowQ = 12;      // float4 per [16,3] SH row
// Every wave stages and reads back ONLY its own rows of the LDS image, so no workgroup barrier is needed: LDS
// operations of one wave complete in order; the fences keep the compiler from reordering across the hand-over.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DEG>
__host__ __device__ constexpr int sh_needed_q() {  // float4 per row that hold the active coefficients
    return DEG < 0 ? 0 : (3 * (DEG + 1) * (DEG + 1) + 3) / 4;
}

// Row staging: a wave moves its 64 coefficient rows through LDS in two passes of 32 WHOLE rows (rows padded to 13
// float4 = 52 dwords: 16 consecutive lanes start in 16 different 4-bank groups), so every global access is a run of
// whole 128-B lines (column halves were tried: the 96-B row segments they write cost the gradient store a quarter of
// its speed).  26 KB per 256-thread block instead of 52 for all 64 rows at once: five blocks a CU fit, not three.
constexpr int kPassRows = 32, kRowPad = 13;
constexpr int kWaveLds = kPassRows * kRowPad;   // float4 per wave
template <int NQ>
__host__ __device__ constexpr int row_loads_per_lane() { return (kPassRows * NQ + 63) / 64; }   // per pass

// All of a wave's row loads, issued together (two passes x NPL whole-wave loads of consecutive float4; the row index
// is CLAMPED to the wave's last live row instead of predicated: predicated loads compile to one branch +
// s_waitcnt vmcnt(0) each, i.e. a dozen serial memory round trips).  stg is a plain float array with compile-time
// indices only: an array of float4 this size, or any dynamic index, is demoted to scratch memory.
template <int NQ>
__device__ __forceinline__ void issue_row_loads(const float4 *__restrict__ src_wave, int last_row, int lane,
                                                float (&stg)[8 * row_loads_per_lane<NQ>()]) {
    constexpr int NPL = row_loads_per_lane<NQ>();
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int idx = min(k * 64 + lane, kPassRows * NQ - 1);
            const int row = min(p * kPassRows + idx / NQ, last_row), col = idx % NQ;
            const float4 v = src_wave[row * kRowQ + col];
            const int o = 4 * (p * NPL + k);
            stg[o] = v.x; stg[o + 1] = v.y; stg[o + 2] = v.z; stg[o + 3] = v.w;
        }
}

// ... and through the wave's LDS image into row-per-lane order: lane l ends up with row l in c (every lane takes part;
// a wave only touches its own image, so no workgroup barrier).
template <int NQ, int NC>
__device__ __forceinline__ void rows_to_lanes(float4 *wrows, int lane, const float (&stg)[8 * row_loads_per_lane<NQ>()],
                                              float (&c)[NC]) {
    constexpr int NPL = row_loads_per_lane<NQ>();
#pragma unroll
    for (int k = 0; k < NC; ++k) c[k] = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p == 1) wave_lds_sync();   // the first 32 lanes have read their rows
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int idx = k * 64 + lane, o = 4 * (p * NPL + k);
            if (idx < kPassRows * NQ)
                wrows[(idx / NQ) * kRowPad + idx % NQ] = make_float4(stg[o], stg[o + 1], stg[o + 2], stg[o + 3]);
        }
        wave_lds_sync();
        if ((lane >> 5) == p) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 v = wrows[(lane & 31) * kRowPad + q];
                if (4 * q + 0 < NC) c[4 * q + 0] = v.x;
                if (4 * q + 1 < NC) c[4 * q + 1] = v.y;
                if (4 * q + 2 < NC) c[4 * q + 2] = v.z;
                if (4 * q + 3 < NC) c[4 * q + 3] = v.w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
#ifndef GSA_PROJ_BATCH_WAVES
#define GSA_PROJ_BATCH_WAVES 1   // register budget of the multi-camera instances (waves per SIMD the compiler must leave room for)
#endif
// PART (r05, gsa_project_fwd_parts): what of the stage's outputs a launch produces.  kPartGeometry: what the BINNING needs --
// radii, tile counts, depth keys, bin rectangles, quadrant boxes, the per-wave key ranges: 44 B in, 28 B out per gaussian and
// camera, no colour, no coefficient row, no record.  kPartRecords: what the SWEEPS need -- the 64-B splat record (with the
// colour) and the coefficient digest.  Both: the whole stage (gsa_project_fwd as before).  Each part recomputes the projection
// (the same function of the same inputs: the same bits), so the two launches are independent of each other: the records part
// runs on a stream of its own BESIDE the latency-bound binning chain, which only waits for the short geometry part -- the
// HBM is idle under the binning (0.13 ms at 1.7 TB/s of traffic), and a view alone no longer pays the 192-B rows in front of it.
constexpr int kPartGeometry = 1, kPartRecords = 2, kPartBoth = 3;
template <int DEG, bool STAGE, bool BATCH, int PART = kPartBoth>
__global__ __launch_bounds__(kBlock, (BATCH ? GSA_PROJ_BATCH_WAVES : GSA_PROJ_FWD_WAVES)) void project_fwd_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ opacities, const float *__restrict__ colors, int sh_coeffs,
    const float *__restrict__ viewmat_all, const float *__restrict__ intrin_all, int n_cams, int intrin_stride, int width,
    int height, int tile_w, int tile_h, int cull_mode, float *__restrict__ splat_all, int32_t *__restrict__ radii_all,
    int32_t *__restrict__ tiles_touched_all, uint32_t *__restrict__ depth_keys_all, uint32_t *__restrict__ tile_rects_all,
    float *__restrict__ sh_aux_all, int bin_tile, uint32_t *__restrict__ key_range_all, uint32_t *__restrict__ qbox_all) {
    // n_cams (gsa_project_fwd_batch): the SAME gaussians seen by several cameras -- the views of one optimisation step.  The
    // thread keeps its gaussian's geometry and coefficient row (236 B of the 376 B a single-camera launch moves) and runs
    // the per-camera part once per camera; camera c's outputs are the c-th slabs of [n_cams, ...] arrays.  Per camera the
    // arithmetic IS the single-camera kernel's (this one, with n_cams = 1): the results are the same bits.
    constexpr bool GEOM = (PART & kPartGeometry) != 0, RECS = (PART & kPartRecords) != 0;
    static_assert(RECS || (DEG < 0 && !STAGE), "the geometry part has one instance per camera mode: no colour");
    __shared__ float4 sh_lds[STAGE ? (kBlock / 64) * kWaveLds : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Blocks walk the gaussians from the LAST to the first (GSA_PROJ_FWD_REVERSE): the kernel that ran before this one in
    // a training loop is the previous view's project_bwd, which streamed the same parameter arrays front to back -- the
    // 256-MB memory-side cache then still holds the rows of the last ~0.45 M gaussians, and reading those first hits
    // them before this kernel's own traffic pushes them out (front to back they would be gone by the time it got there).
    // (the records part may be launched with FEWER workgroups than blocks of gaussians -- a low-occupancy stream beside the
    // binning chain of the same view, which needs the CUs' LDS: every workgroup then walks the blocks gridDim.x apart)
    const long n_blk = ((long)n + kBlock - 1) / kBlock;
    for (long blk_i = blockIdx.x; blk_i < (PART == kPartRecords ? n_blk : (long)blockIdx.x + 1); blk_i += gridDim.x) {
    if (PART == kPartRecords && blk_i != (long)blockIdx.x) wave_lds_sync();   // (the wave's LDS image: the previous block's stores have read it)
    const long blk = GSA_PROJ_FWD_REVERSE ? n_blk - 1 - blk_i : blk_i;
    const long i = blk * kBlock + threadIdx.x;

    // geometry first: these loads are in flight while the SH rows are staged (one memory round trip per wave less;
    // the kernel is latency-bound: ~80 % of its wave-cycles are waits on memory)
    float g_mean[3] = {0.f, 0.f, 0.f}, g_quat[4] = {1.f, 0.f, 0.f, 0.f}, g_scale[3] = {1.f, 1.f, 1.f}, g_opac = 0.f;
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { g_mean[k] = means[i * 3 + k]; g_scale[k] = scales[i * 3 + k]; }
        const float4 q4 = *reinterpret_cast<const float4 *>(quats + i * 4);
        g_quat[0] = q4.x; g_quat[1] = q4.y; g_quat[2] = q4.z; g_quat[3] = q4.w;
        g_opac = opacities[i];
    }

    // SH rows: all of the wave's loads are issued now; they pass through LDS further down, when the coefficients are
    // needed
    constexpr int NQ = sh_needed_q<DEG>();
    constexpr bool ROWS_IN = STAGE && DEG >= 1 && RECS;
    float stg[ROWS_IN ? 8 * row_loads_per_lane<NQ>() : 8];
    if constexpr (ROWS_IN) {
        const long wave_base = min(blk * kBlock + wave * 64, (long)n - 1);
        const int last = (int)min((long)63, (long)n - 1 - wave_base);
        issue_row_loads<NQ>(reinterpret_cast<const float4 *>(colors) + wave_base * kRowQ, last, lane, stg);
    }
    const bool inside = i < n;   // lanes beyond n stay for the wave-cooperative record store below
    const float opac = g_opac;
    const float sig_max = logf(fmaxf(opac, GSA_ALPHA_THRESHOLD) * 255.0f);  // alpha >= 1/255  <=>  sigma <= ln(255 o)
    const float ext = fminf(GSA_RADIUS_EXTEND, sqrtf(2.0f * sig_max));

    // the coefficient rows, through the wave's LDS image half by half: lane l ends up with row l (every lane takes
    // part, visible or not; a wave only touches its own 64 rows, so no workgroup barrier)
    constexpr int NC = DEG < 0 ? 1 : 3 * (DEG + 1) * (DEG + 1);
    float c[NC];
    if constexpr (ROWS_IN) rows_to_lanes<NQ, NC>(&sh_lds[wave * kWaveLds], lane, stg, c);

    const long n_waves = ((long)n + 63) / 64;
    const int cams = BATCH ? n_cams : 1;   // (one camera: a compile-time 1 -- the single-camera instances keep their code)
    for (int ci = 0; ci < cams; ++ci) {
    const float *__restrict__ viewmat = viewmat_all + 16 * ci;
    const float *__restrict__ intrin = intrin_all + (long)intrin_stride * ci;
    float *__restrict__ splat = splat_all + (long)ci * n * GSA_SPLAT_STRIDE;
    int32_t *__restrict__ radii = radii_all + (long)ci * n * 2;
    int32_t *__restrict__ tiles_touched = tiles_touched_all + (long)ci * n;
    uint32_t *__restrict__ depth_keys = depth_keys_all + (long)ci * n;
    uint32_t *__restrict__ tile_rects = tile_rects_all ? tile_rects_all + (long)ci * n * 2 : nullptr;
    float *__restrict__ sh_aux = sh_aux_all ? sh_aux_all + (long)ci * n * 12 : nullptr;
    uint32_t *__restrict__ key_range = key_range_all ? key_range_all + (long)ci * n_waves * 4 : nullptr;
    uint32_t *__restrict__ qbox = qbox_all ? qbox_all + (long)ci * n : nullptr;

    GsaCamera cam = gsa_load_camera(viewmat, intrin);
    GsaProj p = gsa_project_one(cam, g_mean, g_quat, g_scale, width, height);

    bool visible = inside && p.ok && (opac >= GSA_ALPHA_THRESHOLD);
    float rxf = ceilf(ext * sqrtf(fmaxf(p.c00, 0.f)));
    float ryf = ceilf(ext * sqrtf(fmaxf(p.c11, 0.f)));
    visible = visible && !(rxf <= GSA_RADIUS_CLIP && ryf <= GSA_RADIUS_CLIP);
    visible = visible && !(p.mx + rxf <= 0.f || p.mx - rxf >= (float)width || p.my + ryf <= 0.f ||
                           p.my - ryf >= (float)height);

    // what a culled gaussian leaves: a zero record, zero radii / count, the largest depth key
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0, o2 = o0;
    float4 aux[3] = {o0, o0, o0};
    int rx = 0, ry = 0, count = 0;
    uint32_t key = 0xFFFFFFFFu, rect = 0u;
    uint2 bin_pair = make_uint2(0u, 0u);   // (packed rectangle, record count) in BIN tiles -- what the fused binning reads
    uint32_t qb = 0u;                      // the quadrant bounding box of the alpha >= 1/255 ellipse (see below)
    if (visible) {
    // ---- colour (A.3)
    float r = 0.f, g = 0.f, b = 0.f;
    if constexpr (!RECS) {
        // (the geometry part: no colour)
    } else if constexpr (DEG < 0) {
        r = colors[i * 3]; g = colors[i * 3 + 1]; b = colors[i * 3 + 2];
    } else {
        constexpr int NB = (DEG + 1) * (DEG + 1);
        float dx = g_mean[0] - cam.campos[0], dy = g_mean[1] - cam.campos[1], dz = g_mean[2] - cam.campos[2];
        float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
        float B[NB];
        gsa_sh_basis<DEG>(dx * inv, dy * inv, dz * inv, B);
        if constexpr (!ROWS_IN) {
            const float *row = colors + i * (long)sh_coeffs * 3;
#pragma unroll
            for (int k = 0; k < NB * 3; ++k) c[k] = row[k];
        }
        r = g = b = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            r += B[j] * c[3 * j]; g += B[j] * c[3 * j + 1]; b += B[j] * c[3 * j + 2];
        }
        r = fmaxf(r + 0.5f, 0.f); g = fmaxf(g + 0.5f, 0.f); b = fmaxf(b + 0.5f, 0.f);
        if constexpr (DEG >= 1) {
            // sh_aux (optional, [N,12]): per channel d colour / d mean through the view direction (the SH sum's
            // gradient in the unit direction, projected off it, over the distance) and the clamp flag -- everything the
            // backward needs of the 192-B coefficient row, in 48 B (the coefficients are in registers here anyway)
#if GSA_EXP_AUX == 2   // experiment: stores without the computation
            if (sh_aux) aux[0] = aux[1] = aux[2] = make_float4(r, g, b, 1.f);
            if (false) {
#else
            if (sh_aux) {
#endif
                const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
                const float pre[3] = {r, g, b};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float vB[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) vB[j] = c[3 * j + ch];
                    float hx, hy, hz;
                    gsa_sh_basis_vjp<DEG>(ux, uy, uz, vB, hx, hy, hz);
                    const float dd = hx * ux + hy * uy + hz * uz;
                    aux[ch] = make_float4((hx - dd * ux) * inv, (hy - dd * uy) * inv, (hz - dd * uz) * inv,
                                          pre[ch] > 0.f ? 1.f : 0.f);
                }
            }
        }
    }

    // ---- tile count (A.4)
    rx = (int)rxf; ry = (int)ryf;
    GsaTileRect tr = gsa_tile_rect(p.mx, p.my, rx, ry, tile_w, tile_h);
    if (!GEOM) {
        // (the records part needs the rectangle for record word 11 only)
    } else if (cull_mode == 0) {
        count = (tr.x1 - tr.x0) * (tr.y1 - tr.y0);
    } else {
        count = 0;
        for (int ty = tr.y0; ty < tr.y1; ++ty)
            for (int tx = tr.x0; tx < tr.x1; ++tx)
                count += gsa_quadrant_mask(p.mx, p.my, p.qa, p.qb, p.qc, sig_max, tx, ty) != 0u ? 1 : 0;
    }
    // record word 11: the tile rectangle, packed (x0 | y0 << 10 | width << 20; 0 when a coordinate needs more than
    // 10 bits) -- tile_emit then needs this one line per gaussian instead of the mean AND the radii
    const bool packable = tile_w < 1024 && tile_h < 1024;
    rect = packable ? ((uint32_t)tr.x0 | ((uint32_t)tr.y0 << 10) | ((uint32_t)(tr.x1 - tr.x0) << 20)) : 0u;
    bin_pair = make_uint2(count > 0 ? rect : 0u, (uint32_t)count);
    if (GEOM && bin_tile != GSA_TILE_SIZE) {   // coarser bins (cull_mode 0 only): the same rule on a grid of bin_tile pixels
        const int bw = (width + bin_tile - 1) / bin_tile, bh = (height + bin_tile - 1) / bin_tile;
        const GsaTileRect br = gsa_tile_rect(p.mx, p.my, rx, ry, bw, bh, bin_tile);
        const int bc = (br.x1 - br.x0) * (br.y1 - br.y0);
        bin_pair = make_uint2(bc > 0 ? ((uint32_t)br.x0 | ((uint32_t)br.y0 << 10) | ((uint32_t)(br.x1 - br.x0) << 20)) : 0u,
                              (uint32_t)bc);
    }
    // qbox (optional): the 8 x 8 quadrants whose pixel centres the BOUNDING BOX of the alpha >= 1/255 ellipse covers --
    // half extents sqrt(2 t Sigma_xx), sqrt(2 t Sigma_yy) with t = ln(255 o), the margin of the compositor's reach test --
    // packed first column | first row << 10 | (columns - 1) << 20 | (rows - 1) << 26; 0xFFFFFFFF: more than 64 quadrants
    // a side or a grid beyond 1024.  What the record placement turns into per-record quadrant masks without looking at the
    // conic again (gsa_tile_bin16): of the records of a 32 x 32 bin 22 % reach a given quadrant, of those inside the box 78 %.
    if (GEOM && qbox) {
        const float t2 = 2.f * (sig_max * 1.0001f + 1e-4f);
        const float hx = __builtin_amdgcn_sqrtf(t2 * fmaxf(p.c00, 0.f)) * 1.00001f + 1e-3f;
        const float hy = __builtin_amdgcn_sqrtf(t2 * fmaxf(p.c11, 0.f)) * 1.00001f + 1e-3f;
        // pixel centres 8 q + 0.5 + i in [m - h, m + h]: (a box one pixel larger than needed at worst)
        const float qw_max = (float)((width + 7) / 8 - 1), qh_max = (float)((height + 7) / 8 - 1);
        const float x0 = fminf(fmaxf(floorf((p.mx - hx - 0.5f) * 0.125f), 0.f), qw_max);
        const float x1 = fminf(fmaxf(floorf((p.mx + hx - 0.5f) * 0.125f), 0.f), qw_max);
        const float y0 = fminf(fmaxf(floorf((p.my - hy - 0.5f) * 0.125f), 0.f), qh_max);
        const float y1 = fminf(fmaxf(floorf((p.my + hy - 0.5f) * 0.125f), 0.f), qh_max);
        const uint32_t ux0 = (uint32_t)x0, uy0 = (uint32_t)y0, uw = (uint32_t)(x1 - x0), uh = (uint32_t)(y1 - y0);
        qb = (qw_max < 1024.f && qh_max < 1024.f && uw < 64u && uh < 64u) ? (ux0 | (uy0 << 10) | (uw << 20) | (uh << 26))
                                                                           : 0xFFFFFFFFu;
    }
    o0 = make_float4(p.mx, p.my, p.qa, p.qb);
    o1 = make_float4(p.qc, opac, r, g);
    o2 = make_float4(b, p.z, sig_max, __uint_as_float(rect));
    key = __float_as_uint(p.z);
    }   // visible

    // ---- the 64-B records: through the wave's (now free) LDS rows, so that every store instruction writes 1 KB of
    // consecutive addresses instead of 16 B at a 64-B stride
    if constexpr (!RECS) {
        // (the geometry part writes no record)
    } else if constexpr (STAGE && DEG >= 1) {
        const long wave_base = blk * kBlock + wave * 64;
        const int rows = (int)min((long)64, (long)n - wave_base);
        float4 *wl = &sh_lds[wave * kWaveLds];
        wave_lds_sync();   // every lane has read its SH row
        wl[lane * 4 + 0] = o0; wl[lane * 4 + 1] = o1; wl[lane * 4 + 2] = o2; wl[lane * 4 + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
        wave_lds_sync();
        float4 *outw = reinterpret_cast<float4 *>(splat + wave_base * GSA_SPLAT_STRIDE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = k * 64 + lane;
            if (idx < rows * 4) outw[idx] = wl[idx];
        }
#if GSA_EXP_AUX == 1   // experiment: the computation without the stores
        if (DEG >= 1 && sh_aux && n == 12345) {
#else
        if (DEG >= 1 && sh_aux) {   // the digest rows leave the same way (zeros for culled gaussians)
#endif
            wave_lds_sync();
            wl[lane * 3 + 0] = aux[0]; wl[lane * 3 + 1] = aux[1]; wl[lane * 3 + 2] = aux[2];
            wave_lds_sync();
            float4 *auxw = reinterpret_cast<float4 *>(sh_aux) + wave_base * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = k * 64 + lane;
                if (idx < rows * 3) auxw[idx] = wl[idx];
            }
        }
    } else if (inside) {
        if (DEG >= 1 && sh_aux) {
            float4 *arow = reinterpret_cast<float4 *>(sh_aux) + i * 3;
            arow[0] = aux[0]; arow[1] = aux[1]; arow[2] = aux[2];
        }
        float4 *out = reinterpret_cast<float4 *>(splat + i * GSA_SPLAT_STRIDE);
        out[0] = o0; out[1] = o1; out[2] = o2; out[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // key_range (optional): per WAVE the smallest / largest depth key of its visible gaussians (0xFFFFFFFF / 0 when it has
    // none) and the keys of its lanes 0 and 32 (a sample of the view's keys), one plain 16-byte store -- the bucketed depth
    // ordering builds THIS view's depth table from the ceil(n / 64) entries (gsa_depth_order_msd's key_range_dev).
    // (Atomic min / max into a few shards were tried first: the waves of the kernel's first round all find the initial
    // values and all add -- 31 k atomics on two cache lines tripled this kernel's time, 64 shards on lines of their own
    // still cost 15 %.)
    if (GEOM && key_range) {
        const uint32_t s0 = (uint32_t)__shfl((int)key, 0), s1 = (uint32_t)__shfl((int)key, 32);
        uint32_t lo = key, hi = key == 0xFFFFFFFFu ? 0u : key;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
        }
        // (the buffer holds ceil(n / 64) entries: the waves of the last block that lie wholly beyond n have none)
        if (lane == 0 && blk * kBlock + wave * 64 < (long)n)
            reinterpret_cast<uint4 *>(key_range)[blk * (kBlock / 64) + wave] = make_uint4(lo, hi, s0, s1);
    }
    if (GEOM && inside) {
        *reinterpret_cast<int2 *>(radii + 2 * i) = make_int2(rx, ry);
        tiles_touched[i] = count;
        depth_keys[i] = key;
        // dense (rectangle, count) pairs for the fused binning: one 8-B gather per gaussian brings both into depth order
        if (tile_rects) reinterpret_cast<uint2 *>(tile_rects)[i] = bin_pair;
        if (qbox) qbox[i] = qb;
    }
    }   // cameras
    }   // blocks of this workgroup
}

// HOT gaussians (composite.hip, hot_row): the reverse sweep spreads the atomics of a gaussian whose splat covers many tiles over
// GSA_HOT_REPLICAS rows of the hot region behind the n gradient records, and leaves entry + 1 in the reserved word
// GSA_HOT_MARK_FIELD of the gaussian's own record.  The record is read here anyway: a set mark adds the replicas (a few
// thousand gaussians of a view from inside the scene, none of the bench scene C3).  Any other value of the word -- zero from
// the sweeps' clearing; whatever a caller's own records hold, short of an integer in 1 .. GSA_HOT_ENTRIES -- leaves the record alone.
__device__ __forceinline__ void fold_hot_replicas(const float *__restrict__ v_splat, long n, float4 &v0, float4 &v1, float4 &v2) {
    static_assert(GSA_HOT_MARK_FIELD == 11, "the mark is the fourth word of the record's third quad");
    const uint32_t mark = __float_as_uint(v2.w);
    if (mark - 1u < (uint32_t)GSA_HOT_ENTRIES) {
        const float4 *rep = reinterpret_cast<const float4 *>(v_splat + (n + (long)(mark - 1u)) * GSA_SPLAT_STRIDE);
#pragma unroll 1
        for (int k = 0; k < GSA_HOT_REPLICAS; ++k) {   // (replica k of entry e: row n + k GSA_HOT_ENTRIES + e)
            const float4 *r = rep + (long)k * GSA_HOT_ENTRIES * (GSA_SPLAT_STRIDE / 4);
            const float4 a = r[0], b = r[1], c = r[2];
            v0.x += a.x; v0.y += a.y; v0.z += a.z; v0.w += a.w;
            v1.x += b.x; v1.y += b.y; v1.z += b.z; v1.w += b.w;
            v2.x += c.x; v2.y += c.y;
        }
    }
}

// The chain rule of the projection for ONE gaussian and ONE camera: the sweep's raw moments (v0, v1, v2: the gradient
// record, composite.hip) and the view-direction part of the colour gradient (vdir) -> d loss / d (mean, quaternion,
// scale, opacity).  One function for the single-camera and the multi-camera backward kernels: the same arithmetic.
__device__ __forceinline__ void project_chain_bwd(const GsaCamera &cam, const float (&g_mean)[3], const float (&g_quat)[4],
                                                  const float (&g_scale)[3], float opac, int width, int height,
                                                  const float4 v0, const float4 v1, const float4 v2, const float (&vdir)[3],
                                                  float (&vm)[3], float (&vq)[4], float (&vs)[3], float &vo) {
    GsaProj p = gsa_project_one(cam, g_mean, g_quat, g_scale, width, height);
    // The sweep leaves the raw moments of u = d loss / d sigma over d = mean2d - pixel (composite.hip):
    // sigma = (a dx^2 + c dy^2) / 2 + b dx dy and alpha = o * exp(-sigma) make every gradient linear in them.
    const float m_dx = v0.x, m_dy = v0.y, m_dxdx = v0.z, m_dxdy = v0.w, m_dydy = v1.x, m_u = v1.y;
    const float s_opac = opac;
    float v_mx = p.qa * m_dx + p.qb * m_dy, v_my = p.qb * m_dx + p.qc * m_dy;
    float v_a = 0.5f * m_dxdx, v_b = m_dxdy, v_c = 0.5f * m_dydy;
    vo = -m_u / s_opac;   // d alpha / d o = alpha / o  and  u = -alpha * v_alpha
    float v_depth = v2.y;


    // ---- conic -> blurred cov2d:  G = -Q V Q,  V = [[v_a, v_b/2], [v_b/2, v_c]]
    float a = p.qa, b = p.qb, c = p.qc, hb = 0.5f * v_b;
    float q00 = a * v_a + b * hb, q01 = a * hb + b * v_c;
    float q10 = b * v_a + c * hb, q11 = b * hb + c * v_c;
    float G00 = -(q00 * a + q01 * b);
    float G01 = -(q00 * b + q01 * c);
    float G11 = -(q10 * b + q11 * c);

    // ---- cov2d = J Sc J^T
    float J00 = p.J00, J02 = p.J02, J11 = p.J11, J12 = p.J12;
    float vS00 = J00 * J00 * G00;
    float vS01 = J00 * J11 * G01;
    float vS02 = J00 * (G00 * J02 + G01 * J12);
    float vS11 = J11 * J11 * G11;
    float vS12 = J11 * (G01 * J02 + G11 * J12);
    float vS22 = J02 * J02 * G00 + 2.f * J02 * J12 * G01 + J12 * J12 * G11;
    float T00 = J00 * p.S00 + J02 * p.S02, T01 = J00 * p.S01 + J02 * p.S12, T02 = J00 * p.S02 + J02 * p.S22;
    float T10 = J11 * p.S01 + J12 * p.S02, T11 = J11 * p.S11 + J12 * p.S12, T12 = J11 * p.S12 + J12 * p.S22;
    float vJ00 = 2.f * (G00 * T00 + G01 * T10);
    float vJ02 = 2.f * (G00 * T02 + G01 * T12);
    float vJ11 = 2.f * (G01 * T01 + G11 * T11);
    float vJ12 = 2.f * (G01 * T02 + G11 * T12);

    // ---- camera-space mean
    float fx = cam.fx, fy = cam.fy, rz = p.rz, rz2 = p.rz2, rz3 = p.rz2 * p.rz;
    float vx = fx * rz * v_mx;
    float vy = fy * rz * v_my;
    float vz = -(fx * p.x * v_mx + fy * p.y * v_my) * rz2 + v_depth;
    if (p.x_in) vx += -fx * rz2 * vJ02; else vz += -fx * rz3 * vJ02 * p.tx;
    if (p.y_in) vy += -fy * rz2 * vJ12; else vz += -fy * rz3 * vJ12 * p.ty;
    vz += -fx * rz2 * vJ00 - fy * rz2 * vJ11 + 2.f * fx * p.tx * rz3 * vJ02 + 2.f * fy * p.ty * rz3 * vJ12;
    const float *R = cam.R;
    vm[0] = R[0] * vx + R[3] * vy + R[6] * vz;
    vm[1] = R[1] * vx + R[4] * vy + R[7] * vz;
    vm[2] = R[2] * vx + R[5] * vy + R[8] * vz;
    vm[0] += vdir[0]; vm[1] += vdir[1]; vm[2] += vdir[2];

    // ---- Sigma_c = R Sigma R^T  ->  vSig = R^T vSc R
    float U00 = vS00 * R[0] + vS01 * R[3] + vS02 * R[6], U01 = vS00 * R[1] + vS01 * R[4] + vS02 * R[7],
          U02 = vS00 * R[2] + vS01 * R[5] + vS02 * R[8];
    float U10 = vS01 * R[0] + vS11 * R[3] + vS12 * R[6], U11 = vS01 * R[1] + vS11 * R[4] + vS12 * R[7],
          U12 = vS01 * R[2] + vS11 * R[5] + vS12 * R[8];
    float U20 = vS02 * R[0] + vS12 * R[3] + vS22 * R[6], U21 = vS02 * R[1] + vS12 * R[4] + vS22 * R[7],
          U22 = vS02 * R[2] + vS12 * R[5] + vS22 * R[8];
    float W[9];
    W[0] = R[0] * U00 + R[3] * U10 + R[6] * U20; W[1] = R[0] * U01 + R[3] * U11 + R[6] * U21;
    W[2] = R[0] * U02 + R[3] * U12 + R[6] * U22;
    W[3] = R[1] * U00 + R[4] * U10 + R[7] * U20; W[4] = R[1] * U01 + R[4] * U11 + R[7] * U21;
    W[5] = R[1] * U02 + R[4] * U12 + R[7] * U22;
    W[6] = R[2] * U00 + R[5] * U10 + R[8] * U20; W[7] = R[2] * U01 + R[5] * U11 + R[8] * U21;
    W[8] = R[2] * U02 + R[5] * U12 + R[8] * U22;

    // ---- Sigma = M M^T, M = Rq diag(s):  vM = 2 vSig M
    const float *Rq = p.Rq;
    float s[3] = {p.sx, p.sy, p.sz};
    float vM[9];
#pragma unroll
    for (int r_ = 0; r_ < 3; ++r_)
#pragma unroll
        for (int c_ = 0; c_ < 3; ++c_)
            vM[r_ * 3 + c_] = 2.f * (W[r_ * 3 + 0] * Rq[0 * 3 + c_] + W[r_ * 3 + 1] * Rq[1 * 3 + c_] +
                                     W[r_ * 3 + 2] * Rq[2 * 3 + c_]) * s[c_];
    float vR[9];
#pragma unroll
    for (int c_ = 0; c_ < 3; ++c_) {
        vs[c_] = Rq[0 * 3 + c_] * vM[0 * 3 + c_] + Rq[1 * 3 + c_] * vM[1 * 3 + c_] + Rq[2 * 3 + c_] * vM[2 * 3 + c_];
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_) vR[r_ * 3 + c_] = vM[r_ * 3 + c_] * s[c_];
    }
    // ---- rotation matrix -> normalised quaternion -> raw quaternion
    float w = p.qw, x = p.qx, y = p.qy, z = p.qz;
    float gw = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    float gx = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    float gy = 2.f * (-2.f * y * (vR[0] + vR[8]) + x * (vR[1] + vR[3]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    float gz = 2.f * (-2.f * z * (vR[0] + vR[4]) + x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) + w * (vR[3] - vR[1]));
    float dotq = gw * w + gx * x + gy * y + gz * z;
    vq[0] = (gw - dotq * w) * p.qinv; vq[1] = (gx - dotq * x) * p.qinv;
    vq[2] = (gy - dotq * y) * p.qinv; vq[3] = (gz - dotq * z) * p.qinv;
}

// ------------------------------------------------------------------------------------------------
// backward (A.7)
// ------------------------------------------------------------------------------------------------
template <int DEG, bool STAGE, bool FACTORED, bool AUX>
__global__ __launch_bounds__(kBlock, GSA_PROJ_BWD_WAVES) void project_bwd_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ opacities, const float *__restrict__ colors, int sh_coeffs,
    const float *__restrict__ viewmat, const float *__restrict__ intrin, int width, int height,
    const int32_t *__restrict__ radii, const float *__restrict__ v_splat, float *__restrict__ v_means,
    float *__restrict__ v_quats, float *__restrict__ v_scales, float *__restrict__ v_opacities,
    float *__restrict__ v_colors, int accumulate, float *__restrict__ v_rgb_out, const float *__restrict__ sh_aux) {
    // The forward's splat records are NOT read back (64 MB of lines at 1 M for 32 B of use each): the projection is
    // recomputed from the raw parameters anyway, the opacity is an input, and the colour clamp mask is re-derived
    // from the SH sum evaluated exactly as the forward evaluates it.
    // AUX: the forward left sh_aux[N,12] (see there): the coefficient rows are not read at all (48 B instead of 192).
    // v_rgb_out (optional, [N,3]): FACTORED colour gradient -- instead of the [N,K,3] SH-coefficient gradient
    // (v_colors, then unused) write only d loss / d rgb after the clamp mask.  The SH gradient is the outer product of
    // the view's SH basis with it (gsa_sh_grad_expand), so 3 floats per gaussian carry what 48 would.
    __shared__ float4 sh_lds[STAGE && ((DEG >= 1 && !AUX) || !FACTORED) ? (kBlock / 64) * kWaveLds : 1];   // rows in, rows out
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long wave_base = (long)blockIdx.x * kBlock + wave * 64;
    const long rows = min((long)64, (long)n - wave_base);
    constexpr int NB = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
    static_assert(!AUX || DEG >= 1, "sh_aux only exists for view-dependent colour");

    // per-gaussian inputs first: in flight while the SH rows are staged (see the forward kernel)
    const bool active = i < n;
    float g_mean[3] = {0.f, 0.f, 0.f}, g_quat[4] = {1.f, 0.f, 0.f, 0.f}, g_scale[3] = {1.f, 1.f, 1.f};
    int g_rx = 0, g_ry = 0;
    float4 g_v0 = make_float4(0.f, 0.f, 0.f, 0.f), g_v1 = g_v0, g_v2 = g_v0;
    float g_opac = 1.f;
    float4 g_aux[3] = {g_v0, g_v0, g_v0};
    if (active) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { g_mean[k] = means[i * 3 + k]; g_scale[k] = scales[i * 3 + k]; }
        const float4 q4 = *reinterpret_cast<const float4 *>(quats + i * 4);
        g_quat[0] = q4.x; g_quat[1] = q4.y; g_quat[2] = q4.z; g_quat[3] = q4.w;
        g_rx = radii[2 * i]; g_ry = radii[2 * i + 1];
        const float4 *vrow = reinterpret_cast<const float4 *>(v_splat + i * GSA_SPLAT_STRIDE);
        g_v0 = vrow[0]; g_v1 = vrow[1]; g_v2 = vrow[2];
        g_opac = opacities[i];
        if constexpr (AUX) {
            const float4 *arow = reinterpret_cast<const float4 *>(sh_aux) + i * 3;
            g_aux[0] = arow[0]; g_aux[1] = arow[1]; g_aux[2] = arow[2];
        }
    }
    // SH rows (staged, without the forward's digest): the wave's loads are issued now, in two column halves (see the
    // forward kernel), and pass through LDS below
    constexpr int NQ = sh_needed_q<DEG>();
    constexpr bool ROWS_IN = STAGE && DEG >= 1 && !AUX;
    float stg[ROWS_IN ? 8 * row_loads_per_lane<NQ>() : 8];
    if constexpr (ROWS_IN) {
        const long src_base = min(wave_base, (long)n - 1);
        const int last = (int)min((long)63, (long)n - 1 - src_base);
        issue_row_loads<NQ>(reinterpret_cast<const float4 *>(colors) + src_base * kRowQ, last, lane, stg);
    }
    // Every per-gaussian load above is issued HERE, together: left alone the compiler sinks all but the radii below the
    // visibility test (they are only used there), which makes two dependent memory round trips out of one.
    if constexpr (AUX)
        asm volatile("" : "+v"(g_mean[0]), "+v"(g_scale[0]), "+v"(g_quat[0]), "+v"(g_v0.x), "+v"(g_v1.x), "+v"(g_v2.x),
                          "+v"(g_opac), "+v"(g_aux[0].x), "+v"(g_aux[1].x), "+v"(g_aux[2].x), "+v"(g_rx));
    else
        asm volatile("" : "+v"(g_mean[0]), "+v"(g_scale[0]), "+v"(g_quat[0]), "+v"(g_v0.x), "+v"(g_v1.x), "+v"(g_v2.x),
                          "+v"(g_opac), "+v"(g_rx));

    constexpr int NC = DEG < 0 ? 1 : 3 * (DEG + 1) * (DEG + 1);
    float cf[NC];
    if constexpr (ROWS_IN) rows_to_lanes<NQ, NC>(&sh_lds[wave * kWaveLds], lane, stg, cf);

    bool visible = active && g_rx > 0 && g_ry > 0;
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vo = 0.f;
    float vc[NB * 3], vr_keep[3] = {0.f, 0.f, 0.f};   // (vc: only the unstaged path keeps the whole row in registers)
    float ud[3] = {0.f, 0.f, 0.f};   // unit view direction (the staged output re-evaluates the SH basis from it: 3 live
                                     // registers across the projection chain instead of 16)
#pragma unroll
    for (int k = 0; k < NB * 3; ++k) vc[k] = 0.f;

    if (visible) {
        GsaCamera cam = gsa_load_camera(viewmat, intrin);
        float4 v0 = g_v0, v1 = g_v1, v2 = g_v2;
        fold_hot_replicas(v_splat, (long)n, v0, v1, v2);
        float v_rgb[3] = {v1.z, v1.w, v2.x};
        // ---- colour FIRST: the coefficient row (48 registers) is dead before the projection chain needs its own
        float vdir[3] = {0.f, 0.f, 0.f};   // d loss / d mean through the view direction
        if constexpr (DEG < 0) {
            vc[0] = v_rgb[0]; vc[1] = v_rgb[1]; vc[2] = v_rgb[2];
        } else {
            float dx = g_mean[0] - cam.campos[0], dy = g_mean[1] - cam.campos[1], dz = g_mean[2] - cam.campos[2];
            float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            float ux = dx * inv, uy = dy * inv, uz = dz * inv;
            ud[0] = ux; ud[1] = uy; ud[2] = uz;
            float B[NB];
            gsa_sh_basis<DEG>(ux, uy, uz, B);
            if constexpr (AUX) {
                float vr[3] = {g_aux[0].w != 0.f ? v_rgb[0] : 0.f, g_aux[1].w != 0.f ? v_rgb[1] : 0.f,
                               g_aux[2].w != 0.f ? v_rgb[2] : 0.f};
                vr_keep[0] = vr[0]; vr_keep[1] = vr[1]; vr_keep[2] = vr[2];
                if constexpr (!FACTORED && !STAGE) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        vc[3 * j] = B[j] * vr[0]; vc[3 * j + 1] = B[j] * vr[1]; vc[3 * j + 2] = B[j] * vr[2];
                    }
                }
                vdir[0] += g_aux[0].x * vr[0] + g_aux[1].x * vr[1] + g_aux[2].x * vr[2];
                vdir[1] += g_aux[0].y * vr[0] + g_aux[1].y * vr[1] + g_aux[2].y * vr[2];
                vdir[2] += g_aux[0].z * vr[0] + g_aux[1].z * vr[1] + g_aux[2].z * vr[2];
            } else {
            if constexpr (!ROWS_IN) {
                const float *row = colors + i * (long)sh_coeffs * 3;
#pragma unroll
                for (int k = 0; k < NB * 3; ++k) cf[k] = row[k];
            }
            // clamp mask of the forward's  max(sum_j B_j c_j + 0.5, 0)  (same loop, same order, same contraction)
            float cr = 0.f, cg = 0.f, cb = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                cr += B[j] * cf[3 * j]; cg += B[j] * cf[3 * j + 1]; cb += B[j] * cf[3 * j + 2];
            }
            float vr[3] = {fmaxf(cr + 0.5f, 0.f) > 0.f ? v_rgb[0] : 0.f, fmaxf(cg + 0.5f, 0.f) > 0.f ? v_rgb[1] : 0.f,
                           fmaxf(cb + 0.5f, 0.f) > 0.f ? v_rgb[2] : 0.f};
            vr_keep[0] = vr[0]; vr_keep[1] = vr[1]; vr_keep[2] = vr[2];
            if constexpr (!FACTORED && !(STAGE && DEG >= 1)) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    vc[3 * j] = B[j] * vr[0]; vc[3 * j + 1] = B[j] * vr[1]; vc[3 * j + 2] = B[j] * vr[2];
                }
            }
            if constexpr (DEG >= 1) {
                float vB[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    vB[j] = cf[3 * j] * vr[0] + cf[3 * j + 1] * vr[1] + cf[3 * j + 2] * vr[2];
                float hx, hy, hz;
                gsa_sh_basis_vjp<DEG>(ux, uy, uz, vB, hx, hy, hz);
                float dd = hx * ux + hy * uy + hz * uz;
                vdir[0] += (hx - dd * ux) * inv; vdir[1] += (hy - dd * uy) * inv; vdir[2] += (hz - dd * uz) * inv;
            }
            }   // !AUX
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from interleaving the two chains: register pressure)

        project_chain_bwd(cam, g_mean, g_quat, g_scale, g_opac, width, height, v0, v1, v2, vdir, vm, vq, vs, vo);

    }

    // accumulate (wave-uniform): add to what the output buffers hold instead of overwriting them -- the views of a
    // batch then sum into ONE gradient bucket with no separate accumulation pass
    if (active) {
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { vm[k] += v_means[i * 3 + k]; vs[k] += v_scales[i * 3 + k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) vq[k] += v_quats[i * 4 + k];
            vo += v_opacities[i];
        }
        v_means[i * 3] = vm[0]; v_means[i * 3 + 1] = vm[1]; v_means[i * 3 + 2] = vm[2];
        v_quats[i * 4] = vq[0]; v_quats[i * 4 + 1] = vq[1]; v_quats[i * 4 + 2] = vq[2]; v_quats[i * 4 + 3] = vq[3];
        v_scales[i * 3] = vs[0]; v_scales[i * 3 + 1] = vs[1]; v_scales[i * 3 + 2] = vs[2];
        v_opacities[i] = vo;
    }

    // ---- colour gradient rows: every element written (zeros above the active degree / when culled)
    if constexpr (FACTORED) {
        if (active) { v_rgb_out[i * 3] = vr_keep[0]; v_rgb_out[i * 3 + 1] = vr_keep[1]; v_rgb_out[i * 3 + 2] = vr_keep[2]; }
    } else if constexpr (DEG < 0) {
        if (active) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v_colors[i * 3 + k] = vc[k] + (accumulate ? v_colors[i * 3 + k] : 0.f);
        }
    } else if constexpr (STAGE) {
        // rows leave through the wave's LDS image, 32 whole rows a pass: whole-wave stores of consecutive float4;
        // element e = 3 j + c of a row is basis_j * (d loss / d rgb_c), formed here
        float4 *wrows = &sh_lds[wave * kWaveLds];
        float4 *dst = reinterpret_cast<float4 *>(v_colors) + wave_base * kRowQ;
        float B[NB];
        gsa_sh_basis<DEG>(ud[0], ud[1], ud[2], B);   // (culled gaussians: vr_keep is zero)
        auto elem = [&](int e) { return e < NB * 3 ? B[(e / 3) % NB] * vr_keep[e % 3] : 0.f; };
        if constexpr (ROWS_IN) wave_lds_sync();   // all lanes have consumed their coefficient rows
        // (an opaque copy of the lane index: the row / column arithmetic below repeats the staging-in's, and sharing it
        // would keep a dozen 64-bit addresses alive across the whole kernel -- spilled at this register budget)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        constexpr int NPL = kPassRows * kRowQ / 64;   // 6 float4 per lane and pass
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p == 1) wave_lds_sync();   // the first 32 rows have left
            if ((lane >> 5) == p) {
#pragma unroll
                for (int q = 0; q < kRowQ; ++q)
                    wrows[(lane & 31) * kRowPad + q] = make_float4(elem(4 * q), elem(4 * q + 1), elem(4 * q + 2), elem(4 * q + 3));
            }
            wave_lds_sync();
            const int live = (int)rows - p * kPassRows;   // rows of this pass that exist (wave-uniform)
            float old[4 * NPL];
            if (accumulate && live > 0) {   // (rows clamped, not predicated: all six loads in flight at once)
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int idx = k * 64 + lane, r_ = min(idx / kRowQ, live - 1), c_ = idx % kRowQ;
                    const float4 o = dst[(p * kPassRows + r_) * kRowQ + c_];
                    old[4 * k] = o.x; old[4 * k + 1] = o.y; old[4 * k + 2] = o.z; old[4 * k + 3] = o.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4 * NPL; ++k) old[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int idx = k * 64 + lane, r_ = idx / kRowQ, c_ = idx - r_ * kRowQ;
                if (r_ < live) {
                    float4 v = wrows[r_ * kRowPad + c_];
                    v.x += old[4 * k]; v.y += old[4 * k + 1]; v.z += old[4 * k + 2]; v.w += old[4 * k + 3];
                    dst[(p * kPassRows) * kRowQ + idx] = v;
                }
            }
        }
    } else if (active) {
        float *row = v_colors + i * (long)sh_coeffs * 3;
        if (!accumulate)
            for (int k = 0; k < sh_coeffs * 3; ++k) row[k] = 0.f;
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) row[k] = vc[k] + (accumulate ? row[k] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// backward for the cameras of one optimisation step in ONE launch (gsa_project_bwd_batch)
// ------------------------------------------------------------------------------------------------
// The B views of a step are rendered from the same parameters and their gradients are summed.  One launch per view reads a
// gaussian's geometry B times, read-modify-writes the geometry gradients B - 1 times and -- in the factored form the step
// uses -- leaves 3 floats per view for one more kernel to expand into the 192-B coefficient gradient.  This kernel keeps the
// gaussian in registers, runs the per-camera chain rule (project_chain_bwd: the single-camera kernel's arithmetic, summed in
// camera order exactly as the accumulating launches sum) once per camera over that camera's gradient record and
// coefficient digest, accumulates  sum_c basis_j(dir_c) x d loss / d rgb_c  in registers, and writes every gradient ONCE:
// 760 B per gaussian at four cameras where four launches + gsa_sh_grad_expand move 1308.
// View-dependent SH colour with the forward's digest only (sh_degree >= 1, sh_aux), rows of 16 coefficients.
constexpr int kMaxBatchCams = 8;
struct BwdCamRows {
    const int32_t *radii[kMaxBatchCams];
    const float *v_splat[kMaxBatchCams];
    const float *sh_aux[kMaxBatchCams];
    const float *v_rgb[kMaxBatchCams];   // GEOM_ONLY: camera c's masked d loss / d rgb [N,3] (the sweep's own output)
};

// GEOM_ONLY (gsa_project_bwd_batch_geometry): the step of a view-parallel RANK.  The sweeps already left every camera's masked
// colour gradient in the 12-byte exchange layout (gsa_composite_bwd_lists_rgb: the ranks' all-gathers travel behind the
// sweeps, and every rank rebuilds the coefficient gradient of ALL views from the gathered slabs: gsa_sh_grad_expand), so this
// launch only sums the geometry gradients of the rank's own cameras -- 44 B in, 44 B out per gaussian plus 124 B per camera
// (gradient record 48, digest 48 + its flags, colour gradient 12, radii 8): the per-view launches it replaces read the
// geometry per view and read-modify-write the sums.  No coefficient accumulators, no LDS rows: half the registers.
template <int DEG, bool GEOM_ONLY = false>
__global__ __launch_bounds__(kBlock, 1) void project_bwd_batch_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ opacities, const float *__restrict__ viewmats, const float *__restrict__ intrins, int n_cams,
    int intrin_stride, int width, int height, BwdCamRows rows_of, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_opacities, float *__restrict__ v_colors, int accumulate) {
    static_assert(DEG >= 1, "view-dependent colour only");
    __shared__ float4 sh_lds[GEOM_ONLY ? 1 : (kBlock / 64) * kWaveLds];   // rows out
    const int wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long wave_base = (long)blockIdx.x * kBlock + wave * 64;
    const long rows = min((long)64, (long)n - wave_base);
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const bool active = i < n;
    const long ic = active ? i : 0;   // (clamped, not predicated: every lane's loads are issued together)

    float g_mean[3], g_quat[4], g_scale[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { g_mean[k] = means[ic * 3 + k]; g_scale[k] = scales[ic * 3 + k]; }
    const float4 q4 = *reinterpret_cast<const float4 *>(quats + ic * 4);
    g_quat[0] = q4.x; g_quat[1] = q4.y; g_quat[2] = q4.z; g_quat[3] = q4.w;
    const float g_opac = opacities[ic];

    // one camera's rows of this gaussian: radii, gradient record, coefficient digest -- loaded one camera AHEAD of the chain
    struct CamRow { int rx, ry; float4 v0, v1, v2, a0, a1, a2; };
    auto load_cam = [&](int ci) -> CamRow {
        CamRow r;
        const int2 rr = *reinterpret_cast<const int2 *>(rows_of.radii[ci] + 2 * ic);
        r.rx = rr.x; r.ry = rr.y;
        const float4 *vrow = reinterpret_cast<const float4 *>(rows_of.v_splat[ci] + ic * GSA_SPLAT_STRIDE);
        r.v0 = vrow[0]; r.v1 = vrow[1]; r.v2 = vrow[2];
        if constexpr (GEOM_ONLY) {   // the colour moments are not in the record: the sweep put them, masked, into v_rgb
            const float *c3 = rows_of.v_rgb[ci] + ic * 3;
            r.v1.z = c3[0]; r.v1.w = c3[1]; r.v2.x = c3[2];
        }
        const float4 *arow = reinterpret_cast<const float4 *>(rows_of.sh_aux[ci]) + ic * 3;
        r.a0 = arow[0]; r.a1 = arow[1]; r.a2 = arow[2];
        return r;
    };
    CamRow cur = load_cam(0);

    float sm[3] = {0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f}, ss[3] = {0.f, 0.f, 0.f}, so = 0.f;
    float vc[GEOM_ONLY ? 1 : NB * 3];
#pragma unroll
    for (int k = 0; k < (GEOM_ONLY ? 1 : NB * 3); ++k) vc[k] = 0.f;

    for (int ci = 0; ci < n_cams; ++ci) {
        CamRow nxt = cur;
        if (ci + 1 < n_cams) nxt = load_cam(ci + 1);   // (uniform)
        if (active && cur.rx > 0 && cur.ry > 0) {
            GsaCamera cam = gsa_load_camera(viewmats + 16 * ci, intrins + (long)intrin_stride * ci);
            float4 v0 = cur.v0, v1 = cur.v1, v2 = cur.v2;
            fold_hot_replicas(rows_of.v_splat[ci], (long)n, v0, v1, v2);
            const float v_rgb[3] = {v1.z, v1.w, v2.x};
            // ---- colour first (as the single-camera kernel): d loss / d rgb after the clamp mask, the view-direction part of
            // the mean gradient from the digest, and this camera's share of the coefficient gradient
            float vdir[3];
            {
                const float dx = g_mean[0] - cam.campos[0], dy = g_mean[1] - cam.campos[1], dz = g_mean[2] - cam.campos[2];
                const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
                const float vr[3] = {cur.a0.w != 0.f ? v_rgb[0] : 0.f, cur.a1.w != 0.f ? v_rgb[1] : 0.f,
                                     cur.a2.w != 0.f ? v_rgb[2] : 0.f};
                if constexpr (!GEOM_ONLY) {
                    float B[NB];
                    gsa_sh_basis<DEG>(dx * inv, dy * inv, dz * inv, B);
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        vc[3 * j] = __builtin_fmaf(B[j], vr[0], vc[3 * j]);
                        vc[3 * j + 1] = __builtin_fmaf(B[j], vr[1], vc[3 * j + 1]);
                        vc[3 * j + 2] = __builtin_fmaf(B[j], vr[2], vc[3 * j + 2]);
                    }
                }
                vdir[0] = cur.a0.x * vr[0] + cur.a1.x * vr[1] + cur.a2.x * vr[2];
                vdir[1] = cur.a0.y * vr[0] + cur.a1.y * vr[1] + cur.a2.y * vr[2];
                vdir[2] = cur.a0.z * vr[0] + cur.a1.z * vr[1] + cur.a2.z * vr[2];
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from interleaving the two chains: register pressure)
            float vm[3], vq[4], vs[3], vo;
            project_chain_bwd(cam, g_mean, g_quat, g_scale, g_opac, width, height, v0, v1, v2, vdir, vm, vq, vs, vo);
            // summed in camera order, as the accumulating single-camera launches sum
#pragma unroll
            for (int k = 0; k < 3; ++k) { sm[k] += vm[k]; ss[k] += vs[k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[k] += vq[k];
            so += vo;
        }
        cur = nxt;
    }

    if (active) {
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { sm[k] += v_means[i * 3 + k]; ss[k] += v_scales[i * 3 + k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[k] += v_quats[i * 4 + k];
            so += v_opacities[i];
        }
        v_means[i * 3] = sm[0]; v_means[i * 3 + 1] = sm[1]; v_means[i * 3 + 2] = sm[2];
        v_quats[i * 4] = sq[0]; v_quats[i * 4 + 1] = sq[1]; v_quats[i * 4 + 2] = sq[2]; v_quats[i * 4 + 3] = sq[3];
        v_scales[i * 3] = ss[0]; v_scales[i * 3 + 1] = ss[1]; v_scales[i * 3 + 2] = ss[2];
        v_opacities[i] = so;
    }

    // ---- the coefficient-gradient rows leave through the wave's LDS image, 32 whole rows a pass (as the single-camera kernel)
    if constexpr (!GEOM_ONLY) {
    float4 *wrows = &sh_lds[wave * kWaveLds];
    float4 *dst = reinterpret_cast<float4 *>(v_colors) + wave_base * kRowQ;
    auto elem = [&](int e) { return e < NB * 3 ? vc[e] : 0.f; };
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    constexpr int NPL = kPassRows * kRowQ / 64;   // 6 float4 per lane and pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p == 1) wave_lds_sync();   // the first 32 rows have left
        if ((lane >> 5) == p) {
#pragma unroll
            for (int q = 0; q < kRowQ; ++q)
                wrows[(lane & 31) * kRowPad + q] = make_float4(elem(4 * q), elem(4 * q + 1), elem(4 * q + 2), elem(4 * q + 3));
        }
        wave_lds_sync();
        const int live = (int)rows - p * kPassRows;   // rows of this pass that exist (wave-uniform)
        float old[4 * NPL];
        if (accumulate && live > 0) {   // (rows clamped, not predicated: all six loads in flight at once)
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int idx = k * 64 + lane, r_ = min(idx / kRowQ, live - 1), c_ = idx % kRowQ;
                const float4 o = dst[(p * kPassRows + r_) * kRowQ + c_];
                old[4 * k] = o.x; old[4 * k + 1] = o.y; old[4 * k + 2] = o.z; old[4 * k + 3] = o.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4 * NPL; ++k) old[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int idx = k * 64 + lane, r_ = idx / kRowQ, c_ = idx - r_ * kRowQ;
            if (r_ < live) {
                float4 v = wrows[r_ * kRowPad + c_];
                v.x += old[4 * k]; v.y += old[4 * k + 1]; v.z += old[4 * k + 2]; v.w += old[4 * k + 3];
                dst[(p * kPassRows) * kRowQ + idx] = v;
            }
        }
    }
    }   // !GEOM_ONLY
}

template <int DEG>
int launch_fwd(bool stage, int n, const float *means, const float *quats, const float *scales,
               const float *opacities, const float *colors, int sh_coeffs, const float *viewmat,
               const float *intrin, int n_cams, int intrin_stride, int width, int height, int cull_mode, float *splat,
               int32_t *radii, int32_t *tiles_touched, uint32_t *depth_keys, uint32_t *tile_rects, float *sh_aux,
               int bin_tile, uint32_t *key_range, uint32_t *qbox, int parts, hipStream_t s, int records_blocks = 0) {
    int tile_w = gsa_div_up(width, GSA_TILE_SIZE), tile_h = gsa_div_up(height, GSA_TILE_SIZE);
    dim3 grid(gsa_div_up(n, kBlock)), block(kBlock);
#define GSA_FWD_LAUNCH(ST, BA, PA)                                                                                     \
    project_fwd_kernel<DEG, ST, BA, PA><<<grid, block, 0, s>>>(n, means, quats, scales, opacities, colors, sh_coeffs, viewmat, \
                                                           intrin, n_cams, intrin_stride, width, height, tile_w, tile_h,     \
                                                           cull_mode, splat, radii, tiles_touched, depth_keys, tile_rects,   \
                                                           sh_aux, bin_tile, key_range, qbox)
    if (parts == kPartRecords) {
        if (records_blocks > 0 && (int)grid.x > records_blocks) grid.x = (unsigned)records_blocks;
        if (n_cams > 1) {
            if (stage) GSA_FWD_LAUNCH(true, true, kPartRecords); else GSA_FWD_LAUNCH(false, true, kPartRecords);
        } else {
            if (stage) GSA_FWD_LAUNCH(true, false, kPartRecords); else GSA_FWD_LAUNCH(false, false, kPartRecords);
        }
    } else if (n_cams > 1) {
        if (stage) GSA_FWD_LAUNCH(true, true, kPartBoth); else GSA_FWD_LAUNCH(false, true, kPartBoth);
    } else {
        if (stage) GSA_FWD_LAUNCH(true, false, kPartBoth); else GSA_FWD_LAUNCH(false, false, kPartBoth);
    }
#undef GSA_FWD_LAUNCH
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

template <int DEG>
int launch_bwd(bool stage, int n, const float *means, const float *quats, const float *scales, const float *opacities,
               const float *colors, int sh_coeffs, const float *viewmat, const float *intrin, int width,
               int height, const int32_t *radii, const float *v_splat, float *v_means,
               float *v_quats, float *v_scales, float *v_opacities, float *v_colors, int accumulate, float *v_rgb_out,
               const float *sh_aux, hipStream_t s) {
    dim3 grid(gsa_div_up(n, kBlock)), block(kBlock);
#define GSA_BWD_LAUNCH(ST, FA, AU)                                                                                    \
    project_bwd_kernel<DEG, ST, FA, AU><<<grid, block, 0, s>>>(n, means, quats, scales, opacities, colors, sh_coeffs,    \
                                                               viewmat, intrin, width, height, radii, v_splat, v_means, \
                                                               v_quats, v_scales, v_opacities, v_colors, accumulate,    \
                                                               v_rgb_out, sh_aux)
    if constexpr (DEG >= 1) {
        if (sh_aux) {   // the forward's 48-B digest of the coefficient row replaces the row
            if (v_rgb_out) GSA_BWD_LAUNCH(false, true, true);
            else if (stage) GSA_BWD_LAUNCH(true, false, true);
            else GSA_BWD_LAUNCH(false, false, true);
            GSA_LAUNCH_CHECK();
            return GSA_OK;
        }
    }
    if constexpr (DEG >= 0) {
        if (v_rgb_out) {
            if (stage) GSA_BWD_LAUNCH(true, true, false); else GSA_BWD_LAUNCH(false, true, false);
            GSA_LAUNCH_CHECK();
            return GSA_OK;
        }
    }
    if (stage) GSA_BWD_LAUNCH(true, false, false); else GSA_BWD_LAUNCH(false, false, false);
#undef GSA_BWD_LAUNCH
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

bool can_stage(const float *colors, int sh_degree, int sh_coeffs) {
    return sh_degree >= 0 && sh_coeffs == 16 && (reinterpret_cast<uintptr_t>(colors) & 15) == 0;
}

}  // namespace

static int project_fwd_impl(int n, const float *means, const float *quats, const float *scales,
                            const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                            const float *viewmat, const float *intrin, int n_cams, int intrin_stride, int width, int height,
                            int cull_mode, float *splat, int32_t *radii, int32_t *tiles_touched, uint32_t *depth_keys,
                            uint32_t *tile_rects, float *sh_aux, int bin_tile, uint32_t *key_range, uint32_t *qbox,
                            gsa_stream_t stream, int parts = kPartBoth) {
    GSA_RETURN_IF(n < 0 || width <= 0 || height <= 0 || n_cams < 1 || (intrin_stride != 0 && intrin_stride != 9),
                  GSA_ERR_INVALID_ARGUMENT);
    const bool no_stage = (parts & 4) != 0;   // (developer A/B: the records part without the LDS row staging)
    const int records_blocks = parts >> 8;     // (bits 8..: workgroups of a low-occupancy records launch; 0 = one per block of gaussians)
    parts &= 3;
    GSA_RETURN_IF(parts != kPartGeometry && parts != kPartRecords && parts != kPartBoth, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_degree < -1 || sh_degree > 3, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_degree >= 0 && sh_coeffs < (sh_degree + 1) * (sh_degree + 1), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(cull_mode != 0 && cull_mode != 1, GSA_ERR_INVALID_ARGUMENT);
    if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
    GSA_RETURN_IF((bin_tile != GSA_TILE_SIZE && bin_tile != 2 * GSA_TILE_SIZE && bin_tile != 4 * GSA_TILE_SIZE) ||
                      (bin_tile != GSA_TILE_SIZE && cull_mode != 0),
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(tile_rects) & 7) != 0 || (reinterpret_cast<uintptr_t>(radii) & 7) != 0 ||
                      (reinterpret_cast<uintptr_t>(splat) & 15) != 0 || (reinterpret_cast<uintptr_t>(sh_aux) & 15) != 0,
                  GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!means || !quats || !scales || !opacities || !viewmat || !intrin, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((parts & kPartRecords) && (!colors || !splat), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((parts & kPartGeometry) && (!radii || !tiles_touched || !depth_keys), GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (parts == kPartGeometry) {   // no colour, no rows: one light instance per camera mode
        int tile_w = gsa_div_up(width, GSA_TILE_SIZE), tile_h = gsa_div_up(height, GSA_TILE_SIZE);
        dim3 grid(gsa_div_up(n, kBlock)), block(kBlock);
        if (n_cams > 1)
            project_fwd_kernel<-1, false, true, kPartGeometry><<<grid, block, 0, s>>>(
                n, means, quats, scales, opacities, nullptr, 0, viewmat, intrin, n_cams, intrin_stride, width, height, tile_w,
                tile_h, cull_mode, nullptr, radii, tiles_touched, depth_keys, tile_rects, nullptr, bin_tile, key_range, qbox);
        else
            project_fwd_kernel<-1, false, false, kPartGeometry><<<grid, block, 0, s>>>(
                n, means, quats, scales, opacities, nullptr, 0, viewmat, intrin, n_cams, intrin_stride, width, height, tile_w,
                tile_h, cull_mode, nullptr, radii, tiles_touched, depth_keys, tile_rects, nullptr, bin_tile, key_range, qbox);
        GSA_LAUNCH_CHECK();
        return GSA_OK;
    }
    // degree 0 needs 12 B of each 192-B row: direct loads beat staging whole rows
    bool stage = can_stage(colors, sh_degree, sh_coeffs) && sh_degree >= 1 && !no_stage;
#define GSA_FWD(D)                                                                                                      \
    return launch_fwd<D>(stage, n, means, quats, scales, opacities, colors, sh_coeffs, viewmat, intrin, n_cams,         \
                         intrin_stride, width, height, cull_mode, splat, radii, tiles_touched, depth_keys, tile_rects, \
                         sh_aux, bin_tile, key_range, qbox, parts, s, records_blocks)
    switch (sh_degree) {
        case -1: GSA_FWD(-1);
        case 0: GSA_FWD(0);
        case 1: GSA_FWD(1);
        case 2: GSA_FWD(2);
        default: GSA_FWD(3);
    }
#undef GSA_FWD
}

extern "C" int gsa_project_fwd(int n, const float *means, const float *quats, const float *scales,
                               const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                               const float *viewmat, const float *intrin, int width, int height, int cull_mode,
                               float *splat, int32_t *radii, int32_t *tiles_touched, uint32_t *depth_keys,
                               uint32_t *tile_rects, float *sh_aux, int bin_tile, uint32_t *key_range, uint32_t *qbox,
                               gsa_stream_t stream) {
    return project_fwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmat, intrin, 1, 0, width,
                            height, cull_mode, splat, radii, tiles_touched, depth_keys, tile_rects, sh_aux, bin_tile,
                            key_range, qbox, stream);
}

extern "C" int gsa_project_fwd_batch(int n, const float *means, const float *quats, const float *scales,
                                     const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                                     int n_cams, const float *viewmats, const float *intrins, int intrin_stride, int width,
                                     int height, int cull_mode, float *splat, int32_t *radii, int32_t *tiles_touched,
                                     uint32_t *depth_keys, uint32_t *tile_rects, float *sh_aux, int bin_tile,
                                     uint32_t *key_range, uint32_t *qbox, gsa_stream_t stream) {
    // (camera c's slabs start c * n rows into the [n_cams, n, ...] arrays: 64 / 48 / 16 / 8 / 4 bytes a row, so every slab keeps
    // the alignment its vector stores need whatever n is)
    return project_fwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmats, intrins, n_cams,
                            intrin_stride, width, height, cull_mode, splat, radii, tiles_touched, depth_keys, tile_rects,
                            sh_aux, bin_tile, key_range, qbox, stream);
}

extern "C" int gsa_project_fwd_parts(int parts, int n, const float *means, const float *quats, const float *scales,
                                     const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                                     int n_cams, const float *viewmats, const float *intrins, int intrin_stride, int width,
                                     int height, int cull_mode, float *splat, int32_t *radii, int32_t *tiles_touched,
                                     uint32_t *depth_keys, uint32_t *tile_rects, float *sh_aux, int bin_tile,
                                     uint32_t *key_range, uint32_t *qbox, gsa_stream_t stream) {
    return project_fwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmats, intrins, n_cams,
                            intrin_stride, width, height, cull_mode, splat, radii, tiles_touched, depth_keys, tile_rects,
                            sh_aux, bin_tile, key_range, qbox, stream, parts);
}

static int project_bwd_impl(int n, const float *means, const float *quats, const float *scales, const float *opacities,
                            const float *colors, int sh_degree, int sh_coeffs, const float *viewmat, const float *intrin,
                            int width, int height, const int32_t *radii, const float *v_splat, float *v_means,
                            float *v_quats, float *v_scales, float *v_opacities, float *v_colors, int accumulate,
                            float *v_rgb_out, const float *sh_aux, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || width <= 0 || height <= 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_degree < -1 || sh_degree > 3, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_degree >= 0 && sh_coeffs < (sh_degree + 1) * (sh_degree + 1), GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!means || !quats || !scales || !opacities || !colors || !viewmat || !intrin || !radii || !v_splat ||
                      !v_means || !v_quats || !v_scales || !v_opacities || (!v_colors && !v_rgb_out),
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(v_rgb_out && sh_degree < 0, GSA_ERR_INVALID_ARGUMENT);   // factored form: SH colours only
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(sh_aux) & 15) != 0, GSA_ERR_INVALID_ARGUMENT);
    if (sh_degree < 1) sh_aux = nullptr;   // no view dependence: nothing the forward could have left
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool stage = can_stage(colors, sh_degree, sh_coeffs) &&
                 (v_rgb_out || (reinterpret_cast<uintptr_t>(v_colors) & 15) == 0);
#define GSA_BWD(D)                                                                                                  \
    return launch_bwd<D>(stage, n, means, quats, scales, opacities, colors, sh_coeffs, viewmat, intrin, width, height, \
                         radii, v_splat, v_means, v_quats, v_scales, v_opacities, v_colors, accumulate ? 1 : 0, v_rgb_out, \
                         sh_aux, s)
    switch (sh_degree) {
        case -1: GSA_BWD(-1);
        case 0: GSA_BWD(0);
        case 1: GSA_BWD(1);
        case 2: GSA_BWD(2);
        default: GSA_BWD(3);
    }
#undef GSA_BWD
}

extern "C" int gsa_project_bwd(int n, const float *means, const float *quats, const float *scales,
                               const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                               const float *viewmat, const float *intrin, int width, int height,
                               const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                               float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                               const float *sh_aux, gsa_stream_t stream) {
    (void)splat;   // kept in the signature: the backward no longer reads the forward's records (see the kernel)
    return project_bwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmat, intrin, width, height,
                            radii, v_splat, v_means, v_quats, v_scales, v_opacities, v_colors, 0, nullptr, sh_aux, stream);
}

extern "C" int gsa_project_bwd_acc(int n, const float *means, const float *quats, const float *scales,
                                   const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                                   const float *viewmat, const float *intrin, int width, int height,
                                   const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                                   float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                                   int accumulate, const float *sh_aux, gsa_stream_t stream) {
    (void)splat;   // kept in the signature: the backward no longer reads the forward's records (see the kernel)
    return project_bwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmat, intrin, width, height,
                            radii, v_splat, v_means, v_quats, v_scales, v_opacities, v_colors, accumulate, nullptr, sh_aux,
                            stream);
}

// ------------------------------------------------------------------------------------------------
// factored SH gradient: project_bwd writes d loss / d rgb [N,3]; the [N,K,3] coefficient gradient is rebuilt from it
// ------------------------------------------------------------------------------------------------
namespace {

// v_colors[g][k][c] (+)= sum over views v of  basis_k(normalise(mean_g - campos_v)) * v_rgb[v][g][c]
// One thread per gaussian; rows above the active degree are written as zeros (or left alone when accumulating).
// STAGED (rows of at most 48 floats): the rows leave through LDS, a wave's 64 rows as one contiguous range of
// consecutive dwords per store instruction, instead of 64 lanes each walking their own 192-B row.
constexpr int kExpHalf = 24, kExpStride = 25;   // LDS image: 24 columns a pass, padded rows (odd stride: lane-per-row
                                                // writes and dword-linear reads are both conflict-free)
template <int DEG, bool STAGED>
__global__ __launch_bounds__(256) void sh_grad_expand_kernel(int n, const float *__restrict__ means, int sh_coeffs,
                                                             int n_views, const float *__restrict__ viewmats,
                                                             const float *__restrict__ intrin,
                                                             const float *__restrict__ v_rgb, float *__restrict__ v_colors,
                                                             int accumulate, float *__restrict__ v_rest) {
    // v_rest (optional): split output -- v_colors is then the [N,1,3] gradient of the DC coefficient and v_rest the
    // [N, sh_coeffs - 1, 3] gradient of the others (the reference model keeps them as two parameters)
    constexpr int NB = (DEG + 1) * (DEG + 1);
    __shared__ float lds[STAGED ? 256 * kExpStride : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (!STAGED && i >= n) return;
    float acc[NB * 3];
#pragma unroll
    for (int k = 0; k < NB * 3; ++k) acc[k] = 0.f;
    if (i < n) {
        const float mx = means[i * 3], my = means[i * 3 + 1], mz = means[i * 3 + 2];
        for (int v = 0; v < n_views; ++v) {
            const float *vr = v_rgb + ((long)v * n + i) * 3;
            const float r = vr[0], g = vr[1], b = vr[2];
            const bool on = r != 0.f || g != 0.f || b != 0.f;        // off: culled in that view (or clamped away)
            if (!on) continue;
            const GsaCamera cam = gsa_load_camera(viewmats + 16 * v, intrin);
            const float dx = mx - cam.campos[0], dy = my - cam.campos[1], dz = mz - cam.campos[2];
            const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            float B[NB];
            gsa_sh_basis<DEG>(dx * inv, dy * inv, dz * inv, B);
#pragma unroll
            for (int k = 0; k < NB; ++k) { acc[3 * k] += B[k] * r; acc[3 * k + 1] += B[k] * g; acc[3 * k + 2] += B[k] * b; }
        }
    }
    if constexpr (STAGED) {
        const int wd = sh_coeffs * 3;                       // <= 48 (checked by the launcher)
        const long wave_base = (long)blockIdx.x * 256 + wave * 64;
        if (wave_base >= n) return;
        const int rows = (int)min((long)64, (long)n - wave_base);
        float *wl = lds + wave * 64 * kExpStride;
        // element e of a store range -> (row, column) of the LDS image, advanced incrementally; the destination is
        // rows of dst_stride floats of which [dst_off, dst_off + width) come from LDS columns [col0, col0 + width)
        auto sweep = [&](float *__restrict__ dst, int dst_stride, int dst_off, int col0, int width, bool add) {
            if (width <= 0) return;
            int row = lane / width, col = lane - row * width;
            const int total = rows * width;
            for (int e = lane; e < total; e += 64) {
                const float val = wl[row * kExpStride + col0 + col];
                float *d = dst + (long)row * dst_stride + dst_off + col;
                *d = add ? *d + val : val;
                col += 64;
                while (col >= width) { col -= width; ++row; }
            }
        };
        // two halves of 24 columns: 6 KB of LDS per wave instead of 12, so twice the waves stay resident
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lo = kExpHalf * h, hi = min(wd, lo + kExpHalf);
            if (lo >= hi) break;
            if (h) wave_lds_sync();   // the first half has been read
#pragma unroll
            for (int k = 0; k < kExpHalf; ++k) {
                const int g = kExpHalf * h + k;
                if (g < hi) wl[lane * kExpStride + k] = g < NB * 3 ? acc[g < NB * 3 ? g : 0] : 0.f;
            }
            wave_lds_sync();
            if (v_rest) {
                if (h == 0) {
                    sweep(v_colors + wave_base * 3, 3, 0, 0, 3, false);
                    sweep(v_rest + wave_base * (long)(wd - 3), wd - 3, 0, 3, hi - 3, false);
                } else {
                    sweep(v_rest + wave_base * (long)(wd - 3), wd - 3, lo - 3, 0, hi - lo, false);
                }
            } else {
                sweep(v_colors + wave_base * (long)wd, wd, lo, 0, hi - lo, accumulate != 0);
            }
        }
        return;
    }
    if (v_rest) {
        float *dc = v_colors + i * 3, *rest = v_rest + i * (long)(sh_coeffs - 1) * 3;
        dc[0] = acc[0]; dc[1] = acc[1]; dc[2] = acc[2];
#pragma unroll
        for (int k = 3; k < NB * 3; ++k) rest[k - 3] = acc[k];
        for (int k = NB * 3; k < sh_coeffs * 3; ++k) rest[k - 3] = 0.f;
        return;
    }
    float *row = v_colors + i * (long)sh_coeffs * 3;
    if (accumulate) {
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) row[k] += acc[k];
    } else {
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) row[k] = acc[k];
        for (int k = NB * 3; k < sh_coeffs * 3; ++k) row[k] = 0.f;
    }
}

}  // namespace

extern "C" int gsa_project_bwd_factored(int n, const float *means, const float *quats, const float *scales,
                                        const float *opacities, const float *colors, int sh_degree, int sh_coeffs,
                                        const float *viewmat, const float *intrin, int width, int height,
                                        const float *splat, const int32_t *radii, const float *v_splat, float *v_means,
                                        float *v_quats, float *v_scales, float *v_opacities, float *v_rgb,
                                        int accumulate, const float *sh_aux, gsa_stream_t stream) {
    (void)splat;
    GSA_RETURN_IF(!v_rgb, GSA_ERR_INVALID_ARGUMENT);
    return project_bwd_impl(n, means, quats, scales, opacities, colors, sh_degree, sh_coeffs, viewmat, intrin, width, height,
                            radii, v_splat, v_means, v_quats, v_scales, v_opacities, nullptr, accumulate, v_rgb, sh_aux,
                            stream);
}

static int sh_grad_expand_impl(int n, const float *means, int sh_degree, int sh_coeffs, int n_views,
                               const float *viewmats, const float *intrin, const float *v_rgb, float *v_colors,
                               int accumulate, float *v_rest, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_coeffs < (sh_degree + 1) * (sh_degree + 1), GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!means || !viewmats || !intrin || !v_rgb || !v_colors, GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = gsa_div_up(n, 256);
    // staging pays while the 192-B rows dominate the traffic; measured: 59 -> 47 us for 1 view, 63 -> 52 us for 4,
    // 87 -> 97 us for 8 (there the kernel streams 8 x 12 MB of factors and the staging only adds to it)
    const bool staged = sh_coeffs * 3 <= 2 * kExpHalf && (!v_rest || sh_coeffs >= 2) && n_views <= 4;
#define GSA_EXP(D)                                                                                                    \
    if (staged)                                                                                                       \
        sh_grad_expand_kernel<D, true><<<grid, 256, 0, s>>>(n, means, sh_coeffs, n_views, viewmats, intrin, v_rgb,    \
                                                            v_colors, accumulate ? 1 : 0, v_rest);                    \
    else                                                                                                              \
        sh_grad_expand_kernel<D, false><<<grid, 256, 0, s>>>(n, means, sh_coeffs, n_views, viewmats, intrin, v_rgb,   \
                                                             v_colors, accumulate ? 1 : 0, v_rest);                   \
    break
    switch (sh_degree) {
        case 0: GSA_EXP(0);
        case 1: GSA_EXP(1);
        case 2: GSA_EXP(2);
        default: GSA_EXP(3);
    }
#undef GSA_EXP
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_sh_grad_expand(int n, const float *means, int sh_degree, int sh_coeffs, int n_views,
                                  const float *viewmats, const float *intrin, const float *v_rgb, float *v_colors,
                                  int accumulate, gsa_stream_t stream) {
    return sh_grad_expand_impl(n, means, sh_degree, sh_coeffs, n_views, viewmats, intrin, v_rgb, v_colors, accumulate,
                               nullptr, stream);
}

extern "C" int gsa_sh_grad_expand_split(int n, const float *means, int sh_degree, int sh_coeffs, int n_views,
                                        const float *viewmats, const float *intrin, const float *v_rgb, float *v_dc,
                                        float *v_rest, gsa_stream_t stream) {
    GSA_RETURN_IF(n > 0 && (!v_rest || sh_coeffs < 2), GSA_ERR_INVALID_ARGUMENT);
    return sh_grad_expand_impl(n, means, sh_degree, sh_coeffs, n_views, viewmats, intrin, v_rgb, v_dc, 0, v_rest, stream);
}

static int project_bwd_batch_impl(int n, const float *means, const float *quats, const float *scales,
                                  const float *opacities, int sh_degree, int sh_coeffs, int n_cams, const float *viewmats,
                                  const float *intrins, int intrin_stride, int width, int height,
                                  const int32_t *const *radii, const float *const *v_splat, const float *const *sh_aux,
                                  const float *const *v_rgb, float *v_means, float *v_quats, float *v_scales,
                                  float *v_opacities, float *v_colors, int accumulate, gsa_stream_t stream) {
    const bool geom = v_rgb != nullptr;
    GSA_RETURN_IF(n < 0 || width <= 0 || height <= 0 || n_cams < 1 || (intrin_stride != 0 && intrin_stride != 9),
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(sh_degree < 1 || sh_degree > 3 || (!geom && sh_coeffs != 16) || n_cams > kMaxBatchCams, GSA_ERR_UNSUPPORTED);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!means || !quats || !scales || !opacities || !viewmats || !intrins || !radii || !v_splat || !sh_aux ||
                      !v_means || !v_quats || !v_scales || !v_opacities || (!geom && !v_colors),
                  GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(v_colors) & 15) != 0 || (reinterpret_cast<uintptr_t>(quats) & 15) != 0,
                  GSA_ERR_INVALID_ARGUMENT);
    BwdCamRows rows{};
    for (int c = 0; c < kMaxBatchCams; ++c) {   // (unused slots repeat the last camera: never read)
        const int k = c < n_cams ? c : n_cams - 1;
        GSA_RETURN_IF(!radii[k] || !v_splat[k] || !sh_aux[k] || (geom && !v_rgb[k]), GSA_ERR_INVALID_ARGUMENT);
        GSA_RETURN_IF(((reinterpret_cast<uintptr_t>(v_splat[k]) | reinterpret_cast<uintptr_t>(sh_aux[k])) & 15) != 0 ||
                          (reinterpret_cast<uintptr_t>(radii[k]) & 7) != 0,
                      GSA_ERR_INVALID_ARGUMENT);
        rows.radii[c] = radii[k]; rows.v_splat[c] = v_splat[k]; rows.sh_aux[c] = sh_aux[k];
        rows.v_rgb[c] = geom ? v_rgb[k] : nullptr;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(gsa_div_up(n, kBlock)), block(kBlock);
#define GSA_BWDB(D, G)                                                                                                     \
    project_bwd_batch_kernel<D, G><<<grid, block, 0, s>>>(n, means, quats, scales, opacities, viewmats, intrins, n_cams,   \
                                                          intrin_stride, width, height, rows, v_means, v_quats, v_scales, \
                                                          v_opacities, v_colors, accumulate ? 1 : 0)
    if (geom) {
        switch (sh_degree) {
            case 1: GSA_BWDB(1, true); break;
            case 2: GSA_BWDB(2, true); break;
            default: GSA_BWDB(3, true); break;
        }
    } else {
        switch (sh_degree) {
            case 1: GSA_BWDB(1, false); break;
            case 2: GSA_BWDB(2, false); break;
            default: GSA_BWDB(3, false); break;
        }
    }
#undef GSA_BWDB
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_project_bwd_batch(int n, const float *means, const float *quats, const float *scales,
                                     const float *opacities, int sh_degree, int sh_coeffs, int n_cams, const float *viewmats,
                                     const float *intrins, int intrin_stride, int width, int height,
                                     const int32_t *const *radii, const float *const *v_splat, const float *const *sh_aux,
                                     float *v_means, float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                                     int accumulate, gsa_stream_t stream) {
    GSA_RETURN_IF(!v_colors && n > 0, GSA_ERR_INVALID_ARGUMENT);
    return project_bwd_batch_impl(n, means, quats, scales, opacities, sh_degree, sh_coeffs, n_cams, viewmats, intrins,
                                  intrin_stride, width, height, radii, v_splat, sh_aux, nullptr, v_means, v_quats, v_scales,
                                  v_opacities, v_colors, accumulate, stream);
}

extern "C" int gsa_project_bwd_batch_geometry(int n, const float *means, const float *quats, const float *scales,
                                              const float *opacities, int sh_degree, int n_cams, const float *viewmats,
                                              const float *intrins, int intrin_stride, int width, int height,
                                              const int32_t *const *radii, const float *const *v_splat,
                                              const float *const *sh_aux, const float *const *v_rgb, float *v_means,
                                              float *v_quats, float *v_scales, float *v_opacities, int accumulate,
                                              gsa_stream_t stream) {
    GSA_RETURN_IF(!v_rgb, GSA_ERR_INVALID_ARGUMENT);
    return project_bwd_batch_impl(n, means, quats, scales, opacities, sh_degree, 16, n_cams, viewmats, intrins, intrin_stride,
                                  width, height, radii, v_splat, sh_aux, v_rgb, v_means, v_quats, v_scales, v_opacities, nullptr,
                                  accumulate, stream);
}
