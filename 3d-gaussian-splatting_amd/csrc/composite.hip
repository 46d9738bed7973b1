// composite.hip -- stage 3 (front-to-back alpha compositing) and stage 4 (reverse gradient sweep).
//
// Mapping (gfx950, wave64): ONE INDEPENDENT WAVE PER 8x8 QUADRANT of a 16x16 tile, one lane per pixel
// (lane l -> (x = l & 7, y = l >> 3) inside the quadrant), launched as one-wave workgroups: no workgroup
// barrier anywhere, so while one wave waits for its gathers the others on the SIMD blend (a tile-wide
// workgroup with barriers spent half its residency staging or waiting for its slowest quadrant:
// tools/trace_tiles.py).  The four quadrants of a tile are dealt to the same XCD (they gather the same records).
//
//  * Staging: the wave gathers 64 sorted splat records at a time (one 64-B record per lane, one cache line,
//    served from L2 / Infinity Cache) into its own 48-B-per-record LDS image, and each lane tests whether the
//    alpha >= 1/255 ellipse of ITS record can reach a pixel centre of the quadrant (ellipse-vs-rectangle,
//    conservative margin).  The gathers are software-pipelined two batches deep in registers (gaussian ids of
//    batch j+2 and records of batch j+1 are in flight while batch j is blended).
//  * Compaction: ballot + mbcnt turn the 64 test bits into a compacted LIST of LDS record offsets.  A record
//    that cannot touch the quadrant costs the wave nothing; about 45 % of (quadrant, record) pairs survive.
//    Results are identical to the plain per-pixel loop: a skipped pair fails the alpha test at every pixel
//    of the quadrant anyway.
//  * Walk: one 8-B LDS broadcast yields four list entries; each record is three LDS broadcasts.  The blend body
//    has no scalar mask logic and, in the common case, no per-record select at all (optimistic group of four;
//    see walk_fwd).  A wave stops as soon as its 64 pixels are finished.
//  * Backward: records are taken FOUR at a time; the 4 x 9 (10) per-pixel moment contributions are summed over
//    the wave by one transpose-reduce (DPP adds with bank masks for the two widest levels, then
//    v_permlane32_swap / v_permlane16_swap, then two select folds: ~18 VALU per record instead of ~55 for
//    per-value butterflies) and added with ONE atomic instruction whose active lanes cover the four 64-B
//    gradient records.  The sweep accumulates raw moments of u = dL/dsigma;
//    project_bwd turns them into mean / conic / opacity gradients once per gaussian.
//
// Instruction costs that shaped the bodies (tools/inst_microbench.hip, cycles per wave64 instruction per SIMD):
// v_fma/v_mul/v_add/v_and ~2.5, v_cmp/v_min/v_cndmask(sgpr)/DPP/SGPR-operand forms ~4.3, v_exp/v_rcp ~8,
// v_permlane*_swap ~8.8, packed v_pk_* ~4.9 (no gain over two scalar ops; the library is built without SLP).
//
// Restates SURVEY.md Appendix A.5 (forward) and A.6 (backward) -- gsplat's rasterize_to_pixels behind
// /root/reference/src/backends/gsplat_backend.py:168-182.

#include "project_math.h"
#include "tile_schedule.h"
#include <type_traits>

namespace {

constexpr int kTile = GSA_TILE_SIZE;
constexpr int kWave = 64;     // threads per workgroup = one wave = one quadrant
constexpr int kBatch = 64;    // records staged per wave and step (one per lane)
constexpr float kLog2e = 1.4426950408889634f;


#ifdef GSA_TRACE   // developer build: per-wave timeline and phase times (tools/trace_tiles.py)
__device__ unsigned long long *g_trace = nullptr;
struct TraceState {
    unsigned long long t0, mark, acc[8];
    int batches, visits;
};
__device__ __forceinline__ void trace_begin(TraceState &ts) {
    ts.t0 = ts.mark = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < 8; ++k) ts.acc[k] = 0;
    ts.batches = ts.visits = 0;
}
__device__ __forceinline__ void trace_mark(TraceState &ts, int k) {
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    ts.acc[k] += now - ts.mark;
    ts.mark = now;
}
__device__ __forceinline__ void trace_end(const TraceState &ts, int kernel, int unit, int work) {
    if (g_trace && threadIdx.x == 0) {
        unsigned long long *r = g_trace + ((size_t)kernel * 262144 + unit) * 16;
        r[0] = ts.t0;
        r[1] = __builtin_amdgcn_s_memrealtime();
        r[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) |
               __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID[3:0], HW_ID[31:0]
        r[3] = work;
        for (int k = 0; k < 8; ++k) r[4 + k] = ts.acc[k];
        r[12] = ts.batches; r[13] = ts.visits;
    }
}
#define GSA_TRACE_BEGIN() TraceState trace_ts; trace_begin(trace_ts)
#define GSA_TRACE_MARK(k) trace_mark(trace_ts, k)
#define GSA_TRACE_COUNT(n) do { ++trace_ts.batches; trace_ts.visits += (n); } while (0)
#define GSA_TRACE_END(kernel, unit, work) trace_end(trace_ts, kernel, unit, work)
#else
#define GSA_TRACE_BEGIN()
#define GSA_TRACE_MARK(k)
#define GSA_TRACE_COUNT(n)
#define GSA_TRACE_END(kernel, unit, work)
#endif

// Workgroup (= wave) -> (tile, quadrant).  Workgroups are dealt round-robin over the 8 XCDs (speed only, never
// correctness): every XCD gets one contiguous band of tiles, and the four quadrants of a tile follow each other
// on the same XCD so that their gathers of the same records meet in one L2.
// Injective from [0, 32 * chunk) onto (tile, quadrant) pairs; callers drop tiles >= n_tiles.
// `tile_order` (optional, gsa_tile_schedule): a permutation of each band that puts the tiles with long record
// lists first, so that the launch ends on its short jobs (a wave runs for ~1/5 of the whole kernel on a full tile;
// in index order the last round of waves left the chip a fifth empty).  Entries of -1 pad the last band.
// The record list a (tile, quadrant) wave walks.  bin_shift 0: the tile's own list, tile_offsets[tile .. tile + 1], four
// quadrant slices per list.  bin_shift 1: the records were binned into 32 x 32 BINS (half the records to sort; a wave
// re-tests every record against its 8 x 8 quadrant anyway): the four tiles of a bin share its list, tile_offsets is
// indexed by bin, and the bin has 16 quadrant slices (4 * sub-tile + quadrant).
__device__ __forceinline__ void tile_list(const int32_t *__restrict__ tile_offsets, int tile, int tile_w, int bin_shift,
                                          int q, int &start, int &end, int &slice) {
    int idx = tile;
    slice = q;
    if (bin_shift) {
        const int tx = tile % tile_w, ty = tile / tile_w, m = (1 << bin_shift) - 1;
        const int bin_w = (tile_w + m) >> bin_shift;
        idx = (ty >> bin_shift) * bin_w + (tx >> bin_shift);
        slice = ((((ty & m) << bin_shift) | (tx & m)) << 2) | q;
    }
    start = __builtin_amdgcn_readfirstlane(tile_offsets[idx]);
    end = __builtin_amdgcn_readfirstlane(tile_offsets[idx + 1]);
}

__device__ __forceinline__ void wave_unit(int b, int n_tiles, const int32_t *__restrict__ tile_order, int &tile, int &q) {
    const int chunk = (n_tiles + 7) >> 3;
    const int within = b >> 3;
    q = within & 3;
    const int slot = (b & 7) * chunk + (within >> 2);
    tile = tile_order ? tile_order[slot] : slot;
    if (tile < 0) tile = n_tiles;
}

// LDS image of the wave's staged batch, F4 float4 per record (3 for RGB, 4 with the depth channel):
//   [0] mean.x  mean.y  log2e*a/2  log2e*b        (conic pre-scaled for exp2: s = log2e * sigma)
//   [1] log2e*c/2  opacity  r  g
//   [2] b  smax = log2e*ln(255 o)  (int) position in the sorted list  (uint) gaussian id
//   [3] depth  -  -  -
// Slot kBatch is a NULL record (opacity 0: blends nothing) that pads the list to a multiple of four; its list
// position is set to that of the last real entry.  list = byte offsets of the records that can touch the
// quadrant, in walk order.
template <int F4>
struct Staged {
    static constexpr bool kDense = false;
    static constexpr int kRecBytes = 16 * F4;
    float4 rec[(kBatch + 1) * F4];
    uint16_t list[kBatch + 8];
};

// One lane's share of a batch in flight: the 64-B splat record of one sorted list entry.
struct Row {
    float4 r0, r1, r2;
    uint32_t gid;
};

__device__ __forceinline__ Row load_row(const float *__restrict__ splat, uint32_t gid, bool valid) {
    Row r;
    r.gid = gid;
    if (valid) {
        const float4 *row = reinterpret_cast<const float4 *>(splat + (size_t)gid * GSA_SPLAT_STRIDE);
        r.r0 = row[0]; r.r1 = row[1]; r.r2 = row[2];
    } else {
        r.r0 = r.r1 = r.r2 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return r;
}

// Can the alpha >= 1/255 region of a splat reach a pixel CENTRE in the rectangle of centres [x0, x1] x [y0, y1] (the
// 8x8 quadrant, or the bounding box of its pixels that are still accumulating)?  Smallest sigma over that rectangle (convex quadratic: inside -> 0, else the best point of
// the four edges) against ln(255 o) with a conservative margin.  The edge minimiser uses v_rcp: an error there
// only moves the evaluated point ALONG the edge, which raises the value by a second-order amount the margin
// covers.  Both kernels call this one function, so they always build the same lists.
__device__ __forceinline__ bool quadrant_reach(float mx, float my, float a, float b, float c, float sig_max,
                                               float x0, float y0, float x1, float y1) {
    const float thr = sig_max * 1.0001f + 1e-4f;
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return true;
    const float nb_c = -b * __builtin_amdgcn_rcpf(c), nb_a = -b * __builtin_amdgcn_rcpf(a);
    const float dlo = my - y1, dhi = my - y0, elo = mx - x1, ehi = mx - x0;
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dx = e == 0 ? ehi : elo;                       // vertical edge x = x0 / x1
        const float dy = fminf(dhi, fmaxf(dlo, nb_c * dx));
        best = fminf(best, 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy);
        const float ey = e == 0 ? dhi : dlo;                       // horizontal edge y = y0 / y1
        const float ex = fminf(ehi, fmaxf(elo, nb_a * ey));
        best = fminf(best, 0.5f * (a * ex * ex + c * ey * ey) + b * ex * ey);
    }
    return best <= thr;
}

// Lane `lane` writes its record (list position idx) into slot `lane` of the wave's LDS image and tests it against
// the quadrant; the wave compacts the hits into st.list.  Returns the number of listed records (wave-uniform);
// `any_cap` tells whether a listed record's opacity exceeds the 0.999 alpha cap.
template <int F4>
__device__ __forceinline__ int stage_batch(Staged<F4> &st, int lane, const Row &row, bool valid, int idx, float qx0,
                                           float qy0, float qx1, float qy1, bool &any_cap) {
    const float4 r0 = row.r0, r1 = row.r1, r2 = row.r2;
    st.rec[F4 * lane + 0] = make_float4(r0.x, r0.y, 0.5f * kLog2e * r0.z, kLog2e * r0.w);
    st.rec[F4 * lane + 1] = make_float4(0.5f * kLog2e * r1.x, r1.y, r1.z, r1.w);
    st.rec[F4 * lane + 2] = make_float4(r2.x, fmaxf(0.f, kLog2e * r2.z), __int_as_float(idx), __uint_as_float(row.gid));
    if (F4 == 4) st.rec[F4 * lane + 3] = make_float4(r2.y, 0.f, 0.f, 0.f);
    const bool hit = valid && quadrant_reach(r0.x, r0.y, r0.z, r0.w, r1.x, r2.z, qx0, qy0, qx1, qy1);
    const unsigned long long b = __ballot(hit);
    any_cap = __ballot(hit && r1.y > GSA_MAX_ALPHA) != 0ull;
    int n = __builtin_popcountll(b);
#ifdef GSA_EXP_NOWALK   // experiment: measure everything but the walk
    n = 0;
#endif
    const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
    if (hit) st.list[pos] = (uint16_t)(lane * Staged<F4>::kRecBytes);
    if (lane < 3) st.list[n + lane] = (uint16_t)(kBatch * Staged<F4>::kRecBytes);
    // the null record: blends nothing, carries the list position of the last listed record
    const int last_lane = b ? 63 - __builtin_clzll(b) : 0;
    const int last_idx = __builtin_amdgcn_readlane(idx, last_lane);
    if (lane < F4) {
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 2) z.z = __int_as_float(last_idx);
        st.rec[kBatch * F4 + lane] = z;
    }
    // LDS operations of one wave complete in order; this only stops the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n;
}

// THE ALPHA EXPONENT of a record at the pixels of one 8 x 8 quadrant, as a POLYNOMIAL in quadrant-local coordinates (r04).
// alpha = o exp(-sigma) = exp2(-s) with s = log2e sigma - log2 o.  With (cx, cy) the quadrant's centre, u = mean - (cx, cy)
// and a pixel centre at (cx + x, cy + y), x, y in {-3.5 .. 3.5}:
//   s(x, y) = A (ux - x)^2 + B (ux - x)(uy - y) + C (uy - y)^2 - log2 o = s0 + x (gx + A x + B y) + y (gy + C y)
//   s0 = A ux^2 + B ux uy + C uy^2 - log2 o,  gx = -(2 A ux + B uy),  gy = -(B ux + 2 C uy)   (A = log2e a / 2, B = log2e b, C = log2e c / 2)
// (s0, gx, gy) are computed ONCE per staged record by the lane that stages it (sigma_form); a pixel then pays five fused
// multiply-adds and the exponential -- where "d = mean - pixel, d^T Q d, o * exp" cost two subtractions, five multiply-adds,
// the exponential and a multiplication: 3 of the ~18 vector instructions of a (quadrant, record) visit, in both sweeps, and
// the record needs neither its opacity nor its threshold in LDS: the skip rule "alpha < 1/255" is s > log2 255 for every
// record.  Every step is an explicit fma and the form is a function of the stored record and the quadrant alone, so the
// forward and the backward sweep evaluate bit-identical exponents: a pixel's threshold decisions never differ between
// them.  Rounding: the terms are of the size of s0 (the exponent at the quadrant's centre, <~ 100 for a record that reaches
// the quadrant at all), so s carries ~1e-5 absolute where the direct form carried 1e-6 -- d alpha / alpha <~ 1e-5, an order
// below the parity bound.  What that would break is the rule "sigma < 0: skip" where a pixel centre coincides with the
// mean of an opaque splat (true s = 0, computed -1e-6): the test therefore compares |s| (a free operand modifier).  A
// positive-definite conic -- every conic the projection emits -- and an opacity <= 1 give no other negative.
constexpr float kLog2Of255 = 7.994353436858858f;
struct SigmaForm { float s0, gx, gy; };
__device__ __forceinline__ SigmaForm sigma_form(float mx, float my, float A, float B, float C, float opacity, float cx,
                                                float cy) {
#pragma clang fp contract(off)
    const float ux = mx - cx, uy = my - cy;   // (exact: |mean - centre| is small against the operands whenever the record reaches the quadrant)
    const float Aux = A * ux, Cuy = C * uy;
    SigmaForm f;
    f.gx = -__builtin_fmaf(B, uy, Aux + Aux);
    f.gy = -__builtin_fmaf(B, ux, Cuy + Cuy);
    // (opacity 0 -- a culled or padding record -- gives s0 = +inf: alpha 0 everywhere)
    f.s0 = __builtin_fmaf(ux, __builtin_fmaf(B, uy, Aux), Cuy * uy) - __builtin_amdgcn_logf(opacity);
    return f;
}
// (x: NaN for a finished pixel -- s is then NaN and fails the alpha test)
__device__ __forceinline__ float alpha_exponent(float s0, float gx, float gy, float A, float B, float C, float x, float y) {
#pragma clang fp contract(off)
    const float t1 = __builtin_fmaf(B, y, __builtin_fmaf(A, x, gx));
    const float t2 = __builtin_fmaf(C, y, gy);
    return __builtin_fmaf(y, t2, __builtin_fmaf(x, t1, s0));
}

// alpha of one record at one pixel, 0 if the record is skipped there.  The skip rule "o exp(-sigma) < 1/255" is ONE
// compare, |s| <= log2 255 (a NaN s fails it); `off` (all ones or zero) switches a lane off by turning s into a NaN
// first.  The 0.999 cap is only compiled in for batches that hold an opacity above it (s >= -log2 o otherwise keeps
// alpha <= o).
template <bool CLAMP>
__device__ __forceinline__ float record_alpha(float s, uint32_t off, float &raw, bool &hit) {
    raw = __builtin_amdgcn_exp2f(-s);
    const float al = CLAMP ? fminf(GSA_MAX_ALPHA, raw) : raw;
    hit = __builtin_fabsf(__uint_as_float(__float_as_uint(s) | off)) <= kLog2Of255;
    return hit ? al : 0.f;
}
template <bool CLAMP>
__device__ __forceinline__ float record_alpha(float s, uint32_t off, float &raw) {
    bool hit;
    return record_alpha<CLAMP>(s, off, raw, hit);
}

template <int F4>
__device__ __forceinline__ const float4 *rec_at(const Staged<F4> &st, uint32_t byte_off) {
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(st.rec) + byte_off);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Per-pixel state of the forward walk.  A pixel is LIVE until the stop rule fires (T (1 - alpha) <= 1e-4 at a record that
// passes the alpha test).  From then on (r05) its T is FROZEN at the transmittance in front of that record -- which is what
// the image wants as the final T, and which is > 1e-4, so a finished pixel never trips the optimistic group test below --
// its x is a NaN (alpha 0 at every later record: the walk needs no per-record "is this lane off" operand) and `alive` is 0.
// (r04 parked T at 1 and kept the final T and a dead flag in two more registers, and every record of a careful group paid
// three selects for them.)
template <int CH>
struct FwdPixel {
    float T, acc[CH];
    float x;          // the pixel centre's x relative to the quadrant's centre (-3.5 .. 3.5) -- NaN once the pixel is finished
    uint32_t alive;   // all ones while the pixel accumulates, 0 once it has stopped (or lies outside the image)
    int last;
};

#ifndef GSA_LIST_MODE
#define GSA_LIST_MODE 2      // forward lists for the backward sweep: 1 = reach-test survivors, 2 = blended records only
#endif
#ifndef GSA_FWD_GROUP
#define GSA_FWD_GROUP 4      // records per trip of the forward walk (2 or 4)
#endif
#ifndef GSA_FWD_WAVES
#define GSA_FWD_WAVES 1      // minimum waves per SIMD the forward kernel is compiled for (register budget)
#endif
#ifndef GSA_BWD_WAVES
#define GSA_BWD_WAVES 1      // the same for the backward kernel
#endif

// One group of G list entries starting at entry i: per-record alpha (0 where skipped / pixel finished), colours,
// list positions.
template <int CH, int F4, bool CLAMP, int G, typename ST>
__device__ __forceinline__ void eval_group(const ST &st, int i, float xrel, float yrel,
                                           float (&a)[G], float (&col)[G][CH], int (&idx)[G],
                                           unsigned long long (&passed)[G]) {
    uint32_t off[G];
    if constexpr (ST::kDense) {   // the staged records ARE the walk list: slot i + e
#pragma unroll
        for (int e = 0; e < G; ++e) off[e] = (uint32_t)(i + e) * (uint32_t)(16 * F4);
    } else if constexpr (G == 4) {   // one LDS broadcast gives G list entries
        const uint2 o4 = *reinterpret_cast<const uint2 *>(&st.list[i]);
        off[0] = o4.x & 0xFFFFu; off[1] = o4.x >> 16; off[G - 2] = o4.y & 0xFFFFu; off[G - 1] = o4.y >> 16;
    } else {
        const uint32_t o2 = *reinterpret_cast<const uint32_t *>(&st.list[i]);
        off[0] = o2 & 0xFFFFu; off[1] = o2 >> 16;
    }
#pragma unroll
    for (int e = 0; e < G; ++e) {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(st.rec) + off[e]);
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        float raw;
        bool hit;
        // (xrel is NaN for a finished pixel)
        a[e] = record_alpha<CLAMP>(alpha_exponent(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, xrel, yrel), 0u, raw, hit);
        passed[e] = __ballot(hit);   // the compare's own lane mask: which pixels take this record (no extra VALU)
        col[e][0] = q1.z; col[e][1] = q1.w; col[e][2] = q2.x;
        if (CH == 4) col[e][CH - 1] = q2.z;
        idx[e] = __float_as_int(q2.y);
    }
}

// contrib |= (taken != 0) << index, on the scalar unit (written as SALU instructions: left to itself the compiler keeps
// the 64-bit set in VGPRs and spends ~6 vector instructions per record on it -- in a VALU-bound loop).
__device__ __forceinline__ void note_taken(unsigned long long &contrib, unsigned long long taken, int index) {
    unsigned long long bit;
    asm volatile("s_lshl_b64 %1, 1, %3\n\t"
                 "s_cmp_lg_u64 %2, 0\n\t"
                 "s_cselect_b64 %1, %1, 0\n\t"
                 "s_or_b64 %0, %0, %1"
                 : "+s"(contrib), "=&s"(bit)
                 : "s"(taken), "s"(index)
                 : "scc");
}

// Walks n list entries.  Returns false when all 64 pixels of the wave are finished.
// LISTS: `contrib` receives one bit per list entry that was BLENDED at one pixel of the quadrant at least (alpha test
// passed, pixel live and not stopping there) -- the records the backward sweep has to visit here.
__device__ __forceinline__ int select_bits(uint32_t mask, int a, int b) {   // mask ? a : b, bit by bit (v_bfi_b32: one plain-rate instruction)
    return (int)(((uint32_t)a & mask) | ((uint32_t)b & ~mask));
}
template <int CH, int F4, bool CLAMP, int LISTS, typename ST>
__device__ __forceinline__ bool walk_fwd(const ST &st, int n, float yrel, FwdPixel<CH> &px,
                                         unsigned long long &contrib) {
    constexpr int G = GSA_FWD_GROUP;
    int i = 0;
    while (i < n) {
        // Optimistic loop: as long as no pixel stops inside a group (T only shrinks, so testing the group's last
        // value is enough; finished pixels have alpha 0 and a frozen T above the threshold) no per-record select is needed.
        float a[G], col[G][CH];
        int idx[G];
        for (; i < n; i += G) {
            float t[G + 1];
            unsigned long long passed[G];
            eval_group<CH, F4, CLAMP, G>(st, i, px.x, yrel, a, col, idx, passed);
            t[0] = px.T;
            // T (1 - alpha) as ONE fused step, here and on the careful path: the rounding must not depend on the path
#pragma unroll
            for (int e = 0; e < G; ++e) t[e + 1] = __builtin_fmaf(-a[e], t[e], t[e]);
            if (__ballot(t[G] > GSA_TRANSMITTANCE_EPS) != ~0ull) break;
            if (LISTS == 2) {   // (no pixel stops in this group: taking a record = blending it)
#pragma unroll
                for (int e = 0; e < G; ++e) note_taken(contrib, passed[e], i + e);
            }
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const float wgt = a[e] * t[e];
                // explicit fused steps: the rounding must not depend on the path or the position in the group
#pragma unroll
                for (int k = 0; k < CH; ++k) px.acc[k] = __builtin_fmaf(col[e][k], wgt, px.acc[k]);
            }
            px.T = t[G];
            px.last = select_bits(px.alive, idx[G - 1], px.last);   // last record seen while live (the backward starts there)
        }
        if (i >= n) break;
        // some pixel stops in the group at i: the careful path, record by record, on the alphas the optimistic evaluation left
        // (a pixel that stops inside the group has its later alphas switched off by `alive`; its x becomes a NaN behind the group)
        {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const float ae = __uint_as_float(__float_as_uint(a[e]) & px.alive);
                const float nT = __builtin_fmaf(-ae, px.T, px.T);
                const bool ok = nT > GSA_TRANSMITTANCE_EPS;   // finished pixels: alpha 0, T frozen above the threshold -> true, no effect
                const float wgt = ok ? ae * px.T : 0.f;
                if (LISTS == 2) note_taken(contrib, __ballot(wgt != 0.f), i + e);
#pragma unroll
                for (int k = 0; k < CH; ++k) px.acc[k] = __builtin_fmaf(col[e][k], wgt, px.acc[k]);
                px.T = ok ? nT : px.T;              // the stop rule fired: this record is not blended, T stays what it was
                px.alive = ok ? px.alive : 0u;
                px.last = select_bits(px.alive, idx[e], px.last);
            }
            px.x = __uint_as_float(__float_as_uint(px.x) | ~px.alive);   // (all ones: a NaN)
            i += G;
            if (__ballot(px.alive != 0u) == 0ull) return false;   // all 64 pixels finished
        }
    }
    return true;
}

// ---- the forward kernel ------------------------------------------------------------------------------------------
// A (tile, quadrant) wave walks its bin's depth-ordered record list in three decoupled steps:
//   SCAN   64 list entries at a time, coalesced: with BOXES (gsa_tile_bin16 given the gaussians' quadrant boxes) the top
//          byte of an entry says which quadrants of the bin the gaussian's bounding box covers -- a 4-bit mask of its
//          columns, a 4-bit mask of its rows -- and only the entries whose box covers this wave's quadrant (one AND, one
//          compare; ~28 % in a 32 x 32 bin) are appended to a small queue in LDS: ballot + mbcnt, half a dozen vector
//          instructions per 64 entries, nothing gathered for the others;
//   STAGE  64 queued candidates at a time: one 64-B splat record per lane (the gather is in flight while the previous
//          batch is blended), re-tested against the bounding box of the pixels that are still accumulating, and the
//          survivors written to CONSECUTIVE slots of the wave's LDS image -- so that
//   WALK   needs no offset list: slot i of the image is entry i of the walk (walk_fwd with a dense image).
// Without boxes (the other binning paths) every entry is a candidate and STAGE's test is the filter.
// The wave's LDS image, three float4 per staged record:
//   [0] s0  gx  gy  A          (the record's alpha exponent for THIS quadrant: sigma_form / alpha_exponent)
//   [1] B  C  r  g
//   [2] b  (int) position in the sorted list  depth  -
constexpr int kQueue = 128;        // candidates: < 64 left over + 64 from one scan step
constexpr int kScanWin = 1;        // windows of 64 entries per scan step (two windows, 128 entries in flight: forward sweep +4 % --
                                   // a wave scans at most 64 entries past what its pixels still need, and waits for one load)
template <int F4>
struct Dense {
    static constexpr bool kDense = true;
    float4 rec[(kBatch + GSA_FWD_GROUP - 1) * F4];   // staged records, then NULL records padding the walk to whole groups
    uint2 queue[kQueue];             // (gaussian id, list position)
};

// LISTS: additionally writes, per quadrant, the list positions of the records that were blended there
// (quad_list[4 * start + q * (end - start) ...], ascending) and their number (quad_count[4 * tile + q]): the backward
// sweep then visits exactly those, without a reach test and in full batches.
// (register budget: seven waves per SIMD for the three-channel instances -- 72 registers; a handful of values of the
// prologue and the staging step then live in scratch, none of the walk's; the four-channel instances keep what they need)
template <int CH, int LISTS, bool HAS_BOX>
__global__ __launch_bounds__(kWave, (CH == 3 ? 7 : GSA_FWD_WAVES)) void composite_fwd_kernel(
    int width, int height, int tile_w, int n_tiles, int bin_shift, const float *__restrict__ splat,
    const uint32_t *__restrict__ sorted_gid, const int32_t *__restrict__ tile_offsets,
    const float *__restrict__ background, const int32_t *__restrict__ tile_order, float *__restrict__ render,
    float *__restrict__ alpha_out, float *__restrict__ final_T, int32_t *__restrict__ last_id,
    uint32_t *__restrict__ quad_list, int32_t *__restrict__ quad_count, float4 *__restrict__ zero4, long n_zero4) {
    constexpr int F4 = 3;   // (the depth channel's value rides in the third quad)
    __shared__ Dense<F4> st;

    // zero4 (optional): the gradient records the backward sweep will accumulate into.  Every wave clears its share on
    // the way in -- two fire-and-forget stores per lane under a walk that is bound by vector issue -- instead of a
    // 64-MB fill pass of its own in front of the backward sweep.
    if (zero4) {
        const long per = (n_zero4 + gridDim.x - 1) / gridDim.x;
        const long z0 = (long)blockIdx.x * per, z1 = min(n_zero4, z0 + per);
        for (long i = z0 + threadIdx.x; i < z1; i += kWave) zero4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    int tile, q;
    wave_unit(blockIdx.x, n_tiles, tile_order, tile, q);
    if (tile >= n_tiles) return;
    GSA_TRACE_BEGIN();
    const int lane = threadIdx.x;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int qx = tx * kTile + (q & 1) * 8, qy = ty * kTile + (q >> 1) * 8;
    const int px_ = qx + (lane & 7), py_ = qy + (lane >> 3);
    const bool inside = px_ < width && py_ < height;
    const float qx0 = qx + 0.5f, qy0 = qy + 0.5f, qcx = qx + 4.f, qcy = qy + 4.f;   // first pixel centre; the quadrant's centre
    const float xrel = (float)(lane & 7) - 3.5f, yrel = (float)(lane >> 3) - 3.5f;    // this lane's pixel centre relative to it
    int start, end, slice;   // this tile's record list (its bin's), and which of the bin's quadrant slices is this wave's
    tile_list(tile_offsets, tile, tile_w, bin_shift, q, start, end, slice);
    // this wave's quadrant column / row inside its bin (the bin's quadrant slices are (sub-tile << 2) | quadrant)
    // (in the units of the records' boxes: quadrants, or pairs of quadrants in a 64 x 64 bin)
    const int box_shift = bin_shift > 1 ? 1 : 0;
    const uint32_t my_col = (uint32_t)((((slice >> 2) & ((1 << bin_shift) - 1)) << 1) | (q & 1)) >> box_shift;
    const uint32_t my_row = (uint32_t)((((slice >> 2) >> bin_shift) << 1) | (q >> 1)) >> box_shift;
    const uint32_t my_sel = (1u << (24 + my_col)) | (1u << (28 + my_row));   // (wave-uniform: an SGPR)

    FwdPixel<CH> px;
    px.T = 1.f; px.alive = inside ? ~0u : 0u; px.last = -1;
    px.x = inside ? xrel : __builtin_nanf("");
#pragma unroll
    for (int k = 0; k < CH; ++k) px.acc[k] = 0.f;
    int listed = 0;   // LISTS: entries written to this quadrant's list so far (wave-uniform)
    uint32_t *my_list = LISTS ? quad_list + (((size_t)start << 2) << (2 * bin_shift)) + (size_t)slice * (size_t)(end - start) : nullptr;

    if (__ballot(inside) != 0ull && start < end) {
        // ---- SCAN state: the entries [p, p + 64 kScanWin) are in registers (loads clamped, not predicated)
        int p = start, head = 0, tail = 0;   // (wave-uniform)
        uint32_t s_gid[kScanWin];
        auto scan_issue = [&]() {
#pragma unroll
            for (int w = 0; w < kScanWin; ++w) {
                const int idx = p + w * kBatch + lane, ic = min(idx, end - 1);
                s_gid[w] = sorted_gid[ic];
            }
        };
        auto scan_until = [&](int want) {   // append candidates until `want` are queued or the list ends
            while (tail - head < want && p < end) {
#pragma unroll
                for (int w = 0; w < kScanWin; ++w) {
                    const int idx = p + w * kBatch + lane;
                    // (box: bits 24-27 the column units, bits 28-31 the row units of the bin the record's box covers -- one AND
                    // and one compare against this wave's two bits)
                    const bool hit = idx < end && (!HAS_BOX || (s_gid[w] & my_sel) == my_sel);
                    const unsigned long long b = __ballot(hit);
                    if (b) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
                        if (hit) st.queue[(tail + rank) & (kQueue - 1)] = make_uint2(HAS_BOX ? s_gid[w] & 0xFFFFFFu : s_gid[w], (uint32_t)idx);
                        tail += __builtin_popcountll(b);
                    }
                }
                p += kScanWin * kBatch;
                if (p < end) scan_issue();
            }
        };
        // ---- a batch in flight: the queue's first `n` candidates, one splat record per lane
        struct InFlight { Row row; int pos, n; };
        auto pop = [&]() -> InFlight {
            InFlight f;
            f.n = min(kBatch, tail - head);
            // (the queue writes above and this read are LDS operations of one wave: they complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint2 e = st.queue[(head + min(lane, max(f.n - 1, 0))) & (kQueue - 1)];
            f.pos = (int)e.y;
            f.row = load_row(splat, e.x, lane < f.n);
            head += f.n;
            return f;
        };
        scan_issue();
        scan_until(kBatch);
        InFlight cur = pop();
        GSA_TRACE_MARK(0);
        while (cur.n > 0) {
            scan_until(kBatch);   // (vector work under the gather of `cur`)
            // ---- STAGE: reach test against the bounding box of the pixels that are still accumulating (lane = 8 y + x) -- as
            // the quadrant saturates, and at the image border, fewer records can reach anything that still matters --
            // survivors to consecutive slots
            const unsigned long long alive = __ballot(px.alive != 0u);
            const uint32_t cols8 = (uint32_t)alive | (uint32_t)(alive >> 32);
            const uint32_t cols = (cols8 | (cols8 >> 16) | (cols8 >> 8) | (cols8 >> 24)) & 0xFFu;
            const float bx0 = qx0 + (float)__builtin_ctz(cols), bx1 = qx0 + (float)(31 - __builtin_clz(cols));
            const float by0 = qy0 + (float)(__builtin_ctzll(alive) >> 3), by1 = qy0 + (float)((63 - __builtin_clzll(alive)) >> 3);
            const float4 r0 = cur.row.r0, r1 = cur.row.r1, r2 = cur.row.r2;
            const bool hit = lane < cur.n && quadrant_reach(r0.x, r0.y, r0.z, r0.w, r1.x, r2.z, bx0, by0, bx1, by1);
            const unsigned long long hb = __ballot(hit);
            const bool any_cap = __ballot(hit && r1.y > GSA_MAX_ALPHA) != 0ull;
            const int n = __builtin_popcountll(hb);
            const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0));
            if (hit) {
                const float A = 0.5f * kLog2e * r0.z, B = kLog2e * r0.w, C = 0.5f * kLog2e * r1.x;
                const SigmaForm f = sigma_form(r0.x, r0.y, A, B, C, r1.y, qcx, qcy);
                st.rec[F4 * slot + 0] = make_float4(f.s0, f.gx, f.gy, A);
                st.rec[F4 * slot + 1] = make_float4(B, C, r1.z, r1.w);
                st.rec[F4 * slot + 2] = make_float4(r2.x, __int_as_float(cur.pos), r2.y, 0.f);
            }
            // NULL records (exponent +inf: alpha 0) pad the walk to whole groups; they carry the list position of the
            // last real entry (a pixel that is live behind the group has "seen" exactly that far)
            const int last_pos = __builtin_amdgcn_readlane(cur.pos, hb ? 63 - __builtin_clzll(hb) : 0);
            if (lane < (GSA_FWD_GROUP - 1) * F4) {
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane % F4 == 0) z.x = __builtin_inff();
                if (lane % F4 == 2) z.y = __int_as_float(last_pos);
                st.rec[F4 * n + lane] = z;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            GSA_TRACE_MARK(1);
            cur = pop();          // the next batch's gather is in flight during the walk
            GSA_TRACE_COUNT(n);
            unsigned long long contrib = 0ull;
            const bool live = any_cap ? walk_fwd<CH, F4, true, LISTS>(st, n, yrel, px, contrib)
                                      : walk_fwd<CH, F4, false, LISTS>(st, n, yrel, px, contrib);
            if (LISTS == 1) contrib = ~0ull;                   // every record the reach test kept
            contrib &= n >= 64 ? ~0ull : (1ull << n) - 1ull;   // the padding (null) entries "pass" with alpha 0
            if (LISTS && contrib) {   // lane i < n <-> walk entry i: append the blended ones, in list order
                const bool mine = (contrib >> lane) & 1ull;
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(contrib >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)contrib, 0));
                if (mine) my_list[listed + rank] = (uint32_t)__float_as_int(st.rec[F4 * lane + 2].y);
                listed += __builtin_popcountll(contrib);
            }
            GSA_TRACE_MARK(4);
            if (!live) break;
        }
    }

    if (inside) {
        const size_t pix = (size_t)py_ * width + px_;
        const float Tf = px.T;   // (a finished pixel's T is frozen at the transmittance in front of the record that stopped it)
#pragma unroll
        for (int k = 0; k < CH; ++k) render[pix * CH + k] = px.acc[k] + (background ? Tf * background[k] : 0.f);
        alpha_out[pix] = 1.f - Tf;
        final_T[pix] = Tf;  // kept beside alpha: 1 - (1 - T) loses the low bits of a small T
        last_id[pix] = px.last;
    }
    if (LISTS && lane == 0) quad_count[4 * tile + q] = listed;
    GSA_TRACE_MARK(6);
    GSA_TRACE_END(0, tile * 4 + q, end - start);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Transpose-reduce steps.  fold32(x, y): lanes [0,32) get x summed over both wave halves, lanes [32,64) the
// same for y (v_permlane32_swap).  fold16: rows 0 and 2 get x summed over their row pair, rows 1 and 3 the same
// for y (v_permlane16_swap).  fold8 / fold4 / fold2: the lanes whose bit 3 / 2 / 1 is clear get x summed with
// the partner lane, the others y.  After the five levels every lane pair holds the wave-wide sum of ONE of the
// inputs -- about one instruction per input at the top, instead of six per input for plain butterflies.
__device__ __forceinline__ float fold32(float x, float y) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float x, float y) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// fold8 / fold4 as two DPP adds with complementary bank masks (a bank = 4 lanes of a row): the first writes
// x + partner(x) into the lanes that keep x, the second y + partner(y) into the others -- no selects.
// s_nop 1: a DPP read of a VGPR the previous VALU instruction wrote needs two wait states, and the compiler's
// hazard pass cannot see into inline assembly.
__device__ __forceinline__ float fold8(float x, float y) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "+v"(x) : "v"(y));
    return x;
}
__device__ __forceinline__ float fold4(float x, float y) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa"
                 : "+v"(x) : "v"(y));
    return x;
}
template <int CTRL>
__device__ __forceinline__ float fold_sel(float x, float y, bool hi) {
    const float keep = hi ? y : x, send = hi ? x : y;
    return keep + dpp_mov<CTRL>(send);
}
constexpr int kQuadXor2 = 0x4E, kQuadXor1 = 0xB1;

// Sums 4 * NF per-lane values (NF fields of FOUR records, v[4 * f + e]) over the wave.  The levels are ordered by
// cost: the two bank-masked DPP levels (lane bits 3, 2) take the 36 (40) -> 9 (10) halvings, the permlane swaps
// (bits 5, 4) the next two, the select folds (bits 1, 0) the last two values.  On return the lane with
//   record e = 2 b2 + b3,   field f = 2 (2 r + b4) + b5  with r = b1 (b0 = 0)  or  r = 2, b1 = b4 = 0 (b0 = 1)
// (b_k = bit k of the lane id) holds the wave-wide total of v[4 * f + e]; the other lanes hold duplicates.
template <int NF>
__device__ __forceinline__ float reduce_quad(const float *v, int lane) {
    static_assert(NF == 9 || NF == 10, "field count");
    float k2[2 * NF];   // after bit 3: index = 2 f + (e >> 1), lanes with b3 hold the odd record of the pair
#pragma unroll
    for (int i = 0; i < 2 * NF; ++i) k2[i] = fold8(v[2 * i], v[2 * i + 1]);
    float m[10];        // after bit 2: index = f
#pragma unroll
    for (int f = 0; f < NF; ++f) m[f] = fold4(k2[2 * f], k2[2 * f + 1]);
    if (NF == 9) m[9] = m[8];
    float n[5];         // after bit 5: index = f >> 1
#pragma unroll
    for (int i = 0; i < 5; ++i) n[i] = fold32(m[2 * i], m[2 * i + 1]);
    const float r0 = fold16(n[0], n[1]), r1 = fold16(n[2], n[3]), r2 = fold16(n[4], n[4]);   // after bit 4
    const float s0 = fold_sel<kQuadXor2>(r0, r1, (lane & 2) != 0);                            // after bit 1
    const float s1 = r2 + dpp_mov<kQuadXor2>(r2);
    return fold_sel<kQuadXor1>(s0, s1, (lane & 1) != 0);                                      // after bit 0
}

// Walks n list entries (records in REVERSE list order), four per trip.
template <int CH, int F4, bool CLAMP>
__device__ __forceinline__ void walk_bwd(const Staged<F4> &st, int n, int lane, float pxf, float pyf, float qcx, float qcy,
                                         int binf, const float (&vout)[CH], float &T, float &Wd, int my_field, bool my_add,
                                         float *__restrict__ v_splat) {
    constexpr int NF = 6 + CH;
    for (int i = 0; i < n; i += 4) {
        const uint2 o4 = *reinterpret_cast<const uint2 *>(&st.list[i]);
        const uint32_t off[4] = {o4.x & 0xFFFFu, o4.x >> 16, o4.y & 0xFFFFu, o4.y >> 16};
        float v[4 * NF];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 *p = rec_at(st, off[e]);
            const float4 q0 = p[0], q1 = p[1], q2 = p[2];
            const float dx = q0.x - pxf, dy = q0.y - pyf;
            // the pixel stopped in front of this record (its list position is beyond the pixel's last one):
            // binf - position is negative, its sign switches the lane off
            const uint32_t off = (uint32_t)((binf - __float_as_int(q2.z)) >> 31);
            // (this sweep is not the hot one: it recomputes the record's sigma form -- the SAME arithmetic on the same
            // stored record as the forward's staging lane, so the same bits -- instead of staging it)
            const SigmaForm f = sigma_form(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, qcx, qcy);
            float raw;
            const float a = record_alpha<CLAMP>(alpha_exponent(f.s0, f.gx, f.gy, q0.z, q0.w, q1.x, pxf - qcx, pyf - qcy),
                                                off, raw);
            // a == 0: ra = 1, fac = 0 -- T, Wd and the sums pass through unchanged, no selects needed
            const float ra = __builtin_amdgcn_rcpf(1.f - a);
            const float Tn = T * ra;             // transmittance in front of this splat
            const float fac = a * Tn;
            float cv = q1.z * vout[0] + q1.w * vout[1] + q2.x * vout[2];
            if (CH == 4) cv += p[F4 - 1].x * vout[CH - 1];
            const float v_a = Tn * cv + ra * Wd;
            Wd -= fac * cv;
            T = Tn;
            // d loss / d sigma = -alpha v_a where the record is blended and (CLAMP) not capped at 0.999 (a == raw)
            const float u = CLAMP ? ((a == raw) ? -raw * v_a : 0.f) : -a * v_a;
            const float udx = u * dx, udy = u * dy;
            v[0 + e] = udx; v[4 + e] = udy; v[8 + e] = udx * dx; v[12 + e] = udx * dy; v[16 + e] = udy * dy;
            v[20 + e] = u;
            v[24 + e] = fac * vout[0]; v[28 + e] = fac * vout[1]; v[32 + e] = fac * vout[2];
            if (CH == 4) v[36 + e] = fac * vout[CH - 1];
        }
        // one transpose-reduce and one atomic instruction (four 64-B requests) per FOUR records
#ifdef GSA_EXP_NOREDUCE   // ablation (profiles/r02_bwd_ablation.md): per-pixel math kept alive, no cross-lane sum
#pragma unroll
        for (int k = 0; k < 4 * NF; ++k) asm volatile("" ::"v"(v[k]));
        const float total = v[0];
#else
        const float total = reduce_quad<NF>(v, lane);
#endif
        // this lane's record: e = 2 b2 + b3 -> its list entry -> the gaussian id stored with the staged record
        const uint32_t word = (lane & 4) ? o4.y : o4.x;
        const uint32_t my_off = (lane & 8) ? word >> 16 : word & 0xFFFFu;
        const int e = ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
#ifdef GSA_EXP_NOATOMIC   // ablation: everything but the atomic instruction (the sum stays alive)
        asm volatile("" ::"v"(total), "v"(my_off), "v"(e));
#else
        if (my_add && i + e < n) {
            const uint32_t g = __float_as_uint(rec_at(st, my_off)[2].w);
#ifdef GSA_EXP_STORE      // ablation: a plain store in the atomic's place (wrong sums; prices the atomic itself)
            v_splat[(size_t)g * GSA_SPLAT_STRIDE + my_field] = total;
#else
            atomicAdd(v_splat + (size_t)g * GSA_SPLAT_STRIDE + my_field, total);
#endif
        }
#endif
    }
}

// Gradient record written by the backward sweep (v_splat[N,16], consumed by project_bwd):
//   [0] sum u dx   [1] sum u dy   [2] sum u dx^2   [3] sum u dx dy   [4] sum u dy^2   [5] sum u
//   [6..8] sum fac * v_out(rgb)   [9] sum fac * v_out(depth)
// with u = d loss / d sigma at one pixel and d = mean2d - pixel: the raw moments of u.  Mean, conic and
// opacity gradients are linear in them (project_bwd finishes that per gaussian, once, instead of per pixel).
template <int CH>
__global__ __launch_bounds__(kWave, GSA_BWD_WAVES) void composite_bwd_kernel(
    int width, int height, int tile_w, int n_tiles, int bin_shift, const float *__restrict__ splat,
    const uint32_t *__restrict__ sorted_gid, uint32_t gid_mask, const int32_t *__restrict__ tile_offsets,
    const float *__restrict__ background, const int32_t *__restrict__ tile_order, const float *__restrict__ final_T,
    const int32_t *__restrict__ last_id, const float *__restrict__ v_render, const float *__restrict__ v_alpha,
    float *__restrict__ v_splat) {
    constexpr int F4 = CH == 4 ? 4 : 3;
    constexpr int NF = 6 + CH;
    __shared__ Staged<F4> st;

    int tile, q;
    wave_unit(blockIdx.x, n_tiles, tile_order, tile, q);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x;
    int start, end, slice;   // this tile's record list (its bin's), and which of the bin's quadrant slices is this wave's
    tile_list(tile_offsets, tile, tile_w, bin_shift, q, start, end, slice);
    if (end <= start) return;
    GSA_TRACE_BEGIN();
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int qx = tx * kTile + (q & 1) * 8, qy = ty * kTile + (q >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < width && py < height;
    const float pxf = px + 0.5f, pyf = py + 0.5f, qx0 = qx + 0.5f, qy0 = qy + 0.5f, qcx = qx + 4.f, qcy = qy + 4.f;

    // per pixel: T = transmittance behind the current record; Wd = T_final (v_alpha - bg.v_out) - sum of
    // (colour . v_out) * weight over the splats behind -- what d(out)/d(alpha_i) multiplies by 1/(1-alpha_i)
    const size_t pix = inside ? (size_t)py * width + px : 0;
    const int binf = inside ? last_id[pix] : -1;   // last list position the pixel saw while live
    int mx = binf;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
    const int qmax = __builtin_amdgcn_readfirstlane(mx);  // last position any pixel of this quadrant saw
    if (qmax < start) return;                              // nothing was blended here (also covers qmax == -1)

    // software pipeline as in the forward; lane l of batch j holds list position qmax - 64 j - l (walk order)
    auto load_gid = [&](int j) -> uint32_t {
        const int p = qmax - j * kBatch - lane;
        return p >= start ? sorted_gid[p] & gid_mask : 0u;
    };
    uint32_t gid_next = load_gid(0);
    const float Tf = inside ? final_T[pix] : 1.f;
    float T = Tf;
    float vout[CH], bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vout[k] = inside ? v_render[pix * CH + k] : 0.f;
        if (background) bgdot += background[k] * vout[k];
    }
    float Wd = Tf * (((inside && v_alpha) ? v_alpha[pix] : 0.f) - bgdot);
    Row row = load_row(splat, gid_next, qmax - lane >= start);
    gid_next = load_gid(1);

    // which field's total this lane holds after reduce_quad (see there), and whether it is the lane that adds it
    const int my_r = (lane & 1) ? 2 : ((lane >> 1) & 1);
    const int my_field = 2 * (2 * my_r + ((lane >> 4) & 1)) + ((lane >> 5) & 1);
    const bool my_add = my_field < NF && (!(lane & 1) || !(lane & (2 | 16)));   // b0 = 1: fields 8, 9 live at b1 = b4 = 0

    GSA_TRACE_MARK(0);
    for (int top = qmax, j = 0; top >= start; top -= kBatch, ++j) {
        const int idx = top - lane;
        bool any_cap;
        const int n = stage_batch(st, lane, row, idx >= start, idx, qx0, qy0, qx0 + 7.f, qy0 + 7.f, any_cap);
        GSA_TRACE_MARK(1);
        row = load_row(splat, gid_next, idx - kBatch >= start);
        gid_next = load_gid(j + 2);
        GSA_TRACE_COUNT(n);
        if (any_cap)
            walk_bwd<CH, F4, true>(st, n, lane, pxf, pyf, qcx, qcy, binf, vout, T, Wd, my_field, my_add, v_splat);
        else
            walk_bwd<CH, F4, false>(st, n, lane, pxf, pyf, qcx, qcy, binf, vout, T, Wd, my_field, my_add, v_splat);
        GSA_TRACE_MARK(4);
    }
    GSA_TRACE_END(1, tile * 4 + q, qmax + 1 - start);
}

// ------------------------------------------------------------------------------------------------
// backward, two phases per 8 records: pixel-parallel recurrence, record-parallel moment sums
// ------------------------------------------------------------------------------------------------
// The sweep over the forward pass's per-quadrant lists of BLENDED records (composite_fwd_kernel<CH, 2>): no reach test, no
// compaction, every staged record is walked -- full batches of 64, and none of the ~10 % of reach-test survivors that blend
// nowhere.  Lane l of batch j stages list entry count - 1 - 64 j - l (walk order = reverse).
// What a pixel-parallel sweep spends on the cross-lane sums of the moments (~20 of its ~58 vector instructions per
// record, profiles/r02_bwd_ablation.md) is avoided by changing the lane mapping between the two halves of the work:
//   phase A, lane = pixel: the sequential part -- alpha, T, Wd -- for 8 records; per (pixel, record) only the two
//            scalars the moments are linear in, u = d loss / d sigma and fac = alpha T, go to an LDS matrix
//            [pixel][record] (rows padded to 9 entries: conflict-free 8-B stores);
//   phase B, lane = (pixel column x = lane >> 3, record = lane & 7), 8 steps over the pixel rows y: every lane reads its
//            record's (u, fac) at pixel (x, y) and ACCUMULATES the nine moments in registers with plain FMAs
//            (d = mean - pixel is a per-lane constant in x and a per-step constant in y);
//   then one short fold over the 8 lanes of a record (3 levels) and a transposition through LDS give two atomic
//   instructions of four whole 64-B gradient records each.
// (An XOR-swizzled matrix removes phase B's 2-way bank conflicts -- SQ_LDS_BANK_CONFLICT 17.2 M -> 0.02 M cycles -- for one
// v_xor per store and the same kernel time, profiles/r03_bwd_ablation.md: not kept.)
//
// LDS per wave (r04: 7936 B, 8192 before):
//   rec   64 x 3 float4    [0] s0 gx gy A  [1] B C r g  [2] b (int) list position  depth  -
//                          (the forward's image: the record's alpha exponent for THIS quadrant, sigma_form / alpha_exponent)
//   mat   64 x 9 float2    [pixel][record] (u, fac) of the current sub-batch; the ninth entry of row l -- the padding that keeps
//                          phase A's stores conflict-free -- holds mean2d of staged record l (phase B's d = mean - pixel);
//                          words 18 r + f (r < 8, f < 6 + CH) of the same rows double as the epilogue's [record][field]
//                          totals once phase B has read the matrix
//   gid   64 uint32        gaussian index of staged record l (the epilogue's atomics)
constexpr int kSub = 8;                 // records per sub-batch
constexpr int kMatPitch = kSub + 1;     // float2 entries per pixel row of the LDS matrix
constexpr int kTilePitch = 2 * kMatPitch;   // words between the epilogue's record rows (= a matrix row)
template <int F4>
struct TwoPhase {
    float4 rec[kBatch * F4];
    float2 mat[kWave * kMatPitch];
    uint32_t gid[kBatch];
};

// Lane l stages ITS record of the batch (list position idx; !valid: a record that blends nothing).  Returns the number of
// valid records (a prefix of the lanes); any_cap: does one of them exceed the 0.999 alpha cap.
// ---- replicas of the gradient records of HOT gaussians (r05) ------------------------------------------------------------------
// Every (quadrant, record) visit of the sweep ends in one 64-B float-atomic request on the gaussian's gradient record.  A
// camera INSIDE the scene sees splats that cover hundreds to thousands of tiles: a full-screen splat takes ~30 000 requests
// on ONE line, which the memory side serialises (measured: bench.py's wide_depth pass, the sweep 0.535 ms with the atomics,
// 0.158 ms with plain stores in their place).  So a gaussian whose tile rectangle is at least GSA_HOT_MIN_WIDTH tiles wide
// (word 11 of its splat record) gets GSA_HOT_REPLICAS gradient rows of its own in the HOT REGION behind the caller's N
// gradient records -- [N rows | GSA_HOT_REPLICAS x GSA_HOT_ENTRIES replica rows | GSA_HOT_ENTRIES tag words], all zero when
// the sweep starts -- and a wave adds into replica (4 tile + quadrant) mod GSA_HOT_REPLICAS.  Entries are claimed on first
// touch: hash of the gaussian index, compare-and-swap on the tag (0 -> index + 1), a few probes; the claiming lane leaves
// entry + 1 in the reserved word 11 of the gaussian's OWN gradient record, where the projection backward -- which reads that
// record anyway -- finds it and adds the replicas (project.hip: fold_hot_replicas).  A gaussian that finds no entry adds to
// its own record as before; nothing changes for gaussians below the width (the bench scene C3 has none above it).
__device__ __forceinline__ uint32_t hot_row(float *__restrict__ v_splat, long n_rows, uint32_t g, uint32_t replica) {
    uint32_t *tags = reinterpret_cast<uint32_t *>(v_splat + (n_rows + (long)GSA_HOT_ENTRIES * GSA_HOT_REPLICAS) * GSA_SPLAT_STRIDE);
    uint32_t h = g * 2654435761u;
#pragma unroll 1
    for (int probe = 0; probe < GSA_HOT_PROBES; ++probe) {
        const uint32_t e = h >> (32 - GSA_HOT_LOG_ENTRIES);
        h = h * 2246822519u + 3266489917u;
        // (agent scope: the XCDs' L2s are not coherent for plain loads; a tag only ever goes 0 -> index + 1 inside a sweep)
        uint32_t t = __hip_atomic_load(&tags[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0u) {
            t = atomicCAS(&tags[e], 0u, g + 1u);
            if (t == 0u) {   // this lane claimed the entry: tell the projection backward where the replicas are
                t = g + 1u;
                reinterpret_cast<uint32_t *>(v_splat)[(size_t)g * GSA_SPLAT_STRIDE + GSA_HOT_MARK_FIELD] = e + 1u;
            }
        }
        if (t == g + 1u) return (uint32_t)n_rows + replica * GSA_HOT_ENTRIES + e;   // (replica-major: a gaussian's replicas lie 256 KB apart)
    }
    return g;
}

// (HOT: `row_of` receives the gradient ROW of the staged record -- the gaussian's own, or one of its replicas)
template <int F4, bool HOT = false>
__device__ __forceinline__ int stage_listed(TwoPhase<F4> &tp, int lane, const Row &row, bool valid, int idx, float qcx,
                                            float qcy, bool &any_cap, float *__restrict__ v_splat = nullptr, long n_rows = 0,
                                            uint32_t replica = 0u) {
    const float4 r0 = row.r0, r1 = row.r1, r2 = row.r2;
    const float A = 0.5f * kLog2e * r0.z, B = kLog2e * r0.w, C = 0.5f * kLog2e * r1.x;
    const SigmaForm f = sigma_form(r0.x, r0.y, A, B, C, r1.y, qcx, qcy);
    tp.rec[F4 * lane + 0] = make_float4(f.s0, f.gx, f.gy, A);
    tp.rec[F4 * lane + 1] = make_float4(B, C, r1.z, r1.w);
    tp.rec[F4 * lane + 2] = make_float4(r2.x, __int_as_float(idx), r2.y, 0.f);
    tp.mat[lane * kMatPitch + kSub] = make_float2(r0.x, r0.y);
    uint32_t grad_row = row.gid;
    if constexpr (HOT) {   // word 11 of the splat record: tile x0 | y0 << 10 | columns << 20 (0: not packable -- never hot)
        if (valid && ((__float_as_uint(r2.w) >> 20) & 0x3FFu) >= (uint32_t)GSA_HOT_MIN_WIDTH)
            grad_row = hot_row(v_splat, n_rows, row.gid, replica);
    }
    tp.gid[lane] = grad_row;
    const unsigned long long b = __ballot(valid);          // valid lanes are a prefix: lane < n
    any_cap = __ballot(valid && r1.y > GSA_MAX_ALPHA) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return __builtin_popcountll(b);
}

// (register budget: five waves per SIMD for the three-channel instance -- 96 registers; the four-channel instance would
// spill there and keeps what it needs)
template <int CH, bool HOT>
__global__ __launch_bounds__(kWave, (CH == 3 ? 5 : GSA_BWD_WAVES)) void composite_bwd_2p_kernel(
    int width, int height, int tile_w, int n_tiles, int bin_shift, const float *__restrict__ splat,
    const uint32_t *__restrict__ sorted_gid, uint32_t gid_mask, const int32_t *__restrict__ tile_offsets,
    const float *__restrict__ background, const int32_t *__restrict__ tile_order, const float *__restrict__ final_T,
    const int32_t *__restrict__ last_id, const float *__restrict__ v_render, const float *__restrict__ v_alpha,
    const uint32_t *__restrict__ quad_list, const int32_t *__restrict__ quad_count, float *__restrict__ v_splat,
    float *__restrict__ v_rgb, long hot_rows_at) {
    // HOT: the hot region lies behind row `hot_rows_at` (= N) of v_splat (hot_row above); v_rgb is not given then.
    // v_rgb (optional, [N,3], zeroed by the caller; gsa_composite_bwd_lists_rgb): the colour moments (fields 6..8) are added
    // THERE instead of into the gradient record, already masked by the colour clamp (a channel the forward clamped at 0 --
    // its staged colour is exactly 0 -- has no gradient): d loss / d rgb per gaussian in the 12-byte layout the view-parallel
    // exchange all-gathers, straight out of the sweep -- no projection backward in front of the collective.
    constexpr int F4 = 3;
    constexpr int NF = 6 + CH;
    static_assert(NF <= 2 * kSub, "the epilogue's totals of a record fit the non-padding words of a matrix row");
    __shared__ TwoPhase<F4> tp;
    float *tile = reinterpret_cast<float *>(tp.mat);   // [record][field] totals at words kTilePitch * r + f

    int tile_id, q;
    wave_unit(blockIdx.x, n_tiles, tile_order, tile_id, q);
    if (tile_id >= n_tiles) return;
    const int lane = threadIdx.x;
    const int count = __builtin_amdgcn_readfirstlane(quad_count[4 * tile_id + q]);
    if (count <= 0) return;
    int start, end, slice;   // this tile's record list (its bin's), and which of the bin's quadrant slices is this wave's
    tile_list(tile_offsets, tile_id, tile_w, bin_shift, q, start, end, slice);
    const uint32_t *my_list = quad_list + (((size_t)start << 2) << (2 * bin_shift)) + (size_t)slice * (size_t)(end - start);
    const int tx = tile_id % tile_w, ty = tile_id / tile_w;
    const int qx = tx * kTile + (q & 1) * 8, qy = ty * kTile + (q >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < width && py < height;
    const float xrel = (float)(lane & 7) - 3.5f, yrel = (float)(lane >> 3) - 3.5f;    // this lane's pixel centre relative to the quadrant's centre

    // software pipeline three deep: list entries of batch j + 2, gaussian ids of batch j + 1, records of batch j
    auto load_pos = [&](int j) -> int {
        const int e = count - 1 - j * kBatch - lane;
        return e >= 0 ? (int)my_list[e] : -1;
    };
    auto load_gid = [&](int pos) -> uint32_t { return pos >= 0 ? sorted_gid[pos] & gid_mask : 0u; };
    int pos_cur = load_pos(0);
    int pos_next = load_pos(1);
    uint32_t gid_cur = load_gid(pos_cur);

    // phase A state (lane = pixel): T = transmittance behind the current record; Wn = (T_final (v_alpha - bg.v_out) - sum of
    // (colour . v_out) * weight over the splats behind) / T -- d(out)/d(alpha_i) = Tn (c_i . v_out + Wn)
    const size_t pix = inside ? (size_t)py * width + px : 0;
    const int binf = inside ? last_id[pix] : -1;   // last list position the pixel saw while live
    const float Tf = inside ? final_T[pix] : 1.f;
    float T = Tf;
    float vout[CH], bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vout[k] = inside ? v_render[pix * CH + k] : 0.f;
        if (background) bgdot += background[k] * vout[k];
    }
    float Wn = ((inside && v_alpha) ? v_alpha[pix] : 0.f) - bgdot;   // (Wd / T with Wd = T_final (v_alpha - bg . v_out) at the start)
    Row row = load_row(splat, gid_cur, pos_cur >= 0);
    uint32_t gid_next = load_gid(pos_next);
    int pos_next2 = load_pos(2);

    // phase B roles: record = lane & 7 of the sub-batch, pixel column = lane >> 3, one pixel row per step
    const int rec = lane & 7, col = lane >> 3;
    const float colx = qx + col + 0.5f, rowy0 = qy + 0.5f;
    float vcol[8][CH];   // v_out of this lane's pixel column (registers: the LDS pipe is the scarcer resource here)
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const bool in = qx + col < width && qy + y < height;
        const size_t pp = in ? (size_t)(qy + y) * width + qx + col : 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) vcol[y][k] = in ? v_render[pp * CH + k] : 0.f;
    }
    const int f0 = 4 * ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1);   // field this lane ends up with

    for (int done = 0, j = 0; done < count; done += kBatch, ++j) {
        bool any_cap;
        // (the quadrant's centre, rebuilt from scalars per batch: two conversions instead of two registers held through the
        // sweep, which runs at its 96-register budget)
        int cxi = qx + 4, cyi = qy + 4;
        asm volatile("" : "+s"(cxi), "+s"(cyi));
        const int n = stage_listed<F4, HOT>(tp, lane, row, pos_cur >= 0, pos_cur, (float)cxi, (float)cyi, any_cap, v_splat,
                                            hot_rows_at, (uint32_t)(4 * tile_id + q) & (GSA_HOT_REPLICAS - 1));
        row = load_row(splat, gid_next, pos_next >= 0);
        pos_cur = pos_next;
        gid_next = load_gid(pos_next2);
        pos_next = pos_next2;
        pos_next2 = load_pos(j + 3);
        for (int s0 = 0; s0 < n; s0 += kSub) {
            const int nv = min(kSub, n - s0);
            // ---- phase A: lane = pixel, the records of the sub-batch in walk order
            auto phase_a = [&](auto clamp_tag, auto all_on_tag, auto full_tag) {
                constexpr bool CLAMP = decltype(clamp_tag)::value;
                constexpr bool ALL_ON = decltype(all_on_tag)::value;   // every pixel is behind every record of the sub-batch
                // FULL: a whole sub-batch (all but a list's last one) -- straight-line code for its eight records instead of
                // a uniform branch and a re-materialised LDS address per record: -1 % of the sweep's time
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int r = 0; r < kSub; ++r) {
                    float u = 0.f, fac = 0.f;
                    if (FULL || r < nv) {   // (uniform)
                        const float4 *p = &tp.rec[F4 * (s0 + r)];
                        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                        // the pixel stopped in front of this record (its list position is beyond the pixel's last one)
                        const uint32_t off = ALL_ON ? 0u : (uint32_t)((binf - __float_as_int(q2.y)) >> 31);
                        float raw;
                        const float a = record_alpha<CLAMP>(alpha_exponent(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, xrel, yrel),
                                                            off, raw);
                        const float ra = __builtin_amdgcn_rcpf(1.f - a);   // a == 0: ra = 1, fac = 0, u = 0
                        const float Tn = T * ra;
                        fac = a * Tn;
                        float cv = q1.z * vout[0] + q1.w * vout[1] + q2.x * vout[2];
                        if (CH == 4) cv += q2.z * vout[CH - 1];
                        // d out / d alpha = Tn (cv + Wn) with Wn = (what lies behind, weighted) / T: the NORMALISED form of the
                        // recurrence (r05) -- Wn' = Wn (1 - a) - a cv = Wn - a (cv + Wn) -- needs no 1 / (1 - a) of its own and one
                        // instruction less per (pixel, record) than carrying Wd = Wn T (v_a = Tn cv + ra Wd; Wd -= fac cv)
                        const float q = cv + Wn;
                        Wn = __builtin_fmaf(-a, q, Wn);
                        T = Tn;
                        u = CLAMP ? ((a == raw) ? -fac * q : 0.f) : -fac * q;
                    }
                    tp.mat[lane * kMatPitch + r] = make_float2(u, fac);
                }
            };
            // the lists are walked from the deep end: once every pixel of the quadrant has reached its last position (the
            // sub-batch's first record is its deepest) the "pixel stopped in front of this record" test is dead code
            const int pos_hi = __builtin_amdgcn_readfirstlane(__float_as_int(tp.rec[F4 * s0 + 2].y));
            const bool all_on = __ballot(binf >= pos_hi) == ~0ull;
            if (any_cap) {
                if (all_on) phase_a(std::true_type{}, std::true_type{}, std::false_type{});
                else phase_a(std::true_type{}, std::false_type{}, std::false_type{});
            } else if (nv == kSub) {
                if (all_on) phase_a(std::false_type{}, std::true_type{}, std::true_type{});
                else phase_a(std::false_type{}, std::false_type{}, std::true_type{});
            } else {
                if (all_on) phase_a(std::false_type{}, std::true_type{}, std::false_type{});
                else phase_a(std::false_type{}, std::false_type{}, std::false_type{});
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- phase B: lane = (pixel column, record); the moments of ITS record over ITS column
            float acc[10];
#pragma unroll
            for (int f = 0; f < 10; ++f) acc[f] = 0.f;
            {
                // d = mean - pixel is constant in x for this lane and, per step, a compile-time offset in y: the lane sums
                // u, u y', u y'^2 (y' = y - 3.5: three FMAs per step) and expands the five moments of u once per sub-batch
                // (records beyond nv: whatever batch slot s0 + rec holds -- finite, and their u and fac are zeros)
                const float2 mean = tp.mat[(s0 + rec) * kMatPitch + kSub];
                const float dx = mean.x - colx, dyc = mean.y - (rowy0 + 3.5f);   // dy = dyc - y'
                float su = 0.f, suy = 0.f, suyy = 0.f;
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const float2 uf = tp.mat[(8 * y + col) * kMatPitch + rec];
                    const float yc = (float)y - 3.5f;
                    const float u = uf.x, fac = uf.y;
                    su += u;
                    suy = __builtin_fmaf(u, yc, suy);
                    suyy = __builtin_fmaf(u, yc * yc, suyy);
                    acc[6] = __builtin_fmaf(fac, vcol[y][0], acc[6]);
                    acc[7] = __builtin_fmaf(fac, vcol[y][1], acc[7]);
                    acc[8] = __builtin_fmaf(fac, vcol[y][2], acc[8]);
                    if (CH == 4) acc[9] = __builtin_fmaf(fac, vcol[y][CH - 1], acc[9]);
                }
                const float sudy = __builtin_fmaf(dyc, su, -suy);                      // sum u dy
                acc[0] = dx * su;                                                       // sum u dx
                acc[1] = sudy;
                acc[2] = dx * acc[0];                                                   // sum u dx^2
                acc[3] = dx * sudy;                                                     // sum u dx dy
                acc[4] = __builtin_fmaf(dyc, sudy, __builtin_fmaf(-dyc, suy, suyy));    // sum u dy^2 = dyc (dyc su - suy) - dyc suy + suyy
                acc[5] = su;
            }
            // fold over the 8 lanes of a record (lane bits 3, 4, 5): after fold8 a lane holds field 2 i + b3 in slot i,
            // after fold16 field 4 j + 2 b4 + b3 in slot j, after fold32 field 8 t + 4 b5 + 2 b4 + b3 in slot t
            const float k0 = fold8(acc[0], acc[1]), k1 = fold8(acc[2], acc[3]), k2 = fold8(acc[4], acc[5]),
                        k3 = fold8(acc[6], acc[7]), k4 = fold8(acc[8], acc[9]);
            const float m0 = fold16(k0, k1), m1 = fold16(k2, k3), m2 = fold16(k4, k4);
            const float t0 = fold32(m0, m1), t1 = fold32(m2, m2);
            // (the totals go where phase B has just read the matrix: same wave, LDS operations complete in order, and the
            // stored values depend on every load)
            tile[kTilePitch * rec + f0] = t0;
            if (8 + f0 < NF) tile[kTilePitch * rec + 8 + f0] = t1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // one atomic instruction per four records: lanes (record 4 t + (lane >> 4), field lane & 15)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = 4 * t + (lane >> 4), f = lane & 15;
                if (4 * t < nv) {
                    float total = tile[kTilePitch * r + min(f, NF - 1)];
                    if (f < NF && r < nv) {
                        const uint32_t g = tp.gid[s0 + r];
                        float *dst = v_splat + (size_t)g * GSA_SPLAT_STRIDE + f;
                        if (v_rgb && f >= 6 && f < 9) {   // (uniform pointer test; the record image's floats 6..8 ARE r, g, b)
                            if (reinterpret_cast<const float *>(tp.rec)[4 * F4 * (s0 + r) + f] == 0.f) total = 0.f;
                            dst = v_rgb + (size_t)g * 3 + (f - 6);
                        }
                        atomicAdd(dst, total);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launch order: longest record lists first, inside each XCD band
// ------------------------------------------------------------------------------------------------
// One workgroup per band: counting sort of the band's tiles by list length (64 buckets of 64 records, longest
// first).  The order inside a bucket comes from LDS atomics and may differ between launches -- it only moves
// work in time, never changes a result.
constexpr int kSchedThreads = 1024;
template <int SHIFT>
__global__ __launch_bounds__(kSchedThreads) void tile_schedule_kernel(int n_tiles, const int32_t *__restrict__ tile_offsets,
                                                                      int32_t *__restrict__ tile_order, int tile_w) {
    __shared__ int cnt[kSchedBuckets], first[kSchedBuckets];
    tile_schedule_band<SHIFT, kSchedThreads>(blockIdx.x, n_tiles, tile_offsets, tile_order, tile_w, cnt, first);
}

// Start of a backward pass over the forward's lists: ONE launch zero-fills the gradient records (the sweep
// accumulates into them with atomics) and, in its first eight workgroups, orders each band's tiles by the number of
// records their four quadrants blended -- the exact work of the sweep, known since the forward pass (list length is
// a poor proxy for it in a dense scene: the pixels saturate long before the lists end).
constexpr int kPrepThreads = 1024, kPrepBuckets = 256;
__global__ __launch_bounds__(kPrepThreads) void bwd_prepare_kernel(long n_rows, float4 *__restrict__ v_splat4, int n_tiles,
                                                                   const int32_t *__restrict__ quad_count,
                                                                   int32_t *__restrict__ tile_order) {
    if (blockIdx.x >= 8) {   // kPrepThreads records of 64 B per workgroup, written as consecutive 16-B words
        const long base = ((long)blockIdx.x - 8) * kPrepThreads * 4, end = 4 * n_rows;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long i = base + k * kPrepThreads + threadIdx.x;
            if (i < end) v_splat4[i] = z;
        }
        return;
    }
    __shared__ int cnt[kPrepBuckets], first[kPrepBuckets];
    const int chunk = (n_tiles + 7) >> 3;
    const int lo = blockIdx.x * chunk, hi = min(n_tiles, lo + chunk);
    if (threadIdx.x < kPrepBuckets) cnt[threadIdx.x] = 0;
    __syncthreads();
    auto bucket = [&](int t) {
        const int4 w = *reinterpret_cast<const int4 *>(quad_count + 4 * (long)t);
        return kPrepBuckets - 1 - min(kPrepBuckets - 1, (w.x + w.y + w.z + w.w) >> 2);   // bucket 0 = most work
    };
    for (int t = lo + threadIdx.x; t < hi; t += kPrepThreads) atomicAdd(&cnt[bucket(t)], 1);
    __syncthreads();
    bucket_starts<kPrepBuckets>(cnt, first);
    __syncthreads();
    for (int t = lo + threadIdx.x; t < hi; t += kPrepThreads) tile_order[lo + atomicAdd(&first[bucket(t)], 1)] = t;
    for (int t = max(hi, lo) + threadIdx.x; t < lo + chunk; t += kPrepThreads) tile_order[t] = -1;   // padding
}

}  // namespace

extern "C" int gsa_composite_bwd_prepare(long long n, float *v_splat, int n_tiles, const int32_t *quad_count,
                                         int32_t *tile_order, gsa_stream_t stream) {
    // (n == 0: the records were cleared already -- gsa_composite_fwd's v_splat_zero -- and only the tiles are ordered)
    GSA_RETURN_IF(n < 0 || n_tiles <= 0 || !quad_count || !tile_order || (n > 0 && !v_splat), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(((reinterpret_cast<uintptr_t>(v_splat) | reinterpret_cast<uintptr_t>(quad_count)) & 15) != 0,
                  GSA_ERR_INVALID_ARGUMENT);
    const long grid = 8 + (n + kPrepThreads - 1) / kPrepThreads;
    bwd_prepare_kernel<<<(unsigned)grid, kPrepThreads, 0, static_cast<hipStream_t>(stream)>>>(
        (long)n, reinterpret_cast<float4 *>(v_splat), n_tiles, quad_count, tile_order);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

static int bin_shift_of(int bin_tile) {
    return bin_tile == 4 * kTile ? 2 : bin_tile == 2 * kTile ? 1 : (bin_tile == 0 || bin_tile == kTile ? 0 : -1);
}

extern "C" int gsa_tile_schedule(int n_tiles, const int32_t *tile_offsets, int32_t *tile_order, int tile_w, int bin_tile,
                                 gsa_stream_t stream) {
    GSA_RETURN_IF(n_tiles <= 0 || !tile_offsets || !tile_order || bin_shift_of(bin_tile) < 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(bin_shift_of(bin_tile) > 0 && (tile_w <= 0 || n_tiles % tile_w != 0), GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (bin_shift_of(bin_tile)) {
        case 0: tile_schedule_kernel<0><<<8, kSchedThreads, 0, s>>>(n_tiles, tile_offsets, tile_order, tile_w); break;
        case 1: tile_schedule_kernel<1><<<8, kSchedThreads, 0, s>>>(n_tiles, tile_offsets, tile_order, tile_w); break;
        default: tile_schedule_kernel<2><<<8, kSchedThreads, 0, s>>>(n_tiles, tile_offsets, tile_order, tile_w); break;
    }
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

#ifdef GSA_TRACE
extern "C" int gsa_debug_set_trace(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf));
}
#endif

extern "C" int gsa_composite_fwd(int width, int height, int channels, const float *splat,
                                 const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                                 const float *background,
                                 const int32_t *tile_order, float *render, float *alpha, float *final_T,
                                 int32_t *last_id, uint32_t *quad_list, int32_t *quad_count, float *v_splat_zero,
                                 long long n_zero_rows, int bin_tile, gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    const int bin_shift = bin_shift_of(bin_tile);
    GSA_RETURN_IF(bin_shift < 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(gid_boxes != 0 && gid_boxes != 1, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !render || !alpha || !final_T || !last_id, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((quad_list == nullptr) != (quad_count == nullptr), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(n_zero_rows < 0 || (reinterpret_cast<uintptr_t>(v_splat_zero) & 15) != 0, GSA_ERR_INVALID_ARGUMENT);
    float4 *zero4 = n_zero_rows > 0 ? reinterpret_cast<float4 *>(v_splat_zero) : nullptr;
    const long n_zero4 = (long)n_zero_rows * (GSA_SPLAT_STRIDE / 4);
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    const int grid = 32 * gsa_div_up(n_tiles, 8);   // one wave per (tile, quadrant)
    hipStream_t s = static_cast<hipStream_t>(stream);
#define GSA_FWD_LAUNCH(C, L, M)                                                                                       \
    composite_fwd_kernel<C, L, M><<<grid, kWave, 0, s>>>(width, height, tile_w, n_tiles, bin_shift, splat, sorted_gid, \
                                                        tile_offsets, background, tile_order, render, alpha,            \
                                                        final_T, last_id, quad_list, quad_count, zero4, n_zero4)
#define GSA_FWD_LAUNCH_CL(C, L) do { if (gid_boxes) GSA_FWD_LAUNCH(C, L, true); else GSA_FWD_LAUNCH(C, L, false); } while (0)
    if (quad_list) {
        if (channels == 3) GSA_FWD_LAUNCH_CL(3, GSA_LIST_MODE); else GSA_FWD_LAUNCH_CL(4, GSA_LIST_MODE);
    } else {
        if (channels == 3) GSA_FWD_LAUNCH_CL(3, 0); else GSA_FWD_LAUNCH_CL(4, 0);
    }
#undef GSA_FWD_LAUNCH_CL
#undef GSA_FWD_LAUNCH
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_composite_bwd(int width, int height, int channels, const float *splat,
                                 const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets, const float *background,
                                 const int32_t *tile_order, const float *final_T, const int32_t *last_id,
                                 const float *v_render, const float *v_alpha, float *v_splat, int bin_tile,
                                 gsa_stream_t stream) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !final_T || !last_id || !v_render || !v_splat, GSA_ERR_INVALID_ARGUMENT);
    const int bin_shift = bin_shift_of(bin_tile);
    GSA_RETURN_IF(bin_shift < 0, GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    const int grid = 32 * gsa_div_up(n_tiles, 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t gid_mask = gid_boxes ? 0xFFFFFFu : 0xFFFFFFFFu;
    if (channels == 3)
        composite_bwd_kernel<3><<<grid, kWave, 0, s>>>(width, height, tile_w, n_tiles, bin_shift, splat, sorted_gid, gid_mask,
                                                      tile_offsets, background, tile_order, final_T, last_id, v_render, v_alpha,
                                                      v_splat);
    else
        composite_bwd_kernel<4><<<grid, kWave, 0, s>>>(width, height, tile_w, n_tiles, bin_shift, splat, sorted_gid, gid_mask,
                                                      tile_offsets, background, tile_order, final_T, last_id, v_render, v_alpha,
                                                      v_splat);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}


static int composite_bwd_lists_impl(int width, int height, int channels, const float *splat,
                                       const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                                       const float *background, const int32_t *tile_order, const float *final_T,
                                       const int32_t *last_id, const float *v_render, const float *v_alpha,
                                       const uint32_t *quad_list, const int32_t *quad_count, float *v_splat,
                                       float *v_rgb, int bin_tile, gsa_stream_t stream, long long hot_n = -1) {
    GSA_RETURN_IF(width <= 0 || height <= 0 || (channels != 3 && channels != 4), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!tile_offsets || !final_T || !last_id || !v_render || !v_splat || !quad_list || !quad_count,
                  GSA_ERR_INVALID_ARGUMENT);
    // (rows are addressed with 32 bits in the wave's LDS image)
    GSA_RETURN_IF(hot_n >= 0 && (v_rgb || hot_n + gsa_hot_rows() >= (1ll << 31)), GSA_ERR_INVALID_ARGUMENT);
    const int bin_shift = bin_shift_of(bin_tile);
    GSA_RETURN_IF(bin_shift < 0, GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(width, kTile), tile_h = gsa_div_up(height, kTile);
    const int n_tiles = tile_w * tile_h;
    const int grid = 32 * gsa_div_up(n_tiles, 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t gid_mask = gid_boxes ? 0xFFFFFFu : 0xFFFFFFFFu;
#define GSA_BWD_2P(CH_, HOT_)                                                                                             \
    composite_bwd_2p_kernel<CH_, HOT_><<<grid, kWave, 0, s>>>(width, height, tile_w, n_tiles, bin_shift, splat, sorted_gid,    \
                                                              gid_mask, tile_offsets, background, tile_order, final_T, last_id, \
                                                              v_render, v_alpha, quad_list, quad_count, v_splat, v_rgb,       \
                                                              (long)hot_n)
    if (channels == 3) {
        if (hot_n >= 0) GSA_BWD_2P(3, true); else GSA_BWD_2P(3, false);
    } else {
        if (hot_n >= 0) GSA_BWD_2P(4, true); else GSA_BWD_2P(4, false);
    }
#undef GSA_BWD_2P
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_composite_bwd_lists(int width, int height, int channels, const float *splat,
                                       const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                                       const float *background, const int32_t *tile_order, const float *final_T,
                                       const int32_t *last_id, const float *v_render, const float *v_alpha,
                                       const uint32_t *quad_list, const int32_t *quad_count, float *v_splat,
                                       int bin_tile, gsa_stream_t stream) {
    return composite_bwd_lists_impl(width, height, channels, splat, sorted_gid, gid_boxes, tile_offsets, background, tile_order,
                                    final_T, last_id, v_render, v_alpha, quad_list, quad_count, v_splat, nullptr, bin_tile, stream);
}

extern "C" long long gsa_hot_rows(void) { return GSA_HOT_ROWS; }

extern "C" int gsa_composite_bwd_lists_hot(int width, int height, int channels, const float *splat,
                                           const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                                           const float *background, const int32_t *tile_order, const float *final_T,
                                           const int32_t *last_id, const float *v_render, const float *v_alpha,
                                           const uint32_t *quad_list, const int32_t *quad_count, long long n, float *v_splat,
                                           int bin_tile, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0, GSA_ERR_INVALID_ARGUMENT);
    return composite_bwd_lists_impl(width, height, channels, splat, sorted_gid, gid_boxes, tile_offsets, background, tile_order,
                                    final_T, last_id, v_render, v_alpha, quad_list, quad_count, v_splat, nullptr, bin_tile, stream, n);
}

extern "C" int gsa_composite_bwd_lists_rgb(int width, int height, int channels, const float *splat,
                                           const uint32_t *sorted_gid, int gid_boxes, const int32_t *tile_offsets,
                                           const float *background, const int32_t *tile_order, const float *final_T,
                                           const int32_t *last_id, const float *v_render, const float *v_alpha,
                                           const uint32_t *quad_list, const int32_t *quad_count, float *v_splat,
                                           float *v_rgb, int bin_tile, gsa_stream_t stream) {
    GSA_RETURN_IF(!v_rgb, GSA_ERR_INVALID_ARGUMENT);
    return composite_bwd_lists_impl(width, height, channels, splat, sorted_gid, gid_boxes, tile_offsets, background, tile_order,
                                    final_T, last_id, v_render, v_alpha, quad_list, quad_count, v_splat, v_rgb, bin_tile, stream);
}
