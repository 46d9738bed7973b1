// sort.hip -- stage 2: tile binning (row a3.3 of the scope table).
//
// The published pipeline sorts M (tile<<32 | depth_bits, gaussian) records with a 64-bit-key radix
// sort.  The required order is (tile, depth bits, gaussian index).  Here it is produced with far less
// HBM traffic by splitting the key:
//   1. stable LSD radix sort of the N gaussians by their 32 depth bits (values = gaussian index, so
//      ties keep index order);
//   2. records are EMITTED in that order (exclusive scan of tiles_touched in depth order);
//   3. stable LSD radix sort of the M records by tile id only (<= 16 bits -> 2 passes instead of 6).
// All kernels are bandwidth-bound integer work: coalesced 4-B-per-lane streams, LDS histograms,
// wave64 ballot ranking.  No MFMA.
//
// Replaces gsplat's isect_tiles / radix sort / isect_offset_encode behind
// /root/reference/src/backends/gsplat_backend.py:168-182.

#include "project_math.h"
#include "tile_schedule.h"

#ifdef GSA_TRACE   // developer build: per-workgroup phase times of the binning kernels (tools/trace_sort.py)
__device__ unsigned long long *g_sort_trace = nullptr;
#define SORT_TRACE_BEGIN() unsigned long long st_t0 = 0, st_mark = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    if (threadIdx.x == 0) st_t0 = st_mark = __builtin_amdgcn_s_memrealtime()
#define SORT_TRACE_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    st_acc[k] += now_ - st_mark; st_mark = now_; } } while (0)
#define SORT_TRACE_END(kernel) do { if (g_sort_trace && threadIdx.x == 0 && blockIdx.x < 32768) { \
    unsigned long long *r_ = g_sort_trace + ((size_t)(kernel) * 32768 + blockIdx.x) * 16; \
    r_[0] = st_t0; r_[1] = __builtin_amdgcn_s_memrealtime(); for (int k_ = 0; k_ < 8; ++k_) r_[2 + k_] = st_acc[k_]; } } while (0)
extern "C" int gsa_debug_set_sort_trace(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sort_trace), &buf, sizeof(buf));
}
#else
#define SORT_TRACE_BEGIN()
#define SORT_TRACE_MARK(k)
#define SORT_TRACE_END(kernel)
#endif

namespace {

constexpr int kSortThreads = 256;
#ifndef GSA_SORT_ITEMS
#define GSA_SORT_ITEMS 16
#endif
constexpr int kSortItems = GSA_SORT_ITEMS;               // keys per thread
constexpr int kSortChunk = kSortThreads * kSortItems;   // keys per block
constexpr int kRadix = 256;
static_assert(kSortChunk == GSA_BIN_BLOCK, "the fused binning's block-owner table is laid out in blocks of GSA_BIN_BLOCK record slots");

// Item counts may live on the device (the record count M is produced by the scan and the host does not wait for
// it): kernels are launched for a CAPACITY and clamp to *n_dev when it is given.  Uniform address -> scalar load.
__device__ __forceinline__ int resolve_count(int n_cap, const uint32_t *__restrict__ n_dev) {
    return n_dev ? (int)min((uint32_t)n_cap, *n_dev) : n_cap;
}

// Digit of a key in one radix pass.  Plain: bits [shift, shift + 8).  MSD bucketing of depth keys (msd flag; `shift`
// then holds a 32-bit SCALE): bucket = ((key - sub) * scale) >> 32 clamped to 254 -- 255 buckets of (nearly) equal key
// width whatever the range, where a power-of-two width left a quarter of them empty and the fullest one at the LDS
// capacity -- and the culled sentinel 0xFFFFFFFF alone in bucket 255 --
// monotone in the key, so buckets are depth ranges (gsa_depth_order_msd).
struct RadixDigit {
    int shift;
    uint32_t sub;
    int msd;                 // 0: plain bits; 1: 255 equal MSD buckets from (sub, shift = scale); 2: a depth table on the device,
                             // not yet resolved (resolve_table); 3: resolved -- bucket = lut[(key >> shift) - sub]
    const uint32_t *table;   // msd >= 2: the view's depth table (kTabWords words, built by the histogram pass: build_depth_table)
    const uint8_t *lut;      // msd == 3: its cell -> bucket map (global memory: 2 KB, served by L1)
    uint32_t cells;          // msd == 3: cells in use
};

// ---- depth buckets that fit the view at hand (gsa_depth_order_msd with key_range_dev) --------------------------------
// gsa_project_fwd leaves, per wave of 64 gaussians, (min, max) of the visible depth keys and two of the keys themselves.
// The first launch of the ordering turns them into THIS view's DEPTH TABLE: the exact key range [lo, hi] is cut into at
// most 4096 aligned CELLS of 2^shift keys (cell = (key >> shift) - g0), the ~2 N / 64 sample keys are counted per cell, and
// consecutive cells are grouped into at most 255 buckets that hold about the same number of keys each -- and never more
// cells than 16-bit key offsets cover (65536 >> shift).  A monotone map: buckets are still depth ranges and the order
// produced is the same whatever the table; what the table decides is how full the fullest bucket is.  Depth is far from
// uniformly distributed in key space (float bits are denser above every power of two; a frustum holds more gaussians far
// away): 255 EQUAL buckets put 12.6 k of 1 M keys into one bucket for the diagonal cameras of the bench ring -- 3.5 x the
// mean, beyond the 8192-key LDS capacity of the local sorts, so those views fell back to the LSD path -- and 6.7 k for the
// axis-aligned ones.  ONE workgroup builds the table, in a launch of its own in front of the histogram pass 
constexpr int kCellsMax = 4096;
constexpr int kTabLut = 0;                   // 4096 bytes: cell -> bucket
constexpr int kTabCs = 1024;                 // 257 words: bucket -> first cell (then the end)
constexpr int kTabPlan = 1284;               // g0, shift, cells, flags
constexpr int kTabWords = 1288;
constexpr uint32_t kPlanWide = 1u;           // the key range is wider than 250 x 2^24 keys (no float depth range is): 24-bit local sorts cannot order it
constexpr uint32_t kLocalBits = 24;          // key bits a bucket's local sort covers (r04; 16 before: two octaves of depth per view)

struct DepthPlan { uint32_t g0, shift, cells, flags; };

// exclusive scan of one value per thread across the kTabThreads threads of the table kernel; total through `total`
constexpr int kTabThreads = 1024;
__device__ __forceinline__ uint32_t tab_excl_scan(uint32_t v, uint32_t *scratch, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < GSA_WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t off = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kTabThreads / 64; ++w) {
        const uint32_t c = scratch[w];
        if (w < wave) off += c;
        sum += c;
    }
    total = sum;
    __syncthreads();
    return off + inc - v;
}

// ONE workgroup, in front of the histogram pass: the view's depth table from what gsa_project_fwd left per wave.
__global__ __launch_bounds__(kTabThreads) void depth_table_kernel(const uint4 *__restrict__ ranges, int n_waves,
                                                                  uint32_t *__restrict__ table_out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_lut[kCellsMax];
    __shared__ uint32_t s_fine[kCellsMax], s_cs[260], scratch[8 * (kTabThreads / 64)];
    const int t = threadIdx.x;
    constexpr int NW = kTabThreads / 64;
    // two totals over the block at once (one barrier; alternating scratch halves, so no second one)
    int phase = 0;
    auto block_sum2 = [&](uint32_t a, uint32_t b_, uint32_t &ta, uint32_t &tb) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += (uint32_t)__shfl_xor((int)a, d); b_ += (uint32_t)__shfl_xor((int)b_, d); }
        uint32_t *sc = scratch + (phase & 1) * 2 * NW;
        ++phase;
        if ((t & 63) == 0) { sc[t >> 6] = a; sc[NW + (t >> 6)] = b_; }
        __syncthreads();
        ta = tb = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { ta += sc[w]; tb += sc[NW + w]; }
    };
    // ---- ONE pass over the per-wave entries: the exact key range now, the sample keys kept in registers for the cells
    constexpr int kMaxTrips = 4;          // 4 x 8 x 1024 waves = 2 M gaussians (beyond that the sample is thinned: every k-th wave)
    const int stride = (n_waves + kMaxTrips * 8 * kTabThreads - 1) / (kMaxTrips * 8 * kTabThreads);
    const int n_eff = (n_waves + stride - 1) / stride;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u, smp[kMaxTrips * 8][2];
#pragma unroll
    for (int trip = 0; trip < kMaxTrips; ++trip) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = (trip * 8 + j) * kTabThreads + t;
            v[j] = ranges[min((long)min(i, n_eff - 1) * stride, (long)n_waves - 1)];
            if (i >= n_eff) v[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lo = min(lo, v[j].x); hi = max(hi, v[j].y);
            smp[trip * 8 + j][0] = v[j].z; smp[trip * 8 + j][1] = v[j].w;
        }
        if ((trip + 1) * 8 * kTabThreads >= n_eff) break;   // (uniform)
    }
    if (stride > 1) {   // the range must be exact: the waves the sample skips still count for it
        for (int i = t; i < n_waves; i += kTabThreads) {
            const uint2 r = *reinterpret_cast<const uint2 *>(&ranges[i]);
            lo = min(lo, r.x); hi = max(hi, r.y);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
    }
    if ((t & 63) == 0) { scratch[4 * NW + (t >> 6)] = lo; scratch[5 * NW + (t >> 6)] = hi; }
    for (int c = t; c < kCellsMax; c += kTabThreads) s_fine[c] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { lo = min(lo, scratch[4 * NW + w]); hi = max(hi, scratch[5 * NW + w]); }
    DepthPlan pl{0u, 0u, 1u, 0u};
    if (lo <= hi) {
        if ((unsigned long long)hi - lo + 1ull > (250ull << kLocalBits)) pl.flags = kPlanWide;   // (250: the cells' alignment and
                                                                                              // rounding need the rest of the 254)
        uint32_t sh = 0;
        while ((hi >> sh) - (lo >> sh) + 1u > (uint32_t)kCellsMax) ++sh;   // (aligned cells of 2^sh keys; <= 21 trips)
        pl.shift = sh;
        pl.g0 = lo >> sh;
        pl.cells = (hi >> sh) - pl.g0 + 1u;
    }
    const uint32_t G = pl.cells;
    const uint32_t max_cells = pl.shift >= kLocalBits ? 1u : ((1u << kLocalBits) >> pl.shift);   // cells whose keys fit the local sorts' offsets (flagged if too few)
    // ---- the sample, per cell
    uint32_t n_smp = 0;
#pragma unroll
    for (int k = 0; k < kMaxTrips * 8; ++k) {
        if ((k / 8) * 8 * kTabThreads >= n_eff && k >= 8) break;   // (uniform: trips that were not loaded)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (smp[k][h] != 0xFFFFFFFFu) { atomicAdd(&s_fine[(smp[k][h] >> pl.shift) - pl.g0], 1u); ++n_smp; }
        }
    }
    uint32_t total, unused;
    block_sum2(n_smp, 0u, total, unused);   // (the barrier in there also covers the counters)
    // ---- cells -> buckets.  Thread t owns the cells [PER t, PER t + PER).  Every cell DEMANDS a fraction of a bucket (16.16 fixed
    // point): its share of the keys at T sample keys per bucket, but never less than 1 / max_cells -- and cell c goes to bucket
    // floor(demand of the cells in front of it).  Monotone; a bucket holds at most T keys + one cell's, and at most max_cells
    // cells (each adds 1 / max_cells or more).  T: the smallest for which the demands add up to 254 buckets or fewer -- first the
    // T that would do if every cell paid both terms (always feasible), then refined from what the cells really pay.
    constexpr int PER = kCellsMax / kTabThreads;
    uint32_t cnt[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) cnt[j] = (uint32_t)(PER * t + j) < G ? s_fine[PER * t + j] : 0u;
    const uint32_t wq = (65536u + max_cells - 1u) / max_cells;                      // a cell's least demand
    const float kBudget = 254.f * 65536.f;
    const float width_all = (float)G * (float)wq;
    // x = 65536 / T (T: sample keys per bucket): demand of a cell = max(cnt x, wq), in 1/65536 buckets
    auto demand = [&](uint32_t c, float x) -> uint32_t { return max((uint32_t)fminf((float)c * x, kBudget), wq); };
    // what the cells pay at x: the count term of the cells it governs (A), the width term of the others (B)
    auto paid = [&](float x, uint32_t &A, uint32_t &B) {
        uint32_t pay_cnt = 0, pay_wid = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if ((uint32_t)(PER * t + j) < G) {
                const uint32_t q = (uint32_t)fminf((float)cnt[j] * x, kBudget);
                if (q > wq) pay_cnt += q; else pay_wid += wq;
            }
        }
        block_sum2(pay_cnt, pay_wid, A, B);
    };
    float x = 0.f;                                                                  // (width only: no sample, or no room)
    if (total && width_all + 4.f * 65536.f < kBudget) x = (kBudget - width_all - 4.f * 65536.f) / (float)total * 0.9999f;
    const float x_feasible = x;
    if (x > 0.f) {
        // The demand D(x) = sum of max(cnt x, w) is convex and increasing.  From the feasible x0: one Newton step (x1: exact if
        // no cell changes sides, otherwise beyond the budget), then the chord between the two, whose root lies on the
        // feasible side of a convex function.
        uint32_t A0, B0, A1, B1;
        paid(x, A0, B0);
        const float D0 = (float)A0 + (float)B0;
        if (A0 > 0u && D0 < kBudget) {
            const float x1 = x * (1.f + (kBudget - D0) / (float)A0);
            paid(x1, A1, B1);
            const float D1 = (float)A1 + (float)B1;
            if (D1 <= kBudget * 0.999f) x = x1;
            else if (D1 > D0) x = fmaxf(x, (x + (x1 - x) * (kBudget - D0) / (D1 - D0)) * 0.995f);
        }
    }
    uint32_t fin[PER];
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t mine = 0, d[PER], all;
#pragma unroll
        for (int j = 0; j < PER; ++j) { d[j] = (uint32_t)(PER * t + j) < G ? demand(cnt[j], x) : 0u; mine += d[j]; }
        uint32_t ex = tab_excl_scan(mine, scratch + 6 * NW, all);
#pragma unroll
        for (int j = 0; j < PER; ++j) { fin[j] = ex >> 16; ex += d[j]; }
        if (all <= (254u << 16) || x == x_feasible) break;   // (block-uniform) -- else: rounding went the wrong way
        x = x_feasible;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) s_lut[PER * t + j] = (uint8_t)min(fin[j], 254u);
    for (int v = t; v < 260; v += kTabThreads) s_cs[v] = G;
    __syncthreads();
    {
        uint32_t prev_f = t > 0 ? (uint32_t)s_lut[PER * t - 1] : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t c = (uint32_t)(PER * t + j), fc = min(fin[j], 254u);
            if (c < G && fc != prev_f) {
                for (uint32_t v = prev_f == 0xFFFFFFFFu ? 0u : prev_f + 1u; v <= fc; ++v) s_cs[v] = c;
            }
            prev_f = fc;
        }
    }
    __syncthreads();
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(s_lut);
    for (int w = t; w < kCellsMax / 4; w += kTabThreads) table_out[kTabLut + w] = lw[w];
    for (int v = t; v < 257; v += kTabThreads) table_out[kTabCs + v] = s_cs[v];
    if (t == 0) {
        table_out[kTabPlan] = pl.g0; table_out[kTabPlan + 1] = pl.shift;
        table_out[kTabPlan + 2] = pl.cells; table_out[kTabPlan + 3] = pl.flags;
    }
}

// msd == 2 -> 3: the plan and the table of THIS view, as the histogram pass left them
__device__ __forceinline__ void resolve_table(RadixDigit &dg) {
    if (dg.msd != 2) return;
    dg.sub = dg.table[kTabPlan];
    dg.shift = (int)dg.table[kTabPlan + 1];
    dg.cells = dg.table[kTabPlan + 2];
    dg.lut = reinterpret_cast<const uint8_t *>(dg.table + kTabLut);
    dg.msd = 3;
}

__device__ __forceinline__ uint32_t digit_of(uint32_t key, const RadixDigit &dg) {
    if (dg.msd == 3) return key == 0xFFFFFFFFu ? 255u : (uint32_t)dg.lut[min((key >> dg.shift) - dg.sub, dg.cells - 1u)];
    if (dg.msd) return key == 0xFFFFFFFFu ? 255u : min(__umulhi(key - dg.sub, (uint32_t)dg.shift), 254u);
    return (key >> dg.shift) & (kRadix - 1);
}

// Which chunk of the input a workgroup of the histogram / scatter kernels owns.  Workgroups go to the eight XCDs round
// robin by their index, and every XCD has its own L2: with chunk = index, the 32 blocks whose histogram words share a
// 128-B line of the digit-major table sit on 8 different XCDs -- every line of the table is pulled into 8 L2s and its
// partial writes meet only at the memory side.  Here the workgroups of ONE XCD own CONSECUTIVE chunks (a bijection on
// [0, gridDim.x) for any grid size).
__device__ __forceinline__ unsigned sort_block_id() {
    const unsigned p = blockIdx.x, nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = p & 7u;
    return x * q + min(x, r) + (p >> 3);
}

// The same for a CAPACITY launch of which only the first `live` chunks hold items: XCD x owns the consecutive live chunks
// [x q + min(x, r), ...) (live = 8 q + r), its surplus workgroups get none (0xFFFFFFFF) -- with sort_block_id over the
// whole grid the live chunks of a launch 25 % above its count filled six XCDs beyond their wave slots and left two idle.
__device__ __forceinline__ unsigned live_block_id(unsigned live) {
    const unsigned p = blockIdx.x, q = live >> 3, r = live & 7u, x = p & 7u, i = p >> 3;
    return i < q + (x < r ? 1u : 0u) ? x * q + min(x, r) + i : 0xFFFFFFFFu;
}

// ---- pass 1: per-block digit histogram, written digit-major: hist[d * nblocks + block] -------------
// Plain per-lane LDS atomics: measured faster than aggregating same-digit lanes with ballots first (the LDS
// resolves same-address adds of one instruction cheaply; the 8 ballots per key cost more than they save).
// block_range (optional, [2][nblocks]): min and max of the block's keys below the culled sentinel 0xFFFFFFFF -- the
// depth sort's first pass leaves them so that the caller can check how many key bits the visible gaussians differ in.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(int n_cap, const uint32_t *__restrict__ n_dev,
                                                                  const KeyT *__restrict__ keys,
                                                                  RadixDigit dg, int nblocks,
                                                                  uint32_t *__restrict__ hist,
                                                                  uint32_t *__restrict__ block_range) {
    __shared__ __attribute__((aligned(16))) uint8_t s_lut[sizeof(KeyT) == 4 ? kCellsMax : 4];
    DepthPlan pl{0u, 0u, 1u, 0u};
    bool tabled = false;
    if constexpr (sizeof(KeyT) == 4) {
        if (dg.msd == 2) {   // this view's depth table (depth_table_kernel): the cell -> bucket map into LDS
            pl.g0 = dg.table[kTabPlan]; pl.shift = dg.table[kTabPlan + 1]; pl.cells = dg.table[kTabPlan + 2];
            uint32_t *lw = reinterpret_cast<uint32_t *>(s_lut);
            for (int w = threadIdx.x; w < kCellsMax / 4; w += kSortThreads) lw[w] = dg.table[kTabLut + w];
            tabled = true;   // (the barrier in front of the counting loop covers these stores)
        }
    }
    // (with a table the digit comes out of the LDS copy; digit_of's msd == 3 form reads the stored one)
    auto digit = [&](uint32_t key) -> uint32_t {
        if (tabled) return key == 0xFFFFFFFFu ? 255u : (uint32_t)s_lut[min((key >> pl.shift) - pl.g0, pl.cells - 1u)];
        return digit_of(key, dg);
    };
    const int shift = dg.shift;
    __shared__ uint32_t h[kRadix];
    __shared__ uint32_t s_lo, s_hi;
    const int n = resolve_count(n_cap, n_dev);
    const long base = (long)sort_block_id() * kSortChunk;
    if (base >= n) {   // capacity launch: nothing live in this block
        hist[(long)threadIdx.x * nblocks + sort_block_id()] = 0;
        return;
    }
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    __syncthreads();
    // A block with all kSortChunk keys live (all but the last one) issues its loads TOGETHER, then counts: loads
    // predicated on i < n compile to one branch + s_waitcnt vmcnt(0) each -- sixteen serial memory round trips.
    const bool full = base + kSortChunk <= n;
    if constexpr (sizeof(KeyT) == 2) {   // two keys per 4-B load (the block's base is a multiple of kSortChunk keys)
        const uint32_t *pairs = reinterpret_cast<const uint32_t *>(keys + base);
        if (full) {
            uint32_t kk[kSortItems / 2];
#pragma unroll
            for (int r = 0; r < kSortItems / 2; ++r) kk[r] = pairs[r * kSortThreads + threadIdx.x];
#pragma unroll
            for (int r = 0; r < kSortItems / 2; ++r) {
                atomicAdd(&h[((kk[r] & 0xFFFFu) >> shift) & (kRadix - 1)], 1u);
                atomicAdd(&h[((kk[r] >> 16) >> shift) & (kRadix - 1)], 1u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < kSortItems / 2; ++r) {
                const int p = r * kSortThreads + threadIdx.x;
                const long i = base + 2 * p;
                if (i + 1 < n) {
                    const uint32_t kk = pairs[p];
                    atomicAdd(&h[((kk & 0xFFFFu) >> shift) & (kRadix - 1)], 1u);
                    atomicAdd(&h[((kk >> 16) >> shift) & (kRadix - 1)], 1u);
                } else if (i < n) {
                    atomicAdd(&h[((uint32_t)keys[i] >> shift) & (kRadix - 1)], 1u);
                }
            }
        }
    } else if (full) {
        uint32_t kk[kSortItems];
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) kk[r] = (uint32_t)keys[base + r * kSortThreads + threadIdx.x];
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            atomicAdd(&h[digit(kk[r])], 1u);
            if (kk[r] != 0xFFFFFFFFu) { lo = min(lo, kk[r]); hi = max(hi, kk[r]); }
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            long i = base + r * kSortThreads + threadIdx.x;
            if (i < n) {
                const uint32_t kk = (uint32_t)keys[i];
                atomicAdd(&h[digit(kk)], 1u);
                if (kk != 0xFFFFFFFFu) { lo = min(lo, kk); hi = max(hi, kk); }
            }
        }
    }
    if (block_range) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    hist[(long)threadIdx.x * nblocks + sort_block_id()] = h[threadIdx.x];
    if (block_range && threadIdx.x == 0) { block_range[sort_block_id()] = s_lo; block_range[nblocks + sort_block_id()] = s_hi; }
}

// ---- pass 2: one block per digit: exclusive scan of its row in place, row total out -----------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < GSA_WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix and
// (through total) the block sum.  scratch: 4 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t &total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
    uint32_t wave_off = wave == 0 ? 0 : wave == 1 ? s0 : wave == 2 ? s0 + s1 : s0 + s1 + s2;
    total = s0 + s1 + s2 + s3;
    __syncthreads();
    return wave_off + inc - v;
}

__global__ __launch_bounds__(kSortThreads) void radix_scan_rows_kernel(int nblocks, uint32_t *__restrict__ hist,
                                                                       uint32_t *__restrict__ digit_total,
                                                                       const uint32_t *__restrict__ block_range = nullptr,
                                                                       uint32_t *__restrict__ key_range = nullptr,
                                                                       uint32_t *__restrict__ zero_words = nullptr,
                                                                       int n_zero = 0) {
    __shared__ uint32_t scratch[4];
    if (zero_words && blockIdx.x == 1)   // control words of the kernel that follows (polled there: must start at zero)
        for (int i = threadIdx.x; i < n_zero; i += kSortThreads) zero_words[i] = 0u;
    if (key_range && blockIdx.x == 0) {   // min / max over the per-block ranges the histogram pass left
        __shared__ uint32_t r_lo, r_hi;
        if (threadIdx.x == 0) { r_lo = 0xFFFFFFFFu; r_hi = 0u; }
        __syncthreads();
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int b = threadIdx.x; b < nblocks; b += kSortThreads) {
            lo = min(lo, block_range[b]);
            hi = max(hi, block_range[nblocks + b]);
        }
        atomicMin(&r_lo, lo);
        atomicMax(&r_hi, hi);
        __syncthreads();
        if (threadIdx.x == 0) {
            key_range[0] = r_lo;
            key_range[1] = r_hi;
            if (zero_words) key_range[2] = 0u;   // (gsa_depth_order_msd: the status word behind the range)
        }
    }
    uint32_t *row = hist + (long)blockIdx.x * nblocks;
    uint32_t carry = 0;
    // four consecutive entries per thread and trip: a long row (27 k blocks at 112 M records) is a chain of trips, each
    // two barriers and a memory round trip long (eight per trip -- one trip for the 2010 blocks of 8.2 M records --
    // measured no faster).  The loads are clamped, not predicated, so that a trip's four are in flight together.
    constexpr int kPer = 4;
    for (int base = 0; base < nblocks; base += kSortThreads * kPer) {
        const int i0 = base + threadIdx.x * kPer;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) v[j] = row[min(i0 + j, nblocks - 1)];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            v[j] = i0 + j < nblocks ? v[j] : 0u;
            sum += v[j];
        }
        uint32_t total;
        uint32_t ex = carry + block_excl_scan(sum, scratch, total);
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (i0 + j < nblocks) row[i0 + j] = ex;
            ex += v[j];
        }
        carry += total;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// Rank of this lane's item among the items of its digit that the wave has seen so far (earlier rounds: cnt[d]; this
// round: the lanes below it holding the same digit).  The lanes of a digit find one another through one 64-bit word
// per digit in LDS: every lane ORs its bit in (ds_or_b64) and reads the word back -- LDS operations of one wave
// complete in order, so the word then holds exactly the round's lanes of that digit -- and the lowest of them clears
// it again and advances cnt[d].  About ten vector and five LDS instructions per round; ranking by eight ballots with
// per-lane mask selects cost ~55 vector instructions per round and was the largest phase of every scatter kernel
// (tools/trace_sort.py).  mask: 256 zeroed words private to the wave.  The fences only pin the compiler's order.
__device__ __forceinline__ uint32_t wave_rank_round(unsigned long long *mask, uint32_t *cnt, uint32_t d, bool valid,
                                                    int lane) {
    const unsigned long long bit = 1ull << lane;
    if (valid) atomicOr(&mask[d], bit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long peers = valid ? mask[d] : 0ull;
    const uint32_t below = __popcll(peers & (bit - 1ull));
    const uint32_t seen = cnt[d];                       // same value for all peers (read before the update)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (valid && below == 0) {
        cnt[d] = seen + (uint32_t)__popcll(peers);
        mask[d] = 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return seen + below;
}

// ---- pass 3: stable scatter ---------------------------------------------------------------------
// A block owns kSortChunk consecutive items, wave w the w-th quarter of them (16 rounds of 64 consecutive items).
//   1. Each wave ranks its own items against WAVE-PRIVATE digit counters in LDS: per round 8 ballots give the
//      lanes holding the same digit (rank inside the round), the counter gives the items of that digit the wave saw
//      in earlier rounds.  No workgroup barrier inside the 16 rounds (the previous version synchronised the four
//      waves twice per round: 32 barriers per block).
//   2. One pass over the 256 digits turns the four waves' counters into block-sorted offsets.
//   3. Items are placed in LDS in block-sorted order, then written out: consecutive threads write consecutive
//      addresses inside each digit's run, so HBM sees 64-B+ runs instead of one 4-B store per 64-B line (the
//      low-digit pass of a tile sort scatters neighbouring records to 256 different buckets).
// keys_out may be null (last pass of a sort whose keys nobody reads).  pairs (optional, [n] uint2 indexed by VALUE):
// the last pass of the depth sort also leaves pairs[value].x / .y in output order (pair_x_out / pair_y_out) -- the
// packed tile rectangle and the record count of each gaussian, which the scan would otherwise gather through `order`.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    int n_cap, const uint32_t *__restrict__ n_dev, const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out, RadixDigit dg, int nblocks,
    const uint32_t *__restrict__ hist, const uint32_t *__restrict__ digit_total,
    const uint2 *__restrict__ pairs = nullptr, uint32_t *__restrict__ pair_x_out = nullptr,
    uint32_t *__restrict__ pair_y_out = nullptr) {
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t gbase[kRadix];        // global slot of this block's first item of digit d
    __shared__ uint32_t lexcl[kRadix];        // slot of that item in the block-sorted order
    __shared__ uint32_t wave_cnt[4][kRadix];  // items of digit d wave w has ranked so far; then: its first block-sorted slot
    __shared__ KeyT s_key[kSortChunk];
    __shared__ __attribute__((aligned(16))) uint32_t s_val[kSortChunk];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = resolve_count(n_cap, n_dev);
    if ((long)sort_block_id() * kSortChunk >= n) return;   // whole block beyond the live items (capacity launch)
    SORT_TRACE_BEGIN();
    resolve_table(dg);

    // (keys and values first: their latency hides behind the histogram columns and the two block scans)
    const long base = (long)sort_block_id() * kSortChunk + wave * (kSortChunk / 4);
    uint32_t k[kSortItems], v[kSortItems], rk[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        long i = base + r * 64 + lane;
        k[r] = i < n ? (uint32_t)keys_in[i] : 0xFFFFFFFFu;
        v[r] = i < n ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }

    {   // digit t: global base (exclusive scan of digit totals + offset of this block inside the digit) and
        // block-local base (exclusive scan of this block's own digit counts, recovered from the scanned rows)
        uint32_t tot, dt = digit_total[t];
        const uint32_t mine = hist[(long)t * nblocks + sort_block_id()];
        const uint32_t next = (int)sort_block_id() + 1 < nblocks ? hist[(long)t * nblocks + sort_block_id() + 1] : dt;
        const uint32_t dbase = block_excl_scan(dt, scratch, tot);
        const uint32_t lbase = block_excl_scan(next - mine, scratch, tot);
        gbase[t] = dbase + mine;
        lexcl[t] = lbase;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wave_cnt[w][t] = 0;
            reinterpret_cast<unsigned long long *>(s_val)[w * kRadix + t] = 0ull;
        }
    }
    __syncthreads();
    SORT_TRACE_MARK(0);   // prologue: histogram columns, block scans, key / value loads

    uint32_t *cnt = wave_cnt[wave];
    // (the value staging area is idle until the reorder: its first 8 KB are the four waves' peer masks, zeroed above)
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(s_val) + wave * kRadix;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        rk[r] = wave_rank_round(mask, cnt, digit_of(k[r], dg), base + r * 64 + lane < n, lane);
    __syncthreads();
    SORT_TRACE_MARK(1);   // ranking
    {   // digit t: first block-sorted slot of each wave's items = lexcl + counts of the waves before it
        uint32_t c0 = wave_cnt[0][t], c1 = wave_cnt[1][t], c2 = wave_cnt[2][t];
        const uint32_t o = lexcl[t];
        wave_cnt[0][t] = o; wave_cnt[1][t] = o + c0; wave_cnt[2][t] = o + c0 + c1; wave_cnt[3][t] = o + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        if (base + r * 64 + lane < n) {
            const uint32_t d = digit_of(k[r], dg);
            const uint32_t lp = cnt[d] + rk[r];
            s_key[lp] = (KeyT)k[r];
            s_val[lp] = v[r];
        }
    }
    __syncthreads();
    SORT_TRACE_MARK(2);   // reorder through LDS
    const int m = (int)min((long)kSortChunk, (long)n - (long)sort_block_id() * kSortChunk);
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        int lp = r * kSortThreads + t;
        if (lp < m) {
            uint32_t key = s_key[lp];
            uint32_t d = digit_of(key, dg);
            uint32_t pos = gbase[d] + ((uint32_t)lp - lexcl[d]);
            if (keys_out) keys_out[pos] = (KeyT)key;
            const uint32_t val = s_val[lp];
            vals_out[pos] = val;
            if (pairs) {
                const uint2 pr = pairs[val];
                pair_x_out[pos] = pr.x;
                pair_y_out[pos] = pr.y;
            }
        }
    }
    SORT_TRACE_MARK(3);   // output stores (issue)
    SORT_TRACE_END(sizeof(KeyT) == 2 ? 1 : 0);
}

// ---- pass 3 for LARGE inputs (r05): the same stable scatter on 22 KB of LDS instead of 38 ---------------------------------
// radix_scatter_kernel stages a block's keys AND values (2 x 16 KB) for the reorder: four blocks a CU, 1024 on the chip -- and
// the depth sort of 5 M gaussians (BASELINE C5; the reference reports "~5.5M gaussians", src/training/trainer.py:92-96) has
// 1221 blocks: a second round of workgroups for the last fifth of them, 61.9 us per pass = 0.16 of the HBM peak
// (profiles/r04_c5_summary.md).  This instance reorders through ONE 16-KB buffer, keys first, values second: the
// destination of a block-sorted slot comes from a 4-KB
// digit image it leaves beside the buffer, so a round is stage -> barrier -> sixteen coalesced runs of stores.  Six blocks a
// CU (1536 on the chip), two barriers more per block.  32-bit keys; same output, bit for bit (same ranks, same destinations).
__global__ __launch_bounds__(kSortThreads, 5) void radix_scatter_compact_kernel(
    int n, const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int shift, int nblocks, const uint32_t *__restrict__ hist,
    const uint32_t *__restrict__ digit_total, const uint2 *__restrict__ pairs, uint32_t *__restrict__ pair_x_out,
    uint32_t *__restrict__ pair_y_out) {
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t gbase[kRadix];        // global slot of this block's first item of digit d
    __shared__ uint32_t lexcl[kRadix];        // slot of that item in the block-sorted order
    __shared__ uint32_t wave_cnt[4][kRadix];  // items of digit d wave w has ranked so far; then: its first block-sorted slot
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[kSortChunk];   // peer masks while ranking; then keys; then values
    __shared__ uint8_t s_dig[kSortChunk];     // digit of the item in block-sorted slot lp
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned blk = sort_block_id();
    if ((long)blk * kSortChunk >= n) return;
    const RadixDigit dg{shift, 0u, 0, nullptr, nullptr, 0u};
    const long base = (long)blk * kSortChunk + wave * (kSortChunk / 4);
    uint32_t k[kSortItems], v[kSortItems], rk[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const long i = base + r * 64 + lane;
        k[r] = i < n ? keys_in[i] : 0xFFFFFFFFu;
        v[r] = i < n ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }
    {
        uint32_t tot, dt = digit_total[t];
        const uint32_t mine = hist[(long)t * nblocks + blk];
        const uint32_t next = (int)blk + 1 < nblocks ? hist[(long)t * nblocks + blk + 1] : dt;
        const uint32_t dbase = block_excl_scan(dt, scratch, tot);
        const uint32_t lbase = block_excl_scan(next - mine, scratch, tot);
        gbase[t] = dbase + mine;
        lexcl[t] = lbase;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wave_cnt[w][t] = 0;
            reinterpret_cast<unsigned long long *>(s_buf)[w * kRadix + t] = 0ull;
        }
    }
    __syncthreads();
    uint32_t *cnt = wave_cnt[wave];
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(s_buf) + wave * kRadix;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        rk[r] = wave_rank_round(mask, cnt, digit_of(k[r], dg), base + r * 64 + lane < n, lane);
    __syncthreads();
    {
        const uint32_t c0 = wave_cnt[0][t], c1 = wave_cnt[1][t], c2 = wave_cnt[2][t], o = lexcl[t];
        wave_cnt[0][t] = o; wave_cnt[1][t] = o + c0; wave_cnt[2][t] = o + c0 + c1; wave_cnt[3][t] = o + c0 + c1 + c2;
    }
    __syncthreads();
    // block-sorted slot of every item (kept in rk), its digit into the digit image, its key into the buffer
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        if (base + r * 64 + lane < n) {
            const uint32_t d = digit_of(k[r], dg);
            rk[r] = cnt[d] + rk[r];
            s_dig[rk[r]] = (uint8_t)d;
            s_buf[rk[r]] = k[r];
        }
    }
    __syncthreads();
    const int m = (int)min((long)kSortChunk, (long)n - (long)blk * kSortChunk);
    // (the destination of slot lp is recomputed in both rounds from the digit image -- three LDS reads -- rather than kept in
    // sixteen registers: the kernel has to stay within six waves' register budget to use the LDS it saves)
    if (keys_out) {
#pragma unroll 4
        for (int r = 0; r < kSortItems; ++r) {
            const int lp = r * kSortThreads + t;
            if (lp < m) {
                const uint32_t d = s_dig[lp];
                keys_out[gbase[d] + ((uint32_t)lp - lexcl[d])] = s_buf[lp];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        if (base + r * 64 + lane < n) s_buf[rk[r]] = v[r];
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < kSortItems; ++r) {
        const int lp = r * kSortThreads + t;
        if (lp < m) {
            const uint32_t d = s_dig[lp];
            const uint32_t pos = gbase[d] + ((uint32_t)lp - lexcl[d]);
            const uint32_t val = s_buf[lp];
            vals_out[pos] = val;
            if (pairs) {
                const uint2 pr = pairs[val];
                pair_x_out[pos] = pr.x;
                pair_y_out[pos] = pr.y;
            }
        }
    }
}

// blocks beyond which the compact scatter pays: radix_scatter_kernel's 38 KB fit four blocks a CU
constexpr int kCompactScatterBlocks = 1024;

// ---- exclusive scan of tiles_touched in depth order ------------------------------------------------
constexpr int kScanItems = 4;
constexpr int kScanChunk = kSortThreads * kScanItems;

__device__ __forceinline__ uint32_t scan_load(const int32_t *__restrict__ tiles_touched,
                                              const uint32_t *__restrict__ order, long i, int n) {
    return i < n ? (uint32_t)tiles_touched[order[i]] : 0u;
}


// `gathered` (the offsets buffer itself): the counts in depth order, so that the final pass streams them instead of
// repeating the random gather through `order`
__global__ __launch_bounds__(kSortThreads) void scan_reduce_kernel(int n, const int32_t *__restrict__ tiles_touched,
                                                                   const uint32_t *__restrict__ order,
                                                                   uint32_t *__restrict__ block_sums,
                                                                   uint32_t *__restrict__ gathered,
                                                                   const uint2 *__restrict__ tile_rects,
                                                                   uint32_t *__restrict__ rects) {
    __shared__ uint32_t scratch[4];
    long base = (long)blockIdx.x * kScanChunk + threadIdx.x * kScanItems;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        uint32_t c = 0;
        if (base + j < n) {
            if (!order) {  // the depth sort's last pass already left the counts in depth order (in `gathered`)
                c = gathered[base + j];
            } else if (rects) {   // fused binning: ONE 8-B gather brings the packed rectangle and the count into depth order
                const uint2 rc = tile_rects[order[base + j]];
                rects[base + j] = rc.x;
                c = rc.y;
            } else {
                c = (uint32_t)tiles_touched[order[base + j]];
            }
            if (order) gathered[base + j] = c;
        }
        s += c;
    }
    uint32_t total;
    block_excl_scan(s, scratch, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// ctl_host (optional; gsa_scan_tiles_sorted_report): the caller's pinned host slot -- this launch's ONE workgroup leaves the
// count and the key range ctl[1..2] (written by an earlier launch of the stream) there and releases the caller's token into
// word 3, as the bucketed depth ordering does: no 16-byte copy kernel and no event behind the scan.
__global__ __launch_bounds__(kSortThreads) void scan_blocks_kernel(int nblocks, uint32_t *__restrict__ block_sums,
                                                                   uint32_t *__restrict__ total_out,
                                                                   const uint32_t *__restrict__ ctl = nullptr,
                                                                   uint32_t *__restrict__ ctl_host = nullptr,
                                                                   uint32_t token = 0u) {
    __shared__ uint32_t scratch[4];
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += kSortThreads) {
        int i = base + threadIdx.x;
        uint32_t v = i < nblocks ? block_sums[i] : 0;
        uint32_t total;
        uint32_t ex = block_excl_scan(v, scratch, total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (ctl_host) {
            ctl_host[0] = carry;
            ctl_host[1] = ctl[1];
            ctl_host[2] = ctl[2];
            __threadfence_system();
            __hip_atomic_store(&ctl_host[3], token << 8, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// in place: offsets[] holds the depth-ordered counts on entry (scan_reduce_kernel), their exclusive prefix on exit
// block_owner (optional): the gaussian whose record list covers slot b * kSortChunk, for every record block b below
// n_owner_blocks -- the fused emit starts from it instead of searching `offsets`
__global__ __launch_bounds__(kSortThreads) void scan_final_kernel(int n, const uint32_t *__restrict__ block_sums,
                                                                  uint32_t *__restrict__ offsets,
                                                                  int32_t *__restrict__ block_owner, int n_owner_blocks,
                                                                  const uint32_t *__restrict__ m_total) {
    __shared__ uint32_t scratch[4];
    long base = (long)blockIdx.x * kScanChunk + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? offsets[base + j] : 0u;
        s += v[j];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan(s, scratch, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (base + j < n) {
            offsets[base + j] = ex;
            if (block_owner && v[j] > 0u) {   // slots [ex, ex + v): owner of every block boundary inside
                const uint32_t b0 = (ex + (uint32_t)kSortChunk - 1u) / (uint32_t)kSortChunk;
                const uint32_t b1 = min((ex + v[j] - 1u) / (uint32_t)kSortChunk, (uint32_t)n_owner_blocks - 1u);
                for (uint32_t b = b0; b <= b1 && (int)b < n_owner_blocks; ++b) block_owner[b] = (int32_t)(base + j);
                // one entry more: the LAST gaussian that has records (culled gaussians sort behind it)
                if (ex + v[j] == *m_total) block_owner[n_owner_blocks] = (int32_t)(base + j);
            }
        }
        ex += v[j];
    }
}

// ---- record emission ------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void tile_emit_kernel(
    int n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets,
    const float *__restrict__ splat, const int32_t *__restrict__ radii, int tile_w, int tile_h, int n_tiles,
    int cull_mode, uint32_t m_cap, const uint32_t *__restrict__ m_dev, uint32_t *__restrict__ isect_tile,
    uint32_t *__restrict__ isect_gid) {
    long k = (long)blockIdx.x * kSortThreads + threadIdx.x;
    if (k >= n) return;
    const uint32_t m_capacity = m_dev ? min(m_cap, *m_dev) : m_cap;
    uint32_t g = order[k];
    int rx = radii[2 * (long)g], ry = radii[2 * (long)g + 1];
    if (rx <= 0 || ry <= 0) return;
    uint32_t begin = offsets[k];
    uint32_t end = (k + 1 < n) ? offsets[k + 1] : m_capacity;  // == begin + tiles_touched[g]
    end = min(end, m_capacity);
    const float4 *row = reinterpret_cast<const float4 *>(splat + (long)g * GSA_SPLAT_STRIDE);
    float4 q0 = row[0], q1 = row[1];
    float mx = q0.x, my = q0.y, a = q0.z, b = q0.w, c = q1.x, opac = q1.y;
    float sig_max = logf(fmaxf(opac, GSA_ALPHA_THRESHOLD) * 255.0f);
    GsaTileRect tr = gsa_tile_rect(mx, my, rx, ry, tile_w, tile_h);
    uint32_t pos = begin;
    for (int ty = tr.y0; ty < tr.y1; ++ty)
        for (int tx = tr.x0; tx < tr.x1; ++tx) {
            const bool hit = cull_mode == 0 || gsa_quadrant_mask(mx, my, a, b, c, sig_max, tx, ty) != 0u;
            if (hit && pos < end) {
                isect_tile[pos] = (uint32_t)(ty * tile_w + tx);
                isect_gid[pos] = g;
                ++pos;
            }
        }
    // count and emit run the same arithmetic on the same stored floats; should they ever disagree the
    // unused slots become sentinel records that sort behind every real tile and are never composited
    for (; pos < end; ++pos) {
        isect_tile[pos] = (uint32_t)n_tiles;
        isect_gid[pos] = g;
    }
}

// Wave-cooperative emission for cull_mode 0 (every tile of the rectangle is a record): a wave owns 64
// depth-consecutive gaussians, whose records are one contiguous output range; its lanes walk that range 64
// slots at a time -- slot -> gaussian by a 6-step binary search over the 64 start offsets kept in LDS, slot ->
// tile by the rectangle arithmetic -- so every store instruction writes 256 contiguous bytes.
__global__ __launch_bounds__(kSortThreads) void tile_emit_coop_kernel(
    int n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets,
    const float *__restrict__ splat, const int32_t *__restrict__ radii, int tile_w, int tile_h,
    uint32_t m_cap, const uint32_t *__restrict__ m_dev, uint32_t *__restrict__ isect_tile,
    uint32_t *__restrict__ isect_gid) {
    __shared__ uint32_t s_start[4][64 + 1];
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    __shared__ uint32_t s_gid[4][64];
    __shared__ int s_x0[4][64], s_y0[4][64], s_w[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long k0 = (long)blockIdx.x * kSortThreads + w * 64;
    if (k0 >= n) return;
    const long k = k0 + lane;
    uint32_t g = 0, begin = m_total;
    int x0 = 0, y0 = 0, wd = 1;
    if (k < n) {
        g = order[k];
        begin = min(offsets[k], m_total);   // clamp: with a device-side count the buffers may be too small (caller re-runs)
        // the projection packed the tile rectangle into word 11 of the record (one gathered line per gaussian);
        // 0 = culled (zero record) or not packable -> recompute from the mean and the radii
        const uint32_t rect = __float_as_uint(splat[(long)g * GSA_SPLAT_STRIDE + 11]);
        if (rect != 0u) {
            x0 = (int)(rect & 1023u); y0 = (int)((rect >> 10) & 1023u); wd = max((int)(rect >> 20), 1);
        } else {
            const int rx = radii[2 * (long)g], ry = radii[2 * (long)g + 1];
            if (rx > 0 && ry > 0) {
                const float2 m2 = *reinterpret_cast<const float2 *>(splat + (long)g * GSA_SPLAT_STRIDE);
                GsaTileRect tr = gsa_tile_rect(m2.x, m2.y, rx, ry, tile_w, tile_h);
                x0 = tr.x0; y0 = tr.y0; wd = max(tr.x1 - tr.x0, 1);
            }
        }
    }
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(begin);
    const long k_end = min((long)n, k0 + 64);
    const uint32_t wave_end = k_end < n ? min(offsets[k_end], m_total) : m_total;   // uniform address
    s_start[w][lane] = begin - wave_base;
    if (lane == 0) s_start[w][64] = 0xFFFFFFFFu;
    s_gid[w][lane] = g; s_x0[w][lane] = x0; s_y0[w][lane] = y0; s_w[w][lane] = wd;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t total = __builtin_amdgcn_readfirstlane(wave_end) - wave_base;
    for (uint32_t s = lane; s < total; s += 64) {
        int l = 0;  // largest l with start[l] <= s (starts are non-decreasing; empty gaussians share a start)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (s_start[w][l + step] <= s) l += step;
        const uint32_t j = s - s_start[w][l];
        const int wl = s_w[w][l];
        const int row = (int)(((float)j + 0.5f) / (float)wl);  // exact: j, wl < 2^15
        const int col = (int)j - row * wl;
        isect_tile[wave_base + s] = (uint32_t)((s_y0[w][l] + row) * tile_w + s_x0[w][l] + col);
        isect_gid[wave_base + s] = s_gid[w][l];
    }
}

// tile_offsets[t] = first index i with sorted_tile[i] >= t,  t in [0, n_tiles]
// tile_offsets[t] = first index i with sorted_tile[i] >= t,  t in [0, n_tiles]
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void tile_offsets_kernel(uint32_t m_cap, const uint32_t *__restrict__ m_dev,
                                                                    const KeyT *__restrict__ sorted_tile,
                                                                    int n_tiles, int32_t *__restrict__ tile_offsets) {
    const uint32_t m = m_dev ? min(m_cap, *m_dev) : m_cap;
    long i = (long)blockIdx.x * kSortThreads + threadIdx.x;
    if (i > m) return;
    long lo = (i == 0) ? 0 : (long)sorted_tile[i - 1] + 1;
    long hi = (i == m) ? n_tiles : min((long)sorted_tile[i], (long)n_tiles);
    for (long t = lo; t <= hi; ++t) tile_offsets[t] = (int32_t)i;
}

// the same by searching, for long lists (M >> n_tiles: a handful of dependent loads per tile beat streaming all M keys).
// One WAVE per tile: a 64-ary search (each round the lanes probe 64 evenly spaced keys of the remaining range; the
// ballot of "key < tile" says which sixty-fourth holds the boundary) -- 4 dependent memory round trips for 8 M records
// where a per-thread binary search takes 23.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void tile_offsets_bsearch_kernel(uint32_t m_cap, const uint32_t *__restrict__ m_dev,
                                                                            const KeyT *__restrict__ sorted_tile,
                                                                            int n_tiles, int32_t *__restrict__ tile_offsets) {
    const uint32_t m = m_dev ? min(m_cap, *m_dev) : m_cap;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (kSortThreads / 64) + (threadIdx.x >> 6);   // wave-uniform
    if (t > n_tiles) return;
    uint32_t lo = 0, hi = m;   // the first index with key >= t lies in [lo, hi] (hi: "none below hi")
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint64_t probe = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1u;
        const bool below = probe < hi && (uint32_t)sorted_tile[probe] < (uint32_t)t;
        const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(below));   // monotone: the first c probes are below
        const uint64_t nlo = (uint64_t)lo + (uint64_t)c * step;                // the answer lies past probe c - 1 ...
        hi = (uint32_t)min((uint64_t)hi, nlo + step - 1u);                     // ... and at or before probe c
        lo = (uint32_t)min(nlo, (uint64_t)hi);
        if (step == 1u) break;   // probes were consecutive: lo is the answer
    }
    if (lane == 0) tile_offsets[t] = (int32_t)lo;
}

// ---- fused emit + first tile-sort pass ---------------------------------------------------------------
// cull_mode 0, tile grid below 1024 x 1024 and at most 65535 tiles (16-bit tile ids).  The records are never written
// in emission order: the first (low-digit) pass of the tile sort GENERATES them from the depth-ordered rectangles --
// once to count digits (emit_hist_kernel), once to place them (emit_scatter_kernel) -- which drops the emit kernel's
// 8 B/record store and the first pass's 8 B/record load, and the keys travel as 16-bit tile ids from there on.
//
// A block owns the kSortChunk record slots [slot0, slot0 + kSortChunk).  slot -> gaussian: the scan noted the owner of
// the block's first slot (block_owner); the gaussians whose list starts inside the block mark their start slot in LDS (atomicMax: gaussians without records share the start of the
// next one that has some, the last one wins), a max-scan carries the owner forward; slot -> tile: the rectangle
// arithmetic of tile_emit_coop_kernel.
struct EmitBlock { uint32_t slot0, live; int k_lo; };   // live: slots of this block below the record count
#ifndef GSA_EMIT_STAGE
#define GSA_EMIT_STAGE 1280
#endif
// gaussians per block whose (offset, rectangle, index, 1 / width) are staged in LDS.  A 4096-slot block holds ~1050
// gaussians when the records are those of 32 x 32 bins (3.9 per gaussian at the bench scene; 460 with 16 x 16 tiles):
// with 768 staged, a quarter of the records took the per-record global look-up below
constexpr int kEmitStage = GSA_EMIT_STAGE;

// the packed 16-B entry of a staged gaussian: ONE LDS read per record instead of three, and the division by the
// rectangle's width becomes a multiplication (exact for the operands in range: see emit_record)
__device__ __forceinline__ uint4 emit_entry(uint32_t off, uint32_t rect, uint32_t gid) {
    const int wl = max((int)(rect >> 20), 1);
    return make_uint4(off, rect, gid, __float_as_uint(__builtin_amdgcn_rcpf((float)wl)));
}

// (the histogram kernel of the direct path needs neither the gaussian index nor the stored reciprocal: 8-B entries leave
// room for its per-bin counters at four workgroups per CU)
__device__ __forceinline__ uint2 emit_entry_small(uint32_t off, uint32_t rect) { return make_uint2(off, rect); }
__device__ __forceinline__ uint4 emit_widen(uint4 e) { return e; }
__device__ __forceinline__ uint4 emit_widen(uint2 e) {
    return make_uint4(e.x, e.y, 0u, __float_as_uint(__builtin_amdgcn_rcpf((float)max((int)(e.y >> 20), 1))));
}
__device__ __forceinline__ void emit_store_entry(uint4 *s_g, uint32_t i, uint32_t off, uint32_t rect, uint32_t gid) {
    s_g[i] = emit_entry(off, rect, gid);
}
__device__ __forceinline__ void emit_store_entry(uint2 *s_g, uint32_t i, uint32_t off, uint32_t rect, uint32_t) {
    s_g[i] = emit_entry_small(off, rect);
}

// The block's gaussians [k_lo, k_hi] (owner of its first slot .. last one that can start inside it) and the global loads
// of the first kEmitStage of them (three per thread), ISSUED here and consumed in emit_owners: the caller puts its own
// prologue (histogram columns, block scans) between the two, under the loads' latency.
static_assert(kEmitStage % kSortThreads == 0, "whole strides of the block's threads are staged");
struct EmitLoads {
    int k_lo, k_hi;
    uint32_t off[kEmitStage / kSortThreads], rect[kEmitStage / kSortThreads], gid[kEmitStage / kSortThreads];
};
__device__ __forceinline__ EmitLoads emit_owner_loads(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rects,
                                                      const uint32_t *__restrict__ order,
                                                      const int32_t *__restrict__ block_owner, int n_owner_blocks,
                                                      uint32_t m_total, bool want_gid, unsigned bid, int n) {
    EmitLoads el;
    const uint32_t slot0 = bid * (uint32_t)kSortChunk;
    const uint32_t slot_end = slot0 + min((uint32_t)kSortChunk, m_total - slot0);
    // owner of the first slot (noted by the scan), and the last gaussian that can start inside this block: the owner
    // of the next block's first slot, or the last gaussian with records when this is the last live block
    // (clamped to [0, n): after a reported failure of the depth ordering the table may hold anything)
    el.k_lo = min(max(block_owner[bid], 0), n - 1);
    el.k_hi = min(max(slot_end < m_total ? block_owner[bid + 1] : block_owner[n_owner_blocks], el.k_lo), n - 1);
#pragma unroll
    for (int j = 0; j < kEmitStage / kSortThreads; ++j) {
        const int k = min(el.k_lo + j * kSortThreads + (int)threadIdx.x, el.k_hi);   // (clamped, not predicated)
        el.off[j] = offsets[k];
        el.rect[j] = rects[k];
        el.gid[j] = want_gid ? order[k] : 0u;
    }
    return el;
}

// s_g [kEmitStage]: the packed entries of the first kEmitStage gaussians of the block, read once (coalesced)
// while their starts are marked; owners beyond (a block of gaussians with fewer than four records each) are looked
// up in global memory.  want_gid: the histogram does not need the gaussian indices.
template <typename ENTRY>
__device__ __forceinline__ EmitBlock emit_owners(const EmitLoads &el, const uint32_t *__restrict__ offsets,
                                                  uint32_t m_total, uint32_t *__restrict__ s_own,
                                                  ENTRY *__restrict__ s_g, uint32_t *__restrict__ scratch, unsigned bid) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    EmitBlock eb;
    eb.slot0 = bid * (uint32_t)kSortChunk;
    eb.live = min((uint32_t)kSortChunk, m_total - eb.slot0);
    eb.k_lo = el.k_lo;
    const int k_hi = el.k_hi;
    uint4 *own4 = reinterpret_cast<uint4 *>(s_own) + t * (kSortItems / 4);
#pragma unroll
    for (int q = 0; q < kSortItems / 4; ++q) own4[q] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kEmitStage / kSortThreads; ++j) {
        const int k = eb.k_lo + j * kSortThreads + t;
        if (k <= k_hi) {
            const uint32_t i = (uint32_t)(k - eb.k_lo);
            emit_store_entry(s_g, i, el.off[j], el.rect[j], el.gid[j]);
            const uint32_t o = el.off[j] - eb.slot0;   // (gaussians without records at the block's end start at slot_end)
            if (i > 0u && o < eb.live) atomicMax(&s_own[o], i);
        }
    }
    for (int k = eb.k_lo + kEmitStage + t; k <= k_hi; k += kSortThreads) {   // (rare: more than kEmitStage gaussians)
        const uint32_t o = offsets[k] - eb.slot0;
        if (o < eb.live) atomicMax(&s_own[o], (uint32_t)(k - eb.k_lo));
    }
    __syncthreads();
    uint32_t v[kSortItems], run = 0;
#pragma unroll
    for (int q = 0; q < kSortItems / 4; ++q) {
        const uint4 x = own4[q];
        v[4 * q] = run = max(run, x.x); v[4 * q + 1] = run = max(run, x.y);
        v[4 * q + 2] = run = max(run, x.z); v[4 * q + 3] = run = max(run, x.w);
    }
    uint32_t inc = run;   // inclusive max-scan over the block's threads
#pragma unroll
    for (int d = 1; d < GSA_WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc = max(inc, o);
    }
    if (lane == 63) scratch[wave] = inc;
    uint32_t before = __shfl_up(inc, 1);
    if (lane == 0) before = 0;
    __syncthreads();
    for (int w = 0; w < wave; ++w) before = max(before, scratch[w]);
#pragma unroll
    for (int q = 0; q < kSortItems / 4; ++q)
        own4[q] = make_uint4(max(v[4 * q], before), max(v[4 * q + 1], before), max(v[4 * q + 2], before),
                             max(v[4 * q + 3], before));
    __syncthreads();
    return eb;
}

// the record of block slot `slot` (< eb.live): tile id and gaussian index
template <bool WANT_GID, typename ENTRY>
__device__ __forceinline__ void emit_record(const EmitBlock &eb, uint32_t slot, const uint32_t *__restrict__ s_own,
                                            const ENTRY *__restrict__ s_g, const uint32_t *__restrict__ offsets,
                                            const uint32_t *__restrict__ rects, const uint32_t *__restrict__ order,
                                            int tile_w, uint32_t &tile, uint32_t &gid, int *bin_x = nullptr,
                                            int *bin_y = nullptr) {
    const uint32_t i = s_own[slot];
    uint4 e;
    if (i < (uint32_t)kEmitStage) {
        e = emit_widen(s_g[i]);
    } else {   // (a block of gaussians with fewer than ~5 records each: the tail is looked up in global memory)
        const int k = eb.k_lo + (int)i;
        e = emit_entry(offsets[k], rects[k], WANT_GID ? order[k] : 0u);
    }
    if (WANT_GID) gid = e.z;
    const uint32_t j = eb.slot0 + slot - e.x;
    const int x0 = (int)(e.y & 1023u), y0 = (int)((e.y >> 10) & 1023u), wl = max((int)(e.y >> 20), 1);
    // row = floor(j / wl) as (int)((j + 0.5) * (1 / wl)): j, wl < 2^15, so (j + 0.5) / wl is at least 0.5 / wl away from
    // an integer -- a relative 2^-16 or more -- while v_rcp and the product are good to ~2^-22: exact
    const int row = (int)(((float)j + 0.5f) * __uint_as_float(e.w));
    const int col = (int)j - row * wl;
    tile = (uint32_t)((y0 + row) * tile_w + x0 + col);
    if (bin_x) { *bin_x = x0 + col; *bin_y = y0 + row; }
}

__global__ __launch_bounds__(kSortThreads) void emit_hist_kernel(int n, const uint32_t *__restrict__ order,
                                                                 const uint32_t *__restrict__ offsets,
                                                                 const uint32_t *__restrict__ rects,
                                                                 const int32_t *__restrict__ block_owner, int tile_w,
                                                                 uint32_t m_cap, const uint32_t *__restrict__ m_dev,
                                                                 int nblocks, uint32_t *__restrict__ hist,
                                                                 uint32_t low_mask) {
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t h[kRadix];
    __shared__ __attribute__((aligned(16))) uint32_t s_own[kSortChunk];
    __shared__ uint4 s_g[kEmitStage];
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    if ((uint64_t)sort_block_id() * kSortChunk >= m_total) {
        hist[(long)threadIdx.x * nblocks + sort_block_id()] = 0;
        return;
    }
    h[threadIdx.x] = 0;
    const EmitLoads el = emit_owner_loads(offsets, rects, order, block_owner, nblocks, m_total, false, sort_block_id(), n);
    const EmitBlock eb = emit_owners(el, offsets, m_total, s_own, s_g, scratch, sort_block_id());
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t slot = r * kSortThreads + threadIdx.x;
        if (slot < eb.live) {
            uint32_t tile, gid = 0;
            emit_record<false>(eb, slot, s_own, s_g, offsets, rects, order, tile_w, tile, gid);
            atomicAdd(&h[tile & low_mask], 1u);
        }
    }
    __syncthreads();
    hist[(long)threadIdx.x * nblocks + sort_block_id()] = h[threadIdx.x];
}

// radix_scatter_kernel (shift 0) with generated instead of loaded items; keys leave as 16-bit tile ids
__global__ __launch_bounds__(kSortThreads) void emit_scatter_kernel(
    int n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rects,
    const int32_t *__restrict__ block_owner, int tile_w, uint32_t m_cap, const uint32_t *__restrict__ m_dev,
    uint16_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int nblocks, const uint32_t *__restrict__ hist,
    const uint32_t *__restrict__ digit_total, uint32_t low_mask) {
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t gbase[kRadix];
    __shared__ uint32_t lexcl[kRadix];
    // two regions, each with two lives: the generation's (slot owners | staged gaussians) and the scatter's (values |
    // 16-bit keys + wave-private digit counters)
    __shared__ __attribute__((aligned(16))) uint32_t s_val[kSortChunk];
    constexpr int kPoolWords = 4 * kEmitStage > kSortChunk / 2 + 4 * kRadix ? 4 * kEmitStage : kSortChunk / 2 + 4 * kRadix;
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[kPoolWords];
    uint4 *s_g = reinterpret_cast<uint4 *>(s_pool);
    uint16_t *s_key = reinterpret_cast<uint16_t *>(s_pool);
    uint32_t(*wave_cnt)[kRadix] = reinterpret_cast<uint32_t(*)[kRadix]>(s_pool + kSortChunk / 2);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    if ((uint64_t)blockIdx.x * kSortChunk >= m_total) return;
    SORT_TRACE_BEGIN();
    // (chunk = workgroup index here, not sort_block_id(): with the XCD-wise assignment this kernel's scattered run
    // stores arrive from eight distant fronts of every digit's segment at once -- measured 41 -> 45 us)
    const EmitLoads el = emit_owner_loads(offsets, rects, order, block_owner, nblocks, m_total, true, blockIdx.x, n);   // (in flight
                                                                                                          // under the scans)
    {
        uint32_t tot, dt = digit_total[t];
        const uint32_t mine = hist[(long)t * nblocks + blockIdx.x];
        const uint32_t next = (int)blockIdx.x + 1 < nblocks ? hist[(long)t * nblocks + blockIdx.x + 1] : dt;
        const uint32_t dbase = block_excl_scan(dt, scratch, tot);
        const uint32_t lbase = block_excl_scan(next - mine, scratch, tot);
        gbase[t] = dbase + mine;
        lexcl[t] = lbase;
    }
    SORT_TRACE_MARK(0);   // histogram columns + block scans
    const EmitBlock eb = emit_owners(el, offsets, m_total, s_val, s_g, scratch, blockIdx.x);
    SORT_TRACE_MARK(1);   // slot owners
    const uint32_t wbase = wave * (kSortChunk / 4);
    uint32_t k[kSortItems], v[kSortItems], rk[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t slot = wbase + r * 64 + lane;
        k[r] = 0xFFFFu; v[r] = 0u;
        if (slot < eb.live) {
            emit_record<true>(eb, slot, s_val, s_g, offsets, rects, order, tile_w, k[r], v[r]);
            v[r] = min(v[r], (uint32_t)n - 1u);   // (the compositor gathers splat rows by this id, unchecked)
        }
    }
    __syncthreads();   // owners and staged data consumed: the regions start their second life
    SORT_TRACE_MARK(2);   // record generation
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        wave_cnt[w][t] = 0;
        reinterpret_cast<unsigned long long *>(s_val)[w * kRadix + t] = 0ull;   // peer masks, in the idle value area
    }
    __syncthreads();

    uint32_t *cnt = wave_cnt[wave];
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(s_val) + wave * kRadix;   // (zeroed above)
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        rk[r] = wave_rank_round(mask, cnt, k[r] & low_mask, wbase + r * 64 + lane < eb.live, lane);
    __syncthreads();
    SORT_TRACE_MARK(3);   // ranking
    {
        uint32_t c0 = wave_cnt[0][t], c1 = wave_cnt[1][t], c2 = wave_cnt[2][t];
        const uint32_t o = lexcl[t];
        wave_cnt[0][t] = o; wave_cnt[1][t] = o + c0; wave_cnt[2][t] = o + c0 + c1; wave_cnt[3][t] = o + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        if (wbase + r * 64 + lane < eb.live) {
            const uint32_t lp = cnt[k[r] & low_mask] + rk[r];
            s_key[lp] = (uint16_t)k[r];
            s_val[lp] = v[r];
        }
    }
    __syncthreads();
    SORT_TRACE_MARK(4);   // reorder through LDS
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t lp = r * kSortThreads + t;
        if (lp < eb.live) {
            const uint32_t key = s_key[lp];
            const uint32_t d = key & low_mask;
            const uint32_t pos = gbase[d] + (lp - lexcl[d]);
            keys_out[pos] = (uint16_t)key;
            vals_out[pos] = s_val[lp];
        }
    }
    SORT_TRACE_MARK(5);   // output stores (issue)
    SORT_TRACE_END(2);
}

// ---- direct binning: ONE generation pass places every record in its bin's list ---------------------------------------
// Bin grids of at most kBinsDirect bins (the 60 x 34 bins of 32 x 32 pixels at 1920 x 1080).  The two-pass tile sort
// above moves every record three times (generated into digit runs, read back for the second histogram, read and
// scattered again) and then searches the sorted ids for the list boundaries.  Here the per-block histogram is over the
// BIN ITSELF: emit_binhist leaves one dense row of counts per 4096-slot block (coalesced 8-KB rows), bin_colscan turns
// the columns into exclusive prefixes in place and -- its last workgroup -- the column totals into tile_offsets, and
// emit_place generates the records once more and stores each gaussian id straight at
//   tile_offsets[bin] + (records of that bin in the blocks in front) + (rank among the block's records of the bin).
// The stores are single words, but a bin's region is filled front to back by consecutive blocks, and consecutive
// blocks run on ONE XCD (live_block_id): the words of a line meet in that XCD's L2 and leave it as whole lines.
// (Measured and dropped: generating the records once -- a ranking kernel that leaves (bin, rank, id) per slot, then a
// placement without LDS that could run beside another stream's sweep: that placement pays two scattered loads per record
// for its bases, 28.9 us, and comes behind the stream's LDS-hungry kernels anyway)
constexpr int kBinsDirect = 2048;
#ifndef GSA_COL_BINS
#define GSA_COL_BINS 32   // bins (128 contiguous bytes of every row) per workgroup of bin_colscan: 8 -> 15.8 us, 16 -> 10.5, 32 -> 8.8, 64 -> 30.5 (32 workgroups)
#endif
constexpr int kColBins = GSA_COL_BINS, kColSegs = 1024 / kColBins, kColThreads = kColSegs * kColBins;
#ifndef GSA_PLACE_XCD
#define GSA_PLACE_XCD 1   // emit_binhist / emit_place: a workgroup's record block by live_block_id (1) or by its index (0: binning +22 us)
#endif

__global__ __launch_bounds__(kSortThreads) void emit_binhist_kernel(int n, const uint32_t *__restrict__ order,
                                                                    const uint32_t *__restrict__ offsets,
                                                                    const uint32_t *__restrict__ rects,
                                                                    const int32_t *__restrict__ block_owner, int tile_w,
                                                                    uint32_t m_cap, const uint32_t *__restrict__ m_dev,
                                                                    int nblocks, int nb, int n_tiles,
                                                                    uint32_t *__restrict__ mat,
                                                                    uint32_t *__restrict__ done_counter) {
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t h[kBinsDirect];
    __shared__ __attribute__((aligned(16))) uint32_t s_own[kSortChunk];
    __shared__ uint2 s_g[kEmitStage];
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    const unsigned live = (unsigned)(((uint64_t)m_total + kSortChunk - 1) / kSortChunk);
    const unsigned bid = GSA_PLACE_XCD ? live_block_id(live) : blockIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *done_counter = 0u;   // (bin_colscan's arrival counter; it runs behind this kernel)
    if (bid >= live) return;                                // (rows of blocks beyond the count are never read)
#pragma unroll
    for (int j = 0; j < kBinsDirect / kSortThreads; ++j) h[j * kSortThreads + threadIdx.x] = 0u;
    const EmitLoads el = emit_owner_loads(offsets, rects, order, block_owner, nblocks, m_total, false, bid, n);
    const EmitBlock eb = emit_owners(el, offsets, m_total, s_own, s_g, scratch, bid);   // (its barriers cover the zeroing)
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t slot = r * kSortThreads + threadIdx.x;
        if (slot < eb.live) {
            uint32_t tile, gid = 0;
            emit_record<false>(eb, slot, s_own, s_g, offsets, rects, order, tile_w, tile, gid);
            atomicAdd(&h[min(tile, (uint32_t)n_tiles - 1u)], 1u);   // (clamped: after a reported ordering failure
                                                                     // the rectangles may hold anything)
        }
    }
    __syncthreads();
    uint32_t *row = mat + (size_t)bid * nb;
    for (int i = threadIdx.x; i < nb; i += kSortThreads) row[i] = h[i];
}

// columns of the [live blocks][nb] count matrix -> exclusive prefixes over the blocks, in place; bin totals; the
// workgroup that arrives last scans the totals into tile_offsets[0 .. n_tiles].  Thread (segment s, bin j): the blocks
// are cut into kColSegs segments, the kColBins bins of a workgroup are contiguous bytes of every row.
// (The totals travel as agent-scope stores, the arrival is ONE relaxed agent-scope add per workgroup behind the storing
// lane's s_waitcnt, and the last workgroup reads them with agent-scope loads: no __threadfence -- an agent-scope release
// writes back the XCD's whole L2, and 128 of them made this kernel 79 us long.)
__global__ __launch_bounds__(kColThreads) void bin_colscan_kernel(uint32_t m_cap, const uint32_t *__restrict__ m_dev, int nb,
                                                                  int n_tiles, uint32_t *__restrict__ mat,
                                                                  uint32_t *__restrict__ totals,
                                                                  int32_t *__restrict__ tile_offsets,
                                                                  uint32_t *__restrict__ done_counter) {
    __shared__ uint32_t seg[kColSegs][kColBins];
    __shared__ uint32_t scratch[kColThreads / 64 + 2];
    __shared__ uint32_t s_last;
    const int t = threadIdx.x, s = t / kColBins, j = t % kColBins;
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    const int live = (int)(((uint64_t)m_total + kSortChunk - 1) / kSortChunk);
    const int seg_len = (live + kColSegs - 1) / kColSegs;
    const int r0 = min(s * seg_len, live), r1 = min(r0 + seg_len, live);
    const int bin = blockIdx.x * kColBins + j;                 // (< nb: the grid is nb / kColBins workgroups)
    uint32_t *col = mat + bin;
    // the first kKeep rows of the segment stay in registers between the two sweeps (all of them at the bench: 883 blocks, 28
    // rows a segment); what a longer segment has beyond them is read in chunks of eight, a chunk's loads in flight together,
    // and read again (out of L2) in the second sweep.  (All rows in chunks of 32 spilled: 8.8 -> 14.8 us.)
    constexpr int kKeep = kColBins, kMore = 8;
    uint32_t v[kKeep], sum = 0;
#pragma unroll
    for (int q = 0; q < kKeep; ++q) v[q] = r0 < r1 ? col[(size_t)min(r0 + q, r1 - 1) * nb] : 0u;   // (clamped: all in flight)
#pragma unroll
    for (int q = 0; q < kKeep; ++q) { v[q] = r0 + q < r1 ? v[q] : 0u; sum += v[q]; }
    for (int b = r0 + kKeep; b < r1; b += kMore) {
        uint32_t w[kMore];
#pragma unroll
        for (int q = 0; q < kMore; ++q) w[q] = col[(size_t)min(b + q, r1 - 1) * nb];
#pragma unroll
        for (int q = 0; q < kMore; ++q) sum += b + q < r1 ? w[q] : 0u;
    }
    seg[s][j] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll 8
    for (int q = 0; q < kColSegs; ++q) {
        const uint32_t c = seg[q][j];
        if (q < s) run += c;
        all += c;
    }
#pragma unroll
    for (int q = 0; q < kKeep; ++q) {
        if (r0 + q < r1) col[(size_t)(r0 + q) * nb] = run;
        run += v[q];
    }
    for (int b = r0 + kKeep; b < r1; b += kMore) {
        uint32_t w[kMore];
#pragma unroll
        for (int q = 0; q < kMore; ++q) w[q] = col[(size_t)min(b + q, r1 - 1) * nb];
#pragma unroll
        for (int q = 0; q < kMore; ++q) {
            if (b + q < r1) col[(size_t)(b + q) * nb] = run;
            run += b + q < r1 ? w[q] : 0u;
        }
    }
    if (t < kColBins) {   // (s == 0: lanes of wave 0)
        __hip_atomic_store(&totals[bin], all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0)
            s_last = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup: exclusive scan of the nb (<= 2048) totals, two per thread
    constexpr int kPerT = 2;
    uint32_t a[kPerT], mine = 0;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        const int i = kPerT * t + q;
        a[q] = i < nb ? __hip_atomic_load(&totals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        mine += a[q];
    }
    const int lane = t & 63, wave = t >> 6;
    const uint32_t inc = wave_incl_scan(mine, lane);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += scratch[w];
    uint32_t ex = off + inc - mine;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        const int i = kPerT * t + q;
        if (i <= n_tiles) tile_offsets[i] = (int32_t)ex;   // (bins at and beyond n_tiles hold no records: entry n_tiles is the count)
        ex += a[q];
    }
    if (t == 0) *done_counter = 0u;
}

// rank of this lane's record among the wave's records of the same bin so far (wave_rank_round for 11-bit ids): the lanes
// of a bin find one another through the peer word of the id's low 8 bits and three ballots over its high bits; the
// counters are 16-bit (a wave ranks 1024 records per block)
template <int BITS = 11>
__device__ __forceinline__ uint32_t wave_rank_round_bins(unsigned long long *mask, uint16_t *cnt, uint32_t bin, bool valid,
                                                         int lane) {
    const unsigned long long bit = 1ull << lane;
    const uint32_t d = bin & (kRadix - 1);
    if (valid) atomicOr(&mask[d], bit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long raw = valid ? mask[d] : 0ull;
    unsigned long long peers = raw;
#pragma unroll
    for (int b = 8; b < BITS; ++b) {
        const bool set = (bin >> b) & 1u;
        const unsigned long long bal = __ballot(valid && set);
        peers &= set ? bal : ~bal;
    }
    const uint32_t below = __popcll(peers & (bit - 1ull));
    const uint32_t seen = cnt[bin];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (valid && below == 0) cnt[bin] = (uint16_t)(seen + (uint32_t)__popcll(peers));
    if (valid && (raw & (bit - 1ull)) == 0ull) mask[d] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return seen + below;
}

// BOX_Q (0: none; 2 / 4: a bin is BOX_Q x BOX_Q quadrants of 8 x 8 pixels -- 16 x 16 tiles / 32 x 32 bins): the top byte of
// every stored gaussian id (ids are below 2^24) receives the part of the gaussian's QUADRANT BOX (gsa_project_fwd's qbox:
// the bounding box of its alpha >= 1/255 ellipse, in quadrants) that lies inside the record's bin -- a 4-bit mask of the
// columns and a 4-bit mask of the rows it covers; zero: none -- which a forward-compositor wave tests against its own quadrant
// before it gathers anything: of the records of a 32 x 32 bin 22 % reach a given quadrant, of those whose box covers it 78 %.
// About fifteen integer instructions per record, one 4-B load from a 4-MB array the records of a gaussian share (they are
// consecutive slots here), and NOT ONE STORE more: what r04 measured on the way -- exact 16-bit ellipse masks in an array
// of their own -- cost this kernel 0.021 ms for the second scattered store stream alone (3.6 M two-byte stores), whatever
// the masks cost to compute, and a pass over the PLACED records (round 3) 0.070 ms.
// Q = quadrants per bin side.  The masks have 4 bits: a 64 x 64 bin (Q = 8) gets its box in units of TWO quadrants
// (16 pixels) -- coarser, still a pre-test the forward sweep applies before it gathers anything.
template <int Q>
__device__ __forceinline__ uint32_t qbox_in_bin(uint32_t qb, int bx, int by) {
    constexpr int SH = Q > 4 ? 1 : 0, U = (Q >> SH) - 1;   // units per side - 1
    // The byte holds the box as two 4-bit MASKS -- bit c of the low nibble: the box covers column c of the bin's (up to) four
    // column units, bit r of the high nibble: row r -- so that a compositor wave tests "does it cover MY quadrant" with one
    // AND and one compare against a wave-uniform two-bit selector (first / last column and row in 2-bit fields cost that test
    // nine vector instructions per scanned entry, in a sweep that is bound by vector issue)
    if (qb == 0xFFFFFFFFu) return ((2u << U) - 1u) * 0x11u;   // unbounded: every column and row unit of the bin
    const int x0 = (int)(qb & 1023u) - Q * bx, y0 = (int)((qb >> 10) & 1023u) - Q * by;
    const int x1 = x0 + (int)((qb >> 20) & 63u), y1 = y0 + (int)(qb >> 26);
    const bool none = x1 < 0 || x0 > Q - 1 || y1 < 0 || y0 > Q - 1;
    const uint32_t c0 = (uint32_t)(max(x0, 0) >> SH), c1 = (uint32_t)(min(x1, Q - 1) >> SH);
    const uint32_t r0 = (uint32_t)(max(y0, 0) >> SH), r1 = (uint32_t)(min(y1, Q - 1) >> SH);
    const uint32_t cols = (2u << c1) - (1u << c0), rows = (2u << r1) - (1u << r0);   // bits c0 .. c1, r0 .. r1
    return none ? 0u : (cols & 15u) | ((rows & 15u) << 4);
}

template <int BOX_Q>
__global__ __launch_bounds__(kSortThreads, 4) void emit_place_kernel(
    int n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rects,
    const int32_t *__restrict__ block_owner, int tile_w, uint32_t m_cap, const uint32_t *__restrict__ m_dev,
    uint16_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int nblocks, int nb, int n_tiles,
    const uint32_t *__restrict__ mat, const int32_t *__restrict__ tile_offsets, const uint32_t *__restrict__ qbox,
    int32_t *__restrict__ sched_order, int sched_tiles, int sched_tile_w, int sched_shift) {
    __shared__ uint32_t scratch[4];
    // two regions, each with two lives: the generation's (slot owners | staged gaussians) and the placement's (peer
    // words + the bins' first slots | wave-private 16-bit bin counters)
    __shared__ __attribute__((aligned(16))) uint32_t s_val[kSortChunk];
    constexpr int kPoolWords = 4 * kEmitStage > 2 * kBinsDirect ? 4 * kEmitStage : 2 * kBinsDirect;
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[kPoolWords];
    static_assert(kSortChunk >= 2 * 4 * kRadix + kBinsDirect, "peer words and bin bases share the owners' region");
    uint4 *s_g = reinterpret_cast<uint4 *>(s_pool);
    uint16_t(*wave_cnt)[kBinsDirect] = reinterpret_cast<uint16_t(*)[kBinsDirect]>(s_pool);
    uint32_t *s_base = s_val + 2 * 4 * kRadix;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // sched_order (gsa_tile_bin16_scheduled): EIGHT workgroups behind the placement's -- one per XCD band -- write the
    // compositor's launch order from the lists' offsets (complete since the column scan) beside the placement, instead of a
    // launch of their own (gsa_tile_schedule: 5 us and a kernel boundary per view)
    if (blockIdx.x >= (unsigned)nblocks) {
        int *cnt = reinterpret_cast<int *>(s_val), *first = cnt + kSchedBuckets;
        const int band = (int)blockIdx.x - nblocks;
        if (sched_shift == 0) tile_schedule_band<0, kSortThreads>(band, sched_tiles, tile_offsets, sched_order, sched_tile_w, cnt, first);
        else if (sched_shift == 1) tile_schedule_band<1, kSortThreads>(band, sched_tiles, tile_offsets, sched_order, sched_tile_w, cnt, first);
        else tile_schedule_band<2, kSortThreads>(band, sched_tiles, tile_offsets, sched_order, sched_tile_w, cnt, first);
        return;
    }
    const uint32_t m_total = m_dev ? min(m_cap, *m_dev) : m_cap;
    const unsigned live = (unsigned)(((uint64_t)m_total + kSortChunk - 1) / kSortChunk);
    const unsigned bid = GSA_PLACE_XCD ? live_block_id(live) : blockIdx.x;
    if (bid >= live) return;
    SORT_TRACE_BEGIN();
    const EmitLoads el = emit_owner_loads(offsets, rects, order, block_owner, nblocks, m_total, true, bid, n);
    // (first slot of every bin for this block: in flight under the generation)
    uint32_t base_r[kBinsDirect / kSortThreads];
    const uint32_t *row = mat + (size_t)bid * nb;
#pragma unroll
    for (int j = 0; j < kBinsDirect / kSortThreads; ++j) {
        const int i = min(j * kSortThreads + t, nb - 1);
        base_r[j] = row[i] + (uint32_t)tile_offsets[min(i, n_tiles)];
    }
    const EmitBlock eb = emit_owners(el, offsets, m_total, s_val, s_g, scratch, bid);
    SORT_TRACE_MARK(0);   // loads + slot owners
    const uint32_t wbase = wave * (kSortChunk / 4);
    uint32_t k[kSortItems], v[kSortItems], rk[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t slot = wbase + r * 64 + lane;
        k[r] = 0u; v[r] = 0u;
        if (slot < eb.live) {
            emit_record<true>(eb, slot, s_val, s_g, offsets, rects, order, tile_w, k[r], v[r]);
            k[r] = min(k[r], (uint32_t)n_tiles - 1u);
            v[r] = min(v[r], (uint32_t)n - 1u);   // (the compositor gathers splat rows by this id, unchecked)
        }
    }
    if constexpr (BOX_Q != 0) {
        // (bin -> (x, y) by a reciprocal: (bin + 0.5) / tile_w is at least 0.5 / tile_w away from an integer, ids and widths
        // are below 2^11 -- exact.  Slots beyond the block's records hold id 0: a valid entry, never stored.)
        const float inv_tw = __builtin_amdgcn_rcpf((float)tile_w);
        uint32_t qb[kSortItems];
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) qb[r] = qbox[v[r]];
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            const int by = (int)(((float)k[r] + 0.5f) * inv_tw), bx = (int)k[r] - by * tile_w;
            v[r] = (v[r] & 0xFFFFFFu) | (qbox_in_bin<BOX_Q>(qb[r], bx, by) << 24);
        }
    }
    __syncthreads();   // owners and staged data consumed: the regions start their second life
    SORT_TRACE_MARK(1);   // record generation
#pragma unroll
    for (int w = 0; w < 4; ++w) reinterpret_cast<unsigned long long *>(s_val)[w * kRadix + t] = 0ull;
#pragma unroll
    for (int j = 0; j < kBinsDirect / kSortThreads; ++j) {
        s_base[j * kSortThreads + t] = base_r[j];
        reinterpret_cast<uint2 *>(s_pool)[j * kSortThreads + t] = make_uint2(0u, 0u);     // 4 x 2048 16-bit counters
    }
    __syncthreads();
    uint16_t *cnt = wave_cnt[wave];
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(s_val) + wave * kRadix;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        rk[r] = wave_rank_round_bins(mask, cnt, k[r], wbase + r * 64 + lane < eb.live, lane);
    __syncthreads();
    SORT_TRACE_MARK(2);   // bases to LDS + ranking
#pragma unroll
    for (int j = 0; j < kBinsDirect / kSortThreads; ++j) {   // per bin: first rank of each wave's records in the block
        const int b = j * kSortThreads + t;
        const uint32_t c0 = wave_cnt[0][b], c1 = wave_cnt[1][b], c2 = wave_cnt[2][b];
        wave_cnt[0][b] = 0; wave_cnt[1][b] = (uint16_t)c0; wave_cnt[2][b] = (uint16_t)(c0 + c1);
        wave_cnt[3][b] = (uint16_t)(c0 + c1 + c2);
    }
    __syncthreads();
    SORT_TRACE_MARK(3);   // the waves' first ranks
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        if (wbase + r * 64 + lane < eb.live) {
            const uint32_t pos = s_base[k[r]] + (uint32_t)cnt[k[r]] + rk[r];
            if (pos < m_total) {   // (always, unless the depth ordering in front failed and reported it)
                vals_out[pos] = v[r];
                if (keys_out) keys_out[pos] = (uint16_t)k[r];
            }
        }
    }
    SORT_TRACE_MARK(4);   // stores (issue)
    SORT_TRACE_END(2);
}

int sort_blocks(int n) { return gsa_div_up(n, kSortChunk); }

}  // namespace

extern "C" size_t gsa_sort_workspace_bytes(int n) {
    if (n < 0) n = 0;
    size_t nb_sort = (size_t)sort_blocks(n) + 1;
    size_t nb_scan = (size_t)gsa_div_up(n, kScanChunk) + 1;
    size_t words = kRadix * nb_sort + kRadix + nb_scan + 64;
    return words * sizeof(uint32_t);
}

static int sort_pairs_impl(int n, const uint32_t *n_dev, const uint32_t *keys_in, const uint32_t *vals_in,
                           uint32_t *keys_out, uint32_t *vals_out, uint32_t *keys_tmp, uint32_t *vals_tmp, int begin_bit,
                           int end_bit, void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) return GSA_OK;
    GSA_RETURN_IF(!keys_in || !keys_out || !vals_out || !keys_tmp || !vals_tmp || !workspace, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_sort_workspace_bytes(n), GSA_ERR_WORKSPACE_TOO_SMALL);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblocks = sort_blocks(n);
    uint32_t *hist = static_cast<uint32_t *>(workspace);
    uint32_t *digit_total = hist + (size_t)kRadix * nblocks;
    const int passes = (end_bit - begin_bit + 7) / 8;
    // ping-pong so that the last pass lands in *_out
    const uint32_t *kin = keys_in, *vin = vals_in;
    for (int p = 0; p < passes; ++p) {
        bool to_out = ((passes - 1 - p) % 2) == 0;
        uint32_t *kout = to_out ? keys_out : keys_tmp;
        uint32_t *vout = to_out ? vals_out : vals_tmp;
        int shift = begin_bit + 8 * p;
        radix_hist_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(n, n_dev, kin, RadixDigit{shift, 0u, 0, nullptr, nullptr, 0u}, nblocks, hist, nullptr);
        GSA_LAUNCH_CHECK();
        radix_scan_rows_kernel<<<kRadix, kSortThreads, 0, s>>>(nblocks, hist, digit_total);
        GSA_LAUNCH_CHECK();
        radix_scatter_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(n, n_dev, kin, vin, kout, vout, RadixDigit{shift, 0u, 0, nullptr, nullptr, 0u},
                                                                        nblocks, hist, digit_total);
        GSA_LAUNCH_CHECK();
        kin = kout;
        vin = vout;
    }
    return GSA_OK;
}

extern "C" int gsa_sort_pairs_u32(int n, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                                  uint32_t *vals_out, uint32_t *keys_tmp, uint32_t *vals_tmp, int begin_bit,
                                  int end_bit, void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    return sort_pairs_impl(n, nullptr, keys_in, vals_in, keys_out, vals_out, keys_tmp, vals_tmp, begin_bit, end_bit,
                           workspace, workspace_bytes, stream);
}

extern "C" int gsa_sort_pairs_u32_dev(int capacity, const uint32_t *n_dev, const uint32_t *keys_in,
                                      const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                                      uint32_t *keys_tmp, uint32_t *vals_tmp, int begin_bit, int end_bit,
                                      void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(!n_dev, GSA_ERR_INVALID_ARGUMENT);
    return sort_pairs_impl(capacity, n_dev, keys_in, vals_in, keys_out, vals_out, keys_tmp, vals_tmp, begin_bit,
                           end_bit, workspace, workspace_bytes, stream);
}

static int scan_tiles_impl(int n, const int32_t *tiles_touched, const uint32_t *order, const uint2 *tile_rects,
                           uint32_t *offsets, uint32_t *rects, int32_t *block_owner, int n_owner_blocks,
                           uint32_t *total, void *workspace, size_t workspace_bytes, gsa_stream_t stream,
                           const uint32_t *ctl = nullptr, uint32_t *ctl_host = nullptr, uint32_t token = 0u) {
    GSA_RETURN_IF(n < 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!total, GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(total, 0, sizeof(uint32_t), s);
        if (ctl_host) {   // (no kernel runs: the host slot -- host memory -- is filled right here)
            ctl_host[0] = 0u; ctl_host[1] = 0xFFFFFFFFu; ctl_host[2] = 0u;
            __atomic_store_n(&ctl_host[3], token << 8, __ATOMIC_RELEASE);
        }
        return (int)e;
    }
    GSA_RETURN_IF(!offsets || !workspace, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(order && (rects ? !tile_rects : !tiles_touched), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_sort_workspace_bytes(n), GSA_ERR_WORKSPACE_TOO_SMALL);
    uint32_t *block_sums = static_cast<uint32_t *>(workspace);
    int nblocks = gsa_div_up(n, kScanChunk);
    scan_reduce_kernel<<<nblocks, kSortThreads, 0, s>>>(n, tiles_touched, order, block_sums, offsets, tile_rects, rects);
    GSA_LAUNCH_CHECK();
    scan_blocks_kernel<<<1, kSortThreads, 0, s>>>(nblocks, block_sums, total, ctl, ctl_host, token);
    GSA_LAUNCH_CHECK();
    scan_final_kernel<<<nblocks, kSortThreads, 0, s>>>(n, block_sums, offsets, n_owner_blocks > 0 ? block_owner : nullptr,
                                                       n_owner_blocks, total);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_scan_tiles(int n, const int32_t *tiles_touched, const uint32_t *order, uint32_t *offsets,
                              uint32_t *total, void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    return scan_tiles_impl(n, tiles_touched, order, nullptr, offsets, nullptr, nullptr, 0, total, workspace,
                           workspace_bytes, stream);
}

extern "C" int gsa_scan_tiles_rect(int n, const uint32_t *order, const uint32_t *tile_rects, uint32_t *offsets,
                                   uint32_t *rects, int32_t *block_owner, int n_blocks, uint32_t *total,
                                   void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(n > 0 && (!tile_rects || !rects), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(tile_rects) & 7) != 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(n_blocks < 0 || (n_blocks > 0 && !block_owner), GSA_ERR_INVALID_ARGUMENT);
    return scan_tiles_impl(n, nullptr, order, reinterpret_cast<const uint2 *>(tile_rects), offsets, rects, block_owner,
                           n_blocks, total, workspace, workspace_bytes, stream);
}

extern "C" int gsa_scan_tiles_sorted(int n, uint32_t *counts_offsets, int32_t *block_owner, int n_blocks, uint32_t *total,
                                     void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(n_blocks < 0 || (n_blocks > 0 && !block_owner), GSA_ERR_INVALID_ARGUMENT);
    return scan_tiles_impl(n, nullptr, nullptr, nullptr, counts_offsets, nullptr, block_owner, n_blocks, total, workspace,
                           workspace_bytes, stream);
}

// (Measured and dropped in r04: LSD passes on 12-bit digits built from the direct binning's kernels -- a per-block row of
// 4096 counts, the column scan, every key stored straight at its place -- two passes instead of three for the 24 key bits
// the visible depths differ in.  At 5 M keys a placement pass took 158 us (+ 14 us histogram + 20 us column scan) against
// 61 + 20 + 6 us for an 8-bit pass: with 4096 bins per block of 4096 keys the per-bin work of a block -- 16 KB of bases, 32 KB
// of wave counters, the waves' first ranks -- is as large as the per-key work, and every key goes to a line of its own.
// C5 binning 0.54 -> 0.65 ms.)
extern "C" int gsa_scan_tiles_sorted_report(int n, uint32_t *counts_offsets, int32_t *block_owner, int n_blocks, uint32_t *ctl,
                                            uint32_t *ctl_host, uint32_t token, void *workspace, size_t workspace_bytes,
                                            gsa_stream_t stream) {
    GSA_RETURN_IF(n_blocks < 0 || (n_blocks > 0 && !block_owner) || !ctl || !ctl_host || token > 0xFFFFFFu,
                  GSA_ERR_INVALID_ARGUMENT);
    return scan_tiles_impl(n, nullptr, nullptr, nullptr, counts_offsets, nullptr, block_owner, n_blocks, ctl, workspace,
                           workspace_bytes, stream, ctl, ctl_host, token);
}

extern "C" size_t gsa_depth_order_workspace_bytes(int n) {
    if (n < 0) n = 0;
    // two key and two value ping-pong buffers, the per-block key ranges, the sort's histograms
    return (size_t)4 * n * sizeof(uint32_t) + (size_t)2 * ((size_t)sort_blocks(n) + 1) * sizeof(uint32_t) +
           gsa_sort_workspace_bytes(n);
}

extern "C" int gsa_depth_order(int n, const uint32_t *depth_keys, const uint32_t *tile_rects, int key_bits,
                               uint32_t *order, uint32_t *rects, uint32_t *counts, uint32_t *key_range, void *workspace,
                               size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || key_bits < 1 || key_bits > 32 || !key_range, GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(key_range, 0xFF, sizeof(uint32_t), s);
        if (e == hipSuccess) e = hipMemsetAsync(key_range + 1, 0, sizeof(uint32_t), s);
        return (int)e;
    }
    GSA_RETURN_IF(!depth_keys || !tile_rects || !order || !rects || !counts || !workspace, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(tile_rects) & 7) != 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_depth_order_workspace_bytes(n), GSA_ERR_WORKSPACE_TOO_SMALL);
    const int nblocks = sort_blocks(n);
    uint32_t *buf = static_cast<uint32_t *>(workspace);
    uint32_t *kbuf[2] = {buf, buf + n}, *vbuf[2] = {buf + 2 * (size_t)n, buf + 3 * (size_t)n};
    uint32_t *block_range = buf + 4 * (size_t)n;
    uint32_t *hist = block_range + 2 * ((size_t)nblocks + 1);
    uint32_t *digit_total = hist + (size_t)kRadix * nblocks;
    const int passes = (key_bits + 7) / 8;
    const uint32_t *kin = depth_keys, *vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool last = p == passes - 1;
        uint32_t *kout = last ? nullptr : kbuf[p & 1];
        uint32_t *vout = last ? order : vbuf[p & 1];
        radix_hist_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(n, nullptr, kin, RadixDigit{8 * p, 0u, 0, nullptr, nullptr, 0u}, nblocks, hist,
                                                                     p == 0 ? block_range : nullptr);
        GSA_LAUNCH_CHECK();
        radix_scan_rows_kernel<<<kRadix, kSortThreads, 0, s>>>(nblocks, hist, digit_total, p == 0 ? block_range : nullptr,
                                                               p == 0 ? key_range : nullptr);
        GSA_LAUNCH_CHECK();
        if (nblocks > kCompactScatterBlocks)   // (populations beyond ~4 M: more blocks than the 38-KB instance has slots)
            radix_scatter_compact_kernel<<<nblocks, kSortThreads, 0, s>>>(
                n, kin, vin, kout, vout, 8 * p, nblocks, hist, digit_total,
                last ? reinterpret_cast<const uint2 *>(tile_rects) : nullptr, last ? rects : nullptr, last ? counts : nullptr);
        else
            radix_scatter_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(
                n, nullptr, kin, vin, kout, vout, RadixDigit{8 * p, 0u, 0, nullptr, nullptr, 0u}, nblocks, hist, digit_total,
                last ? reinterpret_cast<const uint2 *>(tile_rects) : nullptr, last ? rects : nullptr, last ? counts : nullptr);
        GSA_LAUNCH_CHECK();
        kin = kout;
        vin = vout;
    }
    return GSA_OK;
}

// ---- depth order by buckets: one MSD pass + a local sort per bucket, fused with the scan ------------------------------
// The LSD depth sort costs three launches per 8 key bits, each at its latency floor for 1 M keys (~29 us per pass at
// C3).  With the range of the visible keys known beforehand (remembered per scene shape, checked after the fact), ONE
// global pass buckets the keys by depth range -- digit = ((key - key_min) * scale) >> 32, 255 buckets of equal key width
// (at most 2^16), the culled sentinel alone in bucket 255 -- and ONE launch finishes the job with a workgroup per bucket:
// it sorts its bucket in LDS on the key's offset inside the bucket (stable counting sort, 8 bits per pass, peer-mask
// ranking), gathers the packed tile rectangles and record counts of its gaussians, scans the counts, learns the
// records of the buckets in front of it from their published totals (one 8-byte {ready, count, last} granule per
// bucket, relaxed agent-scope atomics, bounded spin; bucket ids are handed out by an atomic ticket, so a waiting block's
// predecessors are always running or finished) and writes order / rects / offsets / block owners.
// A bucket larger than the LDS capacity, or a key outside the assumed range, is REPORTED (ctl[3] / the key range in
// ctl[1..2]); the caller then repeats with the LSD path.
constexpr int kLocalThreads = 1024, kLocalWaves = kLocalThreads / 64, kLocalCap = 8192;
constexpr unsigned long long kGranuleReady = 1ull << 63;
constexpr uint32_t kStatusOverflow = 1u, kStatusTimeout = 2u, kStatusWide = 4u;

struct BucketLds {
    uint32_t val[2][kLocalCap];           // gaussian ids, ping-pong; after the sort the idle half holds the record counts
    uint16_t key[2][kLocalCap];           // the key's offset inside its bucket (wide buckets: its bits 8 .. 23)
    uint8_t low[kLocalCap];               // wide buckets (more than 65536 keys wide): bits 0 .. 7 of the offset -- the first local pass's digit
    uint16_t rk[kLocalCap];               // rank of an item among the items of its digit in its wave's range
    uint32_t wcnt[kLocalWaves][kRadix];   // per wave range and digit: count, then first slot
    uint32_t dbase[kRadix];
    uint32_t scratch[kLocalWaves + 2];
};

__device__ __forceinline__ uint32_t block1024_excl_scan(uint32_t v, uint32_t *scratch, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t off = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kLocalWaves; ++w) {
        const uint32_t c = scratch[w];
        if (w < wave) off += c;
        sum += c;
    }
    total = sum;
    __syncthreads();
    return off + inc - v;
}

__device__ __forceinline__ uint32_t block1024_max(uint32_t v, uint32_t *scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < kLocalWaves; ++w) m = max(m, scratch[w]);
    __syncthreads();
    return m;
}

// one stable counting-sort pass over the mm items of L.key/val[src] on key bits [bs, bs + 8) into [src ^ 1]
// (bs < 0: on the byte L.low[i] of every item -- the first pass of a wide bucket, over items still in load order)
__device__ __forceinline__ void bucket_local_pass(BucketLds &L, int src, int mm, int bs) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int R = ((mm + kLocalThreads - 1) / kLocalThreads) * 64;   // items per wave range (a multiple of 64)
    // (the destination half of the value buffer is idle until the placement: it holds the 16 waves' peer masks)
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(L.val[src ^ 1]);
    static_assert(sizeof(L.val[0]) >= sizeof(unsigned long long) * kLocalWaves * kRadix, "peer masks must fit");
    for (int i = t; i < kLocalWaves * kRadix; i += kLocalThreads) {
        (&L.wcnt[0][0])[i] = 0u;
        masks[i] = 0ull;
    }
    __syncthreads();
    {
        uint32_t *cnt = L.wcnt[wave];
        unsigned long long *mask = masks + wave * kRadix;
        const int end = min(mm, (wave + 1) * R);
        for (int i0 = wave * R; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < end;
            const uint32_t d = !valid ? 0u : bs < 0 ? (uint32_t)L.low[i] : ((uint32_t)L.key[src][i] >> bs) & (kRadix - 1);
            const uint32_t rank = wave_rank_round(mask, cnt, d, valid, lane);
            if (valid) L.rk[i] = (uint16_t)rank;
        }
    }
    __syncthreads();
    uint32_t total = 0;
    if (t < kRadix) {   // digit t: first slot of every wave range's items inside the digit; digit total
#pragma unroll
        for (int w = 0; w < kLocalWaves; ++w) {
            const uint32_t c = L.wcnt[w][t];
            L.wcnt[w][t] = total;
            total += c;
        }
    }
    uint32_t all;
    const uint32_t ex = block1024_excl_scan(t < kRadix ? total : 0u, L.scratch, all);   // (threads >= 256 add zeros behind)
    if (t < kRadix) L.dbase[t] = ex;
    __syncthreads();
    for (int i = t; i < mm; i += kLocalThreads) {
        const uint32_t k = L.key[src][i];
        const uint32_t d = bs < 0 ? (uint32_t)L.low[i] : (k >> bs) & (kRadix - 1);
        const uint32_t pos = L.dbase[d] + L.wcnt[i / R][d] + L.rk[i];
        L.key[src ^ 1][pos] = (uint16_t)k;
        L.val[src ^ 1][pos] = L.val[src][i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(kLocalThreads) void depth_bucket_sort_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, const uint32_t *__restrict__ rects_in,
    const uint32_t *__restrict__ counts_in, const uint32_t *__restrict__ digit_total,
    uint32_t key_min, uint32_t scale, uint32_t *__restrict__ order,
    uint32_t *__restrict__ rects, uint32_t *__restrict__ offsets, int32_t *__restrict__ block_owner, int n_owner_blocks,
    unsigned long long *granule, uint32_t *ticket, uint32_t *__restrict__ ctl, uint32_t *__restrict__ ctl_host,
    uint32_t token, const uint32_t *__restrict__ table) {
    __shared__ BucketLds L;
    __shared__ uint32_t s_bucket;
    const int t = threadIdx.x;
    SORT_TRACE_BEGIN();
    const uint32_t dt = t < kRadix ? digit_total[t] : 0u;   // (in flight beside the ticket, not behind its barrier)
    if (t == 0) s_bucket = atomicAdd(ticket, 1u);
    __syncthreads();
    const int d = (int)s_bucket;                       // this block's bucket, in order of arrival
    // bucket d = items [lo, lo + m) of the first pass's output
    uint32_t all;
    const uint32_t lo_t = block1024_excl_scan(dt, L.scratch, all);   // (thread t: start of bucket t)
    if (t == d) { L.dbase[0] = lo_t; L.dbase[1] = dt; }
    __syncthreads();
    const uint32_t lo = L.dbase[0], m = L.dbase[1];
    __syncthreads();
    const bool culled_bucket = d == kRadix - 1;        // their entries of order / rects / offsets are never read
    const int mm = culled_bucket ? 0 : (int)min(m, (uint32_t)kLocalCap);
    if (!culled_bucket && m > (uint32_t)kLocalCap && t == 0) {
        atomicOr(&ctl[3], kStatusOverflow);
        __threadfence();   // (ahead of this block's granule: the last block reads the status after it has seen them all)
    }

    int cur = 0;
    // bucket d holds the keys x = key - key_min with (x * scale) >> 32 == d: x in [lo_d, lo_(d+1)), lo_d = ceil(d 2^32 /
    // scale); the local sorts run on x - lo_d (< 2^16 by the caller's choice of scale; the last bucket is clamped, a key
    // beyond the assumed range wraps there and the caller, who checks the measured range, repeats)
    // (with a depth table: bucket d = the cells [cs[d], cs[d + 1]) of 2^shift keys each, counted from cell g0)
    uint32_t sub, width;
    if (table) {
        const uint32_t c0 = table[kTabCs + min(d, 255)], c1 = table[kTabCs + min(d + 1, 256)], sh = table[kTabPlan + 1];
        sub = (table[kTabPlan] + c0) << sh;
        width = (c1 - c0) << sh;
    } else {
        const uint32_t lo_d = (uint32_t)((((unsigned long long)d << 32) + scale - 1u) / scale);
        width = (uint32_t)((((unsigned long long)(d + 1) << 32) + scale - 1u) / scale) - lo_d;
        sub = key_min + lo_d;
    }
    // a bucket wider than 65536 keys (a view whose visible depths span more than ~2 octaves) sorts on 24 bits: the low
    // byte of the offset goes to L.low and is the first pass's digit, bits 8 .. 23 are the 16-bit key of the other two
    const bool wide = width > 65536u;                  // (uniform)
    // (loads in groups of four strides with CLAMPED indices, all of a group in flight at once: a plain strided loop
    // waits for every load before it issues the next -- one memory round trip per 1024 items)
    constexpr int kStrides = kLocalCap / kLocalThreads;
    static_assert(kStrides % 4 == 0, "groups of four strides");
#pragma unroll
    for (int g4 = 0; g4 < kStrides; g4 += 4) {
        if (g4 * kLocalThreads >= mm) break;   // (uniform)
        uint32_t kk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min((g4 + j) * kLocalThreads + t, mm - 1);
            kk[j] = keys_in[lo + i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (g4 + j) * kLocalThreads + t;
            if (i < mm) {
                const uint32_t x = kk[j] - sub;        // (< width <= 2^24, unless a key lies outside the planned range: reported)
                L.key[0][i] = (uint16_t)((wide ? x >> 8 : x) & 0xFFFFu);
                if (wide) L.low[i] = (uint8_t)x;
                // (r05) the local sorts carry the item's PLACE in the bucket, not its gaussian id: id, rectangle and record
                // count then come out of the bucket's own 32-KB windows of the first pass's output (which brought the
                // rectangles along) instead of one random 8-byte gather per gaussian over the whole scene
                L.val[0][i] = (uint32_t)i;
            }
        }
    }
    __syncthreads();
    SORT_TRACE_MARK(0);   // ticket, bucket bounds, key / value loads
    const uint32_t width16 = wide ? (width + 255u) >> 8 : width;   // distinct values of the 16-bit key
    if (mm > 0 && wide) { bucket_local_pass(L, cur, mm, -1); cur ^= 1; }
    if (mm > 0 && width16 > 1u) { bucket_local_pass(L, cur, mm, 0); cur ^= 1; }   // (mm, width: uniform)
    if (mm > 0 && width16 > 256u) { bucket_local_pass(L, cur, mm, 8); cur ^= 1; }

    SORT_TRACE_MARK(1);   // local passes
    // rectangles and record counts of the bucket's gaussians, now in depth order
    uint32_t *cnt = L.val[cur ^ 1];
#pragma unroll
    for (int g4 = 0; g4 < kStrides; g4 += 4) {   // (the random 8-B gathers: four in flight per thread, as above)
        if (g4 * kLocalThreads >= mm) break;
        uint32_t gg[4];
        uint2 pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = L.val[cur][min((g4 + j) * kLocalThreads + t, mm - 1)];   // where the item sat in the bucket
            gg[j] = vals_in[lo + p];
            pr[j] = make_uint2(rects_in[lo + p], counts_in[lo + p]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (g4 + j) * kLocalThreads + t;
            if (i < mm) {
                order[lo + i] = gg[j];
                rects[lo + i] = pr[j].x;
                cnt[i] = pr[j].y;
            }
        }
    }
    __syncthreads();
    SORT_TRACE_MARK(2);   // gathers + stores of order / rects
    // exclusive scan of the counts: 8 consecutive items per thread
    constexpr int kPer = kStrides;
    uint32_t c[kPer], sum = 0, last = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = t * kPer + j;
        c[j] = i < mm ? cnt[i] : 0u;
        sum += c[j];
        if (c[j]) last = (uint32_t)i + 1u;
    }
    uint32_t bucket_total;
    uint32_t ex = block1024_excl_scan(sum, L.scratch, bucket_total);
    const uint32_t last_local = block1024_max(last, L.scratch);   // 1 + index of the bucket's last gaussian with records
    if (t == 0)
        __hip_atomic_store(&granule[d], kGranuleReady | ((unsigned long long)bucket_total << 32) | last_local,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SORT_TRACE_MARK(3);   // scan of the counts, granule published
    // records of the buckets in front: thread t < d waits for bucket t's granule (bounded)
    uint32_t before = 0, last_k = 0;
    if (t < d) {
        unsigned long long gr = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            gr = __hip_atomic_load(&granule[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gr & kGranuleReady) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (!(gr & kGranuleReady)) atomicOr(&ctl[3], kStatusTimeout);
        before = (uint32_t)((gr >> 32) & 0x7FFFFFFFu);
        const uint32_t lk = (uint32_t)gr;
        if (lk) last_k = lo_t + lk;                    // 1 + global index (depth order) of bucket t's last owner
    }
    uint32_t base_total;
    block1024_excl_scan(before, L.scratch, base_total);
    const uint32_t base = base_total;                  // records in front of this bucket
    SORT_TRACE_MARK(4);   // wait for the buckets in front
    if (culled_bucket) {                               // the last bucket: it has seen every other one
        const uint32_t lk = block1024_max(last_k, L.scratch);
        if (t == 0) {
            ctl[0] = base;                             // M
            if (block_owner && n_owner_blocks > 0 && lk) block_owner[n_owner_blocks] = (int32_t)(lk - 1u);
            // The control words go straight to the caller's pinned host slot (no copy kernel, no event on the stream):
            // count and key range first, then -- released at system scope -- the status tagged with the caller's token,
            // which the host polls for.  (A block can only time out on a granule this block waits for as well, so the
            // status read here is final.)
            if (table && (table[kTabPlan + 3] & kPlanWide)) atomicOr(&ctl[3], kStatusWide);   // the range was too wide for 16-bit local sorts
            if (ctl_host) {
                __threadfence();
                const uint32_t st = __hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ctl_host[0] = base;
                ctl_host[1] = ctl[1];
                ctl_host[2] = ctl[2];
                __threadfence_system();
                __hip_atomic_store(&ctl_host[3], (st & 0xFFu) | (token << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        SORT_TRACE_END(3);
        return;
    }
    ex += base;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = t * kPer + j;
        if (i < mm) {
            offsets[lo + i] = ex;
            if (block_owner && n_owner_blocks > 0 && c[j] > 0u) {   // owner of every record-block boundary inside [ex, ex + c)
                const uint32_t b0 = (ex + (uint32_t)kSortChunk - 1u) / (uint32_t)kSortChunk;
                const uint32_t b1 = min((ex + c[j] - 1u) / (uint32_t)kSortChunk, (uint32_t)n_owner_blocks - 1u);
                for (uint32_t b = b0; b <= b1 && (int)b < n_owner_blocks; ++b) block_owner[b] = (int32_t)(lo + i);
            }
        }
        ex += c[j];
    }
    // A bucket beyond the LDS capacity (reported through ctl[3]; the caller repeats with the LSD path) still leaves every
    // entry of its range DEFINED: the kernels queued behind this one run before the host has seen the report, and they
    // must not chase recycled memory.  The tail entries own no records (count 0 at the bucket's end).
    for (uint32_t i = (uint32_t)mm + (uint32_t)t; i < m; i += kLocalThreads) {
        order[lo + i] = 0u;
        rects[lo + i] = 0u;
        offsets[lo + i] = base + bucket_total;
    }
    SORT_TRACE_MARK(5);   // offsets + block owners
    SORT_TRACE_END(3);
}

extern "C" size_t gsa_depth_order_msd_workspace_bytes(int n) {
    if (n < 0) n = 0;
    // first-pass output (keys, values), per-block key ranges, histograms, granules + ticket, the depth table -- and, behind
    // those, the rectangles and record counts the first pass brings into bucket order (r05)
    return (size_t)2 * n * sizeof(uint32_t) + (size_t)2 * ((size_t)sort_blocks(n) + 1) * sizeof(uint32_t) +
           gsa_sort_workspace_bytes(n) + (size_t)(2 * kRadix + 16 + kTabWords) * sizeof(uint32_t) + 16 +
           (size_t)2 * n * sizeof(uint32_t) + 16;
}

extern "C" int gsa_depth_order_msd_max_width(void) { return 1 << kLocalBits; }

// where gsa_depth_order_msd(key_range_dev) leaves the view's depth table inside its workspace (diagnostics and tests):
// byte offset from a 16-byte-aligned workspace base; gsa_depth_order_msd_table_words() uint32 -- bytes [0, 4096): bucket of
// every cell; words [1024, 1281): first cell of every bucket, then the end; words 1284..1287: g0, shift, cells, flags
extern "C" size_t gsa_depth_order_msd_table_offset(int n) {
    if (n < 0) n = 0;
    const size_t nblocks = (size_t)sort_blocks(n);
    size_t words = (size_t)2 * n + 2 * (nblocks + 1) + (size_t)kRadix * nblocks + kRadix;
    size_t bytes = (words * sizeof(uint32_t) + 15) & ~(size_t)15;
    return bytes + (size_t)kRadix * sizeof(unsigned long long) + 16 * sizeof(uint32_t);
}
extern "C" int gsa_depth_order_msd_table_words(void) { return kTabWords; }

extern "C" int gsa_depth_order_msd(int n, const uint32_t *depth_keys, const uint32_t *tile_rects, uint32_t key_min,
                                   uint32_t scale, const uint32_t *key_range_dev, uint32_t *order, uint32_t *rects,
                                   uint32_t *offsets, int32_t *block_owner,
                                   int n_blocks, uint32_t *ctl, uint32_t *ctl_host, uint32_t token, void *workspace,
                                   size_t workspace_bytes, gsa_stream_t stream) {
    // (scale > 2^8: a bucket is at most 2^32 / scale + 1 <= 2^24 keys wide -- what the 24-bit local sorts cover)
    GSA_RETURN_IF(n < 0 || (!key_range_dev && scale <= 256u) || !ctl || n_blocks < 0 ||
                  (n_blocks > 0 && !block_owner), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(key_range_dev) & 15) != 0, GSA_ERR_INVALID_ARGUMENT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    GSA_RETURN_IF(token > 0xFFFFFFu, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(ctl, 0, 4 * sizeof(uint32_t), s);
        if (e == hipSuccess) e = hipMemsetAsync(ctl + 1, 0xFF, sizeof(uint32_t), s);
        if (ctl_host) {   // (no kernel runs: the host slot -- host memory -- is filled right here)
            ctl_host[0] = 0u; ctl_host[1] = 0xFFFFFFFFu; ctl_host[2] = 0u;
            __atomic_store_n(&ctl_host[3], token << 8, __ATOMIC_RELEASE);
        }
        return (int)e;
    }
    GSA_RETURN_IF(!depth_keys || !tile_rects || !order || !rects || !offsets || !workspace, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF((reinterpret_cast<uintptr_t>(tile_rects) & 7) != 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_depth_order_msd_workspace_bytes(n), GSA_ERR_WORKSPACE_TOO_SMALL);
    const int nblocks = sort_blocks(n);
    uint32_t *buf = static_cast<uint32_t *>(workspace);
    uint32_t *kbuf = buf, *vbuf = buf + n;
    uint32_t *block_range = buf + 2 * (size_t)n;
    uint32_t *hist = block_range + 2 * ((size_t)nblocks + 1);
    uint32_t *digit_total = hist + (size_t)kRadix * nblocks;
    uintptr_t g = reinterpret_cast<uintptr_t>(digit_total + kRadix);
    g = (g + 15) & ~(uintptr_t)15;
    unsigned long long *granule = reinterpret_cast<unsigned long long *>(g);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(granule + kRadix);
    // key_range_dev: what gsa_project_fwd left per wave -> one workgroup builds this view's depth table behind the ticket
    // for the launches that follow (the row scan zeroes only granules, ticket and ctl[3])
    uint32_t *table = key_range_dev ? ticket + 16 : nullptr;
    // (behind everything else: the packed rectangles and record counts in the first pass's output order)
    uint32_t *rbuf = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(ticket + 16 + kTabWords) + 15) & ~(uintptr_t)15);
    uint32_t *cbuf = rbuf + n;
    const RadixDigit dg{(int)scale, key_min, table ? 2 : 1, table, nullptr, 0u};
    if (key_range_dev) {
        depth_table_kernel<<<1, kTabThreads, 0, s>>>(reinterpret_cast<const uint4 *>(key_range_dev), gsa_div_up(n, 64), table);
        GSA_LAUNCH_CHECK();
    }
    radix_hist_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(n, nullptr, depth_keys, dg, nblocks, hist, block_range);
    GSA_LAUNCH_CHECK();
    // (block 0 also reduces the key range into ctl[1..2]; block 1 zeroes the granules, the ticket and ctl[3])
    radix_scan_rows_kernel<<<kRadix, kSortThreads, 0, s>>>(nblocks, hist, digit_total, block_range, ctl + 1,
                                                           reinterpret_cast<uint32_t *>(granule), 2 * kRadix + 4);
    GSA_LAUNCH_CHECK();
    radix_scatter_kernel<uint32_t><<<nblocks, kSortThreads, 0, s>>>(n, nullptr, depth_keys, nullptr, kbuf, vbuf, dg, nblocks,
                                                                    hist, digit_total,
                                                                    reinterpret_cast<const uint2 *>(tile_rects), rbuf, cbuf);
    GSA_LAUNCH_CHECK();
    depth_bucket_sort_kernel<<<kRadix, kLocalThreads, 0, s>>>(kbuf, vbuf, rbuf, cbuf, digit_total, key_min, scale, order, rects,
                                                              offsets, n_blocks > 0 ? block_owner : nullptr, n_blocks,
                                                              granule, ticket, ctl, ctl_host, token, table);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

static int tile_emit_impl(int n, const uint32_t *order, const uint32_t *offsets, const float *splat,
                          const int32_t *radii, int width, int height, int cull_mode, uint32_t m_capacity,
                          const uint32_t *m_dev, uint32_t *isect_tile, uint32_t *isect_gid, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || width <= 0 || height <= 0, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(cull_mode != 0 && cull_mode != 1, GSA_ERR_INVALID_ARGUMENT);
    if (n == 0 || m_capacity == 0) return GSA_OK;
    GSA_RETURN_IF(!order || !offsets || !splat || !radii || !isect_tile || !isect_gid, GSA_ERR_INVALID_ARGUMENT);
    int tile_w = gsa_div_up(width, GSA_TILE_SIZE), tile_h = gsa_div_up(height, GSA_TILE_SIZE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cull_mode == 0 && tile_w < (1 << 15) && tile_h < (1 << 15))
        tile_emit_coop_kernel<<<gsa_div_up(n, kSortThreads), kSortThreads, 0, s>>>(
            n, order, offsets, splat, radii, tile_w, tile_h, m_capacity, m_dev, isect_tile, isect_gid);
    else
        tile_emit_kernel<<<gsa_div_up(n, kSortThreads), kSortThreads, 0, s>>>(
            n, order, offsets, splat, radii, tile_w, tile_h, tile_w * tile_h, cull_mode, m_capacity, m_dev, isect_tile,
            isect_gid);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_tile_emit(int n, const uint32_t *order, const uint32_t *offsets, const float *splat,
                             const int32_t *radii, int width, int height, int cull_mode, uint32_t m_capacity,
                             uint32_t *isect_tile, uint32_t *isect_gid, gsa_stream_t stream) {
    return tile_emit_impl(n, order, offsets, splat, radii, width, height, cull_mode, m_capacity, nullptr, isect_tile,
                          isect_gid, stream);
}

extern "C" int gsa_tile_emit_dev(int n, const uint32_t *order, const uint32_t *offsets, const float *splat,
                                 const int32_t *radii, int width, int height, int cull_mode, uint32_t m_capacity,
                                 const uint32_t *m_dev, uint32_t *isect_tile, uint32_t *isect_gid,
                                 gsa_stream_t stream) {
    GSA_RETURN_IF(!m_dev, GSA_ERR_INVALID_ARGUMENT);
    return tile_emit_impl(n, order, offsets, splat, radii, width, height, cull_mode, m_capacity, m_dev, isect_tile,
                          isect_gid, stream);
}

template <typename KeyT>
static int tile_offsets_impl(uint32_t m, const uint32_t *m_dev, const KeyT *sorted_tile, int n_tiles,
                             int32_t *tile_offsets, gsa_stream_t stream) {
    GSA_RETURN_IF(n_tiles <= 0 || !tile_offsets, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(m > 0 && !sorted_tile, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(m >= 0x7FFFFFFFu, GSA_ERR_UNSUPPORTED);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((long long)m > 1024LL * n_tiles)
        tile_offsets_bsearch_kernel<KeyT><<<gsa_div_up(n_tiles + 1, kSortThreads / 64), kSortThreads, 0, s>>>(
            m, m_dev, sorted_tile, n_tiles, tile_offsets);
    else
        tile_offsets_kernel<KeyT><<<gsa_div_up((long long)m + 1, kSortThreads), kSortThreads, 0, s>>>(
            m, m_dev, sorted_tile, n_tiles, tile_offsets);
    GSA_LAUNCH_CHECK();
    return GSA_OK;
}

extern "C" int gsa_tile_offsets(uint32_t m, const uint32_t *sorted_tile, int n_tiles, int32_t *tile_offsets,
                                gsa_stream_t stream) {
    return tile_offsets_impl<uint32_t>(m, nullptr, sorted_tile, n_tiles, tile_offsets, stream);
}

extern "C" int gsa_tile_offsets_dev(uint32_t m_capacity, const uint32_t *m_dev, const uint32_t *sorted_tile,
                                    int n_tiles, int32_t *tile_offsets, gsa_stream_t stream) {
    GSA_RETURN_IF(!m_dev, GSA_ERR_INVALID_ARGUMENT);
    return tile_offsets_impl<uint32_t>(m_capacity, m_dev, sorted_tile, n_tiles, tile_offsets, stream);
}

extern "C" int gsa_tile_bins_supported(int width, int height, int bin_tile) {
    if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
    if (width <= 0 || height <= 0 || (bin_tile != GSA_TILE_SIZE && bin_tile != 2 * GSA_TILE_SIZE && bin_tile != 4 * GSA_TILE_SIZE))
        return 0;
    const long tile_w = gsa_div_up(width, bin_tile), tile_h = gsa_div_up(height, bin_tile);
    return tile_w < 1024 && tile_h < 1024 && tile_w * tile_h <= 65535 ? 1 : 0;
}

extern "C" int gsa_tile_bin16_supported(int width, int height) { return gsa_tile_bins_supported(width, height, GSA_TILE_SIZE); }

static int bin_columns(int n_tiles) { return ((n_tiles + 1 + 63) / 64) * 64; }   // columns of the direct path's count matrix

// 1: gsa_tile_bin16 called without tile_tmp / gid_tmp takes the direct path for this grid (per-bin histograms)
extern "C" int gsa_tile_bin16_direct(int width, int height, int bin_tile) {
    if (!gsa_tile_bins_supported(width, height, bin_tile)) return 0;
    if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
    const int n_tiles = gsa_div_up(width, bin_tile) * gsa_div_up(height, bin_tile);
    return n_tiles > kRadix && n_tiles + 1 <= kBinsDirect ? 1 : 0;
}

// 1: gsa_tile_bin16 given qbox leaves every record's quadrant box in the top byte of its id for this grid and population:
// the one-pass placement, bins of at most 4 x 4 quadrants (16 x 16 tiles or 32 x 32 bins), ids below 2^24
extern "C" int gsa_tile_bin16_boxes(int width, int height, int bin_tile, int n) {
    if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
    return gsa_tile_bin16_direct(width, height, bin_tile) && bin_tile <= 4 * GSA_TILE_SIZE && n >= 0 && n <= (1 << 24) ? 1 : 0;
}

// workspace of gsa_tile_bin16 for either of its paths
extern "C" size_t gsa_tile_bin16_workspace_bytes(int width, int height, int bin_tile, uint32_t m_capacity) {
    size_t bytes = gsa_sort_workspace_bytes((int)min(m_capacity, 0x7FFFFFFFu));
    if (gsa_tile_bin16_direct(width, height, bin_tile)) {
        if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
        const int n_tiles = gsa_div_up(width, bin_tile) * gsa_div_up(height, bin_tile);
        const size_t nb = (size_t)bin_columns(n_tiles);
        const size_t direct = ((size_t)sort_blocks((int)min(m_capacity, 0x7FFFFFFFu)) * nb + nb + 16) * sizeof(uint32_t);
        if (direct > bytes) bytes = direct;
    }
    return bytes;
}

static int tile_bin16_impl(int n, const uint32_t *order, const uint32_t *offsets, const uint32_t *rects,
                           const int32_t *block_owner, int width, int height, int bin_tile, uint32_t m_capacity,
                           const uint32_t *m_dev, uint16_t *tile_tmp,
                           uint32_t *gid_tmp, uint16_t *sorted_tile, uint32_t *sorted_gid, int32_t *tile_offsets,
                           const uint32_t *qbox, int32_t *tile_order,
                           void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(n < 0 || !tile_offsets, GSA_ERR_INVALID_ARGUMENT);
    // (the launch order comes out of the one-pass placement's last launch only, and not for an empty view)
    GSA_RETURN_IF(tile_order && (!gsa_tile_bin16_direct(width, height, bin_tile) || tile_tmp || gid_tmp || n == 0 ||
                                 m_capacity == 0),
                  GSA_ERR_INVALID_ARGUMENT);
    // (boxes: only where the records are placed in one pass, a bin has at most 4 x 4 quadrants and ids fit 24 bits --
    // gsa_tile_bin16_boxes)
    GSA_RETURN_IF(qbox && (!gsa_tile_bin16_boxes(width, height, bin_tile, n) || tile_tmp || gid_tmp), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!gsa_tile_bins_supported(width, height, bin_tile), GSA_ERR_UNSUPPORTED);
    GSA_RETURN_IF(m_capacity >= 0x7FFFFFFFu - (uint32_t)kSortChunk, GSA_ERR_UNSUPPORTED);
    // bin_tile: the side of the tiles the records are binned into -- GSA_TILE_SIZE (0 = the same), or twice that for
    // callers whose compositor maps its 16 x 16 tiles onto 32 x 32 bins (gsa_composite_*'s bin_tile)
    if (bin_tile == 0) bin_tile = GSA_TILE_SIZE;
    GSA_RETURN_IF(bin_tile != GSA_TILE_SIZE && bin_tile != 2 * GSA_TILE_SIZE && bin_tile != 4 * GSA_TILE_SIZE,
                  GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(width, bin_tile), tile_h = gsa_div_up(height, bin_tile);
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0 || m_capacity == 0) {   // no records: every tile's list is empty
        return (int)hipMemsetAsync(tile_offsets, 0, sizeof(int32_t) * ((size_t)n_tiles + 1), s);
    }
    GSA_RETURN_IF(!order || !offsets || !rects || !block_owner || !sorted_gid || !workspace, GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(workspace_bytes < gsa_sort_workspace_bytes((int)m_capacity), GSA_ERR_WORKSPACE_TOO_SMALL);
    const bool two = n_tiles > kRadix;   // tile ids above 8 bits: a second (high-digit) pass
    // ... unless the caller brought no buffers for one (tile_tmp == gid_tmp == NULL) and the grid is small enough for a
    // per-bin histogram in LDS: the direct path above (sorted_tile optional there: the lists are tile_offsets' alone)
    if (two && !tile_tmp && !gid_tmp && n_tiles + 1 <= kBinsDirect) {
        GSA_RETURN_IF(workspace_bytes < gsa_tile_bin16_workspace_bytes(width, height, bin_tile, m_capacity),
                      GSA_ERR_WORKSPACE_TOO_SMALL);
        const int nblocks = sort_blocks((int)m_capacity), nb = bin_columns(n_tiles);
        uint32_t *mat = static_cast<uint32_t *>(workspace);
        uint32_t *totals = mat + (size_t)nblocks * nb;
        uint32_t *done_counter = totals + nb;
        emit_binhist_kernel<<<nblocks, kSortThreads, 0, s>>>(n, order, offsets, rects, block_owner, tile_w, m_capacity, m_dev,
                                                             nblocks, nb, n_tiles, mat, done_counter);
        GSA_LAUNCH_CHECK();
        bin_colscan_kernel<<<nb / kColBins, kColThreads, 0, s>>>(m_capacity, m_dev, nb, n_tiles, mat, totals, tile_offsets,
                                                                  done_counter);
        GSA_LAUNCH_CHECK();
        // (tile_order: eight more workgroups write the compositor's launch order beside the placement)
        const int sched_shift = bin_tile == 4 * GSA_TILE_SIZE ? 2 : bin_tile == 2 * GSA_TILE_SIZE ? 1 : 0;
        const int sched_w = gsa_div_up(width, GSA_TILE_SIZE), sched_tiles = sched_w * gsa_div_up(height, GSA_TILE_SIZE);
        const int place_grid = nblocks + (tile_order ? 8 : 0);
#define GSA_PLACE(Q)                                                                                                     \
    emit_place_kernel<Q><<<place_grid, kSortThreads, 0, s>>>(n, order, offsets, rects, block_owner, tile_w, m_capacity, m_dev, \
                                                             sorted_tile, sorted_gid, nblocks, nb, n_tiles, mat, tile_offsets, \
                                                             qbox, tile_order, sched_tiles, sched_w, sched_shift)
        if (!qbox) GSA_PLACE(0);
        else if (bin_tile == GSA_TILE_SIZE) GSA_PLACE(2);
        else if (bin_tile == 2 * GSA_TILE_SIZE) GSA_PLACE(4);
        else GSA_PLACE(8);
#undef GSA_PLACE
        GSA_LAUNCH_CHECK();
        return GSA_OK;
    }
    GSA_RETURN_IF(!sorted_tile, GSA_ERR_INVALID_ARGUMENT);
    // Two passes split the id bits EVENLY (13 bits: 7 + 6, not 8 + 5): a block's 4096 records then leave the first
    // pass in 128 runs of ~32 (full 128-B lines of values) instead of 256 runs of ~16, and the second pass still writes
    // runs of 64+.
    int id_bits = 0;
    while ((1 << id_bits) < n_tiles) ++id_bits;
    const int low_bits = two ? (id_bits + 1) / 2 : 8;
    const uint32_t low_mask = (1u << low_bits) - 1u;
    GSA_RETURN_IF(two && (!tile_tmp || !gid_tmp || (reinterpret_cast<uintptr_t>(tile_tmp) & 3) != 0),
                  GSA_ERR_INVALID_ARGUMENT);
    const int nblocks = sort_blocks((int)m_capacity);
    uint32_t *hist = static_cast<uint32_t *>(workspace);
    uint32_t *digit_total = hist + (size_t)kRadix * nblocks;
    emit_hist_kernel<<<nblocks, kSortThreads, 0, s>>>(n, order, offsets, rects, block_owner, tile_w, m_capacity, m_dev,
                                                      nblocks, hist, low_mask);
    GSA_LAUNCH_CHECK();
    radix_scan_rows_kernel<<<kRadix, kSortThreads, 0, s>>>(nblocks, hist, digit_total);
    GSA_LAUNCH_CHECK();
    emit_scatter_kernel<<<nblocks, kSortThreads, 0, s>>>(n, order, offsets, rects, block_owner, tile_w, m_capacity, m_dev,
                                                         two ? tile_tmp : sorted_tile, two ? gid_tmp : sorted_gid,
                                                         nblocks, hist, digit_total, low_mask);
    GSA_LAUNCH_CHECK();
    if (two) {
        radix_hist_kernel<uint16_t><<<nblocks, kSortThreads, 0, s>>>((int)m_capacity, m_dev, tile_tmp, RadixDigit{low_bits, 0u, 0, nullptr, nullptr, 0u}, nblocks, hist, nullptr);
        GSA_LAUNCH_CHECK();
        radix_scan_rows_kernel<<<kRadix, kSortThreads, 0, s>>>(nblocks, hist, digit_total);
        GSA_LAUNCH_CHECK();
        radix_scatter_kernel<uint16_t><<<nblocks, kSortThreads, 0, s>>>((int)m_capacity, m_dev, tile_tmp, gid_tmp, sorted_tile,
                                                                        sorted_gid, RadixDigit{low_bits, 0u, 0, nullptr, nullptr, 0u}, nblocks, hist,
                                                                        digit_total);
        GSA_LAUNCH_CHECK();
    }
    return tile_offsets_impl<uint16_t>(m_capacity, m_dev, sorted_tile, n_tiles, tile_offsets, stream);
}

extern "C" int gsa_tile_bin16(int n, const uint32_t *order, const uint32_t *offsets, const uint32_t *rects,
                              const int32_t *block_owner, int width, int height, int bin_tile, uint32_t m_capacity,
                              const uint32_t *m_dev, uint16_t *tile_tmp,
                              uint32_t *gid_tmp, uint16_t *sorted_tile, uint32_t *sorted_gid, int32_t *tile_offsets,
                              const uint32_t *qbox,
                              void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    return tile_bin16_impl(n, order, offsets, rects, block_owner, width, height, bin_tile, m_capacity, m_dev, tile_tmp, gid_tmp,
                           sorted_tile, sorted_gid, tile_offsets, qbox, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int gsa_tile_bin16_scheduled(int n, const uint32_t *order, const uint32_t *offsets, const uint32_t *rects,
                                        const int32_t *block_owner, int width, int height, int bin_tile,
                                        uint32_t m_capacity, const uint32_t *m_dev, uint16_t *sorted_tile,
                                        uint32_t *sorted_gid, int32_t *tile_offsets, const uint32_t *qbox,
                                        int32_t *tile_order, void *workspace, size_t workspace_bytes, gsa_stream_t stream) {
    GSA_RETURN_IF(!tile_order, GSA_ERR_INVALID_ARGUMENT);
    return tile_bin16_impl(n, order, offsets, rects, block_owner, width, height, bin_tile, m_capacity, m_dev, nullptr, nullptr,
                           sorted_tile, sorted_gid, tile_offsets, qbox, tile_order, workspace, workspace_bytes, stream);
}
