// tile_schedule.h -- launch order of the compositor kernels: per XCD band, the tiles with the longest record lists first.
// Shared by composite.hip (gsa_tile_schedule: a launch of its own) and sort.hip (the one-pass placement runs it in eight
// extra workgroups of its last launch, beside the placement: gsa_tile_bin16_scheduled).
#pragma once
#include "gsa_common.h"

namespace {

// first[b] = number of tiles in the buckets before b: wave 0 scans the NB counters (a serial loop by one thread -- 256
// dependent LDS round trips -- was most of the 7 us these single-wave-of-work kernels took)
template <int NB>
__device__ __forceinline__ void bucket_starts(const int *__restrict__ cnt, int *__restrict__ first) {
    static_assert(NB % 64 == 0, "whole counters per lane");
    if (threadIdx.x < 64) {
        constexpr int PER = NB / 64;
        const int lane = threadIdx.x;
        int v[PER], s = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { v[j] = cnt[lane * PER + j]; s += v[j]; }
        int incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        int run = incl - s;
#pragma unroll
        for (int j = 0; j < PER; ++j) { first[lane * PER + j] = run; run += v[j]; }
    }
}

constexpr int kSchedBuckets = 64;
// One band (of eight) per call, by a workgroup of THREADS threads; cnt / first: kSchedBuckets ints of LDS each.
template <int SHIFT, int THREADS>
__device__ __forceinline__ void tile_schedule_band(int band, int n_tiles, const int32_t *__restrict__ tile_offsets,
                                                   int32_t *__restrict__ tile_order, int tile_w, int *cnt, int *first) {
    constexpr int kSchedThreads = THREADS;
    constexpr int SIDE = 1 << SHIFT, MEMBERS = SIDE * SIDE;
    const int chunk = (n_tiles + 7) >> 3;
    const int lo = band * chunk, hi = min(n_tiles, lo + chunk);
    if (threadIdx.x < kSchedBuckets) cnt[threadIdx.x] = 0;
    __syncthreads();
    // One tile per thread (a band has at most kSchedThreads tiles at the sizes the compositor launches; the loops below
    // cover more).  With bins (SHIFT > 0: the list a tile walks is its bin's, see tile_list) the tiles of one bin take
    // CONSECUTIVE places -- their waves then run together and gather the bin's records out of the same L2 at the same
    // time -- and the member with the smallest index inside this band speaks for its bin.
    const int tile_h = n_tiles / max(tile_w, 1), bin_w = (tile_w + SIDE - 1) >> SHIFT;
    // member j of t's bin, if it is a tile of the image inside this band (else -1)
    auto member = [&](int tx0, int ty0, int j) -> int {
        const int x = tx0 + (j & (SIDE - 1)), y = ty0 + (j >> SHIFT), u = y * tile_w + x;
        return x < tile_w && y < tile_h && u >= lo && u < hi ? u : -1;
    };
    auto plan = [&](int t, int &tx0, int &ty0, int &bkt) {   // -> number of places t reserves (0: another tile speaks for it)
        const int ty = t / max(tile_w, 1), tx = t - ty * tile_w;
        const int idx = SHIFT ? (ty >> SHIFT) * bin_w + (tx >> SHIFT) : t;
        const int len = tile_offsets[idx + 1] - tile_offsets[idx];
        bkt = kSchedBuckets - 1 - min(kSchedBuckets - 1, len >> 6);   // bucket 0 = longest lists
        tx0 = (tx >> SHIFT) << SHIFT; ty0 = (ty >> SHIFT) << SHIFT;
        int k = 0, head = -1;
#pragma unroll
        for (int j = 0; j < MEMBERS; ++j) {
            const int u = member(tx0, ty0, j);
            if (u >= 0) { if (head < 0) head = u; ++k; }
        }
        return head == t ? k : 0;
    };
    for (int t = lo + threadIdx.x; t < hi; t += kSchedThreads) {
        int tx0, ty0, bkt;
        const int k = plan(t, tx0, ty0, bkt);
        if (k) atomicAdd(&cnt[bkt], k);
    }
    __syncthreads();
    bucket_starts<kSchedBuckets>(cnt, first);
    __syncthreads();
    for (int t = lo + threadIdx.x; t < hi; t += kSchedThreads) {
        int tx0, ty0, bkt;
        const int k = plan(t, tx0, ty0, bkt);
        if (k) {
            int at = lo + atomicAdd(&first[bkt], k);
#pragma unroll
            for (int j = 0; j < MEMBERS; ++j) {
                const int u = member(tx0, ty0, j);
                if (u >= 0) tile_order[at++] = u;
            }
        }
    }
    for (int t = max(hi, lo) + threadIdx.x; t < lo + chunk; t += kSchedThreads) tile_order[t] = -1;   // padding
}

}  // namespace
