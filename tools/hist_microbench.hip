// Developer aid: how fast can one workgroup-per-4096-keys digit histogram go on gfx950?  (The radix passes' counting
// kernels run at ~1.2 keys per clock per CU whatever the key width -- is that the LDS atomic rate?)
//   hipcc -O3 --offload-arch=gfx950 tools/hist_microbench.hip -o /tmp/hist_mb && /tmp/hist_mb [n_keys] [n_digits]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

constexpr int kThreads = 256, kItems = 16, kChunk = kThreads * kItems;

// V0: loads only
__global__ __launch_bounds__(kThreads) void v0_load(const uint16_t *keys, int n, uint32_t *out) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(keys + (long)blockIdx.x * kChunk);
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r) acc ^= p[r * kThreads + threadIdx.x];
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}
// V1: LDS atomics, one per key
__global__ __launch_bounds__(kThreads) void v1_atomic(const uint16_t *keys, int n, uint32_t *out) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t *p = reinterpret_cast<const uint32_t *>(keys + (long)blockIdx.x * kChunk);
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r) {
        const uint32_t kk = p[r * kThreads + threadIdx.x];
        atomicAdd(&h[(kk >> 8) & 255], 1u);
        atomicAdd(&h[kk >> 24], 1u);
    }
    __syncthreads();
    out[(long)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}
// V2: per-wave private histograms
__global__ __launch_bounds__(kThreads) void v2_wavepriv(const uint16_t *keys, int n, uint32_t *out) {
    __shared__ uint32_t h[4][256];
    for (int w = 0; w < 4; ++w) h[w][threadIdx.x] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    const uint32_t *p = reinterpret_cast<const uint32_t *>(keys + (long)blockIdx.x * kChunk);
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r) {
        const uint32_t kk = p[r * kThreads + threadIdx.x];
        atomicAdd(&mine[(kk >> 8) & 255], 1u);
        atomicAdd(&mine[kk >> 24], 1u);
    }
    __syncthreads();
    out[(long)threadIdx.x * gridDim.x + blockIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}
// V3: byte-private counters: h8[digit][thread] (64 KB), no atomics; reduced with dot4
__global__ __launch_bounds__(kThreads) void v3_bytes(const uint16_t *keys, int n, uint32_t *out) {
    __shared__ uint32_t h8[256 * 64];   // [digit][thread / 4] packed bytes
    uint4 *z = reinterpret_cast<uint4 *>(h8);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k * kThreads + threadIdx.x] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint8_t *b = reinterpret_cast<uint8_t *>(h8);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(keys + (long)blockIdx.x * kChunk);
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r) {
        const uint32_t kk = p[r * kThreads + threadIdx.x];
        const uint32_t d0 = (kk >> 8) & 255, d1 = kk >> 24;
        b[d0 * 256 + threadIdx.x] += 1;
        b[d1 * 256 + threadIdx.x] += 1;
    }
    __syncthreads();
    const uint4 *row = reinterpret_cast<const uint4 *>(h8 + threadIdx.x * 64);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint4 v = row[k];
        s = __builtin_amdgcn_udot4(v.x, 0x01010101u, s, false);
        s = __builtin_amdgcn_udot4(v.y, 0x01010101u, s, false);
        s = __builtin_amdgcn_udot4(v.z, 0x01010101u, s, false);
        s = __builtin_amdgcn_udot4(v.w, 0x01010101u, s, false);
    }
    out[(long)threadIdx.x * gridDim.x + blockIdx.x] = s;
}
// V4: atomics with a returning form avoided? (ds_add_u32 without return is what atomicAdd-with-unused-result compiles to);
// here: halve the atomics by merging equal digits of a pair
__global__ __launch_bounds__(kThreads) void v4_pairmerge(const uint16_t *keys, int n, uint32_t *out) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t *p = reinterpret_cast<const uint32_t *>(keys + (long)blockIdx.x * kChunk);
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r) {
        const uint32_t kk = p[r * kThreads + threadIdx.x];
        const uint32_t d0 = (kk >> 8) & 255, d1 = kk >> 24;
        if (d0 == d1) atomicAdd(&h[d0], 2u);
        else { atomicAdd(&h[d0], 1u); atomicAdd(&h[d1], 1u); }
    }
    __syncthreads();
    out[(long)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

template <typename K>
float time_it(K kern, int grid, const uint16_t *keys, int n, uint32_t *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) kern<<<grid, kThreads>>>(keys, n, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) kern<<<grid, kThreads>>>(keys, n, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main(int argc, char **argv) {
    const int n = (argc > 1 ? atoi(argv[1]) : 8232492) / kChunk * kChunk;
    const int nd = argc > 2 ? atoi(argv[2]) : 32;     // distinct high digits (1080p: 32, 4K: 127)
    std::vector<uint16_t> h(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)((((s >> 16) % nd) << 8) | ((i / 997) & 255)); }
    uint16_t *keys; uint32_t *out;
    hipMalloc(&keys, (size_t)n * 2); hipMalloc(&out, (size_t)(n / kChunk) * 256 * 4);
    hipMemcpy(keys, h.data(), (size_t)n * 2, hipMemcpyHostToDevice);
    const int grid = n / kChunk;
    printf("n = %d keys, %d blocks, %d distinct high digits\n", n, grid, nd);
    printf("V0 loads only          %7.1f us\n", time_it(v0_load, grid, keys, n, out)); fflush(stdout);
    printf("V1 LDS atomics         %7.1f us\n", time_it(v1_atomic, grid, keys, n, out)); fflush(stdout);
    printf("V2 per-wave private    %7.1f us\n", time_it(v2_wavepriv, grid, keys, n, out)); fflush(stdout);
    printf("V3 byte-private + dot4 %7.1f us\n", time_it(v3_bytes, grid, keys, n, out)); fflush(stdout);
    printf("V4 pair merge          %7.1f us\n", time_it(v4_pairmerge, grid, keys, n, out)); fflush(stdout);
    return 0;
}
