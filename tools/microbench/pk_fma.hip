// Issue rate of v_pk_fma_f32 against v_fma_f32 on one wave per SIMD and on full occupancy (is packed fp32 full rate on gfx950?)
// build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/microbench/pk_fma.hip -o /tmp/pk_fma ; run: /tmp/pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kChains = 8;

__global__ void scalar_fma(float *out, float m) {
    float v[kChains];
    for (int c = 0; c < kChains; ++c) v[c] = threadIdx.x + c;
    for (int i = 0; i < kIters; ++i)
#pragma unroll
        for (int c = 0; c < kChains; ++c) v[c] = __builtin_fmaf(v[c], m, 1.0f);
    float s = 0;
    for (int c = 0; c < kChains; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void packed_fma(float *out, float m) {
    f32x2 v[kChains];
    const f32x2 mm = {m, m}, one = {1.0f, 1.0f};
    for (int c = 0; c < kChains; ++c) v[c] = f32x2{(float)threadIdx.x + c, (float)c};
    for (int i = 0; i < kIters; ++i)
#pragma unroll
        for (int c = 0; c < kChains; ++c) v[c] = __builtin_elementwise_fma(v[c], mm, one);
    float s = 0;
    for (int c = 0; c < kChains; ++c) s += v[c].x + v[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
static float time_ms(K k, int blocks, int threads, float *out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<<<blocks, threads>>>(out, 0.999f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int r = 0; r < 10; ++r) k<<<blocks, threads>>>(out, 0.999f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * waves_per_simd / 4, threads = 256;
        const float s = time_ms(scalar_fma, blocks, threads, out), p = time_ms(packed_fma, blocks, threads, out);
        const double instr = (double)kIters * kChains;   // per wave
        printf("%d waves/SIMD: v_fma_f32 %.3f ms (%.2f clk/instr/wave at 2.4 GHz), v_pk_fma_f32 %.3f ms (%.2f clk)\n", waves_per_simd, s,
               s * 1e-3 * 2.4e9 / (instr * waves_per_simd), p, p * 1e-3 * 2.4e9 / (instr * waves_per_simd));
    }
    return 0;
}
