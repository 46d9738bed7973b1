// Which lane's A and which lane's B meet in output register v of lane l of v_mfma_f32_4x4x1_16b_f32 (gfx950)?
// Expectation (16 blocks of 4 lanes): D[v][l] = A[4 * (l / 4) + v] * B[l] + C[v][l].
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_4x4x1_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
    const int l = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 100.f + (float)l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, 0.5f, c, 0, 0, 0);   // accumulates: + 0.5
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const float want = (float)(4 * (l / 4) + v) * (100.f + (float)l) + 0.5f;
            if (h[l * 4 + v] != want) { if (bad < 8) printf("lane %d v %d: got %g want %g\n", l, v, h[l * 4 + v], want); ++bad; }
        }
    printf("mfma_f32_4x4x1f32 layout: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as expected: D[v][l] = A[4(l/4)+v] * B[l] + C[v][l]", bad);
    return bad != 0;
}
