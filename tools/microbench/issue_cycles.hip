// Developer aid (r05): issue cost of gfx950 vector instructions in SHADER CYCLES (s_memtime inside the kernel, not a
// host clock times an assumed 2.4 GHz), at the occupancies the compositor sweeps run at (5 and 7 waves per SIMD), plus the
// two sweeps' inner bodies as straight-line mixes -- the denominators of bench.py's `valu_roofline`.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/issue_cycles.hip -o /tmp/issue_cycles && /tmp/issue_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
#define BODY(S) for (int i = 0; i < iters; ++i) { REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) }

#define KERNEL(NAME, S)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, int iters, float a, float b) { \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,     \
              x6 = x0 + 6, x7 = x0 + 7;                                                              \
        unsigned long long sm = 0x5555555555555555ull + (unsigned long long)iters;                   \
        float sa = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a)));                \
        __shared__ float4 lds[512];                                                                  \
        lds[threadIdx.x] = make_float4(a, b, a, b);                                                  \
        lds[threadIdx.x + 256] = make_float4(a, b, a, b);                                            \
        __syncthreads();                                                                             \
        unsigned laddr = (unsigned)(iters & 7) * 16;                                                 \
        (void)sm; (void)sa; (void)laddr;                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        BODY(S)                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                 \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
    }

#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define S_FMAC(x) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define S_FMA_SGPR(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(sa), "v"(b));
#define S_MUL(x) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_MUL_NEG(x) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(x) : "v"(a));
#define S_SUB1(x) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(x));
#define S_MIN(x) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_MOV(x) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x) : "v"(a));
#define S_CND_VCC(x) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(x) : : );
#define S_CND_SGPR(x) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(x) : "s"(sm));
#define S_CMP_VCC(x) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
#define S_CMP_ABS_SGPR(x) asm volatile("v_cmp_le_f32_e64 %0, |%1|, %2" : "=s"(sm) : "v"(x), "s"(sa));
#define S_EXP(x) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x));
#define S_EXP_NEG(x) asm volatile("v_exp_f32_e64 %0, -%0" : "+v"(x));
#define S_RCP(x) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x));
#define S_DPP_ROR(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_SWAP32(x) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(x), "+v"(x7));
#define S_LDS128(x) { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr)); }
#define S_LDS64(x) { float2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(laddr)); }
// the backward sweep's phase A, one record (ALL_ON, no cap): 5 fma, exp, cmp |s| <= sgpr, cndmask vcc, 1 - a, rcp, 3 for c.v,
// T / ra, v_a (2), fac, u, Wd -- 18 vector instructions on one register (dependent like the real chain; 8 waves interleave)
#define S_PHASE_A(x) asm volatile(                                                                                      \
    "v_fma_f32 %0, %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n"     \
    "v_fmac_f32_e32 %0, %1, %2\n v_exp_f32_e64 %0, -%0\n v_cmp_le_f32_e64 vcc, |%0|, %3\n v_cndmask_b32_e32 %0, 0, %0, vcc\n" \
    "v_sub_f32_e32 %0, 1.0, %0\n v_rcp_f32_e32 %0, %0\n v_mul_f32_e32 %0, %0, %1\n v_fmac_f32_e32 %0, %1, %2\n"          \
    "v_fmac_f32_e32 %0, %1, %2\n v_mul_f32_e32 %0, %0, %1\n v_mul_f32_e32 %0, %0, %1\n v_fmac_f32_e32 %0, %1, %2\n"      \
    "v_mul_f32_e32 %0, %0, %1\n v_mul_f32_e64 %0, %0, -%1\n v_fma_f32 %0, -%0, %1, %2"                                   \
    : "+v"(x) : "v"(a), "v"(b), "s"(sa) : "vcc");
// the forward walk, one record of an optimistic group: 5 fma, exp, cmp -> sgpr pair, cndmask e64, T fma, weight mul, 3 colour fma
#define S_WALK(x) asm volatile(                                                                                          \
    "v_fma_f32 %0, %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n"     \
    "v_fmac_f32_e32 %0, %1, %2\n v_exp_f32_e64 %0, -%0\n v_cmp_le_f32_e64 %4, |%0|, %3\n v_cndmask_b32_e64 %0, 0, %0, %4\n" \
    "v_fma_f32 %0, -%0, %1, %1\n v_mul_f32_e32 %0, %0, %1\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n"     \
    "v_fmac_f32_e32 %0, %1, %2"                                                                                         \
    : "+v"(x) : "v"(a), "v"(b), "s"(sa), "s"(sm));

KERNEL(k_fma, S_FMA) KERNEL(k_fmac, S_FMAC) KERNEL(k_fma_sgpr, S_FMA_SGPR) KERNEL(k_mul, S_MUL) KERNEL(k_mul_neg, S_MUL_NEG)
KERNEL(k_sub1, S_SUB1) KERNEL(k_min, S_MIN) KERNEL(k_mov, S_MOV) KERNEL(k_cnd_vcc, S_CND_VCC) KERNEL(k_cnd_sgpr, S_CND_SGPR)
KERNEL(k_cmp_vcc, S_CMP_VCC) KERNEL(k_cmp_abs, S_CMP_ABS_SGPR) KERNEL(k_exp, S_EXP) KERNEL(k_exp_neg, S_EXP_NEG)
KERNEL(k_rcp, S_RCP) KERNEL(k_dpp, S_DPP_ROR) KERNEL(k_swap32, S_SWAP32) KERNEL(k_lds128, S_LDS128) KERNEL(k_lds64, S_LDS64)
KERNEL(k_phase_a, S_PHASE_A) KERNEL(k_walk, S_WALK)

struct Test { const char *name; void (*fn)(float *, unsigned long long *, int, float, float); int per_inst; };

int main(int argc, char **argv) {
    float *out;
    unsigned long long *cyc, *hcyc = (unsigned long long *)malloc(256 * 8 * 4 * 8);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 256 * 8 * 4 * 8);
    Test tests[] = {
        {"v_fma_f32", k_fma, 1}, {"v_fmac_f32_e32", k_fmac, 1}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 1}, {"v_mul_f32", k_mul, 1},
        {"v_mul_f32_e64 (neg)", k_mul_neg, 1}, {"v_sub_f32 1.0", k_sub1, 1}, {"v_min_f32", k_min, 1}, {"v_mov_b32", k_mov, 1},
        {"v_cndmask e32 vcc", k_cnd_vcc, 1}, {"v_cndmask e64 sgpr", k_cnd_sgpr, 1}, {"v_cmp e32 vcc", k_cmp_vcc, 1},
        {"v_cmp e64 |x| sgpr -> sgpr", k_cmp_abs, 1}, {"v_exp_f32", k_exp, 1}, {"v_exp_f32_e64 (neg)", k_exp_neg, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_add_f32_dpp row_ror:8", k_dpp, 1}, {"v_permlane32_swap", k_swap32, 1},
        {"ds_read_b128 (broadcast)", k_lds128, 1}, {"ds_read_b64 (broadcast)", k_lds64, 1},
        {"bwd phase A, per record (19 inst)", k_phase_a, 1}, {"fwd walk, per record (13 inst)", k_walk, 1},
    };
    int wps_list[] = {8, 7, 5, 2, 1};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-36s  cycles per wave-instruction (or per record) per SIMD, shader clock; [effective GHz]\n", "instruction");
    for (auto &t : tests) {
        printf("%-36s", t.name);
        for (int wps : wps_list) {
            int iters = 2000, grid = 256 * wps;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                t.fn<<<grid, 256>>>(out, cyc, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(hcyc, cyc, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < grid * 4; ++i) sum += (double)hcyc[i];
            const double avg = sum / (grid * 4);                       // cycles one wave was in its loop
            const double per = avg / ((double)iters * 64 * wps);       // the SIMD issued wps x iters x 64 of them meanwhile
            printf("  %dw: %6.2f [%.2f]", wps, per, avg / (ms * 1e6));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
