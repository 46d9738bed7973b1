// Developer aid: issue cost of individual gfx950 instructions, as the compositor uses them.
// Each kernel runs a loop of 64 independent copies of ONE instruction (8 registers x 8) at 8 waves per SIMD on
// every CU and reports cycles per wave-instruction per SIMD (2.0 = the SIMD-32 full rate for wave64).
//   hipcc -O3 --offload-arch=gfx950 tools/inst_microbench.hip -o /tmp/inst_microbench && /tmp/inst_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
#define BODY(S) for (int i = 0; i < iters; ++i) { REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) }

#define KERNEL(NAME, S)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float a, float b) {           \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,     \
              x6 = x0 + 6, x7 = x0 + 7;                                                              \
        unsigned long long sm = 0x5555555555555555ull + (unsigned long long)iters;                   \
        float sa = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a)));                   \
        unsigned si = (unsigned)iters;                                                               \
        __shared__ float4 lds[512];                                                                  \
        lds[threadIdx.x] = make_float4(a, b, a, b);                                                  \
        lds[threadIdx.x + 256] = make_float4(a, b, a, b);                                            \
        __syncthreads();                                                                             \
        unsigned laddr = (unsigned)(iters & 7) * 16;                                                 \
        (void)sm; (void)sa; (void)laddr; (void)si;                                                             \
        BODY(S)                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                 \
    }

#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define S_FMA_SGPR(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(sa), "v"(b));
#define S_MUL(x) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_MUL_LIT(x) asm volatile("v_mul_f32_e32 %0, 0x3fb8aa3b, %0" : "+v"(x));
#define S_ADD(x) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_SUB1(x) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(x));
#define S_MIN(x) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define S_MOV(x) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x) : "v"(a));
#define S_AND(x) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
#define S_LSHR(x) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(x));
#define S_CND_VCC(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
#define S_CND_SGPR(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(sm));
#define S_CND_ABS(x) asm volatile("v_cndmask_b32_e64 %0, -|%0|, %1, %2" : "+v"(x) : "v"(a), "s"(sm));
#define S_CMP_VCC(x) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
#define S_CMP_SGPR(x) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm) : "v"(x), "v"(a));
#define S_CMPCND(x) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define S_EXP(x) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x));
#define S_RCP(x) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x));
#define S_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##x) : "v"(pa));
#define S_DPP_QUAD(x) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_DPP_ROR(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_DPP_MIRROR(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_DPP_BCAST(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
#define S_DPP_WSHR(x) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_SWAP32(x) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(x), "+v"(x7));
#define S_SWAP16(x) asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(x), "+v"(x7));
#define S_READLANE(x) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(si) : "v"(x));
#define S_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x));
#define S_NOP(x) asm volatile("s_nop 0");
#define S_SALU(x) asm volatile("s_add_u32 %0, %0, 1" : "+s"(si) : : "scc");
#define S_LDS128(x) { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr)); }
#define S_LDS64(x) { float2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(laddr)); }
#define S_LDS32(x) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(laddr)); }

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters, float a, float b) {
    f2 px0 = {a, b}, px1 = px0 + 1.f, px2 = px0 + 2.f, px3 = px0 + 3.f, px4 = px0 + 4.f, px5 = px0 + 5.f, px6 = px0 + 6.f,
       px7 = px0 + 7.f, pa = {a, a};
#define S_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##x) : "v"(pa));
    BODY(S_PKFMA)
    out[blockIdx.x * 256 + threadIdx.x] = px0.x + px1.y + px2.x + px3.x + px4.x + px5.x + px6.x + px7.x;
}

KERNEL(k_fma, S_FMA) KERNEL(k_fma_sgpr, S_FMA_SGPR) KERNEL(k_mul, S_MUL) KERNEL(k_mul_lit, S_MUL_LIT) KERNEL(k_add, S_ADD)
KERNEL(k_sub1, S_SUB1) KERNEL(k_min, S_MIN) KERNEL(k_med3, S_MED3) KERNEL(k_mov, S_MOV) KERNEL(k_and, S_AND)
KERNEL(k_lshr, S_LSHR) KERNEL(k_cnd_vcc, S_CND_VCC) KERNEL(k_cnd_sgpr, S_CND_SGPR) KERNEL(k_cnd_abs, S_CND_ABS)
KERNEL(k_cmp_vcc, S_CMP_VCC) KERNEL(k_cmp_sgpr, S_CMP_SGPR) KERNEL(k_cmpcnd, S_CMPCND) KERNEL(k_exp, S_EXP)
KERNEL(k_rcp, S_RCP) KERNEL(k_dpp_quad, S_DPP_QUAD) KERNEL(k_dpp_ror, S_DPP_ROR) KERNEL(k_dpp_mirror, S_DPP_MIRROR)
KERNEL(k_dpp_bcast, S_DPP_BCAST) KERNEL(k_dpp_wshr, S_DPP_WSHR) KERNEL(k_swap32, S_SWAP32) KERNEL(k_swap16, S_SWAP16)
KERNEL(k_readlane, S_READLANE) KERNEL(k_mbcnt, S_MBCNT) KERNEL(k_nop, S_NOP) KERNEL(k_salu, S_SALU)
KERNEL(k_lds128, S_LDS128) KERNEL(k_lds64, S_LDS64) KERNEL(k_lds32, S_LDS32)

struct Test { const char *name; void (*fn)(float *, int, float, float); int per_inst; };

int main(int argc, char **argv) {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    Test tests[] = {
        {"v_fma_f32", k_fma, 1}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 1}, {"v_mul_f32", k_mul, 1},
        {"v_mul_f32 (literal)", k_mul_lit, 1}, {"v_add_f32", k_add, 1}, {"v_sub_f32 1.0", k_sub1, 1},
        {"v_min_f32", k_min, 1}, {"v_med3_f32", k_med3, 1}, {"v_mov_b32", k_mov, 1}, {"v_and_b32", k_and, 1},
        {"v_lshrrev_b32", k_lshr, 1}, {"v_cndmask e32 vcc", k_cnd_vcc, 1}, {"v_cndmask e64 sgpr", k_cnd_sgpr, 1},
        {"v_cndmask e64 -|x|", k_cnd_abs, 1}, {"v_cmp e32 vcc", k_cmp_vcc, 1}, {"v_cmp e64 sgpr", k_cmp_sgpr, 1},
        {"v_cmp+v_cndmask pair", k_cmpcnd, 2}, {"v_exp_f32", k_exp, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_pk_fma_f32", k_pkfma, 1}, {"v_add_f32_dpp quad_perm", k_dpp_quad, 1}, {"v_add_f32_dpp row_ror:8", k_dpp_ror, 1},
        {"v_add_f32_dpp half_mirror", k_dpp_mirror, 1}, {"v_add_f32_dpp row_bcast15", k_dpp_bcast, 1},
        {"v_mov_b32_dpp wave_shr:1", k_dpp_wshr, 1}, {"v_permlane32_swap", k_swap32, 1}, {"v_permlane16_swap", k_swap16, 1},
        {"v_readlane_b32", k_readlane, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"s_nop 0", k_nop, 1}, {"s_add_u32", k_salu, 1},
        {"ds_read_b128 (broadcast)", k_lds128, 1}, {"ds_read_b64 (broadcast)", k_lds64, 1}, {"ds_read_b32 (broadcast)", k_lds32, 1},
    };
    int wps_list[] = {8, 2, 1};
    int only = argc > 1 ? atoi(argv[1]) : -1, idx = -1;
    for (auto &t : tests) {
        ++idx;
        if (only >= 0 && idx != only) continue;
        printf("%-28s", t.name);
        for (int wps : wps_list) {
            int iters = 4000, grid = 256 * wps;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                t.fn<<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double insts_per_simd = (double)iters * 64 * wps * t.per_inst;
            printf("  %dw/SIMD: %6.2f cyc", wps, ms * 1e6 / insts_per_simd * 2.4);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
