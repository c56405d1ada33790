// Issue rate of the VALU instruction classes the render kernels are made of, at their occupancy (4 waves per SIMD):
// cycles per wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 op_rates.hip -o op_rates && ./op_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)
#define BODY(name, INSTR)                                                                                              \
    __global__ __launch_bounds__(256) void name(float* out, int iters, float c, float sv) {                             \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv)));                \
        float d = c * 1.5f + threadIdx.x;                                                                                 \
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_mov_b64 s[22:23], vcc" : : "v"(a0), "v"(c) : "vcc", "s22", "s23");                \
        for (int it = 0; it < iters; ++it) {                                                                             \
            REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR)                                                              \
        }                                                                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s + d;                           \
    }

#define I_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r) : "v"(c));
#define I_FMAC(r) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r) : "v"(c));
#define I_FMA_S(r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r) : "s"(s));
#define I_MUL(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MAX(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MIN3(r) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
#define I_CND(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : );
#define I_CMP(r) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r), "v"(c) : "vcc");
#define I_CMP_E64(r) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" : : "v"(r), "v"(c) : "s20", "s21");
#define I_BFI(r) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r) : "v"(c));
#define I_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MOV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(c));
#define I_DPP(r) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
#define I_EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
#define I_RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define I_SQRT(r) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r));
#define I_RDL(r) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r) : "s20");
#define I_SUBABS(r) asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(r) : "v"(c));
#define I_CND64(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(r) : "v"(c));
#define I_CMPCND(r) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");
#define I_CND0(r) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "+v"(r) : "v"(c));
#define I_MAXS(r) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r) : "s"(s));
#define I_MULS(r) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "s"(s));
#define I_FMA3(r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(d));
#define I_MUL3(r) asm volatile("v_mul_f32 %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_FMAC3(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_FMAC_S(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "s"(s), "v"(d));
#define I_FMA_E64_SAME(r) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r) : "v"(c));
#define I_MIX(r) asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_max_f32 %0, %0, %1" : "+v"(r) : "v"(c), "v"(d));
#define I_PKFMA(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p##r) : "v"(pc));

BODY(k_fma, I_FMA) BODY(k_fmac, I_FMAC) BODY(k_fma_s, I_FMA_S) BODY(k_mul, I_MUL) BODY(k_add, I_ADD) BODY(k_max, I_MAX) BODY(k_min3, I_MIN3)
BODY(k_cnd, I_CND) BODY(k_cmp, I_CMP) BODY(k_cmp64, I_CMP_E64) BODY(k_bfi, I_BFI) BODY(k_and, I_AND) BODY(k_mov, I_MOV) BODY(k_dpp, I_DPP)
BODY(k_cnd64, I_CND64) BODY(k_cmpcnd, I_CMPCND) BODY(k_cnd0, I_CND0) BODY(k_maxs, I_MAXS) BODY(k_muls, I_MULS) BODY(k_fma3, I_FMA3) BODY(k_mul3, I_MUL3)
BODY(k_fmac3, I_FMAC3) BODY(k_fmac_s, I_FMAC_S) BODY(k_fma_same, I_FMA_E64_SAME) BODY(k_mix, I_MIX)
BODY(k_exp, I_EXP) BODY(k_rcp, I_RCP) BODY(k_sqrt, I_SQRT) BODY(k_rdl, I_RDL) BODY(k_subabs, I_SUBABS)

struct Entry { const char* name; void (*fn)(float*, int, float, float); };

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1.0e6;
    const int cus = prop.multiProcessorCount;
    const Entry entries[] = {{"v_fma_f32 (vgpr)", k_fma}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 (sgpr operand)", k_fma_s}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add},
                             {"v_max_f32", k_max}, {"v_min3_f32", k_min3}, {"v_cndmask_b32", k_cnd}, {"v_cmp_gt_f32 vcc", k_cmp}, {"v_cmp_gt_f32 sgpr pair", k_cmp64},
                             {"v_bfi_b32", k_bfi}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_add_f32 dpp row_shr", k_dpp}, {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64}, {"v_cmp + v_cndmask pair (x2)", k_cmpcnd}, {"v_cndmask_b32 0, v, vcc", k_cnd0}, {"v_max_f32 (sgpr operand)", k_maxs}, {"v_mul_f32 (sgpr operand)", k_muls}, {"v_fma_f32 d, a, b, d", k_fma3}, {"v_mul_f32 d, a, b", k_mul3}, {"v_fmac_f32 d, a, b", k_fmac3}, {"v_fmac_f32 d, s, b", k_fmac_s}, {"v_fma_f32 d, a, a, d", k_fma_same}, {"v_fma + v_max pair (x2)", k_mix}, {"v_exp_f32", k_exp},
                             {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_readlane_b32", k_rdl}, {"v_sub_f32 |abs|", k_subabs}};
    const int iters = 4000;
    const int blocks = cus * 4;            // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    printf("%d CUs at %.2f GHz, %d workgroups x 256 threads\n", cus, ghz, blocks);
    for (const Entry& e : entries) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double wave_instrs_per_simd = 4.0 * iters * 32;                      // 4 waves per SIMD
        printf("%-28s %7.3f ms   %.2f cycles per wave-instruction\n", e.name, ms, ms * 1e-3 * ghz * 1e9 / wave_instrs_per_simd);
    }
    return 0;
}
