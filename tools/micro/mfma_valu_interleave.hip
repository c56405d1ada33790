// Does fp32 VALU work hide under matrix-core instructions on ONE SIMD of gfx950?  mfma_overlap.hip / mfma_bf16_overlap.hip let the compiler
// schedule, and it clusters the MFMAs (10 of 16 back to back at the loop's end), so all waves of a SIMD are in the same phase and "both = sum"
// says nothing.  Here the order is pinned with asm volatile: [one MFMA, F independent v_fma_f32] x 16 per iteration, four accumulators.
// If the VALU instructions hide under the MFMA, a group costs max(mfma, F x fma); if not, the sum.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_interleave.hip -o mfma_valu_interleave && ./mfma_valu_interleave
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int kKind, int kFillers>     // kKind 0: no MFMA, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(256) void kernel(float* out, int iters, float c) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = a + 1u;
    float fa = threadIdx.x * 0.001f, fb = 1.0f + threadIdx.x * 0.002f;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (kKind == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(a), "v"(b));
            if (kKind == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int f = 0; f < kFillers; ++f) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[f & 7]) : "v"(c));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) total += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

template <typename K>
static double cycles_per_group(K k, int blocks, float* out, int iters, double ghz, int waves) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float t = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&t, a, b);
    }
    return t * 1e-3 * ghz * 1e9 / (double(iters) * 16 * waves);      // cycles of one SIMD per (MFMA + fillers) group of one wave
}

int main() {
    float* out; (void)hipMalloc(&out, 8192 * 256 * 4);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1.0e6;
    const int cus = prop.multiProcessorCount, iters = 20000;
    printf("cycles of one SIMD per group [1 MFMA + F v_fma_f32] of one wave (order pinned); 'sum' = MFMA alone + fillers alone\n");
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = cus * waves;
#define ROW(F)                                                                                                                        \
        {                                                                                                                             \
            const double v = cycles_per_group(kernel<0, F>, blocks, out, iters, ghz, waves);                                          \
            const double mb = cycles_per_group(kernel<1, 0>, blocks, out, iters, ghz, waves), bb = cycles_per_group(kernel<1, F>, blocks, out, iters, ghz, waves); \
            const double mf = cycles_per_group(kernel<2, 0>, blocks, out, iters, ghz, waves), bf = cycles_per_group(kernel<2, F>, blocks, out, iters, ghz, waves); \
            printf("  %d wave(s)/SIMD, F = %2d: fillers %.1f | bf16 16x16x32: alone %.1f, with fillers %.1f (sum %.1f, max %.1f) | f32 16x16x4: alone %.1f, with fillers %.1f (sum %.1f, max %.1f)\n", \
                   waves, F, v, mb, bb, mb + v, mb > v ? mb : v, mf, bf, mf + v, mf > v ? mf : v);                                     \
        }
        ROW(2) ROW(4) ROW(8) ROW(12)
#undef ROW
    }
    return 0;
}
