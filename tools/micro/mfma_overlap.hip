// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and fp32 VALU work overlap on one SIMD of gfx950?  The residual kernels' time model
// depends on it.  Three kernels with the same loop structure: 16 independent-accumulator MFMAs per iteration, 64
// independent-chain v_fma_f32 per iteration, and both interleaved (4 VALU after each MFMA); at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;

#define VALU4(r0, r1, r2, r3)                                            \
    asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\t" \
                 "v_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3"    \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));

template <bool kMfma, bool kValu>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float c) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (kMfma) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q & 3], 0, 0, 0);
            if (kValu) VALU4(v[(4 * q) & 15], v[(4 * q + 1) & 15], v[(4 * q + 2) & 15], v[(4 * q + 3) & 15])
        }
    }
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) total += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1.0e6;
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    printf("%d CUs at %.2f GHz nominal; per iteration: 16 MFMA 16x16x4 f32 (4 accumulators), 64 v_fma_f32 (16 chains)\n", cus, ghz);
    for (int waves = 1; waves <= 4; waves *= 2) {
        double ms[3];
        for (int mode = 0; mode < 3; ++mode) {
            float t = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL((loop_kernel<true, false>), dim3(cus * waves), dim3(256), 0, 0, out, iters, 1.0001f);
                if (mode == 1) hipLaunchKernelGGL((loop_kernel<false, true>), dim3(cus * waves), dim3(256), 0, 0, out, iters, 1.0001f);
                if (mode == 2) hipLaunchKernelGGL((loop_kernel<true, true>), dim3(cus * waves), dim3(256), 0, 0, out, iters, 1.0001f);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&t, a, b);
            }
            ms[mode] = t;
        }
        const double per_iter = 1e-3 * ghz * 1e9 / (double(iters) * waves);
        printf("%d wave(s)/SIMD: mfma only %.3f ms (%.1f cyc/MFMA)   valu only %.3f ms (%.2f cyc/FMA)   both %.3f ms  -> both / (mfma + valu) = %.2f, both / max = %.2f\n",
               waves, ms[0], ms[0] * per_iter / 16, ms[1], ms[1] * per_iter / 64, ms[2], ms[2] / (ms[0] + ms[1]), ms[2] / (ms[0] > ms[1] ? ms[0] : ms[1]));
    }
    return 0;
}
