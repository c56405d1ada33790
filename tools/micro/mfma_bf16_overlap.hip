// Do bf16 MFMAs (v_mfma_f32_16x16x32_bf16) overlap with fp32 VALU work on one SIMD of gfx950 -- unlike the exact-fp32 MFMA, which
// shares the VALU's datapath (mfma_overlap.hip: both / (mfma + valu) = 1.0)?  Round 5: the split-bf16 form of the residual MLP
// (residual.h) is priced with these numbers.  Same loop structure as mfma_overlap.hip: 16 MFMAs on 4 accumulators per iteration,
// 64 v_fma_f32 on 16 chains, both interleaved; and a DEPENDENT form (one accumulator chain: MFMA -> 4 VALU on its result ->
// v_cvt_pk_bf16_f32 into the next B operand -> MFMA), the shape of one MLP tile.  1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_overlap.hip -o mfma_bf16_overlap && ./mfma_bf16_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define VALU4(r0, r1, r2, r3)                                            \
    asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\t" \
                 "v_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3"    \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));

template <bool kMfma, bool kValu, bool kF32>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float c) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    u32x4 ab = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, ab), b = __builtin_bit_cast(bf16x8, ab + 1u);
    float fa = threadIdx.x * 0.001f, fb = 1.0f + threadIdx.x * 0.002f;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (kMfma) {
                if (kF32) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[q & 3], 0, 0, 0);
                else acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q & 3], 0, 0, 0);
            }
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

// One dependent chain per wave, the shape of an MLP tile: z = W a (kMfmas MFMAs into one accumulator), then kValu VALU instructions
// on z (the LayerNorm / GELU algebra), the hi / lo split of the 4 results into the next B operand, and around again.
template <int kMfmas, int kValu, bool kF32>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, float c) {
    u32x4 ab = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, ab);
    bf16x8 b = __builtin_bit_cast(bf16x8, ab + 1u);
    float fa = threadIdx.x * 0.001f;
    f32x4 z = {0.1f, 0.2f, 0.3f, 0.4f};
    for (int it = 0; it < iters; ++it) {
        f32x4 acc = {0, 0, 0, 0};
        if (kF32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, z[k], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < kMfmas; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        z = acc;
#pragma unroll
        for (int k = 0; k < kValu / 4; ++k) VALU4(z[0], z[1], z[2], z[3])
        if (!kF32) {        // hi / lo split (round to nearest even twice): 12 VALU
            unsigned h01, h23, l01, l23;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h01) : "v"(z[0]), "v"(z[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h23) : "v"(z[2]), "v"(z[3]));
            const float r0 = z[0] - __uint_as_float(h01 << 16), r1 = z[1] - __uint_as_float(h01 & 0xffff0000u);
            const float r2 = z[2] - __uint_as_float(h23 << 16), r3 = z[3] - __uint_as_float(h23 & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l01) : "v"(r0), "v"(r1));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l23) : "v"(r2), "v"(r3));
            b = __builtin_bit_cast(bf16x8, u32x4{h01, h23, l01, l23});
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = z[0] + z[1] + z[2] + z[3];
}

template <typename K>
static float time_kernel(K kernel, int blocks, float* out, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float t = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&t, a, b);
    }
    return t;
}

int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1.0e6;
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    printf("%d CUs at %.2f GHz nominal; per iteration: 16 MFMA (4 accumulators), 64 v_fma_f32 (16 chains)\n", cus, ghz);
    for (int f32 = 0; f32 < 2; ++f32) {
        printf("%s\n", f32 ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_16x16x32_bf16");
        for (int waves = 1; waves <= 4; waves *= 2) {
            double ms[3];
            if (f32) {
                ms[0] = time_kernel(loop_kernel<true, false, true>, cus * waves, out, iters);
                ms[1] = time_kernel(loop_kernel<false, true, true>, cus * waves, out, iters);
                ms[2] = time_kernel(loop_kernel<true, true, true>, cus * waves, out, iters);
            } else {
                ms[0] = time_kernel(loop_kernel<true, false, false>, cus * waves, out, iters);
                ms[1] = time_kernel(loop_kernel<false, true, false>, cus * waves, out, iters);
                ms[2] = time_kernel(loop_kernel<true, true, false>, cus * waves, out, iters);
            }
            const double per_iter = 1e-3 * ghz * 1e9 / (double(iters) * waves);
            printf("  %d wave(s)/SIMD: mfma only %.3f ms (%.1f cyc/MFMA)   valu only %.3f ms (%.2f cyc/FMA)   both %.3f ms  -> both / (mfma + valu) = %.2f, both / max = %.2f\n",
                   waves, ms[0], ms[0] * per_iter / 16, ms[1], ms[1] * per_iter / 64, ms[2], ms[2] / (ms[0] + ms[1]), ms[2] / (ms[0] > ms[1] ? ms[0] : ms[1]));
        }
    }
    printf("dependent chain per wave (cycles per link = MFMAs + VALU + split, per SIMD):\n");
    for (int waves = 1; waves <= 4; waves *= 2) {
        const double per_iter = 1e-3 * ghz * 1e9 / (double(iters) * waves);
        const float f = time_kernel(chain_kernel<4, 40, true>, cus * waves, out, iters);
        const float b2 = time_kernel(chain_kernel<2, 40, false>, cus * waves, out, iters);
        const float b3 = time_kernel(chain_kernel<3, 40, false>, cus * waves, out, iters);
        const float b2v = time_kernel(chain_kernel<2, 0, false>, cus * waves, out, iters);
        const float fv = time_kernel(chain_kernel<4, 0, true>, cus * waves, out, iters);
        printf("  %d wave(s)/SIMD: fp32 4 MFMA + 40 VALU %.0f cyc | bf16 2 MFMA + 40 VALU + split %.0f | bf16 3 MFMA + 40 VALU + split %.0f | no VALU: fp32 %.0f, bf16 2 MFMA + split %.0f\n",
               waves, f * per_iter, b2 * per_iter, b3 * per_iter, fv * per_iter, b2v * per_iter);
    }
    return 0;
}
