// What does one bf16 hi / lo split of a PAIR of fp32 values cost on one SIMD of gfx950, and is v_dot2_f32_bf16 a way to make it cheaper?
//   form A (residual.h: split4):  h = cvt_pk(x0, x1); r0 = x0 - (h << 16); r1 = x1 - (h & 0xffff0000); l = cvt_pk(r0, r1)          6 VALU
//   form B:                       h = cvt_pk(x0, x1); r0 = dot2(h, {-1, 0}, x0); r1 = dot2(h, {0, -1}, x1); l = cvt_pk(r0, r1)      4 VALU
// (v_dot2_f32_bf16 D = S0.lo * S1.lo + S0.hi * S1.hi + S2: with S1 = {-1, 0} it is x0 - hi0 in one instruction -- if the instruction
//  issues at the rate of a plain one.)  Also timed: streams of the single instructions.  Order pinned with asm volatile.
//   hipcc --offload-arch=gfx950 -O3 split_rates.hip -o split_rates && ./split_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kKind>
__global__ __launch_bounds__(256) void kernel(float* out, int iters, float c) {
    float x[8];
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.37f + k * 1.001f;
    const unsigned minus_lo = 0x0000bf80u, minus_hi = 0xbf800000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // four independent pairs per iteration
            float& x0 = x[2 * q];
            float& x1 = x[2 * q + 1];
            if (kKind == 0) {                            // form A
                float r0, r1;
                unsigned t0, t1;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[q]) : "v"(x0), "v"(x1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0) : "v"(h[q]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1) : "v"(h[q]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(t0));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(t1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[q]) : "v"(r0), "v"(r1));
                x0 += r0 * c; x1 += r1 * c;              // (keeps the chain alive: 2 more VALU in every form)
            } else if (kKind == 1) {                     // form B
                float r0, r1;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[q]) : "v"(x0), "v"(x1));
                asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r0) : "v"(h[q]), "v"(minus_lo), "v"(x0));
                asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1) : "v"(h[q]), "v"(minus_hi), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[q]) : "v"(r0), "v"(r1));
                x0 += r0 * c; x1 += r1 * c;
            } else if (kKind == 2) {                     // 6 plain v_fma_f32
#pragma unroll
                for (int f = 0; f < 6; ++f) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(2 * q + f) & 7]) : "v"(c));
                x0 += x1 * c; x1 += x0 * c;
            } else if (kKind == 3) {                     // 6 v_dot2_f32_bf16
#pragma unroll
                for (int f = 0; f < 6; ++f) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[(2 * q + f) & 7]) : "v"(minus_lo), "v"(minus_hi));
                x0 += x1 * c; x1 += x0 * c;
            } else {                                     // 6 v_cvt_pk_bf16_f32
#pragma unroll
                for (int f = 0; f < 6; ++f) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[f & 3]) : "v"(x[(2 * q + f) & 7]), "v"(x[(2 * q + f + 1) & 7]));
                x0 += x1 * c; x1 += x0 * c;
            }
        }
    }
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += x[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) total += __uint_as_float(h[k]) + __uint_as_float(l[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

// exactness of form B against form A over random values
__global__ void check(const float* x, unsigned* bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned h, la, lb;
    float r0, r1, s0, s1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    r0 = x0 - __uint_as_float(h << 16); r1 = x1 - __uint_as_float(h & 0xffff0000u);
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(la) : "v"(r0), "v"(r1));
    const unsigned minus_lo = 0x0000bf80u, minus_hi = 0xbf800000u;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(s0) : "v"(h), "v"(minus_lo), "v"(x0));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(s1) : "v"(h), "v"(minus_hi), "v"(x1));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lb) : "v"(s0), "v"(s1));
    if (la != lb) atomicAdd(bad, 1u);
}

template <typename K>
static double cycles(K k, int blocks, float* out, int iters, double ghz, int waves) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float t = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0e-6f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&t, a, b);
    }
    return t * 1e-3 * ghz * 1e9 / (double(iters) * 4 * waves);      // cycles of one SIMD per pair (group) of one wave
}

int main() {
    float* out; (void)hipMalloc(&out, 8192 * 256 * 4);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1.0e6;
    const int cus = prop.multiProcessorCount, iters = 20000;
    printf("cycles of one SIMD per group of one wave; every group ends in the same 2 VALU (fma chain)\n");
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = cus * waves;
        printf("  %d wave(s)/SIMD: split form A (6 VALU) %.1f | form B (cvt, 2 dot2, cvt) %.1f | 6 v_fma_f32 %.1f | 6 v_dot2_f32_bf16 %.1f | 6 v_cvt_pk_bf16_f32 %.1f\n", waves,
               cycles(kernel<0>, blocks, out, iters, ghz, waves), cycles(kernel<1>, blocks, out, iters, ghz, waves), cycles(kernel<2>, blocks, out, iters, ghz, waves),
               cycles(kernel<3>, blocks, out, iters, ghz, waves), cycles(kernel<4>, blocks, out, iters, ghz, waves));
    }
    const int n = 1 << 22;
    float* x; unsigned* bad; (void)hipMalloc(&x, n * 4); (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    float* host = new float[n];
    unsigned state = 12345u;
    for (int i = 0; i < n; ++i) { state = state * 1664525u + 1013904223u; const int e = int(state >> 27) - 16; host[i] = (float((state >> 3) & 0xffffff) / 8388608.0f - 1.0f) * __builtin_ldexpf(1.0f, e); }
    (void)hipMemcpy(x, host, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, x, bad, n);
    unsigned count = 0; (void)hipMemcpy(&count, bad, 4, hipMemcpyDeviceToHost);
    printf("form B's low parts differ from form A's on %u of %d pairs\n", count, n / 2);
    return 0;
}
