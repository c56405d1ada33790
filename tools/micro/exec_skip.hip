// Does a wave64 VALU instruction cost less when one half (or three quarters) of EXEC is off?  (Would make per-half instance culling
// free in the render loops.)   hipcc --offload-arch=gfx950 -O3 exec_skip.hip -o exec_skip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float c, unsigned long long mask) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = threadIdx.x + k;
    const unsigned long long saved = __builtin_amdgcn_read_exec();
    asm volatile("s_mov_b64 exec, %0" : : "s"(mask));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k]) : "v"(c));
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0xFFFFull, 0x0000FFFF0000FFFFull, 0x1ull};
    const char* names[] = {"all 64 lanes", "lanes 0-31", "lanes 32-63", "lanes 0-15", "lanes 0-15 + 32-47", "lane 0"};
    for (int waves = 2; waves <= 4; waves *= 2)
        for (int m = 0; m < 6; ++m) {
            float t = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(loop_kernel, dim3(cus * waves), dim3(256), 0, 0, out, iters, 1.0001f, masks[m]);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&t, a, b);
            }
            printf("%d waves/SIMD, exec = %-20s %.3f ms  (%.2f cycles per v_fma per SIMD at 2.4 GHz)\n", waves, names[m], t, t * 1e-3 * 2.4e9 / (double(iters) * 64 * waves));
        }
    return 0;
}
