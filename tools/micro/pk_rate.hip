// Issue cost of packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 results per lane) against their scalar forms
// on gfx950, and of the ops that have no packed form (v_max_f32, v_cndmask_b32).  Decides whether evaluating two instances per lane
// with packed arithmetic can pay in the box-only render loops.   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int kOp>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float c) {
    float2v v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = float2v{threadIdx.x + k + 0.5f, threadIdx.x - k - 0.5f};
    float2v cc = float2v{c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k].x) : "v"(c));
                if (kOp == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[k]) : "v"(cc));
                if (kOp == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(cc));
                if (kOp == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(cc));
                if (kOp == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[k].x) : "v"(c));
                if (kOp == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k].x) : "v"(c));
                if (kOp == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[k]) : "s"(cc));
                if (kOp == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k].x) : "s"(c));
                if (kOp == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(v[k]) : "v"(cc));   // broadcast the low half of src1
                if (kOp == 9) asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(v[k].x) : "v"(c));
                if (kOp == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k].x) : "v"(c));
                if (kOp == 11) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(v[k]) : "v"(cc));
            }
    }
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += v[k].x + v[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

template <int kOp>
void run(const char* name, float* out, int cus) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int waves = 1; waves <= 4; waves *= 2) {
        float t = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(loop_kernel<kOp>, dim3(cus * waves), dim3(256), 0, 0, out, iters, 1.0001f);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&t, a, b);
        }
        printf("%-44s %d wave(s)/SIMD  %.3f ms  %.2f cycles per instruction per SIMD (2.4 GHz nominal)\n", name, waves, t, t * 1e-3 * 2.4e9 / (double(iters) * 64 * waves));
    }
}

int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    run<0>("v_fma_f32 v, v, v", out, cus);
    run<1>("v_pk_fma_f32 v, v, v", out, cus);
    run<2>("v_pk_mul_f32", out, cus);
    run<3>("v_pk_add_f32", out, cus);
    run<10>("v_mul_f32", out, cus);
    run<4>("v_max_f32", out, cus);
    run<9>("v_max_f32 with |src0|", out, cus);
    run<5>("v_cndmask_b32", out, cus);
    run<6>("v_pk_fma_f32 with an SGPR-pair operand", out, cus);
    run<7>("v_fma_f32 with an SGPR operand", out, cus);
    run<8>("v_pk_fma_f32 op_sel_hi broadcast", out, cus);
    run<11>("v_pk_mov_b32", out, cus);
    return 0;
}
