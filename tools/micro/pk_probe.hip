// Does hand-packed fp32 (two samples per lane in a float2, wave-uniform operands splat from SGPRs) beat scalar code for the
// rotate / accumulate arithmetic of the adjoint's per-instance phase?   hipcc --offload-arch=gfx950 -O3 pk_probe.hip -o pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void scalar_kernel(const float* __restrict__ inst, float* out, int iters) {
    float x0 = threadIdx.x * 0.01f, y0 = 1.0f + x0, z0 = 2.0f - x0, x1 = x0 + 0.5f, y1 = y0 + 0.5f, z1 = z0 + 0.5f;
    float acc[18] = {0};
    for (int it = 0; it < iters; ++it) {
        const float* p = inst + (it & 15) * 16;                         // wave-uniform -> SGPRs
        const float tx = p[0], ty = p[1], tz = p[2], r00 = p[3], r01 = p[4], r02 = p[5], r10 = p[6], r11 = p[7], r12 = p[8], r20 = p[9], r21 = p[10], r22 = p[11];
#define STEP(x, y, z, o)                                                                       \
        { const float rx = x - tx, ry = y - ty, rz = z - tz;                                    \
          const float px = rx * r00 + ry * r10 + rz * r20, py = rx * r01 + ry * r11 + rz * r21, pz = rx * r02 + ry * r12 + rz * r22; \
          const float gx = r00 * px + r01 * py + r02 * pz, gy = r10 * px + r11 * py + r12 * pz, gz = r20 * px + r21 * py + r22 * pz; \
          acc[o + 0] += rx * px + gx * py; acc[o + 1] += rx * py + gx * pz; acc[o + 2] += rx * pz + gx * px;                       \
          acc[o + 3] += ry * px + gy * py; acc[o + 4] += ry * py + gy * pz; acc[o + 5] += ry * pz + gy * px;                       \
          acc[o + 6] += rz * px + gz * py; acc[o + 7] += rz * py + gz * pz; acc[o + 8] += rz * pz + gz * px; x += gx * 1e-9f; }
        STEP(x0, y0, z0, 0) STEP(x1, y1, z1, 9)
    }
    float s = 0; for (int i = 0; i < 18; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void packed_kernel(const float* __restrict__ inst, float* out, int iters) {
    const float x0 = threadIdx.x * 0.01f;
    float2v x = {x0, x0 + 0.5f}, y = {1.0f + x0, 1.5f + x0}, z = {2.0f - x0, 2.5f - x0};
    float2v acc[9] = {};
    for (int it = 0; it < iters; ++it) {
        const float* p = inst + (it & 15) * 16;
        const float tx = p[0], ty = p[1], tz = p[2], r00 = p[3], r01 = p[4], r02 = p[5], r10 = p[6], r11 = p[7], r12 = p[8], r20 = p[9], r21 = p[10], r22 = p[11];
        const float2v rx = x - tx, ry = y - ty, rz = z - tz;
        const float2v px = rx * r00 + ry * r10 + rz * r20, py = rx * r01 + ry * r11 + rz * r21, pz = rx * r02 + ry * r12 + rz * r22;
        const float2v gx = r00 * px + r01 * py + r02 * pz, gy = r10 * px + r11 * py + r12 * pz, gz = r20 * px + r21 * py + r22 * pz;
        acc[0] += rx * px + gx * py; acc[1] += rx * py + gx * pz; acc[2] += rx * pz + gx * px;
        acc[3] += ry * px + gy * py; acc[4] += ry * py + gy * pz; acc[5] += ry * pz + gy * px;
        acc[6] += rz * px + gz * py; acc[7] += rz * py + gz * pz; acc[8] += rz * pz + gz * px;
        x += gx * 1e-9f;
    }
    float s = 0; for (int i = 0; i < 9; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *inst, *out; hipMalloc(&inst, 1024); hipMalloc(&out, 4096 * 256 * 4);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.01f * (i % 7) + 0.1f;
    hipMemcpy(inst, h, 1024, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(scalar_kernel, dim3(4096), dim3(256), 0, 0, inst, out, iters);
            else hipLaunchKernelGGL(packed_kernel, dim3(4096), dim3(256), 0, 0, inst, out, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s: %.2f ms  (%.1f TFLOP/s of the 2 x 66 flop-pairs)\n", which ? "packed" : "scalar", ms, 4096.0 * 256 * iters * 2 * 114 / ms / 1e9);
        }
    }
    return 0;
}
