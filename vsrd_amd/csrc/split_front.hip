// Second translation unit of libvsrd_hip: the front kernels of vsrd_render_residual_step with the per-instance MLP's products on
// v_mfma_f32_16x16x32_bf16, both operands split into two bfloat16 parts (VSRD_FLAG_MLP_SPLIT_BF16; residual.h: forward_tile_split).
// Compiled WITHOUT -amdgpu-sched-strategy=iterative-ilp (see residual.h and __graft_entry__.py) and with the headers' namespace renamed,
// so that its copies of the kernels do not collide with api.hip's at link time.
#define VSRD_SPLIT_BF16 1
#define vsrd vsrd_split
#include "render_kernels.h"
#undef vsrd
#include <cstring>
#include "split_front.h"

namespace vsrd_split_front {

int pack_images(const float* weights, int num_instances, int centred, unsigned* images, int frames, long long frame_stride, hipStream_t stream) {
    hipLaunchKernelGGL(vsrd_split::pack_mlp_images_kernel, dim3(num_instances, frames), dim3(256), 0, stream, weights, centred, images, frame_stride);
    return hipGetLastError() == hipSuccess ? kOk : kLaunchFailed;
}

namespace {
template <typename Kernel>
bool opt_in(Kernel kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess;
}
}  // namespace

int launch_front(const FrontLaunch& a, hipStream_t stream) {
    using namespace vsrd_split;
    if (a.field_args_bytes != sizeof(FieldArgs) || a.render_args_bytes != sizeof(RenderArgs)) return kUnsupported;
    FieldArgs f;
    RenderArgs c;
    std::memcpy(&f, a.field_args, sizeof f);
    std::memcpy(&c, a.render_args, sizeof c);
#define VSRD_ARGS f, a.instances, a.images, c, a.origins, a.directions, a.u_coarse, a.u_fine, a.targets, a.instance_weights, a.loss_scale, a.eikonal_scale, a.eikonal_norm, \
                  a.labels, a.box_partials, static_cast<float4*>(a.jets), a.loss_partials, a.seeds, a.masks, a.slots_per_instance, a.first, a.rays, a.accumulate
#define VSRD_FRONT(K)                                                                                                                         \
    do {                                                                                                                                      \
        if (a.export_samples) return kUnsupported;      /* (the two-round instantiation below writes them) */                                 \
        if (!opt_in(residual_step_front_kernel<K>, a.lds_bytes)) return kLdsRefused;                                                          \
        if (a.frames > 1) return kUnsupported;          /* (a batch's frames are launches of <= 2048 rays: the pair kernel) */                \
        hipLaunchKernelGGL(residual_step_front_kernel<K>, dim3(a.blocks), dim3(kBlockThreads), a.lds_bytes, stream, VSRD_ARGS);               \
    } while (0)
#define VSRD_PAIR(K)                                                                                                                          \
    do {                                                                                                                                      \
        if (a.frames > 1) {                                                                                                                   \
            if (!opt_in(residual_step_pair_kernel<K, true>, a.lds_bytes)) return kLdsRefused;                                                 \
            hipLaunchKernelGGL((residual_step_pair_kernel<K, true>), dim3(a.blocks, a.frames), dim3(kPairWaves * kWave), a.lds_bytes, stream, VSRD_ARGS); \
        } else {                                                                                                                              \
            if (!opt_in(residual_step_pair_kernel<K>, a.lds_bytes)) return kLdsRefused;                                                       \
            hipLaunchKernelGGL(residual_step_pair_kernel<K>, dim3(a.blocks), dim3(kPairWaves * kWave), a.lds_bytes, stream, VSRD_ARGS);       \
        }                                                                                                                                     \
    } while (0)
    if (a.pair) {
        switch (a.rounds) {
            case 2: VSRD_PAIR(2); break;
            case 4: VSRD_PAIR(4); break;
            default: return kUnsupported;
        }
    } else if (a.rounds == 2 && a.export_samples && a.frames <= 1) {
        if (!opt_in(residual_step_front_kernel<2, true>, a.lds_bytes)) return kLdsRefused;
        hipLaunchKernelGGL((residual_step_front_kernel<2, true>), dim3(a.blocks), dim3(kBlockThreads), a.lds_bytes, stream, VSRD_ARGS);
    } else {
        switch (a.rounds) {
            case 1: VSRD_FRONT(1); break;
            case 2: VSRD_FRONT(2); break;
            case 4: VSRD_FRONT(4); break;
            default: return kUnsupported;
        }
    }
#undef VSRD_FRONT
#undef VSRD_PAIR
#undef VSRD_ARGS
    return hipGetLastError() == hipSuccess ? kOk : kLaunchFailed;
}

int launch_adjoint(int blocks, const float* instances, const float* images, int num_instances, const float* seeds, const unsigned char* masks,
                   long long slots_per_instance, long long used_slots, int items_per_instance, int slots_per_item, unsigned* next_item, float* item_rows,
                   unsigned char* item_flags, int frames, long long frame_stride, hipStream_t stream) {
    using namespace vsrd_split;
    const size_t lds = (static_cast<size_t>(kMlpImageWords) + static_cast<size_t>(kMlpSplitScratchTiles) * kTileFloats) * sizeof(float);
    hipLaunchKernelGGL(residual_mlp_adjoint_split_kernel, dim3(blocks, frames), dim3(kWave), lds, stream, instances, images, num_instances, 0u, seeds, masks,
                       slots_per_instance, used_slots, items_per_instance, slots_per_item, next_item, item_rows, item_flags, frame_stride);
    return hipGetLastError() == hipSuccess ? kOk : kLaunchFailed;
}

}  // namespace vsrd_split_front
