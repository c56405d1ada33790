// Interface between the two translation units of libvsrd_hip: api.hip (everything, compiled with the library's scheduler strategy) and
// split_front.hip (the residual step's front kernels with the split-bf16 products of VSRD_FLAG_MLP_SPLIT_BF16, compiled with the default
// strategy: residual.h says why).  Plain pointers and bytes only: the kernels' argument blocks are the same structs in both units, but
// they live in different namespaces there (split_front.hip compiles the headers as `vsrd_split`), so they cross as bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vsrd_split_front {

struct FrontLaunch {
    const void* field_args; size_t field_args_bytes;       // vsrd::FieldArgs
    const void* render_args; size_t render_args_bytes;     // vsrd::RenderArgs
    const float* instances; const float* images;           // images: the table pack_images wrote (kMlpImageWords words per instance)
    const float* origins; const float* directions; const float* u_coarse; const float* u_fine;
    const float* targets; const float* instance_weights;
    float loss_scale, eikonal_scale, eikonal_norm;
    float* labels; float* box_partials; void* jets; float* loss_partials; float* seeds; unsigned char* masks;
    long long slots_per_instance;
    int first, rays, accumulate;
    bool pair;                                             // residual_step_pair_kernel (a ray over two waves) or residual_step_front_kernel
    int rounds, blocks;
    size_t lds_bytes;
    bool export_samples;                                   // vsrd_render_config::out_* set: the instantiation that writes the step's samples (two rounds only)
    int frames;                                            // frame batch (include/vsrd_hip.h, ABI 8): the grid's y extent; the stride travels in RenderArgs
};

enum { kOk = 0, kLdsRefused = 1, kUnsupported = 2, kLaunchFailed = 3 };

int pack_images(const float* weights, int num_instances, int centred, unsigned* images, int frames, long long frame_stride, hipStream_t stream);
int launch_front(const FrontLaunch& launch, hipStream_t stream);
// residual_mlp_adjoint_split_kernel: the arguments of residual_mlp_adjoint_kernel with the image table in the place of the weights
int launch_adjoint(int blocks, const float* instances, const float* images, int num_instances, const float* seeds, const unsigned char* masks,
                   long long slots_per_instance, long long used_slots, int items_per_instance, int slots_per_item, unsigned* next_item, float* item_rows,
                   unsigned char* item_flags, int frames, long long frame_stride, hipStream_t stream);

}  // namespace vsrd_split_front
