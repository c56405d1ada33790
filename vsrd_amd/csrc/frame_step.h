// The "glue" of one optimisation step around the render launch, as two single-workgroup kernels (native mode, hipGraph replay):
//
//   frame_prologue_kernel   box decode -> corners -> multi-view projection -> Hungarian matching on the target view -> projection
//                           losses AND their gradient w.r.t. the raw box parameters -> schedules -> the renderer's instance block
//   frame_epilogue_kernel   render adjoint [N,16] -> chain through the decode -> + projection gradients -> Adam on the three box
//                           tensors (moments / step counters / learning rates are torch.optim.Adam's own device tensors) ->
//                           ExponentialLR decay of every group's rate -> step counter
//
// Reference semantics (CPU restatement: oracle/step.py, oracle/geometry.py, oracle/losses.py):
//   BoxParameters3D.forward                                   vsrd/models/detectors/box_parameters.py:60-90,124-146
//   projection + clip_boxes_to_image                          scripts/main.py:339-367, geometric_operations.py:343-389
//   matching (negative DIoU, scipy linear_sum_assignment)     scripts/main.py:374-386
//   distance_box_iou_loss / smooth_l1_loss over kept pairs    scripts/main.py:391-415 (torchvision 0.14 formulas)
//   cosine-annealed schedules                                 scripts/main.py:420-431
//   Adam + ExponentialLR                                      configs/.../config.json:166-215, scripts/main.py:863-865
//
// With torch element-wise kernels these are ~330 graph nodes of ~2 us around a 53 us render launch; the work itself
// is a few hundred flops for each of <= 64 boxes x <= 32 views.
#pragma once
#include "matching.h"
#include "projection.h"

namespace vsrd {

constexpr int kFrameThreads = 256;
constexpr int kFrameMaxBoxes = 64;
constexpr int kFrameMaxViews = 32;
constexpr int kBoxEdges = 12;

struct FrameStepArgs {
    int num_boxes, num_views;
    float height, width, epsilon;
    float location_lo[3], location_hi[3], dimension_lo[3], dimension_hi[3];
    int num_steps;
    float max_temperature, min_temperature, max_std, min_std;
    float weight_iou, weight_l1, weight_silhouette;
    float beta1, beta2, adam_epsilon, lr_gamma;
    long long frame_stride;            // vsrd_frame_config::frame_stride (bytes; wave.h: frame batches), 0 for one frame
};

// scratch (global, one per frame): boxes_2d [V,N,4], selection [V,N,4] (int), grad_boxes [V,N,4], grad_world [V,N,8,3]
__host__ __device__ constexpr size_t frame_scratch_floats(int num_views, int num_boxes) {
    return static_cast<size_t>(num_views) * num_boxes * (4 + 4 + 4 + 24);
}

__device__ __forceinline__ float lerp_torch(float a, float b, float w) {           // ATen's lerp kernel
    const float diff = b - a;
    return (fabsf(w) < 0.5f) ? (a + w * diff) : (b - diff * (1.0f - w));
}
__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }

struct DecodedBox {
    float sig_l[3], sig_d[3];
    float loc[3], dim[3];
    float norm, c, s;          // max(|raw orientation|, 1e-12) and the unit heading
};

__device__ __forceinline__ DecodedBox decode_box(const FrameStepArgs& a, const float* raw_loc, const float* raw_dim, const float* raw_ori) {
    DecodedBox b;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        b.sig_l[j] = sigmoid_precise(raw_loc[j]);
        b.sig_d[j] = sigmoid_precise(raw_dim[j]);
        b.loc[j] = lerp_torch(a.location_lo[j], a.location_hi[j], b.sig_l[j]);
        b.dim[j] = lerp_torch(a.dimension_lo[j], a.dimension_hi[j], b.sig_d[j]);
    }
    b.norm = fmaxf(sqrtf(raw_ori[0] * raw_ori[0] + raw_ori[1] * raw_ori[1]), 1.0e-12f);       // F.normalize eps
    b.c = raw_ori[0] / b.norm;
    b.s = raw_ori[1] / b.norm;
    return b;
}

// box_parameters.py:78-90: corner k = R_y (u_k * dim) + loc
__device__ __forceinline__ void unit_corner(int k, float* u) {
    const int table[8][3] = {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, 1}, {1, 1, 1}, {1, 1, -1}, {-1, 1, -1}};
    u[0] = static_cast<float>(table[k][0]); u[1] = static_cast<float>(table[k][1]); u[2] = static_cast<float>(table[k][2]);
}

// d max(a, b) / d a with torch.maximum's tie rule (half each); d min(a, b) / d a likewise
__device__ __forceinline__ float dmax_first(float a, float b) { return (a > b) ? 1.0f : ((a == b) ? 0.5f : 0.0f); }
__device__ __forceinline__ float dmin_first(float a, float b) { return (a < b) ? 1.0f : ((a == b) ? 0.5f : 0.0f); }

// distance_box_iou_loss(pd, gt) (vsrd_amd/losses.py, torchvision 0.14) and its gradient w.r.t. pd = (x1, y1, x2, y2)
__device__ __forceinline__ float diou_loss(const float* p, const float* g, float* grad) {
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], x1g = g[0], y1g = g[1], x2g = g[2], y2g = g[3];
    const float eps = 1.0e-7f;
    const float ix1 = fmaxf(x1, x1g), iy1 = fmaxf(y1, y1g), ix2 = fminf(x2, x2g), iy2 = fminf(y2, y2g);
    const bool overlap = (iy2 > iy1) && (ix2 > ix1);
    const float iw = ix2 - ix1, ih = iy2 - iy1;
    const float inter = overlap ? iw * ih : 0.0f;
    const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
    const float wx = fmaxf(x2, x2g) - fminf(x1, x1g), wy = fmaxf(y2, y2g) - fminf(y1, y1g);
    const float diagonal = wx * wx + wy * wy + eps;
    const float cxd = (x1 + x2) / 2.0f - (x1g + x2g) / 2.0f, cyd = (y1 + y2) / 2.0f - (y1g + y2g) / 2.0f;
    const float centres = cxd * cxd + cyd * cyd;
    const float ue = uni + eps;
    const float dI[4] = {overlap ? -ih * dmax_first(x1, x1g) : 0.0f, overlap ? -iw * dmax_first(y1, y1g) : 0.0f,
                         overlap ? ih * dmin_first(x2, x2g) : 0.0f, overlap ? iw * dmin_first(y2, y2g) : 0.0f};
    const float dA[4] = {-(y2 - y1), -(x2 - x1), (y2 - y1), (x2 - x1)};                       // area of pd
    const float dD[4] = {-2.0f * wx * dmin_first(x1, x1g), -2.0f * wy * dmin_first(y1, y1g), 2.0f * wx * dmax_first(x2, x2g), 2.0f * wy * dmax_first(y2, y2g)};
    const float dC[4] = {cxd, cyd, cxd, cyd};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dU = dA[k] - dI[k];
        grad[k] = -(dI[k] * ue - inter * dU) / (ue * ue) + (dC[k] * diagonal - centres * dD[k]) / (diagonal * diagonal);
    }
    return 1.0f - inter / ue + centres / diagonal;
}

struct FrameBuffers {
    const float* raw_locations;      // [N,3]
    const float* raw_dimensions;     // [N,3]
    const float* raw_orientations;   // [N,2]
    const float* extrinsics;         // [V,16]
    const float* intrinsics;         // [V,9]
    const float* gt_boxes;           // [V,N,4]
    const unsigned char* visible;    // [V,N] (by ground-truth instance)
    const long long* step;           // device step counter
    float* scratch;                  // frame_scratch_floats(V, N)
    float* instances;                // out [N,16]
    long long* pd_indices;           // out [N]
    long long* gt_indices;           // out [N]
    int* target_map;                 // out [N]: ground-truth column of predicted instance n
    float* instance_weights;         // out [N]: 1 for matched predictions
    float* schedule;                 // out [3]: temperature, std, cosine ratio
    float* losses;                   // out [2]: iou, l1 projection loss
    float* grad_raw;                 // out [N,8]: weighted projection-loss gradient w.r.t. (raw loc 3, raw dim 3, raw ori 2)
};

// (the body is a function of the workgroup's size: frame_prologue_kernel runs it with 256 threads, frame_prologue_sample_kernel --
// prologue and the step's ray draw in ONE launch, see below -- with the 1024 of its sampling workgroup)
template <int kThreads>
__device__ __forceinline__ void frame_prologue_body(const FrameStepArgs& a, const FrameBuffers& b) {
    __shared__ double cost[kFrameMaxBoxes * kFrameMaxBoxes];
    __shared__ float corners[kFrameMaxBoxes * 24];
    __shared__ float gcorners[kFrameMaxBoxes * 24];
    __shared__ int match_gt[kFrameMaxBoxes];          // gt column of prediction n (every prediction is matched: the problem is square)
    __shared__ float partial[kThreads / kWave][3];
    const int tid = static_cast<int>(threadIdx.x);
    const int N = a.num_boxes, V = a.num_views;
#ifdef VSRD_PHASE_TIMERS
    const unsigned long long t_start = wall_clock64();                 // 100 MHz; mark k = time of the k-th barrier (tools/prologue_timers.py)
    int mark = 8;
#define VSRD_PROLOGUE_MARK() do { if (tid == 0 && mark < 16) g_phase_cycles[mark] = wall_clock64() - t_start; ++mark; } while (0)
#else
#define VSRD_PROLOGUE_MARK() do { } while (0)
#endif
    const int edges[kBoxEdges][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};   // main.py:26-30
    // What the phases hand each other -- 2-D boxes, which edge end gave each extreme, their gradients, the corner gradients of every
    // (view, box) -- goes through LDS while V N <= 256 (the reference's 17 views x 8 boxes: 136), through the global scratch above
    // that: a global store -> barrier -> load is a ~2 us round trip for this lone workgroup, and there were four of them.  (The corner
    // gradients reuse the cost matrix, which is dead once the matching is done.)
    __shared__ float small_boxes[kFrameThreads * 4];
    __shared__ int small_selection[kFrameThreads * 4];
    __shared__ float small_grad_boxes[kFrameThreads * 4];
    static_assert(sizeof(cost) >= sizeof(float) * kFrameThreads * 24, "the corner gradients of 256 (view, box) pairs fit the cost matrix");
    const bool small = V * N <= kFrameThreads;
    float* boxes_2d = small ? small_boxes : b.scratch;
    int* selection = small ? small_selection : reinterpret_cast<int*>(b.scratch + static_cast<size_t>(V) * N * 4);
    float* grad_boxes = small ? small_grad_boxes : b.scratch + static_cast<size_t>(V) * N * 8;
    float* grad_world = small ? reinterpret_cast<float*>(cost) : b.scratch + static_cast<size_t>(V) * N * 12;
    // the views' matrices (16 + 9 floats each) into LDS while the boxes are decoded: both projection phases then start without a global
    // round trip of their own
    __shared__ float view_matrices[kFrameMaxViews * 25];
    for (int idx = tid; idx < V * 25; idx += kThreads) {
        const int v = idx / 25, r = idx - 25 * v;
        view_matrices[idx] = r < 16 ? b.extrinsics[v * 16 + r] : b.intrinsics[v * 9 + (r - 16)];
    }
    // ---- schedules (main.py:420-431) ----------------------------------------------------------------------------------------
    if (tid == 0) {
        const float x = static_cast<float>(*b.step) / static_cast<float>(a.num_steps);
        const float anneal = (cosf(3.14159265358979323846f * x) + 1.0f) / 2.0f;
        b.schedule[0] = anneal * (a.max_temperature - a.min_temperature) + a.min_temperature;
        b.schedule[1] = anneal * (a.max_std - a.min_std) + a.min_std;
        b.schedule[2] = x;
    }
    // ---- decode, instance block, corners ----------------------------------------------------------------------------------------
    if (tid < N) {
        const DecodedBox d = decode_box(a, b.raw_locations + 3 * tid, b.raw_dimensions + 3 * tid, b.raw_orientations + 2 * tid);
        float* row = b.instances + 16 * tid;
        row[0] = d.loc[0]; row[1] = d.loc[1]; row[2] = d.loc[2];
        row[3] = d.c; row[4] = 0.0f; row[5] = d.s; row[6] = 0.0f; row[7] = 1.0f; row[8] = 0.0f; row[9] = -d.s; row[10] = 0.0f; row[11] = d.c;
        row[12] = d.dim[0]; row[13] = d.dim[1]; row[14] = d.dim[2]; row[15] = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float u[3];
            unit_corner(k, u);
            const float lx = u[0] * d.dim[0], ly = u[1] * d.dim[1], lz = u[2] * d.dim[2];
            corners[tid * 24 + 3 * k + 0] = d.c * lx + d.s * lz + d.loc[0];
            corners[tid * 24 + 3 * k + 1] = ly + d.loc[1];
            corners[tid * 24 + 3 * k + 2] = -d.s * lx + d.c * lz + d.loc[2];
        }
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    // ---- projection of every (view, box) (projection.h: project_boxes_kernel) -------------------------------------------------
    for (int idx = tid; idx < V * N; idx += kThreads) {
        const int v = idx / N, n = idx - v * N;
        const float* E = view_matrices + v * 25;
        const float* K = E + 16;
        float cam[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* p = corners + n * 24 + 3 * k;
            const float w = E[12] * p[0] + E[13] * p[1] + E[14] * p[2] + E[15];
#pragma unroll
            for (int j = 0; j < 3; ++j) cam[k][j] = (E[4 * j] * p[0] + E[4 * j + 1] * p[1] + E[4 * j + 2] * p[2] + E[4 * j + 3]) / w;
        }
        float lo_u = 0.0f, lo_v = 0.0f, hi_u = 0.0f, hi_v = 0.0f;
        int s_lo_u = -1, s_lo_v = -1, s_hi_u = -1, s_hi_v = -1;
        bool any = false;
#pragma unroll                                                   // (static corner indices: a rolled loop indexes cam[] dynamically, i.e. through scratch memory)
        for (int e = 0; e < kBoxEdges; ++e) {
            const ClippedEdge c = clip_edge(cam[edges[e][0]], cam[edges[e][1]], a.epsilon);
            if (!c.front) continue;
            const EdgePoint pf = project_point(K, c.fx, c.fy, c.fz, a.epsilon, nullptr, nullptr);
            const EdgePoint pn = project_point(K, c.nx, c.ny, c.nz, a.epsilon, nullptr, nullptr);
            if (!any) {
                lo_u = hi_u = pf.u; lo_v = hi_v = pf.v;
                s_lo_u = s_lo_v = s_hi_u = s_hi_v = 2 * e;
                any = true;
            } else {
                if (pf.u < lo_u) { lo_u = pf.u; s_lo_u = 2 * e; }
                if (pf.v < lo_v) { lo_v = pf.v; s_lo_v = 2 * e; }
                if (pf.u > hi_u) { hi_u = pf.u; s_hi_u = 2 * e; }
                if (pf.v > hi_v) { hi_v = pf.v; s_hi_v = 2 * e; }
            }
            if (pn.u < lo_u) { lo_u = pn.u; s_lo_u = 2 * e + 1; }
            if (pn.v < lo_v) { lo_v = pn.v; s_lo_v = 2 * e + 1; }
            if (pn.u > hi_u) { hi_u = pn.u; s_hi_u = 2 * e + 1; }
            if (pn.v > hi_v) { hi_v = pn.v; s_hi_v = 2 * e + 1; }
        }
        float out[4] = {lo_u, lo_v, hi_u, hi_v};
        int sel[4] = {s_lo_u, s_lo_v, s_hi_u, s_hi_v};
#pragma unroll
        for (int k = 0; k < 4; ++k) {                     // clip_boxes_to_image: a clamped coordinate has zero gradient
            const float limit = (k & 1) ? a.height : a.width;
            if (out[k] < 0.0f) { out[k] = 0.0f; sel[k] = -1; }
            if (out[k] > limit) { out[k] = limit; sel[k] = -1; }
            boxes_2d[idx * 4 + k] = out[k];
            selection[idx * 4 + k] = any ? sel[k] : -1;
        }
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    // ---- matching on the target view (matching.h) -------------------------------------------------------------------------------
    for (int idx = tid; idx < N * N; idx += kThreads) {
        const int p = idx / N, g = idx - p * N;
        float c = negative_distance_iou(boxes_2d + 4 * p, b.gt_boxes + 4 * g);
        if (!(c == c) || c < -3.0e38f || c > 3.0e38f) c = 3.0e38f;
        cost[p * N + g] = static_cast<double>(c);
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    if (tid < kWave) {                                     // wave 0 (all 64 lanes take part in the wave-cooperative solver)
        const int col4row = lsap_rows_le_cols(cost, N, N);
        if (tid < N) {
            match_gt[tid] = col4row;
            b.pd_indices[tid] = tid;
            b.gt_indices[tid] = col4row;
            b.target_map[tid] = col4row;
            b.instance_weights[tid] = 1.0f;
        }
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    // ---- projection losses over kept (view, matched pair)s and their gradient w.r.t. the predicted 2-D boxes (main.py:391-415) ----
    float iou_sum = 0.0f, l1_sum = 0.0f, kept = 0.0f;
    for (int idx = tid; idx < V * N; idx += kThreads) {
        const int v = idx / N, n = idx - v * N;
        const int g = match_gt[n];
        const bool keep = b.visible[v * N + g] != 0;
        float gi[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (keep) {
            const float* pd = boxes_2d + idx * 4;
            const float* gt = b.gt_boxes + (static_cast<size_t>(v) * N + g) * 4;
            iou_sum += diou_loss(pd, gt, gi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                 // smooth_l1_loss, beta = 1
                const float d = pd[k] - gt[k];
                const float ad = fabsf(d);
                l1_sum += (ad < 1.0f) ? 0.5f * d * d : ad - 0.5f;
                gl[k] = (ad < 1.0f) ? d : ((d > 0.0f) ? 1.0f : -1.0f);
            }
            kept += 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) grad_boxes[idx * 4 + k] = a.weight_iou * gi[k] + a.weight_l1 * gl[k] / 4.0f;   // (divided by the count below)
    }
    iou_sum = wave_sum(iou_sum); l1_sum = wave_sum(l1_sum); kept = wave_sum(kept);
    if ((tid & (kWave - 1)) == 0) { partial[tid >> 6][0] = iou_sum; partial[tid >> 6][1] = l1_sum; partial[tid >> 6][2] = kept; }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    float count = 0.0f, iou_total = 0.0f, l1_total = 0.0f;
    for (int w = 0; w < kThreads / kWave; ++w) { iou_total += partial[w][0]; l1_total += partial[w][1]; count += partial[w][2]; }
    count = fmaxf(count, 1.0f);
    if (tid == 0) { b.losses[0] = iou_total / count; b.losses[1] = l1_total / (count * 4.0f); }
    // ---- adjoint of the projection: 2-D box gradients -> world corners, per view (projection.h: project_boxes_backward_kernel) ----
    for (int idx = tid; idx < V * N; idx += kThreads) {
        const int v = idx / N, n = idx - v * N;
        const float* E = view_matrices + v * 25;
        const float* K = E + 16;
        float cam[8][3], hw[8], gcam[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* p = corners + n * 24 + 3 * k;
            hw[k] = E[12] * p[0] + E[13] * p[1] + E[14] * p[2] + E[15];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                cam[k][j] = (E[4 * j] * p[0] + E[4 * j + 1] * p[1] + E[4 * j + 2] * p[2] + E[4 * j + 3]) / hw[k];
                gcam[k][j] = 0.0f;
            }
        }
        // Which corners an extreme came from is data: cam[] / gcam[] indexed by it would live in scratch memory (a ~2 us round trip per
        // access from this lone workgroup: 14 of the kernel's 42 us).  The two corners of the selected edge are picked out of the
        // registers by compare-and-select chains instead, and their gradients go back the same way.
        constexpr unsigned kEdgeFirst = 0x76543210u, kEdgeSecondLow = 0x47650321u;       // edges[e][0] / edges[e][1] of e = 0 .. 7, 4 bits each
        constexpr unsigned kEdgeFirstHigh = 0x3210u, kEdgeSecondHigh = 0x7654u;          // e = 8 .. 11
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sel = selection[idx * 4 + k];
            const float g = grad_boxes[idx * 4 + k] / count;
            if (sel < 0 || g == 0.0f) continue;
            const int e = sel >> 1;
            const bool near_end = (sel & 1) != 0;
            const int ia = static_cast<int>(((e < 8 ? kEdgeFirst : kEdgeFirstHigh) >> (4 * (e & 7))) & 7u);
            const int ib = static_cast<int>(((e < 8 ? kEdgeSecondLow : kEdgeSecondHigh) >> (4 * (e & 7))) & 7u);
            float pa[3], pb[3];
            pick_corner(cam, ia, pa);                                  // (projection.h)
            pick_corner(cam, ib, pb);
            const ClippedEdge c = clip_edge(pa, pb, a.epsilon);
            const int i_far = c.a_is_far ? ia : ib, i_near = c.a_is_far ? ib : ia;
            const float px = near_end ? c.nx : c.fx, py = near_end ? c.ny : c.fy, pz = near_end ? c.nz : c.fz;
            float w; bool w_clamped;
            const EdgePoint pt = project_point(K, px, py, pz, a.epsilon, &w, &w_clamped);
            const int r = k & 1;
            const float coord = r ? pt.v : pt.u;
            float gp[3], to_far[3], to_near[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) gp[j] = g * (K[3 * r + j] - (w_clamped ? 0.0f : coord * K[6 + j])) / w;
            if (!near_end) {
#pragma unroll
                for (int j = 0; j < 3; ++j) to_far[j] = gp[j];
            } else {
                float gt = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float far_j = c.a_is_far ? pa[j] : pb[j], near_j = c.a_is_far ? pb[j] : pa[j];
                    to_far[j] = gp[j] * (1.0f - c.t);
                    to_near[j] = gp[j] * c.t;
                    gt += gp[j] * (near_j - far_j);
                }
                if (!c.t_clamped) {
                    const float zf = c.a_is_far ? pa[2] : pb[2], zn = c.a_is_far ? pb[2] : pa[2];
                    const float den = c.den_clamped ? a.epsilon : (zf - zn);
                    float dzf = 1.0f / den, dzn = 0.0f;
                    if (!c.den_clamped) { dzf -= zf / (den * den); dzn = zf / (den * den); }
                    to_far[2] += gt * dzf;
                    to_near[2] += gt * dzn;
                }
            }
            add_to_corner(gcam, i_far, to_far);
            add_to_corner(gcam, i_near, to_near);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float* o = grad_world + (static_cast<size_t>(idx) * 8 + k) * 3;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; ++j) acc += gcam[k][j] * (E[4 * j + m] - cam[k][j] * E[12 + m]);
                o[m] = acc / hw[k];
            }
        }
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    // ---- sum over the views in a fixed order, then corners -> decoded parameters -> raw parameters ------------------------------
    for (int idx = tid; idx < N * 24; idx += kThreads) {
        float acc = 0.0f;
        for (int v = 0; v < V; ++v) acc += grad_world[(static_cast<size_t>(v) * N) * 24 + idx];
        gcorners[idx] = acc;
    }
    __syncthreads();
    VSRD_PROLOGUE_MARK();
    if (tid < N) {
        const DecodedBox d = decode_box(a, b.raw_locations + 3 * tid, b.raw_dimensions + 3 * tid, b.raw_orientations + 2 * tid);
        float g_loc[3] = {0.0f, 0.0f, 0.0f}, g_dim[3] = {0.0f, 0.0f, 0.0f}, g_c = 0.0f, g_s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float u[3];
            unit_corner(k, u);
            const float gx = gcorners[tid * 24 + 3 * k], gy = gcorners[tid * 24 + 3 * k + 1], gz = gcorners[tid * 24 + 3 * k + 2];
            const float lx = u[0] * d.dim[0], lz = u[2] * d.dim[2];
            g_loc[0] += gx; g_loc[1] += gy; g_loc[2] += gz;
            // corner = (c lx + s lz, ly, -s lx + c lz) + loc
            g_dim[0] += u[0] * (d.c * gx - d.s * gz);
            g_dim[1] += u[1] * gy;
            g_dim[2] += u[2] * (d.s * gx + d.c * gz);
            g_c += gx * lx + gz * lz;
            g_s += gx * lz - gz * lx;
        }
        float* out = b.grad_raw + 8 * tid;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            out[j] = g_loc[j] * (a.location_hi[j] - a.location_lo[j]) * d.sig_l[j] * (1.0f - d.sig_l[j]);
            out[3 + j] = g_dim[j] * (a.dimension_hi[j] - a.dimension_lo[j]) * d.sig_d[j] * (1.0f - d.sig_d[j]);
        }
        const float dot = d.c * g_c + d.s * g_s;                  // (c, s) = raw / |raw|: the tangential part, over |raw|
        out[6] = (g_c - d.c * dot) / d.norm;
        out[7] = (g_s - d.s * dot) / d.norm;
    }
    VSRD_PROLOGUE_MARK();
}
#undef VSRD_PROLOGUE_MARK

// Frame f of a batch (wave.h): the pointer block moves as a whole.
__device__ __forceinline__ void shift_frame(FrameBuffers& b, long long shift) {
    b.raw_locations = of_frame(b.raw_locations, shift); b.raw_dimensions = of_frame(b.raw_dimensions, shift); b.raw_orientations = of_frame(b.raw_orientations, shift);
    b.extrinsics = of_frame(b.extrinsics, shift); b.intrinsics = of_frame(b.intrinsics, shift); b.gt_boxes = of_frame(b.gt_boxes, shift);
    b.visible = of_frame(b.visible, shift); b.step = of_frame(b.step, shift); b.scratch = of_frame(b.scratch, shift); b.instances = of_frame(b.instances, shift);
    b.pd_indices = of_frame(b.pd_indices, shift); b.gt_indices = of_frame(b.gt_indices, shift); b.target_map = of_frame(b.target_map, shift);
    b.instance_weights = of_frame(b.instance_weights, shift); b.schedule = of_frame(b.schedule, shift); b.losses = of_frame(b.losses, shift);
    b.grad_raw = of_frame(b.grad_raw, shift);
}

__global__ __launch_bounds__(kFrameThreads) void frame_prologue_kernel(FrameStepArgs a, FrameBuffers b) {
    if (a.frame_stride != 0) shift_frame(b, frame_shift(a.frame_stride, blockIdx.y));
    frame_prologue_body<kFrameThreads>(a, b);
}

// The prologue and the draw of the step's rays from the frame's sampling table (ray_sampling.h) in ONE launch of two workgroups: neither
// needs the other, and as two launches on two streams they met again through a cross-queue dependency of ~10 us.
__global__ __launch_bounds__(kTableThreads) void frame_prologue_sample_kernel(FrameStepArgs a, FrameBuffers b, RayTableHeader* table, long long count, int num_rays,
                                                                              unsigned long long seed, const unsigned long long* __restrict__ device_step,
                                                                              const long long* __restrict__ remap, long long* __restrict__ ray_indices) {
    if (a.frame_stride != 0) {             // a batch of frames (wave.h): two workgroups per frame, frame blockIdx.y
        const long long shift = frame_shift(a.frame_stride, blockIdx.y);
        shift_frame(b, shift);
        VSRD_OF_FRAME(table, shift); VSRD_OF_FRAME(device_step, shift); VSRD_OF_FRAME(remap, shift); VSRD_OF_FRAME(ray_indices, shift);
    }
    if (blockIdx.x == 0) frame_prologue_body<kTableThreads>(a, b);
    else sample_table_body(table, count, num_rays, seed, 0ull, device_step, remap, ray_indices);
}

struct AdamTensors {                   // torch.optim.Adam(capturable=True) state of one parameter tensor, all on the device
    float* parameter;
    float* exp_avg;
    float* exp_avg_sq;
    float* step;                       // float32 scalar
    float* learning_rate;              // float32 scalar (decayed in place: ExponentialLR)
};

struct EpilogueBuffers {
    const float* grad_instances;       // [N,16] from the render adjoint: d (silhouette [+ eikonal]) / d (loc 3 | R 9 | dim 3)
    const float* grad_raw_projection;  // [N,8] from the prologue
    const float* projection_losses;    // [2]
    const float* render_losses;        // [2]: silhouette, eikonal (eikonal 0 for box-only fields)
    float eikonal_ratio;               // weight of the eikonal term relative to the silhouette term (already inside grad_instances)
    AdamTensors locations, dimensions, orientations;
    float* other_learning_rates[2];    // embeddings / hypernetwork groups: decayed here as well (may be null)
    long long* step;                   // device step counter, incremented
    float* record;                     // out [5]: iou, l1, silhouette, eikonal, total
    float* raw_gradients;              // out [N,8] (diagnostics / tests)
};

// (old_m, old_v, old_p: the moments and the parameter as the caller read them -- together with everything else, in one round trip)
__device__ __forceinline__ void adam_update(const FrameStepArgs& a, const AdamTensors& t, int index, float grad, float step_new, float lr,
                                            float old_m, float old_v, float old_p) {
    const float m = old_m + (grad - old_m) * (1.0f - a.beta1);                                    // lerp_(grad, 1 - beta1)
    const float v = old_v * a.beta2 + (1.0f - a.beta2) * grad * grad;
    t.exp_avg[index] = m;
    t.exp_avg_sq[index] = v;
    const float bc1 = 1.0f - powf(a.beta1, step_new), bc2 = 1.0f - powf(a.beta2, step_new);
    t.parameter[index] = old_p - (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + a.adam_epsilon);
}

__device__ __forceinline__ void shift_frame(AdamTensors& t, long long shift) {
    t.parameter = of_frame(t.parameter, shift); t.exp_avg = of_frame(t.exp_avg, shift); t.exp_avg_sq = of_frame(t.exp_avg_sq, shift);
    t.step = of_frame(t.step, shift); t.learning_rate = of_frame(t.learning_rate, shift);
}
__device__ __forceinline__ void shift_frame(EpilogueBuffers& e, long long shift) {
    e.grad_instances = of_frame(e.grad_instances, shift); e.grad_raw_projection = of_frame(e.grad_raw_projection, shift);
    e.projection_losses = of_frame(e.projection_losses, shift); e.render_losses = of_frame(e.render_losses, shift);
    shift_frame(e.locations, shift); shift_frame(e.dimensions, shift); shift_frame(e.orientations, shift);
    e.other_learning_rates[0] = of_frame(e.other_learning_rates[0], shift); e.other_learning_rates[1] = of_frame(e.other_learning_rates[1], shift);
    e.step = of_frame(e.step, shift); e.record = of_frame(e.record, shift); e.raw_gradients = of_frame(e.raw_gradients, shift);
}

__global__ __launch_bounds__(kFrameMaxBoxes) void frame_epilogue_kernel(FrameStepArgs a, EpilogueBuffers e) {
    if (a.frame_stride != 0) shift_frame(e, frame_shift(a.frame_stride, blockIdx.y));      // a batch of frames (wave.h): one workgroup per frame
    const int tid = static_cast<int>(threadIdx.x);
    const int N = a.num_boxes;
    // every thread reads the scalars before anyone updates them -- and with them everything else it will need (its box's raw
    // parameters, their Adam moments, the two gradient rows; thread 0 the step counter, the losses, the other groups' rates): ONE global
    // round trip of ~2 us for this lone workgroup instead of four in a row
    const float step_l = *e.locations.step + 1.0f, step_d = *e.dimensions.step + 1.0f, step_o = *e.orientations.step + 1.0f;
    const float lr_l = *e.locations.learning_rate, lr_d = *e.dimensions.learning_rate, lr_o = *e.orientations.learning_rate;
    const int box = tid < N ? tid : 0;
    float raw_loc[3], raw_dim[3], raw_ori[2], m_loc[3], v_loc[3], m_dim[3], v_dim[3], m_ori[2], v_ori[2], gi[16], projection[8];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        raw_loc[j] = e.locations.parameter[3 * box + j]; m_loc[j] = e.locations.exp_avg[3 * box + j]; v_loc[j] = e.locations.exp_avg_sq[3 * box + j];
        raw_dim[j] = e.dimensions.parameter[3 * box + j]; m_dim[j] = e.dimensions.exp_avg[3 * box + j]; v_dim[j] = e.dimensions.exp_avg_sq[3 * box + j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        raw_ori[j] = e.orientations.parameter[2 * box + j]; m_ori[j] = e.orientations.exp_avg[2 * box + j]; v_ori[j] = e.orientations.exp_avg_sq[2 * box + j];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) gi[k] = e.grad_instances[16 * box + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) projection[k] = e.grad_raw_projection[8 * box + k];
    float other_lr[2] = {0.0f, 0.0f};
    long long step_counter = 0;
    float iou = 0.0f, l1 = 0.0f, sil = 0.0f, eik = 0.0f;
    if (tid == 0) {
        for (int k = 0; k < 2; ++k) other_lr[k] = e.other_learning_rates[k] ? *e.other_learning_rates[k] : 0.0f;
        step_counter = *e.step;
        iou = e.projection_losses[0]; l1 = e.projection_losses[1]; sil = e.render_losses[0]; eik = e.render_losses[1];
    }
    __syncthreads();
    if (tid < N) {
        const DecodedBox d = decode_box(a, raw_loc, raw_dim, raw_ori);
        float grads[8];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            grads[j] = gi[j] * (a.location_hi[j] - a.location_lo[j]) * d.sig_l[j] * (1.0f - d.sig_l[j]);
            grads[3 + j] = gi[12 + j] * (a.dimension_hi[j] - a.dimension_lo[j]) * d.sig_d[j] * (1.0f - d.sig_d[j]);
        }
        // R = [[c, 0, s], [0, 1, 0], [-s, 0, c]] (row-major 3..11 of the instance row)
        const float g_c = gi[3] + gi[11], g_s = gi[5] - gi[9];
        const float dot = d.c * g_c + d.s * g_s;
        grads[6] = (g_c - d.c * dot) / d.norm;
        grads[7] = (g_s - d.s * dot) / d.norm;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            grads[k] = a.weight_silhouette * grads[k] + projection[k];
            if (e.raw_gradients) e.raw_gradients[8 * tid + k] = grads[k];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            adam_update(a, e.locations, 3 * tid + j, grads[j], step_l, lr_l, m_loc[j], v_loc[j], raw_loc[j]);
            adam_update(a, e.dimensions, 3 * tid + j, grads[3 + j], step_d, lr_d, m_dim[j], v_dim[j], raw_dim[j]);
        }
        adam_update(a, e.orientations, 2 * tid + 0, grads[6], step_o, lr_o, m_ori[0], v_ori[0], raw_ori[0]);
        adam_update(a, e.orientations, 2 * tid + 1, grads[7], step_o, lr_o, m_ori[1], v_ori[1], raw_ori[1]);
    }
    if (tid == 0) {
        *e.locations.step = step_l; *e.dimensions.step = step_d; *e.orientations.step = step_o;
        *e.locations.learning_rate = lr_l * a.lr_gamma; *e.dimensions.learning_rate = lr_d * a.lr_gamma; *e.orientations.learning_rate = lr_o * a.lr_gamma;
        for (int k = 0; k < 2; ++k)
            if (e.other_learning_rates[k]) *e.other_learning_rates[k] = other_lr[k] * a.lr_gamma;
        *e.step = step_counter + 1;
        e.record[0] = iou; e.record[1] = l1; e.record[2] = sil; e.record[3] = eik;
        e.record[4] = a.weight_iou * iou + a.weight_l1 * l1 + a.weight_silhouette * (sil + e.eikonal_ratio * eik);
    }
}

}  // namespace vsrd
