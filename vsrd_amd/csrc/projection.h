// Multi-view projection of 3-D boxes to clipped 2-D boxes, forward and adjoint, one thread per (view, box).
//
// Reference semantics (CPU restatement: oracle/geometry.py project_boxes*):
//   world -> camera:  einsum("bmn,b...n->b...m", E, [corner, 1]) and the divide by w       scripts/main.py:341-344
//   clip_lines_to_front (swap so point 1 is the deeper end, pull the nearer end to z = 0+)   geometric_operations.py:343-365
//   project_box_3d (@ K^T, divide by clamp(z, eps), min/max over edges whose far end has z>0) geometric_operations.py:368-389
//   torchvision.ops.clip_boxes_to_image                                                      scripts/main.py:359-362
//
// The reference calls project_box_3d V*N times from Python with a host sync each (`torch.any`, :376); here all
// V*N boxes are one launch.  The adjoint follows torch's min/max backward: the gradient of each extreme goes to the
// first edge end point that attains it (recorded by the forward as edge*2 + endpoint).
#pragma once
#include "wave.h"

namespace vsrd {

constexpr int kMaxEdges = 32;

struct EdgePoint { float u, v; };

struct ClippedEdge {
    float fx, fy, fz;   // far (deeper) end, camera frame
    float nx, ny, nz;   // near end after clipping
    float t;            // interpolation weight actually used
    bool a_is_far;      // which original corner is the far end
    bool t_clamped;     // t hit the upper clamp 1 (near end untouched)
    bool den_clamped;   // (zf - zn) hit the lower clamp eps
    bool front;         // far end has z > 0
};

__device__ __forceinline__ ClippedEdge clip_edge(const float* a, const float* b, float eps) {
    ClippedEdge e;
    e.a_is_far = a[2] > b[2];
    const float* f = e.a_is_far ? a : b;
    const float* n = e.a_is_far ? b : a;
    e.fx = f[0]; e.fy = f[1]; e.fz = f[2];
    const float den_raw = f[2] - n[2];
    e.den_clamped = den_raw < eps;
    const float den = e.den_clamped ? eps : den_raw;
    const float t_raw = f[2] / den;
    e.t_clamped = t_raw > 1.0f;
    e.t = e.t_clamped ? 1.0f : t_raw;
    e.nx = f[0] + (n[0] - f[0]) * e.t;
    e.ny = f[1] + (n[1] - f[1]) * e.t;
    e.nz = f[2] + (n[2] - f[2]) * e.t;
    e.front = f[2] > 0.0f;
    return e;
}

__device__ __forceinline__ EdgePoint project_point(const float* K, float x, float y, float z, float eps, float* w_out, bool* w_clamped) {
    const float pu = K[0] * x + K[1] * y + K[2] * z;
    const float pv = K[3] * x + K[4] * y + K[5] * z;
    const float pw = K[6] * x + K[7] * y + K[8] * z;
    const bool clamped = pw < eps;
    const float w = clamped ? eps : pw;
    if (w_out) *w_out = w;
    if (w_clamped) *w_clamped = clamped;
    return {pu / w, pv / w};
}

// Corner `index` (data, 0..7) of a per-thread array of eight: register arrays indexed by data live in scratch memory (a memory round
// trip per access), so the corner is picked / added to by compare-and-select chains over the eight instead.
__device__ __forceinline__ void pick_corner(const float (&corners)[8][3], int index, float (&out)[3]) {
    out[0] = out[1] = out[2] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = (c == index) ? corners[c][j] : out[j];
    }
}
__device__ __forceinline__ void add_to_corner(float (&corners)[8][3], int index, const float (&v)[3]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) corners[c][j] += (c == index) ? v[j] : 0.0f;
    }
}

// world corners [N,8,3], extrinsics [V,16], intrinsics [V,9], edges [E,2] ->
//   boxes_2d [V,N,4] (x1,y1,x2,y2, clipped to the image), camera_corners [V,N,8,3] (optional), selection [V,N,4] (int32).
__global__ __launch_bounds__(256) void project_boxes_kernel(
    const float* __restrict__ world_corners, const float* __restrict__ extrinsics, const float* __restrict__ intrinsics,
    const int* __restrict__ edges, int num_edges, int num_views, int num_boxes, float height, float width, float eps,
    float* __restrict__ boxes_2d, float* __restrict__ camera_corners, int* __restrict__ selection) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= num_views * num_boxes) return;
    const int v = idx / num_boxes, n = idx - v * num_boxes;
    const float* E = extrinsics + v * 16;
    const float* K = intrinsics + v * 9;
    float cam[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float* p = world_corners + (n * 8 + k) * 3;
        const float w = E[12] * p[0] + E[13] * p[1] + E[14] * p[2] + E[15];
#pragma unroll
        for (int j = 0; j < 3; ++j) cam[k][j] = (E[4 * j] * p[0] + E[4 * j + 1] * p[1] + E[4 * j + 2] * p[2] + E[4 * j + 3]) / w;
        if (camera_corners) {
            float* o = camera_corners + (static_cast<size_t>(idx) * 8 + k) * 3;
            o[0] = cam[k][0]; o[1] = cam[k][1]; o[2] = cam[k][2];
        }
    }
    float lo_u = 0.0f, lo_v = 0.0f, hi_u = 0.0f, hi_v = 0.0f;
    int s_lo_u = -1, s_lo_v = -1, s_hi_u = -1, s_hi_v = -1;
    bool any = false;
    for (int e = 0; e < num_edges; ++e) {
        float pa[3], pb[3];
        pick_corner(cam, edges[2 * e], pa);
        pick_corner(cam, edges[2 * e + 1], pb);
        const ClippedEdge c = clip_edge(pa, pb, eps);
        if (!c.front) continue;
        const EdgePoint pf = project_point(K, c.fx, c.fy, c.fz, eps, nullptr, nullptr);
        const EdgePoint pn = project_point(K, c.nx, c.ny, c.nz, eps, nullptr, nullptr);
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
    // clip_boxes_to_image; a coordinate outside [0, size] has zero gradient (torch.clamp backward): selection -1
    float out[4] = {lo_u, lo_v, hi_u, hi_v};
    int sel[4] = {s_lo_u, s_lo_v, s_hi_u, s_hi_v};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float limit = (k & 1) ? height : width;
        if (limit > 0.0f) {                       // height/width <= 0: plain project_box_3d, no image clamp
            if (out[k] < 0.0f) { out[k] = 0.0f; sel[k] = -1; }
            if (out[k] > limit) { out[k] = limit; sel[k] = -1; }
        }
        boxes_2d[idx * 4 + k] = out[k];
        selection[idx * 4 + k] = any ? sel[k] : -1;
    }
}

// grad_boxes_2d [V,N,4] -> grad_world_corners_per_view [V,N,8,3] (summed over V by the caller: deterministic).
__global__ __launch_bounds__(256) void project_boxes_backward_kernel(
    const float* __restrict__ world_corners, const float* __restrict__ extrinsics, const float* __restrict__ intrinsics,
    const int* __restrict__ edges, int num_views, int num_boxes, float eps,
    const float* __restrict__ grad_boxes_2d, const int* __restrict__ selection, float* __restrict__ grad_world_per_view) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= num_views * num_boxes) return;
    const int v = idx / num_boxes, n = idx - v * num_boxes;
    const float* E = extrinsics + v * 16;
    const float* K = intrinsics + v * 9;
    float cam[8][3], hw[8];
    float gcam[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float* p = world_corners + (n * 8 + k) * 3;
        hw[k] = E[12] * p[0] + E[13] * p[1] + E[14] * p[2] + E[15];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cam[k][j] = (E[4 * j] * p[0] + E[4 * j + 1] * p[1] + E[4 * j + 2] * p[2] + E[4 * j + 3]) / hw[k];
            gcam[k][j] = 0.0f;
        }
    }
    for (int k = 0; k < 4; ++k) {
        const int sel = selection[idx * 4 + k];
        const float g = grad_boxes_2d[idx * 4 + k];
        if (sel < 0 || g == 0.0f) continue;
        const int e = sel >> 1;
        const bool near_end = (sel & 1) != 0;
        const int ia = edges[2 * e], ib = edges[2 * e + 1];
        float pa[3], pb[3];
        pick_corner(cam, ia, pa);
        pick_corner(cam, ib, pb);
        const ClippedEdge c = clip_edge(pa, pb, eps);
        const int i_far = c.a_is_far ? ia : ib, i_near = c.a_is_far ? ib : ia;
        const float px = near_end ? c.nx : c.fx, py = near_end ? c.ny : c.fy, pz = near_end ? c.nz : c.fz;
        float w; bool w_clamped;
        const EdgePoint pt = project_point(K, px, py, pz, eps, &w, &w_clamped);
        // d(coordinate)/d(point): row r of K (r = 0 for u, 1 for v) minus coordinate * row 2, over w
        const int r = k & 1;
        const float coord = r ? pt.v : pt.u;
        float gp[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gp[j] = g * (K[3 * r + j] - (w_clamped ? 0.0f : coord * K[6 + j])) / w;
        float to_far[3], to_near[3] = {0.0f, 0.0f, 0.0f};
        if (!near_end) {
#pragma unroll
            for (int j = 0; j < 3; ++j) to_far[j] = gp[j];
        } else {
            // near' = far + (near - far) * t
            float gt = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float far_j = c.a_is_far ? pa[j] : pb[j], near_j = c.a_is_far ? pb[j] : pa[j];
                to_far[j] = gp[j] * (1.0f - c.t);
                to_near[j] = gp[j] * c.t;
                gt += gp[j] * (near_j - far_j);
            }
            if (!c.t_clamped) {              // t = zf / den
                const float zf = c.a_is_far ? pa[2] : pb[2], zn = c.a_is_far ? pb[2] : pa[2];
                const float den = c.den_clamped ? eps : (zf - zn);
                float dzf = 1.0f / den, dzn = 0.0f;
                if (!c.den_clamped) { dzf -= zf / (den * den); dzn = zf / (den * den); }
                to_far[2] += gt * dzf;
                to_near[2] += gt * dzn;
            }
        }
        add_to_corner(gcam, i_far, to_far);
        add_to_corner(gcam, i_near, to_near);
    }
    // camera -> world:  c_j = (E_j . ph) / w  =>  dc_j/dp_k = (E[j][k] - c_j E[3][k]) / w
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float* o = grad_world_per_view + (static_cast<size_t>(idx) * 8 + k) * 3;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += gcam[k][j] * (E[4 * j + m] - cam[k][j] * E[12 + m]);
            o[m] = acc / hw[k];
        }
    }
}

}  // namespace vsrd
