// Hungarian matching on the device: predicted vs ground-truth 2-D boxes of the target view.
//
// Reference semantics:
//   scripts/main.py:374-386   cost = -torchvision.ops.distance_box_iou(pd, gt);  scipy.optimize.linear_sum_assignment(cost.cpu())
//   scipy 1.x rectangular_lsap.cpp (Crouse, "On implementing 2D rectangular assignment algorithms", 2016): shortest augmenting
//   paths with dual variables in float64, the remaining columns visited in scipy's (reversed, swap-removed) order and its
//   tie rule ("among equal minima prefer a column that is still free"), so that degenerate cost matrices (identical boxes,
//   constant costs) give the SAME assignment as the reference; the tests compare with scipy.optimize.linear_sum_assignment itself.
// The reference pays one device-to-host synchronisation per optimisation step for this; here it is one single-wave launch,
// which is what lets the whole step be captured in a hipGraph.
//
// One wave, lane j = column j (and row j for the row-indexed state); P, G <= 64.  ~n^2 wave-steps of a 64-lane argmin.
#pragma once
#include "wave.h"

namespace vsrd {

constexpr double kLsapInfinity = 1.0e300;

__device__ __forceinline__ double shuffle_double(double v, int src_lane) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __shfl(static_cast<int>(bits), src_lane, kWave);
    const int hi = __shfl(static_cast<int>(bits >> 32), src_lane, kWave);
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// Lane `src_lane` (wave-uniform) of an int / a double: v_readlane with a scalar lane index instead of a ds_bpermute round trip.
__device__ __forceinline__ int read_lane_int(int v, int src_lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src_lane)); }
__device__ __forceinline__ double read_lane_double(double v, int src_lane) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = read_lane_int(static_cast<int>(bits), src_lane), hi = read_lane_int(static_cast<int>(bits >> 32), src_lane);
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// Wave-wide minimum / maximum by DPP (quad swaps, row mirrors, row broadcasts; the result is read from lane 63): the solver below runs
// ~n^2 / 2 of each, and through __shfl (ds_bpermute) one 64-bit minimum alone was 12 LDS round trips.
template <int kCtrl, int kRowMask = 0xf>
__device__ __forceinline__ int dpp_int(int v) { return __builtin_amdgcn_update_dpp(v, v, kCtrl, kRowMask, 0xf, false); }
template <int kCtrl, int kRowMask = 0xf>
__device__ __forceinline__ double dpp_double(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = dpp_int<kCtrl, kRowMask>(static_cast<int>(bits)), hi = dpp_int<kCtrl, kRowMask>(static_cast<int>(bits >> 32));
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
// kSmall: every lane that matters sits in the first row of 16 (the others hold the neutral element): four steps, read from lane 0.
template <bool kSmall = false>
__device__ __forceinline__ double wave_min_double(double v) {
    v = fmin(v, dpp_double<kDppQuadXor1>(v));
    v = fmin(v, dpp_double<kDppQuadXor2>(v));
    v = fmin(v, dpp_double<kDppRowHalfMirror>(v));
    v = fmin(v, dpp_double<kDppRowMirror>(v));
    if (kSmall) return read_lane_double(v, 0);
    v = fmin(v, dpp_double<kDppRowBcast15, 0xa>(v));
    v = fmin(v, dpp_double<kDppRowBcast31, 0xc>(v));
    return read_lane_double(v, 63);
}
template <bool kSmall = false>
__device__ __forceinline__ int wave_max_int(int v) {
    v = max(v, dpp_int<kDppQuadXor1>(v));
    v = max(v, dpp_int<kDppQuadXor2>(v));
    v = max(v, dpp_int<kDppRowHalfMirror>(v));
    v = max(v, dpp_int<kDppRowMirror>(v));
    if (kSmall) return read_lane_int(v, 0);
    v = max(v, dpp_int<kDppRowBcast15, 0xa>(v));
    v = max(v, dpp_int<kDppRowBcast31, 0xc>(v));
    return read_lane_int(v, 63);
}

// -DIoU of predicted box i and ground-truth box j (vsrd_amd/losses.py::distance_box_iou, torchvision 0.14 formula).
__device__ __forceinline__ float negative_distance_iou(const float* a, const float* b) {
    const float x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3];
    const float x1g = b[0], y1g = b[1], x2g = b[2], y2g = b[3];
    const float inter = fmaxf(fminf(x2, x2g) - fmaxf(x1, x1g), 0.0f) * fmaxf(fminf(y2, y2g) - fmaxf(y1, y1g), 0.0f);
    const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
    const float w = fmaxf(fmaxf(x2, x2g) - fminf(x1, x1g), 0.0f), h = fmaxf(fmaxf(y2, y2g) - fminf(y1, y1g), 0.0f);
    const float diagonal = w * w + h * h + 1.0e-7f;
    const float cx = (x1 + x2) - (x1g + x2g), cy = (y1 + y2) - (y1g + y2g);
    const float centres = cx * cx / 4.0f + cy * cy / 4.0f;
    return -(inter / uni - centres / diagonal);
}

// Solve the nr x nc (nr <= nc <= 64) assignment problem on `cost` (LDS, row-major with pitch nc, float64).
// Returns col4row of row `lane` (valid for lane < nr).
template <bool kSmall>
__device__ __forceinline__ int lsap_rows_le_cols_impl(const double* cost, int nr, int nc) {
    const int lane = lane_id();
    // column state (lane = j)
    double v = 0.0, spc = kLsapInfinity;
    int path = -1, row4col = -1, position = 0;
    bool removed = false;
    // row state (lane = i)
    double u = 0.0;
    int col4row = -1;
    for (int cur_row = 0; cur_row < nr; ++cur_row) {
        // ---- augmenting path from cur_row ----------------------------------------------------------------------
        double min_val = 0.0;
        int num_remaining = nc;
        position = nc - lane - 1;                     // remaining[it] = nc - it - 1  <=>  column j sits at it = nc - j - 1
        removed = lane >= nc;
        bool in_sr = false;
        spc = kLsapInfinity;
        int i = cur_row, sink = -1;
        while (sink == -1) {
            if (lane == i) in_sr = true;
            const double ui = read_lane_double(u, i);
            if (!removed) {
                const double r = min_val + cost[i * nc + lane] - ui - v;
                if (r < spc) { path = i; spc = r; }
            }
            // scipy's sequential scan over remaining[0..num_remaining): strictly lower wins; on a tie a FREE column wins (the last
            // free one in scan order); otherwise the first column in scan order keeps the lead
            const double lowest = wave_min_double<kSmall>(removed ? kLsapInfinity : spc);
            if (!(lowest < kLsapInfinity)) return -2;                 // infeasible (never for finite costs)
            const bool at_min = !removed && spc == lowest;
            // one reduction for both rules: a free column at the minimum scores 64 + position (the last free one wins), any other
            // column at the minimum 63 - position (the first one wins, and only if no free one is there)
            const int score = wave_max_int<kSmall>(at_min ? ((row4col == -1) ? 64 + position : 63 - position) : -1);
            const int index = (score >= 64) ? score - 64 : 63 - score;
            min_val = lowest;
            const unsigned long long chosen = __ballot(!removed && position == index);
            const int j = __ffsll(static_cast<long long>(chosen)) - 1;
            const int owner = read_lane_int(row4col, j);
            if (owner == -1) sink = j; else i = owner;
            // SC[j] = true; remaining[index] = remaining[--num_remaining]
            --num_remaining;
            if (!removed && position == num_remaining) position = index;    // the last remaining column moves into the hole
            if (lane == j) removed = true;
        }
        // ---- dual updates -------------------------------------------------------------------------------------------
        {
            const int safe = (col4row >= 0) ? col4row : 0;
            const double spc_of_mine = shuffle_double(spc, safe);
            if (lane == cur_row) u += min_val;
            else if (in_sr && lane < nr) u += min_val - spc_of_mine;
            if (removed && lane < nc) v -= min_val - spc;
        }
        // ---- augment along the path ---------------------------------------------------------------------------------------
        int j = sink;
        while (true) {
            const int pi = read_lane_int(path, j);
            if (lane == j) row4col = pi;
            const int previous = read_lane_int(col4row, pi);
            if (lane == pi) col4row = j;
            j = previous;
            if (pi == cur_row) break;
        }
    }
    return col4row;
}

__device__ __forceinline__ int lsap_rows_le_cols(const double* cost, int nr, int nc) {
    return nc <= 16 ? lsap_rows_le_cols_impl<true>(cost, nr, nc) : lsap_rows_le_cols_impl<false>(cost, nr, nc);
}

// cost [P,G] (float, row-major) or boxes -> matched (pd, gt) index pairs sorted by pd index, as scipy returns them.
__global__ __launch_bounds__(kWave) void match_kernel(const float* __restrict__ cost_in, const float* __restrict__ pd_boxes, const float* __restrict__ gt_boxes,
                                                      int P, int G, long long* __restrict__ pd_indices, long long* __restrict__ gt_indices) {
    __shared__ double cost[64 * 64];
    const int lane = lane_id();
    const bool transposed = G < P;                      // scipy transposes when there are more rows than columns
    const int nr = transposed ? G : P, nc = transposed ? P : G;
    for (int idx = lane; idx < P * G; idx += kWave) {
        const int p = idx / G, g = idx % G;
        float c = cost_in ? cost_in[idx] : negative_distance_iou(pd_boxes + 4 * p, gt_boxes + 4 * g);
        // scipy raises on NaN / -inf entries and on an infeasible (+inf) matrix; a kernel cannot, and an unsolved problem would leave
        // the index buffers undefined (they are used for gathers right after): NaN and +-inf entries become finite "never match"
        // costs, so the assignment always is a valid permutation (and equals scipy's wherever scipy has one that avoids them)
        if (!(c == c) || c < -3.0e38f || c > 3.0e38f) c = 3.0e38f;
        cost[transposed ? (g * nc + p) : (p * nc + g)] = static_cast<double>(c);
    }
    __syncthreads();
    const int col4row = lsap_rows_le_cols(cost, nr, nc);
    if (!transposed) {
        if (lane < nr) { pd_indices[lane] = lane; gt_indices[lane] = col4row; }
    } else {
        // rows of the transposed problem are gt boxes: pairs (pd = col4row[g], gt = g), to be listed by increasing pd
        const int pd = (lane < nr) ? col4row : 0x7fffffff;
        int rank = 0;
        for (int other = 0; other < nr; ++other) rank += (__shfl(pd, other, kWave) < pd) ? 1 : 0;
        if (lane < nr) { pd_indices[rank] = pd; gt_indices[rank] = lane; }
    }
}

}  // namespace vsrd
