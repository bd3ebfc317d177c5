// The RePRE replay classifier loss of StandardMultiPrototypeReplayHead.replay_loss
// (mmdet/models/roi_heads/standard_roi_replay_head.py:499):
//     loss = F.cross_entropy(cls_score.softmax(dim=-1), labels)          (mean over the K prototypes)
// i.e. a softmax FOLLOWED by the log-softmax inside cross_entropy -- the reference's double softmax,
// reproduced.  PyTorch runs softmax, log_softmax, nll_loss (+ their three backward kernels) on a
// [K x C] matrix with K <= 400 rows and C <= 81 columns: pure launch overhead.  Here one wave64 owns
// a row: two wave-wide max/sum reductions give q = softmax(s) and lse = logsumexp(q); the forward is a
// single 16-wave workgroup (ordered, deterministic mean), the backward one wave per row:
//     dL/dq_j = (softmax(q)_j - [j == y]) / K,      ds_i = q_i * (dq_i - sum_j dq_j q_j).
#include "common.hpp"
#include "replay_ce.hpp"

namespace nsgp {

constexpr int CE_FWD_WAVES = 16;   // one workgroup of 16 waves: K = 150 rows -> <= 10 dependent row passes per wave
constexpr int CE_FWD_ROWS = 32;    // rows per wave the narrow form keeps in registers: K <= 512

// One row already in registers (lane = column, C <= 64): -log_softmax(softmax(row))[y]
__device__ __forceinline__ float row_double_softmax_narrow(float v, int C, int lane, int y) {
    const bool in = lane < C;
    const float m = wave_max(in ? v : -INFINITY);
    float q = in ? expf(v - m) : 0.0f;
    const float s = wave_sum(q);
    q = q / s;
    const float m2 = wave_max(in ? q : -INFINITY);
    const float s2 = wave_sum(in ? expf(q - m2) : 0.0f);
    const float qy = wave_sum(lane == y ? q : 0.0f);
    return m2 + logf(s2) - qy;
}

// NARROW (C <= 64 and K <= 16 * 32): every wave first issues the loads of ALL its rows (one dword per lane and row), then works
// through them from registers -- the general form below pays one global-memory round trip per row on its dependent chain
// (10 of them for K = 150: 20 us for a [150 x 21] matrix).  Same arithmetic, same summation order.
template <bool NARROW>
__global__ __launch_bounds__(64 * CE_FWD_WAVES) void repre_replay_ce_fwd_kernel(const float* __restrict__ scores,
                                                                               const long long* __restrict__ labels, int K, int C,
                                                                               float* __restrict__ loss_out) {
    __shared__ float part[CE_FWD_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    if (NARROW) {
        float v[CE_FWD_ROWS];
        int y[CE_FWD_ROWS];
#pragma unroll
        for (int i = 0; i < CE_FWD_ROWS; ++i) {
            const int r = wave + CE_FWD_WAVES * i;
            v[i] = (r < K && lane < C) ? scores[(long)r * C + lane] : 0.0f;
            y[i] = r < K ? (int)labels[r] : 0;
        }
#pragma unroll
        for (int i = 0; i < CE_FWD_ROWS; ++i)
            if (wave + CE_FWD_WAVES * i < K) acc += row_double_softmax_narrow(v[i], C, lane, y[i]);
    } else {
        for (int r = wave; r < K; r += CE_FWD_WAVES) {
            float q[4];
            const float lse = row_double_softmax(scores + (long)r * C, C, lane, q);
            const int y = (int)labels[r];
            float qy = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (lane + 64 * e == y) qy = q[e];
            qy = wave_sum(qy);
            acc += lse - qy;                 // -log_softmax(q)[y]
        }
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {              // fixed order: deterministic
        float t = 0.0f;
        for (int w = 0; w < CE_FWD_WAVES; ++w) t += part[w];
        *loss_out = t / (float)K;
    }
}

__global__ __launch_bounds__(256) void repre_replay_ce_bwd_kernel(const float* __restrict__ scores, const long long* __restrict__ labels,
                                                                  int K, int C, const float* __restrict__ grad_out,
                                                                  float* __restrict__ grad_scores) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= K) return;
    float q[4];
    const float lse = row_double_softmax(scores + (long)r * C, C, lane, q);
    const int y = (int)labels[r];
    const float go = *grad_out / (float)K;
    float dq[4], dot = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = lane + 64 * e;
        dq[e] = (c < C) ? go * (expf(q[e] - lse) - (c == y ? 1.0f : 0.0f)) : 0.0f;
        dot += dq[e] * q[e];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = lane + 64 * e;
        if (c < C) grad_scores[(long)r * C + c] = q[e] * (dq[e] - dot);
    }
}

}  // namespace nsgp

using namespace nsgp;

extern "C" int repre_replay_ce_forward(const float* scores, const int64_t* labels, int n_rows, int n_cols, float* loss_out, void* stream_) {
    if (!scores || !labels || !loss_out || n_rows <= 0 || n_cols <= 0) return fail(NSGP_ERR_INVALID, "repre_replay_ce_forward: bad argument");
    if (n_cols > CE_MAX_COLS) return fail(NSGP_ERR_LIMIT, "repre_replay_ce_forward: %d columns > %d", n_cols, CE_MAX_COLS);
    if (n_cols <= 64 && n_rows <= CE_FWD_WAVES * CE_FWD_ROWS)
        hipLaunchKernelGGL(repre_replay_ce_fwd_kernel<true>, dim3(1), dim3(64 * CE_FWD_WAVES), 0, static_cast<hipStream_t>(stream_), scores,
                           reinterpret_cast<const long long*>(labels), n_rows, n_cols, loss_out);
    else
        hipLaunchKernelGGL(repre_replay_ce_fwd_kernel<false>, dim3(1), dim3(64 * CE_FWD_WAVES), 0, static_cast<hipStream_t>(stream_), scores,
                           reinterpret_cast<const long long*>(labels), n_rows, n_cols, loss_out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int repre_replay_ce_backward(const float* scores, const int64_t* labels, int n_rows, int n_cols, const float* grad_out,
                                        float* grad_scores, void* stream_) {
    if (!scores || !labels || !grad_out || !grad_scores || n_rows <= 0 || n_cols <= 0) return fail(NSGP_ERR_INVALID, "repre_replay_ce_backward: bad argument");
    if (n_cols > CE_MAX_COLS) return fail(NSGP_ERR_LIMIT, "repre_replay_ce_backward: %d columns > %d", n_cols, CE_MAX_COLS);
    hipLaunchKernelGGL(repre_replay_ce_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), scores,
                       reinterpret_cast<const long long*>(labels), n_rows, n_cols, grad_out, grad_scores);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
