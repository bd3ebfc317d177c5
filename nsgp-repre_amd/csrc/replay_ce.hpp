// Device helpers of the double-softmax CE (standard_roi_replay_head.py:499), shared by replay_ce.hip and replay_head.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nsgp {

constexpr int CE_MAX_COLS = 256;   // 4 columns per lane

__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// q (4 per lane) = softmax(row); returns logsumexp(q)
__device__ __forceinline__ float row_double_softmax(const float* __restrict__ row, int C, int lane, float (&q)[4]) {
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = lane + 64 * e; q[e] = c < C ? row[c] : -INFINITY; m = fmaxf(m, q[e]); }
    m = wave_max(m);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { q[e] = (lane + 64 * e < C) ? expf(q[e] - m) : 0.0f; s += q[e]; }
    s = wave_sum(s);
    float m2 = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) { q[e] = q[e] / s; if (lane + 64 * e < C) m2 = fmaxf(m2, q[e]); }
    m2 = wave_max(m2);
    float s2 = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (lane + 64 * e < C) s2 += expf(q[e] - m2);
    s2 = wave_sum(s2);
    return m2 + logf(s2);
}

}  // namespace nsgp
