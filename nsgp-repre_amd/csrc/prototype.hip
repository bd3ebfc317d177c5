// K6 / K7: Regional-Prototype-Replay bank construction kernels.  Replace, per old class,
// mmdet/models/roi_heads/standard_roi_replay_head.py:417-423
//     Fn  = F / F.norm(dim=-1, keepdim=True)
//     sim = Fn @ Fn.t();  sim_mask = sim >= 0.6;  counts = sim_mask.long().sum(-1)
// and the (masked) row means of :413 / :443.
//
// The reference materialises the N x N fp32 similarity AND an N x N int64 copy of the mask.
// Here the similarity tile lives only in MFMA accumulators: rows are normalised while they are
// staged into LDS (true division, like the reference), only tiles on or above the diagonal are
// computed, and the epilogue turns each 32x32 accumulator block into 32-bit mask words with
// wave ballots -- stored once for (row, col-segment) and once, transposed, for (col, row-segment).
// Every 32-bit word of the bit matrix is written by exactly one wave: no atomics, deterministic.
#include <algorithm>

#include "common.hpp"
#include "gemm_core.hpp"

namespace nsgp {

// ||row||_2, one workgroup per row (torch `norm(dim=-1)`).
__global__ __launch_bounds__(256) void repre_row_norm_kernel(const float* __restrict__ F, int N, int D, float* __restrict__ nrm) {
    __shared__ float red[4];
    const float* row = F + (long)blockIdx.x * D;
    float s = 0.0f;
    for (int i = threadIdx.x; i < D; i += 256) s = fmaf(row[i], row[i], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) nrm[blockIdx.x] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

template <bool FAST>
__global__ __launch_bounds__(256, 2) void repre_sim_mask_kernel(const float* __restrict__ F, int N, int D,
                                                                const float* __restrict__ nrm, float thr,
                                                                uint32_t* __restrict__ mask32, int words32) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nb = (N + BM - 1) / BM;
    int ti = 0, rem = blockIdx.x;
    while (rem >= nb - ti) { rem -= nb - ti; ++ti; }
    const int m0 = ti * BM, n0 = (ti + rem) * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    zero_acc(acc);
    float da[4], db[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ra_ = m0 + (t >> 3) + 32 * j, rb_ = n0 + (t >> 3) + 32 * j;
        da[j] = ra_ < N ? nrm[ra_] : 1.0f;
        db[j] = rb_ < N ? nrm[rb_] : 1.0f;
    }
    float ra[2][4][4], rb[2][4][4];
    mfma_pipeline<true>(
        (D + BK - 1) / BK, smem, acc,
        [&](int k, auto s) {
            stage_rows<FAST>(F, D, N, D, m0, k * BK, ra[decltype(s)::value]);
            stage_rows<FAST>(F, D, N, D, n0, k * BK, rb[decltype(s)::value]);
        },
        [&](float* img, int, auto s) { write_rows_div(img, ra[decltype(s)::value], da); },
        [&](float* img, int, auto s) { write_rows_div(img, rb[decltype(s)::value], db); });
    // ---- epilogue: accumulators -> bit matrix ---------------------------------------------
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int row_blk = m0 + wm * 64 + mi * 32;  // 32-row block
            const int col_blk = n0 + wn * 64 + ni * 32;  // 32-col block
            const int col = col_blk + c;
            uint32_t colbits = 0;  // this lane's 16 rows of column `col`, as bits of the 32-row block
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = acc_row(r, lane);  // row inside the block (depends on h)
                const bool p = (row_blk + rr < N) && (col < N) && (acc[mi][ni][r] >= thr);
                const unsigned long long b = __ballot(p);  // low half: row (r,h=0), high half: row (r,h=1)
                if (p) colbits |= (1u << rr);
                // lane 0 stores the h=0 row's word, lane 32 the h=1 row's word
                if (c == 0) {
                    const int row = row_blk + rr;
                    if (row < N && col_blk / 32 < words32)
                        mask32[(long)row * words32 + col_blk / 32] = (uint32_t)(h ? (b >> 32) : (b & 0xffffffffull));
                }
            }
            if (m0 != n0) {  // mirror: (col, row-block) word = this lane's 16 bits | partner half's 16 bits
                const uint32_t other = __shfl_xor(colbits, 32, 64);
                if (h == 0 && col < N && row_blk / 32 < words32)
                    mask32[(long)col * words32 + row_blk / 32] = colbits | other;
            }
        }
}

// counts[i] = popcount of row i (the reference's `.long().sum(-1)`, int64)
__global__ __launch_bounds__(256) void repre_count_kernel(const uint32_t* __restrict__ mask32, int N, int words32,
                                                          long long* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    long long s = 0;
    for (int w = 0; w < words32; ++w) s += __popc(mask32[(long)i * words32 + w]);
    counts[i] = s;
}

// partial[seg][d] = sum of the selected rows of row segment `seg` (rows in ascending order)
__global__ __launch_bounds__(256) void repre_masked_sum_kernel(const float* __restrict__ F, int N, int D,
                                                               const unsigned long long* __restrict__ rowmask,
                                                               int rows_per_seg, float* __restrict__ partial) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_seg, r1 = min(r0 + rows_per_seg, N);
    if (d >= D) return;
    float s = 0.0f;
    for (int r = r0; r < r1; ++r) {
        const bool sel = rowmask ? ((rowmask[r >> 6] >> (r & 63)) & 1ull) : true;
        if (sel) s += F[(long)r * D + d];
    }
    partial[(long)blockIdx.y * D + d] = s;
}

__global__ __launch_bounds__(256) void repre_mean_final_kernel(const float* __restrict__ partial, int n_seg, int D,
                                                               int n_selected, float* __restrict__ out) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float s = partial[d];
    for (int k = 1; k < n_seg; ++k) s += partial[(long)k * D + d];
    out[d] = s / (float)n_selected;
}

static int mean_segments(int n) { return std::max(1, std::min(64, (n + 127) / 128)); }

}  // namespace nsgp

using namespace nsgp;

extern "C" int repre_sim_counts(const float* feats, int n, int d, float thr, float* norm_scratch, int64_t* counts,
                                uint64_t* bitmask, void* stream_) {
    if (!feats || !norm_scratch || !counts || !bitmask || n <= 0 || d <= 0) return fail(NSGP_ERR_INVALID, "repre_sim_counts: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int words32 = 2 * ((n + 63) / 64);
    hipLaunchKernelGGL(repre_row_norm_kernel, dim3(n), dim3(256), 0, stream, feats, n, d, norm_scratch);
    NSGP_LAUNCH_CHECK();
    // words past the last 128-column tile boundary are never touched by a tile: clear the matrix first
    NSGP_HIP(hipMemsetAsync(bitmask, 0, (size_t)n * words32 * 4, stream));
    const int nb = (n + BM - 1) / BM;
    const int tiles = nb * (nb + 1) / 2;
    const bool fast = (d % BK == 0) && (d % 4 == 0) && aligned16(feats) && (n % BM == 0);
    uint32_t* m32 = reinterpret_cast<uint32_t*>(bitmask);
    if (fast) {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_mask_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(repre_sim_mask_kernel<true>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, thr, m32, words32);
    } else {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_mask_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(repre_sim_mask_kernel<false>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, thr, m32, words32);
    }
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(repre_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, m32, n, words32, reinterpret_cast<long long*>(counts));
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" size_t repre_masked_mean_workspace_bytes(int n, int d) {
    if (n <= 0 || d <= 0) return 0;
    return (size_t)mean_segments(n) * d * 4;
}

extern "C" int repre_masked_mean(const float* feats, int n, int d, const uint64_t* rowmask, int n_selected, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    if (!feats || !out || n <= 0 || d <= 0 || n_selected <= 0) return fail(NSGP_ERR_INVALID, "repre_masked_mean: bad argument");
    if (!workspace || workspace_bytes < repre_masked_mean_workspace_bytes(n, d)) return fail(NSGP_ERR_WORKSPACE, "repre_masked_mean: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int segs = mean_segments(n);
    const int rows_per_seg = (n + segs - 1) / segs;
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(repre_masked_sum_kernel, dim3((d + 255) / 256, segs), dim3(256), 0, stream, feats, n, d,
                       reinterpret_cast<const unsigned long long*>(rowmask), rows_per_seg, partial);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(repre_mean_final_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, partial, segs, d, n_selected, out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
