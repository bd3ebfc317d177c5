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
// A class with few RoIs has few tiles (N=300: 6) and a long contraction (D=12544: 392 K-steps) -- six
// workgroups on a 256-CU chip.  Below 256 tiles the flattened (tile, K-step) space is therefore cut
// stream-K style into equal ranges over up to 512 resident workgroups (as in covariance.hip): segments
// write fp32 partial tiles, a second launch sums each tile's segments in workgroup order, thresholds
// and emits the same mask words.
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

// FAST operand pointers with the row index clamped to N-1: tiles that hang over the last row re-read it
// (in-bounds, unguarded float4 loads); whatever those rows produce is dropped by the `row < N` tests downstream.
__device__ __forceinline__ void clamped_row_bases(const float* base, long ld, int row0, int n_rows, const float* (&ptr)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) ptr[j] = base + (long)min(row0 + (t >> 3) + 32 * j, n_rows - 1) * ld + (t & 7) * 4;
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
    const float *pa[4], *pb[4];
    if (FAST) {
        clamped_row_bases(F, D, m0, N, pa);
        clamped_row_bases(F, D, n0, N, pb);
    }
    mfma_pipeline<true>(
        (D + BK - 1) / BK, smem, acc,
        [&](int k, auto s) {
            if (FAST) {
                load4(pa, (long)k * BK, ra[decltype(s)::value]);
                load4(pb, (long)k * BK, rb[decltype(s)::value]);
            } else {
                stage_rows<false>(F, D, N, D, m0, k * BK, ra[decltype(s)::value]);
                stage_rows<false>(F, D, N, D, n0, k * BK, rb[decltype(s)::value]);
            }
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

__host__ __device__ __forceinline__ long sim_sk_begin(long w, long G, long P) { return w * G / P; }
__host__ __device__ __forceinline__ long sim_sk_owner(long gidx, long G, long P) { return ((gidx + 1) * P - 1) / G; }
__device__ __forceinline__ void sim_tile_of(int t, int nb, int& ti, int& tj) {
    ti = 0;
    int rem = t;
    while (rem >= nb - ti) { rem -= nb - ti; ++ti; }
    tj = ti + rem;
}

// stream-K phase 1: workgroup w contracts its contiguous range of (tile, K-step) units; segment -> slab (tile + w)
template <bool FAST>
__global__ __launch_bounds__(256, 2) void repre_sim_partial_kernel(const float* __restrict__ F, int N, int D,
                                                                   const float* __restrict__ nrm, int nk, long G,
                                                                   float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nb = (N + BM - 1) / BM;
    const long P = gridDim.x, w = blockIdx.x;
    const int t = threadIdx.x;
    long gi = sim_sk_begin(w, G, P);
    const long g_end = sim_sk_begin(w + 1, G, P);
    float ra[2][4][4], rb[2][4][4];
    while (gi < g_end) {
        const int tile = (int)(gi / nk);
        const int ja = (int)(gi - (long)tile * nk);
        const int jb = (int)min((long)nk, ja + (g_end - gi));
        int ti, tj;
        sim_tile_of(tile, nb, ti, tj);
        const int m0 = ti * BM, n0 = tj * BN;
        float da[4], db[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ra_ = m0 + (t >> 3) + 32 * j, rb_ = n0 + (t >> 3) + 32 * j;
            da[j] = ra_ < N ? nrm[ra_] : 1.0f;
            db[j] = rb_ < N ? nrm[rb_] : 1.0f;
        }
        f32x16 acc[2][2];
        zero_acc(acc);
        const float *pa[4], *pb[4];
        if (FAST) {
            clamped_row_bases(F, D, m0, N, pa);
            clamped_row_bases(F, D, n0, N, pb);
        }
        mfma_pipeline<true>(
            jb - ja, smem, acc,
            [&](int k, auto s) {
                if (FAST) {
                    load4(pa, (long)(ja + k) * BK, ra[decltype(s)::value]);
                    load4(pb, (long)(ja + k) * BK, rb[decltype(s)::value]);
                } else {
                    stage_rows<false>(F, D, N, D, m0, (ja + k) * BK, ra[decltype(s)::value]);
                    stage_rows<false>(F, D, N, D, n0, (ja + k) * BK, rb[decltype(s)::value]);
                }
            },
            [&](float* img, int, auto s) { write_rows_div(img, ra[decltype(s)::value], da); },
            [&](float* img, int, auto s) { write_rows_div(img, rb[decltype(s)::value], db); });
        float* out = slabs + ((long)tile + w) * (BM * BN);
        acc_to_lds(smem, acc);
        for_each_row4(smem, [&](int r, int col, float4 v) {
            f32x4 q;
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            *(gf32x4*)(out + r * BN + col) = q;
        });
        __syncthreads();
        gi += jb - ja;
    }
}

constexpr int SIM_LD = BN + 1;
constexpr int SIM_BAND = 32;     // rows per finishing workgroup = one 32-bit word of the mirrored mask

// stream-K phase 2, grid (tiles, 4): workgroup (t, b) owns the 32-row band b of tile t: similarity = sum of the
// tile's segments (workgroup order) for those rows, then the mask words: (row, 32-column word) x 4 for each of
// its rows and, off the diagonal, the one (column, 32-row word) its band contributes to each mirrored row.
__global__ __launch_bounds__(256) void repre_sim_finish_kernel(const float* __restrict__ slabs, int N, int nk, long G, long P,
                                                               float thr, uint32_t* __restrict__ mask32, int words32) {
    __shared__ float tile[SIM_BAND * SIM_LD];
    const int nb = (N + BM - 1) / BM;
    const int t = blockIdx.x, band = blockIdx.y, r0 = band * SIM_BAND;
    int ti, tj;
    sim_tile_of(t, nb, ti, tj);
    const int m0 = ti * BM, n0 = tj * BN;
    if (m0 + r0 >= N) return;
    const long w_first = sim_sk_owner((long)t * nk, G, P), w_last = sim_sk_owner((long)(t + 1) * nk - 1, G, P);
    for (int idx = threadIdx.x; idx < SIM_BAND * BN / 4; idx += 256) {
        const int r = idx / (BN / 4), c4 = (idx - r * (BN / 4)) * 4;
        const long off = (long)(r0 + r) * BN + c4;
        f32x4 sum = *(const gf32x4*)(slabs + ((long)t + w_first) * (BM * BN) + off);
        for (long w = w_first + 1; w <= w_last; ++w) {
            const f32x4 v = *(const gf32x4*)(slabs + ((long)t + w) * (BM * BN) + off);
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r * SIM_LD + c4 + e] = sum[e];
    }
    __syncthreads();
    if (threadIdx.x < SIM_BAND * 4) {
        const int r = threadIdx.x >> 2, wq = threadIdx.x & 3;
        const int row = m0 + r0 + r, word = n0 / 32 + wq;
        if (row < N && word < words32) {
            uint32_t bits = 0;
            for (int c = 0; c < 32; ++c)
                if (n0 + wq * 32 + c < N && tile[r * SIM_LD + wq * 32 + c] >= thr) bits |= 1u << c;
            mask32[(long)row * words32 + word] = bits;
        }
    } else if (ti != tj) {                                      // threads 128..255: one mirrored word per tile column
        const int c = threadIdx.x - SIM_BAND * 4;
        const int row = n0 + c, word = (m0 + r0) / 32;
        if (row < N && word < words32) {
            uint32_t bits = 0;
            for (int r = 0; r < SIM_BAND; ++r)
                if (m0 + r0 + r < N && tile[r * SIM_LD + c] >= thr) bits |= 1u << r;
            mask32[(long)row * words32 + word] = bits;
        }
    }
}

struct SimPlan {
    int nk;
    long tiles, G, P;
    bool stream_k;
};

static SimPlan sim_plan(int n, int d) {
    const int nb = (n + BM - 1) / BM;
    SimPlan p;
    p.nk = (d + BK - 1) / BK;
    p.tiles = (long)nb * (nb + 1) / 2;
    p.G = p.tiles * p.nk;
    p.P = std::max<long>(1, std::min<long>(512, p.G / 16));
    p.stream_k = p.tiles < 256 && p.P > p.tiles;     // enough tiles fill the chip by themselves
    return p;
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

extern "C" size_t repre_sim_workspace_bytes(int n, int d) {
    if (n <= 0 || d <= 0) return 0;
    const SimPlan p = sim_plan(n, d);
    return p.stream_k ? (size_t)(p.tiles + p.P) * BM * BN * 4 : 0;
}

extern "C" int repre_sim_counts(const float* feats, int n, int d, float thr, float* norm_scratch, int64_t* counts,
                                uint64_t* bitmask, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!feats || !norm_scratch || !counts || !bitmask || n <= 0 || d <= 0) return fail(NSGP_ERR_INVALID, "repre_sim_counts: bad argument");
    const SimPlan p = sim_plan(n, d);
    if (p.stream_k && (!workspace || workspace_bytes < repre_sim_workspace_bytes(n, d)))
        return fail(NSGP_ERR_WORKSPACE, "repre_sim_counts: workspace %zu < %zu", workspace_bytes, repre_sim_workspace_bytes(n, d));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int words32 = 2 * ((n + 63) / 64);
    hipLaunchKernelGGL(repre_row_norm_kernel, dim3(n), dim3(256), 0, stream, feats, n, d, norm_scratch);
    NSGP_LAUNCH_CHECK();
    // words past the last 128-column tile boundary are never touched by a tile: clear the matrix first
    NSGP_HIP(hipMemsetAsync(bitmask, 0, (size_t)n * words32 * 4, stream));
    const bool fast = (d % BK == 0) && aligned16(feats);   // any N: rows past the end are clamped (clamped_row_bases)
    uint32_t* m32 = reinterpret_cast<uint32_t*>(bitmask);
    if (p.stream_k) {
        float* slabs = static_cast<float*>(workspace);
        if (fast) {
            NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_partial_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
            hipLaunchKernelGGL(repre_sim_partial_kernel<true>, dim3((unsigned)p.P), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, p.nk, p.G, slabs);
        } else {
            NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_partial_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
            hipLaunchKernelGGL(repre_sim_partial_kernel<false>, dim3((unsigned)p.P), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, p.nk, p.G, slabs);
        }
        NSGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(repre_sim_finish_kernel, dim3((unsigned)p.tiles, BM / SIM_BAND), dim3(256), 0, stream, slabs, n, p.nk, p.G, p.P, thr, m32, words32);
    } else if (fast) {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_mask_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(repre_sim_mask_kernel<true>, dim3((unsigned)p.tiles), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, thr, m32, words32);
    } else {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(repre_sim_mask_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(repre_sim_mask_kernel<false>, dim3((unsigned)p.tiles), dim3(THREADS), SMEM_BYTES, stream, feats, n, d, norm_scratch, thr, m32, words32);
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
