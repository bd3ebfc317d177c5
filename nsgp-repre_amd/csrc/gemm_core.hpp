// fp32 MFMA tile core shared by the NSGP-RePRE kernels (gfx950 / CDNA4 only).
//
// One workgroup = 256 threads = 4 wave64 arranged 2x2; workgroup tile 128x128,
// K-step 32; each wave owns a 64x64 sub-tile as 2x2 blocks of
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, 64 FLOP/clk/SIMD,
// the only fp32-in matrix path on gfx950 -- there is no xf32).
//
// Data path per K-step: global -> registers (issued before the MFMA block of the
// previous K-step, so HBM/L2 latency hides under 64 MFMAs x 64 cycles per wave)
// -> LDS (double buffered, one barrier per K-step) -> one f32 VGPR per MFMA
// operand by ds_read_b32.  fp32 MFMA is so slow relative to LDS (4 reads feed
// 256 MFMA cycles) that LDS bandwidth is irrelevant; the layouts below are
// chosen only to be bank-conflict free:
//   "row" image  T[128][33]: 128 rows (m or n) x 32 k, padded to 33 so that
//       lane (l&31) -> row, (l>>5) -> k reads 32 distinct banks;
//   "KN" image   T[32][128]: k rows x 128 n, read with n on the lane.
#pragma once
#include <hip/hip_runtime.h>

namespace nsgp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int THREADS = 256;
constexpr int ROW_LD = BK + 1;                       // padded row image
constexpr int ROW_IMG = BM * ROW_LD;                 // floats
constexpr int KN_IMG = BK * BN;                      // floats
// LDS carve (floats): [A0][A1][B0][B1]; B images sized for the larger (row) form.
constexpr int SMEM_FLOATS = 2 * ROW_IMG + 2 * ROW_IMG;
constexpr int SMEM_BYTES = SMEM_FLOATS * 4;          // 67,584 B -> 2 workgroups / CU

__device__ __forceinline__ float* a_img(float* smem, int i) { return smem + i * ROW_IMG; }
__device__ __forceinline__ float* b_img(float* smem, int i) { return smem + (2 + i) * ROW_IMG; }

// ---- register staging ------------------------------------------------------
// Row-image operand: thread t stages rows (t>>3)+32*j, j=0..3, k = (t&7)*4 .. +3.
// KN-image operand:  thread t stages k rows (t>>5)+8*j, n = (t&31)*4 .. +3.

template <bool FAST>
__device__ __forceinline__ void fetch4(const float* __restrict__ base, long ld, int n_rows, int n_cols,
                                       int row, int col, float (&v)[4]) {
    if (FAST) {
        const float4 q = *reinterpret_cast<const float4*>(base + (long)row * ld + col);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = (row < n_rows && col + e < n_cols) ? base[(long)row * ld + col + e] : 0.0f;
    }
}

template <bool FAST>
__device__ __forceinline__ void stage_rows(const float* __restrict__ base, long ld, int n_rows, int n_k,
                                           int row0, int k0, float (&r)[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch4<FAST>(base, ld, n_rows, n_k, row0 + (t >> 3) + 32 * j, k0 + (t & 7) * 4, r[j]);
}

template <bool FAST>
__device__ __forceinline__ void stage_kn(const float* __restrict__ base, long ld, int n_k, int n_n,
                                         int k0, int n0, float (&r)[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch4<FAST>(base, ld, n_k, n_n, k0 + (t >> 5) + 8 * j, n0 + (t & 31) * 4, r[j]);
}

__device__ __forceinline__ void write_rows(float* __restrict__ img, const float (&r)[4][4], float scale) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = img + ((t >> 3) + 32 * j) * ROW_LD + (t & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = scale * r[j][e];
    }
}

__device__ __forceinline__ void write_rows_noscale(float* __restrict__ img, const float (&r)[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = img + ((t >> 3) + 32 * j) * ROW_LD + (t & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = r[j][e];
    }
}

__device__ __forceinline__ void write_kn(float* __restrict__ img, const float (&r)[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 q = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
        *reinterpret_cast<float4*>(img + ((t >> 5) + 8 * j) * BN + (t & 31) * 4) = q;
    }
}

// ---- one K-step of MFMAs for this wave's 64x64 sub-tile --------------------
// A operand of v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5];
// B operand: lane l holds B[k = l>>5][j = l&31].
template <bool B_ROWS>
__device__ __forceinline__ void mfma_kstep(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[2][2], int wm, int wn) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const float* a_base = As + (wm * 64 + r) * ROW_LD + h;
    const float* b_base = B_ROWS ? (Bs + (wn * 64 + r) * ROW_LD + h) : (Bs + h * BN + wn * 64 + r);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const float a0 = a_base[kk];
        const float a1 = a_base[32 * ROW_LD + kk];
        float b0, b1;
        if (B_ROWS) {
            b0 = b_base[kk];
            b1 = b_base[32 * ROW_LD + kk];
        } else {
            b0 = b_base[kk * BN];
            b1 = b_base[kk * BN + 32];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Row-image write with a per-row divisor (prototype similarity: rows are L2-normalised on
// the fly, `F / F.norm(dim=-1, keepdim=True)`) or with k < k_lo zeroed (projector build: the
// basis is the column range [k_lo, D) of V but loads start at an aligned k).
__device__ __forceinline__ void write_rows_div(float* __restrict__ img, const float (&r)[4][4], const float (&div)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = img + ((t >> 3) + 32 * j) * ROW_LD + (t & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = r[j][e] / div[j];
    }
}

__device__ __forceinline__ void write_rows_klo(float* __restrict__ img, const float (&r)[4][4], int k0, int k_lo) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = img + ((t >> 3) + 32 * j) * ROW_LD + (t & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = (k0 + (t & 7) * 4 + e >= k_lo) ? r[j][e] : 0.0f;
    }
}

// ---- generic dense tile: acc = (scale*A[m0.., :]) x B ------------------------
// A: [M x K] row-major (lda).  B_ROWS=false: B is [K x N] row-major (ldb), n contiguous.
// B_ROWS=true: B is given as [N x K] row-major (ldb), i.e. acc = A x B^T.
template <bool FAST_A, bool FAST_B, bool B_ROWS>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                          long ldb, int M, int N, int K, int m0, int n0, float a_scale,
                                          float* smem, f32x16 (&acc)[2][2]) {
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float ra[4][4], rb[4][4];
    const int nk = (K + BK - 1) / BK;

    stage_rows<FAST_A>(A, lda, M, K, m0, 0, ra);
    if (B_ROWS) stage_rows<FAST_B>(B, ldb, N, K, n0, 0, rb);
    else stage_kn<FAST_B>(B, ldb, K, N, 0, n0, rb);
    write_rows(a_img(smem, 0), ra, a_scale);
    if (B_ROWS) write_rows_noscale(b_img(smem, 0), rb);
    else write_kn(b_img(smem, 0), rb);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            stage_rows<FAST_A>(A, lda, M, K, m0, (t + 1) * BK, ra);
            if (B_ROWS) stage_rows<FAST_B>(B, ldb, N, K, n0, (t + 1) * BK, rb);
            else stage_kn<FAST_B>(B, ldb, K, N, (t + 1) * BK, n0, rb);
        }
        mfma_kstep<B_ROWS>(a_img(smem, cur), b_img(smem, cur), acc, wm, wn);
        if (t + 1 < nk) {
            write_rows(a_img(smem, cur ^ 1), ra, a_scale);
            if (B_ROWS) write_rows_noscale(b_img(smem, cur ^ 1), rb);
            else write_kn(b_img(smem, cur ^ 1), rb);
        }
        __syncthreads();
    }
}

}  // namespace nsgp
