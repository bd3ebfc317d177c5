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

// Pointers that reach a kernel through a table in memory are "generic" to the compiler and would
// be accessed with flat_load/flat_store (which tick BOTH vmcnt and lgkmcnt and so serialise against
// the LDS pipeline).  Everything here is hipMalloc'ed global memory: say so.
typedef __attribute__((address_space(1))) float gfloat;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;
__device__ __forceinline__ const gfloat* as_global(const float* p) { return (const gfloat*)p; }
__device__ __forceinline__ gfloat* as_global(float* p) { return (gfloat*)p; }

template <bool FAST>
__device__ __forceinline__ void fetch4(const float* __restrict__ base, long ld, int n_rows, int n_cols,
                                       int row, int col, float (&v)[4]) {
    if (FAST) {
        const f32x4 q = *(const gf32x4*)(base + (long)row * ld + col);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
        const gfloat* g = as_global(base);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = (row < n_rows && col + e < n_cols) ? g[(long)row * ld + col + e] : 0.0f;
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
template <bool B_ROWS, int KK0 = 0, int KK1 = BK>
__device__ __forceinline__ void mfma_kstep(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[2][2], int wm, int wn) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const float* a_base = As + (wm * 64 + r) * ROW_LD + h;
    const float* b_base = B_ROWS ? (Bs + (wn * 64 + r) * ROW_LD + h) : (Bs + h * BN + wn * 64 + r);
#pragma unroll
    for (int kk = KK0; kk < KK1; kk += 2) {
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

// k >= k_hi zeroed (low-rank projection: the operands' K extent is the rank r, loads run to the
// next multiple of 32 inside valid memory)
__device__ __forceinline__ void write_rows_khi(float* __restrict__ img, const float (&r)[4][4], int k0, int k_hi) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = img + ((t >> 3) + 32 * j) * ROW_LD + (t & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = (k0 + (t & 7) * 4 + e < k_hi) ? r[j][e] : 0.0f;
    }
}

// ---- the software pipeline shared by every MFMA kernel ------------------------------------
// Two register sets and two LDS buffers.  K-step t (reads LDS buffer t%2):
//     issue the global loads of tile t+2 into register set t%2       (pinned ABOVE the MFMAs)
//     32 MFMAs
//     LDS-write tile t+1's A image from register set (t+1)%2          (loaded during step t-1)
//     16 MFMAs, LDS-write tile t+1's B image, 16 MFMAs, barrier
// so every global load has a full K-step (>= 4096 MFMA cycles) to land before its first use.
//
// Why the pins: left alone, hipcc sinks each global load down to its first use and emits
// `load; s_waitcnt vmcnt(0); ds_write` eight times per K-step -- the whole memory latency,
// serially, with the MFMA pipe idle (measured on 4096^3: 72 TF at one workgroup per CU, 105 at
// two, against 147 for the same loop without staging).  The masked sched_barriers only forbid
// VMEM reads and MFMAs from crossing the first pin and VALU / DS writes from crossing the
// others; LDS reads and SALU still move freely.  RB is a template constant so LDS reads and
// writes are provably disjoint ranges and register sets are statically indexed.
// (Where the LDS writes sit inside the K-step -- between MFMA quarters as here, all before the
//  MFMAs, or all after them -- measured the same within 2 % on 4096^3: 132.4 / 132.4 / 129.7 TF.)
constexpr int SCHED_PIN_VMEM_READ = 0x2 | 0x4 | 0x100;   // VALU, SALU, DS-read may cross
constexpr int SCHED_PIN_STAGING = 0x4 | 0x8 | 0x100;     // SALU, MFMA, DS-read may cross

template <int I>
struct IC {
    static constexpr int value = I;
};

template <bool B_ROWS, int RB, class LoadF, class WriteAF, class WriteBF>
__device__ __forceinline__ void kstep_pipelined(float* smem, f32x16 (&acc)[2][2], int wm, int wn, int t_load,
                                                int t_write, LoadF load, WriteAF write_a, WriteBF write_b) {
    load(t_load, IC<RB>{});
    __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
    const float* Ar = smem + RB * ROW_IMG;
    const float* Br = smem + (2 + RB) * ROW_IMG;
    mfma_kstep<B_ROWS, 0, BK / 2>(Ar, Br, acc, wm, wn);
    __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
    write_a(smem + (1 - RB) * ROW_IMG, t_write, IC<1 - RB>{});
    mfma_kstep<B_ROWS, BK / 2, 3 * BK / 4>(Ar, Br, acc, wm, wn);
    __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
    write_b(smem + (3 - RB) * ROW_IMG, t_write, IC<1 - RB>{});
    mfma_kstep<B_ROWS, 3 * BK / 4, BK>(Ar, Br, acc, wm, wn);
    __syncthreads();
}

// Drives `nk` K-steps.  load(t, IC<S>) fetches the operands of K-step t into the caller's register
// set S; write_a(img, t, IC<S>) / write_b(...) store set S into the given LDS image.  Tile indices
// past the end are clamped to nk-1 (two redundant, in-bounds tile loads per output tile) so the
// loop body is branch-free.
template <bool B_ROWS, class LoadF, class WriteAF, class WriteBF>
__device__ __forceinline__ void mfma_pipeline(int nk, float* smem, f32x16 (&acc)[2][2], LoadF load, WriteAF write_a,
                                              WriteBF write_b) {
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (nk <= 0) return;
    const int last = nk - 1;
    load(0, IC<0>{});
    write_a(a_img(smem, 0), 0, IC<0>{});
    write_b(b_img(smem, 0), 0, IC<0>{});
    load(min(1, last), IC<1>{});
    __syncthreads();
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        kstep_pipelined<B_ROWS, 0>(smem, acc, wm, wn, min(t + 2, last), min(t + 1, last), load, write_a, write_b);
        kstep_pipelined<B_ROWS, 1>(smem, acc, wm, wn, min(t + 3, last), min(t + 2, last), load, write_a, write_b);
    }
    if (t < nk) kstep_pipelined<B_ROWS, 0>(smem, acc, wm, wn, last, last, load, write_a, write_b);
}

// ---- dense tile: acc = (scale*A[m0.., :]) x B ------------------------------------------------
// A: [M x K] row-major (lda).  B_ROWS=false: B is [K x N] row-major (ldb), n contiguous.
// B_ROWS=true: B is given as [N x K] row-major (ldb), i.e. acc = A x B^T.
// (A finer interleave -- 16 slots of 4 MFMAs with one load / LDS-write chunk pinned after each --
//  was measured 5 % SLOWER than the three-pin form on 4096^3, 126.7 vs 132.8 TF: the per-slot pins
//  stop the compiler from batching LDS reads ahead of the MFMAs.)
template <bool FAST_A, bool FAST_B, bool B_ROWS>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                          long ldb, int M, int N, int K, int m0, int n0, float a_scale,
                                          float* smem, f32x16 (&acc)[2][2]) {
    float ra[2][4][4], rb[2][4][4];
    mfma_pipeline<B_ROWS>(
        (K + BK - 1) / BK, smem, acc,
        [&](int t, auto s) {
            constexpr int S = decltype(s)::value;
            stage_rows<FAST_A>(A, lda, M, K, m0, t * BK, ra[S]);
            if (B_ROWS) stage_rows<FAST_B>(B, ldb, N, K, n0, t * BK, rb[S]);
            else stage_kn<FAST_B>(B, ldb, K, N, t * BK, n0, rb[S]);
        },
        [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value], a_scale); },
        [&](float* img, int, auto s) {
            if (B_ROWS) write_rows_noscale(img, rb[decltype(s)::value]);
            else write_kn(img, rb[decltype(s)::value]);
        });
}

}  // namespace nsgp
