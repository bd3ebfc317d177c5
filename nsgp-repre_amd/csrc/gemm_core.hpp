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
//   "row" image  T[8][128][4] (+4 floats of padding per k-quad plane): 128 rows (m or n) x 32 k
//       stored as k-quads, so a staged float4 (4 consecutive k of one row) is ONE ds_write_b128 and
//       a lane's operand for four consecutive MFMAs is ONE ds_read_b128 (row on the lane, lanes
//       0-31 take quad q, lanes 32-63 quad q+1: MFMA j then contracts k = 4(q + l>>5) + j on both
//       operands).  The plane padding spreads the 8 lanes that stage one row's 8 quads over 8 slots;
//   "KN" image   T[32][128]: k rows x 128 n, read with n on the lane (ds_read_b32).
#pragma once
#include <hip/hip_runtime.h>

namespace nsgp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int THREADS = 256;
constexpr int QPLANE = BM * 4 + 4;                   // one k-quad plane of the row image (floats), padded
constexpr int ROW_IMG = (BK / 4) * QPLANE;           // floats (4128)
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

__device__ __forceinline__ float* row_slot(float* img, int j) {
    const int t = threadIdx.x;   // row (t>>3)+32j, k-quad t&7
    return img + (t & 7) * QPLANE + ((t >> 3) + 32 * j) * 4;
}

__device__ __forceinline__ void write_rows(float* __restrict__ img, const float (&r)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(row_slot(img, j)) = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
}

__device__ __forceinline__ void write_kn(float* __restrict__ img, const float (&r)[4][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 q = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
        *reinterpret_cast<float4*>(img + ((t >> 5) + 8 * j) * BN + (t & 31) * 4) = q;
    }
}

// ---- MFMAs of k in [KK0, KK1) (multiples of 8) for this wave's 64x64 sub-tile -------------------
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k'] and B[k'][j = l&31] with k' chosen by l>>5.
// Per pair of k-quads (q, q+1): lane half h reads quad q+h of its A row(s) with one ds_read_b128 and
// MFMA j (j = 0..3) contracts k = 4(q+h) + j.
template <bool B_ROWS, int KK0 = 0, int KK1 = BK>
__device__ __forceinline__ void mfma_kstep(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[2][2], int wm, int wn) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const float* a_base = As + h * QPLANE + (wm * 64 + r) * 4;
    const float* b_rows = Bs + h * QPLANE + (wn * 64 + r) * 4;   // B_ROWS
    const float* b_kn = Bs + (4 * h) * BN + wn * 64 + r;          // KN
#pragma unroll
    for (int q = KK0 / 4; q < KK1 / 4; q += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(a_base + q * QPLANE);
        const float4 a1 = *reinterpret_cast<const float4*>(a_base + q * QPLANE + 32 * 4);
        float b0[4], b1[4];
        if (B_ROWS) {
            const float4 x = *reinterpret_cast<const float4*>(b_rows + q * QPLANE);
            const float4 y = *reinterpret_cast<const float4*>(b_rows + q * QPLANE + 32 * 4);
            b0[0] = x.x; b0[1] = x.y; b0[2] = x.z; b0[3] = x.w;
            b1[0] = y.x; b1[1] = y.y; b1[2] = y.z; b1[3] = y.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b0[j] = b_kn[(4 * q + j) * BN];
                b1[j] = b_kn[(4 * q + j) * BN + 32];
            }
        }
        const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[j], b1[j], acc[1][1], 0, 0, 0);
        }
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
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(row_slot(img, j)) = make_float4(r[j][0] / div[j], r[j][1] / div[j], r[j][2] / div[j], r[j][3] / div[j]);
}

__device__ __forceinline__ void write_rows_klo(float* __restrict__ img, const float (&r)[4][4], int k0, int k_lo) {
    const int k = k0 + (threadIdx.x & 7) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(row_slot(img, j)) = make_float4(k + 0 >= k_lo ? r[j][0] : 0.0f, k + 1 >= k_lo ? r[j][1] : 0.0f,
                                                                   k + 2 >= k_lo ? r[j][2] : 0.0f, k + 3 >= k_lo ? r[j][3] : 0.0f);
}

// k >= k_hi zeroed (low-rank projection: the operands' K extent is the rank r, loads run to the
// next multiple of 32 inside valid memory)
__device__ __forceinline__ void write_rows_khi(float* __restrict__ img, const float (&r)[4][4], int k0, int k_hi) {
    const int k = k0 + (threadIdx.x & 7) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(row_slot(img, j)) = make_float4(k + 0 < k_hi ? r[j][0] : 0.0f, k + 1 < k_hi ? r[j][1] : 0.0f,
                                                                   k + 2 < k_hi ? r[j][2] : 0.0f, k + 3 < k_hi ? r[j][3] : 0.0f);
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

// Per-thread base pointers of the four float4 slots a thread stages per operand (FAST path):
// hoists the 64-bit row*ld products out of the K loop; inside it a load address is base + k offset.
__device__ __forceinline__ void row_bases(const float* base, long ld, int row0, const float* (&ptr)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) ptr[j] = base + (long)(row0 + (t >> 3) + 32 * j) * ld + (t & 7) * 4;
}
__device__ __forceinline__ void kn_bases(const float* base, long ld, int n0, const float* (&ptr)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) ptr[j] = base + (long)((t >> 5) + 8 * j) * ld + n0 + (t & 31) * 4;
}
__device__ __forceinline__ void load4(const float* const (&ptr)[4], long off, float (&r)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 q = *(const gf32x4*)(ptr[j] + off);
        r[j][0] = q[0]; r[j][1] = q[1]; r[j][2] = q[2]; r[j][3] = q[3];
    }
}

// ---- epilogue re-layout: accumulators -> LDS -> one float4 per lane along the row ---------------
// The MFMA C/D layout gives a lane 64 scattered scalars (column on the lane, 16 rows in registers):
// an epilogue that touches 2-3 global arrays per element then issues ~200 dword memory instructions
// per wave.  For short-K tiles that IS the kernel.  After the last barrier of the K loop each wave
// parks its 64x64 block in its own LDS slice (row stride 64 floats: conflict-free dword writes, 16-byte
// aligned rows) and re-reads it as float4 rows: lane l, pass i -> row 4i + (l>>4), columns 4(l&15)..+3,
// i.e. every wave-instruction moves four full 256-byte row segments.
constexpr int EPI_LD = 64;
static_assert(4 * 64 * EPI_LD <= SMEM_FLOATS, "epilogue slices must fit the K-loop LDS");

__device__ __forceinline__ void acc_to_lds(float* smem, const f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = smem + wave * 64 * EPI_LD;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(mi * 32 + acc_row(r, lane)) * EPI_LD + ni * 32 + (lane & 31)] = acc[mi][ni][r];
}

// f(row_in_tile, col_in_tile, float4 acc_values) for this wave's 64x64 block, 16 passes
template <class F>
__device__ __forceinline__ void for_each_row4(const float* smem, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* mine = smem + wave * 64 * EPI_LD;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int row = 4 * i + (lane >> 4), col = 4 * (lane & 15);
        const float4 v = *reinterpret_cast<const float4*>(mine + row * EPI_LD + col);
        f(wm * 64 + row, wn * 64 + col, v);
    }
}

// ---- dense tile: acc = A[m0.., :] x B (callers apply any scalar factor in their epilogue) ------------------------------------------------
// A: [M x K] row-major (lda).  B_ROWS=false: B is [K x N] row-major (ldb), n contiguous.
// B_ROWS=true: B is given as [N x K] row-major (ldb), i.e. acc = A x B^T.
// (A finer interleave -- 16 slots of 4 MFMAs with one load / LDS-write chunk pinned after each --
//  was measured 5 % SLOWER than the three-pin form on 4096^3, 126.7 vs 132.8 TF: the per-slot pins
//  stop the compiler from batching LDS reads ahead of the MFMAs.)
template <bool FAST_A, bool FAST_B, bool B_ROWS>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                          long ldb, int M, int N, int K, int m0, int n0,
                                          float* smem, f32x16 (&acc)[2][2]) {
    float ra[2][4][4], rb[2][4][4];
    const float *pa[4], *pb[4];
    if (FAST_A) row_bases(A, lda, m0, pa);
    if (FAST_B) {
        if (B_ROWS) row_bases(B, ldb, n0, pb);
        else kn_bases(B, ldb, n0, pb);
    }
    const long bstep = B_ROWS ? (long)BK : (long)BK * ldb;
    mfma_pipeline<B_ROWS>(
        (K + BK - 1) / BK, smem, acc,
        [&](int t, auto s) {
            constexpr int S = decltype(s)::value;
            if (FAST_A) load4(pa, (long)t * BK, ra[S]);
            else stage_rows<false>(A, lda, M, K, m0, t * BK, ra[S]);
            if (FAST_B) load4(pb, t * bstep, rb[S]);
            else if (B_ROWS) stage_rows<false>(B, ldb, N, K, n0, t * BK, rb[S]);
            else stage_kn<false>(B, ldb, K, N, t * BK, n0, rb[S]);
        },
        [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value]); },
        [&](float* img, int, auto s) {
            if (B_ROWS) write_rows(img, rb[decltype(s)::value]);
            else write_kn(img, rb[decltype(s)::value]);
        });
}

}  // namespace nsgp
