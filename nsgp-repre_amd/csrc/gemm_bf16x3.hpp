// fp32-accurate projection GEMM on the bf16 matrix cores: every fp32 operand is split into three bf16 terms
// (a = a0 + a1 + a2, each the round-to-nearest bf16 of the remaining residual: 3 x 8 = 24 mantissa bits) and
//     a*b  ~=  a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0          (the dropped terms are <= 2^-24 |a b|)
// is accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 per 32x32x16 block.  gfx950 runs bf16 MFMAs at 16x the
// rate of v_mfma_f32_32x32x2_f32, so the six-term product has 2.7x the fp32 matrix roof.
//
// Operands of the tile function:
//   A  [M x K] fp32 row-major (the update, read as it is and split while it is written to LDS);
//   Bt the three bf16 terms of the TRANSPOSED right operand (the projector, split once per task by
//      nsgp_split_transpose_bf16x3_kernel), so both operands are "rows" images and a lane's MFMA operand
//      (8 consecutive k of one row / column) is one ds_read_b128.  Global layout [n][k/8][term][8]: the 48 B a
//      thread stages per step are contiguous and a row streams sequentially (6 B per element).
// LDS image per plane and step: [k/8][row][8 bf16] (octet planes padded by 16 B): conflict-free ds_read_b128.
#pragma once
#include "gemm_core.hpp"

namespace nsgp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) bf16x8 g_bf16x8;

constexpr int X3_BK = 16;                      // one MFMA k16 step per pipeline step
constexpr int X3_OCT = BM * 8 + 32;            // bf16 elements of one k-octet plane ([row][8]); the 64-B pad puts the two octets a
                                               // thread pair writes 16 banks apart (ds_write_b128: 8-lane groups, 32 banks)
constexpr int X3_PLANE = (X3_BK / 8) * X3_OCT; // one split plane of one operand, one step (2 octets)
constexpr int X3_STAGE = 6 * X3_PLANE;         // A0 A1 A2 B0 B1 B2   (24,768 B)
constexpr int X3_SMEM_BYTES = (2 * X3_STAGE * 2 > SMEM_BYTES) ? 2 * X3_STAGE * 2 : SMEM_BYTES;   // two stages; the fp32 epilogue re-layout needs 64 KB
static_assert(2 * X3_STAGE * 2 <= 66 * 1024, "two workgroups per CU");

struct X3Regs {
    f32x4 a[2];           // row t>>1, octet t&1: 8 consecutive fp32 of A
    bf16x8 b[3];          // the same (row, octet) of the three planes of Bt
};

__device__ __forceinline__ void x3_split(const f32x4 lo4, const f32x4 hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo4[e] : hi4[e - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p0[e] = h;
        p1[e] = m;
        p2[e] = (__bf16)r2;
    }
}

__device__ __forceinline__ void x3_load(const float* pa, const __bf16* pb, long k0, X3Regs& r) {
    r.a[0] = *(const gf32x4*)(pa + k0);
    r.a[1] = *(const gf32x4*)(pa + k0 + 4);
#pragma unroll
    for (int p = 0; p < 3; ++p) r.b[p] = *(const g_bf16x8*)(pb + 3 * k0 + p * 8);
}

__device__ __forceinline__ int x3_slot() { return (threadIdx.x & 1) * X3_OCT + (threadIdx.x >> 1) * 8; }

__device__ __forceinline__ void x3_write_a(__bf16* stage, const X3Regs& r) {
    bf16x8 p0, p1, p2;
    x3_split(r.a[0], r.a[1], p0, p1, p2);
    const int off = x3_slot();
    *reinterpret_cast<bf16x8*>(stage + 0 * X3_PLANE + off) = p0;
    *reinterpret_cast<bf16x8*>(stage + 1 * X3_PLANE + off) = p1;
    *reinterpret_cast<bf16x8*>(stage + 2 * X3_PLANE + off) = p2;
}
__device__ __forceinline__ void x3_write_b(__bf16* stage, const X3Regs& r) {
    const int off = x3_slot();
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(stage + (3 + p) * X3_PLANE + off) = r.b[p];
}

struct X3Frags {
    bf16x8 a[2][3], b[2][3];
};
__device__ __forceinline__ void x3_read(const __bf16* stage, int wm, int wn, X3Frags& f) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const __bf16* base = stage + h * X3_OCT;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.a[i][p] = *reinterpret_cast<const bf16x8*>(base + p * X3_PLANE + (wm * 64 + i * 32 + r) * 8);
            f.b[i][p] = *reinterpret_cast<const bf16x8*>(base + (3 + p) * X3_PLANE + (wn * 64 + i * 32 + r) * 8);
        }
}
// Terms (a_i, b_j) in the order smallest first -- the fp32 accumulator rounds them before a0 b0 swamps them -- two
// terms of every block per call, so that consecutive MFMAs belong to four independent accumulator chains.
template <int PAIR>
__device__ __forceinline__ void x3_terms(const X3Frags& f, f32x16 (&acc)[2][2]) {
    constexpr int AI[6] = {2, 1, 0, 1, 0, 0}, BJ[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int q = 2 * PAIR; q < 2 * PAIR + 2; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mi][AI[q]], f.b[ni][BJ[q]], acc[mi][ni], 0, 0, 0);
}

// acc += A[m0.., :] x B[:, n0..] with B given as the split planes of its transpose.  Whole tiles, K % 16 == 0,
// A rows 16-byte aligned.
// Pipeline: two LDS stages + three register sets.  In step t (stage t&1 holds k-chunk t): 8 MFMAs | split + write the A
// part of chunk t+1 into the other stage | 8 MFMAs | write its B part, refill that register set with chunk t+4 |
// 8 MFMAs | barrier.  A load is consumed three steps (~2 us) after it is issued.
// ABLATE (tools/bf16x3_bench.hip only; results are garbage for != 0): 1 = no global loads inside the loop, 2 = also no
// split / LDS writes, 3 = also no barrier.
template <int ABLATE = 0>
__device__ __forceinline__ void gemm_tile_bf16x3(const float* __restrict__ A, long lda, const __bf16* __restrict__ Bt, int K,
                                                 int m0, int n0, float* smem_f, f32x16 (&acc)[2][2]) {
    __bf16* smem = reinterpret_cast<__bf16*>(smem_f);
    const int t = threadIdx.x, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* pa = A + (long)(m0 + (t >> 1)) * lda + (t & 1) * 8;
    const __bf16* pb = Bt + (long)(n0 + (t >> 1)) * K * 3 + (t & 1) * 24;
    const int nk = K / X3_BK, last = nk - 1;
    X3Regs regs[3];
    x3_load(pa, pb, 0, regs[0]);
    x3_load(pa, pb, (long)min(1, last) * X3_BK, regs[1]);
    x3_load(pa, pb, (long)min(2, last) * X3_BK, regs[2]);
    x3_write_a(smem, regs[0]);
    x3_write_b(smem, regs[0]);
    x3_load(pa, pb, (long)min(3, last) * X3_BK, regs[0]);
    __syncthreads();
    auto step = [&](int kt, auto rb, auto s) {
        constexpr int RB = decltype(rb)::value, S = decltype(s)::value;   // stage being read, register set holding chunk kt+1
        const __bf16* cur = smem + RB * X3_STAGE;
        __bf16* nxt = smem + (1 - RB) * X3_STAGE;
        X3Frags f;
        x3_read(cur, wm, wn, f);
        x3_terms<0>(f, acc);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        if (ABLATE < 2) x3_write_a(nxt, regs[S]);
        x3_terms<1>(f, acc);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        if (ABLATE < 2) x3_write_b(nxt, regs[S]);
        if (ABLATE < 1) x3_load(pa, pb, (long)min(kt + 4, last) * X3_BK, regs[S]);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
        x3_terms<2>(f, acc);
        if (ABLATE < 3) __syncthreads();
    };
    int kt = 0;
    for (; kt + 5 < nk; kt += 6) {
        step(kt, IC<0>{}, IC<1>{});
        step(kt + 1, IC<1>{}, IC<2>{});
        step(kt + 2, IC<0>{}, IC<0>{});
        step(kt + 3, IC<1>{}, IC<1>{});
        step(kt + 4, IC<0>{}, IC<2>{});
        step(kt + 5, IC<1>{}, IC<0>{});
    }
    // tail (< 6 steps): the same rotation, stopping early
    if (kt < nk) { step(kt, IC<0>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<2>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<0>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<2>{}); ++kt; }
}

// P [K x N] fp32 row-major -> the split terms of its transpose, layout [n][k/8][term][8] bf16 (once per projector per task)
static __global__ __launch_bounds__(256) void nsgp_split_transpose_bf16x3_kernel(const float* __restrict__ P, int K, int N, __bf16* __restrict__ Bt) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8)
        tile[i][tx] = (k0 + i < K && n0 + tx < N) ? P[(long)(k0 + i) * N + n0 + tx] : 0.0f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        if (n < N && k < K) {
            const float x = tile[tx][i];
            const __bf16 h = (__bf16)x;
            const float r1 = x - (float)h;
            const __bf16 m = (__bf16)r1;
            const float r2 = r1 - (float)m;
            __bf16* dst = Bt + ((long)n * K + (k & ~7)) * 3 + (k & 7);
            dst[0] = h;
            dst[8] = m;
            dst[16] = (__bf16)r2;
        }
    }
}

}  // namespace nsgp
