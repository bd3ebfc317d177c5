// fp32-accurate projection GEMM on the fp16 matrix cores: two-term split.
//   a = a0 + a1 with a0 = fp16(a), a1 = fp16(a - a0): 2 x 11 = 22 mantissa bits per operand, and
//       a*b ~= a1 b0 + a0 b1 + a0 b0            (the dropped a1 b1 is <= 2^-22 |a b|)
//   is THREE v_mfma_f32_32x32x16_f16 per fp32-equivalent product, accumulated in fp32 -- half the MFMAs and two thirds of the
//   operand bytes of the three-term bf16 split this repo shipped until round 2, which the L2 -> CU path bounds.  Measured against fp64 on 4096^3
//   (a quarter of the entries 4-5 decades smaller than the rest): max error 1.4e-6 of max|C|, the fp32 MFMA's own level.
// fp16 has a 5-bit exponent, so each operand matrix carries ONE power-of-two scale that puts its largest magnitude in
// [2^13, 2^14) (fp16 max 65504): elements down to 2^-27 of the largest stay normal in a0, anything smaller loses only bits
// that are below 2^-24 of the largest; products and sums live in the fp32 accumulators.  The scales are exact to undo
// (the caller folds 1/(sa*sb) into its epilogue factor).
//   A  [M x K] fp32 row-major, scaled by `a_scale` and split while it is written to LDS;
//   Bt the two fp16 terms of the TRANSPOSED, pre-scaled right operand, global layout [n][k/8][term][8] (4 bytes per
//      element), written once per projector by nsgp_split_transpose_f16x2_kernel.
// Tile / pipeline: 128 x 128, 4 waves, k16 steps, [k/8][row][8] LDS images (octet planes
// padded by 64 B), two LDS stages + three register sets.
#pragma once
#include "gemm_core.hpp"

namespace nsgp {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) h16x8 g_h16x8;

constexpr int F2_BK = 16;
constexpr int F2_OCT = BM * 8 + 32;
constexpr int F2_PLANE = (F2_BK / 8) * F2_OCT;
constexpr int F2_STAGE = 4 * F2_PLANE;                // A0 A1 B0 B1  (16,896 fp16 = 33,792 B)
constexpr int F2_SMEM_BYTES = (2 * F2_STAGE * 2 > SMEM_BYTES) ? 2 * F2_STAGE * 2 : SMEM_BYTES;   // the fp32 epilogue re-layout needs 64 KB

// power-of-two scale that puts a largest magnitude with IEEE bit pattern `amax_bits` into [2^13, 2^14); 1 for an all-zero matrix
__host__ __device__ __forceinline__ float f2_scale_from_amax_bits(unsigned amax_bits) {
    const unsigned e = (amax_bits >> 23) & 0xffu;
    if ((amax_bits & 0x7fffffffu) == 0u || e == 0xffu) return 1.0f;      // zeros; inf / nan propagate on their own
    int se = 267 - (int)(e == 0 ? 1 : e);                                // biased exponent of 2^(13 - (e - 127))
    se = se > 254 ? 254 : se;
    union { unsigned u; float f; } c;
    c.u = (unsigned)se << 23;
    return c.f;
}

struct F2Regs {
    f32x4 a[2];
    h16x8 b[2];
};

__device__ __forceinline__ void f2_split(const f32x4 lo4, const f32x4 hi4, float scale, h16x8& p0, h16x8& p1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e < 4 ? lo4[e] : hi4[e - 4]) * scale;
        const _Float16 h = (_Float16)x;
        p0[e] = h;
        p1[e] = (_Float16)(x - (float)h);
    }
}

// acc += (a_scale * A[m0.., :]) x (pre-scaled B)[:, n0..].  Whole tiles, K % 16 == 0, A rows 16-byte aligned.
__device__ __forceinline__ void gemm_tile_f16x2(const float* __restrict__ A, long lda, const _Float16* __restrict__ Bt, int K,
                                                int m0, int n0, float a_scale, float* smem_f, f32x16 (&acc)[2][2]) {
    _Float16* smem = reinterpret_cast<_Float16*>(smem_f);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const float* pa = A + (long)(m0 + (t >> 1)) * lda + (t & 1) * 8;
    const _Float16* pb = Bt + (long)(n0 + (t >> 1)) * K * 2 + (t & 1) * 16;
    const int slot = (t & 1) * F2_OCT + (t >> 1) * 8;
    const int nk = K / F2_BK, last = nk - 1;
    auto load = [&](long k0, F2Regs& r) {
        r.a[0] = *(const gf32x4*)(pa + k0);
        r.a[1] = *(const gf32x4*)(pa + k0 + 4);
        r.b[0] = *(const g_h16x8*)(pb + 2 * k0);
        r.b[1] = *(const g_h16x8*)(pb + 2 * k0 + 8);
    };
    auto write_a = [&](_Float16* st, const F2Regs& r) {
        h16x8 p0, p1;
        f2_split(r.a[0], r.a[1], a_scale, p0, p1);
        *reinterpret_cast<h16x8*>(st + 0 * F2_PLANE + slot) = p0;
        *reinterpret_cast<h16x8*>(st + 1 * F2_PLANE + slot) = p1;
    };
    auto write_b = [&](_Float16* st, const F2Regs& r) {
        *reinterpret_cast<h16x8*>(st + 2 * F2_PLANE + slot) = r.b[0];
        *reinterpret_cast<h16x8*>(st + 3 * F2_PLANE + slot) = r.b[1];
    };
    F2Regs regs[3];
    load(0, regs[0]);
    load((long)min(1, last) * F2_BK, regs[1]);
    load((long)min(2, last) * F2_BK, regs[2]);
    write_a(smem, regs[0]);
    write_b(smem, regs[0]);
    load((long)min(3, last) * F2_BK, regs[0]);
    __syncthreads();
    // step kt: stage kt&1 holds chunk kt; register set (kt+1) % 3 holds chunk kt+1 and is refilled with chunk kt+4
    auto step = [&](int kt, auto rb, auto s) {
        constexpr int RB = decltype(rb)::value, S = decltype(s)::value;
        const _Float16* cur = smem + RB * F2_STAGE;
        _Float16* nxt = smem + (1 - RB) * F2_STAGE;
        const int r = lane & 31, h = lane >> 5;
        h16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][p] = *reinterpret_cast<const h16x8*>(cur + p * F2_PLANE + h * F2_OCT + (wm * 64 + i * 32 + r) * 8);
                fb[i][p] = *reinterpret_cast<const h16x8*>(cur + (2 + p) * F2_PLANE + h * F2_OCT + (wn * 64 + i * 32 + r) * 8);
            }
        // smallest terms first; consecutive MFMAs belong to four independent accumulator chains
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        write_a(nxt, regs[S]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        write_b(nxt, regs[S]);
        load((long)min(kt + 4, last) * F2_BK, regs[S]);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
        __syncthreads();
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
    if (kt < nk) { step(kt, IC<0>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<2>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<0>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<2>{}); ++kt; }
    __syncthreads();                                       // LDS is free for the caller's epilogue
}

// P [K x N] fp32 row-major -> the two fp16 terms of scale * P^T, layout [n][k/8][term][8] (once per projector per task)
static __global__ __launch_bounds__(256) void nsgp_split_transpose_f16x2_kernel(const float* __restrict__ P, int K, int N, float scale,
                                                                         _Float16* __restrict__ Bt) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        tile[i][tx] = (k0 + i < K && n0 + tx < N) ? P[(long)(k0 + i) * N + n0 + tx] : 0.0f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        if (n < N && k < K) {
            const float x = tile[tx][i] * scale;
            const _Float16 h = (_Float16)x;
            _Float16* dst = Bt + ((long)n * K + (k & ~7)) * 2 + (k & 7);
            dst[0] = h;
            dst[8] = (_Float16)(x - (float)h);
        }
    }
}

}  // namespace nsgp
