// K3: feature-covariance accumulation.  Replaces BRNullSpaceRunner.compute_cov + update_cov
// (mmdet/engine/runner/nsrunner_roi_replay.py:876-916, 923-934):
//     X = F.unfold(mean(x, 0, keepdim), k, padding, stride).permute(0,2,1).reshape(-1, D)   # [L x D]
//     C (=|+=) X^T X
// The reference materialises X (up to 619 MB for one FPN conv).  Here X is never built:
//   1. nsgp_batch_mean_pad_kernel  batch mean written into a zero-bordered image xm[Cin][Hp][Wp]
//      (so the gather needs no bounds tests);
//   2. nsgp_cov_syrk_kernel        implicit-im2col SYRK on the fp32 MFMA core: both operands are
//      "rows" images whose row d=(c,i,j) at column l=(oy,ox) is xm[c][oy*sh+i][ox*sw+j]; only the
//      128x128 tiles on or above the diagonal are computed.  Work is split stream-K style: the
//      flattened (tile, K-step) space is cut into P equal contiguous ranges, one per resident
//      workgroup (P <= 512 = 2 per CU), so every workgroup does the same number of K-steps whatever
//      D and L are (equal-sized (tile, L-chunk) units left up to half the chip idle: 513 units for
//      D=2304, 342 for the six layer3 3x3s).  A range covers the tail of one tile, whole tiles, and
//      the head of another; each such segment writes its 128x128 partial to slab (tile + workgroup);
//   3. nsgp_cov_reduce_kernel      per tile: C (=|+=) sum of its segments' slabs in workgroup order
//      (deterministic; no float atomics), mirrored into the lower triangle through LDS.
#include <algorithm>
#include <new>
#include <vector>

#include "common.hpp"
#include "gemm_core.hpp"
#include "gemm_f16x2.hpp"
#include "gemm_f16x2_v2.hpp"

namespace nsgp {

__global__ __launch_bounds__(256) void nsgp_batch_mean_pad_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                                                  int ph, int pw, float* __restrict__ xm) {
    const int Hp = H + 2 * ph, Wp = W + 2 * pw;
    const long n = (long)C * Hp * Wp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int xx = (int)(idx % Wp), yy = (int)((idx / Wp) % Hp), c = (int)(idx / ((long)Wp * Hp));
        float v = 0.0f;
        if (yy >= ph && yy < ph + H && xx >= pw && xx < pw + W) {
            const long off = ((long)c * H + (yy - ph)) * W + (xx - pw);
            const long bs = (long)C * H * W;
            float s = x[off];
            for (int b = 1; b < B; ++b) s += x[off + b * bs];  // torch.mean(x, 0, True): sum over b, then / B
            v = (B > 1) ? s / (float)B : s;
        }
        xm[idx] = v;
    }
}

struct ConvGeom {
    int D, L, Wo, kh, kw, sh, sw, Hp, Wp;
};

// Offsets (in floats, relative to a row's base) of the 4 consecutive output positions l0+e this thread
// stages in one K-step: l = oy*Wo + ox  ->  oy*sh*Wp + ox*sw.  The division by Wo uses a float reciprocal
// with an exact integer correction (l < 2^24 always: L <= 268,800).  `contig` = the four positions sit in
// one output row of a stride-1 convolution, i.e. 4 consecutive floats in memory.
struct PatchCols {
    long off[4];   // -1 = past the end of this K range
    bool contig;
};

__device__ __forceinline__ PatchCols patch_cols(const ConvGeom& g, float inv_wo, int l0, int l_end) {
    PatchCols pc;
    const int l = l0 + (threadIdx.x & 7) * 4;
    int oy = (int)((float)l * inv_wo);
    oy -= (oy * g.Wo > l);
    oy += ((oy + 1) * g.Wo <= l);
    const int ox = l - oy * g.Wo;
    const long row = (long)oy * g.sh * g.Wp;
    pc.contig = (g.sw == 1) && (ox + 3 < g.Wo) && (l + 3 < l_end);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int oxe = ox + e;
        long r = row;
        while (oxe >= g.Wo) { oxe -= g.Wo; r += (long)g.sh * g.Wp; }   // at most once when Wo >= 4
        pc.off[e] = (l + e < l_end) ? (r + (long)oxe * g.sw) : -1;
    }
    return pc;
}

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) f32x4_u gf32x4_u;

// Stage the 4 rows x 4 columns this thread owns of an implicit X^T tile (rows = d, columns = l).
__device__ __forceinline__ void stage_im2col(const float* __restrict__ xm, const long (&rowbase)[4], const PatchCols& pc,
                                             float (&r)[4][4]) {
    if (pc.contig) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (rowbase[j] >= 0) {
                const f32x4_u q = *(const gf32x4_u*)(xm + rowbase[j] + pc.off[0]);   // 4-byte aligned 16-byte load
                r[j][0] = q[0]; r[j][1] = q[1]; r[j][2] = q[2]; r[j][3] = q[3];
            } else {
                r[j][0] = r[j][1] = r[j][2] = r[j][3] = 0.0f;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r[j][e] = (rowbase[j] >= 0 && pc.off[e] >= 0) ? as_global(xm)[rowbase[j] + pc.off[e]] : 0.0f;
    }
}

__device__ __forceinline__ void im2col_rowbase(const ConvGeom& g, int d0, long (&rowbase)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = d0 + (t >> 3) + 32 * j;
        if (d < g.D) {
            const int kk = g.kh * g.kw;
            const int c = d / kk, rem = d - c * kk, i = rem / g.kw, jj = rem - i * g.kw;
            rowbase[j] = ((long)c * g.Hp + i) * g.Wp + jj;
        } else {
            rowbase[j] = -1;
        }
    }
}

// first flattened index of workgroup w's range, and the workgroup that owns flattened index gidx
__host__ __device__ __forceinline__ long sk_begin(long w, long G, long P) { return w * G / P; }
__host__ __device__ __forceinline__ long sk_owner(long gidx, long G, long P) { return ((gidx + 1) * P - 1) / G; }

__device__ __forceinline__ void tile_of(int t, int nb, int& ti, int& tj) {
    ti = 0;
    int rem = t;
    while (rem >= nb - ti) { rem -= nb - ti; ++ti; }
    tj = ti + rem;
}

__global__ __launch_bounds__(256, 2) void nsgp_cov_syrk_kernel(const float* __restrict__ xm, ConvGeom g, int nk, long G,
                                                               float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nb = (g.D + BM - 1) / BM;
    const long P = gridDim.x, w = blockIdx.x;
    long gi = sk_begin(w, G, P);
    const long g_end = sk_begin(w + 1, G, P);
    const float inv_wo = 1.0f / (float)g.Wo;
    float ra[2][4][4], rb[2][4][4];
    while (gi < g_end) {                                       // uniform across the workgroup
        const int t = (int)(gi / nk);
        const int ja = (int)(gi - (long)t * nk);
        const int jb = (int)min((long)nk, ja + (g_end - gi));
        int ti, tj;
        tile_of(t, nb, ti, tj);
        const int m0 = ti * BM, n0 = tj * BN;
        const int l_beg = ja * BK, l_end = min(jb * BK, g.L);
        f32x16 acc[2][2];
        zero_acc(acc);
        long base_a[4], base_b[4];
        im2col_rowbase(g, m0, base_a);
        im2col_rowbase(g, n0, base_b);
        mfma_pipeline<true>(
            jb - ja, smem, acc,
            [&](int step, auto s) {
                const PatchCols pc = patch_cols(g, inv_wo, l_beg + step * BK, l_end);   // shared by both operands
                stage_im2col(xm, base_a, pc, ra[decltype(s)::value]);
                stage_im2col(xm, base_b, pc, rb[decltype(s)::value]);
            },
            [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value]); },
            [&](float* img, int, auto s) { write_rows(img, rb[decltype(s)::value]); });
        // park the block in LDS and store float4 rows into this segment's slab (always a full 128x128: rows >= D are zero)
        float* out = slabs + ((long)t + w) * (BM * BN);
        acc_to_lds(smem, acc);
        for_each_row4(smem, [&](int r, int col, float4 v) {
            f32x4 q;
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            *(gf32x4*)(out + r * BN + col) = q;
        });
        __syncthreads();                                        // LDS is free again for the next segment's prologue
        gi += jb - ja;
    }
}

// ---- two-term fp16 split variant of the SYRK (gemm_f16x2.hpp): three fp16 MFMAs per fp32-equivalent product ---------
// Both operands are the implicit X^T, so ONE power-of-two scale s (largest |activation| of the batch mean -> [2^13, 2^14)) serves
// both and the reduce kernel undoes s^2 exactly.  Steps are k16 (16 output positions); thread t stages row (t>>1), positions
// 8*(t&1) .. +7 of each operand as fp32 and splits them while writing to LDS.
__global__ __launch_bounds__(256) void nsgp_amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    float am = 0.0f;
    const long n4 = (((uintptr_t)x & 15u) == 0) ? n / 4 : 0;      // float4 body when the base is 16-byte aligned
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 q = *(const gf32x4*)(x + 4 * i);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) am = fmaxf(am, fabsf(as_global(x)[i]));
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    __shared__ float wmax[4];                                   // one atomic per workgroup: thousands of them on one address serialise
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

struct Patch8 {
    long off[8];     // -1 = past the end of this K range
    bool contig[2];  // each group of 4 positions sits in one output row of a stride-1 convolution
};

__device__ __forceinline__ Patch8 patch8(const ConvGeom& g, float inv_wo, int l, int l_end) {
    Patch8 pc;
    int oy = (int)((float)l * inv_wo);
    oy -= (oy * g.Wo > l);
    oy += ((oy + 1) * g.Wo <= l);
    int ox = l - oy * g.Wo;
    long row = (long)oy * g.sh * g.Wp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if ((e & 3) == 0) pc.contig[e >> 2] = (g.sw == 1) && (ox + 3 < g.Wo) && (l + e + 3 < l_end);
        pc.off[e] = (l + e < l_end) ? (row + (long)ox * g.sw) : -1;
        if (++ox >= g.Wo) { ox = 0; row += (long)g.sh * g.Wp; }
    }
    return pc;
}

__device__ __forceinline__ void stage8(const float* __restrict__ xm, long rowbase, const Patch8& pc, f32x4 (&r)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (rowbase >= 0 && pc.contig[h]) {
            const f32x4_u q = *(const gf32x4_u*)(xm + rowbase + pc.off[4 * h]);
            r[h][0] = q[0]; r[h][1] = q[1]; r[h][2] = q[2]; r[h][3] = q[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r[h][e] = (rowbase >= 0 && pc.off[4 * h + e] >= 0) ? as_global(xm)[rowbase + pc.off[4 * h + e]] : 0.0f;
        }
    }
}

__device__ __forceinline__ long im2col_rowbase1(const ConvGeom& g, int d) {
    if (d >= g.D) return -1;
    const int kk = g.kh * g.kw;
    const int c = d / kk, rem = d - c * kk, i = rem / g.kw, jj = rem - i * g.kw;
    return ((long)c * g.Hp + i) * g.Wp + jj;
}

struct CovF2Regs {
    f32x4 a[2], b[2];
};

__global__ __launch_bounds__(256, 2) void nsgp_cov_syrk_f16_kernel(const float* __restrict__ xm, ConvGeom g, int nk, long G,
                                                                   const unsigned* __restrict__ amax, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16* smem = reinterpret_cast<_Float16*>(smem_f);
    const int nb = (g.D + BM - 1) / BM;
    const long P = gridDim.x, w = blockIdx.x;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1;
    const float scale = f2_scale_from_amax_bits(*amax);
    const float inv_wo = 1.0f / (float)g.Wo;
    const int slot = (t & 1) * F2_OCT + (t >> 1) * 8;
    long gi = sk_begin(w, G, P);
    const long g_end = sk_begin(w + 1, G, P);
    while (gi < g_end) {                                       // uniform across the workgroup
        const int tile = (int)(gi / nk);
        const int ja = (int)(gi - (long)tile * nk);
        const int jb = (int)min((long)nk, ja + (g_end - gi));
        int ti, tj;
        tile_of(tile, nb, ti, tj);
        const int m0 = ti * BM, n0 = tj * BN;
        const int l_beg = ja * F2_BK, l_end = min(jb * F2_BK, g.L);
        const int nsteps = jb - ja, last = nsteps - 1;
        const long base_a = im2col_rowbase1(g, m0 + (t >> 1)), base_b = im2col_rowbase1(g, n0 + (t >> 1));
        f32x16 acc[2][2];
        zero_acc(acc);
        auto load = [&](int step, CovF2Regs& r) {
            const Patch8 pc = patch8(g, inv_wo, l_beg + step * F2_BK + (t & 1) * 8, l_end);   // shared by both operands
            stage8(xm, base_a, pc, r.a);
            stage8(xm, base_b, pc, r.b);
        };
        auto write = [&](_Float16* st, const CovF2Regs& r) {
            h16x8 p0, p1;
            f2_split(r.a[0], r.a[1], scale, p0, p1);
            *reinterpret_cast<h16x8*>(st + 0 * F2_PLANE + slot) = p0;
            *reinterpret_cast<h16x8*>(st + 1 * F2_PLANE + slot) = p1;
            f2_split(r.b[0], r.b[1], scale, p0, p1);
            *reinterpret_cast<h16x8*>(st + 2 * F2_PLANE + slot) = p0;
            *reinterpret_cast<h16x8*>(st + 3 * F2_PLANE + slot) = p1;
        };
        CovF2Regs regs[3];
        load(0, regs[0]);
        load(min(1, last), regs[1]);
        load(min(2, last), regs[2]);
        write(smem, regs[0]);
        load(min(3, last), regs[0]);
        __syncthreads();
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
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
            write(nxt, regs[S]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
            load(min(kt + 4, last), regs[S]);
            __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 5 < nsteps; kt += 6) {
            step(kt, IC<0>{}, IC<1>{});
            step(kt + 1, IC<1>{}, IC<2>{});
            step(kt + 2, IC<0>{}, IC<0>{});
            step(kt + 3, IC<1>{}, IC<1>{});
            step(kt + 4, IC<0>{}, IC<2>{});
            step(kt + 5, IC<1>{}, IC<0>{});
        }
        if (kt < nsteps) { step(kt, IC<0>{}, IC<1>{}); ++kt; }
        if (kt < nsteps) { step(kt, IC<1>{}, IC<2>{}); ++kt; }
        if (kt < nsteps) { step(kt, IC<0>{}, IC<0>{}); ++kt; }
        if (kt < nsteps) { step(kt, IC<1>{}, IC<1>{}); ++kt; }
        if (kt < nsteps) { step(kt, IC<0>{}, IC<2>{}); ++kt; }
        float* out = slabs + ((long)tile + w) * (BM * BN);
        acc_to_lds(smem_f, acc);
        for_each_row4(smem_f, [&](int r, int col, float4 v) {
            f32x4 q;
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            *(gf32x4*)(out + r * BN + col) = q;
        });
        __syncthreads();
        gi += nsteps;
    }
}

// ---- second generation of the fp16-split SYRK (wide layers, D >= 512) -------------------------------------------------------------
// The first-generation kernel above gathers the implicit X^T tile by tile: every element is fetched (4-byte-aligned 16-byte
// loads at best) and split by each of the D / 128 tiles that use its row, and the kernel is bound by those memory instructions
// (fpn 3x3: 2.2 ms for 357 GFLOP of upper triangle).  With 288 GB of HBM the reference's own answer -- materialise the unfold -- is
// affordable, if it is materialised ONCE in the form the matrix cores want: X^T as the pre-tiled, pre-scaled two-term fp16
// operand of gemm_f16x2_v2.hpp ([d/64][l/8][term][d%64][8 halves], 4 bytes per element: 619 MB for the fpn 3x3 at 800 x 1344,
// exactly the size of the reference's fp32 unfold buffer).  Then C = X^T X is the projection kernel's tile -- LDS-DMA loaders,
// consumers with nothing but LDS reads and MFMAs -- over the 256 x 128 tiles that touch the upper triangle, split-K into S
// aligned ranges (workgroups of one range walk the same columns of X^T at the same time, so its row panels are shared through
// L2), slab per (tile, range), and an ordered reduce that unscales, keeps diagonal tiles bit-symmetric and mirrors.
//   nsgp_cov_im2col_split_kernel -> nsgp_cov_syrk_v2_kernel -> nsgp_cov_reduce_v2_kernel
__host__ __device__ __forceinline__ int cov2_pad_d(int D) { return (D + 127) / 128 * 128; }       // rows: whole 128-row tiles
__host__ __device__ __forceinline__ int cov2_pad_l(int L) { return (L + V2_BK - 1) / V2_BK * V2_BK; }
// tiles: row tile r covers 64-row blocks [4r, 4r + mb), mb = 4 or 2 (remainder); column tile c blocks [2c, 2c + 2); a tile is
// needed when its columns reach the row tile's first row: 2c >= 4r.
__host__ __device__ __forceinline__ int cov2_tiles(int Dp) {
    const int nblk = Dp / 64;
    int n = 0;
    for (int rb0 = 0; rb0 < nblk; rb0 += 4) n += (nblk - rb0) / 2;
    return n;
}
__host__ __device__ __forceinline__ void cov2_tile_of(int t, int Dp, int& rb0, int& cb0, int& mb) {
    const int nblk = Dp / 64;
    rb0 = 0;
    while (t >= (nblk - rb0) / 2) { t -= (nblk - rb0) / 2; rb0 += 4; }
    cb0 = rb0 + 2 * t;
    mb = nblk - rb0 >= 4 ? 4 : 2;
}
__host__ __device__ __forceinline__ int cov2_tile_index(int rb0, int cb0, int Dp) {
    const int nblk = Dp / 64;
    int n = 0;
    for (int r = 0; r < rb0; r += 4) n += (nblk - r) / 2;
    return n + (cb0 - rb0) / 2;
}

// one wave = one (64-row block, l-octet): lane = row d; 8 consecutive output positions per lane -> the two 16-byte pieces
__global__ __launch_bounds__(256) void nsgp_cov_im2col_split_kernel(const float* __restrict__ xm, ConvGeom g, int Dp, int Lp,
                                                                    const unsigned* __restrict__ amax, void* __restrict__ xt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x * 4 + wave;                 // l-octet
    if (o * 8 >= Lp) return;
    const int d = blockIdx.y * 64 + lane;
    const float scale = f2_scale_from_amax_bits(*amax);
    f32x4 r[2];
    r[0] = f32x4{0, 0, 0, 0};
    r[1] = f32x4{0, 0, 0, 0};
    if (d < g.D && o * 8 < g.L) {
        const Patch8 pc = patch8(g, 1.0f / (float)g.Wo, o * 8, g.L);
        stage8(xm, im2col_rowbase1(g, d), pc, r);
    }
    v2_store_pieces(xt, d, o, Lp, r[0], r[1], scale);
    (void)Dp;
}

__global__ __launch_bounds__(V2L_THREADS, 3) void nsgp_cov_syrk_v2_kernel(const void* __restrict__ xt, int Dp, int Lp, int S, int steps_per_split,
                                                                         float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tile = blockIdx.x / S, sp = blockIdx.x - tile * S;
    int rb0, cb0, mb;
    cov2_tile_of(tile, Dp, rb0, cb0, mb);
    const int nk = Lp / V2_BK;
    const int s0 = sp * steps_per_split;
    const int ns = min(steps_per_split, nk - s0);
    f32x16 acc[2][2];
    zero_acc(acc);
    if (ns > 0) {
        if (mb == 4) gemm_tile_f16x2_v2l<4>(xt, rb0, xt, cb0, Lp, smem_c, acc, s0, ns);
        else gemm_tile_f16x2_v2l<2>(xt, rb0, xt, cb0, Lp, smem_c, acc, s0, ns);
    }
    if ((int)(threadIdx.x >> 6) >= 2 * mb) return;
    float* smem = reinterpret_cast<float*>(smem_c);
    float* out = slabs + (size_t)blockIdx.x * (256 * 128);
    acc_to_lds(smem, acc);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for_each_row4(smem, [&](int r, int col, float4 v) {
        f32x4 q;
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
        *(gf32x4*)(out + r * 128 + col) = q;
    });
}

constexpr int RED_LD = BN + 1;
constexpr int RED_MAX_ROWS = 64;

// grid (128 x 128 upper-triangle tiles of C, bands): sums the S range slabs of the parent 256 x 128 tile in range order.
__global__ __launch_bounds__(256) void nsgp_cov_reduce_v2_kernel(const float* __restrict__ slabs, int D, int Dp, int S, int rows,
                                                                 float* __restrict__ cov, int accumulate, const unsigned* __restrict__ amax) {
    __shared__ float tile[RED_MAX_ROWS * RED_LD];
    const float sc = f2_scale_from_amax_bits(*amax);
    const float unscale = (1.0f / sc) * (1.0f / sc);
    const int nb = Dp / 128;
    int ti, tj;
    tile_of(blockIdx.x, nb, ti, tj);
    const int r0 = blockIdx.y * rows;
    const int m0 = ti * 128, n0 = tj * 128;
    if (m0 + r0 >= D) return;
    const int rb0 = (2 * ti) / 4 * 4, sub = (2 * ti - rb0) / 2;          // parent row tile and which 128-row half of its slab
    const float* base = slabs + (size_t)cov2_tile_index(rb0, 2 * tj, Dp) * S * (256 * 128) + (size_t)sub * 128 * 128;
    if (ti != tj) {
        for (int idx = threadIdx.x; idx < rows * 32; idx += 256) {
            const int r = idx >> 5, c4 = (idx & 31) * 4;
            const size_t off = (size_t)(r0 + r) * 128 + c4;
            f32x4 sum = *(const gf32x4*)(base + off);
            for (int s = 1; s < S; ++s) {
                const f32x4 v = *(const gf32x4*)(base + (size_t)s * (256 * 128) + off);
                sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r * RED_LD + c4 + e] = sum[e] * unscale;
        }
    } else {    // diagonal tile: (i,j) and (j,i) hold the same products in a different order -- take the upper triangle for both
        for (int idx = threadIdx.x; idx < rows * 128; idx += 256) {
            const int r = idx >> 7, c = idx & 127, ra = r0 + r;
            const size_t off = (c >= ra) ? (size_t)ra * 128 + c : (size_t)c * 128 + ra;
            float sum = as_global(base)[off];
            for (int s = 1; s < S; ++s) sum += as_global(base)[(size_t)s * (256 * 128) + off];
            tile[r * RED_LD + c] = sum * unscale;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * 128; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        if (m0 + r0 + r < D && n0 + c < D) {
            const long o = (long)(m0 + r0 + r) * D + n0 + c;
            const float v = tile[r * RED_LD + c];
            as_global(cov)[o] = accumulate ? (as_global(cov)[o] + v) : v;
        }
    }
    if (ti != tj) {
        for (int idx = threadIdx.x; idx < rows * 128; idx += 256) {
            const int c = idx / rows, r = idx - c * rows;
            if (m0 + r0 + r < D && n0 + c < D) {
                const long o = (long)(n0 + c) * D + m0 + r0 + r;
                const float v = tile[r * RED_LD + c];
                as_global(cov)[o] = accumulate ? (as_global(cov)[o] + v) : v;
            }
        }
    }
}

// grid (tiles, R): workgroup (t, y) owns rows [y*rows, (y+1)*rows) of upper-triangle tile t: sums that band over
// the tile's segments in workgroup order, writes it and its mirror.  R grows when there are few tiles -- a D=64
// layer is ONE tile cut into 512 segments, and a single workgroup would stream all 32 MB of slabs by itself.
__global__ __launch_bounds__(256) void nsgp_cov_reduce_kernel(const float* __restrict__ slabs, int D, int nk, long G, long P,
                                                              int rows, float* __restrict__ cov, int accumulate,
                                                              const unsigned* __restrict__ amax) {
    __shared__ float tile[RED_MAX_ROWS * RED_LD];
    float unscale = 1.0f;                                       // fp16 path: the operands were multiplied by s (a power of two)
    if (amax) { const float sc = f2_scale_from_amax_bits(*amax); unscale = (1.0f / sc) * (1.0f / sc); }
    const int nb = (D + BM - 1) / BM;
    const int t = blockIdx.x, r0 = blockIdx.y * rows;
    int ti, tj;
    tile_of(t, nb, ti, tj);
    const int m0 = ti * BM, n0 = tj * BN;
    if (m0 + r0 >= D) return;                                   // band entirely below the matrix edge
    const long w_first = sk_owner((long)t * nk, G, P), w_last = sk_owner((long)(t + 1) * nk - 1, G, P);
    if (ti != tj || !amax) {
        for (int idx = threadIdx.x; idx < rows * BN / 4; idx += 256) {
            const int r = idx / (BN / 4), c4 = (idx - r * (BN / 4)) * 4;
            const long off = (long)(r0 + r) * BN + c4;
            f32x4 sum = *(const gf32x4*)(slabs + ((long)t + w_first) * (BM * BN) + off);
            for (long w = w_first + 1; w <= w_last; ++w) {
                const f32x4 v = *(const gf32x4*)(slabs + ((long)t + w) * (BM * BN) + off);
                sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r * RED_LD + c4 + e] = sum[e] * unscale;
        }
    } else {
        // Diagonal tile of the split path: (i,j) and (j,i) hold the same three products per position in a different order,
        // so they can differ in the last bit.  Take the upper triangle for both: C stays exactly symmetric, as on the fp32 path.
        for (int idx = threadIdx.x; idx < rows * BN; idx += 256) {
            const int r = idx >> 7, c = idx & 127, ra = r0 + r;
            const long off = (c >= ra) ? (long)ra * BN + c : (long)c * BN + ra;
            float sum = as_global(slabs)[((long)t + w_first) * (BM * BN) + off];
            for (long w = w_first + 1; w <= w_last; ++w) sum += as_global(slabs)[((long)t + w) * (BM * BN) + off];
            tile[r * RED_LD + c] = sum * unscale;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * BN; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        if (m0 + r0 + r < D && n0 + c < D) {
            const long o = (long)(m0 + r0 + r) * D + n0 + c;
            const float v = tile[r * RED_LD + c];
            as_global(cov)[o] = accumulate ? (as_global(cov)[o] + v) : v;
        }
    }
    if (ti != tj) {                                             // mirror: element (r, c) of the tile lands at row n0+c, column m0+r
        for (int idx = threadIdx.x; idx < rows * BN; idx += 256) {
            const int c = idx / rows, r = idx - c * rows;       // r fastest: runs of `rows` consecutive floats
            if (m0 + r0 + r < D && n0 + c < D) {
                const long o = (long)(n0 + c) * D + m0 + r0 + r;
                const float v = tile[r * RED_LD + c];
                as_global(cov)[o] = accumulate ? (as_global(cov)[o] + v) : v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void nsgp_cov_linear_kernel(const float* __restrict__ x, int B, int Fdim,
                                                              float* __restrict__ cov, int accumulate) {
    const long n = (long)Fdim * Fdim;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx / Fdim), j = (int)(idx - (long)i * Fdim);
        float a = x[i], b = x[j];
        for (int k = 1; k < B; ++k) { a += x[(long)k * Fdim + i]; b += x[(long)k * Fdim + j]; }
        if (B > 1) { a = a / (float)B; b = b / (float)B; }
        const float v = a * b;
        cov[idx] = accumulate ? (cov[idx] + v) : v;
    }
}

// stream-K launch shape: G = tiles x K-steps flattened units over P workgroups (all resident: 2 per CU)
struct CovPlan {
    int nk;
    long tiles, G, P;
};

static int g_cov_gen2 = 1;     // fp16-split path: wide layers (D >= 512) take the materialised-operand / LDS-DMA generation; 0 keeps the gather kernel (tests, A/B)
static int g_cov_split = 1;    // 0: fp32 MFMA SYRK, 1: auto (default: the fp16 split where the extra amax launch pays), 2: always the fp16 split

static CovPlan cov_plan(int D, int L, int kstep) {
    const int nb = (D + BM - 1) / BM;
    CovPlan p;
    p.nk = (L + kstep - 1) / kstep;
    p.tiles = (long)nb * (nb + 1) / 2;
    p.G = p.tiles * p.nk;
    p.P = std::max<long>(1, std::min<long>(512, p.G / 16));    // at least 16 K-steps per workgroup (each segment costs a 64 KB slab)
    return p;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// second-generation split path: wide layers only (the padding to 128-row tiles and the 619 MB-class operand do not pay below)
struct Cov2Plan {
    bool use;
    int Dp, Lp, tiles, S, steps;
    size_t xt_bytes, slab_bytes;
};
static Cov2Plan cov2_plan(int D, int L) {
    Cov2Plan q{};
    q.use = D >= 512 && D % 64 == 0 && L >= 256;
    if (!q.use) return q;
    q.Dp = cov2_pad_d(D);
    q.Lp = cov2_pad_l(L);
    q.tiles = cov2_tiles(q.Dp);
    const int nk = q.Lp / V2_BK;
    // split-K ranges: one workgroup per CU at a time, so the launch takes ceil(tiles * S / 256) rounds of (steps per range + a
    // fixed prologue / epilogue cost of ~20 steps), and every range adds a 128 KiB slab per tile to write and reduce (~0.055
    // steps each).  Take the cheapest S <= 32 with at least 8 steps per range, preferring fewer ranges when within 5 %.
    // (fpn 3x3: 90 tiles x 2100 steps -> S = 5; layer1 3x3: 9 tiles x 2100 -> S = 28; layer3 3x3: 90 x 132 -> S = 2.)
    int bestS = 1;
    double best = -1.0;
    for (int S = 1; S <= 32 && nk / S >= 8; ++S) {
        const long rounds = ((long)q.tiles * S + 255) / 256;
        const double cost = (double)rounds * ((nk + S - 1) / S + 20) + 0.055 * S * q.tiles;
        if (best < 0 || cost < best * 0.95) { best = cost; bestS = S; }
    }
    q.steps = (nk + bestS - 1) / bestS;
    q.S = (nk + q.steps - 1) / q.steps;
    q.xt_bytes = align256(v2_operand_bytes(q.Dp, q.Lp));
    q.slab_bytes = (size_t)q.tiles * q.S * 256 * 128 * 4;
    return q;
}

}  // namespace nsgp

using namespace nsgp;

extern "C" size_t nsgp_cov_workspace_bytes(int cin, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw) {
    if (cin <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return 0;
    const int Hp = h + 2 * ph, Wp = w + 2 * pw;
    const int Ho = (Hp - kh) / sh + 1, Wo = (Wp - kw) / sw + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const int D = cin * kh * kw;
    const CovPlan p16 = cov_plan(D, Ho * Wo, F2_BK), p32 = cov_plan(D, Ho * Wo, BK);     // enough for either path
    const Cov2Plan q = cov2_plan(D, Ho * Wo);
    const size_t slabs1 = (size_t)(p32.tiles + std::max(p16.P, p32.P)) * BM * BN * 4;
    // [batch-mean image][slabs (either generation)][gen-2: the materialised split X^T][amax word]
    return align256((size_t)cin * Hp * Wp * 4) + align256(std::max(slabs1, q.use ? q.slab_bytes : (size_t)0)) + (q.use ? q.xt_bytes : 0) + 256;
}

extern "C" int nsgp_cov_accumulate_conv2d(const float* x, int batch, int cin, int h, int w, int kh, int kw, int sh,
                                          int sw, int ph, int pw, float* cov, int accumulate, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    if (!x || !cov || batch <= 0) return fail(NSGP_ERR_INVALID, "nsgp_cov_accumulate_conv2d: null pointer / empty batch");
    const size_t need = nsgp_cov_workspace_bytes(cin, h, w, kh, kw, sh, sw, ph, pw);
    if (need == 0) return fail(NSGP_ERR_INVALID, "nsgp_cov_accumulate_conv2d: bad geometry");
    if (!workspace || workspace_bytes < need) return fail(NSGP_ERR_WORKSPACE, "nsgp_cov_accumulate_conv2d: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int Hp = h + 2 * ph, Wp = w + 2 * pw;
    const int Ho = (Hp - kh) / sh + 1, Wo = (Wp - kw) / sw + 1;
    ConvGeom g{cin * kh * kw, Ho * Wo, Wo, kh, kw, sh, sw, Hp, Wp};
    // the split path costs two extra tiny launches (clear + amax): measured worth it from ~10^4 (tile, k32-step) units on
    const CovPlan p32 = cov_plan(g.D, g.L, BK);
    const Cov2Plan q = cov2_plan(g.D, g.L);
    // auto: the split pays from ~10^4 (tile, k32-step) units on, and -- with its second generation -- on every wide layer
    const bool split = g_cov_split == 2 || (g_cov_split == 1 && (p32.G >= 10000 || (q.use && g_cov_gen2)));
    const CovPlan p = cov_plan(g.D, g.L, split ? F2_BK : BK);
    const CovPlan p16 = cov_plan(g.D, g.L, F2_BK);
    const size_t slabs1 = (size_t)(p32.tiles + std::max(p16.P, p32.P)) * BM * BN * 4;
    void* xt = static_cast<char*>(workspace) + align256((size_t)cin * Hp * Wp * 4) + align256(std::max(slabs1, q.use ? q.slab_bytes : (size_t)0));
    float* xm = static_cast<float*>(workspace);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)cin * Hp * Wp * 4));
    unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + need - 256);
    const long n_img = (long)cin * Hp * Wp;
    if (batch == 1 && ph == 0 && pw == 0) xm = const_cast<float*>(x);   // the image IS its own batch mean: no copy (all 1x1 convs at batch 1)
    else hipLaunchKernelGGL(nsgp_batch_mean_pad_kernel, dim3((unsigned)std::min<long>(4096, (n_img + 255) / 256)), dim3(256), 0, stream,
                       x, batch, cin, h, w, ph, pw, xm);
    NSGP_LAUNCH_CHECK();
    if (split) {
        NSGP_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned), stream));
        hipLaunchKernelGGL(nsgp_amax_kernel, dim3((unsigned)std::min<long>(512, (n_img / 4 + 255) / 256 + 1)), dim3(256), 0, stream, xm, n_img, amax);
        NSGP_LAUNCH_CHECK();
        if (q.use && g_cov_gen2) {
            hipLaunchKernelGGL(nsgp_cov_im2col_split_kernel, dim3((unsigned)((q.Lp / 8 + 3) / 4), (unsigned)(q.Dp / 64)), dim3(256), 0, stream,
                               xm, g, q.Dp, q.Lp, amax, xt);
            NSGP_LAUNCH_CHECK();
            NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_cov_syrk_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM_BYTES));
            hipLaunchKernelGGL(nsgp_cov_syrk_v2_kernel, dim3((unsigned)(q.tiles * q.S)), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, xt, q.Dp, q.Lp, q.S, q.steps, slabs);
            NSGP_LAUNCH_CHECK();
            const int nb128 = q.Dp / 128;
            const long t128 = (long)nb128 * (nb128 + 1) / 2;
            int bands2 = 2;
            while (bands2 < 32 && t128 * bands2 < 256) bands2 *= 2;
            hipLaunchKernelGGL(nsgp_cov_reduce_v2_kernel, dim3((unsigned)t128, bands2), dim3(256), 0, stream, slabs, g.D, q.Dp, q.S, 128 / bands2, cov, accumulate, amax);
            NSGP_LAUNCH_CHECK();
            return NSGP_OK;
        }
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_cov_syrk_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM_BYTES));
        hipLaunchKernelGGL(nsgp_cov_syrk_f16_kernel, dim3((unsigned)p.P), dim3(THREADS), F2_SMEM_BYTES, stream, xm, g, p.nk, p.G, amax, slabs);
    } else {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_cov_syrk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(nsgp_cov_syrk_kernel, dim3((unsigned)p.P), dim3(THREADS), SMEM_BYTES, stream, xm, g, p.nk, p.G, slabs);
    }
    NSGP_LAUNCH_CHECK();
    int bands = 2;                                              // >= 256 reduce workgroups, bands of 64 .. 4 rows
    while (bands < 32 && p.tiles * bands < 256) bands *= 2;
    hipLaunchKernelGGL(nsgp_cov_reduce_kernel, dim3((unsigned)p.tiles, bands), dim3(256), 0, stream, slabs, g.D, p.nk, p.G, p.P,
                       BM / bands, cov, accumulate, split ? amax : nullptr);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

namespace nsgp {

// ---- grouped pass: every eligible hooked layer of ONE forward in a handful of launches ------------------------------------------
// One hooked forward of R-50-FPN is 61 accumulations; issued layer by layer they are 4-5 launches each, most of them far too small
// to fill 256 CUs, every wide layer needs split-K slabs + a reduce launch to fill the chip on its own, and each pays a memset +
// amax launch for its fp16 scale (round 2: 7.4 ms of back-to-back GPU time, 5.6 ms over four streams).  Here the hooks only stash
// their inputs; at the end of the forward ONE plan run does, for all layers with D % 64 == 0 together:
//   nsgp_cov_group_mean_kernel   batch mean + zero border, for the layers that need it (B > 1 or padding);
//   nsgp_cov_group_amax_kernel   largest |activation| per layer -> its power-of-two fp16 scale;
//   nsgp_cov_group_split_kernel  the pre-tiled, pre-scaled two-term fp16 operand of every layer (gemm_f16x2_v2.hpp);
//   nsgp_cov_group_syrk_kernel   ONE tile table over all layers, longest K first (in-order dispatch = LPT list scheduling on the
//                                256 one-workgroup CUs).  With thousands of tiles from 60 layers in one launch the chip is full
//                                without split-K: a tile contracts its whole L, and its epilogue unscales and writes (or accumulates
//                                into) C directly -- upper-triangle blocks plus their mirrors, diagonal blocks symmetrised through
//                                LDS.  Every element of C is written by exactly one tile in a fixed k order: bitwise reproducible.
// Layers the tile cannot take (D % 64 != 0: the 7x7 stem; Linear) stay on the single-layer entry points.
//
// CORRELATION FORM of the 3x3 / stride 1 / padding 1 convolutions (where almost all of the work is: the seventeen such layers are
// 90 % of an R-50-FPN forward's im2col tile-steps, the one on the stride-4 FPN level alone 35 %).  Their covariance is 81 blocks of C x C,
//     Cov[(c1,ky1,kx1), (c2,ky2,kx2)] = sum over output positions (y,x) of X[c1][y+ky1-1][x+kx1-1] X[c2][y+ky2-1][x+kx2-1],
// and a block depends on its two kernel taps almost only through their DIFFERENCE (dy, dx) = (ky2-ky1, kx2-kx1): summed over the
// EXTENDED position set y in [-1, H], x in [-1, W] (one ring more than the convolution has), every tap sweeps the whole image and
//     Cov_ext[(c1,ky1,kx1), (c2,ky2,kx2)] = R[dy,dx][c1,c2] := sum over all pixels q of X[c1][q] X[c2][q + (dy,dx)]      (zero outside).
// So  Cov = Cov_ext - Cov_ring:  25 shifted C x C correlations (13 up to transposition: R[-d] = R[d]^T) instead of 81 blocks
// (40.5 up to symmetry), minus the covariance of the ring's 2(H + W) + 4 positions.  In memory a shift is an OFFSET: with the image
// laid out flat at a row pitch Wq >= W + 4 (two zero rows above and below, zero columns in between), q + (dy,dx) = q + dy Wq + dx.
//   operand:   five copies of the flat image shifted by dx = -2..2, rows (copy j, channel c), pre-tiled as always:  5 C x (H+4) Wq
//              instead of 9 C x H W  (the LDS-DMA loader needs 16-byte-aligned pieces: dx cannot be an address offset, dy Wq can);
//   R tiles:   A = the dx = 0 copy (C rows), B = all copies read dy Wq / 32 k-steps further on -- the same tile function, a byte
//              offset on B.  dy = 0 needs dx >= 0 only.  13 C^2 instead of 40.5 C^2 products per position: ~2.5-2.8 x fewer
//              tile-steps net of the row-pitch padding.  Contractions are cut into ranges of <= CG_CORR_MAX_STEPS, raw sums to slabs;
//   nsgp_cov_corr_reduce_kernel    R[dy][dx][c1][c2] = ordered sum of the range slabs, and its transpose into R[-dy][-dx] (all 25 shifts);
//   nsgp_cov_corr_assemble_kernel  C[d1][d2] (=|+=) unscale * R[ky2-ky1][kx2-kx1][c1][c2] - strips: (d1,d2) and (d2,d1) read two copies of the same
//              number, C stays bit-symmetric;
//   ring:      at a ring position only one row or column of the 3x3 window is inside the image, so the ring's covariance lives in the tap
//              pairs of that row / column: four STRIP layers per parent (top y = -1: taps ky = 2; bottom y = H: ky = 0; left x = -1:
//              kx = 2; right x = W: kx = 0; rows (c, the free tap index), D = 3 C, L = W + 2 or H) go through the ordinary tiles of the
//              same launch into [3C x 3C] buffers, and the assemble pass subtracts them where they apply.
// Chosen per layer by tile-step count (plan creation; nsgp_cov_set_corr_mode forces it on or off for tests and studies).
struct CovGroupLayer {
    ConvGeom g;              // kinds 0 / 2: im2col geometry over the window; kind 1: D = 9 C, L = H W, Hp = H, Wp = W (the raw image)
    int kind;                // 0 im2col layer; 1 correlation form; 2 one of the four ring strips of a kind-1 layer (D = 3 C)
    int cin, H, W, batch;    // the hooked input [batch x cin x H x W]
    int oy, ox;              // window origin in image coordinates (window = g.Hp x g.Wp; outside the image: zero)
    int Dp, Lp;              // operand rows, contraction length (kind 1: pad128(5 Cp) rows, (H + 4) Wq)
    int needs_mean;          // 0: the input is its own batch mean and needs no border (B == 1 and no padding, or kind 1 with B == 1)
    int src;                 // slot of x / cov / accumulate / amax in the per-run tables (a strip: its parent's)
    int Wq, Cp;              // kind 1: flat row pitch (a multiple of 32), channels padded to 64
    int strip;               // kind 2: 0 top (y = -1), 1 bottom (y = H), 2 left (x = -1), 3 right (x = W)
    long xm_off, xt_off, r_off;   // byte offsets into the workspace (-1: none); r_off: kind 1 its R, kind 2 its [3C x 3C] strip covariance
    long s_off[4];           // kind 1: the r_off of its four strips
    long n_img;              // elements of the window image
};
struct CovGroupTile {
    int layer, rb0, cb0, mb;
    int step0, nsteps;       // the k32 steps this workgroup contracts (of the A operand)
    int bstep;               // the B operand is read this many k-steps further on (correlation tiles: dy Wq / 32)
    long slab;               // >= 0: index of the 256 x 128 slab this K range writes; -1: direct epilogue
};
struct CovCorrUnit {         // one R tile: its S range slabs -> R[dy][ca0 ..][col0 ..]
    int layer, dy, ca0, mb, col0, S;
    long slab0;
};

__device__ __forceinline__ int cov_group_find(const int* __restrict__ prefix, int n, int unit) {
    int lo = 0, hi = n;                                      // the layer with prefix[lo] <= unit < prefix[lo + 1]: uniform binary search
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (unit >= prefix[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

constexpr int CG_CHUNK = 256 * 64;       // elements per workgroup of the mean / amax launches (64 KB of fp32)
// A tile's contraction is ONE fp32 accumulator chain: over L = 67,200 positions (2,100 steps) its rounding reached 9.7e-6 of a row's
// maximum against fp64 -- the whole 1e-5 gate (profiles/r03/covariance_true_size.json, first form).  Contractions longer than this many
// steps are therefore cut into equal K ranges whose slabs an ordered reduce sums (a blocked summation: 2.5e-6).
constexpr int CG_MAX_STEPS = 600;
constexpr int CG_CORR_MAX_STEPS = 320;   // correlation tiles: few tiles per layer (26 for C = 256), so shorter ranges also balance the launch
constexpr int CG_SPLIT_OCTETS = 16;      // l-octets per workgroup of the operand-split launch (4 per wave)
constexpr int CG_ST_LD = CG_SPLIT_OCTETS * 8 + 4 + 1;      // floats per channel of the staged tile: 128 columns + a halo of 2 either side, odd pitch
constexpr int CG_ASM_ROWS = 8;           // rows of C per workgroup of the assemble launch

// dyn: [x pointers n][cov pointers n][accumulate flags n]
__device__ __forceinline__ const float* cg_x(const void* dyn, int i) { return reinterpret_cast<const float* const*>(dyn)[i]; }
__device__ __forceinline__ float* cg_cov(const void* dyn, int n, int i) { return reinterpret_cast<float* const*>(dyn)[n + i]; }
__device__ __forceinline__ int cg_acc(const void* dyn, int n, int i) { return reinterpret_cast<const int*>(reinterpret_cast<const float* const*>(dyn) + 2 * n)[i]; }

__global__ __launch_bounds__(256) void nsgp_cov_group_mean_kernel(const CovGroupLayer* __restrict__ layers, const int* __restrict__ prefix, int n,
                                                                  const void* __restrict__ dyn, char* __restrict__ ws) {
    const int li = cov_group_find(prefix, n, blockIdx.x);
    const CovGroupLayer L = layers[li];
    const float* __restrict__ x = cg_x(dyn, L.src);
    float* __restrict__ xm = reinterpret_cast<float*>(ws + L.xm_off);
    const int Hp = L.g.Hp, Wp = L.g.Wp;
    const long base = (long)(blockIdx.x - prefix[li]) * CG_CHUNK, end = min(base + CG_CHUNK, L.n_img);
    const long bs = (long)L.cin * L.H * L.W;
    // one division per thread, then an incremental (row, column) walk in steps of 256 elements
    long idx = base + threadIdx.x;
    if (idx >= end) return;
    int row = (int)(idx / Wp), xx = (int)(idx - (long)row * Wp);      // row = c * Hp + yy
    int c = row / Hp, yy = row - c * Hp;
    const int qr = 256 / Wp, rr = 256 - qr * Wp;
    for (; idx < end; idx += 256) {
        float v = 0.0f;
        const int iy = yy + L.oy, ix = xx + L.ox;
        if (iy >= 0 && iy < L.H && ix >= 0 && ix < L.W) {
            const long off = ((long)c * L.H + iy) * L.W + ix;
            float s = x[off];
            for (int b = 1; b < L.batch; ++b) s += x[off + b * bs];      // torch.mean(x, 0, True): sum over b, then / B
            v = (L.batch > 1) ? s / (float)L.batch : s;
        }
        xm[idx] = v;
        xx += rr;
        yy += qr;
        if (xx >= Wp) { xx -= Wp; ++yy; }
        while (yy >= Hp) { yy -= Hp; ++c; }
    }
}

__device__ __forceinline__ const float* cg_image(const CovGroupLayer& L, const void* dyn, const char* ws) {
    return L.needs_mean ? reinterpret_cast<const float*>(ws + L.xm_off) : cg_x(dyn, L.src);
}

__global__ __launch_bounds__(256) void nsgp_cov_group_amax_kernel(const CovGroupLayer* __restrict__ layers, const int* __restrict__ prefix, int n,
                                                                  const void* __restrict__ dyn, const char* __restrict__ ws, unsigned* __restrict__ amax) {
    const int li = cov_group_find(prefix, n, blockIdx.x);
    const CovGroupLayer L = layers[li];
    const float* __restrict__ xm = cg_image(L, dyn, ws);
    const long base = (long)(blockIdx.x - prefix[li]) * CG_CHUNK, end = min(base + CG_CHUNK, L.n_img);
    float am = 0.0f;
    if ((((uintptr_t)xm) & 15u) == 0 && end - base == CG_CHUNK) {
#pragma unroll 8
        for (int i = 0; i < CG_CHUNK / 1024; ++i) {
            const f32x4 q = *(const gf32x4*)(xm + base + 4 * (threadIdx.x + 256 * i));
            am = fmaxf(fmaxf(am, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) am = fmaxf(am, fabsf(as_global(xm)[i]));
    }
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax + li, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// one workgroup = one (64-row block, 16 l-octets) of one layer; gather path: one wave per octet as nsgp_cov_im2col_split_kernel
__global__ __launch_bounds__(256) void nsgp_cov_group_split_kernel(const CovGroupLayer* __restrict__ layers, const int* __restrict__ prefix, int n,
                                                                   const void* __restrict__ dyn, char* __restrict__ ws, const unsigned* __restrict__ amax) {
    const int li = cov_group_find(prefix, n, blockIdx.x);
    const CovGroupLayer L = layers[li];
    const int u = blockIdx.x - prefix[li];
    const int nob = (L.Lp / 8 + CG_SPLIT_OCTETS - 1) / CG_SPLIT_OCTETS;      // workgroups along l
    const int ob = u % nob, db = u / nob;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ xm = cg_image(L, dyn, ws);
    const int d = db * 64 + lane;
    const float scale = f2_scale_from_amax_bits(amax[L.src]);
    if (L.kind == 1) {
        // STAGED: row (copy j, channel c) of the correlation operand is the flat zero-bordered image F (element q = r Wq + s <-
        // X[c][r-2][s-2]) shifted by dx = j - 2, copy_j[q] = F[q + j - 2].  The workgroup owns 64 channels x 128 consecutive q: it
        // reads F[128 ob - 2 .. 128 ob + 130) of its channels as rows (~512 contiguous bytes per channel), parks them in LDS, and writes
        // ALL FIVE copies from there (lane = channel: a plane is 1 KiB contiguous).  Against one gather per copy (32 bytes per lane at a
        // stride of one channel image): 0.58 against 0.65 ms for the launch; the same staging for the 1x1 layers -- whose gather is
        // 32-byte aligned -- measured slower (0.71) and is not used (profiles/r03/cov_split_study.log).
        __shared__ float stage[64 * CG_ST_LD];
        const int cb = db;                                       // channel block (units of a kind-1 layer: Cp / 64 x nob)
        constexpr int tw = CG_SPLIT_OCTETS * 8 + 4;
        const float inv_wq = 1.0f / (float)L.Wq;
        const long chan = (long)L.H * L.W;
        for (int idx = threadIdx.x; idx < 64 * tw; idx += 256) {
            const int cl = idx / tw, t = idx - cl * tw, c = cb * 64 + cl;
            const int q = ob * (CG_SPLIT_OCTETS * 8) - 2 + t;
            float v = 0.0f;
            if (c < L.cin && q >= 0) {
                int r = (int)((float)q * inv_wq);                // q < 2^24: exact after the correction
                r -= (r * L.Wq > q);
                r += ((r + 1) * L.Wq <= q);
                const int iy = r - 2, ix = q - r * L.Wq - 2;
                if (iy >= 0 && iy < L.H && ix >= 0 && ix < L.W) v = as_global(xm)[c * chan + (long)iy * L.W + ix];
            }
            stage[cl * CG_ST_LD + t] = v;
        }
        __syncthreads();
        const int ncopy = 5 + (cb == 0 && L.Dp > 5 * L.Cp ? 1 : 0);      // + the zero block that pads 5 Cp rows to whole 128-row tiles
        for (int p = wave; p < ncopy * CG_SPLIT_OCTETS; p += 4) {
            const int j = p / CG_SPLIT_OCTETS, ol = p - j * CG_SPLIT_OCTETS, o = ob * CG_SPLIT_OCTETS + ol;
            if (o * 8 >= L.Lp) continue;                         // uniform per wave
            f32x4 v[2];
            v[0] = f32x4{0, 0, 0, 0};
            v[1] = f32x4{0, 0, 0, 0};
            if (j < 5) {
                const float* src = stage + lane * CG_ST_LD + ol * 8 + j;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = src[e];
            }
            v2_store_pieces<true>(ws + L.xt_off, j < 5 ? j * L.Cp + cb * 64 + lane : 5 * L.Cp + lane, o, L.Lp, v[0], v[1], scale);
        }
        return;
    }
    if (L.kind == 2) {
        // strip operand: row d = (c, k), column l = position along the strip.  top / bottom: X[c][0 | H-1][l + k - 2] (l = x + 1, k = kx);
        // left / right: X[c][l + k - 1][0 | W-1] (l = y, k = ky); zero outside the image
        const int c = d / 3, k = d - 3 * c;
        const float* __restrict__ img = xm + (long)c * L.H * L.W;
#pragma unroll
        for (int i = 0; i < CG_SPLIT_OCTETS / 4; ++i) {
            const int o = ob * CG_SPLIT_OCTETS + 4 * i + wave;
            if (o * 8 >= L.Lp) break;                            // uniform per wave
            f32x4 v[2];
            v[0] = f32x4{0, 0, 0, 0};
            v[1] = f32x4{0, 0, 0, 0};
            if (d < L.g.D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int l = o * 8 + e;
                    if (l >= L.g.L) break;
                    int iy, ix;
                    if (L.strip < 2) { iy = L.strip == 0 ? 0 : L.H - 1; ix = l + k - 2; }
                    else { iy = l + k - 1; ix = L.strip == 2 ? 0 : L.W - 1; }
                    if (iy >= 0 && iy < L.H && ix >= 0 && ix < L.W) v[e >> 2][e & 3] = as_global(img)[(long)iy * L.W + ix];
                }
            }
            v2_store_pieces<true>(ws + L.xt_off, d, o, L.Lp, v[0], v[1], scale);
        }
        return;
    }
    const long rowbase = im2col_rowbase1(L.g, d);
    const float inv_wo = 1.0f / (float)L.g.Wo;
#pragma unroll
    for (int i = 0; i < CG_SPLIT_OCTETS / 4; ++i) {
        const int o = ob * CG_SPLIT_OCTETS + 4 * i + wave;
        if (o * 8 >= L.Lp) break;                            // uniform per wave
        f32x4 r[2];
        r[0] = f32x4{0, 0, 0, 0};
        r[1] = f32x4{0, 0, 0, 0};
        if (d < L.g.D && o * 8 < L.g.L) {
            const Patch8 pc = patch8(L.g, inv_wo, o * 8, L.g.L);
            stage8(xm, rowbase, pc, r);
        }
        v2_store_pieces<true>(ws + L.xt_off, d, o, L.Lp, r[0], r[1], scale);
    }
}

constexpr int CG_TLD = 129;              // floats per row of the epilogue tile: transposed reads stay conflict-free
static_assert(256 * CG_TLD * 4 <= V2_SMEM_BYTES, "the epilogue tile must fit the ring");

__global__ __launch_bounds__(V2L_THREADS, 3) void nsgp_cov_group_syrk_kernel(const CovGroupTile* __restrict__ tiles, const CovGroupLayer* __restrict__ layers,
                                                                            int n, const void* __restrict__ dyn, const char* __restrict__ ws,
                                                                            const unsigned* __restrict__ amax, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const CovGroupTile t = tiles[blockIdx.x];
    const CovGroupLayer L = layers[t.layer];
    const char* xt = ws + L.xt_off;
    const char* xb = xt + (size_t)t.bstep * V2_STEP;             // the same operand, read bstep k-steps further on (0 but for correlation tiles)
    f32x16 acc[2][2];
    zero_acc(acc);
    if (t.mb == 4) gemm_tile_f16x2_v2l<4>(xt, t.rb0, xb, t.cb0, L.Lp, smem_c, acc, t.step0, t.nsteps);
    else gemm_tile_f16x2_v2l<2>(xt, t.rb0, xb, t.cb0, L.Lp, smem_c, acc, t.step0, t.nsteps);
    // every wave is back (the loaders too) and the ring is free
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (t.slab >= 0) {                                           // one K range of a long contraction / of an R tile: raw partial sums to its slab
        if (wave >= 2 * t.mb) return;
        float* smem = reinterpret_cast<float*>(smem_c);
        float* out = slabs + (size_t)t.slab * (256 * 128);
        acc_to_lds(smem, acc);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for_each_row4(smem, [&](int r, int col, float4 v) {
            *(gf32x4*)(out + r * 128 + col) = f32x4{v.x, v.y, v.z, v.w};
        });
        return;
    }
    // park the unscaled 256 (128) x 128 block in LDS
    float* T = reinterpret_cast<float*>(smem_c);
    const float sc = f2_scale_from_amax_bits(amax[L.src]);
    const float unscale = (1.0f / sc) * (1.0f / sc);
    if (wave < 2 * t.mb) {
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[(wm * 64 + mi * 32 + acc_row(r, lane)) * CG_TLD + wn * 64 + ni * 32 + (lane & 31)] = acc[mi][ni][r] * unscale;
    }
    __syncthreads();
    // a strip's covariance goes to its buffer in the workspace (the assemble pass of its parent reads it)
    gfloat* cov = as_global(L.kind == 2 ? reinterpret_cast<float*>(const_cast<char*>(ws) + L.r_off) : cg_cov(dyn, n, L.src));
    const bool accumulate = L.kind != 2 && cg_acc(dyn, n, L.src) != 0;
    const int D = L.g.D;
    for (int sb = 0; sb < t.mb / 2; ++sb) {                      // the 128 x 128 blocks of this tile
        const int ti = t.rb0 / 2 + sb, tj = t.cb0 / 2;
        if (ti > tj) continue;                                   // below the diagonal: the mirror of a block another tile owns
        const int m0 = ti * 128, n0 = tj * 128;
        const float* Tb = T + sb * 128 * CG_TLD;
        if (ti == tj) {
            // diagonal block: (i,j) and (j,i) hold the same products in a different order -- take the upper triangle for both
            for (int idx = threadIdx.x; idx < 128 * 128; idx += V2L_THREADS) {
                const int r = idx >> 7, c = idx & 127;
                if (m0 + r < D && n0 + c < D) {
                    const float v = (c >= r) ? Tb[r * CG_TLD + c] : Tb[c * CG_TLD + r];
                    const long o = (long)(m0 + r) * D + n0 + c;
                    cov[o] = accumulate ? (cov[o] + v) : v;
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < 128 * 128; idx += V2L_THREADS) {
                const int r = idx >> 7, c = idx & 127;
                if (m0 + r < D && n0 + c < D) {
                    const long o = (long)(m0 + r) * D + n0 + c;
                    const float v = Tb[r * CG_TLD + c];
                    cov[o] = accumulate ? (cov[o] + v) : v;
                }
            }
            for (int idx = threadIdx.x; idx < 128 * 128; idx += V2L_THREADS) {      // mirror: r fastest
                const int c = idx >> 7, r = idx & 127;
                if (m0 + r < D && n0 + c < D) {
                    const long o = (long)(n0 + c) * D + m0 + r;
                    const float v = Tb[r * CG_TLD + c];
                    cov[o] = accumulate ? (cov[o] + v) : v;
                }
            }
        }
    }
}

// R[dy+2][dx+2][ca][cb] = sum of the tile's range slabs, in range order (raw scaled sums; the assemble pass unscales), AND its transpose
// into the opposite shift, R[2-dy][2-dx][cb][ca]: the assemble pass then reads every tap pair along a row of some R (cb contiguous
// across lanes), and (d1,d2) / (d2,d1) read two copies of the same number.  Shift 0: the upper triangle serves both halves.
__global__ __launch_bounds__(256) void nsgp_cov_corr_reduce_kernel(const CovCorrUnit* __restrict__ units, const CovGroupLayer* __restrict__ layers,
                                                                   const float* __restrict__ slabs, char* __restrict__ ws) {
    const CovCorrUnit u = units[blockIdx.x];
    const CovGroupLayer L = layers[u.layer];
    gfloat* R = as_global(reinterpret_cast<float*>(ws + L.r_off));
    const int Cp = L.Cp;
    const long plane = (long)Cp * Cp;
    const float* __restrict__ base = slabs + (size_t)u.slab0 * (256 * 128);
    const int band_rows = u.mb * 64 / (int)gridDim.y;            // grid.y bands of rows: 26 tiles per 256-channel layer alone would leave most CUs idle
    for (int idx = threadIdx.x; idx < band_rows * 32; idx += 256) {
        const int r = blockIdx.y * band_rows + (idx >> 5), c4 = (idx & 31) * 4;
        const int ca = u.ca0 + r, col = u.col0 + c4;             // col = j Cp + cb; Cp % 64 == 0: the four columns share j
        if (ca >= Cp || col >= 5 * Cp) continue;                 // the junk half of a 64-channel layer's tile; column padding
        f32x4 sum = *(const gf32x4*)(base + r * 128 + c4);
        for (int s = 1; s < u.S; ++s) sum += *(const gf32x4*)(base + (size_t)s * (256 * 128) + r * 128 + c4);
        const int j = col / Cp, cb = col - j * Cp, dx = j - 2;
        if (u.dy == 0 && dx < 0) continue;                       // not needed (a tile that straddles the dx = -1 / 0 boundary)
        gfloat* fwd = R + ((long)(u.dy + 2) * 5 + j) * plane;
        gfloat* bwd = R + ((long)(2 - u.dy) * 5 + (2 - dx)) * plane;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (u.dy == 0 && dx == 0) {
                if (ca <= cb + e) { fwd[(long)ca * Cp + cb + e] = sum[e]; fwd[(long)(cb + e) * Cp + ca] = sum[e]; }
            } else {
                fwd[(long)ca * Cp + cb + e] = sum[e];
                bwd[(long)(cb + e) * Cp + ca] = sum[e];
            }
        }
    }
}

// C[d1][d2] (=|+=) unscale * R[ky2-ky1+2][kx2-kx1+2][c1][c2] - the strips that hold the tap pair.
// A workgroup owns CG_ASM_ROWS rows of C and walks the columns 64 channels (576 columns) at a time in two phases:
//   gather   one (row, column tap) pair per wave-iteration, lane = channel c2: R, the vertical and the horizontal strip are read ALONG c2
//            (256 contiguous bytes of an R row per wave; a stride of three floats in a strip row) and the value is parked in LDS at its
//            place in the row, column 9 c2 + tap (odd stride: conflict-free);
//   stream   the eight finished row pieces (2,304 B each) leave LDS as 16-byte pieces: C is read-modified-written coalesced, past L2.
// (First form: a thread per column, R and the strips gathered 4 bytes at a time in column order -- three scattered loads per element: 0.23 ms
// for the 14 layers of an R-50-FPN forward, 0.11 of it the stream.)
constexpr int CG_ASM_CC = 64;
__global__ __launch_bounds__(256) void nsgp_cov_corr_assemble_kernel(const int* __restrict__ corr_layers, const int* __restrict__ prefix, int ncorr,
                                                                     const CovGroupLayer* __restrict__ layers, int n, const void* __restrict__ dyn,
                                                                     const char* __restrict__ ws, const unsigned* __restrict__ amax) {
    __shared__ __attribute__((aligned(16))) float tile[CG_ASM_ROWS * CG_ASM_CC * 9];
    const int k = cov_group_find(prefix, ncorr, blockIdx.x);
    const CovGroupLayer L = layers[corr_layers[k]];
    const int D = L.g.D, ld3 = 3 * L.cin;
    const gfloat* R = as_global(reinterpret_cast<const float*>(ws + L.r_off));
    const gfloat* s_top = as_global(reinterpret_cast<const float*>(ws + L.s_off[0]));
    const gfloat* s_bot = as_global(reinterpret_cast<const float*>(ws + L.s_off[1]));
    const gfloat* s_lft = as_global(reinterpret_cast<const float*>(ws + L.s_off[2]));
    const gfloat* s_rgt = as_global(reinterpret_cast<const float*>(ws + L.s_off[3]));
    const float sc = f2_scale_from_amax_bits(amax[L.src]);
    const float unscale = (1.0f / sc) * (1.0f / sc);
    gfloat* cov = as_global(cg_cov(dyn, n, L.src));
    const bool accumulate = cg_acc(dyn, n, L.src) != 0;
    const int d1_0 = (blockIdx.x - prefix[k]) * CG_ASM_ROWS;      // D = 9 C, C % 64 == 0: D % CG_ASM_ROWS == 0, every workgroup has all its rows
    const long plane = (long)L.Cp * L.Cp;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int TW = CG_ASM_CC * 9;                             // columns of a chunk
    for (int c20 = 0; c20 < L.cin; c20 += CG_ASM_CC) {
        const int c2 = c20 + lane;
        for (int p = wave; p < CG_ASM_ROWS * 9; p += 4) {         // uniform: (row, column tap)
            const int row = p / 9, t2 = p - 9 * row, ky2 = t2 / 3, kx2 = t2 - 3 * ky2;
            const int d1 = d1_0 + row, c1 = d1 / 9, t1 = d1 - 9 * c1, ky1 = t1 / 3, kx1 = t1 - 3 * ky1;
            float v = R[((long)(ky2 - ky1 + 2) * 5 + (kx2 - kx1 + 2)) * plane + (long)c1 * L.Cp + c2] * unscale;
            float sub = 0.0f;                                     // vertical term first, then horizontal: (d1,d2) and (d2,d1) subtract the same numbers in the same order
            if (ky1 == ky2 && ky1 != 1) sub += (ky1 == 2 ? s_top : s_bot)[(long)(3 * c1 + kx1) * ld3 + 3 * c2 + kx2];
            if (kx1 == kx2 && kx1 != 1) sub += (kx1 == 2 ? s_lft : s_rgt)[(long)(3 * c1 + ky1) * ld3 + 3 * c2 + ky2];
            tile[row * TW + 9 * lane + t2] = v - sub;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < CG_ASM_ROWS * (TW / 4); idx += 256) {
            const int row = idx / (TW / 4), c4 = idx - row * (TW / 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * TW + 4 * c4);
            gf32x4* dst = (gf32x4*)(cov + (long)(d1_0 + row) * D + 9 * c20 + 4 * c4);      // C is touched once per forward: streamed past L2
            if (accumulate) v += __builtin_nontemporal_load(dst);
            __builtin_nontemporal_store(v, dst);
        }
        __syncthreads();
    }
}

}  // namespace nsgp

struct nsgp_cov_plan {
    int n = 0;                       // layers handed to create
    int n_group = 0;                 // of which on the grouped launches
    int n_slots = 0;                 // entries of the layer table: grouped layers + one ring layer per correlation-form layer
    int n_corr = 0;                  // layers in the correlation form
    std::vector<int> route;          // per layer: index into the grouped tables, or -1 (single-layer entry points)
    std::vector<char> is_corr;       // per table entry: correlation form (its C is accessed 16 bytes at a time)
    std::vector<int> ring_parent;   // per table entry: its parent's index (strips) or -1
    size_t ws_bytes = 0, amax_off = 0;
    int mean_units = 0, amax_units = 0, split_units = 0, n_tiles = 0, corr_units = 0, asm_units = 0;
    nsgp::CovGroupLayer* d_layers = nullptr;
    nsgp::CovGroupTile* d_tiles = nullptr;
    nsgp::CovCorrUnit* d_corr = nullptr;
    int* d_prefix = nullptr;         // prefix arrays: mean, amax, split (n_slots + 1 each), assemble (n_corr + 1), then the n_corr layer indices
    size_t dyn_bytes = 0;
    char* h_dyn[4] = {nullptr, nullptr, nullptr, nullptr};
    char* d_dyn[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_used[4] = {false, false, false, false};
    int slot = 0;
    double flops_upper = 0;          // sum over grouped layers of L * D * (D + 128): the upper-triangle work actually needed
    double tile_steps = 0;           // k32 steps of 256 x 128 tiles the SYRK launches execute (a 128-row tile counts half)
    size_t slab_off = 0;
    struct SplitLayer { int group_index, D, Dp, S; size_t slab_base; };   // im2col layers whose contraction is cut into S K ranges
    std::vector<SplitLayer> split_layers;
};

namespace nsgp {
static int g_cov_corr_mode = 1;      // 0: never the correlation form; 1: where it saves tile-steps (default); 2: wherever it applies

static bool cov_group_eligible(const nsgp_cov_geom_t& q, int& D, int& L, int& Wo, int& Hp, int& Wp) {
    if (q.cin <= 0 || q.h <= 0 || q.w <= 0 || q.kh <= 0 || q.kw <= 0 || q.sh <= 0 || q.sw <= 0 || q.ph < 0 || q.pw < 0 || q.batch <= 0) return false;
    Hp = q.h + 2 * q.ph;
    Wp = q.w + 2 * q.pw;
    const int Ho = (Hp - q.kh) / q.sh + 1;
    Wo = (Wp - q.kw) / q.sw + 1;
    if (Ho <= 0 || Wo <= 0) return false;
    D = q.cin * q.kh * q.kw;
    L = Ho * Wo;
    return D % 64 == 0 && L >= 32;
}
static int cov_group_ranges(int nk, int max_steps, int& steps) {      // equal K ranges of at most max_steps
    const int S0 = (nk + max_steps - 1) / max_steps;
    steps = (nk + S0 - 1) / S0;
    return (nk + steps - 1) / steps;
}
static double cov_im2col_tile_steps(int Dp, int Lp) {
    double w = 0;
    const int nt = cov2_tiles(Dp), nk = Lp / V2_BK;
    for (int t = 0; t < nt; ++t) {
        int rb0, cb0, mb;
        cov2_tile_of(t, Dp, rb0, cb0, mb);
        w += nk * (mb / 4.0);
    }
    return w;
}
// the correlation form's tiles: A row tiles over the dx = 0 copy, per dy the column tiles that hold a needed copy
struct CorrShape { int Cp, Wq, Kq, rows_q, nk, step_a0, bstep1, n_rt, ct0_dy0, nct; };
static CorrShape corr_shape(int C, int H, int W) {
    CorrShape s;
    s.Cp = (C + 63) / 64 * 64;
    s.Wq = (W + 4 + 31) / 32 * 32;
    s.Kq = (H + 4) * s.Wq;
    s.rows_q = (5 * s.Cp + 127) / 128 * 128;
    s.nk = H * s.Wq / V2_BK;                    // the image rows of the A operand: flat rows 2 .. H + 1
    s.step_a0 = 2 * s.Wq / V2_BK;
    s.bstep1 = s.Wq / V2_BK;
    s.n_rt = (s.Cp / 64 + 3) / 4;               // 256-channel row tiles (the last may have 1-2 blocks: mb = 2)
    s.ct0_dy0 = (2 * s.Cp / 64) / 2;            // dy = 0 needs the copies dx >= 0 only: column tiles from the one holding block 2 Cp / 64
    s.nct = s.rows_q / 128;
    return s;
}
static int corr_tile_mb(const CorrShape& s, int rt) { return (s.Cp / 64 - 4 * rt) >= 3 ? 4 : 2; }      // 3 blocks left: a full tile, its last block junk (never read back)
static double corr_tile_steps(const CorrShape& s) {
    double w = 0;
    for (int rt = 0; rt < s.n_rt; ++rt) w += (corr_tile_mb(s, rt) / 4.0) * s.nk * ((s.nct - s.ct0_dy0) + 2 * s.nct);
    return w;
}
}  // namespace nsgp

using namespace nsgp;

extern "C" int nsgp_cov_plan_destroy(nsgp_cov_plan_t* P) {
    if (!P) return NSGP_OK;
    hipError_t first = hipSuccess;
    auto keep = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    for (int s = 0; s < 4; ++s)
        if (P->ev[s] && P->ev_used[s]) keep(hipEventSynchronize(P->ev[s]));
    for (int s = 0; s < 4; ++s) {
        if (P->ev[s]) keep(hipEventDestroy(P->ev[s]));
        if (P->h_dyn[s]) keep(hipHostFree(P->h_dyn[s]));
        if (P->d_dyn[s]) keep(hipFree(P->d_dyn[s]));
    }
    if (P->d_layers) keep(hipFree(P->d_layers));
    if (P->d_tiles) keep(hipFree(P->d_tiles));
    if (P->d_corr) keep(hipFree(P->d_corr));
    if (P->d_prefix) keep(hipFree(P->d_prefix));
    delete P;
    if (first != hipSuccess) return fail(NSGP_ERR_HIP, "nsgp_cov_plan_destroy: %s", hipGetErrorString(first));
    return NSGP_OK;
}

extern "C" int nsgp_cov_plan_create(nsgp_cov_plan_t** out, const nsgp_cov_geom_t* geoms, int n) {
    if (!out || !geoms || n <= 0) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_create: null / empty layer list");
    nsgp_cov_plan* P = new (std::nothrow) nsgp_cov_plan();
    if (!P) return fail(NSGP_ERR_INVALID, "out of host memory");
    P->n = n;
    P->route.assign(n, -1);
    std::vector<CovGroupLayer> ld;
    for (int i = 0; i < n; ++i) {
        const nsgp_cov_geom_t& q = geoms[i];
        int D, L, Wo, Hp, Wp;
        if (!cov_group_eligible(q, D, L, Wo, Hp, Wp)) continue;
        CovGroupLayer c{};
        c.cin = q.cin; c.H = q.h; c.W = q.w; c.batch = q.batch;
        c.src = (int)ld.size();
        c.xm_off = c.xt_off = c.r_off = -1;
        // correlation form?  3 x 3, stride 1, padding 1, whole 64-channel blocks; by rule: its tile-steps (R tiles + the four strips, plus
        // the assemble pass priced at 5.3e-4 tile-steps per element of C: 12 B at ~4 TB/s against 5.7 ns per tile-step) under 0.8 x the im2col tiles'
        bool corr = g_cov_corr_mode != 0 && q.kh == 3 && q.kw == 3 && q.sh == 1 && q.sw == 1 && q.ph == 1 && q.pw == 1 && q.cin % 64 == 0 && q.h >= 4 && q.w >= 4;
        if (corr && g_cov_corr_mode == 1) {
            const CorrShape s = corr_shape(q.cin, q.h, q.w);
            const double old_steps = cov_im2col_tile_steps(cov2_pad_d(D), cov2_pad_l(L));
            const double ring = 2 * cov_im2col_tile_steps(cov2_pad_d(3 * q.cin), cov2_pad_l(q.w + 2)) + 2 * cov_im2col_tile_steps(cov2_pad_d(3 * q.cin), cov2_pad_l(q.h));
            const double new_steps = corr_tile_steps(s) + ring + 5.3e-4 * (double)D * D;
            corr = new_steps < 0.8 * old_steps;
        }
        P->route[i] = (int)ld.size();
        P->flops_upper += (double)L * D * (D + 128.0);
        if (!corr) {
            c.kind = 0;
            c.g = ConvGeom{D, L, Wo, q.kh, q.kw, q.sh, q.sw, Hp, Wp};
            c.oy = -q.ph; c.ox = -q.pw;
            c.Dp = cov2_pad_d(D);
            c.Lp = cov2_pad_l(L);
            c.needs_mean = !(q.batch == 1 && q.ph == 0 && q.pw == 0);
            c.n_img = (long)q.cin * Hp * Wp;
            ld.push_back(c);
            continue;
        }
        const CorrShape s = corr_shape(q.cin, q.h, q.w);
        c.kind = 1;
        c.g = ConvGeom{D, L, q.w, 3, 3, 1, 1, q.h, q.w};
        c.oy = c.ox = 0;
        c.Dp = s.rows_q;
        c.Lp = s.Kq;
        c.Wq = s.Wq; c.Cp = s.Cp;
        c.needs_mean = q.batch > 1;
        c.n_img = (long)q.cin * q.h * q.w;
        const int parent = (int)ld.size();
        ld.push_back(c);
        ++P->n_corr;
        // the ring: rows y = -1 and y = H (x in [-1, W]), columns x = -1 and x = W (y in [0, H - 1]) as four strip layers of D = 3 C
        for (int k = 0; k < 4; ++k) {
            CovGroupLayer r{};
            r.kind = 2;
            r.strip = k;
            r.cin = q.cin; r.H = q.h; r.W = q.w; r.batch = q.batch;
            r.g = ConvGeom{3 * q.cin, k < 2 ? q.w + 2 : q.h, 1, 3, 3, 1, 1, q.h, q.w};
            r.Dp = cov2_pad_d(r.g.D);
            r.Lp = cov2_pad_l(r.g.L);
            r.needs_mean = c.needs_mean;      // reads its parent's image (xm_off copied below)
            r.src = parent;
            r.xm_off = r.xt_off = r.r_off = -1;
            r.n_img = 0;
            ld.push_back(r);
        }
    }
    P->n_group = 0;
    for (int i = 0; i < n; ++i) P->n_group += P->route[i] >= 0;
    P->n_slots = (int)ld.size();
    P->ring_parent.assign(ld.size(), -1);
    P->is_corr.assign(ld.size(), 0);
    for (size_t li = 0; li < ld.size(); ++li) P->is_corr[li] = ld[li].kind == 1;
    size_t off = 0;
    for (size_t li = 0; li < ld.size(); ++li) {
        CovGroupLayer& c = ld[li];
        if (c.kind == 2) { P->ring_parent[li] = c.src; c.xm_off = ld[c.src].xm_off; continue; }      // a strip follows its parent in the table
        if (c.needs_mean) { c.xm_off = (long)off; off += align256((size_t)c.n_img * 4); }
    }
    for (CovGroupLayer& c : ld) { c.xt_off = (long)off; off += align256(v2_operand_bytes(c.Dp, c.Lp)); }
    for (size_t li = 0; li < ld.size(); ++li) {
        CovGroupLayer& c = ld[li];
        if (c.kind == 1) { c.r_off = (long)off; off += align256((size_t)25 * c.Cp * c.Cp * 4); }
        if (c.kind == 2) {
            c.r_off = (long)off;
            off += align256((size_t)c.g.D * c.g.D * 4);
            ld[c.src].s_off[c.strip] = c.r_off;
        }
    }
    P->amax_off = off;
    off += align256(std::max<size_t>(1, ld.size()) * sizeof(unsigned));
    P->ws_bytes = off;
    std::vector<int> pm{0}, pa{0}, ps{0}, pasm{0}, corr_ids;
    std::vector<CovGroupTile> tiles;
    std::vector<CovCorrUnit> cunits;
    size_t n_slabs = 0;
    for (size_t li = 0; li < ld.size(); ++li) {
        const CovGroupLayer& c = ld[li];
        const int chunks = (int)((c.n_img + CG_CHUNK - 1) / CG_CHUNK);
        pm.push_back(pm.back() + (c.needs_mean && c.kind != 2 ? chunks : 0));
        pa.push_back(pa.back() + (c.kind == 2 ? 0 : chunks));
        ps.push_back(ps.back() + ((c.Lp / 8 + CG_SPLIT_OCTETS - 1) / CG_SPLIT_OCTETS) * ((c.kind == 1 ? c.Cp : c.Dp) / 64));      // kind 1: a workgroup writes all five copies of its channels
        if (c.kind == 1) {
            const CorrShape s = corr_shape(c.cin, c.H, c.W);
            int steps;
            const int S = cov_group_ranges(s.nk, CG_CORR_MAX_STEPS, steps);
            for (int rt = 0; rt < s.n_rt; ++rt) {
                const int mb = corr_tile_mb(s, rt);
                for (int dy = 0; dy < 3; ++dy)
                    for (int ct = (dy == 0 ? s.ct0_dy0 : 0); ct < s.nct; ++ct) {
                        cunits.push_back(CovCorrUnit{(int)li, dy, rt * 256, mb, ct * 128, S, (long)n_slabs});
                        for (int sp = 0; sp < S; ++sp)
                            tiles.push_back(CovGroupTile{(int)li, 2 * s.Cp / 64 + 4 * rt, 2 * ct, mb, s.step_a0 + sp * steps, std::min(steps, s.nk - sp * steps),
                                                         dy * s.bstep1, (long)(n_slabs + sp)});
                        n_slabs += S;
                        P->tile_steps += (mb / 4.0) * s.nk;
                    }
            }
            corr_ids.push_back((int)li);
            pasm.push_back(pasm.back() + (c.g.D + CG_ASM_ROWS - 1) / CG_ASM_ROWS);
            continue;
        }
        const int nt = cov2_tiles(c.Dp), nk = c.Lp / V2_BK;
        int steps;
        const int S = c.kind == 2 ? 1 : cov_group_ranges(nk, CG_MAX_STEPS, steps);
        if (c.kind == 2) steps = nk;
        if (S > 1) P->split_layers.push_back(nsgp_cov_plan::SplitLayer{(int)li, c.g.D, c.Dp, S, n_slabs});
        std::vector<CovGroupTile>& dst = tiles;
        for (int t = 0; t < nt; ++t) {
            int rb0, cb0, mb;
            cov2_tile_of(t, c.Dp, rb0, cb0, mb);
            P->tile_steps += (mb / 4.0) * nk;
            if (S == 1) dst.push_back(CovGroupTile{(int)li, rb0, cb0, mb, 0, nk, 0, -1});
            else
                for (int sp = 0; sp < S; ++sp)      // slab layout of nsgp_cov_reduce_v2_kernel: [tile][range]
                    dst.push_back(CovGroupTile{(int)li, rb0, cb0, mb, sp * steps, std::min(steps, nk - sp * steps), 0, (long)(n_slabs + (size_t)t * S + sp)});
        }
        if (S > 1) n_slabs += (size_t)nt * S;
    }
    // longest contraction first: in-order dispatch is then LPT list scheduling on the one-workgroup CUs
    auto by_cost = [](const CovGroupTile& a, const CovGroupTile& b) { return (long)a.nsteps * a.mb > (long)b.nsteps * b.mb; };
    std::stable_sort(tiles.begin(), tiles.end(), by_cost);
    P->slab_off = P->ws_bytes;
    P->ws_bytes += align256(n_slabs * (size_t)(256 * 128) * 4);
    P->mean_units = pm.back();
    P->amax_units = pa.back();
    P->split_units = ps.back();
    P->n_tiles = (int)tiles.size();
    P->corr_units = (int)cunits.size();
    P->asm_units = pasm.back();
    P->dyn_bytes = align256((size_t)std::max(1, P->n_slots) * (2 * sizeof(void*) + sizeof(int)));
#define COVP_HIP(call)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            nsgp_cov_plan_destroy(P);                                                                \
            return fail(NSGP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                \
        }                                                                                            \
    } while (0)
    if (P->n_slots > 0) {
        COVP_HIP(hipMalloc(&P->d_layers, sizeof(CovGroupLayer) * ld.size()));
        COVP_HIP(hipMemcpy(P->d_layers, ld.data(), sizeof(CovGroupLayer) * ld.size(), hipMemcpyHostToDevice));
        COVP_HIP(hipMalloc(&P->d_tiles, sizeof(CovGroupTile) * tiles.size()));
        COVP_HIP(hipMemcpy(P->d_tiles, tiles.data(), sizeof(CovGroupTile) * tiles.size(), hipMemcpyHostToDevice));
        if (!cunits.empty()) {
            COVP_HIP(hipMalloc(&P->d_corr, sizeof(CovCorrUnit) * cunits.size()));
            COVP_HIP(hipMemcpy(P->d_corr, cunits.data(), sizeof(CovCorrUnit) * cunits.size(), hipMemcpyHostToDevice));
        }
        std::vector<int> prefix(pm);
        prefix.insert(prefix.end(), pa.begin(), pa.end());
        prefix.insert(prefix.end(), ps.begin(), ps.end());
        prefix.insert(prefix.end(), pasm.begin(), pasm.end());
        prefix.insert(prefix.end(), corr_ids.begin(), corr_ids.end());
        COVP_HIP(hipMalloc(&P->d_prefix, sizeof(int) * prefix.size()));
        COVP_HIP(hipMemcpy(P->d_prefix, prefix.data(), sizeof(int) * prefix.size(), hipMemcpyHostToDevice));
        for (int s = 0; s < 4; ++s) {
            COVP_HIP(hipHostMalloc(reinterpret_cast<void**>(&P->h_dyn[s]), P->dyn_bytes, hipHostMallocDefault));
            COVP_HIP(hipMalloc(reinterpret_cast<void**>(&P->d_dyn[s]), P->dyn_bytes));
            COVP_HIP(hipEventCreateWithFlags(&P->ev[s], hipEventDisableTiming));
        }
        COVP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_cov_group_syrk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM_BYTES));
    }
#undef COVP_HIP
    *out = P;
    return NSGP_OK;
}

extern "C" size_t nsgp_cov_plan_workspace_bytes(const nsgp_cov_plan_t* P) { return P ? P->ws_bytes : 0; }

extern "C" int nsgp_cov_plan_routes(const nsgp_cov_plan_t* P, int* routes, int n) {
    if (!P || !routes || n != P->n) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_routes: bad argument");
    for (int i = 0; i < n; ++i) routes[i] = P->route[i] >= 0 ? 1 : 0;
    return NSGP_OK;
}

extern "C" int nsgp_cov_plan_stats(const nsgp_cov_plan_t* P, int* n_grouped, int* n_tiles, double* upper_flops) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_stats: null plan");
    if (n_grouped) *n_grouped = P->n_group;
    if (n_tiles) *n_tiles = P->n_tiles;
    if (upper_flops) *upper_flops = P->flops_upper;
    return NSGP_OK;
}

extern "C" int nsgp_cov_plan_forms(const nsgp_cov_plan_t* P, int* n_correlation_form, double* tile_steps) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_forms: null plan");
    if (n_correlation_form) *n_correlation_form = P->n_corr;
    if (tile_steps) *tile_steps = P->tile_steps;
    return NSGP_OK;
}

extern "C" int nsgp_cov_set_corr_mode(int mode) {
    const int prev = g_cov_corr_mode;
    g_cov_corr_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return prev;
}

extern "C" int nsgp_cov_plan_run(nsgp_cov_plan_t* P, const float* const* x, float* const* cov, const int* accumulate, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
    if (!P || !x || !cov || !accumulate) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_run: null argument");
    if (P->n_slots == 0) return NSGP_OK;
    if (!workspace || workspace_bytes < P->ws_bytes) return fail(NSGP_ERR_WORKSPACE, "nsgp_cov_plan_run: workspace %zu < %zu", workspace_bytes, P->ws_bytes);
    if (!aligned16(workspace)) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_run: workspace must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int s = P->slot;
    P->slot = (s + 1) % 4;
    if (P->ev_used[s]) NSGP_HIP(hipEventSynchronize(P->ev[s]));
    const int ng = P->n_slots;
    const float** hx = reinterpret_cast<const float**>(P->h_dyn[s]);
    float** hc = reinterpret_cast<float**>(P->h_dyn[s]) + ng;
    int* ha = reinterpret_cast<int*>(reinterpret_cast<float**>(P->h_dyn[s]) + 2 * ng);
    for (int i = 0; i < P->n; ++i) {
        const int gi = P->route[i];
        if (gi < 0) continue;
        if (!x[i] || !cov[i]) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_run: layer %d: null input or covariance", i);
        if (P->is_corr[gi] && !aligned16(cov[i])) return fail(NSGP_ERR_INVALID, "nsgp_cov_plan_run: layer %d: the covariance must be 16-byte aligned", i);
        hx[gi] = x[i];
        hc[gi] = cov[i];
        ha[gi] = accumulate[i];
    }
    for (int gi = 0; gi < ng; ++gi) {                            // the kernels address a strip through its parent's slot; keep its own slot valid
        const int parent = P->ring_parent[gi];
        if (parent >= 0) { hx[gi] = hx[parent]; hc[gi] = hc[parent]; ha[gi] = 1; }
    }
    NSGP_HIP(hipMemcpyAsync(P->d_dyn[s], P->h_dyn[s], P->dyn_bytes, hipMemcpyHostToDevice, stream));
    char* ws = static_cast<char*>(workspace);
    unsigned* amax = reinterpret_cast<unsigned*>(ws + P->amax_off);
    NSGP_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned) * ng, stream));
    const int* pm = P->d_prefix, *pa = pm + (ng + 1), *ps = pa + (ng + 1), *pasm = ps + (ng + 1), *corr_ids = pasm + (P->n_corr + 1);
    if (P->mean_units > 0) {
        hipLaunchKernelGGL(nsgp_cov_group_mean_kernel, dim3(P->mean_units), dim3(256), 0, stream, P->d_layers, pm, ng, (const void*)P->d_dyn[s], ws);
        NSGP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nsgp_cov_group_amax_kernel, dim3(P->amax_units), dim3(256), 0, stream, P->d_layers, pa, ng, (const void*)P->d_dyn[s], (const char*)ws, amax);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nsgp_cov_group_split_kernel, dim3(P->split_units), dim3(256), 0, stream, P->d_layers, ps, ng, (const void*)P->d_dyn[s], ws, (const unsigned*)amax);
    NSGP_LAUNCH_CHECK();
    float* slabs = reinterpret_cast<float*>(ws + P->slab_off);
    hipLaunchKernelGGL(nsgp_cov_group_syrk_kernel, dim3(P->n_tiles), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, P->d_tiles, P->d_layers, ng,
                       (const void*)P->d_dyn[s], (const char*)ws, (const unsigned*)amax, slabs);
    NSGP_LAUNCH_CHECK();
    for (const nsgp_cov_plan::SplitLayer& sl : P->split_layers) {      // the long im2col contractions: ordered sum of their K ranges
        const int nb128 = sl.Dp / 128;
        const long t128 = (long)nb128 * (nb128 + 1) / 2;
        int bands = 2;
        while (bands < 32 && t128 * bands < 256) bands *= 2;
        hipLaunchKernelGGL(nsgp_cov_reduce_v2_kernel, dim3((unsigned)t128, bands), dim3(256), 0, stream, slabs + sl.slab_base * (size_t)(256 * 128), sl.D, sl.Dp, sl.S,
                           128 / bands, hc[sl.group_index], ha[sl.group_index], (const unsigned*)(amax + sl.group_index));
        NSGP_LAUNCH_CHECK();
    }
    if (P->n_corr > 0) {                                         // correlation form: R from the slabs, then C = R laid out over the tap pairs - the strips
        hipLaunchKernelGGL(nsgp_cov_corr_reduce_kernel, dim3(P->corr_units, 8), dim3(256), 0, stream, P->d_corr, P->d_layers, (const float*)slabs, ws);
        NSGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(nsgp_cov_corr_assemble_kernel, dim3(P->asm_units), dim3(256), 0, stream, corr_ids, pasm, P->n_corr, P->d_layers, ng,
                           (const void*)P->d_dyn[s], (const char*)ws, (const unsigned*)amax);
        NSGP_LAUNCH_CHECK();
    }
    NSGP_HIP(hipEventRecord(P->ev[s], stream));
    P->ev_used[s] = true;
    return NSGP_OK;
}

extern "C" int nsgp_cov_set_split_mfma(int mode) {
    // 0: fp32 MFMA; 1: auto (default); 2: always the fp16 split; 3: always the split, first-generation (gather) kernel only
    const int prev = g_cov_gen2 ? g_cov_split : 3;
    g_cov_gen2 = mode != 3;
    g_cov_split = mode < 0 ? 0 : (mode >= 2 ? 2 : mode);
    return prev;
}

extern "C" int nsgp_cov_accumulate_linear(const float* x, int batch, int features, float* cov, int accumulate, void* stream_) {
    if (!x || !cov || batch <= 0 || features <= 0) return fail(NSGP_ERR_INVALID, "nsgp_cov_accumulate_linear: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long n = (long)features * features;
    hipLaunchKernelGGL(nsgp_cov_linear_kernel, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, stream, x, batch, features, cov, accumulate);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
