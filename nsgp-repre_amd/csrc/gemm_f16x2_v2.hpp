// fp32-accurate projection GEMM on the fp16 matrix cores, second generation: BOTH operands arrive pre-split and
// pre-tiled, and travel global -> LDS by LDS-DMA (global_load_lds_dwordx4) issued by dedicated LOADER waves; the consumer
// waves' main loop holds MFMAs and LDS reads only.
//
// Arithmetic (as gemm_f16x2.hpp): x = x0 + x1, x0 = fp16(s x), x1 = fp16(s x - x0), a*b ~= a1 b0 + a0 b1 + a0 b0 -- three
// v_mfma_f32_32x32x16_f16 per fp32-equivalent product, fp32 accumulation, smallest terms first.  What changed is the scale s:
// one power of two PER ROW of the left operand (the update: found by the elementwise launch that produces the row) and one PER
// COLUMN of the right operand (the projector: found once per task), each putting its row's / column's largest magnitude
// into [2^13, 2^14).  Every output element C[m][n] is therefore computed with ~22 significant bits relative to
// max_k|A[m][k]| * max_k|P[k][n]| -- per-row accuracy like the reference's fp32 torch.mm (SGD_NSCL.py:85-90), not relative to
// the largest entry of the whole tensor.  The scales are undone exactly in the epilogue (rinv[m] * cinv[n]).
//
// Operand storage ("pre-tiled", the SAME for both operands; `row` = m for the update, n for the projector's transpose):
//     [row / 64][k / 8][term 0|1][row % 64][8 halves]           4 bytes per element
// i.e. one (64-row block, k-octet, term) PLANE is 1 KiB of contiguous memory holding the 16-byte MFMA operand piece of each
// of its 64 rows.  One LDS-DMA wave-instruction moves exactly one plane (64 lanes x 16 B, lane-linear on both sides: fully
// coalesced in memory, no swizzle needed), and a ds_read_b128 of 32 lanes along the rows of a plane is bank-conflict free
// (the hardware's 16-lane groups cover rows that are distinct mod 16).  Per 64-row block a k32 step is 8 planes = 8 KiB
// contiguous.
//
// Tile: 256 x 128 (MB = 4 row blocks x 2 column blocks), ONE workgroup of 768 threads per CU: waves 0-7 are consumers (4 (M) x
// 2 (N), 64 x 64 per wave as 2 x 2 MFMA blocks), waves 8-11 loaders.  The projector -- the operand that comes from HBM, 4 B per
// element -- is shared by twice the rows of the first-generation 128 x 128 tile: 32 B/clk/CU from L2 at full MFMA rate
// instead of 43.  LDS: three 48 KiB stages (144 of the CU's 160 KiB).  K-step t: each loader waits for ITS OWN pieces of
// stage t (counted vmcnt that leaves step t+1's in flight), all twelve waves meet at ONE raw s_barrier (everybody's pieces of t
// have landed AND every consumer is done reading stage t-1); the loaders issue the pieces of step t+2 into the stage t-1
// used, the consumers do 16 ds_read_b128 + 24 MFMAs.  Prefetch distance two full steps (~2.4 us).  The consumers' two k16 halves
// are software-pipelined ACROSS the barrier (PIPE, see the loop): bit-identical results, +0..4 % (proj_v2_pipe_study.log).
// MB = 2 serves the 128-row layers (and a trailing 128-row remainder): the same code with consumer waves 4-7 idle.
//
// Measured and not kept -- the variants live in tools/proj_v2_bench.hip, their logs in profiles/r02/proj_v2_*_study.log:
//   * every wave issuing its own six pieces right after the barrier (512 threads, no loaders): the first form of this tile,
//     418 against 426-434 TF-eq on full rounds of a uniform GEMM, 0.340 against 0.320 ms on the R-50 table;
//   * staggering that DMA issue between the two waves of a SIMD (-5 %); other LDS-read / MFMA interleaves (+-2 %);
//     the v_mfma_f32_16x16x32_f16 form of the step (same rate);
//   * 128 x 128 tiles of 256 threads, two independent workgroups per CU (bit-identical; up to +40 % on launches that
//     under-fill the chip, -5 % on full rounds, +-1 % on the tables);
//   * an L2 prefetch of the projector stream by 4-byte "touch" loads 3-12 steps ahead: with every launch streaming its
//     projector from HBM (700 MB of copies in rotation) the kernel already runs at the cache-resident rate; touches cost 2.5 %.
#pragma once
#include "gemm_core.hpp"
#include "gemm_f16x2.hpp"

namespace nsgp {

constexpr int V2_BLOCK_ROWS = 64;
constexpr int V2_PLANE = 1024;                          // bytes of one (block, octet, term) plane
constexpr int V2_BK = 32;
constexpr int V2_STEP = 8 * V2_PLANE;                   // bytes per 64-row block per k32 step (4 octets x 2 terms)
constexpr int V2_NB = 2;                                // column blocks of a tile (128 columns)
constexpr int V2_MB_MAX = 4;                            // row blocks of a full tile (256 rows)
constexpr int V2_STAGE = (V2_MB_MAX + V2_NB) * V2_STEP; // 48 KiB
constexpr int V2_STAGES = 3;
constexpr int V2_RING_BYTES = V2_STAGES * V2_STAGE;     // 147,456 B
constexpr int V2_SMEM_BYTES = V2_RING_BYTES;
static_assert(8 * 64 * EPI_LD * 4 <= V2_RING_BYTES, "the epilogue re-layout (16 KiB per wave) must fit the ring");

typedef __attribute__((address_space(3))) char lds_char;

// bytes from the start of a pre-tiled operand to the plane (block, octet, term) for an operand with K columns
__host__ __device__ __forceinline__ size_t v2_plane_offset(int block, int octet, int term, int K) {
    return (((size_t)block * (K / 8) + octet) * 2 + term) * V2_PLANE;
}
__host__ __device__ __forceinline__ size_t v2_operand_bytes(int rows, int K) { return (size_t)rows * K * 4; }

// one LDS-DMA piece: the 1 KiB at `src` + lane * 16 -> LDS byte address `lds` + lane * 16
__device__ __forceinline__ void v2_dma_one(unsigned long long src, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds), "s"(src));
}

// raw barrier (a __syncthreads() would drain the LDS-DMA prefetch with vmcnt(0)); the empty asms keep the compiler from
// moving LDS accesses across it
__device__ __forceinline__ void v2_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned long long v2_uniform(unsigned long long x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ---- the tile: acc += A[rows of blocks a_block0 .. +MB) x B[cols of blocks b_block0, +1], both pre-tiled / pre-split, K % 32 == 0
// 768 threads: waves 0-7 are CONSUMERS (4 x 2, 64 x 64 each: LDS reads + MFMAs, nothing else), waves 8-11 are LOADERS that
// issue all 48 (MB = 2: 32) DMA pieces of a step and never touch the matrix pipe.  (When every wave loads, each spends the
// head of each step issuing its six LDS-DMA pieces -- ~100-180 cycles apiece while the CU's memory pipeline is taking 48 KiB --
// with BOTH waves of a SIMD stuck there together and the matrix pipe idle; here the matrix work starts right behind the
// barrier.)  One barrier per step for all twelve waves.  On return every wave has passed a barrier after the last LDS read and
// no DMA is in flight: the ring is the caller's (only waves 0 .. 2 MB - 1 hold accumulators).
//   loader l (0..3): planes 2l, 2l+1 (octet l, both terms) of every block slot -> 12 (8) pieces per step, counted vmcnt(12 / 8).
constexpr int V2L_THREADS = 768;

template <int MB, bool PIPE = true>
__device__ __forceinline__ void gemm_tile_f16x2_v2l(const void* __restrict__ Asplit, int a_block0, const void* __restrict__ Bsplit,
                                                    int b_block0, int K, char* smem, f32x16 (&acc)[2][2], int step0 = 0, int nsteps = -1) {
    // K = the operands' full contraction length (it sets the block stride); the tile contracts k32 steps [step0, step0 + nsteps)
    // (default: all of them) -- split-K callers give each workgroup a range.
    static_assert(MB == 2 || MB == 4, "MB");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t blk = (size_t)K * 256;
    const int nk = nsteps < 0 ? K / V2_BK : nsteps;
    if (wave >= 8) {
        // ------------------------------------------------------------------ loader
        constexpr int NPIECE = 2 * (MB + V2_NB);
        const int l = wave - 8;
        unsigned long long src[V2_MB_MAX + V2_NB];
#pragma unroll
        for (int b = 0; b < V2_MB_MAX; ++b)
            src[b] = v2_uniform((unsigned long long)(uintptr_t)Asplit + (size_t)(a_block0 + (b < MB ? b : 0)) * blk + (size_t)step0 * V2_STEP + (size_t)l * (2 * V2_PLANE));
#pragma unroll
        for (int b = 0; b < V2_NB; ++b)
            src[V2_MB_MAX + b] = v2_uniform((unsigned long long)(uintptr_t)Bsplit + (size_t)(b_block0 + b) * blk + (size_t)step0 * V2_STEP + (size_t)l * (2 * V2_PLANE));
        const unsigned voff = lane * 16;
        const unsigned my_planes = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem + l * (2 * V2_PLANE));
        auto issue = [&](int stage) {
#pragma unroll
            for (int b = 0; b < V2_MB_MAX + V2_NB; ++b) {
                if (MB == 2 && (b == 2 || b == 3)) continue;
                v2_dma_one(src[b], voff, my_planes + stage * V2_STAGE + b * V2_STEP);
                v2_dma_one(src[b] + V2_PLANE, voff, my_planes + stage * V2_STAGE + b * V2_STEP + V2_PLANE);
                src[b] += V2_STEP;
            }
        };
        auto lstep = [&](int t, auto st_next2) {
            if (t + 1 < nk) { if constexpr (NPIECE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v2_barrier();
            if (t + 2 < nk) issue(decltype(st_next2)::value);
        };
        issue(0);
        if (nk > 1) issue(1);
        int t = 0;
        for (; t + 2 < nk; t += 3) { lstep(t, IC<2>{}); lstep(t + 1, IC<0>{}); lstep(t + 2, IC<1>{}); }
        if (t < nk) { lstep(t, IC<2>{}); ++t; }
        if (t < nk) { lstep(t, IC<0>{}); ++t; }
        v2_barrier();
        return;
    }
    // ---------------------------------------------------------------------- consumer
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const lds_char* abase = (const lds_char*)smem + wm * V2_STEP + h * (2 * V2_PLANE) + r * 16;
    const lds_char* bbase = (const lds_char*)smem + (V2_MB_MAX + wn) * V2_STEP + h * (2 * V2_PLANE) + r * 16;
    const bool active = wm < MB;
    typedef const __attribute__((address_space(3))) h16x8* lds_frag;
    auto compute = [&](auto st) {
        constexpr int ST = decltype(st)::value;
        if (!active) return;
        h16x8 fa[2][2][2], fb[2][2][2];     // [k16 half][32-row block][term]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[ks][i][p] = *reinterpret_cast<lds_frag>(abase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + i * 512);
                    fb[ks][i][p] = *reinterpret_cast<lds_frag>(bbase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + i * 512);
                }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][mi][1], fb[ks][ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][mi][0], fb[ks][ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][mi][0], fb[ks][ni][0], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    };
    if constexpr (!PIPE) {
        int t = 0;
        for (; t + 2 < nk; t += 3) {
            v2_barrier(); compute(IC<0>{});
            v2_barrier(); compute(IC<1>{});
            v2_barrier(); compute(IC<2>{});
        }
        if (t < nk) { v2_barrier(); compute(IC<0>{}); ++t; }
        if (t < nk) { v2_barrier(); compute(IC<1>{}); ++t; }
        v2_barrier();
        return;
    }
    // ---- PIPE: the k16 halves of a step are software-pipelined ACROSS the step barrier.  In the plain schedule above every
    // consumer wave leaves the barrier, issues its first 8 fragment reads and stalls until they land -- all eight waves at once,
    // 64 KiB through a 128 B/clk LDS, with both waves of every SIMD waiting together and the matrix pipe idle.  Here the second
    // half's fragments of stage s-1 stay in registers over the barrier and their 12 MFMAs run while the first-half reads of
    // stage s are in flight; the second-half reads of stage s then fly under the first half's MFMAs.  Same 16 fragment registers.
    // The reads of a stage must have LANDED before the barrier that hands the stage back to the loaders: lgkmcnt(0) in front of it.
    h16x8 f0a[2][2], f0b[2][2], f1a[2][2], f1b[2][2];      // [32-row block][term] of k16 half 0 / half 1
    auto read_half = [&](auto st, auto ks_, h16x8 (&fa)[2][2], h16x8 (&fb)[2][2]) {
        constexpr int ST = decltype(st)::value, ks = decltype(ks_)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][p] = *reinterpret_cast<lds_frag>(abase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + i * 512);
                fb[i][p] = *reinterpret_cast<lds_frag>(bbase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + i * 512);
            }
    };
    auto mfma_half = [&](const h16x8 (&fa)[2][2], const h16x8 (&fb)[2][2]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
    };
    auto pin_half = [&]() {      // 8 x (one LDS read, one MFMA), then the remaining 4 MFMAs
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };
    auto first = [&](auto st) {            // stage 0 of the range: nothing to overlap the first reads with
        v2_barrier();
        if (!active) return;
        read_half(st, IC<0>{}, f0a, f0b);
        read_half(st, IC<1>{}, f1a, f1b);
        mfma_half(f0a, f0b);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    };
    auto steady = [&](auto st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // half 1 of the previous stage is in registers: its LDS may be refilled
        v2_barrier();
        if (!active) return;
        read_half(st, IC<0>{}, f0a, f0b);
        mfma_half(f1a, f1b);                                    // previous stage, half 1
        pin_half();
        read_half(st, IC<1>{}, f1a, f1b);
        mfma_half(f0a, f0b);
        pin_half();
    };
    first(IC<0>{});
    int t = 1;
    for (; t + 2 < nk; t += 3) { steady(IC<1>{}); steady(IC<2>{}); steady(IC<0>{}); }
    if (t < nk) { steady(IC<1>{}); ++t; }
    if (t < nk) { steady(IC<2>{}); ++t; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    v2_barrier();
    if (active) mfma_half(f1a, f1b);
}

// ---- operand preparation ------------------------------------------------------------------------------------------------
// One 16-byte piece pair: 8 consecutive k of one row, scaled, split and stored to its two planes.
template <bool NT = false>
__device__ __forceinline__ void v2_store_pieces(void* __restrict__ dst, int row, int octet, int K, const f32x4 lo, const f32x4 hi, float scale) {
    h16x8 p0, p1;
    f2_split(lo, hi, scale, p0, p1);
    char* base = static_cast<char*>(dst) + v2_plane_offset(row >> 6, octet, 0, K) + (row & 63) * 16;
    if constexpr (NT) {      // a stream written once and read back from HBM by a later launch: keep it out of the way of what L2 holds
        __builtin_nontemporal_store(p0, (g_h16x8*)(base));
        __builtin_nontemporal_store(p1, (g_h16x8*)(base + V2_PLANE));
    } else {
        *(g_h16x8*)(base) = p0;
        *(g_h16x8*)(base + V2_PLANE) = p1;
    }
}

// Stand-alone row split of a [rows x K] fp32 matrix (rows % 8 == 0, K % 8 == 0): one workgroup per band of 8 rows finds
// the 8 row maxima, publishes rinv[row] = 1 / scale, and writes the pre-tiled split.  (Inside the optimizer step this work
// is fused into the elementwise launch, projected_step.hip; this kernel serves nsgp_project_f16x2 and the benches.)
static __global__ __launch_bounds__(256) void nsgp_split_rows_f16x2_kernel(const float* __restrict__ A, int rows, int K,
                                                                          void* __restrict__ out, float* __restrict__ rinv) {
    __shared__ unsigned rowmax[8];
    __shared__ float rowscale[8];
    const int row0 = blockIdx.x * 8, tid = threadIdx.x;
    if (tid < 8) rowmax[tid] = 0u;
    __syncthreads();
    const float* band = A + (size_t)row0 * K;
    for (int r = 0; r < 8; ++r) {
        float am = 0.0f;
        for (int k = tid * 4; k < K; k += 1024) {
            const f32x4 v = *(const gf32x4*)(band + (size_t)r * K + k);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
        if ((tid & 63) == 0) atomicMax(&rowmax[r], __float_as_uint(am));
    }
    __syncthreads();
    if (tid < 8) {
        const float s = f2_scale_from_amax_bits(rowmax[tid]);
        rowscale[tid] = s;
        rinv[row0 + tid] = 1.0f / s;
    }
    __syncthreads();
    const int items = K;                // 8 rows x K / 8 octets
    for (int id = tid; id < items; id += 256) {
        const int r = id & 7, o = id >> 3;
        const float* src = band + (size_t)r * K + o * 8;
        v2_store_pieces(out, row0 + r, o, K, *(const gf32x4*)src, *(const gf32x4*)(src + 4), rowscale[r]);
    }
}

// Column maxima of P [K x N] fp32 row-major -> the power-of-two scale of each column and its inverse.
static __global__ __launch_bounds__(256) void nsgp_col_scales_f16x2_kernel(const float* __restrict__ P, int K, int N,
                                                                          float* __restrict__ cscale, float* __restrict__ cinv) {
    __shared__ float part[8][33];
    const int n = blockIdx.x * 32 + (threadIdx.x & 31), ty = threadIdx.x >> 5;
    float am = 0.0f;
    if (n < N)
        for (int k = ty; k < K; k += 8) am = fmaxf(am, fabsf(P[(size_t)k * N + n]));
    part[ty][threadIdx.x & 31] = am;
    __syncthreads();
    if (ty == 0 && n < N) {
        for (int j = 1; j < 8; ++j) am = fmaxf(am, part[j][threadIdx.x & 31]);
        const float s = f2_scale_from_amax_bits(__float_as_uint(am));
        cscale[n] = s;
        cinv[n] = 1.0f / s;
    }
}

// P [K x N] fp32 row-major -> pre-tiled two-term split of diag(cscale) * P^T ("row" = n, contraction index k).
static __global__ __launch_bounds__(256) void nsgp_split_transpose_f16x2_v2_kernel(const float* __restrict__ P, int K, int N,
                                                                                 const float* __restrict__ cscale, void* __restrict__ out) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        tile[i][tx] = (k0 + i < K && n0 + tx < N) ? P[(size_t)(k0 + i) * N + n0 + tx] : 0.0f;
    __syncthreads();
    // 32 n x 4 octets = 128 piece pairs, one per thread of the first two waves
    if (threadIdx.x < 128) {
        const int n = n0 + (threadIdx.x & 31), o = threadIdx.x >> 5;
        if (n < N && k0 + o * 8 < K) {
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = tile[o * 8 + e][threadIdx.x & 31]; hi[e] = tile[o * 8 + 4 + e][threadIdx.x & 31]; }
            v2_store_pieces(out, n, (k0 >> 3) + o, K, lo, hi, cscale[n]);
        }
    }
}

}  // namespace nsgp
