// Ablation micro-benchmark for the fp32 MFMA tile core (timing only; variants >0 compute garbage).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nsgp-repre_amd/csrc -o gpurun_out/gemm_bench tools/gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cmath>
#include "gemm_core.hpp"
using namespace nsgp;

namespace nsgp {
// EXPERIMENT (measured and rejected): results bit-identical to the production tile, 121 TF vs 137 TF.
// ---- LDS-DMA variant of the dense tile (global_load_lds_dwordx4: no VGPR staging, no ds_write) ------------------
// One wave-instruction moves 64 x 16 B from per-lane global addresses to 1 KiB of CONTIGUOUS LDS (base in M0,
// lane l lands at base + 16 l), so the image layout is fixed by the hardware and the freedom is in WHICH element
// each lane fetches:
//   A ("rows" operand): piece = 8 rows x 128 B (one full line per row -> coalesced); lane (row = l>>3, slot = l&7)
//     fetches k-quad (slot ^ f(row)), f(row) = (row>>1)&7.  Reading quad Q of row R = chunk (Q ^ f(R)) of the row:
//     a ds_read_b128 lane group (rows distinct mod 16, same Q) then covers all 16 slots of the 256-B bank row.
//   B (KN operand, [k][n] image): piece = 2 k-rows x 512 B, lane-linear = the layout mfma_kstep<false> reads.
// Two LDS buffers, loads of K-step t+1 issued at the top of K-step t (the barrier that ended t-1 freed the buffer),
// drained by the vmcnt(0) the compiler places before the barrier that ends K-step t.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;
constexpr int DMA_IMG = BM * BK;                       // 4096 floats = 16 KiB per operand image
static_assert(4 * DMA_IMG <= SMEM_FLOATS, "DMA images must fit the K-loop LDS");

__device__ __forceinline__ int dma_f(int row) { return (row >> 1) & 7; }

// The instruction is written as inline asm on purpose: through the builtin the compiler's wait-count pass sees a
// VMEM operation that writes LDS and places `s_waitcnt vmcnt(0)` before the NEXT ds_read of any LDS address --
// i.e. it waits for the prefetch of K-step t+1 before the first MFMA of K-step t.  The asm form is invisible to that
// pass; the one wait that is needed (before the barrier that publishes the buffer) is dma_wait() below.
__device__ __forceinline__ void dma_piece(const float* src, float* lds_dst) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"((gl_void_t*)src) : "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void dma_issue(const float* const (&pa)[4], const float* const (&pb)[4], long a_off, long b_off,
                                          float* a_img_, float* b_img_) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dma_piece(pa[i] + a_off, a_img_ + (wave * 4 + i) * 256);
        dma_piece(pb[i] + b_off, b_img_ + (wave * 4 + i) * 256);
    }
}

__device__ __forceinline__ void mfma_kstep_dma(const float* __restrict__ As, const float* __restrict__ Bs,
                                               f32x16 (&acc)[2][2], int wm, int wn) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int R = wm * 64 + r;
    const float* a_row = As + (R >> 3) * 256 + (R & 7) * 32;       // +1024 floats for row R + 32 (same f)
    const int f = dma_f(R);
    const float* b_kn = Bs + (4 * h) * BN + wn * 64 + r;
#pragma unroll
    for (int q = 0; q < BK / 4; q += 2) {
        const int x = ((q + h) ^ f) * 4;
        const float4 a0 = *reinterpret_cast<const float4*>(a_row + x);
        const float4 a1 = *reinterpret_cast<const float4*>(a_row + 1024 + x);
        float b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b0[j] = b_kn[(4 * q + j) * BN];
            b1[j] = b_kn[(4 * q + j) * BN + 32];
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

// acc = A[m0.., :] x B[:, n0..]; A [M x K] row-major, B [K x N] row-major; whole tiles, K % BK == 0, 16-byte aligned rows.
__device__ __forceinline__ void gemm_tile_dma(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                              int K, int m0, int n0, float* smem, f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float *pa[4], *pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        pa[i] = A + (long)(m0 + row) * lda + (((lane & 7) ^ dma_f(row)) * 4);
        pb[i] = B + (long)((wave * 4 + i) * 2 + (lane >> 5)) * ldb + n0 + (lane & 31) * 4;
    }
    const int nk = K / BK;
    // buffer indices are compile-time (two K-steps per trip): a run-time select between LDS pointers makes the
    // compiler fall back to flat loads, which also tick vmcnt and would re-serialise the loop on the prefetch
    auto step = [&](int t, auto rb) {
        constexpr int RB = decltype(rb)::value;
        if (t + 1 < nk)
            dma_issue(pa, pb, (long)(t + 1) * BK, (long)(t + 1) * BK * ldb, smem + (1 - RB) * DMA_IMG, smem + (3 - RB) * DMA_IMG);
        mfma_kstep_dma(smem + RB * DMA_IMG, smem + (2 + RB) * DMA_IMG, acc, wm, wn);
        dma_wait();
        __syncthreads();
    };
    dma_issue(pa, pb, 0, 0, smem, smem + 2 * DMA_IMG);
    dma_wait();
    __syncthreads();
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        step(t, IC<0>{});
        step(t + 1, IC<1>{});
    }
    if (t < nk) step(t, IC<0>{});
}

}  // namespace nsgp

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// VARIANT 6: the LDS-DMA tile above.
// VARIANT 0: production gemm_tile.  1: no barriers in the loop.  2: no global loads in the loop.
// 3: MFMA + LDS reads only (no staging, no barrier).  4: MFMA only, operands in registers.
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void bench_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[2][2];
    zero_acc(acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    if (VARIANT == 0) {
        gemm_tile<true, true, false>(A, K, B, N, M, N, K, m0, n0, smem, acc);
    } else if (VARIANT == 5) {
        gemm_tile<true, true, false>(A, K, B, N, M, N, K, m0, n0, smem, acc);
    } else if (VARIANT == 6) {
        gemm_tile_dma(A, K, B, N, K, m0, n0, smem, acc);
    } else {
        const int nk = K / BK;
        float ra[4][4], rb[4][4];
        stage_rows<true>(A, K, M, K, m0, 0, ra);
        stage_kn<true>(B, N, K, N, 0, n0, rb);
        write_rows(a_img(smem, 0), ra); write_kn(b_img(smem, 0), rb);
        write_rows(a_img(smem, 1), ra); write_kn(b_img(smem, 1), rb);
        __syncthreads();
        if (VARIANT == 4) {
            float a0 = A[lane], b0 = B[lane];
            for (int t = 0; t < nk; ++t) {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][1], 0, 0, 0);
                }
            }
        } else {
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                if (VARIANT == 1) {
                    stage_rows<true>(A, K, M, K, m0, ((t + 1) % nk) * BK, ra);
                    stage_kn<true>(B, N, K, N, ((t + 1) % nk) * BK, n0, rb);
                }
                mfma_kstep<false>(a_img(smem, cur), b_img(smem, cur), acc, wm, wn);
                if (VARIANT == 1) { write_rows(a_img(smem, cur ^ 1), ra); write_kn(b_img(smem, cur ^ 1), rb); }
                if (VARIANT == 2) __syncthreads();
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                float* dst = &C[(long)(m0 + wm * 64 + mi * 32 + acc_row(r, lane)) * N + n0 + wn * 64 + ni * 32 + (lane & 31)];
                *dst = (VARIANT == 5) ? (*dst + -0.02f * acc[mi][ni][r]) : acc[mi][ni][r];
            }
}

static float* g_flush = nullptr;   // when set: stream 1 GiB through the caches before every timed launch
template <int V>
static float run(const float* A, const float* B, float* C, int M, int N, int K, int extra_lds, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES + extra_lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(N / BN, M / BM);
    std::vector<float> ts;
    for (int i = 0; i < reps + 2; ++i) {
        if (g_flush) hipMemsetAsync(g_flush, i, (size_t)1 << 30, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench_kernel<V>, grid, dim3(256), SMEM_BYTES + extra_lds, 0, A, B, C, M, N, K);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = N;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)K * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    std::vector<float> h((size_t)std::max(M, K) * std::max(K, N));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * (double)K;
    if (argc > 3 && atoi(argv[3])) CK(hipMalloc(&g_flush, (size_t)1 << 30));
    const char* names[] = {"production", "no-barrier", "no-global-loads", "mfma+ldsread", "mfma-only"};
    for (int round = 0; round < 2; ++round) {
        float t[5];
        t[0] = run<0>(A, B, C, M, N, K, 0, 8); t[1] = run<1>(A, B, C, M, N, K, 0, 8); t[2] = run<2>(A, B, C, M, N, K, 0, 8);
        t[3] = run<3>(A, B, C, M, N, K, 0, 8); t[4] = run<4>(A, B, C, M, N, K, 0, 8);
        float t5 = run<5>(A, B, C, M, N, K, 0, 8);
        {   // LDS-DMA tile: correctness against the production tile, then timing
            std::vector<float> c0((size_t)M * N), c6((size_t)M * N);
            run<0>(A, B, C, M, N, K, 0, 1); hipMemcpy(c0.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
            hipMemset(C, 0, (size_t)M * N * 4);
            float t6 = run<6>(A, B, C, M, N, K, 0, 8); hipMemcpy(c6.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
            double md = 0, mx = 0;
            for (size_t i = 0; i < c0.size(); ++i) { md = std::max(md, (double)fabsf(c0[i] - c6[i])); mx = std::max(mx, (double)fabsf(c0[i])); }
            printf("round %d  %-16s 2wg/cu  %.3f ms  %.1f TF   max|diff| vs production %.3g (max|C| %.3g)\n", round, "lds-dma", t6, fl / t6 / 1e9, md, mx);
        }
        printf("round %d  %-16s 2wg/cu  %.3f ms  %.1f TF\n", round, "prod+scale+rmw", t5, fl / t5 / 1e9);
        for (int v = 0; v < 5; ++v) printf("round %d  %-16s 2wg/cu  %.3f ms  %.1f TF\n", round, names[v], t[v], fl / t[v] / 1e9);
        float s0 = run<0>(A, B, C, M, N, K, 40000, 8), s4 = run<4>(A, B, C, M, N, K, 40000, 8), s3 = run<3>(A, B, C, M, N, K, 40000, 8);
        printf("round %d  production 1wg/cu %.3f ms %.1f TF | mfma+ldsread 1wg/cu %.1f TF | mfma-only 1wg/cu %.1f TF\n", round, s0, fl / s0 / 1e9, fl / s3 / 1e9, fl / s4 / 1e9);
    }
    return 0;
}
