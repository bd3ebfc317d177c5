// Ablation micro-benchmark for the fp32 MFMA tile core (timing only; variants >0 compute garbage).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nsgp-repre_amd/csrc -o gpurun_out/gemm_bench tools/gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "gemm_core.hpp"
using namespace nsgp;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

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
        printf("round %d  %-16s 2wg/cu  %.3f ms  %.1f TF\n", round, "prod+scale+rmw", t5, fl / t5 / 1e9);
        for (int v = 0; v < 5; ++v) printf("round %d  %-16s 2wg/cu  %.3f ms  %.1f TF\n", round, names[v], t[v], fl / t[v] / 1e9);
        float s0 = run<0>(A, B, C, M, N, K, 40000, 8), s4 = run<4>(A, B, C, M, N, K, 40000, 8), s3 = run<3>(A, B, C, M, N, K, 40000, 8);
        printf("round %d  production 1wg/cu %.3f ms %.1f TF | mfma+ldsread 1wg/cu %.1f TF | mfma-only 1wg/cu %.1f TF\n", round, s0, fl / s0 / 1e9, fl / s3 / 1e9, fl / s4 / 1e9);
    }
    return 0;
}
