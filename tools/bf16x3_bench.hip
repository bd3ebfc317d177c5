// Ceiling experiment for a 3-way bf16 split of the fp32 projection GEMM (a = a0 + a1 + a2 in bf16, six
// v_mfma_f32_32x32x16_bf16 per fp32-equivalent product: a0b0 a0b1 a1b0 a0b2 a1b1 a2b0).  Timing only: operand
// images are filled once, the loop is MFMA + LDS reads (the analogue of gemm_bench variant 3 for fp32).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bf16x3 tools/bf16x3_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cmath>

namespace ceiling {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BK = 32;                      // K-step = 32 k = two k16 MFMA steps
constexpr int OCT = BM * 8 + 8;                       // one k-octet plane: [row][8 bf16], padded by 16 B (in bf16 units)
constexpr int PLANE_IMG = (BK / 8) * OCT;             // one split plane of one operand per K-step
constexpr int STAGE = 6 * PLANE_IMG;                  // A0 A1 A2 B0 B1 B2

template <int MODE>   // 0: MFMA + LDS reads   1: MFMA only
__global__ __launch_bounds__(256, 2) void k(const __bf16* __restrict__ src, float* __restrict__ C, int N, int nk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    for (int i = threadIdx.x; i < STAGE; i += 256) smem[i] = src[i % 4096];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[2][3], fb[2][3];
    for (int t = 0; t < nk; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {              // k16 step: lane half h reads octet 2*ks + h
            if (MODE == 0 || t == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        fa[mi][p] = *reinterpret_cast<const bf16x8*>(smem + p * PLANE_IMG + (2 * ks + h) * OCT + (wm * 64 + mi * 32 + r) * 8);
                        fb[mi][p] = *reinterpret_cast<const bf16x8*>(smem + (3 + p) * PLANE_IMG + (2 * ks + h) * OCT + (wn * 64 + mi * 32 + r) * 8);
                    }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][2], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][1], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][2], fb[ni][0], acc[mi][ni], 0, 0, 0);
                }
        }
    }
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e)
        C[(long)(m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * N + n0 + wn * 64 + ni * 32 + r] = acc[mi][ni][e];
}

template <int MODE>
static float run(const __bf16* src, float* C, int M, int N, int K) {
    using namespace ceiling;
    const int lds = STAGE * 2;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 8; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(N / 128, M / 128), dim3(256), lds, 0, src, C, N, K / BK);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

}  // namespace ceiling

// ---- full data path: gemm_tile_bf16x3 against the production fp32 tile ------------------------------------------------
#include "gemm_bf16x3.hpp"
using namespace nsgp;

__device__ __forceinline__ void store_plain(float* C, int N, int m0, int n0, const f32x16 (&acc)[2][2]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e)
        C[(long)(m0 + wm * 64 + mi * 32 + acc_row(e, lane)) * N + n0 + wn * 64 + ni * 32 + (lane & 31)] = acc[mi][ni][e];
}

__global__ __launch_bounds__(256, 2) void x3_kernel(const float* __restrict__ A, const __bf16* __restrict__ Bt, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    f32x16 acc[2][2];
    zero_acc(acc);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    gemm_tile_bf16x3(A, K, Bt, K, m0, n0, sm, acc);
    store_plain(C, N, m0, n0, acc);
}
__global__ __launch_bounds__(256, 2) void f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    f32x16 acc[2][2];
    zero_acc(acc);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    gemm_tile<true, true, false>(A, K, B, N, M, N, K, m0, n0, sm, acc);
    store_plain(C, N, m0, n0, acc);
}

template <class F>
static float time_it(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 8; ++i) {
        (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static int full_path(int M, int N, int K) {
    float *A, *B, *C0, *C1; __bf16* Bt;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)K * N * 4); (void)hipMalloc(&C0, (size_t)M * N * 4); (void)hipMalloc(&C1, (size_t)M * N * 4);
    (void)hipMalloc(&Bt, (size_t)3 * N * K * 2);
    std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * 1e-3f;
    for (auto& v : hb) v = rnd() * 0.05f;
    (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X3_SMEM_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    const float t_split = time_it([&] { hipLaunchKernelGGL(nsgp_split_transpose_bf16x3_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, 0, B, K, N, Bt); });
    const float t0 = time_it([&] { hipLaunchKernelGGL(f32_kernel, dim3(N / BN, M / BM), dim3(256), SMEM_BYTES, 0, A, B, C0, M, N, K); });
    const float t1 = time_it([&] { hipLaunchKernelGGL(x3_kernel, dim3(N / BN, M / BM), dim3(256), X3_SMEM_BYTES, 0, A, Bt, C1, M, N, K); });
    std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
    (void)hipMemcpy(c0.data(), C0, c0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost);
    // fp64 reference on a sample of entries
    double e0 = 0, e1 = 0, mx = 0;
    for (int smp = 0; smp < 4000; ++smp) {
        s = s * 1664525u + 1013904223u; const int i = (s >> 8) % M; s = s * 1664525u + 1013904223u; const int j = (s >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)i * K + k] * (double)hb[(size_t)k * N + j];
        e0 = std::max(e0, fabs(c0[(size_t)i * N + j] - ref)); e1 = std::max(e1, fabs(c1[(size_t)i * N + j] - ref)); mx = std::max(mx, fabs(ref));
    }
    double dm = 0;
    for (size_t i = 0; i < c0.size(); ++i) dm = std::max(dm, (double)fabsf(c0[i] - c1[i]));
    const double fl = 2.0 * M * N * (double)K;
    printf("M=%d N=%d K=%d  fp32 tile %.3f ms %.1f TF | bf16x3 tile %.3f ms %.1f TF-eq | split+transpose of B %.3f ms\n", M, N, K, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t_split);
    printf("   max|err| vs fp64 (sampled): fp32 %.3g  bf16x3 %.3g   (max|C| %.3g)   max|fp32 - bf16x3| %.3g  = %.2g of max|C|\n", e0, e1, mx, dm, dm / mx);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C0); (void)hipFree(C1); (void)hipFree(Bt);
    return 0;
}

int main() {
    const int M = 4096, N = 4096, K = 4096;
    __bf16* src; float* C;
    (void)hipMalloc(&src, 4096 * 2); (void)hipMalloc(&C, (size_t)M * N * 4);
    std::vector<unsigned short> hsrc(4096);          // random normal-range bf16 bit patterns: zero operands flatter the clock
    for (int i = 0; i < 4096; ++i) hsrc[i] = (unsigned short)(0x3c00u + ((i * 2654435761u) >> 20) % 0x0400u) | ((i & 1) ? 0x8000u : 0u);
    (void)hipMemcpy(src, hsrc.data(), 4096 * 2, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * N * (double)K;      // fp32-equivalent FLOPs (the bf16 MFMAs execute 6x that)
    for (int round = 0; round < 2; ++round) {
        const float t0 = ceiling::run<0>(src, C, M, N, K), t1 = ceiling::run<1>(src, C, M, N, K);
        printf("round %d  bf16x3 mfma+ldsread %.3f ms = %.1f TF fp32-equivalent (%.0f TF bf16)   mfma-only %.3f ms = %.1f TF eq (%.0f TF bf16)\n", round,
               t0, fl / t0 / 1e9, 6 * fl / t0 / 1e9, t1, fl / t1 / 1e9, 6 * fl / t1 / 1e9);
    }
    full_path(512, 4608, 4608);
    full_path(4096, 4096, 4096);
    full_path(1024, 256, 256);
    full_path(256, 2304, 2304);
    return 0;
}
