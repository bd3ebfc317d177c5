// Study: the 256 x 128 LDS-DMA tile of gemm_f16x2_v2.hpp with FOUR consumer waves of 128 x 64 (4 x 2 MFMA blocks, 128 accumulator registers)
// instead of eight of 64 x 64: per k16 half a wave reads 8 A + 4 B fragments for 24 MFMAs (0.5 LDS reads per MFMA against 0.67), 96 KB
// instead of 128 KB of fragment reads per step and CU -- the shipped tile's LDS traffic (1,400 of 1,536 cycles per step) is as long as its
// matrix work.  One consumer wave and one loader wave per SIMD.  Same products in the same order per accumulator: bitwise equal output.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nsgp-repre_amd/csrc -I include tools/tile_wide_wave_bench.hip -o tools/_build/tile_wide_wave_bench
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "gemm_core.hpp"
#include "gemm_f16x2.hpp"
#include "gemm_f16x2_v2.hpp"

using namespace nsgp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int V2W_THREADS = 512;

__device__ __forceinline__ void gemm_tile_f16x2_v2w(const void* __restrict__ Asplit, int a_block0, const void* __restrict__ Bsplit, int b_block0, int K,
                                                    char* smem, f32x16 (&acc)[4][2]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t blk = (size_t)K * 256;
    const int nk = K / V2_BK;
    if (wave >= 4) {
        const int l = wave - 4;
        unsigned long long src[V2_MB_MAX + V2_NB];
#pragma unroll
        for (int b = 0; b < V2_MB_MAX; ++b) src[b] = v2_uniform((unsigned long long)(uintptr_t)Asplit + (size_t)(a_block0 + b) * blk + (size_t)l * (2 * V2_PLANE));
#pragma unroll
        for (int b = 0; b < V2_NB; ++b) src[V2_MB_MAX + b] = v2_uniform((unsigned long long)(uintptr_t)Bsplit + (size_t)(b_block0 + b) * blk + (size_t)l * (2 * V2_PLANE));
        const unsigned voff = lane * 16;
        const unsigned my_planes = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem + l * (2 * V2_PLANE));
        auto issue = [&](int stage) {
#pragma unroll
            for (int b = 0; b < V2_MB_MAX + V2_NB; ++b) {
                v2_dma_one(src[b], voff, my_planes + stage * V2_STAGE + b * V2_STEP);
                v2_dma_one(src[b] + V2_PLANE, voff, my_planes + stage * V2_STAGE + b * V2_STEP + V2_PLANE);
                src[b] += V2_STEP;
            }
        };
        auto lstep = [&](int t, auto st_next2) {
            if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
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
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const lds_char* abase = (const lds_char*)smem + (2 * wm) * V2_STEP + h * (2 * V2_PLANE) + r * 16;
    const lds_char* bbase = (const lds_char*)smem + (V2_MB_MAX + wn) * V2_STEP + h * (2 * V2_PLANE) + r * 16;
    typedef const __attribute__((address_space(3))) h16x8* lds_frag;
    h16x8 f0a[4][2], f0b[2][2], f1a[4][2], f1b[2][2];      // [32-row block][term] of k16 half 0 / half 1
    auto read_half = [&](auto st, auto ks_, h16x8 (&fa)[4][2], h16x8 (&fb)[2][2]) {
        constexpr int ST = decltype(st)::value, ks = decltype(ks_)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j][p] = *reinterpret_cast<lds_frag>(abase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + (j >> 1) * V2_STEP + (j & 1) * 512);
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[i][p] = *reinterpret_cast<lds_frag>(bbase + ST * V2_STAGE + ks * (4 * V2_PLANE) + p * V2_PLANE + i * 512);
        }
    };
    auto mfma_half = [&](const h16x8 (&fa)[4][2], const h16x8 (&fb)[2][2]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
    };
    auto pin_half = [&]() {      // 12 x (one LDS read, one MFMA), then the remaining 12 MFMAs
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    };
    auto first = [&](auto st) {
        v2_barrier();
        read_half(st, IC<0>{}, f0a, f0b);
        read_half(st, IC<1>{}, f1a, f1b);
        mfma_half(f0a, f0b);
        __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
    };
    auto steady = [&](auto st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v2_barrier();
        read_half(st, IC<0>{}, f0a, f0b);
        mfma_half(f1a, f1b);
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
    mfma_half(f1a, f1b);
}

__global__ __launch_bounds__(V2W_THREADS, 2) void v2w_kernel(const void* As, const void* Bs, const float* rinv, const float* cinv, float* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 128;
    gemm_tile_f16x2_v2w(As, m0 / 64, Bs, n0 / 64, K, smem_c, acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + 128 * wm + 32 * mi + acc_row(v, lane), col = n0 + 64 * wn + 32 * ni + (lane & 31);
                C[(size_t)row * N + col] = rinv[row] * (cinv[col] * acc[mi][ni][v]);
            }
}

__global__ __launch_bounds__(V2L_THREADS, 3) void v2l_kernel(const void* As, const void* Bs, const float* rinv, const float* cinv, float* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    f32x16 acc[2][2];
    zero_acc(acc);
    const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 128;
    gemm_tile_f16x2_v2l<4>(As, m0 / 64, Bs, n0 / 64, K, smem_c, acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 8) return;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + 64 * wm + 32 * mi + acc_row(v, lane), col = n0 + 64 * wn + 32 * ni + (lane & 31);
                C[(size_t)row * N + col] = rinv[row] * (cinv[col] * acc[mi][ni][v]);
            }
}

template <class F>
static float time_it(F f, int reps = 30, int warm = 30) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < warm; ++i) f();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

static int run(int M, int N, int K) {
    float *A, *B, *C, *rinv, *cscale, *cinv; void *As, *Bs;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)K * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&As, v2_operand_bytes(M, K))); CK(hipMalloc(&Bs, v2_operand_bytes(N, K)));
    CK(hipMalloc(&rinv, M * 4)); CK(hipMalloc(&cscale, N * 4)); CK(hipMalloc(&cinv, N * 4));
    std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
    unsigned s = 777u + M;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * 1e-3f;
    for (auto& v : hb) v = rnd() * 0.05f;
    CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(nsgp_split_rows_f16x2_kernel, dim3(M / 8), dim3(256), 0, 0, A, M, K, As, rinv);
    hipLaunchKernelGGL(nsgp_col_scales_f16x2_kernel, dim3((N + 31) / 32), dim3(256), 0, 0, B, K, N, cscale, cinv);
    hipLaunchKernelGGL(nsgp_split_transpose_f16x2_v2_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, 0, B, K, N, cscale, Bs);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v2l_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v2w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM_BYTES));
    CK(hipDeviceSynchronize());
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
    hipLaunchKernelGGL(v2l_kernel, dim3(N / 128, M / 256), dim3(V2L_THREADS), V2_SMEM_BYTES, 0, As, Bs, rinv, cinv, C, M, N, K);
    CK(hipGetLastError());
    CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    hipLaunchKernelGGL(v2w_kernel, dim3(N / 128, M / 256), dim3(V2W_THREADS), V2_SMEM_BYTES, 0, As, Bs, rinv, cinv, C, M, N, K);
    CK(hipGetLastError());
    CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < c1.size(); ++i) bad += c1[i] != c2[i];
    const double fl = 2.0 * M * N * (double)K;
    for (int rep = 0; rep < 2; ++rep) {
        const float tl = time_it([&] { hipLaunchKernelGGL(v2l_kernel, dim3(N / 128, M / 256), dim3(V2L_THREADS), V2_SMEM_BYTES, 0, As, Bs, rinv, cinv, C, M, N, K); });
        const float tw = time_it([&] { hipLaunchKernelGGL(v2w_kernel, dim3(N / 128, M / 256), dim3(V2W_THREADS), V2_SMEM_BYTES, 0, As, Bs, rinv, cinv, C, M, N, K); });
        printf("M %d N %d K %d: 8 consumers of 64 x 64: %.3f ms = %.1f TF-eq | 4 consumers of 128 x 64: %.3f ms = %.1f TF-eq | differing elements %zu\n", M, N, K, tl, fl / tl / 1e9, tw,
               fl / tw / 1e9, bad);
    }
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(As); (void)hipFree(Bs); (void)hipFree(rinv); (void)hipFree(cscale); (void)hipFree(cinv);
    return 0;
}

int main() {
    int rc = 0;
    rc |= run(512, 512, 512);
    rc |= run(4096, 4096, 4096);
    rc |= run(2048, 4608, 4608);
    rc |= run(8192, 8192, 2048);
    return rc;
}
