// EXPERIMENT for the next round: two-term fp16 split instead of the three-term bf16 split of the projection GEMM.
//   a = a0 + a1 in fp16 (2 x 11 = 22 mantissa bits), a*b ~= a1 b0 + a0 b1 + a0 b0: THREE v_mfma_f32_32x32x16_f16 per
//   fp32-equivalent product and 4 bytes per element of the split projector (the bf16x3 path: six MFMAs, 6 bytes), at the
//   price of fp16's 5-bit exponent: operands must be scaled into range (here: one power of two per matrix, undone in fp32).
// Same tile / pipeline as csrc/gemm_bf16x3.hpp (128 x 128, k16 steps, two LDS stages, three register sets).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nsgp-repre_amd/csrc -o /tmp/f16x2 tools/f16x2_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "gemm_core.hpp"
using namespace nsgp;

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) h16x8 g_h16x8;

constexpr int F2_BK = 16;
constexpr int F2_OCT = BM * 8 + 32;
constexpr int F2_PLANE = 2 * F2_OCT;
constexpr int F2_STAGE = 4 * F2_PLANE;            // A0 A1 B0 B1
constexpr int F2_SMEM = SMEM_BYTES;

struct F2Regs { f32x4 a[2]; h16x8 b[2]; };

__device__ __forceinline__ void f2_split(const f32x4 lo4, const f32x4 hi4, float scale, h16x8& p0, h16x8& p1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e < 4 ? lo4[e] : hi4[e - 4]) * scale;
        const _Float16 h = (_Float16)x;
        p0[e] = h;
        p1[e] = (_Float16)(x - (float)h);
    }
}

__global__ __launch_bounds__(256) void f2_split_transpose(const float* __restrict__ P, int K, int N, float scale, _Float16* __restrict__ Bt) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = P[(long)(k0 + i) * N + n0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        const float x = tile[tx][i] * scale;
        const _Float16 h = (_Float16)x;
        _Float16* dst = Bt + ((long)n * K + (k & ~7)) * 2 + (k & 7);      // [n][k/8][term][8]
        dst[0] = h;
        dst[8] = (_Float16)(x - (float)h);
    }
}

template <int ABL>
__global__ __launch_bounds__(256, 2) void f2_kernel(const float* __restrict__ A, const _Float16* __restrict__ Bt, float* __restrict__ C,
                                                    int M, int N, int K, float a_scale, float unscale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    _Float16* smem = reinterpret_cast<_Float16*>(sm);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[2][2];
    zero_acc(acc);
    const float* pa = A + (long)(m0 + (t >> 1)) * K + (t & 1) * 8;
    const _Float16* pb = Bt + (long)(n0 + (t >> 1)) * K * 2 + (t & 1) * 16;
    const int slot = (t & 1) * F2_OCT + (t >> 1) * 8;
    const int nk = K / F2_BK, last = nk - 1;
    auto load = [&](long k0, F2Regs& r) {
        r.a[0] = *(const gf32x4*)(pa + k0);
        r.a[1] = *(const gf32x4*)(pa + k0 + 4);
        r.b[0] = *(const g_h16x8*)(pb + 2 * k0);
        r.b[1] = *(const g_h16x8*)(pb + 2 * k0 + 8);
    };
    auto write = [&](_Float16* st, const F2Regs& r) {
        h16x8 p0, p1;
        f2_split(r.a[0], r.a[1], a_scale, p0, p1);
        *reinterpret_cast<h16x8*>(st + 0 * F2_PLANE + slot) = p0;
        *reinterpret_cast<h16x8*>(st + 1 * F2_PLANE + slot) = p1;
        *reinterpret_cast<h16x8*>(st + 2 * F2_PLANE + slot) = r.b[0];
        *reinterpret_cast<h16x8*>(st + 3 * F2_PLANE + slot) = r.b[1];
    };
    F2Regs regs[3];
    load(0, regs[0]); load((long)min(1, last) * F2_BK, regs[1]); load((long)min(2, last) * F2_BK, regs[2]);
    write(smem, regs[0]);
    load((long)min(3, last) * F2_BK, regs[0]);
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
        if (ABL < 2) write(nxt, regs[S]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        if (ABL < 1) load((long)min(kt + 4, last) * F2_BK, regs[S]);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 5 < nk; kt += 6) {
        step(kt, IC<0>{}, IC<1>{}); step(kt + 1, IC<1>{}, IC<2>{}); step(kt + 2, IC<0>{}, IC<0>{});
        step(kt + 3, IC<1>{}, IC<1>{}); step(kt + 4, IC<0>{}, IC<2>{}); step(kt + 5, IC<1>{}, IC<0>{});
    }
    if (kt < nk) { step(kt, IC<0>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<2>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<0>{}); ++kt; }
    if (kt < nk) { step(kt, IC<1>{}, IC<1>{}); ++kt; }
    if (kt < nk) { step(kt, IC<0>{}, IC<2>{}); ++kt; }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e)
        C[(long)(m0 + wm * 64 + mi * 32 + acc_row(e, lane)) * N + n0 + wn * 64 + ni * 32 + (lane & 31)] = acc[mi][ni][e] * unscale;
}

// ---- variant: k32 steps (two MFMA k16 sub-steps per barrier), full 128-B lines per row and thread pair -----------------
constexpr int G2_OCT = BM * 8 + 32;
constexpr int G2_PLANE = 4 * G2_OCT;             // 4 octets per step
constexpr int G2_STAGE = 4 * G2_PLANE;           // A0 A1 B0 B1: 33,792 B
struct G2Regs { f32x4 a[4]; h16x8 b[4]; };       // octets 2*(t&1), 2*(t&1)+1: A 16 floats, B 2 octets x 2 terms

template <int NSETS>
__global__ __launch_bounds__(256, 2) void g2_kernel(const float* __restrict__ A, const _Float16* __restrict__ Bt, float* __restrict__ C,
                                                    int M, int N, int K, float a_scale, float unscale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    _Float16* smem = reinterpret_cast<_Float16*>(sm);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[2][2];
    zero_acc(acc);
    const float* pa = A + (long)(m0 + (t >> 1)) * K + (t & 1) * 16;
    const _Float16* pb = Bt + (long)(n0 + (t >> 1)) * K * 2 + (t & 1) * 32;
    const int slot = (t & 1) * 2 * G2_OCT + (t >> 1) * 8;
    const int nk = K / 32, last = nk - 1;
    auto load = [&](long k0, G2Regs& r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.a[i] = *(const gf32x4*)(pa + k0 + 4 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.b[i] = *(const g_h16x8*)(pb + 2 * k0 + 8 * i);     // oct0 t0, oct0 t1, oct1 t0, oct1 t1
    };
    auto write_a = [&](_Float16* st, const G2Regs& r) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            h16x8 p0, p1;
            f2_split(r.a[2 * o], r.a[2 * o + 1], a_scale, p0, p1);
            *reinterpret_cast<h16x8*>(st + 0 * G2_PLANE + slot + o * G2_OCT) = p0;
            *reinterpret_cast<h16x8*>(st + 1 * G2_PLANE + slot + o * G2_OCT) = p1;
        }
    };
    auto write_b = [&](_Float16* st, const G2Regs& r) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            *reinterpret_cast<h16x8*>(st + 2 * G2_PLANE + slot + o * G2_OCT) = r.b[2 * o];
            *reinterpret_cast<h16x8*>(st + 3 * G2_PLANE + slot + o * G2_OCT) = r.b[2 * o + 1];
        }
    };
    G2Regs regs[NSETS];
#pragma unroll
    for (int i = 0; i < NSETS; ++i) load((long)min(i, last) * 32, regs[i]);
    write_a(smem, regs[0]); write_b(smem, regs[0]);
    load((long)min(NSETS, last) * 32, regs[0]);
    __syncthreads();
    auto step = [&](int kt, auto rb, auto s) {
        constexpr int RB = decltype(rb)::value, S = decltype(s)::value;
        const _Float16* cur = smem + RB * G2_STAGE;
        _Float16* nxt = smem + (1 - RB) * G2_STAGE;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i][p] = *reinterpret_cast<const h16x8*>(cur + p * G2_PLANE + (2 * ks + h) * G2_OCT + (wm * 64 + i * 32 + r) * 8);
                    fb[i][p] = *reinterpret_cast<const h16x8*>(cur + (2 + p) * G2_PLANE + (2 * ks + h) * G2_OCT + (wn * 64 + i * 32 + r) * 8);
                }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
            if (ks == 0) write_a(nxt, regs[S]); else write_b(nxt, regs[S]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
            if (ks == 1) { load((long)min(kt + 1 + NSETS, last) * 32, regs[S]); __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ); }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    };
    int kt = 0;
    if (NSETS == 2) {
        for (; kt + 1 < nk; kt += 2) { step(kt, IC<0>{}, IC<1>{}); step(kt + 1, IC<1>{}, IC<0>{}); }
        if (kt < nk) step(kt, IC<0>{}, IC<1>{});
    } else {
        for (; kt + 5 < nk; kt += 6) {
            step(kt, IC<0>{}, IC<1 % NSETS>{}); step(kt + 1, IC<1>{}, IC<2 % NSETS>{}); step(kt + 2, IC<0>{}, IC<0>{});
            step(kt + 3, IC<1>{}, IC<1 % NSETS>{}); step(kt + 4, IC<0>{}, IC<2 % NSETS>{}); step(kt + 5, IC<1>{}, IC<0>{});
        }
        if (kt < nk) { step(kt, IC<0>{}, IC<1 % NSETS>{}); ++kt; }
        if (kt < nk) { step(kt, IC<1>{}, IC<2 % NSETS>{}); ++kt; }
        if (kt < nk) { step(kt, IC<0>{}, IC<0>{}); ++kt; }
        if (kt < nk) { step(kt, IC<1>{}, IC<1 % NSETS>{}); ++kt; }
        if (kt < nk) { step(kt, IC<0>{}, IC<2 % NSETS>{}); ++kt; }
    }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e)
        C[(long)(m0 + wm * 64 + mi * 32 + acc_row(e, lane)) * N + n0 + wn * 64 + ni * 32 + (lane & 31)] = acc[mi][ni][e] * unscale;
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

int main() {
    const int M = 4096, N = 4096, K = 4096;
    float *A, *B, *C; _Float16* Bt;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)K * N * 4); (void)hipMalloc(&C, (size_t)M * N * 4); (void)hipMalloc(&Bt, (size_t)2 * N * K * 2);
    std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * 1e-3f * ((s & 0x300) == 0 ? 1e-4f : 1.0f);     // a quarter of the entries 4 decades smaller
    for (auto& v : hb) v = rnd() * 0.05f * ((s & 0xc00) == 0 ? 1e-5f : 1.0f);
    (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    // power-of-two scales that bring max|a|, max|b| to ~2^0 (exact to undo)
    float ma = 0, mb = 0; for (float v : ha) ma = std::max(ma, fabsf(v)); for (float v : hb) mb = std::max(mb, fabsf(v));
    const float sa = exp2f(-ceilf(log2f(ma))), sb = exp2f(-ceilf(log2f(mb)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM);
    hipLaunchKernelGGL(f2_split_transpose, dim3(N / 32, K / 32), dim3(256), 0, 0, B, K, N, sb, Bt);
    const double fl = 2.0 * M * N * (double)K;
    for (int round = 0; round < 2; ++round) {
        const float t0 = time_it([&] { hipLaunchKernelGGL(f2_kernel<0>, dim3(N / BN, M / BM), dim3(256), F2_SMEM, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb)); });
        const float t1 = time_it([&] { hipLaunchKernelGGL(f2_kernel<1>, dim3(N / BN, M / BM), dim3(256), F2_SMEM, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb)); });
        printf("round %d  f16x2 tile %.3f ms = %.1f TF fp32-equivalent   (no global loads in the loop: %.1f)\n", round, t0, fl / t0 / 1e9, fl / t1 / 1e9);
    }
    {   // k32-step variants: same results expected (same summation order per accumulator)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(g2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_STAGE * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(g2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_STAGE * 2);
        hipLaunchKernelGGL(f2_kernel<0>, dim3(N / BN, M / BM), dim3(256), F2_SMEM, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb));
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        (void)hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
        const float g2 = time_it([&] { hipLaunchKernelGGL(g2_kernel<2>, dim3(N / BN, M / BM), dim3(256), 2 * G2_STAGE * 2, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb)); });
        (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
        double dm = 0; for (size_t i = 0; i < c0.size(); ++i) dm = std::max(dm, (double)fabsf(c0[i] - c1[i]));
        const float g3 = time_it([&] { hipLaunchKernelGGL(g2_kernel<3>, dim3(N / BN, M / BM), dim3(256), 2 * G2_STAGE * 2, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb)); });
        printf("k32-step variant: 2 register sets %.3f ms = %.1f TF-eq | 3 sets %.3f ms = %.1f TF-eq   max|diff| vs k16 tile %.3g\n", g2, fl / g2 / 1e9, g3, fl / g3 / 1e9, dm);
    }
    hipLaunchKernelGGL(f2_kernel<0>, dim3(N / BN, M / BM), dim3(256), F2_SMEM, 0, A, Bt, C, M, N, K, sa, 1.0f / (sa * sb));
    std::vector<float> c((size_t)M * N);
    (void)hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
    double e = 0, mx = 0;
    for (int smp = 0; smp < 4000; ++smp) {
        s = s * 1664525u + 1013904223u; const int i = (s >> 8) % M; s = s * 1664525u + 1013904223u; const int j = (s >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)i * K + k] * (double)hb[(size_t)k * N + j];
        e = std::max(e, fabs(c[(size_t)i * N + j] - ref)); mx = std::max(mx, fabs(ref));
    }
    printf("f16x2 max|err| vs fp64 (sampled) %.3g at max|C| %.3g = %.2g of max|C|   (the fp32 MFMA and the bf16x3 tile: ~2e-6 on comparable data)\n", e, mx, e / mx);
    return 0;
}
