// What the matrix pipe sustains on this box: v_mfma_f32_32x32x16_f16 (and v_mfma_f32_32x32x2_f32) back to back from registers, no memory traffic,
// 1 / 2 / 3 waves per SIMD with 4 independent accumulators each, ~1 ms per launch.  The two-term fp16 tile's 425-437 TF fp32-equivalent
// (x 3 executed products) and the fp32 kernels' rates are to be read against THESE numbers, not only the data-sheet peaks.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak_bench.hip -o tools/_build/mfma_peak_bench
#include <cstdio>
#include <hip/hip_runtime.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool F16>
__global__ void mfma_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int b = 0; b < 4; ++b)
        for (int v = 0; v < 16; ++v) acc[b][v] = 0.0f;
    h16x8 a, bb;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); bb[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    const float fa = 0.001f * threadIdx.x, fb = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (F16) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc[b], 0, 0, 0);
                else acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[b], 0, 0, 0);
            }
    }
    float s = 0;
    for (int b = 0; b < 4; ++b)
        for (int v = 0; v < 16; ++v) s += acc[b][v];
    if (s == 12345.678f) out[0] = s;
}

template <bool F16>
static void run(int waves_per_simd) {
    float* out;
    (void)hipMalloc(&out, 4);
    const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = F16 ? 6000 : 3000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<F16>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<F16>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops_per_mfma = F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double total = (double)blocks * 4 * waves_per_simd * iters * 16 * flops_per_mfma;
    printf("%s, %d wave(s) per SIMD: %.3f ms per launch, %.1f TFLOP/s\n", F16 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_32x32x2_f32", waves_per_simd, ms, total / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main() {
    for (int w = 1; w <= 3; ++w) run<true>(w);
    for (int w = 1; w <= 3; ++w) run<false>(w);
    return 0;
}
