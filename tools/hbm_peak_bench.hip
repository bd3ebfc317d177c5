// What HBM sustains on this box for the access mixes of the step's launches: pure read, copy (1 read + 1 write), the un-projected update's mix
// (3 reads + 3 writes), the fused launch's (3 reads + 2 writes) and the apply launch's (2 reads + 1 write); 16-byte accesses, linear chunks
// of 4,096 elements per workgroup (the plain update kernel's pattern), non-temporal.  The roofline fractions in bench.py are priced against the
// 8 TB/s data-sheet figure; these are the rates to read them against.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_peak_bench.hip -o tools/_build/hbm_peak_bench
#include <cstdio>
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NR, int NW>
__global__ __launch_bounds__(256) void stream_kernel(float* const* bufs, long n) {
    const long base = (long)blockIdx.x * 4096;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long i = base + 4 * (threadIdx.x + 256 * it);
        if (i >= n) return;
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < NR; ++r) acc += __builtin_nontemporal_load((const f32x4*)(bufs[r] + i));
        if (NW == 0) { if (acc[0] == 123.456f) bufs[0][i] = acc[1]; }
#pragma unroll
        for (int w = 0; w < NW; ++w) __builtin_nontemporal_store(acc * (float)(w + 1), (f32x4*)(bufs[w] + i));
    }
}

template <int NR, int NW>
static void run(const char* what, float** d_ptrs, long n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = (int)((n + 4095) / 4096);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((stream_kernel<NR, NW>), dim3(blocks), dim3(256), 0, 0, d_ptrs, n);
    (void)hipEventRecord(e0, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_kernel<NR, NW>), dim3(blocks), dim3(256), 0, 0, d_ptrs, n);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-46s %ld M elements: %.3f ms per launch, %.2f TB/s\n", what, n >> 20, ms, (double)(NR + NW) * n * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    const long n = 64L << 20;      // 256 MB per buffer
    float* h[4];
    for (int i = 0; i < 4; ++i) { if (hipMalloc(&h[i], n * 4) != hipSuccess) return 1; (void)hipMemset(h[i], 0, n * 4); }
    float** d;
    (void)hipMalloc(&d, sizeof(h));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (long m : {n, n / 4}) {
        run<1, 0>("read", d, m);
        run<1, 1>("copy in place (1 read + 1 write)", d, m);
        run<2, 1>("apply launch's mix (2 reads + 1 write)", d, m);
        run<3, 2>("fused launch's mix (3 reads + 2 writes)", d, m);
        run<3, 3>("un-projected update's mix (3 reads + 3 writes)", d, m);
    }
    return 0;
}
