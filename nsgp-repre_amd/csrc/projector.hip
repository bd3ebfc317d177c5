// K5: null-space projector build.  Replaces get_transforms
// (mmdet/engine/optimizers/SGD_NSCL.py:270-285):
//     basis = V[:, ind];  P = basis @ basis.T;  P /= ||P||_F   (if normalise)
// where `ind` is always a column suffix [first_col, D) of V (adaptive_threshold sets
// mask[i_thres:] = True, SGD_NSCL.py:174-175; the NoAdaptive rule `s <= s_min*thres`
// on a descending spectrum is a suffix too).
//
// P is symmetric: only the 128x128 tiles on or above the diagonal are computed on
// the fp32 MFMA core (A = B = V[:, first_col:], "rows" image for both operands) and
// each is stored twice.  Loads start at the 32-aligned column below first_col with
// the leading columns zeroed in the LDS write, so the aligned float4 path is kept.
#include <algorithm>

#include "common.hpp"
#include "gemm_core.hpp"

namespace nsgp {

constexpr int NORM_BLOCKS = 1024;

template <bool FAST>
__global__ __launch_bounds__(256, 2) void nsgp_projector_kernel(const float* __restrict__ V, float* __restrict__ P,
                                                                int D, int first_col) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // blockIdx.x enumerates the upper-triangular tiles row by row
    const int nb = (D + BM - 1) / BM;
    int ti = 0, rem = blockIdx.x;
    while (rem >= nb - ti) { rem -= nb - ti; ++ti; }
    const int m0 = ti * BM, n0 = (ti + rem) * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    zero_acc(acc);
    float ra[2][4][4], rb[2][4][4];
    const int kbeg = first_col & ~(BK - 1);
    mfma_pipeline<true>(
        (D - kbeg + BK - 1) / BK, smem, acc,
        [&](int t, auto s) {
            stage_rows<FAST>(V, D, D, D, m0, kbeg + t * BK, ra[decltype(s)::value]);
            stage_rows<FAST>(V, D, D, D, n0, kbeg + t * BK, rb[decltype(s)::value]);
        },
        [&](float* img, int t, auto s) { write_rows_klo(img, ra[decltype(s)::value], kbeg + t * BK, first_col); },
        [&](float* img, int t, auto s) { write_rows_klo(img, rb[decltype(s)::value], kbeg + t * BK, first_col); });
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + acc_row(r, lane);
                if (row < D && col < D) {
                    as_global(P)[(long)row * D + col] = acc[mi][ni][r];
                    if (m0 != n0) as_global(P)[(long)col * D + row] = acc[mi][ni][r];
                }
            }
        }
}

// Frobenius norm, deterministic two-stage sum in double.
__global__ __launch_bounds__(256) void nsgp_sumsq_partial_kernel(const float* __restrict__ P, long n, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double v = P[i];
        s += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void nsgp_norm_final_kernel(double* __restrict__ partial, int n_partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    // slot NORM_BLOCKS holds the fp32 norm (torch.norm returns fp32) for the divide pass
    if (threadIdx.x == 0) reinterpret_cast<float*>(partial + NORM_BLOCKS)[0] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void nsgp_divide_kernel(float* __restrict__ P, long n, const double* __restrict__ partial) {
    const float nrm = reinterpret_cast<const float*>(partial + NORM_BLOCKS)[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) P[i] = P[i] / nrm;
}

// Head form: P = I - U U^T from the r REMOVED directions, U [D][rpad] row-major fp32 (columns >= rank zero, rpad a
// multiple of 32 up to 128, or 256), D % 32 == 0.  One wave per (32 rows x 256 columns) strip on v_mfma_f32_32x32x2_f32 with K = rpad;
// P[m][n] and P[n][m] are the same k-ordered chain of the same (commuting) products, so P is bit-symmetric.  This is the
// form the low-rank step applies (projected_step.hip), so `u @ P` and `c (u - (u U) U^T)` agree to the rounding of P's entries.
template <int NJ>
__device__ __forceinline__ void projector_head_body(const float* __restrict__ U, float* __restrict__ P, int D, int rpad, int m0, int n0, int ncols) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    f32x4 ta[4 * NJ];
    const float* trow = U + (long)(m0 + i) * rpad + 4 * h;
#pragma unroll
    for (int q = 0; q < 4 * NJ; ++q) ta[q] = *(const gf32x4*)(trow + 8 * q);
    for (int n = n0; n < n0 + ncols; n += 32) {
        f32x4 ub[4 * NJ];
        const float* urow = U + (long)(n + i) * rpad + 4 * h;
#pragma unroll
        for (int q = 0; q < 4 * NJ; ++q) ub[q] = *(const gf32x4*)(urow + 8 * q);
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4 * NJ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[q][e], ub[q][e], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + acc_row(v, lane), col = n + i;
            as_global(P)[(long)row * D + col] = (row == col ? 1.0f : 0.0f) - acc[v];
        }
    }
}

// rpad a multiple of 128 above 128: the same chain 128 k at a time (the fragments of a whole 256-wide row pair would not fit the registers)
__device__ __forceinline__ void projector_head_wide(const float* __restrict__ U, float* __restrict__ P, int D, int rpad, int m0, int n0, int ncols) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const float* trow = U + (long)(m0 + i) * rpad + 4 * h;
    for (int n = n0; n < n0 + ncols; n += 32) {
        const float* urow = U + (long)(n + i) * rpad + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
        for (int k0 = 0; k0 < rpad; k0 += 128) {
            f32x4 ta[16], ub[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { ta[q] = *(const gf32x4*)(trow + k0 + 8 * q); ub[q] = *(const gf32x4*)(urow + k0 + 8 * q); }
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[q][e], ub[q][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + acc_row(v, lane), col = n + i;
            as_global(P)[(long)row * D + col] = (row == col ? 1.0f : 0.0f) - acc[v];
        }
    }
}

__global__ __launch_bounds__(256) void nsgp_projector_head_kernel(const float* __restrict__ U, float* __restrict__ P, int D, int rpad) {
    const int strips = (D + 255) / 256;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (D / 32) * strips) return;
    const int m0 = (unit / strips) * 32, n0 = (unit % strips) * 256;
    const int ncols = min(256, D - n0);
    switch (rpad >> 5) {
        case 1: projector_head_body<1>(U, P, D, rpad, m0, n0, ncols); break;
        case 2: projector_head_body<2>(U, P, D, rpad, m0, n0, ncols); break;
        case 3: projector_head_body<3>(U, P, D, rpad, m0, n0, ncols); break;
        case 4: projector_head_body<4>(U, P, D, rpad, m0, n0, ncols); break;
        default: projector_head_wide(U, P, D, rpad, m0, n0, ncols); break;      // rpad = 256 (129 .. 256 removed directions)
    }
}

}  // namespace nsgp

using namespace nsgp;

extern "C" size_t nsgp_projector_scratch_bytes(int D) {
    const int nb = (D + BM - 1) / BM;
    (void)nb;
    return (NORM_BLOCKS + 2) * sizeof(double);
}

extern "C" int nsgp_build_projector(const float* V, int D, int first_col, int normalise, float* P, void* scratch,
                                    size_t scratch_bytes, void* stream_) {
    if (!V || !P || D <= 0 || first_col < 0 || first_col >= D) return fail(NSGP_ERR_INVALID, "nsgp_build_projector: bad argument (D=%d first_col=%d)", D, first_col);
    if (!scratch || scratch_bytes < nsgp_projector_scratch_bytes(D)) return fail(NSGP_ERR_WORKSPACE, "nsgp_build_projector: scratch too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int nb = (D + BM - 1) / BM;
    const int tiles = nb * (nb + 1) / 2;
    double* partial = static_cast<double*>(scratch);
    const bool fast = (D % BM == 0) && aligned16(V);
    if (fast) {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_projector_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(nsgp_projector_kernel<true>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, V, P, D, first_col);
    } else {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_projector_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        hipLaunchKernelGGL(nsgp_projector_kernel<false>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, V, P, D, first_col);
    }
    NSGP_LAUNCH_CHECK();
    if (normalise) {
        const long n = (long)D * D;
        int blocks = (int)std::min<long>(NORM_BLOCKS, (n + 255) / 256);
        hipLaunchKernelGGL(nsgp_sumsq_partial_kernel, dim3(blocks), dim3(256), 0, stream, P, n, partial);
        NSGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(nsgp_norm_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks);
        NSGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(nsgp_divide_kernel, dim3(2048), dim3(256), 0, stream, P, n, partial);
        NSGP_LAUNCH_CHECK();
    }
    return NSGP_OK;
}

static int normalise_projector(float* P, int D, double* partial, hipStream_t stream) {
    const long n = (long)D * D;
    int blocks = (int)std::min<long>(NORM_BLOCKS, (n + 255) / 256);
    hipLaunchKernelGGL(nsgp_sumsq_partial_kernel, dim3(blocks), dim3(256), 0, stream, P, n, partial);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nsgp_norm_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nsgp_divide_kernel, dim3(2048), dim3(256), 0, stream, P, n, partial);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int nsgp_build_projector_head(const float* U, int D, int rpad, int normalise, float* P, void* scratch,
                                         size_t scratch_bytes, void* stream_) {
    if (!U || !P || D <= 0 || D % 32 != 0 || rpad <= 0 || rpad % 32 != 0 || (rpad > 128 && rpad != 256) || !aligned16(U))
        return fail(NSGP_ERR_INVALID, "nsgp_build_projector_head: bad argument (D=%d rpad=%d; D %% 32 == 0, rpad in {32,64,96,128,256}, U 16-byte aligned)", D, rpad);
    if (!scratch || scratch_bytes < nsgp_projector_scratch_bytes(D)) return fail(NSGP_ERR_WORKSPACE, "nsgp_build_projector_head: scratch too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int units = (D / 32) * ((D + 255) / 256);
    hipLaunchKernelGGL(nsgp_projector_head_kernel, dim3((units + 3) / 4), dim3(256), 0, stream, U, P, D, rpad);
    NSGP_LAUNCH_CHECK();
    if (normalise) return normalise_projector(P, D, static_cast<double*>(scratch), stream);
    return NSGP_OK;
}
