// K3: feature-covariance accumulation.  Replaces BRNullSpaceRunner.compute_cov + update_cov
// (mmdet/engine/runner/nsrunner_roi_replay.py:876-916, 923-934):
//     X = F.unfold(mean(x, 0, keepdim), k, padding, stride).permute(0,2,1).reshape(-1, D)   # [L x D]
//     C (=|+=) X^T X
// The reference materialises X (up to 619 MB for one FPN conv).  Here X is never built:
//   1. nsgp_batch_mean_pad_kernel  batch mean written into a zero-bordered image xm[Cin][Hp][Wp]
//      (so the gather needs no bounds tests);
//   2. nsgp_cov_syrk_kernel        implicit-im2col SYRK on the fp32 MFMA core: both operands are
//      "rows" images whose row d=(c,i,j) at column l=(oy,ox) is xm[c][oy*sh+i][ox*sw+j]; only the
//      128x128 tiles on or above the diagonal are computed, the L dimension is split S ways so
//      that even a D=64 layer fills the chip, each split writing its partial tile to workspace;
//   3. nsgp_cov_reduce_kernel      C (=|+=) sum_s partial[s], mirrored into the lower triangle,
//      in a fixed order (deterministic; no float atomics).
#include <algorithm>

#include "common.hpp"
#include "gemm_core.hpp"

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

__global__ __launch_bounds__(256, 2) void nsgp_cov_syrk_kernel(const float* __restrict__ xm, ConvGeom g, int l_chunk,
                                                               float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nb = (g.D + BM - 1) / BM;
    int ti = 0, rem = blockIdx.x;
    while (rem >= nb - ti) { rem -= nb - ti; ++ti; }
    const int m0 = ti * BM, n0 = (ti + rem) * BN;
    const int l_beg = blockIdx.y * l_chunk;
    const int l_end = min(l_beg + l_chunk, g.L);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    zero_acc(acc);
    long base_a[4], base_b[4];
    im2col_rowbase(g, m0, base_a);
    im2col_rowbase(g, n0, base_b);
    float ra[2][4][4], rb[2][4][4];
    const float inv_wo = 1.0f / (float)g.Wo;
    mfma_pipeline<true>(
        (l_end - l_beg + BK - 1) / BK, smem, acc,
        [&](int t, auto s) {
            const PatchCols pc = patch_cols(g, inv_wo, l_beg + t * BK, l_end);   // shared by both operands
            stage_im2col(xm, base_a, pc, ra[decltype(s)::value]);
            stage_im2col(xm, base_b, pc, rb[decltype(s)::value]);
        },
        [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value]); },
        [&](float* img, int, auto s) { write_rows(img, rb[decltype(s)::value]); });
    float* out = partial + (long)blockIdx.y * g.D * g.D;
    if (g.D % BM == 0) {   // full tiles: park the block in LDS, store float4 rows (short K chunks make the store matter)
        acc_to_lds(smem, acc);
        for_each_row4(smem, [&](int r, int col, float4 v) {
            f32x4 q;
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            *(gf32x4*)(out + (long)(m0 + r) * g.D + n0 + col) = q;
        });
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + acc_row(r, lane);
                if (row < g.D && col < g.D) as_global(out)[(long)row * g.D + col] = acc[mi][ni][r];
            }
        }
}

__global__ __launch_bounds__(256) void nsgp_cov_reduce_kernel(const float* __restrict__ partial, int S, int D,
                                                              float* __restrict__ cov, int accumulate) {
    const long n = (long)D * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / D), c = (int)(idx - (long)r * D);
        // the tile that holds (r,c) was computed iff its tile row <= tile column; else read the mirror
        const long src = (r / BM <= c / BN) ? idx : ((long)c * D + r);
        float s = partial[src];
        for (int k = 1; k < S; ++k) s += partial[(long)k * n + src];
        cov[idx] = accumulate ? (cov[idx] + s) : s;
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

// split factor along L: enough workgroups to fill 256 CUs x 2, each split >= 4 K-steps
static void cov_split(int D, int L, int* S, int* l_chunk) {
    const int nb = (D + BM - 1) / BM;
    const int tiles = nb * (nb + 1) / 2;
    int s = (512 + tiles - 1) / tiles;
    const int max_s = std::max(1, L / (4 * BK));
    s = std::max(1, std::min(s, max_s));
    int chunk = (L + s - 1) / s;
    chunk = ((chunk + BK - 1) / BK) * BK;
    s = (L + chunk - 1) / chunk;
    *S = s;
    *l_chunk = chunk;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace nsgp

using namespace nsgp;

extern "C" size_t nsgp_cov_workspace_bytes(int cin, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw) {
    if (cin <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return 0;
    const int Hp = h + 2 * ph, Wp = w + 2 * pw;
    const int Ho = (Hp - kh) / sh + 1, Wo = (Wp - kw) / sw + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const int D = cin * kh * kw;
    int S, chunk;
    cov_split(D, Ho * Wo, &S, &chunk);
    return align256((size_t)cin * Hp * Wp * 4) + (size_t)S * D * D * 4;
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
    int S, chunk;
    cov_split(g.D, g.L, &S, &chunk);
    float* xm = static_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)cin * Hp * Wp * 4));
    const long n_img = (long)cin * Hp * Wp;
    hipLaunchKernelGGL(nsgp_batch_mean_pad_kernel, dim3((unsigned)std::min<long>(4096, (n_img + 255) / 256)), dim3(256), 0, stream,
                       x, batch, cin, h, w, ph, pw, xm);
    NSGP_LAUNCH_CHECK();
    NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nsgp_cov_syrk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    const int nb = (g.D + BM - 1) / BM;
    hipLaunchKernelGGL(nsgp_cov_syrk_kernel, dim3(nb * (nb + 1) / 2, S), dim3(THREADS), SMEM_BYTES, stream, xm, g, chunk, partial);
    NSGP_LAUNCH_CHECK();
    const long n = (long)g.D * g.D;
    hipLaunchKernelGGL(nsgp_cov_reduce_kernel, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, stream, partial, S, g.D, cov, accumulate);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int nsgp_cov_accumulate_linear(const float* x, int batch, int features, float* cov, int accumulate, void* stream_) {
    if (!x || !cov || batch <= 0 || features <= 0) return fail(NSGP_ERR_INVALID, "nsgp_cov_accumulate_linear: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long n = (long)features * features;
    hipLaunchKernelGGL(nsgp_cov_linear_kernel, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, stream, x, batch, features, cov, accumulate);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
