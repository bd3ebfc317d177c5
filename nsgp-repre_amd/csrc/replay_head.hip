// RePRE per-step replay pass: the prototype bank through the task bbox head and the double-softmax CE, forward and backward,
// as a handful of launches -- StandardMultiPrototypeReplayHead.replay_loss
// (mmdet/models/roi_heads/standard_roi_replay_head.py:468-501) over Shared2FCBBoxHeadTask.forward
// (mmdet/models/roi_heads/bbox_heads/convfc_bbox_head_task.py:235-276):
//
//     H1 = relu(X W1^T + b1)        X = the bank [K x 12544] (a constant: no dX), W1 = shared_fcs.0 [1024 x 12544]
//     H2 = relu(H1 W2^T + b2)       W2 = shared_fcs.1 [1024 x 1024]
//     S  = H2 Wc^T + bc             Wc = the rows of the per-task fc_cls heads seen so far + background, stacked [C x 1024]
//     loss = CE(softmax(S), labels) (head:499, the reference's double softmax; replay_ce.hip)
//
// and the weight gradients  dW1 = dZ1^T X,  dW2 = dZ2^T H1,  dWc = dS^T H2  (+ bias gradients).  The reference runs this through
// autograd: ~15 library GEMM / elementwise launches forward and ~25 backward for K = 150 rows, i.e. pure launch overhead around
// one 51 MB weight pass each way (0.69 ms per step in round 2, host-bound).  Here: 6 launches forward, 5 backward, everything exact
// fp32 on v_mfma_f32_32x32x2_f32 in a fixed order (the bank pass is kept at the reference's fp32 width; bf16 autocast in the
// reference would only lower it).
//
// Kernels
//   rh_skinny_kernel   C[M x N] = A[M x K] B^T (or A B) for M <= 160 rows per workgroup: a workgroup owns ALL (<= 5) 32-row blocks of
//                      the skinny operand x 128 columns (one 32-column block per wave, 5 accumulators) x one K range (>= 4 k32 steps);
//                      the K ranges of one column tile write slabs that the next launch sums in range order (deterministic).  M = 150
//                      wastes 6 % of the matrix work (a 128 x 128 tile: 41 %).  K ranges are dealt to XCDs (blockIdx % 8) so that the
//                      skinny operand's K slice is fetched into one L2.
//   rh_reduce_kernel   H = relu(sum of slabs + bias).
//   rh_scores_kernel   S = H2 Wc^T + bc, one ROW per workgroup (thread = (class, k slice)), then the row's double-softmax CE term;
//                      the class rows are read in place through a table of the per-task heads (RhHeads), no stacked copy.
//   rh_mean_kernel     loss = ordered mean of the row terms.
//   rh_dz_kernel       dZ = (upstream) * (H > 0) with its transposed, zero-padded copy (the A operand of the weight-gradient GEMM)
//                      and the bias gradient (ordered column sums); upstream = the slab sum (dZ2 W2) or dS Wc; in the second mode also
//                      the class heads' gradients dWc = dS^T H2 (written straight into the per-head buffers) and dbc.
//   rh_tn_kernel       dW[No x Ni] = dZ^T[No x Mp] X[M x Ni] on the 128 x 128 fp32 tile of gemm_core.hpp, grouped over jobs.
#include <algorithm>

#include "common.hpp"
#include "gemm_core.hpp"
#include "replay_ce.hpp"

namespace nsgp {

constexpr int RH_MAX_MB = 5;          // 32-row blocks per workgroup of the skinny kernel
constexpr int RH_MAX_ROWS = 512;      // K <= 10 prototypes x 40 old classes = 400 (COCO 40+40)
constexpr int RH_MAX_COLS = 256;      // the CE kernels' limit
constexpr int RH_TARGET_WGS = 256;    // one workgroup per CU (measured on the first FC at K = 150: 51 us; 512: 54-57, 1024: 60)

template <int MB> constexpr int rh_aplane() { return MB * 32 * 4 + 4; }
template <int MB> constexpr int rh_smem_floats() { return 2 * 8 * rh_aplane<MB>() + 2 * ROW_IMG; }

struct RhGemm {
    const float* A; long lda;      // [M x K] row-major (rows beyond M are never read: the row index is clamped)
    const float* B; long ldb;      // B_ROWS: [N x K] row-major; else [K x N] row-major
    float* slabs;                  // [S][Mpad][N]
    int M, N, K;
    int mchunks, ntn, S, per;      // row chunks of MB*32, column tiles of 128, K ranges of `per` k-steps
    int Mpad;
};

template <int MB, bool B_ROWS, bool FAST>
__global__ __launch_bounds__(256, 2) void rh_skinny_kernel(const RhGemm g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int APLANE = rh_aplane<MB>();
    constexpr int AIMG = 8 * APLANE;
    float* const A0 = smem;
    float* const A1 = smem + AIMG;
    float* const B0 = smem + 2 * AIMG;
    float* const B1 = B0 + ROW_IMG;
    // blockIdx -> (K range, column tile, row chunk): ranges s = 8 a + x go to XCD x (blockIdx % 8)
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int mc = q % g.mchunks, q2 = q / g.mchunks;
    const int nt = q2 % g.ntn, s = (q2 / g.ntn) * 8 + x;
    if (s >= g.S) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = mc * MB * 32, n0 = nt * 128;
    const int nk = (g.K + 31) >> 5;
    const int kt0 = s * g.per, kt1 = min(nk, kt0 + g.per);
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
    float ra[MB][4], rb[4][4];
    const gfloat* Ag = as_global(g.A);
    const gfloat* Bg = as_global(g.B);
    auto load = [&](int kt) {
        const int k = kt * 32 + (t & 7) * 4;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int row = min(m0 + (t >> 3) + 32 * j, g.M - 1);
            if (FAST) {
                const f32x4 v = *(const gf32x4*)(g.A + (long)row * g.lda + k);
                ra[j][0] = v[0]; ra[j][1] = v[1]; ra[j][2] = v[2]; ra[j][3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[j][e] = (k + e < g.K) ? Ag[(long)row * g.lda + k + e] : 0.0f;
            }
        }
        if (B_ROWS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = n0 + (t >> 3) + 32 * j;
                if (FAST) {
                    const f32x4 v = *(const gf32x4*)(g.B + (long)row * g.ldb + k);
                    rb[j][0] = v[0]; rb[j][1] = v[1]; rb[j][2] = v[2]; rb[j][3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[j][e] = (row < g.N && k + e < g.K) ? Bg[(long)row * g.ldb + k + e] : 0.0f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = kt * 32 + (t >> 5) + 8 * j, n = n0 + (t & 31) * 4;
                if (FAST) {
                    const f32x4 v = *(const gf32x4*)(g.B + (long)kr * g.ldb + n);
                    rb[j][0] = v[0]; rb[j][1] = v[1]; rb[j][2] = v[2]; rb[j][3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[j][e] = (kr < g.K && n + e < g.N) ? Bg[(long)kr * g.ldb + n + e] : 0.0f;
                }
            }
        }
    };
    auto stage_write = [&](float* Ai, float* Bi) {
#pragma unroll
        for (int j = 0; j < MB; ++j)
            *reinterpret_cast<float4*>(Ai + (t & 7) * APLANE + ((t >> 3) + 32 * j) * 4) = make_float4(ra[j][0], ra[j][1], ra[j][2], ra[j][3]);
        if (B_ROWS) write_rows(Bi, rb);
        else write_kn(Bi, rb);
    };
    auto compute = [&](const float* Ai, const float* Bi) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int qq = 0; qq < 8; qq += 2) {
            float b[4];
            if (B_ROWS) {
                const float4 v = *reinterpret_cast<const float4*>(Bi + (qq + h) * QPLANE + (wave * 32 + r) * 4);
                b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bi[(4 * (qq + h) + j) * BN + wave * 32 + r];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float4 a = *reinterpret_cast<const float4*>(Ai + (qq + h) * APLANE + (mb * 32 + r) * 4);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc[mb], 0, 0, 0);
            }
        }
    };
    // two LDS buffers, one register set: the loads of step k+1 are issued before the MFMAs of step k (pinned there) and
    // written to the other buffer after them
    load(kt0);
    stage_write(A0, B0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
        const bool more1 = kt + 1 < kt1;
        if (more1) load(kt + 1);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
        compute(A0, B0);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        if (more1) stage_write(A1, B1);
        __syncthreads();
        if (!more1) break;
        const bool more2 = kt + 2 < kt1;
        if (more2) load(kt + 2);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_VMEM_READ);
        compute(A1, B1);
        __builtin_amdgcn_sched_barrier(SCHED_PIN_STAGING);
        if (more2) stage_write(A0, B0);
        __syncthreads();
    }
    gfloat* slab = as_global(g.slabs) + ((long)s * g.Mpad + m0) * g.N;
    const int col = n0 + wave * 32 + (lane & 31);
    if (FAST || col < g.N) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(long)(mb * 32 + acc_row(r, lane)) * g.N + col] = acc[mb][r];
    }
}

// H[m][n] = relu(bias[n] + sum_s slab[s][m][n]), ranges summed in order (loads of eight ranges in flight together)
template <bool VEC>
__global__ __launch_bounds__(256) void rh_reduce_kernel(const float* __restrict__ slabs, int S, int Mpad, int M, int N, const float* __restrict__ bias,
                                                        float* __restrict__ out) {
    const long stride = (long)Mpad * N;
    if (VEC) {
        const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
        if (i4 * 4 >= (long)M * N) return;
        const long i = i4 * 4;
        const int n = (int)(i % N);
        f32x4 sum = *(const gf32x4*)(slabs + i);
        int s = 1;
        for (; s + 8 <= S; s += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const gf32x4*)(slabs + (s + u) * stride + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; s < S; ++s) sum += *(const gf32x4*)(slabs + s * stride + i);
        const f32x4 b = *(const gf32x4*)(bias + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(sum[e] + b[e], 0.0f);
        *(gf32x4*)(out + i) = o;
    } else {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= (long)M * N) return;
        float sum = slabs[i];
        for (int s = 1; s < S; ++s) sum += slabs[s * stride + i];
        out[i] = fmaxf(sum + bias[i % N], 0.0f);
    }
}

// scores[m][c] = bc[c] + sum_k H[m][k] Wc[c][k]: ONE ROW per workgroup (K = 150 rows -> 150 workgroups), thread = (class, k slice):
// 8 consecutive lanes read 128 contiguous bytes of one class row, 32 classes per pass, the 8 slices of a class are summed with
// three shuffles; the row of H sits in LDS.  Then the row's double-softmax CE term rowloss[m] = logsumexp(softmax(s_m)) -
// softmax(s_m)[y_m] (head:499) from the scores still in LDS.
constexpr int RS_MAX_HIDDEN = 4096;

// The kept class rows live in the per-task fc_cls heads (tasks 1 .. task_id, then the background head): the kernels read them
// through this table instead of a stacked copy (the torch.cat of the rows and the split of their gradient were two launches and
// half a dozen autograd nodes per step on a host-bound pass).  Row c of the stacked view belongs to head h with row0[h] <= c < row0[h+1].
constexpr int RH_MAX_HEADS = 16;
struct RhHeads {
    const float* w[RH_MAX_HEADS];
    const float* b[RH_MAX_HEADS];
    float* gw[RH_MAX_HEADS];
    float* gb[RH_MAX_HEADS];
    int row0[RH_MAX_HEADS + 1];
    int n;
};
__device__ __forceinline__ int rh_head_of(const RhHeads& hd, int c) {
    int h = 0;
    while (h + 1 < hd.n && c >= hd.row0[h + 1]) ++h;
    return h;
}
__global__ __launch_bounds__(256) void rh_scores_kernel(const float* __restrict__ H, int M, int hidden, const RhHeads hd, int C,
                                                        const long long* __restrict__ labels,
                                                        float* __restrict__ scores, float* __restrict__ rowloss) {
    __shared__ __attribute__((aligned(16))) float hs[RS_MAX_HIDDEN];
    __shared__ float sc[RH_MAX_COLS];
    const int m = blockIdx.x, t = threadIdx.x, ks = t & 7, cl = t >> 3;
    const bool staged = hidden <= RS_MAX_HIDDEN;
    if (staged) {
        for (int k = t; k < hidden; k += 256) hs[k] = H[(long)m * hidden + k];
        __syncthreads();
    }
    bool vec = staged && (hidden & 3) == 0;
    for (int h = 0; h < hd.n; ++h) vec = vec && (((uintptr_t)hd.w[h]) & 15u) == 0;
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cl;
        float acc = 0.0f;
        if (c < C) {
            const int hh = rh_head_of(hd, c);
            const float* w = hd.w[hh] + (long)(c - hd.row0[hh]) * hidden;
            if (vec) {
                for (int k = 4 * ks; k < hidden; k += 32) {
                    const f32x4 wv = *(const gf32x4*)(w + k);
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + k);
                    acc += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
                }
            } else {
                const float* hrow = staged ? hs : H + (long)m * hidden;
                for (int k = ks; k < hidden; k += 8) acc += hrow[k] * w[k];
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (ks == 0 && c < C) {
            const int hh = rh_head_of(hd, c);
            const float v = acc + hd.b[hh][c - hd.row0[hh]];
            sc[c] = v;
            scores[(long)m * C + c] = v;
        }
    }
    __syncthreads();
    if (t < 64) {
        float q[4];
        const float lse = row_double_softmax(sc, C, t, q);
        const int y = (int)labels[m];
        float qy = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (t + 64 * e == y) qy = q[e];
        qy = wave_sum(qy);
        if (t == 0) rowloss[m] = lse - qy;
    }
}

// loss = mean of the row terms, fixed summation order
__global__ __launch_bounds__(256) void rh_mean_kernel(const float* __restrict__ rowloss, int M, float* __restrict__ loss_out) {
    __shared__ float part[256];
    float s = 0.0f;
    for (int m = threadIdx.x; m < M; m += 256) s += rowloss[m];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = part[0] / (float)M;
}

// dZ[m][n] = up[m][n] * (H[m][n] > 0) for a block of 16 columns and ALL rows; up = the sum of the K-range slabs (MODE 0: dZ2 W2)
// or dS Wc (MODE 1).  Outputs: dZ (optional), dZT[n][m] (zero-padded to Mp columns: the A operand of the weight-gradient GEMM),
// db[n] = sum_m dZ[m][n] (row groups summed in order).  MODE 1 also produces the class heads' gradients for its columns,
// dWc[c][n] = sum_m dS[m][c] H[m][n] (rows in order, 64 at a time through LDS), and one extra workgroup writes dbc = column sums of dS.
constexpr int DZ_COLS = 16, DZ_RC = 64, DZ_THREADS = 512, DZ_GROUPS = DZ_THREADS / DZ_COLS, DZ_WACC = RH_MAX_COLS * DZ_COLS / DZ_THREADS;
struct RhDz {
    const float* slabs; int S; int Mpad;         // MODE 0
    const float* dS; int C;                      // MODE 1 (+ hd: the class heads' rows and gradient buffers)
    const float* H;                              // [M x N] the layer's OUTPUT (post-ReLU): relu'(z) = (H > 0)
    int M, N, Mp;
    float* dZ;                                   // [M x N] or null
    float* dZT;                                  // [N x Mp]
    float* db;                                   // [N]
    RhHeads hd;                                  // MODE 1
};

template <int MODE>
__global__ __launch_bounds__(DZ_THREADS) void rh_dz_kernel(const RhDz a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int nblocks = (a.N + DZ_COLS - 1) / DZ_COLS;
    if (MODE == 1 && (int)blockIdx.x == nblocks) {        // the extra workgroup: column sums of dS
        for (int c = t; c < a.C; c += DZ_THREADS) {
            float s = 0.0f;
            for (int m = 0; m < a.M; ++m) s += a.dS[(long)m * a.C + c];
            const int hh = rh_head_of(a.hd, c);
            a.hd.gb[hh][c - a.hd.row0[hh]] = s;
        }
        return;
    }
    const int c = t & (DZ_COLS - 1), grp = t / DZ_COLS, n0 = blockIdx.x * DZ_COLS, n = n0 + c;
    const int TLD = a.Mp + 1;
    float* tile = lds;                                   // [16][Mp + 1]
    float* part = tile + DZ_COLS * TLD;                  // [32 groups][16]
    float* wct = part + DZ_GROUPS * DZ_COLS;                    // MODE 1: [C][16]
    float* dsc = wct + (MODE == 1 ? a.C * DZ_COLS : 0);  // MODE 1: [DZ_RC][C]
    float* hc = dsc + (MODE == 1 ? DZ_RC * a.C : 0);     // MODE 1: [DZ_RC][16]
    if (MODE == 1) {
        for (int idx = t; idx < a.C * DZ_COLS; idx += DZ_THREADS) {
            const int cc = idx / DZ_COLS, col = n0 + (idx & (DZ_COLS - 1));
            const int hh = rh_head_of(a.hd, cc);
            wct[idx] = (col < a.N) ? a.hd.w[hh][(long)(cc - a.hd.row0[hh]) * a.N + col] : 0.0f;
        }
    }
    const long stride = (long)a.Mpad * a.N;
    float colsum = 0.0f;
    float wacc[DZ_WACC];
#pragma unroll
    for (int j = 0; j < DZ_WACC; ++j) wacc[j] = 0.0f;
    for (int m0 = 0; m0 < a.M; m0 += DZ_RC) {
        const int rows = min(DZ_RC, a.M - m0);
        if (MODE == 1) {
            __syncthreads();
            for (int idx = t; idx < rows * a.C; idx += DZ_THREADS) dsc[idx] = a.dS[(long)m0 * a.C + idx];
            for (int idx = t; idx < rows * DZ_COLS; idx += DZ_THREADS) {
                const int rr = idx / DZ_COLS, col = n0 + (idx & (DZ_COLS - 1));
                hc[idx] = (col < a.N) ? a.H[(long)(m0 + rr) * a.N + col] : 0.0f;
            }
            __syncthreads();
        }
        for (int rr = grp; rr < rows; rr += DZ_GROUPS) {
            const int m = m0 + rr;
            float v = 0.0f;
            if (n < a.N) {
                float h;
                if (MODE == 0) {
                    const float* sp = a.slabs + (long)m * a.N + n;
                    h = a.H[(long)m * a.N + n];
                    v = sp[0];
                    int s = 1;
                    for (; s + 8 <= a.S; s += 8) {
                        float u[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) u[e] = sp[(s + e) * stride];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v += u[e];
                    }
                    for (; s < a.S; ++s) v += sp[s * stride];
                } else {
                    for (int cc = 0; cc < a.C; ++cc) v += dsc[rr * a.C + cc] * wct[cc * DZ_COLS + c];
                    h = hc[rr * DZ_COLS + c];
                }
                v = (h > 0.0f) ? v : 0.0f;
                if (a.dZ) a.dZ[(long)m * a.N + n] = v;
            }
            tile[c * TLD + m] = v;
            colsum += v;
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < DZ_WACC; ++j) {
                const int idx = t + DZ_THREADS * j;
                if (idx < a.C * DZ_COLS) {
                    const int cc = idx / DZ_COLS, nn = idx & (DZ_COLS - 1);
                    float s = 0.0f;
                    for (int rr = 0; rr < rows; ++rr) s += dsc[rr * a.C + cc] * hc[rr * DZ_COLS + nn];
                    wacc[j] += s;
                }
            }
        }
    }
    part[grp * DZ_COLS + c] = colsum;
    __syncthreads();
    if (grp == 0 && n < a.N) {
        float s = part[c];
        for (int k = 1; k < DZ_GROUPS; ++k) s += part[k * DZ_COLS + c];
        a.db[n] = s;
    }
    for (int idx = t; idx < DZ_COLS * a.Mp; idx += DZ_THREADS) {
        const int cc = idx / a.Mp, m = idx - cc * a.Mp;
        if (n0 + cc < a.N) a.dZT[(long)(n0 + cc) * a.Mp + m] = (m < a.M) ? tile[cc * TLD + m] : 0.0f;
    }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < DZ_WACC; ++j) {
            const int idx = t + DZ_THREADS * j;
            if (idx < a.C * DZ_COLS) {
                const int cc = idx / DZ_COLS, col = n0 + (idx & (DZ_COLS - 1));
                const int hh = rh_head_of(a.hd, cc);
                if (col < a.N) a.hd.gw[hh][(long)(cc - a.hd.row0[hh]) * a.N + col] = wacc[j];
            }
        }
    }
}

// dW[No x Ni] = AT[No x Mp] X[M x Ni]  (AT = dZ^T, zero beyond column M; X rows clamped), 128 x 128 tiles over a job table
struct RhTnJob {
    const float* AT; const float* X; float* out;
    long ldx, ldo;
    int No, Ni, M, Mp, tiles_n, tile0;
};
struct RhTnArgs {
    RhTnJob job[3];
    int njobs;
};

template <bool FAST>
__global__ __launch_bounds__(256, 2) void rh_tn_kernel(const RhTnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ji = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (k < args.njobs && (int)blockIdx.x >= args.job[k].tile0) ji = k;
    const RhTnJob J = args.job[ji];
    const int tile = blockIdx.x - J.tile0;
    const int m0 = (tile / J.tiles_n) * BM, n0 = (tile % J.tiles_n) * BN;
    const int t = threadIdx.x;
    f32x16 acc[2][2];
    zero_acc(acc);
    float ra[2][4][4], rb[2][4][4];
    const float* pa[4];
    if (FAST) row_bases(J.AT, J.Mp, m0, pa);
    const gfloat* Xg = as_global(J.X);
    mfma_pipeline<false>(
        J.Mp / BK, smem, acc,
        [&](int kt, auto s) {
            constexpr int S = decltype(s)::value;
            if (FAST) load4(pa, (long)kt * BK, ra[S]);
            else stage_rows<false>(J.AT, J.Mp, J.No, J.Mp, m0, kt * BK, ra[S]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = min(kt * BK + (t >> 5) + 8 * j, J.M - 1), n = n0 + (t & 31) * 4;
                if (FAST) {
                    const f32x4 v = *(const gf32x4*)(J.X + (long)kr * J.ldx + n);
                    rb[S][j][0] = v[0]; rb[S][j][1] = v[1]; rb[S][j][2] = v[2]; rb[S][j][3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[S][j][e] = (n + e < J.Ni) ? Xg[(long)kr * J.ldx + n + e] : 0.0f;
                }
            }
        },
        [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value]); },
        [&](float* img, int, auto s) { write_kn(img, rb[decltype(s)::value]); });
    if (FAST) {
        acc_to_lds(smem, acc);
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's own LDS writes have landed
        for_each_row4(smem, [&](int r, int col, float4 v) {
            *(gf32x4*)(J.out + (long)(m0 + r) * J.ldo + n0 + col) = f32x4{v.x, v.y, v.z, v.w};
        });
    } else {
        const int lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + acc_row(r, lane);
                    if (row < J.No && col < J.Ni) as_global(J.out)[(long)row * J.ldo + col] = acc[mi][ni][r];
                }
            }
    }
}

// ---- host -----------------------------------------------------------------------
struct RhPlan {
    int MB, mchunks, ntn, S, per, Mpad, grid;
};

static RhPlan rh_plan(int M, int N, int K) {
    RhPlan p;
    const int nblocks = (M + 31) / 32;
    p.mchunks = (nblocks + RH_MAX_MB - 1) / RH_MAX_MB;
    p.MB = (nblocks + p.mchunks - 1) / p.mchunks;
    p.Mpad = p.mchunks * p.MB * 32;
    p.ntn = (N + 127) / 128;
    const int nk = (K + 31) / 32;
    int S = std::max(1, std::min(std::max(1, nk / 4), RH_TARGET_WGS / std::max(1, p.ntn * p.mchunks)));     // >= 4 k-steps per range
    p.per = (nk + S - 1) / S;
    p.S = (nk + p.per - 1) / p.per;
    p.grid = 8 * p.ntn * p.mchunks * ((p.S + 7) / 8);
    return p;
}

static size_t rh_pad(size_t x) { return (x + 255) & ~(size_t)255; }

struct RhWorkspace {
    size_t slabs, dS, rowloss, dZ2, dZ2T, dZ1T, total;
};

static RhWorkspace rh_workspace(int M, int in_f, int hidden, int C) {
    const RhPlan f1 = rh_plan(M, hidden, in_f), f2 = rh_plan(M, hidden, hidden);
    const size_t slab_f = std::max((size_t)f1.S * f1.Mpad, (size_t)f2.S * f2.Mpad) * hidden * 4;
    const int Mp = (M + 31) / 32 * 32;
    RhWorkspace w;
    size_t off = 0;
    w.slabs = off; off += rh_pad(slab_f);
    w.dS = off; off += rh_pad((size_t)M * C * 4);
    w.rowloss = off; off += rh_pad((size_t)M * 4);
    w.dZ2 = off; off += rh_pad((size_t)M * hidden * 4);
    w.dZ2T = off; off += rh_pad((size_t)hidden * Mp * 4);
    w.dZ1T = off; off += rh_pad((size_t)hidden * Mp * 4);
    w.total = off;
    return w;
}

template <int MB, bool B_ROWS>
static int rh_launch_skinny_mb(const RhGemm& g, const RhPlan& p, bool fast, hipStream_t stream) {
    const size_t smem = (size_t)rh_smem_floats<MB>() * 4;
    static bool armed[2] = {false, false};       // the LDS opt-in is per kernel and sticks: once per process and instantiation
    if (fast) {
        if (!armed[1]) {
            NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_skinny_kernel<MB, B_ROWS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            armed[1] = true;
        }
        hipLaunchKernelGGL((rh_skinny_kernel<MB, B_ROWS, true>), dim3(p.grid), dim3(256), smem, stream, g);
    } else {
        if (!armed[0]) {
            NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_skinny_kernel<MB, B_ROWS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            armed[0] = true;
        }
        hipLaunchKernelGGL((rh_skinny_kernel<MB, B_ROWS, false>), dim3(p.grid), dim3(256), smem, stream, g);
    }
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

// C slabs = A[M x K] (B^T | B);  B_ROWS: B is [N x K], else [K x N]
template <bool B_ROWS>
static int rh_launch_skinny(const float* A, long lda, const float* B, long ldb, float* slabs, int M, int N, int K, RhPlan& p, hipStream_t stream) {
    p = rh_plan(M, N, K);
    RhGemm g{A, lda, B, ldb, slabs, M, N, K, p.mchunks, p.ntn, p.S, p.per, p.Mpad};
    const bool fast = (K % 32 == 0) && (N % 128 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A) && aligned16(B);
    switch (p.MB) {
        case 1: return rh_launch_skinny_mb<1, B_ROWS>(g, p, fast, stream);
        case 2: return rh_launch_skinny_mb<2, B_ROWS>(g, p, fast, stream);
        case 3: return rh_launch_skinny_mb<3, B_ROWS>(g, p, fast, stream);
        case 4: return rh_launch_skinny_mb<4, B_ROWS>(g, p, fast, stream);
        default: return rh_launch_skinny_mb<5, B_ROWS>(g, p, fast, stream);
    }
}

static int rh_launch_reduce(const float* slabs, const RhPlan& p, int M, int N, const float* bias, float* out, hipStream_t stream) {
    if (N % 4 == 0 && aligned16(out) && aligned16(bias)) {
        const long n4 = ((long)M * N + 3) / 4;
        hipLaunchKernelGGL(rh_reduce_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, slabs, p.S, p.Mpad, M, N, bias, out);
    } else {
        hipLaunchKernelGGL(rh_reduce_kernel<false>, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, stream, slabs, p.S, p.Mpad, M, N, bias, out);
    }
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

static int rh_check_dims(const char* who, int n_rows, int in_f, int hidden, int n_cols) {
    if (n_rows <= 0 || in_f <= 0 || hidden <= 0 || n_cols <= 0) return fail(NSGP_ERR_INVALID, "%s: non-positive size", who);
    if (n_rows > RH_MAX_ROWS) return fail(NSGP_ERR_LIMIT, "%s: %d bank rows > %d", who, n_rows, RH_MAX_ROWS);
    if (n_cols > RH_MAX_COLS) return fail(NSGP_ERR_LIMIT, "%s: %d class columns > %d", who, n_cols, RH_MAX_COLS);
    return NSGP_OK;
}

}  // namespace nsgp

using namespace nsgp;

extern "C" size_t repre_replay_head_workspace_bytes(int n_rows, int in_features, int hidden, int n_cols) {
    if (n_rows <= 0 || in_features <= 0 || hidden <= 0 || n_cols <= 0 || n_rows > RH_MAX_ROWS || n_cols > RH_MAX_COLS) return 0;
    return rh_workspace(n_rows, in_features, hidden, n_cols).total;
}

static int rh_make_heads(const char* who, RhHeads& hd, const float* const* w, const float* const* b, float* const* gw, float* const* gb,
                         const int* rows, int n_heads, int n_cols) {
    if (!w || !b || !rows || n_heads <= 0 || n_heads > RH_MAX_HEADS) return fail(NSGP_ERR_INVALID, "%s: 1..%d class heads expected", who, RH_MAX_HEADS);
    hd.n = n_heads;
    hd.row0[0] = 0;
    for (int h = 0; h < n_heads; ++h) {
        if (!w[h] || !b[h] || rows[h] <= 0 || (gw && (!gw[h] || !gb[h]))) return fail(NSGP_ERR_INVALID, "%s: class head %d: null pointer or no rows", who, h);
        hd.w[h] = w[h];
        hd.b[h] = b[h];
        hd.gw[h] = gw ? gw[h] : nullptr;
        hd.gb[h] = gb ? gb[h] : nullptr;
        hd.row0[h + 1] = hd.row0[h] + rows[h];
    }
    if (hd.row0[n_heads] != n_cols) return fail(NSGP_ERR_INVALID, "%s: the heads hold %d rows, n_cols = %d", who, hd.row0[n_heads], n_cols);
    return NSGP_OK;
}

extern "C" int repre_replay_head_forward(const float* bank, int n_rows, int in_features, const float* w1, const float* b1, const float* w2,
                                         const float* b2, const float* const* wc_heads, const float* const* bc_heads, const int* head_rows,
                                         int n_heads, int hidden, int n_cols, const int64_t* labels,
                                         float* h1, float* h2, float* scores, float* loss_out, void* workspace, size_t workspace_bytes,
                                         void* stream_) {
    int rc = rh_check_dims("repre_replay_head_forward", n_rows, in_features, hidden, n_cols);
    if (rc) return rc;
    if (!bank || !w1 || !b1 || !w2 || !b2 || !labels || !h1 || !h2 || !scores || !loss_out || !workspace)
        return fail(NSGP_ERR_INVALID, "repre_replay_head_forward: null argument");
    RhHeads hd{};
    if ((rc = rh_make_heads("repre_replay_head_forward", hd, wc_heads, bc_heads, nullptr, nullptr, head_rows, n_heads, n_cols))) return rc;
    const RhWorkspace W = rh_workspace(n_rows, in_features, hidden, n_cols);
    if (workspace_bytes < W.total) return fail(NSGP_ERR_WORKSPACE, "repre_replay_head_forward: workspace %zu < %zu bytes", workspace_bytes, W.total);
    if (!aligned16(workspace)) return fail(NSGP_ERR_INVALID, "repre_replay_head_forward: workspace must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + W.slabs);
    RhPlan p;
    if ((rc = rh_launch_skinny<true>(bank, in_features, w1, in_features, slabs, n_rows, hidden, in_features, p, stream))) return rc;
    if ((rc = rh_launch_reduce(slabs, p, n_rows, hidden, b1, h1, stream))) return rc;
    if ((rc = rh_launch_skinny<true>(h1, hidden, w2, hidden, slabs, n_rows, hidden, hidden, p, stream))) return rc;
    if ((rc = rh_launch_reduce(slabs, p, n_rows, hidden, b2, h2, stream))) return rc;
    float* rowloss = reinterpret_cast<float*>(static_cast<char*>(workspace) + W.rowloss);
    hipLaunchKernelGGL(rh_scores_kernel, dim3(n_rows), dim3(256), 0, stream, h2, n_rows, hidden, hd, n_cols,
                       reinterpret_cast<const long long*>(labels), scores, rowloss);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(rh_mean_kernel, dim3(1), dim3(256), 0, stream, rowloss, n_rows, loss_out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int repre_replay_head_backward(const float* bank, int n_rows, int in_features, const float* w2, const float* const* wc_heads,
                                          const float* const* bc_heads, const int* head_rows, int n_heads, int hidden,
                                          int n_cols, const int64_t* labels, const float* h1, const float* h2, const float* scores,
                                          const float* grad_out, float* gw1, float* gb1, float* gw2, float* gb2, float* const* gwc_heads,
                                          float* const* gbc_heads,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = rh_check_dims("repre_replay_head_backward", n_rows, in_features, hidden, n_cols);
    if (rc) return rc;
    if (!bank || !w2 || !labels || !h1 || !h2 || !scores || !grad_out || !gw1 || !gb1 || !gw2 || !gb2 || !gwc_heads || !gbc_heads || !workspace)
        return fail(NSGP_ERR_INVALID, "repre_replay_head_backward: null argument");
    RhHeads hd{};
    if ((rc = rh_make_heads("repre_replay_head_backward", hd, wc_heads, bc_heads, gwc_heads, gbc_heads, head_rows, n_heads, n_cols))) return rc;
    const RhWorkspace W = rh_workspace(n_rows, in_features, hidden, n_cols);
    if (workspace_bytes < W.total) return fail(NSGP_ERR_WORKSPACE, "repre_replay_head_backward: workspace %zu < %zu bytes", workspace_bytes, W.total);
    if (!aligned16(workspace)) return fail(NSGP_ERR_INVALID, "repre_replay_head_backward: workspace must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    float* slabs = reinterpret_cast<float*>(ws + W.slabs);
    float* dS = reinterpret_cast<float*>(ws + W.dS);
    float* dZ2 = reinterpret_cast<float*>(ws + W.dZ2);
    float* dZ2T = reinterpret_cast<float*>(ws + W.dZ2T);
    float* dZ1T = reinterpret_cast<float*>(ws + W.dZ1T);
    const int M = n_rows, Mp = (M + 31) / 32 * 32;
    if ((rc = repre_replay_ce_backward(scores, labels, M, n_cols, grad_out, dS, stream_))) return rc;
    const int nblocks = (hidden + DZ_COLS - 1) / DZ_COLS;
    const size_t dz_lds = ((size_t)DZ_COLS * (Mp + 1) + DZ_GROUPS * DZ_COLS + (size_t)n_cols * DZ_COLS + (size_t)DZ_RC * n_cols + DZ_RC * DZ_COLS) * 4;
    static size_t dz_armed = 0;                  // the LDS opt-in sticks: raise it only when a larger tile comes along
    if (dz_lds > dz_armed) {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_dz_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dz_lds));
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_dz_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dz_lds));
        dz_armed = dz_lds;
    }
    {   // dZ2 = (dS Wc) * (H2 > 0), its transposed copy, db2; the class heads' dWc = dS^T H2 and dbc
        RhDz a{nullptr, 0, 0, dS, n_cols, h2, M, hidden, Mp, dZ2, dZ2T, gb2, hd};
        hipLaunchKernelGGL(rh_dz_kernel<1>, dim3(nblocks + 1), dim3(DZ_THREADS), dz_lds, stream, a);
        NSGP_LAUNCH_CHECK();
    }
    RhPlan p;    // dH1 = dZ2 W2   (W2 is [hidden(k) x hidden(n)] row-major for this product)
    if ((rc = rh_launch_skinny<false>(dZ2, hidden, w2, hidden, slabs, M, hidden, hidden, p, stream))) return rc;
    {   // dZ1 = dH1 * (H1 > 0): only its transposed copy is needed (dX is not: the bank is a constant), db1
        RhDz a{slabs, p.S, p.Mpad, nullptr, 0, h1, M, hidden, Mp, nullptr, dZ1T, gb1, RhHeads{}};
        hipLaunchKernelGGL(rh_dz_kernel<0>, dim3(nblocks), dim3(DZ_THREADS), dz_lds, stream, a);
        NSGP_LAUNCH_CHECK();
    }
    // weight gradients of the two shared FCs: dW1 = dZ1^T X, dW2 = dZ2^T H1 (one grouped launch when both take the whole-tile path)
    auto job = [&](const float* AT, const float* X, long ldx, float* out, int No, int Ni) {
        RhTnJob j{AT, X, out, ldx, (long)Ni, No, Ni, M, Mp, (Ni + BN - 1) / BN, 0};
        return j;
    };
    auto fast = [&](const RhTnJob& j) { return j.No % BM == 0 && j.Ni % BN == 0 && aligned16(j.AT) && aligned16(j.X) && aligned16(j.out) && j.ldx % 4 == 0; };
    RhTnJob jobs[2] = {job(dZ1T, bank, in_features, gw1, hidden, in_features), job(dZ2T, h1, hidden, gw2, hidden, hidden)};
    static bool tn_armed = false;
    if (!tn_armed) {
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rh_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        tn_armed = true;
    }
    for (int pass = 0; pass < 2; ++pass) {
        RhTnArgs args{};
        int tiles = 0;
        for (const RhTnJob& j0 : jobs) {
            if (fast(j0) != (pass == 0)) continue;
            RhTnJob j = j0;
            j.tile0 = tiles;
            tiles += ((j.No + BM - 1) / BM) * j.tiles_n;
            args.job[args.njobs++] = j;
        }
        if (!args.njobs) continue;
        if (pass == 0) hipLaunchKernelGGL(rh_tn_kernel<true>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, args);
        else hipLaunchKernelGGL(rh_tn_kernel<false>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, args);
        NSGP_LAUNCH_CHECK();
    }
    return NSGP_OK;
}
