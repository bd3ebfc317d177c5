// Projected optimizer step: K1 (grouped fp32-MFMA projection GEMM) + K2 (multi-tensor
// elementwise update).  Replaces the Python per-parameter loop of the reference's
// SGDNSCL / AdamWNSCL / AdamNSCL / SGDNSCLNA `.step()`
// (mmdet/engine/optimizers/SGD_NSCL.py:59-96,387-415; AdamW_NSCL.py:66-103,212-250;
// Adam_NSCL.py:66-102,207-247).
//
// Launch 1  nsgp_update_kernel    HBM-bound. One pass over every listed tensor:
//           weight decay, momentum / Adam moments, and `p += update` for tensors
//           without a projector.  Projected tensors keep their update source for
//           launch 2 (SGD: the momentum buffer or the mutated grad; Adam: workspace U).
//           On the fp16-split path (the default) a workgroup of this launch owns 8 WHOLE ROWS of a
//           projected tensor: it finds each row's largest |update|, and writes the update once more as
//           the pre-tiled, row-scaled two-term fp16 split the projection streams (gemm_f16x2_v2.hpp).
// Launch 2  nsgp_project_v2_kernel / nsgp_project_kernel   MFMA-bound. For every projected tensor
//           p[Cout x D] += (scale * S[Cout x D]) @ P[D x D]  as 256x128 (fp16 split) or 128x128
//           (fp32 MFMA, bf16 split) output tiles drawn from ONE cost-sorted tile table spanning all
//           layers (longest K first = LPT list scheduling on the 256 CUs), interleaved so that tiles
//           that share a P column panel land on the same XCD (blockIdx % 8).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "common.hpp"
#include "gemm_core.hpp"
#include "gemm_bf16x3.hpp"
#include "gemm_f16x2.hpp"
#include "gemm_f16x2_v2.hpp"

namespace nsgp {

// ---- device-side tables ------------------------------------------------------
struct TensorDev {
    float* p;
    float* s0;
    float* s1;
    float* s2;
    float* u;  // Adam: slice of the update workspace (projected tensors only)
    long numel;
    int hyper;
    int projected;
    // fp16-split path: the update of this tensor is also written as its pre-tiled row-scaled split (workspace), with 1/scale per row
    void* a_split;
    float* rinv;
    int cols;
    int pad;
};

struct LayerDev {
    float* p;
    const float* s0;
    const float* u;
    const float* proj;
    int tensor;
    int rows, cols;
    int hyper;
    // low-rank form (rank > 0): V, r, T = (scale*S) U  [rows x rpad], its split-K slabs, 1/||P||_F
    const float* basis;
    float* T;
    float* slabs;
    int rank, rpad, nsplit, kchunk;
    float basis_scale;
    const void* split;     // split copy of proj^T: kind 1 = three bf16 terms (gemm_bf16x3.hpp), 2 = pre-tiled column-scaled fp16 pair (gemm_f16x2_v2.hpp)
    int split_kind;
    int pad;
    const float* cinv;     // kind 2: 1 / (power-of-two scale of each projector column), D floats behind the split copy
    const void* a_split;   // kind 2: this step's update, pre-tiled row-scaled fp16 pair (written by the elementwise launch)
    const float* rinv;     // kind 2: 1 / (scale of each update row)
};

struct TileDev {
    int layer, m0, n0, pad;  // pad: split-K slice index (low-rank phase 1)
};

constexpr int LR_REDUCE_CHUNK = 2048;   // elements per workgroup of the slab reduce (many small blocks: T is small)
constexpr int LR_KCHUNK_DEFAULT = 512;  // K extent of one low-rank phase-1 tile (the plan may pick another)

struct ChunkDev {
    int tensor;
    int band;    // 1: rows [start / cols, +8) of a projected tensor on the fp16-split path; 0: a linear chunk of CHUNK elements
    long start;
};

constexpr int CHUNK = 16384;  // elements per workgroup of the elementwise kernel
constexpr int NSLOT = 4;      // depth of the per-step upload ring

struct DynBlock {  // uploaded every step: hyper sets + current grad pointers
    nsgp_hyper_t hyper[NSGP_MAX_HYPER];
    float* grads[1];  // n_tensors entries follow
};

// ---- launch 1: multi-tensor elementwise update -------------------------------
// Rounding mirrors the ATen CPU kernels the reference runs: `x.add_(alpha, y)` is one
// fused multiply-add per element, `mul_` then `add_` are two roundings.  The file is
// built with -ffp-contract=off so nothing else is contracted.

__device__ __forceinline__ void sgd_elem(float& p, float& g, float& b, const nsgp_hyper_t& h, bool projected) {
    if (h.weight_decay != 0.0f) g = fmaf(h.weight_decay, p, g);  // grad.add_(wd, p)            :400
    float d = g;
    if (h.momentum != 0.0f) {
        if (h.first_step) b = b + g;                                // exp_avg.add_(grad)          :406
        else b = fmaf(h.one_minus_dampening, g, h.momentum * b);    // mul_(m).add_(1-damp, grad)  :404
        if (h.nesterov) { g = fmaf(h.momentum, b, g); d = g; }      // grad.add_(m, exp_avg)       :409
        else d = b;                                                 // grad = exp_avg              :411
    }
    if (!projected) p = p + (-(h.lr * d));                          // p.add_(-(lr*grad))          :413-414,95
}

__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, float& vmax, float& u,
                                          const nsgp_hyper_t& h, bool projected) {
    if (h.weight_decay != 0.0f) g = fmaf(h.weight_decay, p, g);    // Adam_NSCL.py:229-230
    m = fmaf(h.one_minus_beta1, g, h.beta1 * m);                    // exp_avg.mul_(b1).add_(1-b1, g)
    v = fmaf(h.one_minus_beta2 * g, g, h.beta2 * v);                // exp_avg_sq.mul_(b2).addcmul_(1-b2, g, g)
    float denom;
    if (h.amsgrad) { vmax = fmaxf(vmax, v); denom = sqrtf(vmax) + h.eps; }
    else denom = sqrtf(v) + h.eps;
    float upd = ((-h.step_size) * m) / denom;                       // -step_size * exp_avg / denom
    if (h.decoupled_decay != 0.0f) upd = upd - h.decoupled_decay * p;  // AdamW_NSCL.py:87
    if (projected) u = upd; else p = p + upd;
}

// Gradients may be views into a flat bucket (DDP's gradient_as_bucket_view: parameters packed back to back, so one odd-sized
// bias shifts everything behind it off the 16-byte grid).  Parameters and optimizer state are whole allocations and stay
// aligned, so only the gradient stream uses 4-byte-aligned 16-byte accesses; the kernel keeps its float4 shape.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) f32x4_a4 gf32x4_a4;

// Row bookkeeping of a band workgroup: a thread walks its elements in steps of `stride` and needs the band-local row of
// each (cols % 4 == 0, so a float4 never straddles rows).  One division per thread up front, increments afterwards.
struct RowWalk {
    int row, k, q, rm, cols;
    __device__ __forceinline__ void init(long local, int stride, int cols_) {
        cols = cols_;
        row = (int)(local / cols_);
        k = (int)(local - (long)row * cols_);
        q = stride / cols_;
        rm = stride - q * cols_;
    }
    __device__ __forceinline__ void next() {
        row += q;
        k += rm;
        if (k >= cols) { k -= cols; ++row; }
    }
};

template <int OPT>
__global__ __launch_bounds__(256) void nsgp_update_kernel(const ChunkDev* __restrict__ chunks,
                                                          const TensorDev* __restrict__ tensors,
                                                          const DynBlock* __restrict__ dyn) {
    __shared__ unsigned s_rowmax[8];
    __shared__ float s_rowscale[8];
    const ChunkDev c = chunks[blockIdx.x];
    const TensorDev T = tensors[c.tensor];
    const nsgp_hyper_t h = dyn->hyper[T.hyper];
    float* __restrict__ gp = dyn->grads[c.tensor];
    const bool band = c.band != 0;                       // uniform per workgroup
    const long span = band ? 8L * T.cols : (long)CHUNK;
    const long end = (c.start + span < T.numel) ? c.start + span : T.numel;
    const bool proj = T.projected != 0;
    const bool vec = (((uintptr_t)T.p | ((uintptr_t)gp & 3u) | (uintptr_t)T.s0 | (uintptr_t)T.s1 | (uintptr_t)T.s2 |
                       (uintptr_t)T.u) & 15u) == 0;
    // the mutated gradient is the GEMM's A operand unless a non-Nesterov momentum buffer is
    bool wg = h.write_grad != 0;
    if (OPT == NSGP_OPT_SGD && proj && !(h.momentum != 0.0f && !h.nesterov)) wg = true;
    // what the projection reads as its A operand: momentum buffer, mutated gradient, or Adam's update
    const bool a_is_buf = (OPT == NSGP_OPT_SGD) && h.momentum != 0.0f && !h.nesterov;
    if (band) {
        if (threadIdx.x < 8) s_rowmax[threadIdx.x] = 0u;
        __syncthreads();
    }
    // band workgroups keep the largest |A| of the row they are in and hand it to LDS whenever the row changes
    RowWalk rw;
    float am = 0.0f;
    int cur_row = -1;
    auto note = [&](float a0, float a1, float a2, float a3) {
        if (rw.row != cur_row) {
            if (cur_row >= 0) atomicMax(&s_rowmax[cur_row], __float_as_uint(am));
            cur_row = rw.row;
            am = 0.0f;
        }
        am = fmaxf(fmaxf(am, fmaxf(fabsf(a0), fabsf(a1))), fmaxf(fabsf(a2), fabsf(a3)));
    };
    long i = c.start + (long)threadIdx.x * 4;
    if (vec) {
        if (band) rw.init((long)threadIdx.x * 4, 256 * 4, T.cols);
        for (; i + 3 < end; i += 256 * 4) {
            float4 p4 = *reinterpret_cast<const float4*>(T.p + i);
            const f32x4_a4 g4 = *(const gf32x4_a4*)(gp + i);
            float pv[4] = {p4.x, p4.y, p4.z, p4.w}, gv[4] = {g4[0], g4[1], g4[2], g4[3]};
            if (OPT == NSGP_OPT_SGD) {
                float bv[4] = {0, 0, 0, 0};
                if (h.momentum != 0.0f) {
                    float4 b4 = *reinterpret_cast<const float4*>(T.s0 + i);
                    bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) sgd_elem(pv[e], gv[e], bv[e], h, proj);
                if (band) {
                    if (a_is_buf) note(bv[0], bv[1], bv[2], bv[3]); else note(gv[0], gv[1], gv[2], gv[3]);
                    rw.next();
                }
                if (h.momentum != 0.0f) *reinterpret_cast<float4*>(T.s0 + i) = make_float4(bv[0], bv[1], bv[2], bv[3]);
                if (wg && (h.weight_decay != 0.0f || (h.momentum != 0.0f && h.nesterov)))
                    *(gf32x4_a4*)(gp + i) = f32x4_a4{gv[0], gv[1], gv[2], gv[3]};
                if (!proj) *reinterpret_cast<float4*>(T.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            } else {
                float4 m4 = *reinterpret_cast<const float4*>(T.s0 + i);
                float4 v4 = *reinterpret_cast<const float4*>(T.s1 + i);
                float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
                float xv[4] = {0, 0, 0, 0}, uv[4];
                if (h.amsgrad) {
                    float4 x4 = *reinterpret_cast<const float4*>(T.s2 + i);
                    xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) adam_elem(pv[e], gv[e], mv[e], vv[e], xv[e], uv[e], h, proj);
                if (band) { note(uv[0], uv[1], uv[2], uv[3]); rw.next(); }
                *reinterpret_cast<float4*>(T.s0 + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                *reinterpret_cast<float4*>(T.s1 + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                if (h.amsgrad) *reinterpret_cast<float4*>(T.s2 + i) = make_float4(xv[0], xv[1], xv[2], xv[3]);
                if (wg && h.weight_decay != 0.0f) *(gf32x4_a4*)(gp + i) = f32x4_a4{gv[0], gv[1], gv[2], gv[3]};
                if (proj) *reinterpret_cast<float4*>(T.u + i) = make_float4(uv[0], uv[1], uv[2], uv[3]);
                else *reinterpret_cast<float4*>(T.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            }
        }
        // scalar tail of a tensor whose numel is not a multiple of 4 (never a band: cols % 128 == 0 there)
        const long tail0 = end - ((end - c.start) & 3);
        i = tail0 + threadIdx.x;
    } else {
        i = c.start + threadIdx.x;
        if (band) rw.init((long)threadIdx.x, 256, T.cols);
    }
    const long stride = vec ? end : 256;  // vec: at most 3 tail elements, one per thread
    for (; i < end; i += stride) {
        float pv = T.p[i], gv = gp[i];
        if (OPT == NSGP_OPT_SGD) {
            float bv = (h.momentum != 0.0f) ? T.s0[i] : 0.0f;
            sgd_elem(pv, gv, bv, h, proj);
            if (band && !vec) { const float a = a_is_buf ? bv : gv; note(a, a, a, a); rw.next(); }
            if (h.momentum != 0.0f) T.s0[i] = bv;
            if (wg && (h.weight_decay != 0.0f || (h.momentum != 0.0f && h.nesterov))) gp[i] = gv;
            if (!proj) T.p[i] = pv;
        } else {
            float mv = T.s0[i], vv = T.s1[i], xv = h.amsgrad ? T.s2[i] : 0.0f, uv;
            adam_elem(pv, gv, mv, vv, xv, uv, h, proj);
            if (band && !vec) { note(uv, uv, uv, uv); rw.next(); }
            T.s0[i] = mv; T.s1[i] = vv;
            if (h.amsgrad) T.s2[i] = xv;
            if (wg && h.weight_decay != 0.0f) gp[i] = gv;
            if (proj) T.u[i] = uv; else T.p[i] = pv;
        }
    }
    if (!band) return;
    // ---- band epilogue: row scales, then the update of these 8 rows once more as its pre-tiled two-term fp16 split.
    // The A values were written to global memory just above by this workgroup (momentum buffer / mutated gradient / u):
    // __syncthreads() makes them visible to the whole workgroup and they are still in this CU's cache hierarchy.
    if (cur_row >= 0) atomicMax(&s_rowmax[cur_row], __float_as_uint(am));
    __syncthreads();
    const int row0 = (int)(c.start / T.cols);
    if (threadIdx.x < 8) {
        const float sc = f2_scale_from_amax_bits(s_rowmax[threadIdx.x]);
        s_rowscale[threadIdx.x] = sc;
        T.rinv[row0 + threadIdx.x] = 1.0f / sc;
    }
    __syncthreads();
    const float* __restrict__ Asrc = (OPT == NSGP_OPT_SGD) ? (a_is_buf ? T.s0 : gp) : T.u;
    const bool a_vec = ((uintptr_t)Asrc & 3u) == 0;     // always, for fp32 data: 4-byte-aligned 16-byte loads
    for (int id = threadIdx.x; id < T.cols; id += 256) {      // 8 rows x cols / 8 octets; 8 consecutive lanes = one full 128-B line per plane
        const int r = id & 7, o = id >> 3;
        const float* src = Asrc + c.start + (long)r * T.cols + o * 8;
        f32x4 lo, hi;
        if (a_vec) {
            const f32x4_a4 l4 = *(const gf32x4_a4*)src, h4 = *(const gf32x4_a4*)(src + 4);
            lo = f32x4{l4[0], l4[1], l4[2], l4[3]};
            hi = f32x4{h4[0], h4[1], h4[2], h4[3]};
        } else {
            const gfloat* g = as_global(src);
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = g[e]; hi[e] = g[4 + e]; }
        }
        v2_store_pieces(T.a_split, row0 + r, o, T.cols, lo, hi, s_rowscale[r]);
    }
}

// ---- launch 2: grouped projection GEMM ---------------------------------------
template <bool FAST, bool ACCUM>
__device__ __forceinline__ void store_tile(float* __restrict__ C, long ldc, int M, int N, int m0, int n0,
                                           const f32x16 (&acc)[2][2], float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + acc_row(r, lane);
                if (FAST || (row < M && col < N)) {
                    gfloat* dst = as_global(C) + (long)row * ldc + col;
                    *dst = ACCUM ? (*dst + scale * acc[mi][ni][r]) : scale * acc[mi][ni][r];
                }
            }
        }
}

template <int OPT, bool FAST, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void nsgp_project_kernel(const TileDev* __restrict__ tiles,
                                                              const LayerDev* __restrict__ layers,
                                                              const DynBlock* __restrict__ dyn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileDev t = tiles[blockIdx.x];
    const LayerDev L = layers[t.layer];
    const float* A;
    float scale;
    if (OPT == NSGP_OPT_SGD) {
        const nsgp_hyper_t& h = dyn->hyper[L.hyper];
        // update = -(lr * grad) where `grad` is the momentum buffer (non-Nesterov momentum)
        // or the (mutated) gradient itself (SGD_NSCL.py:408-414)
        A = (h.momentum != 0.0f && !h.nesterov) ? L.s0 : dyn->grads[L.tensor];
        scale = -h.lr;
    } else {
        A = L.u;
        scale = 1.0f;
    }
    f32x16 acc[2][2];
    zero_acc(acc);
    // the grad pointer is only known at step time: a misaligned one (e.g. a view into a flat
    // bucket) takes the guarded scalar loader for the A operand only
    if (SPLIT == 1 && ((uintptr_t)A & 15u) == 0)                 // six bf16 MFMAs per fp32-equivalent product (gemm_bf16x3.hpp)
        gemm_tile_bf16x3(A, L.cols, static_cast<const __bf16*>(L.split), L.cols, t.m0, t.n0, smem, acc);
    else if (!FAST || ((uintptr_t)A & 15u) == 0)
        gemm_tile<FAST, FAST, false>(A, L.cols, L.proj, L.cols, L.rows, L.cols, L.cols, t.m0, t.n0, smem, acc);
    else
        gemm_tile<false, true, false>(A, L.cols, L.proj, L.cols, L.rows, L.cols, L.cols, t.m0, t.n0, smem, acc);
    // p.data.add_(update_) :95 with update_ = (-(lr*S)) @ P; the scalar is applied once per output
    // element here instead of once per staged operand element (differs by one fp32 rounding per term)
    if (FAST) {   // full tile, 16-byte aligned p: re-layout through LDS and update p with float4 read-modify-writes
        acc_to_lds(smem, acc);
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's own LDS writes have landed
        for_each_row4(smem, [&](int r, int col, float4 v) {
            gf32x4* pp = (gf32x4*)(L.p + (long)(t.m0 + r) * L.cols + t.n0 + col);
            f32x4 pv = *pp;
            pv[0] = pv[0] + scale * v.x;
            pv[1] = pv[1] + scale * v.y;
            pv[2] = pv[2] + scale * v.z;
            pv[3] = pv[3] + scale * v.w;
            *pp = pv;
        });
    } else {
        store_tile<FAST, true>(L.p, L.cols, L.rows, L.cols, t.m0, t.n0, acc, scale);
    }
}

// fp16-split path (gemm_f16x2_v2.hpp): p += scale * rinv[m] * cinv[n] * (A_split x B_split), 256 x 128 tiles (t.pad = 4 row
// blocks) or 128 x 128 (t.pad = 2); 768 threads = 8 consumer + 4 loader waves, one workgroup per CU.
template <int OPT>
__global__ __launch_bounds__(V2L_THREADS, 3) void nsgp_project_v2_kernel(const TileDev* __restrict__ tiles,
                                                                        const LayerDev* __restrict__ layers,
                                                                        const DynBlock* __restrict__ dyn) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const TileDev t = tiles[blockIdx.x];
    const LayerDev L = layers[t.layer];
    const float scale = (OPT == NSGP_OPT_SGD) ? -dyn->hyper[L.hyper].lr : 1.0f;   // update = -(lr * grad), SGD_NSCL.py:413
    f32x16 acc[2][2];
    zero_acc(acc);
    if (t.pad == 4) gemm_tile_f16x2_v2l<4>(L.a_split, t.m0 >> 6, L.split, t.n0 >> 6, L.cols, smem_c, acc);
    else gemm_tile_f16x2_v2l<2>(L.a_split, t.m0 >> 6, L.split, t.n0 >> 6, L.cols, smem_c, acc);
    if ((int)(threadIdx.x >> 6) >= 2 * t.pad) return;      // the four loader waves; and waves 4-7 of a 128-row tile, which have no rows
    float* smem = reinterpret_cast<float*>(smem_c);
    acc_to_lds(smem, acc);
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's own LDS writes have landed
    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
    const f32x4 ci = *(const gf32x4*)(L.cinv + t.n0 + wn * 64 + 4 * (lane & 15));    // this lane's four columns, every pass
    for_each_row4(smem, [&](int r, int col, float4 v) {
        const float ri = L.rinv[t.m0 + r];
        gf32x4* pp = (gf32x4*)(L.p + (long)(t.m0 + r) * L.cols + t.n0 + col);
        f32x4 pv = *pp;
        pv[0] = pv[0] + scale * (ri * (ci[0] * v.x));      // ri, ci: powers of two -- exact
        pv[1] = pv[1] + scale * (ri * (ci[1] * v.y));
        pv[2] = pv[2] + scale * (ri * (ci[2] * v.z));
        pv[3] = pv[3] + scale * (ri * (ci[3] * v.w));
        *pp = pv;
    });
}

template <bool FAST>
__global__ __launch_bounds__(256, 2) void nsgp_project_single_kernel(const float* __restrict__ A,
                                                                     const float* __restrict__ P,
                                                                     float* __restrict__ out, int rows, int cols,
                                                                     float scale, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 acc[2][2];
    zero_acc(acc);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    gemm_tile<FAST, FAST, false>(A, cols, P, cols, rows, cols, cols, m0, n0, smem, acc);
    if (accumulate) store_tile<FAST, true>(out, cols, rows, cols, m0, n0, acc, scale);
    else store_tile<FAST, false>(out, cols, rows, cols, m0, n0, acc, scale);
}

// ---- low-rank form:  p += c * (u - (u U) U^T),  u = scale*S,  U = V[:, :r]  ------------------------
// Same result as u @ (c * V_tail V_tail^T) up to the orthogonality error of V, in 4*Cout*D*r FLOP.
// Phase 1: T = u U as split-K slabs (K chunks of 512 so that the skinny [Cout x r] product still fills
// the chip); a deterministic reduce sums the slabs; phase 2: p += c*(u - T U^T), K = r.
template <int OPT>
__device__ __forceinline__ void lowrank_source(const LayerDev& L, const DynBlock* dyn, const float*& A, float& scale) {
    if (OPT == NSGP_OPT_SGD) {
        const nsgp_hyper_t& h = dyn->hyper[L.hyper];
        A = (h.momentum != 0.0f && !h.nesterov) ? L.s0 : dyn->grads[L.tensor];
        scale = -h.lr;
    } else {
        A = L.u;
        scale = 1.0f;
    }
}

template <int OPT>
__global__ __launch_bounds__(256, 2) void nsgp_lowrank_p1_kernel(const TileDev* __restrict__ tiles,
                                                                 const LayerDev* __restrict__ layers,
                                                                 const DynBlock* __restrict__ dyn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileDev t = tiles[blockIdx.x];
    const LayerDev L = layers[t.layer];
    const float* A;
    float scale;
    lowrank_source<OPT>(L, dyn, A, scale);
    const int k0 = t.pad * L.kchunk;
    const int k1 = min(k0 + L.kchunk, L.cols);
    f32x16 acc[2][2];
    zero_acc(acc);
    float ra[2][4][4], rb[2][4][4];
    const bool a_fast = ((uintptr_t)A & 15u) == 0;
    mfma_pipeline<false>(
        (k1 - k0) / BK, smem, acc,
        [&](int kt, auto s) {
            constexpr int S = decltype(s)::value;
            if (a_fast) stage_rows<true>(A, L.cols, L.rows, L.cols, t.m0, k0 + kt * BK, ra[S]);
            else stage_rows<false>(A, L.cols, L.rows, L.cols, t.m0, k0 + kt * BK, ra[S]);
            stage_kn<true>(L.basis, L.cols, L.cols, L.cols, k0 + kt * BK, t.n0, rb[S]);
        },
        [&](float* img, int, auto s) { write_rows(img, ra[decltype(s)::value]); },
        [&](float* img, int, auto s) { write_kn(img, rb[decltype(s)::value]); });
    float* slab = L.slabs + (long)t.pad * L.rows * L.rpad;
    acc_to_lds(smem, acc);
    for_each_row4(smem, [&](int r, int col, float4 v) {
        f32x4 q;
        q[0] = scale * v.x; q[1] = scale * v.y; q[2] = scale * v.z; q[3] = scale * v.w;
        *(gf32x4*)(slab + (long)(t.m0 + r) * L.rpad + t.n0 + col) = q;
    });
}

__global__ __launch_bounds__(256) void nsgp_lowrank_reduce_kernel(const ChunkDev* __restrict__ chunks,
                                                                  const LayerDev* __restrict__ layers) {
    const ChunkDev c = chunks[blockIdx.x];
    const LayerDev L = layers[c.tensor];
    const long n = (long)L.rows * L.rpad;
    const long end = (c.start + LR_REDUCE_CHUNK < n) ? c.start + LR_REDUCE_CHUNK : n;
    for (long i = c.start + (long)threadIdx.x * 4; i < end; i += 256 * 4) {   // rpad % 128 == 0 -> n % 4 == 0
        f32x4 s = *(const gf32x4*)(L.slabs + i);
        for (int k = 1; k < L.nsplit; ++k) s += *(const gf32x4*)(L.slabs + (long)k * n + i);
        *(gf32x4*)(L.T + i) = s;
    }
}

template <int OPT>
__global__ __launch_bounds__(256, 2) void nsgp_lowrank_p2_kernel(const TileDev* __restrict__ tiles,
                                                                 const LayerDev* __restrict__ layers,
                                                                 const DynBlock* __restrict__ dyn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileDev t = tiles[blockIdx.x];
    const LayerDev L = layers[t.layer];
    const float* A;
    float scale;
    lowrank_source<OPT>(L, dyn, A, scale);
    f32x16 acc[2][2];
    zero_acc(acc);
    float ra[2][4][4], rb[2][4][4];
    mfma_pipeline<true>(
        (L.rank + BK - 1) / BK, smem, acc,
        [&](int kt, auto s) {
            constexpr int S = decltype(s)::value;
            stage_rows<true>(L.T, L.rpad, L.rows, L.rpad, t.m0, kt * BK, ra[S]);          // T[m][k]
            stage_rows<true>(L.basis, L.cols, L.cols, L.cols, t.n0, kt * BK, rb[S]);       // U[n][k] = V[n][k]
        },
        [&](float* img, int kt, auto s) { write_rows_khi(img, ra[decltype(s)::value], kt * BK, L.rank); },
        [&](float* img, int kt, auto s) { write_rows_khi(img, rb[decltype(s)::value], kt * BK, L.rank); });
    // p += c * (scale*S - T U^T), 16 bytes per lane (the K loop is only r/32 steps long: the epilogue,
    // three global arrays per element, would otherwise dominate this kernel)
    __syncthreads();                       // every wave is done reading the K-loop images
    acc_to_lds(smem, acc);
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's own LDS writes have landed
    const float c = L.basis_scale;
    for_each_row4(smem, [&](int r, int col, float4 v) {
        const long idx = (long)(t.m0 + r) * L.cols + t.n0 + col;
        const f32x4 a = *(const gf32x4*)(A + idx);
        gf32x4* pp = (gf32x4*)(L.p + idx);
        f32x4 pv = *pp;
        pv[0] = pv[0] + c * (scale * a[0] - v.x);
        pv[1] = pv[1] + c * (scale * a[1] - v.y);
        pv[2] = pv[2] + c * (scale * a[2] - v.z);
        pv[3] = pv[3] + c * (scale * a[3] - v.w);
        *pp = pv;
    });
}

// ---- host: plan ---------------------------------------------------------------
template <typename K>
static int enable_big_lds(K kernel) {
    NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, X3_SMEM_BYTES > SMEM_BYTES ? X3_SMEM_BYTES : SMEM_BYTES));
    return NSGP_OK;
}

template <typename K>
static int enable_v2_lds(K kernel) {
    NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM_BYTES));
    return NSGP_OK;
}

}  // namespace nsgp

using namespace nsgp;

struct nsgp_plan {
    int optimizer = 0;
    int n_tensors = 0;
    int n_layers = 0;
    int n_chunks = 0;
    int n_tiles_fast = 0, n_tiles_generic = 0;
    int n_tiles_v2 = 0;         // 256 x 128 / 128 x 128 tiles of the fp16-split kernel (split_kind 2); they replace the fast tiles
    int split_kind = 0;         // dense fast tiles: 0 fp32 MFMA, 1 three-term bf16 split, 2 two-term fp16 split
    double gemm_flops = 0, bytes = 0;
    TensorDev* d_tensors = nullptr;
    LayerDev* d_layers = nullptr;
    TileDev* d_tiles = nullptr;  // dense fast tiles | dense generic tiles | low-rank phase-1 | phase-2
    int n_tiles_lr1 = 0, n_tiles_lr2 = 0, n_chunks_lr = 0, n_lowrank = 0;
    double lowrank_flops = 0;
    ChunkDev* d_chunks_lr = nullptr;
    ChunkDev* d_chunks = nullptr;
    size_t dyn_bytes = 0;
    char* h_dyn[NSLOT] = {nullptr, nullptr, nullptr, nullptr};  // pinned
    char* d_dyn[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_used[NSLOT] = {false, false, false, false};
    int slot = 0;
    // optional per-launch timing: 3 events per recorded step (before update, between, after GEMM)
    std::vector<hipEvent_t> prof_ev;
    int prof_cap = 0, prof_n = 0;
};

static bool tensor_fast(const nsgp_tensor_t& t) {
    return t.rows % BM == 0 && t.cols % BN == 0 && t.cols % BK == 0 && aligned16(t.proj);
}
// the fp16-split kernel takes a layer when it has the fast shape and carries a kind-2 split copy of its projector
static bool tensor_v2(const nsgp_tensor_t& t) {
    return tensor_fast(t) && t.split_kind == 2 && t.proj_split && aligned16(t.proj_split) && aligned16(t.param) && aligned16(t.state0);
}

// low-rank form: needs the fast shape, an aligned basis, and r <= D/4 (else the dense GEMM is cheaper)
static bool tensor_lowrank(const nsgp_tensor_t& t) {
    return t.basis && t.rank > 0 && 4 * (long)t.rank <= t.cols && t.rows % BM == 0 && t.cols % BN == 0 &&
           aligned16(t.basis) && aligned16(t.param) && t.proj;
}
static int lr_rpad(int rank) { return (rank + BN - 1) / BN * BN; }
static int lr_nsplit(int cols, int kchunk) { return (cols + kchunk - 1) / kchunk; }

// One K chunk for all low-rank layers of a plan: the candidate whose phase-1 grid makes the fewest,
// fullest rounds on the 512 workgroup slots (cost model: rounds x (fixed tile overhead + K-steps)).
static int lr_pick_kchunk(const nsgp_tensor_t* tensors, int n) {
    static const int cand[] = {512, 768, 1024, 1536};   // >= 512: every extra slab is re-read by the reduce
    int best = LR_KCHUNK_DEFAULT;
    double best_cost = 1e300;
    for (int kc : cand) {
        long tiles = 0;
        int maxsteps = 0;
        for (int i = 0; i < n; ++i) {
            const nsgp_tensor_t& t = tensors[i];
            if (!tensor_lowrank(t)) continue;
            tiles += (long)(t.rows / BM) * (lr_rpad(t.rank) / BN) * lr_nsplit(t.cols, kc);
            maxsteps = std::max(maxsteps, std::min(kc, t.cols) / BK);
        }
        if (tiles == 0) return LR_KCHUNK_DEFAULT;
        const double rounds = std::ceil(tiles / 512.0);
        const double cost = rounds * (3.0 + maxsteps);   // in K-step units; 3 ~ prologue + epilogue of a tile
        if (cost < best_cost) { best_cost = cost; best = kc; }
    }
    return best;
}
static size_t pad256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t nsgp_plan_workspace_bytes(const nsgp_tensor_t* tensors, int n, int optimizer) {
    if (!tensors) return 0;
    size_t s = 0;
    for (int i = 0; i < n; ++i) {
        const nsgp_tensor_t& t = tensors[i];
        if (!t.proj) continue;
        if (optimizer == NSGP_OPT_ADAM) s += pad256((size_t)t.numel * 4);
        if (tensor_v2(t) && !tensor_lowrank(t)) s += pad256(v2_operand_bytes(t.rows, t.cols)) + pad256((size_t)t.rows * 4);
        if (tensor_lowrank(t)) s += pad256((size_t)t.rows * lr_rpad(t.rank) * 4) * (1 + lr_nsplit(t.cols, 256));  // worst case split
    }
    return s;
}

extern "C" int nsgp_plan_create(nsgp_plan_t** out, const nsgp_tensor_t* tensors, int n, int optimizer,
                                void* workspace, size_t workspace_bytes) {
    if (!out || !tensors || n <= 0) return fail(NSGP_ERR_INVALID, "nsgp_plan_create: null/empty tensor list");
    if (optimizer != NSGP_OPT_SGD && optimizer != NSGP_OPT_ADAM)
        return fail(NSGP_ERR_INVALID, "nsgp_plan_create: unknown optimizer %d", optimizer);
    const size_t need = nsgp_plan_workspace_bytes(tensors, n, optimizer);
    if (need > workspace_bytes || (need && !workspace))
        return fail(NSGP_ERR_WORKSPACE, "nsgp_plan_create: workspace %zu < %zu bytes", workspace_bytes, need);

    std::vector<TensorDev> td(n);
    std::vector<LayerDev> ld;
    std::vector<ChunkDev> cd;
    std::vector<char> layer_fast;
    bool all_split = true;     // every fast dense layer carries a split copy of its projector, all of the same kind
    int split_kind = 0;
    double flops = 0, bytes = 0, lr_flops = 0;
    int n_lowrank = 0;
    const int lr_kchunk = lr_pick_kchunk(tensors, n);
    size_t ws_off = 0;
    for (int i = 0; i < n; ++i) {
        const nsgp_tensor_t& t = tensors[i];
        if (!t.param || t.numel <= 0) return fail(NSGP_ERR_INVALID, "tensor %d: null param or numel<=0", i);
        if (t.hyper < 0 || t.hyper >= NSGP_MAX_HYPER) return fail(NSGP_ERR_LIMIT, "tensor %d: hyper index %d", i, t.hyper);
        if (!t.state0) return fail(NSGP_ERR_INVALID, "tensor %d: state0 is null", i);
        if (optimizer == NSGP_OPT_ADAM && !t.state1) return fail(NSGP_ERR_INVALID, "tensor %d: Adam needs state1", i);
        TensorDev d{t.param, t.state0, t.state1, t.state2, nullptr, (long)t.numel, t.hyper, t.proj ? 1 : 0, nullptr, nullptr, t.cols, 0};
        if (t.proj) {
            if (t.rows <= 0 || t.cols <= 0 || (int64_t)t.rows * t.cols != t.numel)
                return fail(NSGP_ERR_INVALID, "tensor %d: rows*cols (%d*%d) != numel %lld", i, t.rows, t.cols,
                            (long long)t.numel);
            if (optimizer == NSGP_OPT_ADAM) {
                d.u = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off);
                ws_off += ((size_t)t.numel * 4 + 255) & ~(size_t)255;
            }
            LayerDev L{t.param, t.state0, d.u, t.proj, i, t.rows, t.cols, t.hyper, nullptr, nullptr, nullptr, 0, 0, 0, 0, 1.0f,
                       t.proj_split, t.split_kind, 0, nullptr, nullptr, nullptr};
            if (tensor_v2(t) && !tensor_lowrank(t)) {
                // the split copy carries its column scales behind the planes: [D x D x 4 B][scale: D floats][1/scale: D floats]
                L.cinv = reinterpret_cast<const float*>(static_cast<const char*>(t.proj_split) + v2_operand_bytes(t.cols, t.cols)) + t.cols;
                d.a_split = static_cast<char*>(workspace) + ws_off;
                ws_off += pad256(v2_operand_bytes(t.rows, t.cols));
                d.rinv = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off);
                ws_off += pad256((size_t)t.rows * 4);
                L.a_split = d.a_split;
                L.rinv = d.rinv;
            }
            if (tensor_lowrank(t)) {
                L.basis = t.basis;
                L.rank = t.rank;
                L.rpad = lr_rpad(t.rank);
                L.kchunk = lr_kchunk;
                L.nsplit = lr_nsplit(t.cols, lr_kchunk);
                L.basis_scale = t.basis_scale;
                const size_t one = pad256((size_t)t.rows * L.rpad * 4);
                L.T = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off);
                L.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off + one);
                ws_off += one * (1 + L.nsplit);
                lr_flops += 4.0 * t.rows * (double)t.cols * t.rank;
                ++n_lowrank;
            }
            ld.push_back(L);
            layer_fast.push_back(tensor_fast(t) && aligned16(t.param) && aligned16(t.state0) ? 1 : 0);
            if (layer_fast.back() && !tensor_lowrank(t)) {
                const bool ok = t.proj_split && aligned16(t.proj_split) && (t.split_kind == 1 || t.split_kind == 2);
                if (!ok) all_split = false;
                else if (split_kind == 0) split_kind = t.split_kind;
                else if (split_kind != t.split_kind) all_split = false;
            }
            flops += 2.0 * t.rows * (double)t.cols * t.cols;
            bytes += 4.0 * (double)t.cols * t.cols;
        }
        bytes += 5.0 * 4.0 * (double)t.numel;
        td[i] = d;
        if (d.a_split) for (long s = 0; s < t.numel; s += 8L * t.cols) cd.push_back(ChunkDev{i, 1, s});   // 8-row bands
        else for (long s = 0; s < t.numel; s += CHUNK) cd.push_back(ChunkDev{i, 0, s});
    }
    // longest workgroups first (a band of a 4608-wide layer is 36,864 elements, a linear chunk 16,384)
    std::stable_sort(cd.begin(), cd.end(), [&](const ChunkDev& a, const ChunkDev& b) {
        const long la = a.band ? 8L * tensors[a.tensor].cols : CHUNK, lb = b.band ? 8L * tensors[b.tensor].cols : CHUNK;
        return la > lb;
    });

    // ---- tile table.  Layers are grouped by K (= cols, the per-tile cost) in descending order, so
    // the hardware's in-order dispatch of blockIdx does longest-first list scheduling on the 256
    // CUs.  Inside one K group the tiles go to 8 queues keyed on their P column panel and the
    // queues are interleaved round-robin: tiles 8 apart (same blockIdx % 8 = same XCD under the
    // observed round-robin placement; speed only, never correctness) walk the M blocks of one
    // panel, so a panel is fetched from HBM once and re-read from that XCD's L2.
    std::vector<int> order(ld.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ld[a].cols > ld[b].cols; });
    // three tile classes: 0 = fast 128 x 128 (fp32 MFMA / bf16 split), 1 = generic guarded 128 x 128, 2 = fp16-split 256 x 128
    std::vector<TileDev> fast_tiles, gen_tiles, v2_tiles;
    auto layer_class = [&](int li) { return ld[li].a_split ? 2 : (layer_fast[li] ? 0 : 1); };
    for (int pass = 0; pass < 3; ++pass) {
        std::vector<TileDev>& dst = pass == 0 ? fast_tiles : (pass == 1 ? gen_tiles : v2_tiles);
        const int TM = pass == 2 ? 4 * V2_BLOCK_ROWS : BM;
        size_t g0 = 0;
        while (g0 < order.size()) {
            size_t g1 = g0;
            while (g1 < order.size() && ld[order[g1]].cols == ld[order[g0]].cols) ++g1;
            std::vector<TileDev> q[8];
            // Panels go to the 8 queues in CONTIGUOUS runs of the group's panel list (layer after layer), balanced by
            // tile count: an XCD then works on one or two layers of the group instead of all of them, so a layer's A
            // operand is pulled into ~3 L2s instead of 8 (dealing the panels round-robin read every A once per XCD;
            // measured 0.681 vs 0.695 ms on the split kernel, no change on the fp32 one).
            long group_tiles = 0;
            for (size_t oi = g0; oi < g1; ++oi) {
                const int li = order[oi];
                if (layer_class(li) != pass || ld[li].rank > 0) continue;
                group_tiles += (long)((ld[li].rows + TM - 1) / TM) * ((ld[li].cols + BN - 1) / BN);
            }
            long seen = 0;
            for (size_t oi = g0; oi < g1; ++oi) {
                const int li = order[oi];
                if (layer_class(li) != pass || ld[li].rank > 0) continue;
                const int mb = (ld[li].rows + TM - 1) / TM, nb = (ld[li].cols + BN - 1) / BN;
                for (int j = 0; j < nb; ++j) {
                    const int qi = (int)std::min<long>(7, seen * 8 / std::max<long>(group_tiles, 1));
                    // class 2: pad = row blocks of this tile (4, or 2 for a 128-row layer / remainder)
                    for (int m = 0; m < mb; ++m)
                        q[qi].push_back(TileDev{li, m * TM, j * BN, pass == 2 ? std::min(4, (ld[li].rows - m * TM) / V2_BLOCK_ROWS) : 0});
                    seen += mb;
                }
            }
            size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            size_t total = 0;
            for (auto& v : q) total += v.size();
            const size_t base = dst.size();
            while (dst.size() - base < total) {
                for (int x = 0; x < 8 && dst.size() - base < total; ++x) {
                    int src = x;
                    if (pos[src] >= q[src].size()) {  // queue ran dry: borrow from the fullest
                        size_t best = 0;
                        src = -1;
                        for (int y = 0; y < 8; ++y)
                            if (q[y].size() - pos[y] > best) { best = q[y].size() - pos[y]; src = y; }
                        if (src < 0) break;
                    }
                    dst.push_back(q[src][pos[src]++]);
                }
            }
            g0 = g1;
        }
    }

    nsgp_plan* P = new (std::nothrow) nsgp_plan();
    if (!P) return fail(NSGP_ERR_INVALID, "out of host memory");
    P->optimizer = optimizer;
    P->n_tensors = n;
    P->n_layers = (int)ld.size();
    P->n_chunks = (int)cd.size();
    P->n_tiles_fast = (int)fast_tiles.size();
    P->n_tiles_v2 = (int)v2_tiles.size();
    P->split_kind = (all_split && !(fast_tiles.empty() && v2_tiles.empty())) ? split_kind : 0;
    P->n_tiles_generic = (int)gen_tiles.size();
    P->gemm_flops = flops;
    P->bytes = bytes;
    // low-rank tiles: phase 1 (T slabs: [rows x rpad] per K chunk), phase 2 (p tiles, K = r); both
    // in descending-cost order (phase 1 tiles all cost one K chunk; phase 2 cost ~ r)
    std::vector<TileDev> lr1, lr2;
    std::vector<ChunkDev> lr_chunks;
    {
        std::vector<int> lo;
        for (size_t li = 0; li < ld.size(); ++li)
            if (ld[li].rank > 0) lo.push_back((int)li);
        std::stable_sort(lo.begin(), lo.end(), [&](int a, int b) { return ld[a].rank > ld[b].rank; });
        for (int li : lo) {
            const LayerDev& L = ld[li];
            for (int ks = 0; ks < L.nsplit; ++ks)
                for (int j = 0; j < L.rpad / BN; ++j)
                    for (int m = 0; m < L.rows / BM; ++m) lr1.push_back(TileDev{li, m * BM, j * BN, ks});
            for (int j = 0; j < L.cols / BN; ++j)
                for (int m = 0; m < L.rows / BM; ++m) lr2.push_back(TileDev{li, m * BM, j * BN, 0});
            for (long st = 0; st < (long)L.rows * L.rpad; st += LR_REDUCE_CHUNK) lr_chunks.push_back(ChunkDev{li, 0, st});
        }
    }
    std::vector<TileDev> all_tiles(fast_tiles);
    all_tiles.insert(all_tiles.end(), gen_tiles.begin(), gen_tiles.end());
    all_tiles.insert(all_tiles.end(), lr1.begin(), lr1.end());
    all_tiles.insert(all_tiles.end(), lr2.begin(), lr2.end());
    all_tiles.insert(all_tiles.end(), v2_tiles.begin(), v2_tiles.end());
    P->n_tiles_lr1 = (int)lr1.size();
    P->n_tiles_lr2 = (int)lr2.size();
    P->n_chunks_lr = (int)lr_chunks.size();
    P->n_lowrank = n_lowrank;
    P->lowrank_flops = lr_flops;
    P->dyn_bytes = (sizeof(nsgp_hyper_t) * NSGP_MAX_HYPER + sizeof(float*) * (size_t)n + 255) & ~(size_t)255;

#define PLAN_HIP(call)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            nsgp_plan_destroy(P);                                                                    \
            return fail(NSGP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                \
        }                                                                                            \
    } while (0)
    PLAN_HIP(hipMalloc(&P->d_tensors, sizeof(TensorDev) * td.size()));
    PLAN_HIP(hipMemcpy(P->d_tensors, td.data(), sizeof(TensorDev) * td.size(), hipMemcpyHostToDevice));
    PLAN_HIP(hipMalloc(&P->d_chunks, sizeof(ChunkDev) * cd.size()));
    PLAN_HIP(hipMemcpy(P->d_chunks, cd.data(), sizeof(ChunkDev) * cd.size(), hipMemcpyHostToDevice));
    if (!lr_chunks.empty()) {
        PLAN_HIP(hipMalloc(&P->d_chunks_lr, sizeof(ChunkDev) * lr_chunks.size()));
        PLAN_HIP(hipMemcpy(P->d_chunks_lr, lr_chunks.data(), sizeof(ChunkDev) * lr_chunks.size(), hipMemcpyHostToDevice));
    }
    if (!ld.empty()) {
        PLAN_HIP(hipMalloc(&P->d_layers, sizeof(LayerDev) * ld.size()));
        PLAN_HIP(hipMemcpy(P->d_layers, ld.data(), sizeof(LayerDev) * ld.size(), hipMemcpyHostToDevice));
        PLAN_HIP(hipMalloc(&P->d_tiles, sizeof(TileDev) * all_tiles.size()));
        PLAN_HIP(hipMemcpy(P->d_tiles, all_tiles.data(), sizeof(TileDev) * all_tiles.size(), hipMemcpyHostToDevice));
    }
    for (int s = 0; s < NSLOT; ++s) {
        PLAN_HIP(hipHostMalloc(reinterpret_cast<void**>(&P->h_dyn[s]), P->dyn_bytes, hipHostMallocDefault));
        PLAN_HIP(hipMalloc(reinterpret_cast<void**>(&P->d_dyn[s]), P->dyn_bytes));
        PLAN_HIP(hipEventCreateWithFlags(&P->ev[s], hipEventDisableTiming));
    }
#undef PLAN_HIP
    if (getenv("NSGP_DEBUG_ALLOC")) {    // diagnostics: where this plan's own buffers live (to place a faulting address)
        fprintf(stderr, "[nsgp alloc] plan %p: tensors %p +%zu, chunks %p +%zu, layers %p, tiles %p, dyn_bytes %zu\n", (void*)P, P->d_tensors,
                sizeof(TensorDev) * td.size(), P->d_chunks, sizeof(ChunkDev) * cd.size(), P->d_layers, P->d_tiles, P->dyn_bytes);
        for (int s = 0; s < NSLOT; ++s) fprintf(stderr, "[nsgp alloc]   slot %d: pinned %p device %p\n", s, (void*)P->h_dyn[s], (void*)P->d_dyn[s]);
    }
    int rc;
    if ((rc = enable_big_lds(nsgp_lowrank_p1_kernel<NSGP_OPT_SGD>)) || (rc = enable_big_lds(nsgp_lowrank_p1_kernel<NSGP_OPT_ADAM>)) ||
        (rc = enable_big_lds(nsgp_lowrank_p2_kernel<NSGP_OPT_SGD>)) || (rc = enable_big_lds(nsgp_lowrank_p2_kernel<NSGP_OPT_ADAM>)) ||
        (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_SGD, true>)) || (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_SGD, false>)) ||
        (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_ADAM, true>)) || (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_ADAM, false>)) ||
        (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_SGD, true, 1>)) || (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_ADAM, true, 1>)) ||
        (rc = enable_v2_lds(nsgp_project_v2_kernel<NSGP_OPT_SGD>)) || (rc = enable_v2_lds(nsgp_project_v2_kernel<NSGP_OPT_ADAM>))) {
        nsgp_plan_destroy(P);
        return rc;
    }
    *out = P;
    return NSGP_OK;
}

extern "C" int nsgp_plan_destroy(nsgp_plan_t* P) {
    if (!P) return NSGP_OK;
    // Every launch of this plan is followed by an event record on its stream (nsgp_plan_step): waiting for the used slots
    // means no kernel still reads the tables, and no copy still reads the pinned ring, when they are released below.
    if (getenv("NSGP_DEBUG_ALLOC")) fprintf(stderr, "[nsgp alloc] destroy plan %p\n", (void*)P);
    hipError_t first = hipSuccess;
    const char* what = "";
    auto keep = [&](hipError_t e, const char* w) { if (e != hipSuccess && first == hipSuccess) { first = e; what = w; } };
    for (int s = 0; s < NSLOT; ++s)
        if (P->ev[s] && P->ev_used[s]) keep(hipEventSynchronize(P->ev[s]), "hipEventSynchronize");
    for (int s = 0; s < NSLOT; ++s) {
        if (P->ev[s]) keep(hipEventDestroy(P->ev[s]), "hipEventDestroy");
        if (P->h_dyn[s]) keep(hipHostFree(P->h_dyn[s]), "hipHostFree");
        if (P->d_dyn[s]) keep(hipFree(P->d_dyn[s]), "hipFree(dyn)");
    }
    for (hipEvent_t e : P->prof_ev) keep(hipEventDestroy(e), "hipEventDestroy(profile)");
    if (P->d_tensors) keep(hipFree(P->d_tensors), "hipFree(tensors)");
    if (P->d_layers) keep(hipFree(P->d_layers), "hipFree(layers)");
    if (P->d_tiles) keep(hipFree(P->d_tiles), "hipFree(tiles)");
    if (P->d_chunks) keep(hipFree(P->d_chunks), "hipFree(chunks)");
    if (P->d_chunks_lr) keep(hipFree(P->d_chunks_lr), "hipFree(chunks_lr)");
    delete P;
    if (first != hipSuccess) return fail(NSGP_ERR_HIP, "nsgp_plan_destroy: %s failed: %s", what, hipGetErrorString(first));
    return NSGP_OK;
}

extern "C" int nsgp_plan_stats(const nsgp_plan_t* P, double* gemm_flops, double* bytes, int* n_tiles, int* n_proj) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_plan_stats: null plan");
    if (gemm_flops) *gemm_flops = P->gemm_flops;  // the dense form's 2*Cout*D^2 for every projected layer
    if (bytes) *bytes = P->bytes;
    if (n_tiles) *n_tiles = P->n_tiles_fast + P->n_tiles_generic + P->n_tiles_v2;
    if (n_proj) *n_proj = P->n_layers;
    return NSGP_OK;
}

extern "C" int nsgp_plan_lowrank_stats(const nsgp_plan_t* P, int* n_lowrank, double* lowrank_flops, int* n_tiles_p1, int* n_tiles_p2) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_plan_lowrank_stats: null plan");
    if (n_lowrank) *n_lowrank = P->n_lowrank;
    if (lowrank_flops) *lowrank_flops = P->lowrank_flops;
    if (n_tiles_p1) *n_tiles_p1 = P->n_tiles_lr1;
    if (n_tiles_p2) *n_tiles_p2 = P->n_tiles_lr2;
    return NSGP_OK;
}

extern "C" int nsgp_plan_uses_split_mfma(const nsgp_plan_t* P) { return P ? P->split_kind : 0; }

extern "C" int nsgp_plan_tile_counts(const nsgp_plan_t* P, int* fast_128, int* generic_128, int* split_f16_256) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_plan_tile_counts: null plan");
    if (fast_128) *fast_128 = P->n_tiles_fast;
    if (generic_128) *generic_128 = P->n_tiles_generic;
    if (split_f16_256) *split_f16_256 = P->n_tiles_v2;
    return NSGP_OK;
}

// [pre-tiled two-term split of diag(cscale) P^T: D*D*4 B][cscale: D floats][1 / cscale: D floats]
extern "C" size_t nsgp_split_projector_f16_bytes(int D) { return D > 0 ? v2_operand_bytes(D, D) + 2 * (size_t)D * 4 : 0; }

extern "C" int nsgp_split_projector_f16(const float* proj, int D, void* out, void* stream_) {
    if (!proj || !out || D <= 0 || D % 64 != 0) return fail(NSGP_ERR_INVALID, "nsgp_split_projector_f16: bad argument (D must be a multiple of 64)");
    if (!aligned16(out)) return fail(NSGP_ERR_INVALID, "nsgp_split_projector_f16: output must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* cscale = reinterpret_cast<float*>(static_cast<char*>(out) + v2_operand_bytes(D, D));
    hipLaunchKernelGGL(nsgp_col_scales_f16x2_kernel, dim3((D + 31) / 32), dim3(256), 0, stream, proj, D, D, cscale, cscale + D);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nsgp_split_transpose_f16x2_v2_kernel, dim3((D + 31) / 32, (D + 31) / 32), dim3(256), 0, stream, proj, D, D, cscale, out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" size_t nsgp_split_projector_bytes(int D) { return D > 0 ? (size_t)D * D * 6 : 0; }

extern "C" int nsgp_split_projector(const float* proj, int D, void* out, void* stream_) {
    if (!proj || !out || D <= 0 || D % 8 != 0) return fail(NSGP_ERR_INVALID, "nsgp_split_projector: bad argument (D must be a multiple of 8)");
    if (!aligned16(out)) return fail(NSGP_ERR_INVALID, "nsgp_split_projector: output must be 16-byte aligned");
    hipLaunchKernelGGL(nsgp_split_transpose_bf16x3_kernel, dim3((D + 31) / 32, (D + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       proj, D, D, static_cast<__bf16*>(out));
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int nsgp_plan_step(nsgp_plan_t* P, float* const* grads, const nsgp_hyper_t* hyper, int n_hyper,
                              void* stream_) {
    if (!P || !grads || !hyper) return fail(NSGP_ERR_INVALID, "nsgp_plan_step: null argument");
    if (n_hyper <= 0 || n_hyper > NSGP_MAX_HYPER)
        return fail(NSGP_ERR_LIMIT, "nsgp_plan_step: n_hyper %d outside 1..%d", n_hyper, NSGP_MAX_HYPER);
    for (int i = 0; i < P->n_tensors; ++i)
        if (!grads[i]) return fail(NSGP_ERR_INVALID, "nsgp_plan_step: grad %d is null (the reference raises too)", i);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int s = P->slot;
    P->slot = (s + 1) % NSLOT;
    if (P->ev_used[s]) NSGP_HIP(hipEventSynchronize(P->ev[s]));  // slot reuse: its last upload was consumed
    DynBlock* h = reinterpret_cast<DynBlock*>(P->h_dyn[s]);
    std::memset(h->hyper, 0, sizeof(h->hyper));
    std::memcpy(h->hyper, hyper, sizeof(nsgp_hyper_t) * n_hyper);
    std::memcpy(h->grads, grads, sizeof(float*) * (size_t)P->n_tensors);
    NSGP_HIP(hipMemcpyAsync(P->d_dyn[s], P->h_dyn[s], P->dyn_bytes, hipMemcpyHostToDevice, stream));
    const DynBlock* d = reinterpret_cast<const DynBlock*>(P->d_dyn[s]);
    const bool prof = P->prof_n < P->prof_cap;
    if (prof) NSGP_HIP(hipEventRecord(P->prof_ev[3 * P->prof_n + 0], stream));

    if (P->optimizer == NSGP_OPT_SGD)
        hipLaunchKernelGGL(nsgp_update_kernel<NSGP_OPT_SGD>, dim3(P->n_chunks), dim3(256), 0, stream, P->d_chunks, P->d_tensors, d);
    else
        hipLaunchKernelGGL(nsgp_update_kernel<NSGP_OPT_ADAM>, dim3(P->n_chunks), dim3(256), 0, stream, P->d_chunks, P->d_tensors, d);
    NSGP_LAUNCH_CHECK();
    if (prof) NSGP_HIP(hipEventRecord(P->prof_ev[3 * P->prof_n + 1], stream));
    if (P->n_tiles_v2 > 0) {
        const TileDev* vt = P->d_tiles + P->n_tiles_fast + P->n_tiles_generic + P->n_tiles_lr1 + P->n_tiles_lr2;
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL(nsgp_project_v2_kernel<NSGP_OPT_SGD>, dim3(P->n_tiles_v2), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, vt, P->d_layers, d);
        else
            hipLaunchKernelGGL(nsgp_project_v2_kernel<NSGP_OPT_ADAM>, dim3(P->n_tiles_v2), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, vt, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
    }
    if (P->n_tiles_fast > 0) {
        if (P->split_kind == 1) {
            if (P->optimizer == NSGP_OPT_SGD)
                hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_SGD, true, 1>), dim3(P->n_tiles_fast), dim3(THREADS), X3_SMEM_BYTES, stream, P->d_tiles, P->d_layers, d);
            else
                hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_ADAM, true, 1>), dim3(P->n_tiles_fast), dim3(THREADS), X3_SMEM_BYTES, stream, P->d_tiles, P->d_layers, d);
        } else if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_SGD, true>), dim3(P->n_tiles_fast), dim3(THREADS), SMEM_BYTES, stream, P->d_tiles, P->d_layers, d);
        else
            hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_ADAM, true>), dim3(P->n_tiles_fast), dim3(THREADS), SMEM_BYTES, stream, P->d_tiles, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
    }
    if (P->n_tiles_generic > 0) {
        const TileDev* gt = P->d_tiles + P->n_tiles_fast;
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_SGD, false>), dim3(P->n_tiles_generic), dim3(THREADS), SMEM_BYTES, stream, gt, P->d_layers, d);
        else
            hipLaunchKernelGGL((nsgp_project_kernel<NSGP_OPT_ADAM, false>), dim3(P->n_tiles_generic), dim3(THREADS), SMEM_BYTES, stream, gt, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
    }
    if (P->n_tiles_lr1 > 0) {
        const TileDev* t1 = P->d_tiles + P->n_tiles_fast + P->n_tiles_generic;
        const TileDev* t2 = t1 + P->n_tiles_lr1;
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL(nsgp_lowrank_p1_kernel<NSGP_OPT_SGD>, dim3(P->n_tiles_lr1), dim3(THREADS), SMEM_BYTES, stream, t1, P->d_layers, d);
        else
            hipLaunchKernelGGL(nsgp_lowrank_p1_kernel<NSGP_OPT_ADAM>, dim3(P->n_tiles_lr1), dim3(THREADS), SMEM_BYTES, stream, t1, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(nsgp_lowrank_reduce_kernel, dim3(P->n_chunks_lr), dim3(256), 0, stream, P->d_chunks_lr, P->d_layers);
        NSGP_LAUNCH_CHECK();
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL(nsgp_lowrank_p2_kernel<NSGP_OPT_SGD>, dim3(P->n_tiles_lr2), dim3(THREADS), SMEM_BYTES, stream, t2, P->d_layers, d);
        else
            hipLaunchKernelGGL(nsgp_lowrank_p2_kernel<NSGP_OPT_ADAM>, dim3(P->n_tiles_lr2), dim3(THREADS), SMEM_BYTES, stream, t2, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
    }
    if (prof) {
        NSGP_HIP(hipEventRecord(P->prof_ev[3 * P->prof_n + 2], stream));
        ++P->prof_n;
    }
    NSGP_HIP(hipEventRecord(P->ev[s], stream));
    P->ev_used[s] = true;
    return NSGP_OK;
}

extern "C" int nsgp_plan_profile_begin(nsgp_plan_t* P, int max_steps) {
    if (!P || max_steps < 0 || max_steps > 4096) return fail(NSGP_ERR_INVALID, "nsgp_plan_profile_begin: bad argument");
    while ((int)P->prof_ev.size() < 3 * max_steps) {
        hipEvent_t e;
        NSGP_HIP(hipEventCreate(&e));
        P->prof_ev.push_back(e);
    }
    P->prof_cap = max_steps;
    P->prof_n = 0;
    return NSGP_OK;
}

extern "C" int nsgp_plan_profile_end(nsgp_plan_t* P, int* n_steps, float* update_ms_avg, float* gemm_ms_avg) {
    if (!P) return fail(NSGP_ERR_INVALID, "nsgp_plan_profile_end: null plan");
    double u = 0, g = 0;
    for (int i = 0; i < P->prof_n; ++i) {
        NSGP_HIP(hipEventSynchronize(P->prof_ev[3 * i + 2]));
        float a = 0, b = 0;
        NSGP_HIP(hipEventElapsedTime(&a, P->prof_ev[3 * i + 0], P->prof_ev[3 * i + 1]));
        NSGP_HIP(hipEventElapsedTime(&b, P->prof_ev[3 * i + 1], P->prof_ev[3 * i + 2]));
        u += a;
        g += b;
    }
    if (n_steps) *n_steps = P->prof_n;
    if (update_ms_avg) *update_ms_avg = P->prof_n ? (float)(u / P->prof_n) : 0.0f;
    if (gemm_ms_avg) *gemm_ms_avg = P->prof_n ? (float)(g / P->prof_n) : 0.0f;
    P->prof_cap = 0;
    P->prof_n = 0;
    return NSGP_OK;
}

extern "C" int nsgp_project(const float* a, const float* proj, float* out, int rows, int cols, float scale,
                            int accumulate, void* stream_) {
    if (!a || !proj || !out || rows <= 0 || cols <= 0) return fail(NSGP_ERR_INVALID, "nsgp_project: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool fast = rows % BM == 0 && cols % BN == 0 && aligned16(a) && aligned16(proj) && aligned16(out);
    dim3 grid((cols + BN - 1) / BN, (rows + BM - 1) / BM);
    int rc;
    if (fast) {
        if ((rc = enable_big_lds(nsgp_project_single_kernel<true>))) return rc;
        hipLaunchKernelGGL(nsgp_project_single_kernel<true>, grid, dim3(THREADS), SMEM_BYTES, stream, a, proj, out, rows, cols, scale, accumulate);
    } else {
        if ((rc = enable_big_lds(nsgp_project_single_kernel<false>))) return rc;
        hipLaunchKernelGGL(nsgp_project_single_kernel<false>, grid, dim3(THREADS), SMEM_BYTES, stream, a, proj, out, rows, cols, scale, accumulate);
    }
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

// ---- misc ABI -----------------------------------------------------------------
// Diagnostics for the GPU test runs: a process that dies with SIGABRT (the HIP / HSA runtimes abort() on a GPU memory
// fault or a queue error, C++ on an escaped exception) normally leaves nothing but Python's "Fatal Python error: Aborted".
// This handler writes the NATIVE call stack of the aborting thread to fd 2 first, then hands over to whatever handler was
// installed before it (Python's faulthandler, or the default action).
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static struct sigaction g_prev_abrt;
static void abort_backtrace_handler(int sig, siginfo_t* info, void* uc) {
    static const char head[] = "\n[nsgp_repre] SIGABRT -- native stack of the aborting thread:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    if (g_prev_abrt.sa_flags & SA_SIGINFO) {
        if (g_prev_abrt.sa_sigaction) { g_prev_abrt.sa_sigaction(sig, info, uc); return; }
    } else if (g_prev_abrt.sa_handler != SIG_DFL && g_prev_abrt.sa_handler != SIG_IGN) {
        g_prev_abrt.sa_handler(sig);
        return;
    }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
extern "C" int nsgp_debug_install_abort_backtrace(void) {
    static bool installed = false;
    if (installed) return NSGP_OK;
    void* warm[4];
    (void)backtrace(warm, 4);                  // loads libgcc now: backtrace() must not dlopen inside a signal handler
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = abort_backtrace_handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGABRT, &sa, &g_prev_abrt) != 0) return fail(NSGP_ERR_INVALID, "sigaction(SIGABRT) failed");
    installed = true;
    return NSGP_OK;
}
extern "C" int nsgp_abi_version(void) { return NSGP_ABI_VERSION; }
extern "C" const char* nsgp_last_error(void) { return err_buf(); }
extern "C" int nsgp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int nsgp_device_arch(char* buf, int buflen) {
    if (!buf || buflen <= 0) return NSGP_ERR_INVALID;
    buf[0] = 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(NSGP_ERR_HIP, "no HIP device");
    std::snprintf(buf, buflen, "%s", prop.gcnArchName);
    return NSGP_OK;
}
