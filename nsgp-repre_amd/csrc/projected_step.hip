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
    // low-rank form (rank > 0): U as k-quads [D/4][rpad][4] and row-major [D][rpad] (columns >= rank are zero),
    // T = (scale*S) U  [rows x rpad], c = 1/||I - U U^T||_F (or 1)
    const float* ukq;
    const float* urm;
    float* T;
    float* slabs;       // nsplit > 1: [nsplit][rows][rpad] partial T of each K range
    int rank, rpad, nsplit;
    float basis_scale;
    const void* split;     // split copy of proj^T: kind 2 = pre-tiled column-scaled fp16 pair (gemm_f16x2_v2.hpp)
    int split_kind;
    int pad;
    const float* cinv;     // kind 2: 1 / (power-of-two scale of each projector column), D floats behind the split copy
    const void* a_split;   // kind 2: this step's update, pre-tiled row-scaled fp16 pair (written by the elementwise launch)
    const float* rinv;     // kind 2: 1 / (scale of each update row)
};

struct TileDev {
    int layer, m0, n0, pad;  // pad: split-K slice index (low-rank phase 1)
};


struct ChunkDev {
    int tensor;
    int band;    // 1: rows [start / cols, +8) of a projected tensor on the fp16-split path; 0: a linear chunk of CHUNK elements
    long start;
};

constexpr int CHUNK = 4096;   // elements per workgroup of the elementwise kernel: 4 float4 rounds per thread.  (A workgroup is a chain of dependent
                              // load -> store rounds; the kernel gets its bandwidth from the number of workgroups in flight, and with 16,384-element
                              // chunks the 14.6 M un-projected elements of R-50-FPN were 3.5 workgroups per CU: 2.5 TB/s.)
constexpr int NSLOT = 4;      // depth of the per-step upload ring
// events per profiled step: start, after the plain launch, after the dense launches, 3 per low-rank group (nsgp_plan::prof_per_step)
// Cache blocking of the low-rank launches (layers cut into G groups whose (update, p) working set fits the 256 MB memory-side cache,
// fused -> reduce -> apply issued group by group) was measured and NOT kept: every extra group costs ~17 us in the fused launch and
// ~10 us in the apply launch (launch boundary: tail + ramp) and the apply launch gets no faster -- 0.28 / 0.31 / 0.35 / 0.36 / 0.41 ms
// per R-50 step at G = 1 / 2 / 3 / 4 / 6 (profiles/r03/lr_groups_study.log).  The group tables stay (one group).
constexpr int g_lr_groups = 1;

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

// One chunk of the multi-tensor update.  BANDS = false compiles the band path (row scales + split copy of the fp16 dense path) out:
// that form is what the fused low-rank launch appends to its grid for the un-projected tensors.
template <int OPT, bool BANDS>
__device__ __forceinline__ void update_chunk(const ChunkDev c, const TensorDev* __restrict__ tensors, const DynBlock* __restrict__ dyn,
                                             unsigned* s_rowmax, float* s_rowscale) {
    const TensorDev T = tensors[c.tensor];
    const nsgp_hyper_t h = dyn->hyper[T.hyper];
    float* __restrict__ gp = dyn->grads[c.tensor];
    const bool band = BANDS && c.band != 0;              // uniform per workgroup
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

template <int OPT>
__global__ __launch_bounds__(256) void nsgp_update_kernel(const ChunkDev* __restrict__ chunks,
                                                          const TensorDev* __restrict__ tensors,
                                                          const DynBlock* __restrict__ dyn) {
    __shared__ unsigned s_rowmax[8];
    __shared__ float s_rowscale[8];
    update_chunk<OPT, true>(chunks[blockIdx.x], tensors, dyn, s_rowmax, s_rowscale);
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

template <int OPT, bool FAST>
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
    if (!FAST || ((uintptr_t)A & 15u) == 0)
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

// ---- low-rank form:  p += c * (u - (u U) U^T),  u = scale*S,  U = the r removed directions ([D x r], r <= 128) ----
// The north star's own form (g <- g - U (U^T g)).  It is used for projectors the optimizer BUILT ITSELF in the head form
// P = c * (I - U U^T) (nsgp_build_projector_head, from the orthonormalised top-r eigenvectors): for those the dense u @ P
// and this form are the same function of (u, U) up to the fp32 rounding of P's entries, and the step costs 4*Cout*D*r
// FLOP and NO projector traffic instead of 2*Cout*D^2 FLOP and D^2 projector bytes -- the step becomes HBM-bound on the
// update itself.  Everything is exact fp32 on v_mfma_f32_32x32x2_f32 (lane l supplies A[i = l & 31][k'] and
// B[k'][j = l & 31], k' = l >> 5): at r <= 128 the matrix work is a few GFLOP per step and hides under the memory stream.
// Both launches serve every rank class (U padded to rpad = 32, 64 or 128 columns) of a plan at once.
//
//   launch U+T (nsgp_update_lr_kernel): the layer's elementwise update fused with T[32 rows x rpad] = scale * (S U), see the
//             kernel.  Wide layers are cut into S K ranges (one workgroup each, ~32 groups of 32 columns) whose slabs a small
//             launch (nsgp_lr_reduce_kernel) sums in range order.  (A "last workgroup to arrive sums the slabs" variant needs an
//             agent-scope release / acquire per workgroup -- an L2 write-back + invalidate on a multi-XCD part -- and measured
//             0.77 ms; a stand-alone T launch re-reading the update measured 0.063 ms, MFMA-latency bound.)
//   launch A  (nsgp_lr_apply_kernel):  p += c * (scale*S - T U^T), K = rpad.  One workgroup per (32 rows x <= 256
//             columns), the four waves on adjacent 32-column blocks: a wave loads S and p in the (transposed) MFMA C layout
//             as 16-byte pieces, T and U rows as 16-byte fragments (L1 / L2 hits), does rpad/2 MFMAs and writes p.
//             HBM-bound: 12 bytes per element.
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

constexpr int LRA_COLS = 256;        // columns of one apply workgroup
static_assert(LRA_COLS == 256, "the apply kernel gives each of its four waves at most two 32-column blocks");
constexpr int LR_MAX_RANK = 256;       // rank classes 32 / 64 / 128 share the launches; 129 .. 256 removed directions (rpad = 256) get their own pair (NB = 2 / MAXR = 256)
constexpr int LR_TILE_LD = 36;       // floats per row of a [32 x 32] LDS tile: 144 B, so that 8 lanes reading 16 B of 8 consecutive rows cover all 32 banks
constexpr int LR_TILE = 32 * LR_TILE_LD;

typedef __attribute__((address_space(3))) f32x4 lf32x4;

// streaming accesses of the low-rank launches (data touched once per launch): the non-temporal hint keeps them from evicting U
// and T, which every row block re-reads, from L2
__device__ __forceinline__ f32x4 load4_nt(const float* p) { return __builtin_nontemporal_load((const gf32x4*)p); }
__device__ __forceinline__ void store4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, (gf32x4*)p); }
__device__ __forceinline__ f32x4 load4_a4_nt(const float* p) {
    const f32x4_a4 v = __builtin_nontemporal_load((const gf32x4_a4*)p);
    return f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ f32x4 load4_a4(const float* p) {
    const f32x4_a4 v = *(const gf32x4_a4*)p;
    return f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void lds_put4(float* tile, int off, f32x4 v) { *(lf32x4*)((__attribute__((address_space(3))) float*)tile + off) = v; }
__device__ __forceinline__ f32x4 lds_get4(const float* tile, int off) { return *(const lf32x4*)((const __attribute__((address_space(3))) float*)tile + off); }

// Global <-> MFMA layout goes through a wave-private [32 rows x 32 floats] LDS tile: in memory a wave-instruction covers 8 rows
// x 128 contiguous bytes (lane l: row l >> 3 (+ 8 per piece), 16-byte chunk l & 7 -- whole cache lines, each touched once),
// the MFMA wants lane (i = l & 31, h = l >> 5) to hold 16 bytes of row i.  (Loading in the MFMA layout directly -- 32 rows x 32
// bytes per instruction, four instructions per line -- measured 0.12 ms for the apply launch against the stream's 0.06.)

// Launch U+T: the elementwise update of a low-rank layer AND its T = (scale*S) U in one pass over the gradient stream.
// One workgroup of 4 waves owns 32 rows x one K range (a run of 32-column groups) of a layer and walks it four groups at a
// time.  Per round
//   PRODUCE  wave w takes group gb + w (512 contiguous bytes per row across the workgroup): it loads g, the momentum buffer /
//            Adam moments and p as 8 rows x 128 B per instruction, applies the same per-element arithmetic as nsgp_update_kernel
//            (sgd_elem / adam_elem), stores the state (and the mutated gradient / Adam's update), and parks the 32 x 32 block of
//            update values in LDS tile w of the round's buffer;
//   one barrier (two tile buffers: the barrier of round k+1 also says that every wave is done reading the tiles of round k-1);
//   CONSUME  wave w owns ONE 32-column block of U, jb = w mod NJ (NJ = rpad / 32 = 1, 2 or 4), and contracts NJ of the four
//            tiles with it -- tiles (w div NJ) + (4 / NJ) x -- 16 x v_mfma_f32_32x32x2_f32 per tile into a single accumulator
//            (U as k-quads: one coalesced 16-byte load per lane and k8 step, L2-resident, issued before the barrier).
// Every wave therefore streams one group and does 16 NJ MFMAs per round whatever the rank: ONE kernel and ONE launch for all
// rank classes, 16 accumulator registers.  The matrix work (2*Cout*D*r FLOP per step, ~20 us of the matrix pipe chip-wide)
// hides under the HBM stream of the update, and the update values are never re-read for T.  At the end the waves that share a
// column block are summed in wave order through LDS into T (one K range per layer) or into the range's slab
// (nsgp_lr_reduce_kernel sums the slabs in range order): deterministic.
// (First form: each wave contracted its own groups with ALL column blocks -- 16 NJ accumulator registers, a launch per rank
// class, the U loads of blocks 1.. on the dependent chain: 78 us for each of the two wide classes of R-50-FPN.)
// NB = 2: the WIDE rank class (129 .. 256 removed directions, rpad = 256): every wave keeps TWO 32-column blocks of U (wave and wave + 4) and
// contracts all four tiles of a round with both -- 32 accumulator registers, 128 MFMAs per round; its layers run in their own launch.
template <int OPT, int NB>
__global__ __launch_bounds__(256) void nsgp_update_lr_kernel(const TileDev* __restrict__ units, const LayerDev* __restrict__ layers,
                                                             const TensorDev* __restrict__ tensors, const DynBlock* __restrict__ dyn,
                                                             int n_units, const ChunkDev* __restrict__ plain_chunks) {
    __shared__ __attribute__((aligned(16))) float lds[8 * LR_TILE];   // 2 buffers x 4 tiles; afterwards the waves' partial T blocks (1024 floats each)
    // The grid's tail: the un-projected tensors' chunks (nsgp_update_kernel's work).  They are independent of the low-rank units, so
    // they ride in this launch -- one launch boundary less (a boundary costs 10-17 us here: tail + ramp, profiles/r03/lr_groups_study.log),
    // and 4,096-element chunks dispatched last fill the tail the 32-row units leave.
    if ((int)blockIdx.x >= n_units) {
        update_chunk<OPT, false>(plain_chunks[blockIdx.x - n_units], tensors, dyn, nullptr, nullptr);
        return;
    }
    const TileDev t = units[blockIdx.x];           // m0 = first row, pad = K range index
    const LayerDev L = layers[t.layer];
    const TensorDev T = tensors[L.tensor];
    const nsgp_hyper_t h = dyn->hyper[T.hyper];
    float* __restrict__ gp = dyn->grads[L.tensor];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, hh = lane >> 5, r8 = lane >> 3, ch = lane & 7;
    const int ngroups = L.cols >> 5;               // groups of 32 k = one 128-byte line per row
    const int per = (ngroups + L.nsplit - 1) / L.nsplit;
    const int g0 = min(ngroups, t.pad * per), g1 = min(ngroups, g0 + per);
    const int nj = NB == 2 ? 4 : (L.rpad >> 5);    // 1, 2 or 4 column blocks across the waves (wide: 4, twice)
    const int jb = wave & (nj - 1), sub = wave / nj, tstride = 4 / nj;
    // what the projection reads as its A operand, and whether the mutated gradient must be stored (as in nsgp_update_kernel)
    const bool a_is_buf = (OPT == NSGP_OPT_SGD) && h.momentum != 0.0f && !h.nesterov;
    bool wg = h.write_grad != 0;
    if (OPT == NSGP_OPT_SGD && !a_is_buf) wg = true;
    const bool store_g = wg && (h.weight_decay != 0.0f || (OPT == NSGP_OPT_SGD && h.momentum != 0.0f && h.nesterov));
    const bool need_p = h.weight_decay != 0.0f || h.decoupled_decay != 0.0f;
    const float scale = (OPT == NSGP_OPT_SGD) ? -h.lr : 1.0f;
    const long rowbase = (long)(t.m0 + r8) * L.cols + 4 * ch;             // + 8 it rows, + 32 g columns
    const long r8s = 8L * L.cols;
    const float* ub = L.ukq + ((long)hh * L.rpad + jb * 32 + i) * 4;      // quad 8 g + 2 st + hh, column 32 jb + i (second block of a wide layer: + 128 columns)
    const long qs = 2L * L.rpad * 4, qg = 8L * L.rpad * 4;                // floats per k8 step, per group
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[b][v] = 0.0f;
    // (Measured and not kept: requesting the NEXT round's stream before the barrier and the matrix work -- in front of or behind
    // the U fragments, 8 or 16 groups per workgroup -- 128-134 us on the R-50 table either way, at 134 instead of 112 VGPRs: vmcnt
    // retires in order, so every fragment wait inside the consume phase waits for the prefetched stream as well.)
    int buf = 0;
    for (int gb = g0; gb < g1; gb += 4, buf ^= 1) {
        float* tiles = lds + buf * (4 * LR_TILE);
        const int g = gb + wave;
        if (g < g1) {
            // ---- produce: the update of group g
            f32x4 a4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const long off = rowbase + it * r8s + 32L * g;
                const f32x4 g4 = load4_a4_nt(gp + off);
                float gv[4] = {g4[0], g4[1], g4[2], g4[3]};
                float pv[4] = {0, 0, 0, 0};
                if (need_p) {
                    const f32x4 p4 = load4_nt(T.p + off);
                    pv[0] = p4[0]; pv[1] = p4[1]; pv[2] = p4[2]; pv[3] = p4[3];
                }
                if (OPT == NSGP_OPT_SGD) {
                    float bv[4] = {0, 0, 0, 0};
                    if (h.momentum != 0.0f) {
                        const f32x4 s4 = *(const gf32x4*)(T.s0 + off);
                        bv[0] = s4[0]; bv[1] = s4[1]; bv[2] = s4[2]; bv[3] = s4[3];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) sgd_elem(pv[e], gv[e], bv[e], h, true);
                    if (h.momentum != 0.0f) *(gf32x4*)(T.s0 + off) = f32x4{bv[0], bv[1], bv[2], bv[3]};
                    a4[it] = a_is_buf ? f32x4{bv[0], bv[1], bv[2], bv[3]} : f32x4{gv[0], gv[1], gv[2], gv[3]};
                } else {
                    const f32x4 m4 = *(const gf32x4*)(T.s0 + off), v4 = *(const gf32x4*)(T.s1 + off);
                    float mv[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]}, xv[4] = {0, 0, 0, 0}, uv[4];
                    if (h.amsgrad) {
                        const f32x4 x4 = *(const gf32x4*)(T.s2 + off);
                        xv[0] = x4[0]; xv[1] = x4[1]; xv[2] = x4[2]; xv[3] = x4[3];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) adam_elem(pv[e], gv[e], mv[e], vv[e], xv[e], uv[e], h, true);
                    *(gf32x4*)(T.s0 + off) = f32x4{mv[0], mv[1], mv[2], mv[3]};
                    *(gf32x4*)(T.s1 + off) = f32x4{vv[0], vv[1], vv[2], vv[3]};
                    if (h.amsgrad) *(gf32x4*)(T.s2 + off) = f32x4{xv[0], xv[1], xv[2], xv[3]};
                    a4[it] = f32x4{uv[0], uv[1], uv[2], uv[3]};
                    *(gf32x4*)(T.u + off) = a4[it];
                }
                if (store_g) __builtin_nontemporal_store(f32x4_a4{gv[0], gv[1], gv[2], gv[3]}, (gf32x4_a4*)(gp + off));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) lds_put4(tiles + wave * LR_TILE, (r8 + 8 * it) * LR_TILE_LD + 4 * ch, a4[it]);
        }
        // ---- consume: my column block of U against nj of the round's tiles
        f32x4 bx[2][NB][4];                        // the fragments of tile x + 1 are requested while tile x is contracted
        auto load_b = [&](int x, auto set_) {
            constexpr int set = decltype(set_)::value;
            const int gx = gb + sub + tstride * x;
            if (x < nj && gx < g1) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int st = 0; st < 4; ++st) bx[set][b][st] = *(const gf32x4*)(ub + gx * qg + st * qs + b * 512);
            }
        };
        auto contract = [&](int x, auto set_) {
            constexpr int set = decltype(set_)::value;
            const int tx = sub + tstride * x;
            if (x < nj && gb + tx < g1) {
                f32x4 f[4];
#pragma unroll
                for (int st = 0; st < 4; ++st) f[st] = lds_get4(tiles + tx * LR_TILE, i * LR_TILE_LD + 4 * (2 * st + hh));   // k = 32 g + 8 st + 4 hh + e
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[st][e], bx[set][b][st][e], acc[b], 0, 0, 0);
            }
        };
        load_b(0, IC<0>{});
        __syncthreads();
        load_b(1, IC<1>{});
        contract(0, IC<0>{});
        load_b(2, IC<0>{});
        contract(1, IC<1>{});
        load_b(3, IC<1>{});
        contract(2, IC<0>{});
        contract(3, IC<1>{});
    }
    // the waves that share a column block (wave = sub * nj + jb), summed in wave order (deterministic)
    float* mine = lds + wave * 1024;
    float* dst = (L.nsplit > 1) ? L.slabs + (long)t.pad * L.rows * L.rpad : L.T;
    const float osc = (L.nsplit > 1) ? 1.0f : scale;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        __syncthreads();                           // every wave is done with the tiles (with the previous block's partials): the buffer becomes four partial blocks
#pragma unroll
        for (int v = 0; v < 16; ++v) mine[acc_row(v, lane) * 32 + i] = acc[b][v];
        __syncthreads();
        for (int idx = threadIdx.x; idx < nj * 1024; idx += 256) {
            const int j = idx >> 10, e = idx & 1023;              // column block, element (row e >> 5, column e & 31)
            float sum = lds[j * 1024 + e];
            for (int k = 1; k < tstride; ++k) sum += lds[(k * nj + j) * 1024 + e];
            as_global(dst)[(long)(t.m0 + (e >> 5)) * L.rpad + 128 * b + j * 32 + (e & 31)] = osc * sum;
        }
    }
}

// T = scale * (sum of the K-range slabs, in range order) for the layers that were split (deterministic; a few MB in all)
constexpr int LR_REDUCE_CHUNK = 1024;
__global__ __launch_bounds__(256) void nsgp_lr_reduce_kernel(const ChunkDev* __restrict__ chunks, const LayerDev* __restrict__ layers,
                                                             const DynBlock* __restrict__ dyn, int optimizer) {
    const ChunkDev c = chunks[blockIdx.x];
    const LayerDev L = layers[c.tensor];
    const float scale = (optimizer == NSGP_OPT_SGD) ? -dyn->hyper[L.hyper].lr : 1.0f;
    const long n = (long)L.rows * L.rpad, i = c.start + (long)threadIdx.x * 4;      // n % 1024 == 0 (rows % 32 == 0, rpad % 32 == 0)
    f32x4 part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < L.nsplit) part[k] = *(const gf32x4*)(L.slabs + (long)k * n + i);    // all loads in flight together
    f32x4 sum = part[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (k < L.nsplit) sum += part[k];
    *(gf32x4*)(L.T + i) = scale * sum;
}

template <int OPT, int MAXR>
__global__ __launch_bounds__(256) void nsgp_lr_apply_kernel(const TileDev* __restrict__ units, const LayerDev* __restrict__ layers,
                                                            const DynBlock* __restrict__ dyn) {
    __shared__ __attribute__((aligned(16))) float t_lds[32 * (MAXR + 4)];             // T[m0 .. m0+32][rpad], rows padded by 4 floats (MAXR = 256: the wide class's launch)
    __shared__ __attribute__((aligned(16))) float w_lds[4 * LR_TILE];                  // per wave: a U chunk, then the output block
    const TileDev t = units[blockIdx.x];           // m0, n0, pad = columns of this workgroup (<= 256, a multiple of 32)
    const LayerDev L = layers[t.layer];
    const float* A;
    float scale;
    lowrank_source<OPT>(L, dyn, A, scale);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5, r8 = lane >> 3, ch = lane & 7;
    const float c = L.basis_scale;
    const int NJ = L.rpad >> 5, tld = L.rpad + 4;
    // The MFMA computes the TRANSPOSED block C[n][m] = sum_k U[n][k] T[m][k] (A operand = U rows, B operand = T rows): lane
    // (i, h) then holds, for row m = m0 + i, the columns n + 8 g + 4 h + (0..3), g = 0..3 -- four 16-byte pieces for the tile.
    // The four waves take ADJACENT 32-column blocks (512 contiguous bytes per row at a time), then the next 128 columns
    // (at most two blocks per wave).  The S / p loads of a wave's first block are issued BEFORE the T tile is staged and those of
    // its second block before the first is computed: the HBM latency of the stream never sits behind an LDS stage or the MFMAs.
    const long r8s = 8L * L.cols;
    const long off0 = (long)(t.m0 + r8) * L.cols + t.n0 + 32 * wave + 4 * ch;
    const bool has0 = 32 * wave < t.pad, has1 = 32 * wave + 128 < t.pad;
    f32x4 av0[4], pv0[4], av1[4], pv1[4];
    if (has0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { av0[it] = load4_a4_nt(A + off0 + it * r8s); pv0[it] = load4_nt(L.p + off0 + it * r8s); }
    }
    for (int x = threadIdx.x; x < 8 * L.rpad; x += 256) {                               // 32 rows x rpad / 4 pieces, contiguous in memory
        const int row = x / (L.rpad >> 2), c4 = x - row * (L.rpad >> 2);
        lds_put4(t_lds, row * tld + 4 * c4, *(const gf32x4*)(L.T + (long)t.m0 * L.rpad + 4L * x));
    }
    if (has1) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { av1[it] = load4_a4_nt(A + off0 + 128 + it * r8s); pv1[it] = load4_nt(L.p + off0 + 128 + it * r8s); }
    }
    __syncthreads();
    float* tile = w_lds + wave * LR_TILE;
    auto block = [&](int n, long off, const f32x4 (&av)[4], const f32x4 (&pv)[4]) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
        const float* ubase = L.urm + (long)(n + r8) * L.rpad + 4 * ch;
        for (int jb = 0; jb < NJ; ++jb) {                              // 32 k at a time
            f32x4 u4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) u4[it] = *(const gf32x4*)(ubase + (long)(8 * it) * L.rpad + 32 * jb);
#pragma unroll
            for (int it = 0; it < 4; ++it) lds_put4(tile, (r8 + 8 * it) * LR_TILE_LD + 4 * ch, u4[it]);
            f32x4 fu[4], ft[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fu[q] = lds_get4(tile, i * LR_TILE_LD + 4 * (2 * q + h));              // U[n + i][32 jb + 8 q + 4 h ..]
                ft[q] = lds_get4(t_lds, i * tld + 32 * jb + 4 * (2 * q + h));          // T[m0 + i][32 jb + 8 q + 4 h ..]
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fu[q][e], ft[q][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) lds_put4(tile, i * LR_TILE_LD + 8 * g + 4 * h, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f32x4 cv = lds_get4(tile, (r8 + 8 * it) * LR_TILE_LD + 4 * ch);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pv[it][e] + c * (scale * av[it][e] - cv[e]);
            store4_nt(L.p + off + it * r8s, o);
        }
    };
    if (has0) block(t.n0 + 32 * wave, off0, av0, pv0);
    if (has1) block(t.n0 + 32 * wave + 128, off0 + 128, av1, pv1);
}

// ---- host: plan ---------------------------------------------------------------
template <typename K>
static int enable_big_lds(K kernel) {
    NSGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
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
    int split_kind = 0;         // dense fast tiles: 0 fp32 MFMA, 2 two-term fp16 split (1 was the three-term bf16 split, removed in ABI 8)
    double gemm_flops = 0, bytes = 0;
    TensorDev* d_tensors = nullptr;
    LayerDev* d_layers = nullptr;
    TileDev* d_tiles = nullptr;  // dense fast tiles | dense generic tiles | low-rank phase-1 | phase-2
    int n_tiles_lr1 = 0, n_tiles_lr2 = 0, n_lowrank = 0;    // low-rank: workgroups of the fused update + T launches and of the apply launch
    int n_chunks_lr = 0;
    int n_chunks_plain = 0;       // chunks of the un-projected tensors appended to the fused low-rank launch (0 without low-rank layers)
    ChunkDev* d_chunks_plain = nullptr;
    // cache blocking of the low-rank launches: the layers are cut into groups whose (update, p) working set fits the 256 MB
    // Infinity Cache, and fused -> reduce -> apply run group by group, so that the apply launch's re-read of the update and of p
    // (8 of its 12 bytes per element) is served by the memory-side cache instead of HBM.  Offsets / counts into d_tiles, d_chunks_lr.
    struct LrGroup { int lr1_off, lr1_n, lr2_off, lr2_n, chunk_off, chunk_n, wide; };      // wide: the rpad = 256 layers (their own kernels)
    std::vector<LrGroup> lr_groups;
    ChunkDev* d_chunks_lr = nullptr;
    double lowrank_flops = 0;
    ChunkDev* d_chunks = nullptr;
    size_t dyn_bytes = 0;
    char* h_dyn[NSLOT] = {nullptr, nullptr, nullptr, nullptr};  // pinned
    char* d_dyn[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_used[NSLOT] = {false, false, false, false};
    int slot = 0;
    // optional per-launch timing: prof_per_step events per recorded step (before the elementwise launch, after it, after the dense
    // GEMM launches, and per low-rank group after its fused update + T launch, its slab reduce and its apply launch)
    std::vector<hipEvent_t> prof_ev;
    int prof_cap = 0, prof_n = 0, prof_per_step = 0;
    float prof_detail[5] = {0, 0, 0, 0, 0};   // averages of the last profile_end: elementwise, fused update + T, dense GEMM, slab reduce, low-rank apply
};

static bool tensor_fast(const nsgp_tensor_t& t) {
    return t.rows % BM == 0 && t.cols % BN == 0 && t.cols % BK == 0 && aligned16(t.proj);
}
// the fp16-split kernel takes a layer when it has the fast shape and carries a kind-2 split copy of its projector
static bool tensor_v2(const nsgp_tensor_t& t) {
    return tensor_fast(t) && t.split_kind == 2 && t.proj_split && aligned16(t.proj_split) && aligned16(t.param) && aligned16(t.state0);
}

static size_t pad256(size_t x) { return (x + 255) & ~(size_t)255; }

// low-rank form: a head-form projector (the caller vouches that proj == basis_scale * (I - U U^T)), r <= 128, 32-aligned shape
static bool tensor_lowrank(const nsgp_tensor_t& t) {
    return t.basis && t.basis_rows && t.rank > 0 && t.rank <= LR_MAX_RANK && t.rows % 32 == 0 && t.cols % 32 == 0 &&
           aligned16(t.basis) && aligned16(t.basis_rows) && t.proj && aligned16(t.param) && aligned16(t.state0) && aligned16(t.state1) &&
           aligned16(t.state2);
}
static int lr_rpad(int rank) { return rank <= 32 ? 32 : (rank <= 64 ? 64 : (rank <= 128 ? 128 : 256)); }   // U's padded width: 32, 64, 128 or 256 columns
// K ranges of a low-rank layer in the fused update + T launch: ~8 groups of 32 columns per workgroup (2 per wave), at most 16
// ranges.  (32 groups per workgroup left the two wide rank classes of R-50-FPN with 240 workgroups each -- under one per CU --
// and 78 us apiece; the slabs this costs are a few MB.)
static int lr_nsplit(int cols, int rank) {
    (void)rank;
    const int groups = cols / 32;
    return std::max(1, std::min(16, (groups + 7) / 8));
}
static size_t lr_workspace(const nsgp_tensor_t& t) {
    const size_t one = pad256((size_t)t.rows * lr_rpad(t.rank) * 4);
    const int S = lr_nsplit(t.cols, t.rank);
    return one + (S > 1 ? (size_t)S * one : 0);
}

extern "C" size_t nsgp_plan_workspace_bytes(const nsgp_tensor_t* tensors, int n, int optimizer) {
    if (!tensors) return 0;
    size_t s = 0;
    for (int i = 0; i < n; ++i) {
        const nsgp_tensor_t& t = tensors[i];
        if (!t.proj) continue;
        if (optimizer == NSGP_OPT_ADAM) s += pad256((size_t)t.numel * 4);
        if (tensor_v2(t) && !tensor_lowrank(t)) s += pad256(v2_operand_bytes(t.rows, t.cols)) + pad256((size_t)t.rows * 4);
        if (tensor_lowrank(t)) s += lr_workspace(t);
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
    std::vector<ChunkDev> cd, cd_plain;       // chunks of the projected tensors on a dense path | of the un-projected tensors
    std::vector<char> layer_fast;
    bool all_split = true;     // every fast dense layer carries a split copy of its projector, all of the same kind
    int split_kind = 0;
    double flops = 0, bytes = 0, lr_flops = 0;
    int n_lowrank = 0;
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
            LayerDev L{};
            L.p = t.param; L.s0 = t.state0; L.u = d.u; L.proj = t.proj; L.tensor = i; L.rows = t.rows; L.cols = t.cols; L.hyper = t.hyper;
            L.basis_scale = 1.0f; L.split = t.proj_split; L.split_kind = t.split_kind;
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
                L.ukq = t.basis;
                L.urm = t.basis_rows;
                L.rank = t.rank;
                L.rpad = lr_rpad(t.rank);
                L.basis_scale = t.basis_scale;
                L.nsplit = lr_nsplit(t.cols, t.rank);
                const size_t one = pad256((size_t)t.rows * L.rpad * 4);
                L.T = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off);
                if (L.nsplit > 1) L.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off + one);
                ws_off += lr_workspace(t);
                lr_flops += 4.0 * t.rows * (double)t.cols * t.rank;
                ++n_lowrank;
            }
            ld.push_back(L);
            layer_fast.push_back(tensor_fast(t) && aligned16(t.param) && aligned16(t.state0) ? 1 : 0);
            if (layer_fast.back() && !tensor_lowrank(t)) {
                const bool ok = t.proj_split && aligned16(t.proj_split) && t.split_kind == 2;
                if (!ok) all_split = false;
                else if (split_kind == 0) split_kind = t.split_kind;
                else if (split_kind != t.split_kind) all_split = false;
            }
            flops += 2.0 * t.rows * (double)t.cols * t.cols;
            bytes += 4.0 * (double)t.cols * t.cols;
        }
        bytes += 5.0 * 4.0 * (double)t.numel;
        td[i] = d;
        if (t.proj && tensor_lowrank(t)) continue;                                                           // updated by the fused update + T launch
        if (d.a_split) for (long s = 0; s < t.numel; s += 8L * t.cols) cd.push_back(ChunkDev{i, 1, s});   // 8-row bands
        else for (long s = 0; s < t.numel; s += CHUNK) (t.proj ? cd : cd_plain).push_back(ChunkDev{i, 0, s});
    }
    // with low-rank layers in the plan the un-projected tensors' chunks ride at the end of the fused launch's grid; otherwise they
    // belong to the multi-tensor launch as before
    if (n_lowrank == 0) {
        cd.insert(cd.end(), cd_plain.begin(), cd_plain.end());
        cd_plain.clear();
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
    P->n_chunks_plain = (int)cd_plain.size();
    P->n_tiles_fast = (int)fast_tiles.size();
    P->n_tiles_v2 = (int)v2_tiles.size();
    P->split_kind = (all_split && !(fast_tiles.empty() && v2_tiles.empty())) ? split_kind : 0;
    P->n_tiles_generic = (int)gen_tiles.size();
    P->gemm_flops = flops;
    P->bytes = bytes;
    // low-rank units.  Fused update + T launch: one workgroup per (32-row block, K range); apply launch: one workgroup per
    // (32 rows x <= 256 columns).  All the row blocks of one K range (one column strip) read the SAME slice of U: a strip's
    // workgroups go to one of 8 queues (the shortest at the time), and the queues are interleaved round-robin, so that they
    // share blockIdx % 8 = one XCD under the observed round-robin placement and the slice is fetched into ONE L2 instead of up to
    // eight (speed only, never correctness; measured 127 vs 128-134 us and 81 vs 83 us on the R-50 table: inside the run-to-run
    // spread).  Layers in descending cost order.
    std::vector<TileDev> lr1, lr2;
    std::vector<ChunkDev> lr_chunks;      // slab reduce of the layers with more than one K range
    std::vector<nsgp_plan::LrGroup> lr_group_recs;
    {
        std::vector<int> lo;
        for (size_t li = 0; li < ld.size(); ++li)
            if (ld[li].rank > 0) lo.push_back((int)li);
        auto xcd_interleave = [](std::vector<TileDev> (&q)[8], std::vector<TileDev>& dst) {
            size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0;
            for (auto& v : q) total += v.size();
            while (dst.size() < total) {
                for (int x = 0; x < 8 && dst.size() < total; ++x) {
                    int src = x;
                    if (pos[src] >= q[src].size()) {      // queue ran dry: borrow from the fullest
                        size_t best = 0;
                        src = -1;
                        for (int y = 0; y < 8; ++y)
                            if (q[y].size() - pos[y] > best) { best = q[y].size() - pos[y]; src = y; }
                        if (src < 0) break;
                    }
                    dst.push_back(q[src][pos[src]++]);
                }
            }
        };
        auto shortest = [](std::vector<TileDev> (&q)[8]) {
            int best = 0;
            for (int y = 1; y < 8; ++y)
                if (q[y].size() < q[best].size()) best = y;
            return best;
        };
        constexpr int lra_cols = LRA_COLS;      // (128-column apply workgroups -- one block per wave, twice the workgroups -- measured 0.098 vs 0.094 ms: round 3)
        auto t_cost = [&](int li) { return (long)ld[li].cols * ld[li].rpad / ld[li].nsplit; };
        std::stable_sort(lo.begin(), lo.end(), [&](int a, int b) { return t_cost(a) > t_cost(b); });
        // layer groups (cache blocking, see nsgp_plan::lr_groups): contiguous runs of the cost-sorted layer list with about equal
        // element counts; one group = today's three launches
        const int G = std::max(1, std::min(std::min(16, g_lr_groups), (int)std::max<size_t>(1, lo.size())));
        double total_el = 0;
        for (int li : lo) total_el += (double)ld[li].rows * ld[li].cols;
        std::vector<std::vector<int>> groups(G + 1);      // the last group: the wide rank class (rpad = 256), launched with its own kernels
        {
            double seen = 0;
            for (int li : lo) {
                if (ld[li].rpad > 128) { groups[G].push_back(li); continue; }
                const int gi = std::min(G - 1, (int)(seen * G / std::max(total_el, 1.0)));
                groups[gi].push_back(li);
                seen += (double)ld[li].rows * ld[li].cols;
            }
        }
        for (auto& grp : groups) {
            if (grp.empty()) continue;
            nsgp_plan::LrGroup rec{(int)lr1.size(), 0, (int)lr2.size(), 0, (int)lr_chunks.size(), 0, &grp == &groups.back() ? 1 : 0};
            {
                std::vector<TileDev> q[8], part;
                for (int li : grp)
                    for (int sp = 0; sp < ld[li].nsplit; ++sp) {
                        std::vector<TileDev>& dq = q[shortest(q)];
                        for (int m = 0; m < ld[li].rows; m += 32) dq.push_back(TileDev{li, m, 0, sp});
                    }
                xcd_interleave(q, part);
                lr1.insert(lr1.end(), part.begin(), part.end());
            }
            for (int li : grp)
                if (ld[li].nsplit > 1)
                    for (long st = 0; st < (long)ld[li].rows * ld[li].rpad; st += LR_REDUCE_CHUNK) lr_chunks.push_back(ChunkDev{li, 0, st});
            std::vector<int> by_rpad(grp);
            std::stable_sort(by_rpad.begin(), by_rpad.end(), [&](int a, int b) { return ld[a].rpad > ld[b].rpad; });
            {
                std::vector<TileDev> q[8], part;
                for (int li : by_rpad)
                    for (int n0 = 0; n0 < ld[li].cols; n0 += lra_cols) {
                        std::vector<TileDev>& dq = q[shortest(q)];
                        for (int m = 0; m < ld[li].rows; m += 32) dq.push_back(TileDev{li, m, n0, std::min(lra_cols, ld[li].cols - n0)});
                    }
                xcd_interleave(q, part);
                lr2.insert(lr2.end(), part.begin(), part.end());
            }
            rec.lr1_n = (int)lr1.size() - rec.lr1_off;
            rec.lr2_n = (int)lr2.size() - rec.lr2_off;
            rec.chunk_n = (int)lr_chunks.size() - rec.chunk_off;
            lr_group_recs.push_back(rec);
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
    P->lr_groups = lr_group_recs;
    P->prof_per_step = 3 + 3 * (int)lr_group_recs.size();
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
    if (!cd.empty()) {
        PLAN_HIP(hipMalloc(&P->d_chunks, sizeof(ChunkDev) * cd.size()));
        PLAN_HIP(hipMemcpy(P->d_chunks, cd.data(), sizeof(ChunkDev) * cd.size(), hipMemcpyHostToDevice));
    }
    if (!cd_plain.empty()) {
        PLAN_HIP(hipMalloc(&P->d_chunks_plain, sizeof(ChunkDev) * cd_plain.size()));
        PLAN_HIP(hipMemcpy(P->d_chunks_plain, cd_plain.data(), sizeof(ChunkDev) * cd_plain.size(), hipMemcpyHostToDevice));
    }
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
    if ((rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_SGD, true>)) || (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_SGD, false>)) ||
        (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_ADAM, true>)) || (rc = enable_big_lds(nsgp_project_kernel<NSGP_OPT_ADAM, false>)) ||
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
    if (P->d_chunks_plain) keep(hipFree(P->d_chunks_plain), "hipFree(chunks_plain)");
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

extern "C" int nsgp_plan_launch_shape(const nsgp_plan_t* P, int* shape5) {
    if (!P || !shape5) return fail(NSGP_ERR_INVALID, "nsgp_plan_launch_shape: null argument");
    shape5[0] = P->n_chunks;            // workgroups of the multi-tensor update launch (0 = no such launch)
    shape5[1] = P->n_tiles_lr1;         // low-rank (32-row block, K range) units of the fused update + T launch
    shape5[2] = P->n_chunks_plain;      // chunks of the un-projected tensors appended to that launch
    shape5[3] = P->n_chunks_lr;         // workgroups of the slab reduce
    shape5[4] = P->n_tiles_lr2;         // workgroups of the apply launch
    return NSGP_OK;
}

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
    hipEvent_t* pe = prof ? &P->prof_ev[(size_t)P->prof_per_step * P->prof_n] : nullptr;
    int ei = 0;
    if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));

    if (P->n_chunks > 0) {
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL(nsgp_update_kernel<NSGP_OPT_SGD>, dim3(P->n_chunks), dim3(256), 0, stream, P->d_chunks, P->d_tensors, d);
        else
            hipLaunchKernelGGL(nsgp_update_kernel<NSGP_OPT_ADAM>, dim3(P->n_chunks), dim3(256), 0, stream, P->d_chunks, P->d_tensors, d);
        NSGP_LAUNCH_CHECK();
    }
    if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));
    // dense projection of the layers that are not on the low-rank form (their update was written by the launch above)
    if (P->n_tiles_v2 > 0) {
        const TileDev* vt = P->d_tiles + P->n_tiles_fast + P->n_tiles_generic + P->n_tiles_lr1 + P->n_tiles_lr2;
        if (P->optimizer == NSGP_OPT_SGD)
            hipLaunchKernelGGL(nsgp_project_v2_kernel<NSGP_OPT_SGD>, dim3(P->n_tiles_v2), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, vt, P->d_layers, d);
        else
            hipLaunchKernelGGL(nsgp_project_v2_kernel<NSGP_OPT_ADAM>, dim3(P->n_tiles_v2), dim3(V2L_THREADS), V2_SMEM_BYTES, stream, vt, P->d_layers, d);
        NSGP_LAUNCH_CHECK();
    }
    if (P->n_tiles_fast > 0) {
        if (P->optimizer == NSGP_OPT_SGD)
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
    if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));
    // low-rank layers, group by group: update fused with T = u U -> ordered slab reduce -> apply
    const TileDev* t1 = P->d_tiles + P->n_tiles_fast + P->n_tiles_generic;
    const TileDev* t2 = t1 + P->n_tiles_lr1;
    for (const nsgp_plan::LrGroup& g : P->lr_groups) {
        const int extra = (&g == &P->lr_groups.front()) ? P->n_chunks_plain : 0;      // the un-projected tensors ride in the first group's launch
#define LR_FUSED(OPTV, NBV)                                                                                                                      \
    hipLaunchKernelGGL((nsgp_update_lr_kernel<OPTV, NBV>), dim3(g.lr1_n + extra), dim3(256), 0, stream, t1 + g.lr1_off, P->d_layers, P->d_tensors, d, \
                       g.lr1_n, P->d_chunks_plain)
        if (P->optimizer == NSGP_OPT_SGD) { if (g.wide) LR_FUSED(NSGP_OPT_SGD, 2); else LR_FUSED(NSGP_OPT_SGD, 1); }
        else { if (g.wide) LR_FUSED(NSGP_OPT_ADAM, 2); else LR_FUSED(NSGP_OPT_ADAM, 1); }
#undef LR_FUSED
        NSGP_LAUNCH_CHECK();
        if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));
        if (g.chunk_n > 0) {
            hipLaunchKernelGGL(nsgp_lr_reduce_kernel, dim3(g.chunk_n), dim3(256), 0, stream, P->d_chunks_lr + g.chunk_off, P->d_layers, d, P->optimizer);
            NSGP_LAUNCH_CHECK();
        }
        if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));
#define LR_APPLY(OPTV, MAXRV) hipLaunchKernelGGL((nsgp_lr_apply_kernel<OPTV, MAXRV>), dim3(g.lr2_n), dim3(256), 0, stream, t2 + g.lr2_off, P->d_layers, d)
        if (P->optimizer == NSGP_OPT_SGD) { if (g.wide) LR_APPLY(NSGP_OPT_SGD, 256); else LR_APPLY(NSGP_OPT_SGD, 128); }
        else { if (g.wide) LR_APPLY(NSGP_OPT_ADAM, 256); else LR_APPLY(NSGP_OPT_ADAM, 128); }
#undef LR_APPLY
        NSGP_LAUNCH_CHECK();
        if (prof) NSGP_HIP(hipEventRecord(pe[ei++], stream));
    }
    if (prof) ++P->prof_n;
    NSGP_HIP(hipEventRecord(P->ev[s], stream));
    P->ev_used[s] = true;
    return NSGP_OK;
}

extern "C" int nsgp_plan_profile_begin(nsgp_plan_t* P, int max_steps) {
    if (!P || max_steps < 0 || max_steps > 4096) return fail(NSGP_ERR_INVALID, "nsgp_plan_profile_begin: bad argument");
    while ((int)P->prof_ev.size() < P->prof_per_step * max_steps) {
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
    // per step: e[0] start, e[1] after the plain launch, e[2] after the dense launches, then per low-rank group: after the fused
    // update + T launch, after the slab reduce, after the apply launch.
    // detail: [0] plain launch, [1] fused launches, [2] dense GEMM, [3] slab reduces, [4] apply launches (summed over the groups);
    // update_ms = [0] + [1], gemm_ms = the rest
    double det[5] = {0, 0, 0, 0, 0};
    const int G = (int)P->lr_groups.size();
    for (int i = 0; i < P->prof_n; ++i) {
        hipEvent_t* e = &P->prof_ev[(size_t)P->prof_per_step * i];
        NSGP_HIP(hipEventSynchronize(e[P->prof_per_step - 1]));
        float ms = 0;
        // an interval without a launch measures only the event records themselves (~5 us each): not counted
        if (P->n_chunks > 0) {
            NSGP_HIP(hipEventElapsedTime(&ms, e[0], e[1]));
            det[0] += ms;
        }
        if (P->n_tiles_v2 + P->n_tiles_fast + P->n_tiles_generic > 0) {
            NSGP_HIP(hipEventElapsedTime(&ms, e[1], e[2]));
            det[2] += ms;
        }
        for (int g = 0; g < G; ++g) {
            hipEvent_t* ge = e + 2 + 3 * g;
            NSGP_HIP(hipEventElapsedTime(&ms, ge[0], ge[1]));
            det[1] += ms;
            NSGP_HIP(hipEventElapsedTime(&ms, ge[1], ge[2]));
            det[3] += ms;
            NSGP_HIP(hipEventElapsedTime(&ms, ge[2], ge[3]));
            det[4] += ms;
        }
    }
    for (int k = 0; k < 5; ++k) P->prof_detail[k] = P->prof_n ? (float)(det[k] / P->prof_n) : 0.0f;
    if (n_steps) *n_steps = P->prof_n;
    if (update_ms_avg) *update_ms_avg = P->prof_detail[0] + P->prof_detail[1];
    if (gemm_ms_avg) *gemm_ms_avg = P->prof_detail[2] + P->prof_detail[3] + P->prof_detail[4];
    P->prof_cap = 0;
    P->prof_n = 0;
    return NSGP_OK;
}

extern "C" int nsgp_plan_profile_detail(const nsgp_plan_t* P, float* ms5) {
    if (!P || !ms5) return fail(NSGP_ERR_INVALID, "nsgp_plan_profile_detail: null argument");
    for (int k = 0; k < 5; ++k) ms5[k] = P->prof_detail[k];
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
