/*
 * nsgp_repre.h -- C ABI of the MI355X-native NSGP-RePRE hot path (libnsgp_repre_hip.so)
 *
 * Drop-in boundary for the path SURVEY.md section 8 scopes.  The reference
 * (yyl404/NSGP-RePRE) is pure Python on PyTorch; each entry point below replaces
 * the torch call sequence at the cited reference file:line (paths relative to
 * the reference repo root).  Plain pointers and sizes only: no torch types.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to contiguous fp32 unless noted;
 *     the caller owns all buffers, the library borrows them for the call;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream) and returns without synchronising;
 *   - return value: NSGP_OK (0) or a negative NSGP_ERR_* code; no exceptions
 *     cross the ABI; nsgp_last_error() gives a thread-local message;
 *   - no allocation on the hot calls: step plans own their small device
 *     tables (allocated in nsgp_plan_create), big workspaces are passed in.
 */
#ifndef NSGP_REPRE_H
#define NSGP_REPRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSGP_OK 0
#define NSGP_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, misaligned) */
#define NSGP_ERR_HIP (-2)       /* a HIP runtime call failed */
#define NSGP_ERR_WORKSPACE (-3) /* workspace too small */
#define NSGP_ERR_LIMIT (-4)     /* a fixed capacity was exceeded (e.g. > NSGP_MAX_HYPER combos) */

#define NSGP_ABI_VERSION 8
#define NSGP_MAX_HYPER 32 /* distinct hyper-parameter sets per plan step */

int nsgp_abi_version(void);
const char* nsgp_last_error(void);
/* Number of HIP devices visible / name of the current one ("" without a GPU). */
int nsgp_device_count(void);
int nsgp_device_arch(char* buf, int buflen);
/* Test-run diagnostics (no reference counterpart): from now on a SIGABRT first writes the native call stack of the
 * aborting thread to fd 2, then runs the handler that was installed before (e.g. Python's faulthandler).  Idempotent.
 * Environment: NSGP_DEBUG_ALLOC=1 makes nsgp_plan_create / nsgp_plan_destroy print the addresses of the plan's own device and
 * pinned buffers to stderr (to place a faulting address reported by the runtime). */
int nsgp_debug_install_abort_backtrace(void);

/* ------------------------------------------------------------------------
 * Projected optimizer step  (K1 + K2 of SURVEY section 2.2)
 *
 * Replaces the per-parameter Python loop of
 *   SGDNSCL.step      mmdet/engine/optimizers/SGD_NSCL.py:59-96   (+ get_update :387-415)
 *   AdamWNSCL.step    mmdet/engine/optimizers/AdamW_NSCL.py:66-103 (+ get_update :212-250)
 *   AdamNSCL.step     mmdet/engine/optimizers/Adam_NSCL.py:66-102  (+ get_update :207-247)
 *   SGDNSCLNA.step    mmdet/engine/optimizers/SGD_NSCL_NoAdaptive.py:59-111
 * by a handful of launches: one multi-tensor elementwise kernel over every listed
 * tensor (momentum / Adam moments / weight decay, `p += update` for the
 * un-projected ones), and for the projected tensors either
 *   - the low-rank form `p += c (u - (u U) U^T)` for layers described with `basis` / `basis_rows`
 *     (head-form projectors, at most 256 removed directions: rank classes 32 / 64 / 128 share one pair of launches, 129 .. 256 have their own): their elementwise update fused with
 *     T = u U, a small ordered slab reduce, and the apply launch -- HBM-bound, exact fp32 MFMA; or
 *   - one grouped MFMA GEMM `p += update.view(Cout,D) @ P` (the `torch.mm(update.view(Cout,-1), P)`
 *     of SGD_NSCL.py:85-90) for every other projected tensor.
 * ------------------------------------------------------------------------ */

typedef struct nsgp_plan nsgp_plan_t;

enum { NSGP_OPT_SGD = 0, NSGP_OPT_ADAM = 1 };

/* Static description of one parameter tensor (one (name, p) pair of the
 * reference's param_groups[i]['names'/'params']). */
typedef struct {
    float* param;       /* p.data, numel fp32 */
    float* state0;      /* SGD: state['previous_grad'];  Adam: state['exp_avg'] */
    float* state1;      /* Adam: state['exp_avg_sq'];    SGD: NULL */
    float* state2;      /* Adam amsgrad: state['max_exp_avg_sq'] or NULL */
    const float* proj;  /* transforms[name]: [cols x cols] row-major, or NULL = not projected */
    int64_t numel;
    int32_t rows;       /* Cout  = update.size(0)            (projected tensors only) */
    int32_t cols;       /* D     = numel / rows = Cin*kh*kw  (projected tensors only) */
    int32_t hyper;      /* index into the per-step hyper array */
    int32_t rank;       /* low-rank form (optional): number of REMOVED directions r (the top eigenvectors = first_col of the projector) */
    const float* basis; /* low-rank form (optional, ABI 7): U as k-quads [cols/4][rpad][4] fp32, rpad = 32 / 64 / 128 (the smallest >= rank),
                           columns >= rank zero (element (k, j) at ((k/4)*rpad + j)*4 + k%4).  With `basis_rows` non-NULL,
                           0 < rank <= 128, rows % 32 == 0 and cols % 32 == 0 the step applies
                           p += basis_scale * (u - (u U) U^T)      (the north star's g - U (U^T g))
                           in 4*Cout*D*r FLOP and no projector traffic instead of u @ proj.  The CALLER vouches that
                           proj == basis_scale * (I - U U^T) as built by nsgp_build_projector_head from the same U;
                           otherwise leave both NULL and the dense `proj` is used */
    float basis_scale;  /* 1/||I - U U^T||_F for Frobenius-normalised projectors (SGD_NSCL.py:282-283), else 1 */
    int32_t split_kind; /* 0: no split copy; 2: proj_split holds the pre-tiled, column-scaled two-term fp16 split
                           (nsgp_split_projector_f16).  (1 was a three-term bf16 split: slower than kind 2 on every table and never a
                           default; removed in ABI 8.) */
    const void* proj_split; /* optional split copy of proj^T.  When every 128-aligned projected tensor of a plan carries one
                           of the same kind, the dense projection runs on the low-precision matrix cores with fp32 accumulation
                           and fp32-level error PER OUTPUT ROW: three fp16 MFMAs per product (a0b0+a0b1+a1b0) with one power-of-two scale per ROW of the
                           update (found by the elementwise launch of the same step, which also writes the update's split
                           copy into the plan workspace) and one per COLUMN of the projector (stored behind the split copy).
                           NULL = fp32 MFMA */
    float split_scale;  /* unused since ABI 6 (the fp16 split carries its per-column scales itself); keep 0 */
    int32_t reserved;
    const float* basis_rows; /* low-rank form (ABI 7): the same U row-major [cols][rpad] fp32 */
} nsgp_tensor_t;

/* Per-step hyper-parameters of one param group (host values, fp64->fp32 as torch does). */
typedef struct {
    float lr;            /* SGD: group['lr'].  Adam: UNUSED for the moment update (see step_size) */
    float momentum;      /* SGD */
    float one_minus_dampening; /* SGD: (1 - group['dampening']) computed in double on the host */
    float weight_decay;  /* SGD/Adam: L2 folded into grad.  AdamW: see decoupled_decay */
    float beta1, beta2, one_minus_beta1, one_minus_beta2, eps; /* Adam */
    float step_size;     /* Adam: lr*sqrt(1-beta2^t)/(1-beta1^t), computed in double on the host */
    float decoupled_decay; /* AdamW: lr*weight_decay (AdamW_NSCL.py:87), else 0 */
    int32_t nesterov;    /* SGD */
    int32_t first_step;  /* SGD: state['step']==1 -> buf = grad (SGD_NSCL.py:403-406) */
    int32_t amsgrad;     /* Adam */
    int32_t write_grad;  /* 1 = mirror the reference's in-place mutation of p.grad */
    int32_t reserved;
} nsgp_hyper_t;

/* Device bytes of workspace a plan needs (Adam kinds: the projected updates; low-rank layers: the
 * [Cout x r] intermediates and their split-K slabs). */
size_t nsgp_plan_workspace_bytes(const nsgp_tensor_t* tensors, int n_tensors, int optimizer);

/* Build a plan: validates shapes, builds the cost-sorted, XCD-interleaved tile
 * table of the grouped GEMM and the chunk table of the elementwise kernel, and
 * uploads them.  `workspace` (device, >= nsgp_plan_workspace_bytes) is borrowed
 * for the plan's lifetime.  Allocates a few hundred KB of device/pinned memory. */
int nsgp_plan_create(nsgp_plan_t** plan, const nsgp_tensor_t* tensors, int n_tensors,
                     int optimizer, void* workspace, size_t workspace_bytes);
/* Waits for the plan's own launches (its per-step events), then releases its device tables, pinned ring and events.
 * Returns NSGP_ERR_HIP (first failing call in nsgp_last_error) if any release fails; the plan is gone either way.
 * Call it from the thread that drives the stream -- never from a finaliser (the Python layer parks handles of
 * garbage-collected optimizers and releases them at its next explicit entry point). */
int nsgp_plan_destroy(nsgp_plan_t* plan);

/* One optimizer step.  `grads[i]` = p.grad.data of tensor i (device pointers in a HOST
 * array; may change from step to step).  `hyper` = n_hyper host structs.  Stream-ordered. */
int nsgp_plan_step(nsgp_plan_t* plan, float* const* grads, const nsgp_hyper_t* hyper,
                   int n_hyper, void* stream);

/* Introspection for measurement: algorithmic FLOPs / bytes of one step of this plan
 * (sum 2*Cout*D^2; sum 4*D^2 + 5*4*numel). */
int nsgp_plan_stats(const nsgp_plan_t* plan, double* gemm_flops, double* algorithmic_bytes,
                    int* n_tiles, int* n_projected);

/* Layers that take the low-rank form and their FLOPs (sum 4*Cout*D*r); workgroups of its T = u U launch (one per 32-row
 * block) and wave units of its apply launch (32 rows x <= 256 columns each, padded to a multiple of four). */
int nsgp_plan_lowrank_stats(const nsgp_plan_t* plan, int* n_lowrank, double* lowrank_flops,
                            int* n_tiles_p1, int* n_tiles_p2);
/* Kind of split the plan's dense projection launch uses: 0 = fp32 MFMA, 1 = three-term bf16, 2 = two-term fp16. */
int nsgp_plan_uses_split_mfma(const nsgp_plan_t* plan);
/* Dense-projection tiles of the plan by kernel: whole 128 x 128 tiles of the fp32-MFMA / bf16-split kernel, guarded 128 x 128
 * tiles (ragged or misaligned layers, fp32 MFMA), and 256 x 128 (or 128 x 128) tiles of the fp16-split kernel. */
int nsgp_plan_tile_counts(const nsgp_plan_t* plan, int* fast_128, int* generic_128, int* split_f16_256);

/* Per-launch timing with HIP events recorded on the launch stream (measurement only): between
 * _begin and _end each nsgp_plan_step records 6 events; _end synchronises on them and returns
 * the average duration of the elementwise launch and of the projection launches (dense GEMM and / or low-rank) together. */
int nsgp_plan_profile_begin(nsgp_plan_t* plan, int max_steps);
int nsgp_plan_profile_end(nsgp_plan_t* plan, int* n_steps, float* update_ms_avg, float* gemm_ms_avg);
/* The last _end launch by launch (ABI 7), averages in ms over the profiled steps, 0 for launches a plan does not make:
 * ms5[0] the multi-tensor elementwise launch, [1] the fused update + T = u U launch of the low-rank layers, [2] the dense GEMM
 * launch(es), [3] the slab reduce of the low-rank T, [4] the low-rank apply launch.  (_end's update_ms = [0] + [1], gemm_ms = the rest.) */
int nsgp_plan_profile_detail(const nsgp_plan_t* plan, float* ms5);
/* Launch shape of a step (ABI 8), workgroups: [0] the multi-tensor update launch (0 = that launch is not made), [1] the low-rank
 * units of the fused update + T launch, [2] the chunks of the UN-PROJECTED tensors that ride at the end of that launch's grid (with
 * low-rank layers in the plan the step has no separate launch for them), [3] the slab reduce, [4] the apply launch. */
int nsgp_plan_launch_shape(const nsgp_plan_t* plan, int* shape5);

/* Two-term fp16 split of diag(c) * proj^T with c[n] the power of two that brings the largest |entry| of projector column n
 * into [2^13, 2^14) (fp16 overflows at 65504).  `out` (16-byte aligned, nsgp_split_projector_f16_bytes(D) bytes, D % 64 == 0):
 *   [n / 64][k / 8][term][n % 64][8 fp16]   D*D*4 bytes -- 1 KiB planes, the unit the projection kernel moves by LDS-DMA
 *   c[D] fp32, then 1/c[D] fp32.
 * Two launches on `stream` (column maxima, split); once per projector per task. */
size_t nsgp_split_projector_f16_bytes(int D);
int nsgp_split_projector_f16(const float* proj, int D, void* out, void* stream);

/* Stand-alone projection `out[rows x cols] (+)= scale * (a[rows x cols] @ proj[cols x cols])`
 * (SGD_NSCL.py:85-90 in isolation; accumulate=0 overwrites `out`).  Used by tests to check
 * the projected update itself at 1e-5 rel, which p += update cannot resolve in fp32. */
int nsgp_project(const float* a, const float* proj, float* out, int rows, int cols, float scale,
                 int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * Covariance accumulation  (K3)
 * Replaces BRNullSpaceRunner.compute_cov + update_cov,
 *   mmdet/engine/runner/nsrunner_roi_replay.py:876-916, 923-934:
 *   X = unfold(mean_batch(x), k, pad, stride) viewed [L x D];  C (+)= X^T X
 * x: [B,Cin,H,W]; cov: [D x D], D=Cin*kh*kw.  X is never materialised as fp32: narrow layers gather it tile by tile
 * (implicit im2col); layers with D >= 512 on the fp16-split path write X^T ONCE as the pre-tiled, pre-scaled two-term fp16
 * operand of the projection kernel (4 bytes per element, the size of the reference's unfold buffer) and contract it with
 * that kernel's LDS-DMA tile.
 * workspace: >= nsgp_cov_workspace_bytes (the zero-padded batch mean + split-K slabs + that operand where it is used).
 * accumulate=0 is the reference's first call (assign), 1 the later ones (add).
 * ------------------------------------------------------------------------ */
size_t nsgp_cov_workspace_bytes(int cin, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw);
/* Process-wide choice of the SYRK's matrix-core path: 0 = fp32 MFMA, 2 = two-term fp16 split (three fp16 MFMAs per
 * fp32-equivalent product, one power-of-two scale per layer found from the batch mean; fp32-level error), 1 (default) = the
 * split for layers large enough to repay its two extra tiny launches; 3 = always the split, restricted to its
 * first-generation (gather) kernel.  Returns the previous setting.  The workspace size covers every path. */
int nsgp_cov_set_split_mfma(int mode);
int nsgp_cov_accumulate_conv2d(const float* x, int batch, int cin, int h, int w, int kh, int kw,
                               int sh, int sw, int ph, int pw, float* cov, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream);
/* Linear branch (runner:901-902): X = mean(x, 0, keepdim) with x [B x F] -> C (+)= X^T X (rank 1). */
/* Grouped covariance pass: every eligible hooked convolution of ONE forward in a handful of launches (mean / amax / operand split /
 * one tile table over all layers, longest K first, no split-K: each tile writes its block of C and the mirror).  Replaces
 * the per-hook launches of compute_cov + update_cov, nsrunner_roi_replay.py:876-934, for a whole forward of cal_fea_in (:705-763).
 * A plan is built once per model geometry; nsgp_cov_plan_routes says which layers the grouped launches take (1) and which stay on
 * nsgp_cov_accumulate_conv2d (0: D = cin*kh*kw not a multiple of 64, or fewer than 32 output positions).  nsgp_cov_plan_run borrows
 * x[i] ([batch x cin x h x w] fp32), cov[i] ([D x D] fp32) for the layers of route 1 (entries of route-0 layers are ignored) and
 * assigns (accumulate[i] == 0) or adds; the covariances of one run must be distinct buffers.  workspace: >=
 * nsgp_cov_plan_workspace_bytes(plan), 16-byte aligned (the materialised two-term fp16 operands of all layers, slabs and correlation tables: ~2.5 GB for R-50-FPN at
 * 800 x 1344).  Deterministic: every element of C is produced by one tile in a fixed order. */
typedef struct nsgp_cov_plan nsgp_cov_plan_t;
typedef struct {
    int32_t batch, cin, h, w, kh, kw, sh, sw, ph, pw;
} nsgp_cov_geom_t;
int nsgp_cov_plan_create(nsgp_cov_plan_t** out, const nsgp_cov_geom_t* layers, int n);
int nsgp_cov_plan_destroy(nsgp_cov_plan_t* plan);
size_t nsgp_cov_plan_workspace_bytes(const nsgp_cov_plan_t* plan);
int nsgp_cov_plan_routes(const nsgp_cov_plan_t* plan, int* routes, int n);
/* n_grouped layers, tiles of the SYRK launch, and sum over the grouped layers of L*D*(D+128) = the FLOPs of the upper triangles (blocks on the diagonal counted whole) */
int nsgp_cov_plan_stats(const nsgp_cov_plan_t* plan, int* n_grouped, int* n_tiles, double* upper_flops);
/* Correlation form.  A 3x3 / stride 1 / padding 1 convolution's covariance is 81 C x C blocks that depend on their two kernel taps
 * almost only through the taps' difference: summed over one ring of positions more than the convolution has, block ((ky1,kx1),(ky2,kx2))
 * IS the shifted correlation R[ky2-ky1, kx2-kx1] = sum_q X[c1][q] X[c2][q + shift] -- 13 distinct C x C products instead of 40.5 --
 * and the ring's own covariance (2(H+W)+4 positions, four strip layers) is subtracted afterwards.  The plan takes this form per layer
 * where it saves tile-steps (the large feature maps; same fp16-split tile, same per-layer scale, sums cut into ordered ranges, C
 * bit-symmetric and bitwise reproducible as before).  nsgp_cov_plan_forms: layers in that form and the k32 steps of 256 x 128 tiles
 * the plan's SYRK launches execute; nsgp_cov_set_corr_mode (process-wide, read at plan creation; returns the previous mode):
 * 0 never, 1 by the rule (default), 2 wherever it applies. */
int nsgp_cov_plan_forms(const nsgp_cov_plan_t* plan, int* n_correlation_form, double* tile_steps);
int nsgp_cov_set_corr_mode(int mode);
int nsgp_cov_plan_run(nsgp_cov_plan_t* plan, const float* const* x, float* const* cov, const int* accumulate, void* workspace,
                      size_t workspace_bytes, void* stream);
int nsgp_cov_accumulate_linear(const float* x, int batch, int features, float* cov, int accumulate,
                               void* stream);

/* ------------------------------------------------------------------------
 * Projector build  (K5)
 * Replaces get_transforms, mmdet/engine/optimizers/SGD_NSCL.py:270-285:
 *   basis = V[:, first_col:];  P = basis basis^T;  P /= ||P||_F if normalise.
 * V: [D x D] row-major eigenvectors in columns (descending eigenvalue order), P: [D x D].
 * scratch: >= nsgp_projector_scratch_bytes(D) of device memory (Frobenius-norm partial sums).
 * ------------------------------------------------------------------------ */
size_t nsgp_projector_scratch_bytes(int D);
int nsgp_build_projector(const float* V, int D, int first_col, int normalise, float* P,
                         void* scratch, size_t scratch_bytes, void* stream);
/* The same projector from the other side (ABI 7): P = I - U U^T (/ ||P||_F if normalise) with U [D x rpad] row-major
 * fp32 = the `first_col` REMOVED directions V[:, :first_col], orthonormal, zero-padded to rpad columns (rpad a multiple of
 * 32, <= 128; D % 32 == 0).  For an orthonormal V this IS basis basis^T of SGD_NSCL.py:270-285; built this way P is exactly
 * idempotent-complement in form, bit-symmetric, and the step may apply it as p += c (u - (u U) U^T) (nsgp_tensor_t.basis). */
int nsgp_build_projector_head(const float* U, int D, int rpad, int normalise, float* P,
                              void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Prototype selection kernels  (K6, K7)
 * Replace standard_roi_replay_head.py:417-423 (row-normalise -> F F^T -> >= thr ->
 * row counts) and :413,443 (masked row means).
 *   feats: [N x D] fp32.  norm_scratch: [N] floats (||row||_2, written by the call).
 *   counts: [N] int64 (the reference's `.long().sum(-1)`).
 *   bitmask: [N x words] uint64, words = (N+63)/64; bit j of row i = (sim[i][j] >= thr).
 *   workspace: >= repre_sim_workspace_bytes(n, d) (fp32 partial tiles of the stream-K split used for
 *   classes with few rows; 0 -- and workspace may be NULL -- when N alone fills the chip).
 * ------------------------------------------------------------------------ */
size_t repre_sim_workspace_bytes(int n, int d);
int repre_sim_counts(const float* feats, int n, int d, float thr, float* norm_scratch,
                     int64_t* counts, uint64_t* bitmask, void* workspace, size_t workspace_bytes,
                     void* stream);
/* out[d] = mean over rows i with bit i set in `rowmask` (words uint64); if rowmask==NULL all rows.
 * n_selected is the popcount (caller-computed).  standard_roi_replay_head.py:413,443.
 * workspace: >= repre_masked_mean_workspace_bytes(n, d) (per-row-segment partial sums). */
size_t repre_masked_mean_workspace_bytes(int n, int d);
int repre_masked_mean(const float* feats, int n, int d, const uint64_t* rowmask, int n_selected,
                      float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Replay classifier loss  (K8, the fork's own line)
 * Replaces F.cross_entropy(cls_score.softmax(dim=-1), labels),
 * mmdet/models/roi_heads/standard_roi_replay_head.py:499 (double softmax, mean over rows), and its
 * backward: scores [n_rows x n_cols] fp32 row-major, labels int64 [n_rows], n_cols <= 256.
 * loss_out / grad_out: device fp32 scalars; grad_scores [n_rows x n_cols].
 * ------------------------------------------------------------------------ */
int repre_replay_ce_forward(const float* scores, const int64_t* labels, int n_rows, int n_cols,
                            float* loss_out, void* stream);
int repre_replay_ce_backward(const float* scores, const int64_t* labels, int n_rows, int n_cols,
                             const float* grad_out, float* grad_scores, void* stream);

/* ------------------------------------------------------------------------
 * The whole per-step replay pass  (SURVEY rows a16 + a17)
 * Replaces StandardMultiPrototypeReplayHead.replay_loss, mmdet/models/roi_heads/standard_roi_replay_head.py:468-501, over
 * Shared2FCBBoxHeadTask.forward, mmdet/models/roi_heads/bbox_heads/convfc_bbox_head_task.py:235-276, and the autograd backward of
 * both (~40 library launches around one 51 MB weight pass each way) by 6 + 5 launches on the exact fp32 MFMA:
 *     H1 = relu(bank W1^T + b1);  H2 = relu(H1 W2^T + b2);  S = H2 Wc^T + bc;  loss = mean CE(softmax(S), labels)
 * bank [n_rows x in_features] (the prototype bank: a constant, no gradient), w1 [hidden x in_features], w2 [hidden x hidden].
 * The kept class rows -- the reference keeps the columns [:task_split[task_id]] and the last one, head:495-497 -- are the rows of the
 * per-task fc_cls heads seen so far followed by the background head: n_heads (<= 16) weight / bias pointers with head_rows[h] rows
 * each, read in place (no stacked copy); n_cols = the sum of head_rows.  labels int64 [n_rows].
 * forward writes h1, h2 [n_rows x hidden], scores [n_rows x n_cols] and the scalar loss (all device fp32; the caller keeps
 * h1 / h2 / scores for backward).  backward writes (not accumulates) gw1 [hidden x in_features], gb1, gw2, gb2 [hidden] and, per
 * head, gwc_heads[h] [head_rows[h] x hidden], gbc_heads[h] [head_rows[h]], for upstream gradient *grad_out (device scalar).
 * Deterministic (fixed summation order).  n_rows <= 512, n_cols <= 256.  workspace: >= repre_replay_head_workspace_bytes(...),
 * 16-byte aligned, shared by both calls.
 * ------------------------------------------------------------------------ */
size_t repre_replay_head_workspace_bytes(int n_rows, int in_features, int hidden, int n_cols);
int repre_replay_head_forward(const float* bank, int n_rows, int in_features, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* const* wc_heads, const float* const* bc_heads, const int* head_rows,
                              int n_heads, int hidden, int n_cols, const int64_t* labels, float* h1, float* h2, float* scores,
                              float* loss_out, void* workspace, size_t workspace_bytes, void* stream);
int repre_replay_head_backward(const float* bank, int n_rows, int in_features, const float* w2, const float* const* wc_heads,
                               const float* const* bc_heads, const int* head_rows, int n_heads, int hidden, int n_cols,
                               const int64_t* labels, const float* h1, const float* h2, const float* scores, const float* grad_out,
                               float* gw1, float* gb1, float* gw2, float* gb2, float* const* gwc_heads, float* const* gbc_heads,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * EWC regulariser on the BatchNorm parameters  
 * Replaces EWCHook.__call__, mmdet/engine/runner/nsrunner_roi_replay.py:1055-1073, which launches
 * ~5 elementwise/reduction kernels per registered parameter (~106 for R-50) every step:
 *   loss = weight * sum_n sum_t sum_i F[n][t][i] * (theta[n][i] - theta_old[n][t][i])^2
 * One launch for the loss (+ a 1-block deterministic finish), one for its gradient
 *   grad[n][i] = grad_out * 2*weight * sum_t F[n][t][i] * (theta[n][i] - theta_old[n][t][i]).
 * table: DEVICE array of n_tensors records of 6 int64 each:
 *   { theta ptr, importance ptr [T x numel], old ptr [T x numel], grad ptr (backward only), numel, T }.
 * partials: >= n_tensors doubles of device scratch.  loss_out / grad_out_scalar: device fp32 scalars.
 * ------------------------------------------------------------------------ */
int nsgp_ewc_loss(const int64_t* table, int n_tensors, float weight, double* partials, float* loss_out,
                  void* stream);
int nsgp_ewc_grad(const int64_t* table, int n_tensors, float weight, const float* grad_out_scalar,
                  void* stream);

/* ------------------------------------------------------------------------
 * Teacher pseudo-label filter  
 * Replaces the per-box Python loop (one host sync per box) of FasterRCNNRoIReplay.loss,
 * mmdet/models/detectors/faster_rcnn_roi_replay.py:78-108, for ONE image: boxes [P x 4] xyxy in the
 * teacher's output order, scores [P], gt_boxes [G x 4].  Box k is dropped if its max IoU with the ground
 * truth AND with the earlier boxes already accepted into the RoI set exceeds iou_thr (0.7); otherwise
 * add_rpn[k] = score > rpn_thr, add_roi[k] = score > roi_thr.  P <= 2048.
 * ------------------------------------------------------------------------ */
int repre_pseudo_label_filter(const float* boxes, const float* scores, int n_boxes, const float* gt_boxes,
                              int n_gt, float iou_thr, float rpn_thr, float roi_thr,
                              unsigned char* add_rpn, unsigned char* add_roi, void* stream);

/* ------------------------------------------------------------------------
 * Greedy box NMS  (support for the teacher's per-step `predict`, det:72-74)
 * Replaces mmcv.ops.nms / batched_nms (mmcv is not part of the reference tree): boxes_sorted [N x 4] xyxy
 * already in DESCENDING score order (class- or level-aware NMS: add idx * (max_coord + 1) to the
 * coordinates first, as batched_nms does).  keep[0 .. *n_keep) receives the kept row indices in
 * score order, at most max_keep of them; both are device pointers (keep: int64[max_keep]).  N <= 65536.
 * ------------------------------------------------------------------------ */
size_t repre_nms_workspace_bytes(int n_boxes);
int repre_nms(const float* boxes_sorted, int n_boxes, float iou_thr, int max_keep, long long* keep,
              int* n_keep, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSGP_REPRE_H */
