// the teacher pseudo-label filter inside FasterRCNNRoIReplay.loss
// (mmdet/models/detectors/faster_rcnn_roi_replay.py:78-108).  Per image the reference walks the
// teacher's predictions in order with a Python loop and one `.item()` host sync per box:
//     max_iou = box_iou(box_k, gt_data_sample.gt_instances.bboxes).max()      (0 if that set is empty)
//     if max_iou > 0.7: continue
//     if score_k > rpn_thresh: rpn set += box_k
//     if score_k > roi_thresh: gt_data_sample.gt_instances += box_k           <-- the set box k+1 is tested against
// i.e. a SEQUENTIAL greedy filter: later boxes are also tested against earlier boxes that were
// accepted into the RoI set.  One wave per image does the whole walk on the GPU: IoU against the
// original ground truth for every box in parallel first, then the ordered scan with the accepted
// flags in LDS and a wave-wide max per step.  `box_iou` is torchvision.ops.box_iou's published
// formula (torchvision is absent from the reference tree and from this image): inter / (a1 + a2 - inter).
#include "common.hpp"

namespace nsgp {

constexpr int PL_MAX_BOXES = 2048;

__device__ __forceinline__ float box_iou1(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}

__global__ __launch_bounds__(64) void repre_pseudo_label_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                                int P, const float4* __restrict__ gt, int G, float iou_thr,
                                                                float rpn_thr, float roi_thr, unsigned char* __restrict__ add_rpn,
                                                                unsigned char* __restrict__ add_roi) {
    __shared__ float gtmax[PL_MAX_BOXES];
    __shared__ unsigned char in_roi[PL_MAX_BOXES];
    __shared__ float4 sbox[PL_MAX_BOXES];          // boxes and scores staged once: the ordered walk below then touches LDS only
    __shared__ float sscore[PL_MAX_BOXES];         // (a global load on every dependent step cost ~0.5 us per box)
    const int lane = threadIdx.x;
    for (int k = lane; k < P; k += 64) {
        float m = 0.0f;
        const float4 b = boxes[k];
        for (int g = 0; g < G; ++g) m = fmaxf(m, box_iou1(b, gt[g]));
        gtmax[k] = m;
        in_roi[k] = 0;
        sbox[k] = b;
        sscore[k] = scores[k];
    }
    __syncthreads();
    for (int k = 0; k < P; ++k) {
        const float4 b = sbox[k];
        float m = gtmax[k];
        for (int j = lane; j < k; j += 64)
            if (in_roi[j]) m = fmaxf(m, box_iou1(b, sbox[j]));
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const bool keep = !(m > iou_thr);          // `if max_iou > 0.7: continue`
        const float s = sscore[k];
        const bool rp = keep && (s > rpn_thr), ro = keep && (s > roi_thr);
        if (lane == 0) {
            add_rpn[k] = rp;
            add_roi[k] = ro;
            in_roi[k] = ro;
        }
        __syncthreads();
    }
}

// P <= 256 (the teacher keeps at most 100 boxes per image): the pairwise "IoU > thr" relation of every box with the EARLIER boxes
// is computed first, in parallel, as four 64-bit words per box; the ordered walk is then bit arithmetic on words read from LDS --
// no IoU, no wave reduction and no barrier on the dependent chain (the walk above costs ~0.5 us per box).  Same decisions:
// `max over a set > thr` == `any member > thr` (fmaxf skips NaNs, and NaN > thr is false).
constexpr int PL_FAST_BOXES = 256;

__global__ __launch_bounds__(256) void repre_pseudo_label_small_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                                       int P, const float4* __restrict__ gt, int G, float iou_thr,
                                                                       float rpn_thr, float roi_thr, unsigned char* __restrict__ add_rpn,
                                                                       unsigned char* __restrict__ add_roi) {
    __shared__ float4 sbox[PL_FAST_BOXES];
    __shared__ float sscore[PL_FAST_BOXES];
    __shared__ unsigned long long rel[PL_FAST_BOXES][4];     // rel[k][w] bit b: IoU(box k, box 64 w + b) > thr, earlier boxes only
    __shared__ unsigned char over_gt[PL_FAST_BOXES];
    const int t = threadIdx.x;
    if (t < P) { sbox[t] = boxes[t]; sscore[t] = scores[t]; }
    __syncthreads();
    if (t < P) {
        const float4 b = sbox[t];
        bool og = false;
        for (int g = 0; g < G; ++g) og |= box_iou1(b, gt[g]) > iou_thr;
        over_gt[t] = og;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned long long w = 0;
            const int jn = min(t - 64 * i, 64);                 // earlier boxes only
            for (int j = 0; j < jn; ++j)
                if (box_iou1(b, sbox[64 * i + j]) > iou_thr) w |= 1ull << j;
            rel[t][i] = w;
        }
    }
    __syncthreads();
    if (t >= 64) return;
    // one wave, every lane the same walk (uniform values); lane 0 stores
    unsigned long long acc[4] = {0, 0, 0, 0};               // boxes accepted into the RoI set so far
#pragma unroll
    for (int w = 0; w < 4; ++w) {                           // word by word so that acc[] is indexed statically
        for (int b = 0; b < 64; ++b) {
            const int k = w * 64 + b;
            if (k >= P) break;
            const bool blocked = over_gt[k] | (((rel[k][0] & acc[0]) | (rel[k][1] & acc[1]) | (rel[k][2] & acc[2]) | (rel[k][3] & acc[3])) != 0);
            const float s = sscore[k];
            const bool rp = !blocked && (s > rpn_thr), ro = !blocked && (s > roi_thr);
            if (ro) acc[w] |= 1ull << b;
            if (t == 0) { add_rpn[k] = rp; add_roi[k] = ro; }
        }
    }
}

}  // namespace nsgp

using namespace nsgp;

extern "C" int repre_pseudo_label_filter(const float* boxes, const float* scores, int n_boxes, const float* gt_boxes, int n_gt,
                                         float iou_thr, float rpn_thr, float roi_thr, unsigned char* add_rpn,
                                         unsigned char* add_roi, void* stream_) {
    if (n_boxes == 0) return NSGP_OK;
    if (!boxes || !scores || !add_rpn || !add_roi || n_boxes < 0 || n_gt < 0 || (n_gt > 0 && !gt_boxes))
        return fail(NSGP_ERR_INVALID, "repre_pseudo_label_filter: bad argument");
    if (n_boxes > PL_MAX_BOXES) return fail(NSGP_ERR_LIMIT, "repre_pseudo_label_filter: %d boxes > %d", n_boxes, PL_MAX_BOXES);
    if (!aligned16(boxes) || (n_gt > 0 && !aligned16(gt_boxes))) return fail(NSGP_ERR_INVALID, "repre_pseudo_label_filter: boxes must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_boxes <= PL_FAST_BOXES)
        hipLaunchKernelGGL(repre_pseudo_label_small_kernel, dim3(1), dim3(256), 0, stream, reinterpret_cast<const float4*>(boxes), scores, n_boxes,
                           reinterpret_cast<const float4*>(gt_boxes), n_gt, iou_thr, rpn_thr, roi_thr, add_rpn, add_roi);
    else
        hipLaunchKernelGGL(repre_pseudo_label_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const float4*>(boxes), scores, n_boxes,
                           reinterpret_cast<const float4*>(gt_boxes), n_gt, iou_thr, rpn_thr, roi_thr, add_rpn, add_roi);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
