// Greedy box NMS for the teacher's per-step `predict` inside FasterRCNNRoIReplay.loss
// (mmdet/models/detectors/faster_rcnn_roi_replay.py:72-74 -> RPN `batched_nms` + per-class NMS of the
// RoI head).  The reference gets this from mmcv.ops.nms (mmcv >=2.0.0rc4,<2.2.0 -- a CUDA extension that
// is absent from the reference tree and from this image); this is the published algorithm:
//     walk the boxes in descending score order; keep a box unless its IoU with an already kept box
//     exceeds the threshold;  IoU = inter / (a + b - inter), widths/heights without the legacy +1.
// Two launches: (1) the strict upper triangle of the "IoU > thr" relation as 64-bit words, fully
// parallel; (2) ONE workgroup walks the kept boxes only -- the removed set lives in LDS, every kept box
// ORs its row in, and the walk stops at `max_keep` (kept boxes are in score order, so the first
// `max_keep` kept are exactly mmcv's `keep[:max_num]`).
#include "common.hpp"

namespace nsgp {

constexpr int NMS_MAX_WORDS = 1024;   // 65536 boxes

__device__ __forceinline__ bool nms_over(const float4 a, const float4 b, float thr) {
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    return inter / (area_a + area_b - inter) > thr;
}

// grid (col blocks, row blocks); only col block >= row block does work, the rest of `mask` is never read
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, int n, float thr, int words,
                                                      unsigned long long* __restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;
    __shared__ float4 col[64];
    const int t = threadIdx.x;
    const int cn = min(64, n - cb * 64);
    if (t < cn) col[t] = boxes[cb * 64 + t];
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const float4 me = boxes[i];
    unsigned long long bits = 0;
    const int j0 = (cb == rb) ? t + 1 : 0;        // strict upper triangle
    for (int j = j0; j < cn; ++j)
        if (nms_over(me, col[j], thr)) bits |= 1ull << j;
    mask[static_cast<size_t>(i) * words + cb] = bits;
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int words, int max_keep,
                                                       long long* __restrict__ keep, int* __restrict__ n_keep) {
    __shared__ unsigned long long removed[NMS_MAX_WORDS];
    const int t = threadIdx.x;
    for (int w = t; w < words; w += 256) removed[w] = 0;
    __syncthreads();
    int nk = 0;
    for (int w = 0; w < words && nk < max_keep; ++w) {
        const int valid = min(64, n - w * 64);
        const unsigned long long vmask = valid == 64 ? ~0ull : ((1ull << valid) - 1);
        // every thread follows the same walk: `removed[w]` is re-read after each barrier
        unsigned long long live = ~removed[w] & vmask;
        while (live && nk < max_keep) {
            const int b = __ffsll(static_cast<long long>(live)) - 1;
            const int i = w * 64 + b;
            if (t == 0) keep[nk] = i;
            ++nk;
            __syncthreads();                       // everyone has read removed[w] for this round
            const unsigned long long* row = mask + static_cast<size_t>(i) * words;
            for (int ww = w + t; ww < words; ww += 256) removed[ww] |= row[ww];
            __syncthreads();
            live = ~removed[w] & vmask & ~((2ull << b) - 1);   // bits above b that are still alive
        }
    }
    if (t == 0) *n_keep = nk;
}

}  // namespace nsgp

using namespace nsgp;

extern "C" size_t repre_nms_workspace_bytes(int n_boxes) {
    if (n_boxes <= 0) return 0;
    const size_t words = (static_cast<size_t>(n_boxes) + 63) / 64;
    return static_cast<size_t>(n_boxes) * words * sizeof(unsigned long long);
}

extern "C" int repre_nms(const float* boxes_sorted, int n_boxes, float iou_thr, int max_keep, long long* keep, int* n_keep,
                         void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!n_keep) return fail(NSGP_ERR_INVALID, "repre_nms: n_keep is null");
    if (n_boxes == 0 || max_keep == 0) {
        NSGP_HIP(hipMemsetAsync(n_keep, 0, sizeof(int), stream));
        return NSGP_OK;
    }
    if (!boxes_sorted || !keep || n_boxes < 0 || max_keep < 0) return fail(NSGP_ERR_INVALID, "repre_nms: bad argument");
    if (!aligned16(boxes_sorted)) return fail(NSGP_ERR_INVALID, "repre_nms: boxes must be 16-byte aligned");
    const int words = (n_boxes + 63) / 64;
    if (words > NMS_MAX_WORDS) return fail(NSGP_ERR_LIMIT, "repre_nms: %d boxes > %d", n_boxes, NMS_MAX_WORDS * 64);
    if (!workspace || workspace_bytes < repre_nms_workspace_bytes(n_boxes))
        return fail(NSGP_ERR_WORKSPACE, "repre_nms: workspace too small (%zu < %zu)", workspace_bytes, repre_nms_workspace_bytes(n_boxes));
    unsigned long long* mask = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, stream, reinterpret_cast<const float4*>(boxes_sorted), n_boxes,
                       iou_thr, words, mask);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), 0, stream, mask, n_boxes, words, max_keep, keep, n_keep);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
