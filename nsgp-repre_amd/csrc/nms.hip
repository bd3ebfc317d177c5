// Greedy box NMS for the teacher's per-step `predict` inside FasterRCNNRoIReplay.loss
// (mmdet/models/detectors/faster_rcnn_roi_replay.py:72-74 -> RPN `batched_nms` + per-class NMS of the
// RoI head).  The reference gets this from mmcv.ops.nms (mmcv >=2.0.0rc4,<2.2.0 -- a CUDA extension that
// is absent from the reference tree and from this image); this is the published algorithm:
//     walk the boxes in descending score order; keep a box unless its IoU with an already kept box
//     exceeds the threshold;  IoU = inter / (a + b - inter), widths/heights without the legacy +1.
// Two launches: (1) the strict upper triangle of the "IoU > thr" relation as 64-bit words, fully
// parallel; (2) ONE workgroup walks the boxes a 64-box block at a time -- the removed set lives in LDS, the kept
// boxes of a block OR their rows in together, and the walk stops at `max_keep` (kept boxes are in score order, so
// the first `max_keep` kept are exactly mmcv's `keep[:max_num]`).
#include "common.hpp"

namespace nsgp {

constexpr int NMS_MAX_WORDS = 1024;   // 65536 boxes

__device__ __forceinline__ bool nms_over(const float4 a, const float4 b, float thr) {
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    if (!(inter > 0.0f)) return false;   // disjoint boxes (every pair from different levels / classes): 0 / u > thr is false for thr >= 0, no division
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

// the readlane / readfirstlane builtins return a SIGNED int: widen through unsigned, or a set bit 31 smears over the high word
__device__ __forceinline__ unsigned long long nms_u64(int hi, int lo) {
    return (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo);
}

// Scan, one workgroup of four waves, one 64-box block per round (two barriers per BLOCK, not per kept box):
//   * wave 0 resolves the block on its own: the block's 64 x 64 diagonal words sit one per lane (fetched a block ahead), the walk
//     over the still-alive bits is scalar work (find-first-set, one v_readlane of the kept box's diagonal word) -- no memory on
//     the dependent chain;
//   * it publishes the kept word; all four waves then OR the kept boxes' rows into the removed set in LDS (wave k takes every
//     fourth kept box, a lane the words lane, lane + 64, ... to the right of the block; four row loads in flight per lane).
// Same greedy walk, same result as keeping boxes one at a time; the walk stops once `max_keep` boxes are kept.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int words, int max_keep,
                                                       long long* __restrict__ keep, int* __restrict__ n_keep) {
    __shared__ unsigned long long removed[NMS_MAX_WORDS];
    __shared__ unsigned long long s_kept;
    __shared__ unsigned char s_idx[64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int w = t; w < words; w += 256) removed[w] = 0;
    unsigned long long diag = 0;                  // wave 0: row (w * 64 + lane), word w
    if (wave == 0 && lane < n) diag = mask[static_cast<size_t>(lane) * words];
    __syncthreads();
    int nk = 0;
    for (int w = 0; w < words && nk < max_keep; ++w) {
        if (wave == 0) {
            unsigned long long next = 0;          // the next block's diagonal: in flight during this block's walk
            const int ni = (w + 1) * 64 + lane;
            if (w + 1 < words && ni < n) next = mask[static_cast<size_t>(ni) * words + (w + 1)];
            const int valid = min(64, n - w * 64);
            const unsigned long long vmask = valid == 64 ? ~0ull : ((1ull << valid) - 1);
            const unsigned long long r = removed[w];
            unsigned long long live = ~r & vmask;
            live = nms_u64(__builtin_amdgcn_readfirstlane(static_cast<unsigned>(live >> 32)), __builtin_amdgcn_readfirstlane(static_cast<unsigned>(live)));
            unsigned long long kept = 0;
            int budget = max_keep - nk;
            const unsigned dlo = static_cast<unsigned>(diag), dhi = static_cast<unsigned>(diag >> 32);
            while (live && budget > 0) {
                const int b = __builtin_ctzll(live);
                kept |= 1ull << b;
                --budget;
                const unsigned long long drow = nms_u64(__builtin_amdgcn_readlane(dhi, b), __builtin_amdgcn_readlane(dlo, b));
                live &= ~drow;                    // strict upper triangle: only bits above b are set in drow
                live &= ~((2ull << b) - 1);
            }
            if ((kept >> lane) & 1) {
                const int pos = __popcll(kept & ((1ull << lane) - 1));
                keep[nk + pos] = static_cast<long long>(w) * 64 + lane;
                s_idx[pos] = static_cast<unsigned char>(lane);
            }
            if (lane == 0) s_kept = kept;
            diag = next;
        }
        __syncthreads();
        const unsigned long long kept = s_kept;
        const int kc = __popcll(kept);
        nk += kc;
        if (w + 1 < words && nk < max_keep) {
            const size_t base = static_cast<size_t>(w) * 64;
            for (int j = wave; j < kc; j += 16) {             // this wave's kept boxes j, j+4, j+8, j+12: four rows in flight
                const unsigned long long* r0 = mask + (base + s_idx[j]) * words;
                const unsigned long long* r1 = mask + (base + s_idx[min(j + 4, kc - 1)]) * words;
                const unsigned long long* r2 = mask + (base + s_idx[min(j + 8, kc - 1)]) * words;
                const unsigned long long* r3 = mask + (base + s_idx[min(j + 12, kc - 1)]) * words;
                for (int ww = w + 1 + lane; ww < words; ww += 64) {
                    const unsigned long long v = r0[ww] | r1[ww] | r2[ww] | r3[ww];   // a clamped duplicate row ORs in nothing new
                    if (v) atomicOr(&removed[ww], v);
                }
            }
        }
        __syncthreads();
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
