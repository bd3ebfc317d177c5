// the EWC regulariser the fork adds to every training step
// (EWCHook.__call__, mmdet/engine/runner/nsrunner_roi_replay.py:1055-1073):
//     ewc_loss = sum_n 1000 * sum( cat(F_n) * (theta_n - cat(theta*_n))**2 )      over "bn" parameters
// The reference runs cat/expand/sub/pow/mul/sum/mul/add per parameter -- ~106 parameters x ~8 tiny
// launches per step for R-50, pure launch overhead.  Here: one multi-tensor launch (one workgroup per
// parameter tensor, fp64 partial per tensor), a one-block ordered finish, and one multi-tensor launch
// for the gradient.  HBM-bound and tiny (53 k BN parameters x T tasks); deterministic (no atomics).
#include "common.hpp"

namespace nsgp {

struct EwcRec {
    long long theta, importance, old, grad, numel, tasks;
};

__global__ __launch_bounds__(256) void nsgp_ewc_loss_kernel(const EwcRec* __restrict__ table, double* __restrict__ partials) {
    __shared__ double red[4];
    const EwcRec r = table[blockIdx.x];
    const float* theta = reinterpret_cast<const float*>(r.theta);
    const float* F = reinterpret_cast<const float*>(r.importance);
    const float* old = reinterpret_cast<const float*>(r.old);
    double s = 0.0;
    for (long long t = 0; t < r.tasks; ++t)
        for (long long i = threadIdx.x; i < r.numel; i += 256) {
            const float d = theta[i] - old[t * r.numel + i];      // fp32 like the reference: F * (new - old)**2
            s += (double)(F[t * r.numel + i] * (d * d));
        }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void nsgp_ewc_finish_kernel(const double* __restrict__ partials, int n, float weight,
                                                             float* __restrict__ loss_out) {
    if (threadIdx.x == 0) {
        float total = 0.0f;   // reg_loss += 1000 * tmp, parameter by parameter, in fp32 (runner:1068)
        for (int i = 0; i < n; ++i) total = total + weight * (float)partials[i];
        *loss_out = total;
    }
}

__global__ __launch_bounds__(256) void nsgp_ewc_grad_kernel(const EwcRec* __restrict__ table, float weight,
                                                            const float* __restrict__ grad_out) {
    const EwcRec r = table[blockIdx.x];
    const float* theta = reinterpret_cast<const float*>(r.theta);
    const float* F = reinterpret_cast<const float*>(r.importance);
    const float* old = reinterpret_cast<const float*>(r.old);
    float* g = reinterpret_cast<float*>(r.grad);
    const float go = 2.0f * weight * (*grad_out);
    for (long long i = threadIdx.x; i < r.numel; i += 256) {
        float s = 0.0f;
        for (long long t = 0; t < r.tasks; ++t) s += F[t * r.numel + i] * (theta[i] - old[t * r.numel + i]);
        g[i] = go * s;
    }
}

}  // namespace nsgp

using namespace nsgp;

extern "C" int nsgp_ewc_loss(const int64_t* table, int n, float weight, double* partials, float* loss_out, void* stream_) {
    if (!table || !partials || !loss_out || n <= 0) return fail(NSGP_ERR_INVALID, "nsgp_ewc_loss: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(nsgp_ewc_loss_kernel, dim3(n), dim3(256), 0, stream, reinterpret_cast<const EwcRec*>(table), partials);
    NSGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nsgp_ewc_finish_kernel, dim3(1), dim3(64), 0, stream, partials, n, weight, loss_out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}

extern "C" int nsgp_ewc_grad(const int64_t* table, int n, float weight, const float* grad_out, void* stream_) {
    if (!table || !grad_out || n <= 0) return fail(NSGP_ERR_INVALID, "nsgp_ewc_grad: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(nsgp_ewc_grad_kernel, dim3(n), dim3(256), 0, stream, reinterpret_cast<const EwcRec*>(table), weight, grad_out);
    NSGP_LAUNCH_CHECK();
    return NSGP_OK;
}
