// Optimizer steps of the reference's training loop (main_cls_dgcnn.py:128-133) on the FLAT parameter / gradient buffers:
// one HBM-bound pass (float4 per lane) instead of a chain of small kernels per parameter tensor.
//   Adam : torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8), L2 weight decay folded into the gradient
//   SGD  : torch.optim.SGD with momentum (dampening 0, no Nesterov), L2 weight decay
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                        float wd, float bc1, float rsqrt_bc2) {
    // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq = b2*v + (1-b2) g^2; denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
    }
}

__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                       int64_t n, float lr, float momentum, float wd, int first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        float bi = gi;
        if (momentum != 0.f) {
            bi = first ? gi : fmaf(momentum, buf[i], gi);
            buf[i] = bi;
        }
        p[i] = pi - lr * bi;
    }
}

// The same steps with their step-dependent scalars read from DEVICE memory, so that the launch can be part of a captured graph (the
// host refreshes `hyper` with one small asynchronous copy before every replay): adam [lr, beta1, beta2, eps, weight_decay, bc1,
// 1/sqrt(bc2)], sgd [lr, momentum, weight_decay, first (0 / 1)].
__global__ __launch_bounds__(256) void adam_step_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, int64_t n, const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], rsqrt_bc2 = hyper[6];
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
    }
}
__global__ __launch_bounds__(256) void sgd_step_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                           int64_t n, const float* __restrict__ hyper) {
    const float lr = hyper[0], momentum = hyper[1], wd = hyper[2];
    const bool first = hyper[3] != 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        float bi = gi;
        if (momentum != 0.f) {
            bi = first ? gi : fmaf(momentum, buf[i], gi);
            buf[i] = bi;
        }
        p[i] = pi - lr * bi;
    }
}

}  // namespace

extern "C" int svnet_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream) {
    SVNET_REQUIRE(p && g && m && v && hyper && n >= 0, SVNET_E_ARG, "svnet_adam_step_dev_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    hipLaunchKernelGGL(adam_step_dev_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper);
    SVNET_CHECK_LAUNCH("adam_step_dev_kernel");
    return SVNET_OK;
}

extern "C" int svnet_sgd_step_dev_f32(float* p, const float* g, float* buf, int64_t n, const float* hyper, void* stream) {
    SVNET_REQUIRE(p && g && buf && hyper && n >= 0, SVNET_E_ARG, "svnet_sgd_step_dev_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    hipLaunchKernelGGL(sgd_step_dev_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, hyper);
    SVNET_CHECK_LAUNCH("sgd_step_dev_kernel");
    return SVNET_OK;
}

extern "C" int svnet_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int64_t step, void* stream) {
    SVNET_REQUIRE(p && g && m && v && n >= 0 && step >= 1, SVNET_E_ARG, "svnet_adam_step_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)));
    SVNET_CHECK_LAUNCH("adam_step_kernel");
    return SVNET_OK;
}

extern "C" int svnet_sgd_step_f32(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                                  int first_step, void* stream) {
    SVNET_REQUIRE(p && g && (buf || momentum == 0.f) && n >= 0, SVNET_E_ARG, "svnet_sgd_step_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, weight_decay,
                       first_step);
    SVNET_CHECK_LAUNCH("sgd_step_kernel");
    return SVNET_OK;
}


// ---- diagnostics: the device's constant-rate clock at the point the stream has reached (svnet_amd._lib.StepClock)
namespace {
__global__ void stamp_kernel(uint64_t* slot) { *slot = __builtin_amdgcn_s_memrealtime(); }
}  // namespace

extern "C" int svnet_stamp_u64(uint64_t* slot, void* stream) {
    SVNET_REQUIRE(slot, SVNET_E_ARG, "svnet_stamp_u64: null slot");
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slot);
    SVNET_CHECK_LAUNCH("svnet_stamp_u64");
    return SVNET_OK;
}
