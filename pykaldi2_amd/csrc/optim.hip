// Global-norm gradient clipping fused into Adam(amsgrad) / SGD(momentum) over one flat f32
// parameter buffer (gfx950, HBM-bound elementwise).
//
// Replaces nn.utils.clip_grad_norm_ + torch.optim.Adam(amsgrad=True) / torch.optim.SGD as the
// reference calls them (bin/train_ce.py:123,195-196; bin/train_chain.py:138,287-288;
// bin/train_se.py:127,253-257).  The clip coefficient min(1, max_norm/(norm+1e-6)) is read
// from device memory, so no host synchronisation sits between backward and the update.
#include <algorithm>

#include "common.h"
#include "persist_guard.h"

namespace pk2 {

constexpr int kOptThreads = 256;
constexpr int kNormBlocks = 1024;

__global__ void __launch_bounds__(kOptThreads) sumsq_partial(const float* __restrict__ g, int64_t n,
                                                             double* __restrict__ partial) {
  __shared__ double red[kOptThreads / 64];
  double acc = 0.0;
  const int64_t n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kOptThreads) {
    const float4 v = g4[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kOptThreads)
    acc += (double)g[i] * (double)g[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kOptThreads) sumsq_final(const double* __restrict__ partial, int count,
                                                           float* __restrict__ norm_out) {
  __shared__ double red[kOptThreads / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kOptThreads) acc += partial[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *norm_out = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
}

__device__ __forceinline__ float clip_coef(float max_norm, const float* norm) {
  if (max_norm <= 0.f || norm == nullptr) return 1.f;
  return fminf(1.f, max_norm / (*norm + 1e-6f));
}

template <bool AMSGRAD>
__global__ void __launch_bounds__(kOptThreads) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vmax, int64_t n, float lr,
                                                           float b1, float b2, float eps, float wd,
                                                           float bc1, float bc2_sqrt, float max_norm,
                                                           const float* norm, float grad_scale,
                                                           const unsigned* guard) {
  if (*guard != 0u) return;       // a persistent kernel gave up in this process: the gradients may be poison (persist_guard.h)
  const float coef = clip_coef(max_norm, norm) * grad_scale;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kOptThreads) {
    float pi = p[i];
    float gi = g[i] * coef;
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom;
    if (AMSGRAD) {
      const float vm = fmaxf(vmax[i], vi);
      vmax[i] = vm;
      denom = sqrtf(vm) / bc2_sqrt + eps;
    } else {
      denom = sqrtf(vi) / bc2_sqrt + eps;
    }
    p[i] = pi - step_size * (mi / denom);
  }
}

template <bool MOMENTUM>
__global__ void __launch_bounds__(kOptThreads) sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ buf, int64_t n, float lr,
                                                          float momentum, float wd, int first_step,
                                                          float max_norm, const float* norm, float grad_scale,
                                                          const unsigned* guard) {
  if (*guard != 0u) return;       // (persist_guard.h)
  const float coef = clip_coef(max_norm, norm) * grad_scale;
  for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kOptThreads) {
    float pi = p[i];
    float d = g[i] * coef;
    if (wd != 0.f) d += wd * pi;
    if (MOMENTUM) {
      const float b = first_step ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = b;
    }
    p[i] = pi - lr * d;
  }
}

static int opt_blocks(int64_t n) { return (int)std::min<int64_t>(4096, (n + kOptThreads - 1) / kOptThreads); }

}  // namespace pk2

using namespace pk2;

extern "C" size_t pk2_grad_norm_workspace_bytes(int64_t n) { (void)n; return kNormBlocks * sizeof(double); }

extern "C" int pk2_grad_norm(const float* grad, int64_t n, float* norm_out, void* workspace,
                             size_t workspace_bytes, void* stream_) {
  PK2_REQUIRE(grad && norm_out && workspace && n > 0, "grad_norm: bad args");
  PK2_REQUIRE(workspace_bytes >= kNormBlocks * sizeof(double), "grad_norm: workspace too small");
  PK2_REQUIRE((reinterpret_cast<uintptr_t>(grad) & 15) == 0, "grad_norm: grad must be 16-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int blocks = (int)std::min<int64_t>(kNormBlocks, (n / 4 + kOptThreads - 1) / kOptThreads + 1);
  hipLaunchKernelGGL(sumsq_partial, dim3(blocks), dim3(kOptThreads), 0, stream, grad, n,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(kOptThreads), 0, stream, static_cast<double*>(workspace),
                     blocks, norm_out);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             float* max_exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float max_norm,
                             const float* norm, float grad_scale, void* stream_) {
  PK2_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad args");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  PersistGuard guard;
  { int rc = persist_guard(&guard); if (rc) return rc; }
  if (max_exp_avg_sq)
    hipLaunchKernelGGL(adam_kernel<true>, dim3(opt_blocks(n)), dim3(kOptThreads), 0, stream, param, grad,
                       exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                       max_norm, norm, grad_scale, guard.dev);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3(opt_blocks(n)), dim3(kOptThreads), 0, stream, param, grad,
                       exp_avg, exp_avg_sq, nullptr, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                       max_norm, norm, grad_scale, guard.dev);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                            float momentum, float weight_decay, int32_t first_step, float max_norm,
                            const float* norm, float grad_scale, void* stream_) {
  PK2_REQUIRE(param && grad && n > 0, "sgd_step: bad args");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PersistGuard guard;
  { int rc = persist_guard(&guard); if (rc) return rc; }
  if (momentum_buf && momentum != 0.f)
    hipLaunchKernelGGL(sgd_kernel<true>, dim3(opt_blocks(n)), dim3(kOptThreads), 0, stream, param, grad,
                       momentum_buf, n, lr, momentum, weight_decay, first_step, max_norm, norm, grad_scale, guard.dev);
  else
    hipLaunchKernelGGL(sgd_kernel<false>, dim3(opt_blocks(n)), dim3(kOptThreads), 0, stream, param, grad,
                       nullptr, n, lr, momentum, weight_decay, first_step, max_norm, norm, grad_scale, guard.dev);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
