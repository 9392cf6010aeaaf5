// Fused log-softmax + NLL + gradient, one pass over the logits (gfx950).
//
// Replaces nn.CrossEntropyLoss(ignore_index=-100) as called by the reference
// (bin/train_ce.py:134,189 mean reduction; bin/train_se.py:214,235 sum reduction).
// HBM-bound: reads rows*P floats once, writes rows*P floats once.  One 256-thread workgroup
// per row keeps the whole row (P <= 8192) in registers between the max / sum-exp reductions
// and the gradient write.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace pk2 {

constexpr int kCeThreads = 256;
constexpr int kCeVpt = 32;  // values per thread held in registers -> P <= 8192

__device__ __forceinline__ float ce_block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float ce_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool IN_REGS>
__global__ void __launch_bounds__(kCeThreads) softmax_ce_kernel(
    const float* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ targets,
    int64_t ignore_index, int64_t rows, int P, float* loss_sum, int32_t* count, float* __restrict__ grad,
    int64_t grad_row_stride, float* __restrict__ logprob, const int32_t* __restrict__ count_in) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  // count_in (pk2_softmax_ce_fwd_bwd_mean): the number of valid targets is known before this launch (count_valid_kernel), so the
  // gradient of the MEAN loss is written as it is -- no scaling pass over [rows, P] behind this kernel; the count is not added again
  const float gs = count_in ? 1.f / (float)max(1, *count_in) : 1.f;
  // A workgroup takes rows blockIdx.x, + gridDim.x, ... and adds its loss and count to the totals ONCE at the end: an atomic
  // pair per row (20 480 rows x 2 on one cache line at the CE configuration) queued in L2 for longer than the rows took to
  // stream -- 518 us for 0.94 GB.
  float loss_local = 0.f;
  int count_local = 0;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
  __syncthreads();                       // (red[] of the previous row has been read by everybody)
  const float* x = logits + row * row_stride;
  const int64_t tgt = targets[row];
  // A label outside [0, P) that is not ignore_index (an alignment with pdf-ids beyond the config's label_size) must not
  // index the row: nn.CrossEntropyLoss asserts on it; here the row contributes nothing and the loss becomes NaN -- loud,
  // without a device-to-host round trip on the hot path.
  const bool out_of_range = tgt != ignore_index && (tgt < 0 || tgt >= P);
  if (out_of_range && tid == 0) loss_local += __int_as_float(0x7fc00000);
  const bool ignored = tgt == ignore_index || out_of_range;
  float* g = grad ? grad + row * grad_row_stride : nullptr;
  if (ignored && !logprob) {
    if (g) for (int p = tid; p < P; p += kCeThreads) g[p] = 0.f;
    continue;
  }
  float v[kCeVpt];
  float m = -INFINITY;
  if (IN_REGS) {
#pragma unroll
    for (int k = 0; k < kCeVpt; ++k) {
      const int p = tid + k * kCeThreads;
      v[k] = p < P ? x[p] : -INFINITY;
      m = fmaxf(m, v[k]);
    }
  } else {
    for (int p = tid; p < P; p += kCeThreads) m = fmaxf(m, x[p]);
  }
  m = ce_block_max(m, red);
  float s = 0.f;
  if (IN_REGS) {
#pragma unroll
    for (int k = 0; k < kCeVpt; ++k) { v[k] = expf(v[k] - m); s += v[k]; }  // exp(-inf) = 0 for the tail
  } else {
    for (int p = tid; p < P; p += kCeThreads) s += expf(x[p] - m);
  }
  s = ce_block_sum(s, red);
  const float lse = m + logf(s);
  const float inv = 1.0f / s;
  if (!ignored && tid == 0) {
    loss_local += lse - x[tgt];
    ++count_local;
  }
  if (IN_REGS) {
#pragma unroll
    for (int k = 0; k < kCeVpt; ++k) {
      const int p = tid + k * kCeThreads;
      if (p < P) {
        const float sm = v[k] * inv;
        if (g) g[p] = ignored ? 0.f : (sm - (p == tgt ? 1.f : 0.f)) * gs;
        if (logprob) logprob[row * (int64_t)P + p] = logf(fmaxf(sm, 1e-45f)) ;
      }
    }
  } else {
    for (int p = tid; p < P; p += kCeThreads) {
      const float lp = x[p] - lse;
      if (g) g[p] = ignored ? 0.f : (expf(lp) - (p == tgt ? 1.f : 0.f)) * gs;
      if (logprob) logprob[row * (int64_t)P + p] = lp;
    }
  }
  }
  if (tid == 0) {
    if (loss_local != 0.f) atomicAdd(loss_sum, loss_local);       // (NaN != 0: the out-of-range marker goes through)
    if (count_local && !count_in) atomicAdd(count, count_local);
  }
}

// Valid targets (neither ignore_index nor outside [0, P)) of the whole call, one workgroup: 20 480 labels at the CE configuration.
__global__ void __launch_bounds__(1024) count_valid_kernel(const int64_t* __restrict__ targets, int64_t rows, int64_t ignore_index, int P,
                                                           int32_t* count) {
  __shared__ int red[16];
  int c = 0;
  for (int64_t r = threadIdx.x; r < rows; r += 1024) {
    const int64_t t = targets[r];
    c += (t != ignore_index && t >= 0 && t < P) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    *count = t;
  }
}

// data[i] *= (*num) / (*den)  (den may be null: 1), nothing touched when the factor is exactly 1: the loss's incoming gradient
// applied IN PLACE to a gradient that is otherwise final -- loss.backward() hands in 1.0 and the pass over [rows, P] does not
// happen (every workgroup reads the two scalars and leaves).  A repeated backward over a retained graph passes the factor
// the buffer already carries as den.
__global__ void scale_inplace_ratio_kernel(float* __restrict__ data, int64_t n, const float* num, const float* den) {
  const float f = (*num) / (den ? *den : 1.f);
  if (f == 1.f) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(data) & 15u) == 0 ? n / 4 : 0;
  for (int64_t i = i0; i < n4; i += stride) {
    float4 v = reinterpret_cast<float4*>(data)[i];
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    reinterpret_cast<float4*>(data)[i] = v;
  }
  for (int64_t i = 4 * n4 + i0; i < n; i += stride) data[i] *= f;
}

__global__ void scale_by_count_kernel(float* data, int64_t n, float numerator, const int32_t* count) {
  const float f = numerator / (float)max(1, *count);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    data[i] *= f;
}

// out[i] = in[i] * (*mul) / max(1, *count): the two scalings of the CE gradient (1 / valid targets of reduction='mean', the
// loss's incoming gradient) in ONE pass over the [rows, P] tensor, both factors read from device memory.
__global__ void scale_by_scalars_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, const float* mul,
                                        const int32_t* count) {
  const float f = (mul ? *mul : 1.f) / (count ? (float)max(1, *count) : 1.f);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 ? n / 4 : 0;
  for (int64_t i = i0; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(in)[i];
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    reinterpret_cast<float4*>(out)[i] = v;
  }
  for (int64_t i = 4 * n4 + i0; i < n; i += stride) out[i] = in[i] * f;
}

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_scale_by_scalars(const float* in, float* out, int64_t n, const float* mul, const int32_t* count_den,
                                    void* stream_) {
  PK2_REQUIRE(in && out && n >= 0, "scale_by_scalars: bad args");
  if (n == 0) return PK2_OK;
  const int blocks = (int)std::min<int64_t>(4096, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(scale_by_scalars_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), in, out, n, mul,
                     count_den);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

static int softmax_ce_launch(const float* logits, int64_t row_stride, const int64_t* targets, int64_t ignore_index, int64_t rows,
                             int32_t P, float* loss_sum, int32_t* count, float* grad, int64_t grad_row_stride, float* logprob_out,
                             bool mean_grad, hipStream_t stream) {
  PK2_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), stream));
  const int32_t* count_in = nullptr;
  if (mean_grad && rows > 0) {
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(1024), 0, stream, targets, rows, ignore_index, P, count);
    count_in = count;
  } else {
    PK2_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), stream));
  }
  if (rows == 0) return PK2_OK;
  // (256 CUs x 8 resident workgroups; a workgroup walks rows blockIdx.x, + grid, ...)
  static const int ce_grid = [] { const char* e = getenv("PK2_CE_GRID"); const int v = e ? atoi(e) : 2048; return v < 1 ? 1 : v; }();
  const unsigned grid = (unsigned)std::min<int64_t>(rows, ce_grid);
  if (P <= kCeThreads * kCeVpt && !logprob_out) {
    hipLaunchKernelGGL(softmax_ce_kernel<true>, dim3(grid), dim3(kCeThreads), 0, stream, logits,
                       row_stride, targets, ignore_index, rows, P, loss_sum, count, grad, grad_row_stride,
                       logprob_out, count_in);
  } else {
    hipLaunchKernelGGL(softmax_ce_kernel<false>, dim3(grid), dim3(kCeThreads), 0, stream, logits,
                       row_stride, targets, ignore_index, rows, P, loss_sum, count, grad, grad_row_stride,
                       logprob_out, count_in);
  }
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_softmax_ce_fwd_bwd(const float* logits, int64_t row_stride, const int64_t* targets,
                                      int64_t ignore_index, int64_t rows, int32_t P, float* loss_sum,
                                      int32_t* count, float* grad, int64_t grad_row_stride,
                                      float* logprob_out, void* stream_) {
  PK2_REQUIRE(logits && targets && loss_sum && count && rows >= 0 && P > 0, "softmax_ce: bad args");
  return softmax_ce_launch(logits, row_stride, targets, ignore_index, rows, P, loss_sum, count, grad, grad_row_stride, logprob_out,
                           false, static_cast<hipStream_t>(stream_));
}

// As above with grad = d(MEAN loss) / d logits: the valid targets are counted by a one-workgroup launch in front, the
// gradient leaves the kernel divided by max(1, count) (loss_sum stays the SUM: the caller divides one scalar).
extern "C" int pk2_softmax_ce_fwd_bwd_mean(const float* logits, int64_t row_stride, const int64_t* targets,
                                           int64_t ignore_index, int64_t rows, int32_t P, float* loss_sum,
                                           int32_t* count, float* grad, int64_t grad_row_stride, void* stream_) {
  PK2_REQUIRE(logits && targets && loss_sum && count && grad && rows >= 0 && P > 0, "softmax_ce (mean): bad args");
  return softmax_ce_launch(logits, row_stride, targets, ignore_index, rows, P, loss_sum, count, grad, grad_row_stride, nullptr,
                           true, static_cast<hipStream_t>(stream_));
}

extern "C" int pk2_scale_inplace_ratio(float* data, int64_t n, const float* num_dev, const float* den_dev, void* stream_) {
  PK2_REQUIRE(data && num_dev && n >= 0, "scale_inplace_ratio: bad args");
  if (n == 0) return PK2_OK;
  const int blocks = (int)std::min<int64_t>(4096, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(scale_inplace_ratio_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), data, n, num_dev,
                     den_dev);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_scale_by_count(float* data, int64_t n, float numerator, const int32_t* count_den,
                                  void* stream_) {
  PK2_REQUIRE(data && count_den && n >= 0, "scale_by_count: bad args");
  if (n == 0) return PK2_OK;
  int blocks = (int)std::min<int64_t>(2048, (n + 255) / 256);
  hipLaunchKernelGGL(scale_by_count_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     data, n, numerator, count_den);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
