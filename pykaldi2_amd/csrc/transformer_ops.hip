// Row-wise kernels of the Transformer acoustic model (gfx950): LayerNorm (+ residual), masked softmax of the
// attention scores, ReLU.  The matrix products go through pk2_gemm_f32 / pk2_gemm_f32_batched.
//
// Replaces the ATen kernels under nn.TransformerEncoderLayer / nn.LayerNorm / F.relu as the reference's
// TransformerAM uses them (reference models/transformer.py:52-94: post-norm encoder layer, ReLU FFN,
// Conv1d(k=3) + ReLU after every layer, final LayerNorm).  All HBM-bound row passes.
#include <algorithm>

#include "common.h"

namespace pk2 {

constexpr int kRowThreads = 256;

__device__ __forceinline__ float row_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float row_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// s = x + res (res may be null); y = (s - mean)/sqrt(var + eps) * gamma + beta; saves s, mean, rstd.
__global__ void __launch_bounds__(kRowThreads) layernorm_fwd_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ res,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int C, float eps,
                                                                    float* __restrict__ s_out, float* __restrict__ y,
                                                                    float* __restrict__ mean_out,
                                                                    float* __restrict__ rstd_out) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * C;
  const float* rr = res ? res + row * C : nullptr;
  float loc = 0.f;
  for (int c = threadIdx.x; c < C; c += kRowThreads) {
    const float v = xr[c] + (rr ? rr[c] : 0.f);
    if (s_out) s_out[row * C + c] = v;
    loc += v;
  }
  const float mean = row_sum(loc, red) / C;
  float var = 0.f;
  for (int c = threadIdx.x; c < C; c += kRowThreads) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
    var += d * d;
  }
  const float rstd = rsqrtf(row_sum(var, red) / C + eps);
  for (int c = threadIdx.x; c < C; c += kRowThreads) {
    const float v = xr[c] + (rr ? rr[c] : 0.f);
    y[row * C + c] = (v - mean) * rstd * gamma[c] + beta[c];
  }
  if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// ds = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-row partial dgamma/dbeta are
// accumulated by a second kernel (column sums over rows).
__global__ void __launch_bounds__(kRowThreads) layernorm_bwd_kernel(const float* __restrict__ dy,
                                                                    const float* __restrict__ s,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma, int C,
                                                                    float* __restrict__ ds) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float m = mean[row], r = rstd[row];
  float a = 0.f, b = 0.f;
  for (int c = threadIdx.x; c < C; c += kRowThreads) {
    const float g = dy[row * C + c] * gamma[c];
    const float xh = (s[row * C + c] - m) * r;
    a += g; b += g * xh;
  }
  const float ma = row_sum(a, red) / C;
  const float mb = row_sum(b, red) / C;
  for (int c = threadIdx.x; c < C; c += kRowThreads) {
    const float g = dy[row * C + c] * gamma[c];
    const float xh = (s[row * C + c] - m) * r;
    ds[row * C + c] = r * (g - ma - xh * mb);
  }
}

// dgamma[c] += sum_rows dy * xhat ; dbeta[c] += sum_rows dy   (rows split over blockIdx.y, atomics)
__global__ void __launch_bounds__(256) layernorm_param_grad_kernel(const float* __restrict__ dy,
                                                                   const float* __restrict__ s,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, int64_t rows, int C,
                                                                   int rows_per_block, float* dgamma, float* dbeta) {
  __shared__ float rg[4][64], rb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float ag = 0.f, ab = 0.f;
  if (c < C)
    for (int64_t r = r0 + part; r < r1; r += 4) {
      const float d = dy[r * C + c];
      ag += d * (s[r * C + c] - mean[r]) * rstd[r];
      ab += d;
    }
  rg[part][threadIdx.x & 63] = ag; rb[part][threadIdx.x & 63] = ab;
  __syncthreads();
  if (part == 0 && c < C) {
    atomicAdd(dgamma + c, (rg[0][threadIdx.x] + rg[1][threadIdx.x]) + (rg[2][threadIdx.x] + rg[3][threadIdx.x]));
    atomicAdd(dbeta + c, (rb[0][threadIdx.x] + rb[1][threadIdx.x]) + (rb[2][threadIdx.x] + rb[3][threadIdx.x]));
  }
}

// scores[z][i][:] (z = b*H + h) <- softmax_j(scores + src_mask[i][j]  with key_pad[b][j] -> -inf), in place.
__global__ void __launch_bounds__(kRowThreads) softmax_mask_kernel(float* __restrict__ scores,
                                                                   const float* __restrict__ src_mask,
                                                                   const uint8_t* __restrict__ key_pad, int H, int T) {
  __shared__ float red[4];
  const int i = blockIdx.x, z = blockIdx.y, b = z / H;
  float* row = scores + ((int64_t)z * T + i) * T;
  const float* mrow = src_mask ? src_mask + (int64_t)i * T : nullptr;
  const uint8_t* kp = key_pad ? key_pad + (int64_t)b * T : nullptr;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < T; j += kRowThreads) {
    float v = row[j] + (mrow ? mrow[j] : 0.f);
    if (kp && kp[j]) v = -INFINITY;
    row[j] = v;
    m = fmaxf(m, v);
  }
  m = row_max(m, red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < T; j += kRowThreads) {
    const float e = (m == -INFINITY) ? 0.f : expf(row[j] - m);
    row[j] = e;
    sum += e;
  }
  sum = row_sum(sum, red);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int j = threadIdx.x; j < T; j += kRowThreads) row[j] *= inv;
}

// dS = P * (dP - sum_j dP*P), in place on dP.
__global__ void __launch_bounds__(kRowThreads) softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                                  int T) {
  __shared__ float red[4];
  const int64_t r = (int64_t)blockIdx.y * T + blockIdx.x;
  const float* p = P + r * T;
  float* d = dP + r * T;
  float dot = 0.f;
  for (int j = threadIdx.x; j < T; j += kRowThreads) dot += p[j] * d[j];
  dot = row_sum(dot, red);
  for (int j = threadIdx.x; j < T; j += kRowThreads) d[j] = p[j] * (d[j] - dot);
}

__global__ void relu_fwd_kernel(float* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* y, float* dy, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) dy[i] = 0.f;
}
__global__ void add_inplace_kernel(float* a, const float* b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    a[i] += b[i];
}

static int ew_blocks(int64_t n) { return (int)std::min<int64_t>(4096, (n + 255) / 256); }

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                                 int64_t rows, int32_t C, float eps, float* sum_out, float* y, float* mean,
                                 float* rstd, void* stream_) {
  PK2_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0, "layernorm_fwd: bad args");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)rows), dim3(kRowThreads), 0, static_cast<hipStream_t>(stream_),
                     x, res, gamma, beta, C, eps, sum_out, y, mean, rstd);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_layernorm_bwd(const float* dy, const float* s, const float* mean, const float* rstd,
                                 const float* gamma, int64_t rows, int32_t C, float* ds, float* dgamma, float* dbeta,
                                 void* stream_) {
  PK2_REQUIRE(dy && s && mean && rstd && gamma && ds && dgamma && dbeta && rows > 0 && C > 0, "layernorm_bwd: bad args");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int col_blocks = (C + 63) / 64;
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 63) / 64, 1024 / col_blocks));
  const int rpb = (int)((rows + splits - 1) / splits);
  hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(col_blocks, (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0,
                     stream, dy, s, mean, rstd, rows, C, rpb, dgamma, dbeta);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)rows), dim3(kRowThreads), 0, stream, dy, s, mean, rstd, gamma,
                     C, ds);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_softmax_mask_fwd(float* scores, const float* src_mask, const uint8_t* key_padding, int32_t B,
                                    int32_t H, int32_t T, void* stream_) {
  PK2_REQUIRE(scores && B > 0 && H > 0 && T > 0 && (int64_t)B * H <= 65535, "softmax_mask_fwd: bad args");
  hipLaunchKernelGGL(softmax_mask_kernel, dim3(T, B * H), dim3(kRowThreads), 0, static_cast<hipStream_t>(stream_), scores,
                     src_mask, key_padding, H, T);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_softmax_bwd(const float* P, float* dP, int32_t BH, int32_t T, void* stream_) {
  PK2_REQUIRE(P && dP && BH > 0 && T > 0 && BH <= 65535, "softmax_bwd: bad args");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(T, BH), dim3(kRowThreads), 0, static_cast<hipStream_t>(stream_), P, dP, T);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_relu_fwd(float* x, int64_t n, void* stream_) {
  PK2_REQUIRE(x && n >= 0, "relu_fwd: bad args");
  if (n) hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), x, n);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_relu_bwd(const float* y, float* dy, int64_t n, void* stream_) {
  PK2_REQUIRE(y && dy && n >= 0, "relu_bwd: bad args");
  if (n) hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), y, dy, n);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_add_inplace(float* a, const float* b, int64_t n, void* stream_) {
  PK2_REQUIRE(a && b && n >= 0, "add_inplace: bad args");
  if (n) hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), a, b, n);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
