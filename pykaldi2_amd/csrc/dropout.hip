// Inverted dropout with a counter-based mask (gfx950, HBM-bound elementwise).
//
// Replaces the inter-layer dropout of nn.LSTM (reference models/lstm.py:49-54, dropout=0.2 in
// configs/ce.yaml:19): y = x * m / (1-p), m ~ Bernoulli(1-p) per element.  The mask is a pure function of
// (seed, element index), so the backward pass re-applies the same call to the gradient and no mask is
// stored.  (The random stream differs from PyTorch's Philox stream; dropout is only comparable in
// distribution.)
#include <algorithm>

#include "common.h"

namespace pk2 {

__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int64_t n, uint32_t keep_threshold, float scale,
                                                      uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t r = mix32(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    y[i] = r < keep_threshold ? x[i] * scale : 0.f;
  }
}

}  // namespace pk2

extern "C" int pk2_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream_) {
  PK2_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad args");
  if (n == 0) return PK2_OK;
  const double keep = 1.0 - (double)p;
  const uint32_t thr = (uint32_t)std::min<double>(4294967295.0, keep * 4294967296.0);
  const int blocks = (int)std::min<int64_t>(4096, (n + 255) / 256);
  hipLaunchKernelGGL(pk2::dropout_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y, n, thr,
                     (float)(1.0 / keep), seed);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
