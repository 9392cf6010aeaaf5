// Reverberation + additive noise of one utterance on the device (single channel): the arithmetic of the reference's
// dynamic data simulation, simulation/_distorter.py (Distorter.apply_rir :118-154, Distorter.add_noise :86-116,
// _comp_noise_scale_given_snr :28-32, _NoiseSampler.sample_noise :36-58) as driven by _Simulator.simulate
// (simulation/simulation.py:55-178) for one speech source, which the reference runs with numpy in DataLoader
// workers (data/sr_dataset.py:321-345).  The random draws (SNR, noise position, which files) stay on the host.
//
//   pk2_sim_apply_rir   time-domain convolution with the room impulse response, "synchronised" output
//                       (the reference convolves by FFT in float64; float32 direct form here, LDS-tiled)
//   pk2_sim_power       sum of squares (float64) and max |x|
//   pk2_sim_add_noise   noise scaled to the requested SNR and added at its sampled position
//   pk2_sim_gain_norm   0.5 / max|x| gain normalisation
// The statistics stay in device memory between the calls: no host round trip inside a simulated utterance.
#include <algorithm>

#include "common.h"

namespace pk2 {

constexpr int kSimThreads = 128;
constexpr int kSimOut = 4;                         // outputs per thread
constexpr int kSimTile = kSimThreads * kSimOut;    // outputs per workgroup: small, so that a 12 s utterance makes
                                                   // more workgroups than the GPU has CUs and stagings overlap
constexpr int kSimTaps = 1024;                     // taps staged per pass (4096 FMAs per thread between two barriers)

// out[i] = sum_j rir[j] * wav[i + base - j]  (wav is zero outside [0, n))
__global__ void __launch_bounds__(kSimThreads) sim_apply_rir_kernel(const float* __restrict__ wav, int64_t n,
                                                                    const float* __restrict__ rir, int k, int64_t base,
                                                                    float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_rir[kSimTaps];
  __shared__ __attribute__((aligned(16))) float s_wav[kSimTile + kSimTaps];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kSimTile;
  float acc[kSimOut] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < k; j0 += kSimTaps) {
    const int nt = min(kSimTaps, k - j0);
    // taps j0 .. j0+nt-1 need wav[i0 + base - j0 - (nt-1) .. i0 + kSimTile - 1 + base - j0]
    const int64_t w0 = i0 + base - j0 - (kSimTaps - 1);
    __syncthreads();
    for (int q = tid; q < kSimTaps; q += kSimThreads) s_rir[q] = q < nt ? rir[j0 + q] : 0.f;
    for (int q = tid; q < kSimTile + kSimTaps; q += kSimThreads) {
      const int64_t w = w0 + q;
      s_wav[q] = (w >= 0 && w < n) ? wav[w] : 0.f;
    }
    __syncthreads();
    // thread `tid` owns the 4 consecutive outputs o = 4 tid + r.  Tap j0 + jj of output o reads
    // s_wav[o + (kSimTaps - 1) - jj]; for 4 taps jj0 .. jj0+3 the 4 outputs need the 7 samples
    // e[0..6] = s_wav[4 tid + 252 - jj0 ...]: two aligned 16-byte LDS reads (conflict-free across lanes) plus one
    // broadcast read of the taps feed 16 FMAs.
#pragma unroll 4
    for (int jj0 = 0; jj0 < kSimTaps; jj0 += 4) {
      const float4 h = *reinterpret_cast<const float4*>(&s_rir[jj0]);
      const float4 lo = *reinterpret_cast<const float4*>(&s_wav[4 * tid + (kSimTaps - 4) - jj0]);
      const float4 hi = *reinterpret_cast<const float4*>(&s_wav[4 * tid + kSimTaps - jj0]);
      const float e[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const float ht[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
      for (int r = 0; r < kSimOut; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[r] = fmaf(ht[t], e[r + 3 - t], acc[r]);
    }
  }
  const int64_t i = i0 + 4 * tid;
  if (i + 3 < n) {
    *reinterpret_cast<float4*>(out + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
#pragma unroll
    for (int r = 0; r < kSimOut; ++r)
      if (i + r < n) out[i + r] = acc[r];
  }
}

__global__ void __launch_bounds__(256) sim_power_kernel(const float* __restrict__ x, int64_t n, double* stats) {
  __shared__ double s_sum[4];
  __shared__ float s_max[4];
  double sum = 0.0;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    sum += (double)v * (double)v;
    mx = fmaxf(mx, fabsf(v));
  }
  sum = wave_sum_d(sum);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    const double m = (double)fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(&stats[1]), (unsigned long long)__double_as_longlong(m));
  }
}

__global__ void __launch_bounds__(256) sim_add_noise_kernel(float* __restrict__ mixed, int64_t n, const float* __restrict__ noise,
                                                            int64_t m, int64_t start, float snr_db, const double* sig_stats,
                                                            const double* noise_stats) {
  // _comp_noise_scale_given_snr: sqrt(Px / Pn * 10^(-snr / 10)), powers = mean squares over the whole arrays
  const double px = sig_stats[0] / (double)n, pn = noise_stats[0] / (double)m;
  const float scale = (float)sqrt(px / pn * pow(10.0, -(double)snr_db / 10.0));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v;
    if (m <= n) v = (i >= start && i < start + m) ? noise[i - start] : 0.f;   // shorter noise placed at `start`
    else v = noise[start + i];                                                  // longer noise cropped from `start`
    mixed[i] += scale * v;
  }
}

__global__ void __launch_bounds__(256) sim_gain_norm_kernel(float* __restrict__ x, int64_t n, const double* stats) {
  const float g = (float)(0.5 / stats[1]);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= g;
}

static inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 2048); }

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_sim_apply_rir(const float* wav, int64_t n, const float* rir, int32_t k, int32_t delay, float* out,
                                 void* stream_) {
  PK2_REQUIRE(wav && rir && out && n > 0 && k > 0 && delay >= 0 && delay < k, "sim_apply_rir: bad arguments");
  PK2_REQUIRE(wav != out, "sim_apply_rir: in-place operation is not supported");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // sync = True keeps reverb[delay - 1 : delay + n - 1] (simulation/_distorter.py:147-148); delay = 0 would make the
  // reference's slice start at -1 (an empty result): the direct path is kept at sample 0 instead
  const int64_t base = delay > 0 ? delay - 1 : 0;
  hipLaunchKernelGGL(sim_apply_rir_kernel, dim3((unsigned)((n + kSimTile - 1) / kSimTile)), dim3(kSimThreads), 0, stream,
                     wav, n, rir, k, base, out);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_sim_power(const float* x, int64_t n, double* stats, void* stream_) {
  PK2_REQUIRE(x && stats && n > 0, "sim_power: bad arguments");
  hipLaunchKernelGGL(sim_power_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), x, n, stats);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_sim_add_noise(float* mixed, int64_t n, const float* noise, int64_t m, int64_t start, float snr_db,
                                 const double* sig_stats, const double* noise_stats, void* stream_) {
  PK2_REQUIRE(mixed && noise && sig_stats && noise_stats && n > 0 && m > 0, "sim_add_noise: bad arguments");
  PK2_REQUIRE(start >= 0 && (m <= n ? start + m <= n : start + n <= m), "sim_add_noise: noise position %lld outside its range",
              (long long)start);
  hipLaunchKernelGGL(sim_add_noise_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), mixed, n,
                     noise, m, start, snr_db, sig_stats, noise_stats);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_sim_gain_norm(float* x, int64_t n, const double* stats, void* stream_) {
  PK2_REQUIRE(x && stats && n > 0, "sim_gain_norm: bad arguments");
  hipLaunchKernelGGL(sim_gain_norm_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream_), x, n, stats);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
