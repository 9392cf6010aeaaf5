// On-device 80-dim log-mel filterbank, CMN and frame subsampling (gfx950).
//
// Replaces, for waveforms already resident in HBM:
//   DataGeneratorTrain._logfbank_extractor   reference data/sr_dataset.py:279-296
//   stft / _enframe                           reference simulation/freq_analysis.py:113-150,41-110
//   preprocess.cmn(axis=0)                    reference reader/preprocess.py:34-41 (sr_dataset.py:365-366)
//   roll + unfold subsampling                 reference bin/train_chain.py:251-255
// Semantics restated from those lines: y = wav[1:] - 0.96*wav[:-1] (float32); frames of 400
// samples every 160, the tail zero padded so T = ceil((N-1-400)/160)+1; symmetric Hamming(400);
// 512-point FFT -> 257 bins; power; mel matrix = rows of mel80_window.txt with every FFT-bin
// column normalised to sum 1 over the filters (all-zero columns stay zero); * 32768^2, + 1, ln.
//
// One 256-thread workgroup per frame: the frame is windowed into LDS, transformed with a
// radix-2 FFT that never leaves LDS (9 stages, one butterfly per thread per stage), and the 80
// mel sums read the power spectrum from LDS; HBM traffic is the 4 B/sample wav read (frames
// overlap 2.5x but hit L2) and the 320 B/frame output.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

struct pk2_fbank {
  std::vector<float> melT;      // [257][80], column-normalised, times 32768^2
  std::vector<float> window;    // [400]
  std::vector<float> tw_re, tw_im;  // [256] exp(-2 pi i k / 512)
  float *d_melT = nullptr, *d_window = nullptr, *d_tw_re = nullptr, *d_tw_im = nullptr;
  bool uploaded = false;
};

namespace pk2 {

constexpr int kNfft = 512, kWin = 400, kHop = 160, kBins = 257, kMel = 80;

__device__ __forceinline__ unsigned rev9(unsigned i) { return __brev(i) >> 23; }

// The utterances of one launch (round 5: a launch per utterance left a 4-utterance minibatch with four 12-37 us kernels one
// after the other, each too small for the chip -- 0.19 ms of the LF-MMI step): sample and feature-row offsets by value, a
// workgroup per frame of the whole batch looks its utterance up.
constexpr int kFbankBatch = 32;
struct FbankBatch { int64_t wav_off[kFbankBatch + 1]; int32_t row_off[kFbankBatch + 1]; int32_t n; };

__global__ void __launch_bounds__(256) fbank_frame_kernel(FbankBatch b, const float* __restrict__ wav_all,
                                                          const float* __restrict__ window,
                                                          const float* __restrict__ tw_re,
                                                          const float* __restrict__ tw_im,
                                                          const float* __restrict__ melT,
                                                          float* __restrict__ feats_all) {
  __shared__ float re[kNfft], im[kNfft];
  __shared__ float pw[kBins + 3];
  int u = 0;
  while (u + 1 < b.n && (int)blockIdx.x >= b.row_off[u + 1]) ++u;
  const float* __restrict__ wav = wav_all + b.wav_off[u];
  const int64_t n_samples = b.wav_off[u + 1] - b.wav_off[u];
  float* __restrict__ feats = feats_all + (int64_t)b.row_off[u] * kMel;
  const int t = (int)blockIdx.x - b.row_off[u], tid = threadIdx.x;
  const int64_t m = n_samples - 1;            // length of the pre-emphasised signal
  const int64_t s0 = (int64_t)t * kHop;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = tid + h * 256;
    float v = 0.f;
    if (i < kWin && s0 + i < m) {
      // float32 like numpy: round the product, then the difference (no FMA contraction)
      const float pe = __fmul_rn(0.96f, wav[s0 + i]);
      v = __fsub_rn(wav[s0 + i + 1], pe) * window[i];
    }
    const unsigned r = rev9((unsigned)i);
    re[r] = v;
    im[r] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 1; s <= 9; ++s) {
    const int half = 1 << (s - 1);
    const int grp = tid >> (s - 1), pos = tid & (half - 1);
    const int i0 = (grp << s) + pos, i1 = i0 + half;
    const int tw = pos << (9 - s);
    const float wr = tw_re[tw], wi = tw_im[tw];
    const float xr = re[i1], xi = im[i1];
    const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
    const float ar = re[i0], ai = im[i0];
    re[i1] = ar - tr; im[i1] = ai - ti;
    re[i0] = ar + tr; im[i0] = ai + ti;
    __syncthreads();
  }
  for (int b = tid; b < kBins; b += 256) pw[b] = re[b] * re[b] + im[b] * im[b];
  __syncthreads();
  if (tid < kMel) {
    float acc = 0.f;
    for (int b = 0; b < kBins; ++b) acc += pw[b] * melT[b * kMel + tid];
    feats[(int64_t)t * kMel + tid] = logf(acc + 1.0f);
  }
}

// feats[t][f] -= mean_t feats[t][f]   (one workgroup per utterance: 12 partitions of the time axis x 80 bins, four
// independent double accumulators per thread -- with 4 partitions and one dependent chain this 0.4 MB job took 145 us)
constexpr int kCmnParts = 12;
__global__ void __launch_bounds__(kCmnParts * kMel) cmn_kernel(float* __restrict__ feats, const int64_t* __restrict__ row_off) {
  __shared__ double part[kCmnParts][kMel];
  const int n = blockIdx.x;
  const int64_t r0 = row_off[n], T = row_off[n + 1] - r0;
  const int f = threadIdx.x % kMel, q = threadIdx.x / kMel;
  float* base = feats + r0 * kMel;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t t = q;
  for (; t + 3 * kCmnParts < T; t += 4 * kCmnParts) {
    a0 += (double)base[t * kMel + f];
    a1 += (double)base[(t + kCmnParts) * kMel + f];
    a2 += (double)base[(t + 2 * kCmnParts) * kMel + f];
    a3 += (double)base[(t + 3 * kCmnParts) * kMel + f];
  }
  for (; t < T; t += kCmnParts) a0 += (double)base[t * kMel + f];
  part[q][f] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < kCmnParts; ++k) sum += part[k][f];
  const float mean = (float)(sum / (double)T);
  for (int64_t u = q; u < T; u += kCmnParts) base[u * kMel + f] -= mean;
}

// x[(n, j)][:] = feats_n[(j*sub - shift) mod max_t]  (zero beyond the utterance's frames)
__global__ void __launch_bounds__(256) pad_roll_subsample_kernel(const float* __restrict__ feats,
                                                                 const int64_t* __restrict__ row_off, int N,
                                                                 int max_t, int shift, int sub, float* __restrict__ x,
                                                                 int out_t, int time_major) {
  const int64_t total = (int64_t)N * out_t * (kMel / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % (kMel / 4));
    const int64_t rj = i / (kMel / 4);
    int n, j;
    if (time_major) { j = (int)(rj / N); n = (int)(rj % N); }
    else            { n = (int)(rj / out_t); j = (int)(rj % out_t); }
    // torch.roll(x, shift, 1)[j*sub] = x[(j*sub - shift) mod max_t]
    int64_t src_t = ((int64_t)j * sub - shift) % max_t;
    if (src_t < 0) src_t += max_t;
    const int64_t T = row_off[n + 1] - row_off[n];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src_t < T) v = *reinterpret_cast<const float4*>(feats + (row_off[n] + src_t) * kMel + c * 4);
    *reinterpret_cast<float4*>(x + rj * kMel + c * 4) = v;
  }
}

// y[r][d] = (x[r][d] - mean[d]) / std[d]   (GlobalMeanVarianceNormalization.apply_on_ndarray,
// reference reader/preprocess.py:211-229)
__global__ void __launch_bounds__(256) mvn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, int64_t total, int D,
                                                        float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    y[i] = (x[i] - mean[d]) / stdv[d];
  }
}

static int fbank_upload(pk2_fbank* fb) {
  if (fb->uploaded) return PK2_OK;
  auto up = [&](const std::vector<float>& h, float** d) -> int {
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(float)));
    PK2_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return PK2_OK;
  };
  int rc;
  if ((rc = up(fb->melT, &fb->d_melT))) return rc;
  if ((rc = up(fb->window, &fb->d_window))) return rc;
  if ((rc = up(fb->tw_re, &fb->d_tw_re))) return rc;
  if ((rc = up(fb->tw_im, &fb->d_tw_im))) return rc;
  fb->uploaded = true;
  return PK2_OK;
}

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_fbank_create(const float* mel, pk2_fbank** out) {
  PK2_REQUIRE(mel && out, "fbank_create: null pointer");
  auto* fb = new pk2_fbank();
  // column normalisation (reference data/sr_dataset.py:283-286): t1 = sum over filters, zeros -> -1
  fb->melT.assign((size_t)kBins * kMel, 0.f);
  for (int b = 0; b < kBins; ++b) {
    float t1 = 0.f;  // np.sum(axis=0) of float32 rows: sequential float32 adds
    for (int f = 0; f < kMel; ++f) t1 += mel[f * kBins + b];
    if (t1 == 0.f) t1 = -1.f;
    const float inv = 1.0f / t1;             // np.diag(1 / t1), float32
    for (int f = 0; f < kMel; ++f)           // window.dot(inv) then * 32768**2, all float32
      fb->melT[(size_t)b * kMel + f] = (mel[f * kBins + b] * inv) * 1073741824.0f;
  }
  fb->window.resize(kWin);
  for (int i = 0; i < kWin; ++i)  // np.hamming(400): 0.54 - 0.46 cos(2 pi i / (M-1))
    fb->window[i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / (kWin - 1)));
  fb->tw_re.resize(kNfft / 2); fb->tw_im.resize(kNfft / 2);
  for (int k = 0; k < kNfft / 2; ++k) {
    fb->tw_re[k] = (float)cos(-2.0 * M_PI * k / kNfft);
    fb->tw_im[k] = (float)sin(-2.0 * M_PI * k / kNfft);
  }
  *out = fb;
  return PK2_OK;
}

extern "C" int pk2_fbank_destroy(pk2_fbank* fb) {
  if (!fb) return PK2_OK;
  if (fb->uploaded) {
    (void)hipFree(fb->d_melT); (void)hipFree(fb->d_window); (void)hipFree(fb->d_tw_re); (void)hipFree(fb->d_tw_im);
  }
  delete fb;
  return PK2_OK;
}

extern "C" int32_t pk2_fbank_num_frames(int64_t num_samples) {
  const int64_t m = num_samples - 1;
  if (m <= kWin) return m >= 1 ? 1 : 0;
  return (int32_t)((m - kWin + kHop - 1) / kHop + 1);
}

extern "C" int pk2_fbank_compute(const pk2_fbank* fbc, const float* wav, const int64_t* wav_off,
                                 int32_t num_utts, float* feats, const int64_t* feat_row_off,
                                 int32_t apply_cmn, void* stream_) {
  PK2_REQUIRE(fbc && wav && wav_off && feats && feat_row_off && num_utts > 0, "fbank_compute: bad args");
  pk2_fbank* fb = const_cast<pk2_fbank*>(fbc);
  int rc = fbank_upload(fb);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // feat_row_off is a DEVICE array of num_utts+1 int64 (also used by cmn / subsample kernels);
  // wav_off is a HOST array.
  int64_t row0 = 0;                      // rows of an utterance start at the host-known prefix of frame counts
  for (int n0 = 0; n0 < num_utts; n0 += kFbankBatch) {
    pk2::FbankBatch b;
    b.n = std::min<int>(kFbankBatch, num_utts - n0);
    int rows = 0;
    for (int k = 0; k < b.n; ++k) {
      const int64_t ns = wav_off[n0 + k + 1] - wav_off[n0 + k];
      const int T = pk2_fbank_num_frames(ns);
      PK2_REQUIRE(T > 0, "fbank_compute: utterance %d has %lld samples", n0 + k, (long long)ns);
      b.wav_off[k] = wav_off[n0 + k] - wav_off[n0]; b.row_off[k] = rows;
      rows += T;
    }
    b.wav_off[b.n] = wav_off[n0 + b.n] - wav_off[n0]; b.row_off[b.n] = rows;
    for (int k = b.n + 1; k <= kFbankBatch; ++k) { b.wav_off[k] = b.wav_off[b.n]; b.row_off[k] = rows; }
    hipLaunchKernelGGL(fbank_frame_kernel, dim3(rows), dim3(256), 0, stream, b, wav + wav_off[n0], fb->d_window,
                       fb->d_tw_re, fb->d_tw_im, fb->d_melT, feats + row0 * kMel);
    row0 += rows;
  }
  if (apply_cmn)
    hipLaunchKernelGGL(cmn_kernel, dim3(num_utts), dim3(kCmnParts * kMel), 0, stream, feats, feat_row_off);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_pad_roll_subsample(const float* feats, const int64_t* feat_row_off, int32_t num_utts,
                                      int32_t max_t, int32_t shift, int32_t subsample, float* x, int32_t out_t,
                                      int32_t time_major, void* stream_) {
  PK2_REQUIRE(feats && feat_row_off && x && num_utts > 0 && max_t > 0 && subsample > 0 && out_t > 0,
              "pad_roll_subsample: bad args");
  const int64_t total = (int64_t)num_utts * out_t * (kMel / 4);
  const int blocks = (int)std::min<int64_t>(4096, (total + 255) / 256);
  hipLaunchKernelGGL(pad_roll_subsample_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     feats, feat_row_off, num_utts, max_t, shift, subsample, x, out_t, time_major);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_mvn_apply(const float* x, const float* mean, const float* stdv, int64_t rows, int32_t dim,
                             float* y, void* stream_) {
  PK2_REQUIRE(x && mean && stdv && y && rows >= 0 && dim > 0, "mvn_apply: bad args");
  const int64_t total = rows * dim;
  if (total == 0) return PK2_OK;
  const int blocks = (int)std::min<int64_t>(4096, (total + 255) / 256);
  hipLaunchKernelGGL(mvn_apply_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), x, mean, stdv,
                     total, dim, y);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
