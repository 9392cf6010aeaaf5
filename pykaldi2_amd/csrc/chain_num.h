// Numerator forward-backward: parameter block and the per-sequence device routine, shared by chain_num.hip (stand-alone
// launch) and chain_den.hip (where the numerator workgroups ride along with the occupancy kernel of the denominator:
// one workgroup per sequence is a 1 ms latency-bound job that would otherwise hold the stream on its own).
#pragma once
#include "chain_internal.h"

namespace pk2 {

constexpr int kNumThreads = 256;

template <int THREADS>
__device__ __forceinline__ float block_sum_f(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();      // (with one wave this also orders the frame's LDS atomics before the next frame's reads)
  if (THREADS == 64) return v;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < THREADS / 64; ++k) s += red[k];
  return s;
}

// Numerator forward-backward of sequence n by the first THREADS threads of the calling workgroup (the others must
// have left the kernel); smem: 2 * states + 8 floats.  A chain supervision has only a few arcs per frame, so the
// recursion is a chain of ~2T tiny dependent steps: with THREADS = 64 (one wave) a frame costs a wave reduction and
// a barrier that no other wave has to reach, instead of two 4-wave barriers.
template <int THREADS = kNumThreads>
__device__ __forceinline__ void num_fwd_bwd_body(const NumParams& p, int n, float* smem) {
  constexpr int kNumThreads = THREADS;
  const int tid = threadIdx.x;
  const int32_t* info = p.seqinfo + n * 8;
  const int fbase = info[0], T = info[1], ns = info[2], flo = info[3], fhi = info[4];
  float* al = smem;        // [ns] alpha, each frame's states carry that frame's (unknown) scale
  float* be = smem + ns;   // [ns]
  float* red = be + ns;    // [4]
  for (int i = tid; i < 2 * ns; i += kNumThreads) smem[i] = 0.f;
  __syncthreads();
  if (tid == 0) al[0] = 1.f;
  __syncthreads();

  const float* fmax = p.frame_max + (fbase - n);
  double logp = 0.0;
  float inv_prev = 1.f;
  for (int t = 0; t < T; ++t) {
    const int lo = p.frame_off[fbase + t], hi = p.frame_off[fbase + t + 1];
    const float m = fmax[t];
    float z = 0.f;
    for (int a = lo + tid; a < hi; a += kNumThreads) {
      const float v = al[p.arc_src[a]] * inv_prev * expf(p.score[a] - m);
      atomicAdd(&al[p.arc_dst[a]], v);
      z += v;
    }
    z = block_sum_f<THREADS>(z, red);  // (also orders the LDS atomics before the next frame's reads)
    logp += (double)m + log((double)z);
    inv_prev = 1.f / z;
  }
  // final states
  float zf = 0.f;
  for (int k = flo + tid; k < fhi; k += kNumThreads) {
    const int s = p.final_state[k];
    const float e = expf(-p.final_w[k]);
    zf += al[s] * inv_prev * e;
    be[s] = e;
  }
  zf = block_sum_f<THREADS>(zf, red);
  logp += log((double)zf);
  if (tid == 0) p.num_lp[n] = (float)logp;

  // backward: posteriors of a frame are normalised by their own sum
  float* grow = p.grad + (int64_t)n * p.gseq_stride;
  inv_prev = 1.f;
  for (int t = T - 1; t >= 0; --t) {
    const int lo = p.frame_off[fbase + t], hi = p.frame_off[fbase + t + 1];
    const float m = fmax[t];
    float zq = 0.f, zb = 0.f;
    // a frame holds at most a few arcs per thread; keep the products in registers
    float q[4]; int na = 0;
    for (int a = lo + tid; a < hi; a += kNumThreads) {
      const int s = p.arc_src[a];
      const float u = expf(p.score[a] - m) * be[p.arc_dst[a]] * inv_prev;
      atomicAdd(&be[s], u);
      const float qq = al[s] * u;
      if (na < 4) q[na] = qq;
      ++na;
      zq += qq; zb += u;
    }
    zq = block_sum_f<THREADS>(zq, red);
    zb = block_sum_f<THREADS>(zb, red);
    const float inv_q = p.scale / zq;
    int k = 0;
    for (int a = lo + tid; a < hi; a += kNumThreads, ++k) {
      float qq;
      if (k < 4) {
        qq = q[k];
      } else {  // > 1024 arcs in one frame: recompute (beta of the source is final by now,
                // so rebuild u from the destination side)
        qq = al[p.arc_src[a]] * expf(p.score[a] - m) * be[p.arc_dst[a]] * inv_prev;
      }
      atomicAdd(grow + (int64_t)t * p.gframe_stride + p.arc_pdf[a], qq * inv_q);
    }
    inv_prev = 1.f / zb;
  }
}

}  // namespace pk2
