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
template <int THREADS = kNumThreads, bool STAGE = false>
__device__ __forceinline__ void num_fwd_bwd_body(const NumParams& p, int n, float* smem) {
  constexpr int kNumThreads = THREADS;
  const int tid = threadIdx.x;
  const int32_t* info = p.seqinfo + n * 8;
  const int fbase = info[0], T = info[1], ns = info[2], flo = info[3], fhi = info[4];
  float* al = smem;        // [ns] alpha, each frame's states carry that frame's (unknown) scale
  float* be = smem + ns;   // [ns]
  float* red = be + ns;    // [8]
  for (int i = tid; i < 2 * ns; i += kNumThreads) smem[i] = 0.f;
  // STAGE: the sequence's frame table and arc arrays are copied into LDS first (one coalesced burst): inside the
  // recursion every global load would put ~1 us of latency into each of the 2T dependent frame steps.
  const int a0 = p.frame_off[fbase];
  int32_t* s_foff = reinterpret_cast<int32_t*>(red + 8);          // [T+1] relative to a0
  float* s_fmax = reinterpret_cast<float*>(s_foff + T + 1);       // [T]
  const int na = STAGE ? p.frame_off[fbase + T] - a0 : 0;
  int32_t* s_src = reinterpret_cast<int32_t*>(s_fmax + T);        // [na]
  int32_t* s_dst = s_src + na;
  int32_t* s_pdf = s_dst + na;
  float* s_score = reinterpret_cast<float*>(s_pdf + na);
  const float* fmax_g = p.frame_max + (fbase - n);
  if (STAGE) {
    for (int t = tid; t <= T; t += kNumThreads) s_foff[t] = p.frame_off[fbase + t] - a0;
    for (int t = tid; t < T; t += kNumThreads) s_fmax[t] = fmax_g[t];
    for (int a = tid; a < na; a += kNumThreads) {
      s_src[a] = p.arc_src[a0 + a]; s_dst[a] = p.arc_dst[a0 + a]; s_pdf[a] = p.arc_pdf[a0 + a]; s_score[a] = p.score[a0 + a];
    }
  }
  auto FOFF = [&](int t) { return STAGE ? s_foff[t] : p.frame_off[fbase + t]; };
  auto FMAX = [&](int t) { return STAGE ? s_fmax[t] : fmax_g[t]; };
  auto SRC = [&](int a) { return STAGE ? s_src[a] : p.arc_src[a]; };
  auto DST = [&](int a) { return STAGE ? s_dst[a] : p.arc_dst[a]; };
  auto PDF = [&](int a) { return STAGE ? s_pdf[a] : p.arc_pdf[a]; };
  auto SCORE = [&](int a) { return STAGE ? s_score[a] : p.score[a]; };
  __syncthreads();
  if (tid == 0) al[0] = 1.f;
  __syncthreads();

  double logp = 0.0;
  float inv_prev = 1.f;
  for (int t = 0; t < T; ++t) {
    const int lo = FOFF(t), hi = FOFF(t + 1);
    const float m = FMAX(t);
    float z = 0.f;
    for (int a = lo + tid; a < hi; a += kNumThreads) {
      const float v = al[SRC(a)] * inv_prev * expf(SCORE(a) - m);
      atomicAdd(&al[DST(a)], v);
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
    const int lo = FOFF(t), hi = FOFF(t + 1);
    const float m = FMAX(t);
    float zq = 0.f, zb = 0.f;
    // a frame holds at most a few arcs per thread; keep the products in registers
    float q[4]; int na_t = 0;
    for (int a = lo + tid; a < hi; a += kNumThreads) {
      const int s = SRC(a);
      const float u = expf(SCORE(a) - m) * be[DST(a)] * inv_prev;
      atomicAdd(&be[s], u);
      const float qq = al[s] * u;
      if (na_t < 4) q[na_t] = qq;
      ++na_t;
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
      } else {  // > 4 * THREADS arcs in one frame: recompute (beta of the source is final by now,
                // so rebuild u from the destination side)
        qq = al[SRC(a)] * expf(SCORE(a) - m) * be[DST(a)] * inv_prev;
      }
      atomicAdd(grow + (int64_t)t * p.gframe_stride + PDF(a), qq * inv_q);
    }
    inv_prev = 1.f / zb;
  }
}

// Two-wave variant of the staged recursion (threads 0..127 of the calling workgroup; the others must have left):
// the alpha chain (wave 0) and the beta chain (wave 1) do not depend on each other -- only the posteriors need
// both -- so they run side by side, each ordered by the in-order LDS pipeline of its own wave (no workgroup barrier
// inside the 2T frame steps), and the posteriors are formed afterwards, a frame per thread, from alpha and the
// per-arc backward factors kept in LDS.  smem: 2 * states + 8 + (2T + 1) + 5 * arcs floats.
//
// Round 2: what sat inside the 2T dependent frame steps and did not have to was moved out -- exp(score - frame max) of
// every arc is computed by all threads before the chains, the per-frame log z of the forward pass afterwards (the chain
// only stores z) -- and a frame's sum over its (usually <= 16) arcs is a 16-lane DPP reduction instead of six
// ds_bpermute rounds.  The state sums stay float atomics: a fixed-point form (fast integer LDS atomics) would lose the
// relative precision of small alphas / betas, which decides posteriors when past and future disagree.  The occupancy
// launch this rides on now takes 0.23 ms; the numerator's 2T-step tail behind it was ~0.4 ms.

// sum over the active lanes of a wave: 16-lane DPP reduction when the frame has at most 16 arcs (the usual case)
__device__ __forceinline__ float num_frame_sum(float v, int count) {
  if (count <= 16) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));   // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));   // row_shr:1
    return __shfl(v, 15, 64);                     // lane 15 holds the sum of lanes 0..15
  }
  return wave_sum(v);
}

// (kNumTwoWaveBarriers __syncthreads() inside: a caller whose workgroup has further waves alive lets those execute the same
// number of barriers -- den_persist2_kernel, where the numerators ride in the idle time of the teams that finish early.)
constexpr int kNumTwoWaveBarriers = 4;
__device__ __forceinline__ void num_fwd_bwd_two_waves(const NumParams& p, int n, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* info = p.seqinfo + n * 8;
  const int fbase = info[0], T = info[1], ns = info[2], flo = info[3], fhi = info[4];
  float* al = smem;                                          // [ns] alpha, each frame's states carry that frame's scale
  float* be = al + ns;                                       // [ns] beta
  float* red = be + ns;
  const int a0 = p.frame_off[fbase];
  const int na = p.frame_off[fbase + T] - a0;
  int32_t* s_foff = reinterpret_cast<int32_t*>(red + 8);
  float* s_fmax = reinterpret_cast<float*>(s_foff + T + 1);     // frame maxima, overwritten with the frame's z by the alpha chain
  int32_t* s_src = reinterpret_cast<int32_t*>(s_fmax + T);
  int32_t* s_dst = s_src + na;
  int32_t* s_pdf = s_dst + na;
  float* s_e = reinterpret_cast<float*>(s_pdf + na);            // exp(score - frame max)
  float* s_u = s_e + na;                                        // backward factor of every arc (any per-frame scale)
  const float* fmax_g = p.frame_max + (fbase - n);
  for (int i = tid; i < 2 * ns; i += 128) al[i] = 0.f;
  for (int t = tid; t <= T; t += 128) s_foff[t] = p.frame_off[fbase + t] - a0;
  double msum = 0.0;
  for (int t = tid; t < T; t += 128) { const float m = fmax_g[t]; s_fmax[t] = m; msum += (double)m; }
  for (int a = tid; a < na; a += 128) {
    s_src[a] = p.arc_src[a0 + a]; s_dst[a] = p.arc_dst[a0 + a]; s_pdf[a] = p.arc_pdf[a0 + a]; s_e[a] = p.score[a0 + a];
  }
  __syncthreads();
  for (int t = tid; t < T; t += 128) {               // a frame per thread: its arcs' exp(score - max)
    const float m = s_fmax[t];
    for (int a = s_foff[t]; a < s_foff[t + 1]; ++a) s_e[a] = __expf(s_e[a] - m);
  }
  if (tid == 0) al[0] = 1.f;
  for (int k = flo + tid; k < fhi; k += 128) be[p.final_state[k]] = __expf(-p.final_w[k]);
  __syncthreads();
  auto wave_order = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  float inv_last = 1.f;
  if (wave == 0) {
    float inv_prev = 1.f;                            // 1 / previous frame's sum
    for (int t = 0; t < T; ++t) {
      const int lo = s_foff[t], hi = s_foff[t + 1];
      float z = 0.f;
      for (int a = lo + lane; a < hi; a += 64) {
        const float v = al[s_src[a]] * inv_prev * s_e[a];
        atomicAdd(&al[s_dst[a]], v);
        z += v;
      }
      z = num_frame_sum(z, hi - lo);
      wave_order();
      if (lane == 0) s_fmax[t] = z;
      inv_prev = 1.f / z;
    }
    inv_last = inv_prev;
  } else {
    float inv_prev = 1.f;
    for (int t = T - 1; t >= 0; --t) {
      const int lo = s_foff[t], hi = s_foff[t + 1];
      float zb = 0.f;
      for (int a = lo + lane; a < hi; a += 64) {
        const float u = s_e[a] * be[s_dst[a]] * inv_prev;
        atomicAdd(&be[s_src[a]], u);
        s_u[a] = u;
        zb += u;
      }
      zb = num_frame_sum(zb, hi - lo);
      wave_order();
      inv_prev = 1.f / zb;
    }
  }
  __syncthreads();
  // log-probability: sum of the frame maxima and of log z (all threads), plus the final states (wave 0 knows 1 / z[T-1])
  double lsum = msum;
  for (int t = tid; t < T; t += 128) lsum += log((double)s_fmax[t]);
  lsum = wave_sum_d(lsum);
  double* redd = reinterpret_cast<double*>(red);
  if (lane == 0) redd[wave] = lsum;
  __syncthreads();
  if (wave == 0) {
    float zf = 0.f;
    for (int k = flo + lane; k < fhi; k += 64) zf += al[p.final_state[k]] * inv_last * __expf(-p.final_w[k]);
    zf = wave_sum(zf);
    if (lane == 0) p.num_lp[n] = (float)(redd[0] + redd[1] + log((double)zf));
  }
  // posteriors: normalised inside each frame, so the frames' unknown scales cancel
  float* grow = p.grad + (int64_t)n * p.gseq_stride;
  for (int t = tid; t < T; t += 128) {
    const int lo = s_foff[t], hi = s_foff[t + 1];
    float zq = 0.f;
    for (int a = lo; a < hi; ++a) zq += al[s_src[a]] * s_u[a];
    const float inv_q = p.scale / zq;
    for (int a = lo; a < hi; ++a) atomicAdd(grow + (int64_t)t * p.gframe_stride + s_pdf[a], al[s_src[a]] * s_u[a] * inv_q);
  }
}

}  // namespace pk2
