// Lattice forward-backward on gfx950 for the MMI, sMBR and MPFE criteria, on the frame-layered lattices that
// lattice_decode.hip leaves in the workspace.
//
// Replaces, per minibatch and on the device (reference ops/ops.py:56-63, 134-143):
//   scale_lattice(lattice_scale(1.0, 0.2)); lattice_forward_backward_mmi(trans_model, lat, trans_ids, True, False, True)
//   lattice_forward_backward_mpe_variants(trans_model, silence_phones, lat, trans_ids, criterion, True)
//   Posterior.to_pdf_matrix(trans_model)
// Arithmetic follows Kaldi's lattice-functions.cc as restated in oracle/lattice_ref.py: log-domain alpha /
// beta in float64; the expected-accuracy recursions (alpha_smbr / beta_smbr) in float64.
//
// One workgroup per utterance.  A lattice is processed frame by frame: the emitting links t-1 -> t in one
// parallel sweep (log-add through a 64-bit compare-and-swap), then the epsilon links inside frame t level by
// level of the epsilon DAG (levels computed by the decoder), workgroup barriers in between.
#include <cmath>

#include "lattice_internal.h"

namespace pk2 {

constexpr int kFbThreads = 1024;
constexpr int kFbWaves = kFbThreads / 64;

template <typename T>
__device__ __forceinline__ T ldc(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double log_add(double a, double b) {
  const double m = fmax(a, b);
  if (m == -INFINITY) return m;
  return m + log1p(exp(-fabs(a - b)));
}
__device__ __forceinline__ void atomic_log_add(double* addr, double v) {
  if (v == -INFINITY) return;
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = ldc(a), assumed;
  do {
    assumed = old;
    const double nv = log_add(__longlong_as_double((long long)assumed), v);
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(nv));
  } while (old != assumed);
}

struct FbParams {
  LatPtrs L;
  const int32_t* ref_tids; int64_t ref_stride;
  const int32_t* tid2pdf; const int32_t* tid2phone; const uint8_t* phone_sil;
  int32_t criterion, one_silence_class, drop_frames;
  double lm_scale, ac_scale;
  float* post; int64_t post_seq_stride, post_frame_stride;
  double* out;   // [N] lat_like (MMI) or expected accuracy (MPE)
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int k = 0; k < kFbWaves; ++k) r += red[k];
  return r;
}
__device__ __forceinline__ double block_max_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int k = 1; k < kFbWaves; ++k) r = fmax(r, red[k]);
  return r;
}

// Frame accuracy of an arc (LatticeForwardBackwardMpeVariants): sMBR compares pdfs, MPFE phones.
__device__ __forceinline__ double frame_acc(const FbParams& p, int tid_arc, int tid_ref) {
  const int phone = p.tid2phone[tid_arc], ref_phone = p.tid2phone[tid_ref];
  const bool phone_sil = p.phone_sil[phone] != 0, both_sil = phone_sil && p.phone_sil[ref_phone] != 0;
  bool ok;
  if (p.criterion == 1) {
    ok = p.one_silence_class ? (phone == ref_phone || both_sil) : (phone == ref_phone && !phone_sil);
  } else {
    const int pdf = p.tid2pdf[tid_arc], ref_pdf = p.tid2pdf[tid_ref];
    ok = p.one_silence_class ? (pdf == ref_pdf || both_sil) : (pdf == ref_pdf && !phone_sil);
  }
  return ok ? 1.0 : 0.0;
}

// MODE 0: MMI, MODE 1: sMBR / MPFE.
template <int MODE>
__global__ void __launch_bounds__(kFbThreads) lat_fb_kernel(FbParams p) {
  __shared__ double red[kFbWaves];
  const int n = blockIdx.x, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) { if (tid == 0) p.out[n] = NAN; return; }
  const int T = U.T, nt = U.n_tok;
  const int32_t* ftok = p.L.frame_tok + U.frame_base;
  const int32_t* seg = p.L.seg_off + U.frame_base;
  const int32_t* kept = p.L.seg_kept + U.frame_base;
  const int32_t* maxlev = p.L.frame_maxlev + U.frame_base;
  const int4* lrec = p.L.link_rec + U.link_base;      // {src token, dst token, transition-id, graph cost bits}
  const float* lac = p.L.link_ac + U.link_base;
  const int32_t* tl = p.L.tok_level + U.tok_base;
  const float* tf = p.L.tok_final + U.tok_base;
  double* alpha = p.L.alpha + U.tok_base; double* beta = p.L.beta + U.tok_base;
  double* af = p.L.acc_f + U.tok_base; double* ab = p.L.acc_b + U.tok_base;
  const int32_t* ref = p.ref_tids + (int64_t)n * p.ref_stride;
  double* ref_post = p.L.ref_post + U.frame_base;
  // fst::ScaleLattice stores the scaled weights as floats; the forward-backward then sums them in double
  auto like = [&](const int4& r, int l) {
    return -((double)(float)(p.lm_scale * (double)__int_as_float(r.w)) + (double)(float)(p.ac_scale * (double)lac[l]));
  };
  auto final_like = [&](int i) { return -(double)(float)(p.lm_scale * (double)tf[i]); };

  for (int i = tid; i < nt; i += kFbThreads) { alpha[i] = -INFINITY; beta[i] = -INFINITY; af[i] = 0.0; ab[i] = 0.0; }
  for (int t = tid; t < T; t += kFbThreads) ref_post[t] = 0.0;
  __syncthreads();
  if (tid == 0) alpha[0] = 0.0;
  __syncthreads();

  // ---- forward ----
  for (int t = 0; t <= T; ++t) {
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = m0 + kept[2 * t - 1];
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 r = lrec[l];
        atomic_log_add(&alpha[r.y], ldc(&alpha[r.x]) + like(r, l));
      }
      __syncthreads();
    }
    const int e0 = seg[2 * t], e1 = e0 + kept[2 * t];
    for (int lev = 0; lev < maxlev[t]; ++lev) {
      for (int l = e0 + tid; l < e1; l += kFbThreads) {
        const int4 r = lrec[l];
        if (tl[r.x] == lev) atomic_log_add(&alpha[r.y], ldc(&alpha[r.x]) + like(r, l));
      }
      __syncthreads();
    }
  }
  // total likelihood over the final tokens (stable log-sum-exp)
  const int fT0 = ftok[T], fT1 = ftok[T + 1];
  double mx = -INFINITY;
  for (int i = fT0 + tid; i < fT1; i += kFbThreads)
    if (tf[i] < INFINITY) mx = fmax(mx, ldc(&alpha[i]) + final_like(i));
  mx = block_max_d(mx, red);
  double sm = 0.0;
  for (int i = fT0 + tid; i < fT1; i += kFbThreads)
    if (tf[i] < INFINITY) sm += exp(ldc(&alpha[i]) + final_like(i) - mx);
  sm = block_sum_d(sm, red);
  const double tot = mx + log(sm);
  // ---- backward ----
  for (int i = fT0 + tid; i < fT1; i += kFbThreads)
    if (tf[i] < INFINITY) beta[i] = final_like(i);
  __syncthreads();
  for (int t = T; t >= 0; --t) {
    const int e0 = seg[2 * t], e1 = e0 + kept[2 * t];
    for (int lev = maxlev[t] - 1; lev >= 0; --lev) {
      for (int l = e0 + tid; l < e1; l += kFbThreads) {
        const int4 r = lrec[l];
        if (tl[r.x] == lev) atomic_log_add(&beta[r.x], ldc(&beta[r.y]) + like(r, l));
      }
      __syncthreads();
    }
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = m0 + kept[2 * t - 1];
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 r = lrec[l];
        atomic_log_add(&beta[r.x], ldc(&beta[r.y]) + like(r, l));
      }
      __syncthreads();
    }
  }
  float* post = p.post + (int64_t)n * p.post_seq_stride;

  if (MODE == 0) {
    // denominator posterior of the reference transition-id per frame (MergePosteriors' drop_frames test)
    for (int t = 0; t < T; ++t) {
      const int m0 = seg[2 * t + 1], m1 = m0 + kept[2 * t + 1];
      const int r = ref[t];
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = lrec[l];
        if (q.z == r) atomicAdd(&ref_post[t], exp(ldc(&alpha[q.x]) + like(q, l) + ldc(&beta[q.y]) - tot));
      }
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (p.drop_frames && ldc(&ref_post[t]) == 0.0) continue;
      const int m0 = seg[2 * t + 1], m1 = m0 + kept[2 * t + 1];
      float* row = post + (int64_t)t * p.post_frame_stride;
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = lrec[l];
        atomicAdd(&row[p.tid2pdf[q.z]], -(float)exp(ldc(&alpha[q.x]) + like(q, l) + ldc(&beta[q.y]) - tot));
      }
      if (tid == 0) atomicAdd(&row[p.tid2pdf[ref[t]]], 1.0f);
    }
    if (tid == 0) p.out[n] = tot;
    return;
  }

  // ---- sMBR / MPFE: expected accuracy forward (alpha_smbr) ----
  for (int t = 0; t <= T; ++t) {
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = m0 + kept[2 * t - 1];
      const int r = ref[t - 1];
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = lrec[l];
        const int s = q.x, d = q.y;
        atomicAdd(&af[d], exp(ldc(&alpha[s]) + like(q, l) - ldc(&alpha[d])) * (ldc(&af[s]) + frame_acc(p, q.z, r)));
      }
      __syncthreads();
    }
    const int e0 = seg[2 * t], e1 = e0 + kept[2 * t];
    for (int lev = 0; lev < maxlev[t]; ++lev) {
      for (int l = e0 + tid; l < e1; l += kFbThreads) {
        const int4 q = lrec[l];
        const int s = q.x, d = q.y;
        if (tl[s] == lev) atomicAdd(&af[d], exp(ldc(&alpha[s]) + like(q, l) - ldc(&alpha[d])) * ldc(&af[s]));
      }
      __syncthreads();
    }
  }
  double sc = 0.0;
  for (int i = fT0 + tid; i < fT1; i += kFbThreads)
    if (tf[i] < INFINITY) sc += exp(ldc(&alpha[i]) + final_like(i) - tot) * ldc(&af[i]);
  const double tot_score = block_sum_d(sc, red);
  // ---- expected accuracy backward (beta_smbr) ----
  for (int t = T; t >= 0; --t) {
    const int e0 = seg[2 * t], e1 = e0 + kept[2 * t];
    for (int lev = maxlev[t] - 1; lev >= 0; --lev) {
      for (int l = e0 + tid; l < e1; l += kFbThreads) {
        const int4 q = lrec[l];
        const int s = q.x, d = q.y;
        const double bs = ldc(&beta[s]), bd = ldc(&beta[d]);
        if (tl[s] == lev && bs > -INFINITY && bd > -INFINITY) atomicAdd(&ab[s], exp(bd + like(q, l) - bs) * ldc(&ab[d]));
      }
      __syncthreads();
    }
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = m0 + kept[2 * t - 1];
      const int r = ref[t - 1];
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = lrec[l];
        const int s = q.x, d = q.y;
        const double bs = ldc(&beta[s]), bd = ldc(&beta[d]);
        if (bs > -INFINITY && bd > -INFINITY)
          atomicAdd(&ab[s], exp(bd + like(q, l) - bs) * (ldc(&ab[d]) + frame_acc(p, q.z, r)));
      }
      __syncthreads();
    }
  }
  for (int t = 0; t < T; ++t) {
    const int m0 = seg[2 * t + 1], m1 = m0 + kept[2 * t + 1];
    const int r = ref[t];
    float* row = post + (int64_t)t * p.post_frame_stride;
    for (int l = m0 + tid; l < m1; l += kFbThreads) {
      const int4 q = lrec[l];
      const int s = q.x, d = q.y;
      const double bd = ldc(&beta[d]);
      if (bd == -INFINITY) continue;
      const double pr = exp(ldc(&alpha[s]) + like(q, l) + bd - tot);
      const double diff = ldc(&af[s]) + frame_acc(p, q.z, r) + ldc(&ab[d]) - tot_score;
      atomicAdd(&row[p.tid2pdf[q.z]], (float)(pr * diff));
    }
  }
  if (tid == 0) p.out[n] = tot_score;
}

}  // namespace pk2

using namespace pk2;

static int fb_launch(int mode, const pk2_lattice_batch* b, void* workspace, FbParams& p, hipStream_t stream) {
  PK2_REQUIRE(b->decoded, "lattice forward-backward: pk2_lattice_decode has not run on this batch");
  lattice_carve(b, workspace, &p.L);
  if (mode == 0) hipLaunchKernelGGL(lat_fb_kernel<0>, dim3(b->N), dim3(kFbThreads), 0, stream, p);
  else hipLaunchKernelGGL(lat_fb_kernel<1>, dim3(b->N), dim3(kFbThreads), 0, stream, p);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_lattice_mmi(const pk2_lattice_batch* b, void* workspace, const int32_t* ref_tids,
                               int64_t ref_stride, const int32_t* tid2pdf, double lm_scale, double acoustic_scale,
                               int32_t drop_frames, float* post, int64_t post_seq_stride, int64_t post_frame_stride,
                               double* lat_like, void* stream_) {
  PK2_REQUIRE(b && workspace && ref_tids && tid2pdf && post && lat_like, "lattice mmi: null pointer");
  FbParams p{};
  p.ref_tids = ref_tids; p.ref_stride = ref_stride; p.tid2pdf = tid2pdf;
  p.drop_frames = drop_frames; p.lm_scale = lm_scale; p.ac_scale = acoustic_scale;
  p.post = post; p.post_seq_stride = post_seq_stride; p.post_frame_stride = post_frame_stride; p.out = lat_like;
  return fb_launch(0, b, workspace, p, static_cast<hipStream_t>(stream_));
}

extern "C" int pk2_lattice_mpe(const pk2_lattice_batch* b, void* workspace, const int32_t* ref_tids,
                               int64_t ref_stride, const int32_t* tid2pdf, const int32_t* tid2phone,
                               const uint8_t* phone_is_silence, int32_t criterion, int32_t one_silence_class,
                               double lm_scale, double acoustic_scale, float* post, int64_t post_seq_stride,
                               int64_t post_frame_stride, double* score, void* stream_) {
  PK2_REQUIRE(b && workspace && ref_tids && tid2pdf && tid2phone && phone_is_silence && post && score,
              "lattice mpe: null pointer");
  PK2_REQUIRE(criterion == 0 || criterion == 1, "lattice mpe: criterion must be 0 (smbr) or 1 (mpfe)");
  FbParams p{};
  p.ref_tids = ref_tids; p.ref_stride = ref_stride; p.tid2pdf = tid2pdf; p.tid2phone = tid2phone;
  p.phone_sil = phone_is_silence; p.criterion = criterion; p.one_silence_class = one_silence_class;
  p.lm_scale = lm_scale; p.ac_scale = acoustic_scale;
  p.post = post; p.post_seq_stride = post_seq_stride; p.post_frame_stride = post_frame_stride; p.out = score;
  return fb_launch(1, b, workspace, p, static_cast<hipStream_t>(stream_));
}
