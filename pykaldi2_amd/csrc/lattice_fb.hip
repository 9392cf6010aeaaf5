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
// A lattice is processed frame by frame: the emitting links t-1 -> t in one parallel sweep (log-add through a 64-bit
// compare-and-swap), then the epsilon links inside frame t level by level of the epsilon DAG (levels computed by
// the pruning pass), workgroup barriers in between.  The alpha and the beta recursion of an utterance do not depend
// on each other: they run side by side in two workgroups (so do the two expected-accuracy recursions of sMBR / MPFE),
// and the posteriors, which are local to a frame, are computed by 64 workgroups per utterance.
#include <algorithm>
#include <cmath>
#include <cstdlib>

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

// Per-utterance views shared by the kernels below.
struct FbView {
  int T, nt;
  const int32_t* ftok; const int32_t* seg; const int32_t* kept; const int32_t* maxlev;
  const int4* lrec; const float* lac; const int32_t* tl; const float* tf;
  double* alpha; double* beta; double* af; double* ab;
  const int32_t* ref; double* ref_post;
  LatFrame* F;
};
__device__ __forceinline__ FbView fb_view(const FbParams& p, int n, const LatUtt& U) {
  FbView v;
  v.T = U.T; v.nt = U.n_tok;
  v.ftok = p.L.frame_tok + U.frame_base; v.seg = p.L.seg_off + U.frame_base;
  v.kept = p.L.seg_kept + U.frame_base; v.maxlev = p.L.frame_maxlev + U.frame_base;
  v.lrec = p.L.link_rec + U.link_base;      // {src token, dst token, transition-id, graph cost bits}
  v.lac = p.L.link_ac + U.link_base;
  v.tl = p.L.tok_level + U.tok_base; v.tf = p.L.tok_final + U.tok_base;
  v.alpha = p.L.alpha + U.tok_base; v.beta = p.L.beta + U.tok_base;
  v.af = p.L.acc_f + U.tok_base; v.ab = p.L.acc_b + U.tok_base;
  v.ref = p.ref_tids + (int64_t)n * p.ref_stride;
  v.ref_post = p.L.ref_post + U.frame_base;
  v.F = p.L.frame + n;
  return v;
}
// fst::ScaleLattice stores the scaled weights as floats; the forward-backward then sums them in double
__device__ __forceinline__ double link_like(const FbParams& p, const FbView& v, const int4& r, int l) {
  return -((double)(float)(p.lm_scale * (double)__int_as_float(r.w)) + (double)(float)(p.ac_scale * (double)v.lac[l]));
}
__device__ __forceinline__ double final_like(const FbParams& p, const FbView& v, int i) {
  return -(double)(float)(p.lm_scale * (double)v.tf[i]);
}

// alpha (workgroup x = 0) and beta (x = 1) are independent recursions: they run side by side.  x = 0 also leaves the
// total log-likelihood in the utterance's frame state.
// The values of the frame being ACCUMULATED live in LDS (one array of doubles; the log-adds are 64-bit LDS
// compare-and-swaps, the epsilon levels of the frame never leave the CU); a finished frame is written back, and the
// next frame reads its sources from there.  A frame with more than `cap` tokens (frame 0: a token per word) is
// accumulated in global memory as before.  (With everything in global memory a frame cost ~14 us of dependent round
// trips and CAS loops through L2.)
constexpr int kFbEps = 4;       // epsilon links per thread kept in registers over the levels of a frame
extern __shared__ __attribute__((aligned(16))) double lat_fb_smem[];

__device__ __forceinline__ void lds_log_add(double* addr, double v) {
  if (v == -INFINITY) return;
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *a, assumed;
  do {
    assumed = old;
    const double nv = log_add(__longlong_as_double((long long)assumed), v);
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(nv));
  } while (old != assumed);
}

__global__ void __launch_bounds__(kFbThreads) lat_fb_alpha_beta(FbParams p, int cap) {
  __shared__ double red[kFbWaves];
  double* cur = lat_fb_smem;
  const int n = blockIdx.y, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) { if (tid == 0 && blockIdx.x == 0) p.out[n] = NAN; return; }
  const FbView v = fb_view(p, n, U);
  const int T = v.T;
  const int fT0 = v.ftok[T], fT1 = v.ftok[T + 1];
  const bool fwd = blockIdx.x == 0;
  double* val = fwd ? v.alpha : v.beta;          // the recursion's values in global memory
  for (int i = tid; i < v.nt; i += kFbThreads) { val[i] = -INFINITY; (fwd ? v.af : v.ab)[i] = 0.0; }
  if (fwd)
    for (int t = tid; t < T; t += kFbThreads) v.ref_post[t] = 0.0;
  __syncthreads();
  // Epsilon links of frame t inside the LDS array (or global memory), level by level; dir = +1: source level ascending
  // and values flow src -> dst (alpha), -1: descending and dst -> src (beta).
  auto eps_levels = [&](int t, int base, bool lds) {
    const int e0 = v.seg[2 * t], e1 = e0 + v.kept[2 * t];
    const int nlev = v.maxlev[t];
    if (nlev <= 0 || e1 <= e0) return;
    int es[kFbEps], ed[kFbEps], el[kFbEps]; double ek[kFbEps];
#pragma unroll
    for (int q = 0; q < kFbEps; ++q) {
      const int l = e0 + tid + q * kFbThreads;
      es[q] = -1; ed[q] = 0; el[q] = -1; ek[q] = 0.0;
      if (l < e1) { const int4 r = v.lrec[l]; es[q] = r.x; ed[q] = r.y; el[q] = v.tl[r.x]; ek[q] = link_like(p, v, r, l); }
    }
    auto one = [&](int s, int d, double like) {
      const int from = fwd ? s : d, to = fwd ? d : s;
      if (lds) lds_log_add(&cur[to - base], cur[from - base] + like);
      else atomic_log_add(&val[to], ldc(&val[from]) + like);
    };
    for (int k = 0; k < nlev; ++k) {
      const int lev = fwd ? k : nlev - 1 - k;
#pragma unroll
      for (int q = 0; q < kFbEps; ++q)
        if (es[q] >= 0 && el[q] == lev) one(es[q], ed[q], ek[q]);
      for (int l = e0 + tid + kFbEps * kFbThreads; l < e1; l += kFbThreads) {
        const int4 r = v.lrec[l];
        if (v.tl[r.x] == lev) one(r.x, r.y, link_like(p, v, r, l));
      }
      __syncthreads();
    }
  };
  if (fwd) {
    for (int t = 0; t <= T; ++t) {
      const int base = v.ftok[t], cnt = v.ftok[t + 1] - base;
      const bool lds = cnt <= cap;
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) cur[i] = (t == 0 && i == 0) ? 0.0 : -INFINITY;
      else if (t == 0 && tid == 0)
        val[0] = 0.0;
      __syncthreads();
      if (t > 0) {
        const int m0 = v.seg[2 * t - 1], m1 = m0 + v.kept[2 * t - 1];
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 r = v.lrec[l];
          const double x = ldc(&val[r.x]) + link_like(p, v, r, l);
          if (lds) lds_log_add(&cur[r.y - base], x); else atomic_log_add(&val[r.y], x);
        }
        __syncthreads();
      }
      eps_levels(t, base, lds);
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) val[base + i] = cur[i];
      __syncthreads();          // (the frame's values are in memory before the next frame gathers them)
    }
    // total likelihood over the final tokens (stable log-sum-exp)
    double mx = -INFINITY;
    for (int i = fT0 + tid; i < fT1; i += kFbThreads)
      if (v.tf[i] < INFINITY) mx = fmax(mx, ldc(&v.alpha[i]) + final_like(p, v, i));
    mx = block_max_d(mx, red);
    double sm = 0.0;
    for (int i = fT0 + tid; i < fT1; i += kFbThreads)
      if (v.tf[i] < INFINITY) sm += exp(ldc(&v.alpha[i]) + final_like(p, v, i) - mx);
    sm = block_sum_d(sm, red);
    if (tid == 0) v.F->fb_tot = mx + log(sm);
  } else {
    int base = fT0, cnt = fT1 - fT0;
    bool lds = cnt <= cap;
    for (int i = tid; i < cnt; i += kFbThreads) {
      const double b0 = v.tf[base + i] < INFINITY ? final_like(p, v, base + i) : -INFINITY;
      if (lds) cur[i] = b0; else val[base + i] = b0;
    }
    __syncthreads();
    for (int t = T; t >= 0; --t) {
      eps_levels(t, base, lds);
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) val[base + i] = cur[i];
      __syncthreads();
      if (t > 0) {
        const int pbase = v.ftok[t - 1], pcnt = base - pbase;
        const bool plds = pcnt <= cap;
        if (plds)
          for (int i = tid; i < pcnt; i += kFbThreads) cur[i] = -INFINITY;
        __syncthreads();
        const int m0 = v.seg[2 * t - 1], m1 = m0 + v.kept[2 * t - 1];
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 r = v.lrec[l];
          const double x = ldc(&val[r.y]) + link_like(p, v, r, l);
          if (plds) lds_log_add(&cur[r.x - pbase], x); else atomic_log_add(&val[r.x], x);
        }
        __syncthreads();
        base = pbase; cnt = pcnt; lds = plds;
      }
    }
  }
}

// sMBR / MPFE: the expected-accuracy recursions, forward (x = 0, needs alpha) and backward (x = 1, needs beta).
__global__ void __launch_bounds__(kFbThreads) lat_fb_accuracy(FbParams p) {
  __shared__ double red[kFbWaves];
  const int n = blockIdx.y, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) return;
  const FbView v = fb_view(p, n, U);
  const int T = v.T;
  if (blockIdx.x == 0) {
    for (int t = 0; t <= T; ++t) {
      if (t > 0) {
        const int m0 = v.seg[2 * t - 1], m1 = m0 + v.kept[2 * t - 1];
        const int r = v.ref[t - 1];
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          const int s = q.x, d = q.y;
          atomicAdd(&v.af[d], exp(v.alpha[s] + link_like(p, v, q, l) - v.alpha[d]) * (ldc(&v.af[s]) + frame_acc(p, q.z, r)));
        }
        __syncthreads();
      }
      const int e0 = v.seg[2 * t], e1 = e0 + v.kept[2 * t];
      for (int lev = 0; lev < v.maxlev[t]; ++lev) {
        for (int l = e0 + tid; l < e1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          const int s = q.x, d = q.y;
          if (v.tl[s] == lev) atomicAdd(&v.af[d], exp(v.alpha[s] + link_like(p, v, q, l) - v.alpha[d]) * ldc(&v.af[s]));
        }
        __syncthreads();
      }
    }
    const int fT0 = v.ftok[T], fT1 = v.ftok[T + 1];
    const double tot = v.F->fb_tot;
    double sc = 0.0;
    for (int i = fT0 + tid; i < fT1; i += kFbThreads)
      if (v.tf[i] < INFINITY) sc += exp(v.alpha[i] + final_like(p, v, i) - tot) * ldc(&v.af[i]);
    sc = block_sum_d(sc, red);
    if (tid == 0) v.F->fb_score = sc;
  } else {
    for (int t = T; t >= 0; --t) {
      const int e0 = v.seg[2 * t], e1 = e0 + v.kept[2 * t];
      for (int lev = v.maxlev[t] - 1; lev >= 0; --lev) {
        for (int l = e0 + tid; l < e1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          const int s = q.x, d = q.y;
          const double bs = v.beta[s], bd = v.beta[d];
          if (v.tl[s] == lev && bs > -INFINITY && bd > -INFINITY) atomicAdd(&v.ab[s], exp(bd + link_like(p, v, q, l) - bs) * ldc(&v.ab[d]));
        }
        __syncthreads();
      }
      if (t > 0) {
        const int m0 = v.seg[2 * t - 1], m1 = m0 + v.kept[2 * t - 1];
        const int r = v.ref[t - 1];
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          const int s = q.x, d = q.y;
          const double bs = v.beta[s], bd = v.beta[d];
          if (bs > -INFINITY && bd > -INFINITY)
            atomicAdd(&v.ab[s], exp(bd + link_like(p, v, q, l) - bs) * (ldc(&v.ab[d]) + frame_acc(p, q.z, r)));
        }
        __syncthreads();
      }
    }
  }
}

// Posteriors, frame by frame in parallel (frames dealt round-robin to the workgroups of an utterance).
// MODE 0: MMI (numerator - denominator with MergePosteriors' drop_frames test), MODE 1: sMBR / MPFE.
template <int MODE>
__global__ void __launch_bounds__(kFbThreads) lat_fb_posteriors(FbParams p) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) return;
  const FbView v = fb_view(p, n, U);
  const double tot = v.F->fb_tot, tot_score = v.F->fb_score;
  float* post = p.post + (int64_t)n * p.post_seq_stride;
  if (blockIdx.x == 0 && tid == 0) p.out[n] = MODE == 0 ? tot : tot_score;
  for (int t = blockIdx.x; t < v.T; t += gridDim.x) {
    const int m0 = v.seg[2 * t + 1], m1 = m0 + v.kept[2 * t + 1];
    const int r = v.ref[t];
    float* row = post + (int64_t)t * p.post_frame_stride;
    if (MODE == 0) {
      // denominator posterior of the reference transition-id (MergePosteriors' drop_frames test)
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = v.lrec[l];
        if (q.z == r) atomicAdd(&v.ref_post[t], exp(v.alpha[q.x] + link_like(p, v, q, l) + v.beta[q.y] - tot));
      }
      __syncthreads();
      const bool drop = p.drop_frames && ldc(&v.ref_post[t]) == 0.0;
      if (!drop) {
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          atomicAdd(&row[p.tid2pdf[q.z]], -(float)exp(v.alpha[q.x] + link_like(p, v, q, l) + v.beta[q.y] - tot));
        }
        if (tid == 0) atomicAdd(&row[p.tid2pdf[r]], 1.0f);
      }
    } else {
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = v.lrec[l];
        const int s = q.x, d = q.y;
        const double bd = v.beta[d];
        if (bd == -INFINITY) continue;
        const double pr = exp(v.alpha[s] + link_like(p, v, q, l) + bd - tot);
        const double diff = v.af[s] + frame_acc(p, q.z, r) + v.ab[d] - tot_score;
        atomicAdd(&row[p.tid2pdf[q.z]], (float)(pr * diff));
      }
    }
  }
}

}  // namespace pk2

using namespace pk2;

static int fb_launch(int mode, const pk2_lattice_batch* b, void* workspace, FbParams& p, hipStream_t stream) {
  PK2_REQUIRE(b->decoded, "lattice forward-backward: pk2_lattice_decode has not run on this batch");
  lattice_carve(b, workspace, &p.L);
  const dim3 two(2, b->N), many(64, b->N), thr(kFbThreads);
  constexpr int kFbCap = 19456;                    // tokens of a frame the LDS array holds (152 KB of doubles)
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lat_fb_alpha_beta), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kFbCap * (int)sizeof(double)));
    attr = true;
  }
  const char* cap_env = getenv("PK2_LAT_FIN_CAP");       // (test hook, shared with the pruning pass of the decoder)
  const int cap = cap_env ? std::max(0, std::min(kFbCap, atoi(cap_env))) : kFbCap;
  hipLaunchKernelGGL(lat_fb_alpha_beta, two, thr, kFbCap * sizeof(double), stream, p, cap);
  if (mode == 0) {
    hipLaunchKernelGGL(lat_fb_posteriors<0>, many, thr, 0, stream, p);
  } else {
    hipLaunchKernelGGL(lat_fb_accuracy, two, thr, 0, stream, p);
    hipLaunchKernelGGL(lat_fb_posteriors<1>, many, thr, 0, stream, p);
  }
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
