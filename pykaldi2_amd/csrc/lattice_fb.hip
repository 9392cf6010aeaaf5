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
// The values of the frame being ACCUMULATED live in LDS (an array of doubles; the log-adds are 64-bit LDS
// compare-and-swaps, the epsilon levels of the frame never leave the CU); a finished frame is written back for the
// posterior pass.  A frame with more than `cap` tokens (frame 0: a token per word) is accumulated in global memory.
// (With everything in global memory a frame cost ~14 us of dependent round trips and CAS loops through L2.)
// Round 5 (18 ms per step on this chain, 7.5 us per frame and direction): a frame was still a string of
// exposed round trips -- link records loaded when the frame began, then the source values of the frame before gathered
// from global memory behind them, then the epsilon records and, behind those, their levels.  Now
//  * TWO frames live in LDS: the one being accumulated and the one before it, which the emitting links gather from
//    (cap = half the array each; a frame with more tokens takes the global path as before, per frame);
//  * the link records of the NEXT frame (emitting links, epsilon links and their levels, the frame's bounds) are loaded
//    one frame ahead into registers: the chain of a frame is LDS traffic, barriers and the write-back stores.
constexpr int kFbEps = 2;       // epsilon links per thread kept in registers over the levels of a frame (a pruned lattice keeps ~1.5 links per thread and frame)
constexpr int kFbEmit = 2;      // emitting links per thread loaded a frame ahead
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

// __syncthreads() is a workgroup-scope release / acquire: it waits for EVERY outstanding global load of the wave first
// (s_waitcnt vmcnt(0)) -- also for the records loaded a frame ahead, which would then be waited for where they were
// issued.  Between phases that only exchange data through LDS the barrier orders LDS accesses alone.
__device__ __forceinline__ void fb_barrier(bool lds_only) {
  if (lds_only) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// The bounds of a frame (its tokens, its emitting and epsilon link segments, its epsilon depth): loaded TWO frames ahead, so
// that the record loads of the next frame can be issued without waiting for them.
struct FbSc { int base, cnt, m0, m1, e0, e1, nlev; };
struct FbEmit { int m0, m1; int src[kFbEmit], dst[kFbEmit]; double like[kFbEmit]; };
struct FbEpsR { int e0, e1, nlev; int src[kFbEps], dst[kFbEps], lev[kFbEps]; double like[kFbEps]; };

__global__ void __launch_bounds__(kFbThreads) lat_fb_alpha_beta(FbParams p, int cap) {
  __shared__ double red[kFbWaves];
  const int n = blockIdx.y, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) { if (tid == 0 && blockIdx.x == 0) p.out[n] = NAN; return; }
  const FbView v = fb_view(p, n, U);
  const int T = v.T;
  const int fT0 = v.ftok[T], fT1 = v.ftok[T + 1];
  const bool fwd = blockIdx.x == 0;
  double* val = fwd ? v.alpha : v.beta;          // the recursion's values in global memory
  double* A = lat_fb_smem;                       // the frame being accumulated / finished
  double* P = lat_fb_smem + cap;                 // its neighbour: the frame before it (alpha), the one being built (beta)
  for (int i = tid; i < v.nt; i += kFbThreads) { val[i] = -INFINITY; (fwd ? v.af : v.ab)[i] = 0.0; }
  if (fwd)
    for (int t = tid; t < T; t += kFbThreads) v.ref_post[t] = 0.0;
  __syncthreads();
  auto load_sc = [&](int t, FbSc& c) {
    c.base = c.cnt = c.m0 = c.m1 = c.e0 = c.e1 = c.nlev = 0;
    if (t < 0 || t > T) return;
    c.base = v.ftok[t]; c.cnt = v.ftok[t + 1] - c.base;
    if (t > 0) { c.m0 = v.seg[2 * t - 1]; c.m1 = c.m0 + v.kept[2 * t - 1]; }
    c.e0 = v.seg[2 * t]; c.e1 = c.e0 + v.kept[2 * t]; c.nlev = v.maxlev[t];
  };
  // the emitting links t-1 -> t (segment 2t-1), the first kFbEmit per thread
  auto load_emit = [&](const FbSc& c, FbEmit& m) {
    m.m0 = c.m0; m.m1 = c.m1;
#pragma unroll
    for (int q = 0; q < kFbEmit; ++q) {
      const int l = m.m0 + tid + q * kFbThreads;
      m.src[q] = -1; m.dst[q] = 0; m.like[q] = 0.0;
      if (l < m.m1) { const int4 r = v.lrec[l]; m.src[q] = r.x; m.dst[q] = r.y; m.like[q] = link_like(p, v, r, l); }
    }
  };
  // the epsilon links inside frame t (segment 2t), the first kFbEps per thread -- and, in a second step issued later (the
  // gather needs the record: issued right behind it, the wave would wait for the record on the spot), the level of each
  // link's source token
  auto load_eps = [&](const FbSc& c, FbEpsR& e) {
    e.e0 = c.e0; e.e1 = c.e1; e.nlev = c.nlev;
#pragma unroll
    for (int q = 0; q < kFbEps; ++q) {
      const int l = e.e0 + tid + q * kFbThreads;
      e.src[q] = -1; e.dst[q] = 0; e.lev[q] = -1; e.like[q] = 0.0;
      if (e.nlev > 0 && l < e.e1) { const int4 r = v.lrec[l]; e.src[q] = r.x; e.dst[q] = r.y; e.like[q] = link_like(p, v, r, l); }
    }
  };
  auto load_eps_levels = [&](FbEpsR& e) {
#pragma unroll
    for (int q = 0; q < kFbEps; ++q)
      if (e.src[q] >= 0) e.lev[q] = v.tl[e.src[q]];
  };
  // Epsilon links of frame t inside the LDS array A (or global memory), level by level; alpha: source level ascending and
  // values flow src -> dst, beta: descending and dst -> src.
  auto eps_levels = [&](const FbEpsR& e, int base, bool lds) {
    if (e.nlev <= 0 || e.e1 <= e.e0) return;
    auto one = [&](int s, int d, double like) {
      const int from = fwd ? s : d, to = fwd ? d : s;
      if (lds) lds_log_add(&A[to - base], A[from - base] + like);
      else atomic_log_add(&val[to], ldc(&val[from]) + like);
    };
    for (int k = 0; k < e.nlev; ++k) {
      const int lev = fwd ? k : e.nlev - 1 - k;
#pragma unroll
      for (int q = 0; q < kFbEps; ++q)
        if (e.src[q] >= 0 && e.lev[q] == lev) one(e.src[q], e.dst[q], e.like[q]);
      for (int l = e.e0 + tid + kFbEps * kFbThreads; l < e.e1; l += kFbThreads) {
        const int4 r = v.lrec[l];
        if (v.tl[r.x] == lev) one(r.x, r.y, link_like(p, v, r, l));
      }
      fb_barrier(lds);
    }
  };
  FbEmit cm, nm; FbEpsR ce, ne;
  FbSc s0, s1, s2;               // the frame at hand, the next one, the one after it (in the direction of the recursion)
  // (-DPK2_FB_PROFILE: thread 0's time in the phases of a forward frame, printed for utterance 1 -- tools/gpu_fb.sh)
#ifdef PK2_FB_PROFILE
  long long ph[6] = {0, 0, 0, 0, 0, 0}, last = wall_clock64(); long long nlevs = 0, nlinks = 0, neps = 0, nglobal = 0, maxcnt = 0;
#define FB_T(k) do { const long long now_ = wall_clock64(); ph[k] += now_ - last; last = now_; } while (0)
#else
#define FB_T(k) do { } while (0)
#endif
  if (fwd) {
    load_sc(0, s0); load_sc(1, s1);
    load_emit(s0, cm);           // (frame 0 has no emitting links: empty)
    load_eps(s0, ce);
    load_eps_levels(ce);
    int pbase = 0; bool plds = false;
    for (int t = 0; t <= T; ++t) {
      const int base = s0.base, cnt = s0.cnt;
      const bool lds = cnt <= cap;
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) A[i] = (t == 0 && i == 0) ? 0.0 : -INFINITY;
      else if (t == 0 && tid == 0)
        val[0] = 0.0;
      load_sc(t + 2, s2);            // bounds two frames ahead, records one frame ahead: on their way while this frame is worked on
      load_emit(s1, nm);
      load_eps(s1, ne);
      fb_barrier(lds);
      FB_T(0);
#ifdef PK2_FB_PROFILE
      nlevs += ce.nlev; nlinks += cm.m1 - cm.m0; neps += ce.e1 - ce.e0; nglobal += lds ? 0 : 1; maxcnt = cnt > maxcnt ? cnt : maxcnt;
#endif
      if (t > 0) {
        auto push = [&](int s, int d, double like) {
          const double x = (plds ? P[s - pbase] : ldc(&val[s])) + like;
          if (lds) lds_log_add(&A[d - base], x); else atomic_log_add(&val[d], x);
        };
#pragma unroll
        for (int q = 0; q < kFbEmit; ++q)
          if (cm.src[q] >= 0) push(cm.src[q], cm.dst[q], cm.like[q]);
        for (int l = cm.m0 + tid + kFbEmit * kFbThreads; l < cm.m1; l += kFbThreads) {
          const int4 r = v.lrec[l];
          push(r.x, r.y, link_like(p, v, r, l));
        }
        fb_barrier(lds);
      }
      FB_T(1);
      eps_levels(ce, base, lds);
      FB_T(2);
      load_eps_levels(ne);         // (the records have arrived by now)
      FB_T(3);
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) val[base + i] = A[i];        // (for the posterior pass: nobody here waits for it)
      else
        __syncthreads();        // a frame on the global path: its values are in memory before the next frame gathers them
      FB_T(4);
      double* tmp = A; A = P; P = tmp;
      pbase = base; plds = lds;
      cm = nm; ce = ne; s0 = s1; s1 = s2;
    }
    __syncthreads();             // the last frame's values are in memory
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
#ifdef PK2_FB_PROFILE
    if (tid == 0 && n == 1) printf("lat_fb alpha utt %d, %d frames, 10 ns ticks per frame: init+prefetch+barrier %lld | emitting %lld | eps levels %lld | level loads %lld | write-back %lld ; per frame: %.1f levels, %.0f emitting links, %.0f epsilon links, %.0f tokens; %lld frames beyond the LDS arrays (cap %d, largest frame %lld)\n", n, T,
                                ph[0] / (T + 1), ph[1] / (T + 1), ph[2] / (T + 1), ph[3] / (T + 1), ph[4] / (T + 1), (double)nlevs / (T + 1), (double)nlinks / (T + 1), (double)neps / (T + 1), (double)v.nt / (T + 1), nglobal, cap, maxcnt);
#endif
  } else {
    int base = fT0, cnt = fT1 - fT0;
    bool lds = cnt <= cap;
    for (int i = tid; i < cnt; i += kFbThreads) {
      const double b0 = v.tf[base + i] < INFINITY ? final_like(p, v, base + i) : -INFINITY;
      if (lds) A[i] = b0; else val[base + i] = b0;
    }
    load_sc(T, s0); load_sc(T - 1, s1);
    load_eps(s0, ce);
    load_eps_levels(ce);
    load_emit(s0, cm);
    __syncthreads();
    for (int t = T; t >= 0; --t) {
      load_sc(t - 2, s2);
      load_eps(s1, ne);              // the records of frame t-1: on their way while this frame is finished
      load_emit(s1, nm);
      eps_levels(ce, base, lds);
      load_eps_levels(ne);
      if (lds)
        for (int i = tid; i < cnt; i += kFbThreads) val[base + i] = A[i];
      if (t > 0) {
        const int pbase = s1.base, pcnt = s1.cnt;
        const bool plds = pcnt <= cap;
        if (plds)
          for (int i = tid; i < pcnt; i += kFbThreads) P[i] = -INFINITY;
        fb_barrier(lds && plds);  // (P is initialised; a frame on the global path: its values are in memory)
        auto push = [&](int s, int d, double like) {
          const double x = (lds ? A[d - base] : ldc(&val[d])) + like;
          if (plds) lds_log_add(&P[s - pbase], x); else atomic_log_add(&val[s], x);
        };
#pragma unroll
        for (int q = 0; q < kFbEmit; ++q)
          if (cm.src[q] >= 0) push(cm.src[q], cm.dst[q], cm.like[q]);
        for (int l = cm.m0 + tid + kFbEmit * kFbThreads; l < cm.m1; l += kFbThreads) {
          const int4 r = v.lrec[l];
          push(r.x, r.y, link_like(p, v, r, l));
        }
        fb_barrier(plds);
        double* tmp = A; A = P; P = tmp;
        base = pbase; cnt = pcnt; lds = plds;
        cm = nm; ce = ne; s0 = s1; s1 = s2;
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
  constexpr int kFbCap = 19456;                    // doubles of LDS (152 KB): two frames of kFbCap / 2 tokens each
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lat_fb_alpha_beta), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kFbCap * (int)sizeof(double)));
    attr = true;
  }
  const char* cap_env = getenv("PK2_LAT_FIN_CAP");       // (test hook, shared with the pruning pass of the decoder)
  const int cap = cap_env ? std::max(0, std::min(kFbCap / 2, atoi(cap_env))) : kFbCap / 2;
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
