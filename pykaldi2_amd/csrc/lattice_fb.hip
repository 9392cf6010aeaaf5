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
  double* lw; double* sca; double* scb;       // linear-domain recursion: link weights, per-frame log scales of alpha / beta
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
  v.lw = p.L.link_w + U.link_base; v.sca = p.L.fb_scale + U.frame_base; v.scb = p.L.fb_scale + p.L.frame_total + U.frame_base;
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

// Round 6, third session: the same two recursions in the LINEAR domain (the default; PK2_FB_LINEAR=0 keeps the kernel above).
// The phase timers of the kernel above (profiles/r06_fb_phases.txt) put 3.1 us of a forward frame's 7.4 into the log-adds of its
// ~1070 emitting links: every link is log1p(exp(.)) in double inside a 64-bit LDS compare-and-swap loop.  A frame's values span the
// decoder's beams (tens of nats), a double spans 1400: here a frame's values are kept as  exp(value - r)  with ONE log scale r per
// frame (every finished frame is normalised to maximum 1 for the next), a link multiplies by its weight w = exp(like) -- computed for
// all kept links by a parallel pass in front of the recursions (lat_fb_weights) -- and adds with the LDS's own 64-bit float add
// (ds_add_f64): no loop and no transcendental in the chain.  The finished frame is written back AS IT IS (linear) with its scale
// in fb_scale; a parallel pass behind the recursions (lat_fb_logs) takes the logs for the posterior passes, which are unchanged.
// (Taking the logs inside the chain was tried first: a frame has 8366 token slots of which ~1000 are in the pruned lattice, and
// 8 logs per thread and frame made the kernel 24 ms against 17.)  Frames that do not fit the LDS arrays (frame 0: a token per
// word) keep the log-domain path through global memory (scale = NaN: "logs already"); an LDS frame filled from one is converted
// when it is finished.  Values differ from the log-add chain's in the last bits only (same sums); the tests bound MMI / sMBR /
// MPFE against the oracle as before.
// (512 threads: at 1024 the compiler has 128 registers per thread and this kernel spilled 70 of them -- the link records held
// across a frame -- into scratch, whose reloads are round trips to L2; 256 registers hold four emitting and two epsilon links per
// thread, a frame of the bench lattices has ~1100)
constexpr int kLinThreads = 512, kLinWaves = kLinThreads / 64, kLinEmit = 4, kLinEps = 2;
__device__ __forceinline__ double lin_block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int q = 1; q < kLinWaves; ++q) r = fmax(r, red[q]);
  return r;
}
__device__ __forceinline__ double lin_block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int q = 0; q < kLinWaves; ++q) r += red[q];
  return r;
}
struct FbEmitW { int m0, m1; int src[kLinEmit], dst[kLinEmit]; double w[kLinEmit]; };      // (a log-domain frame reads its records again)
struct FbEpsW { int e0, e1, nlev; int src[kLinEps], dst[kLinEps], lev[kLinEps]; double w[kLinEps]; };

__device__ __forceinline__ double fb_block_max(double v, double* red) {     // red: kLinWaves doubles nobody else touches until the next barrier
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  fb_barrier(true);
  double r = red[0];
#pragma unroll
  for (int k = 1; k < kLinWaves; ++k) r = fmax(r, red[k]);
  return r;
}

// In front of the recursions, in parallel (64 workgroups per utterance): w[l] = exp(like(l)) for the kept links of every
// segment, alpha = beta = -inf for every token slot ("nothing reaches it": a linear value is never negative, so -inf stays
// recognisable), the accumulators of the posterior passes.  (The recursion kernel's own workgroup used to fill the utterance's
// 13.5 M token slots by itself: ~2 ms at the head of the longest chain of the step.)
__global__ void __launch_bounds__(256) lat_fb_prep(FbParams p, int with_acc) {
  const int n = blockIdx.y;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) return;
  const FbView v = fb_view(p, n, U);
  for (int sgm = blockIdx.x; sgm < 2 * (v.T + 1); sgm += gridDim.x) {
    const int l0 = v.seg[sgm], l1 = l0 + v.kept[sgm];
    for (int l = l0 + threadIdx.x; l < l1; l += 256) v.lw[l] = exp(link_like(p, v, v.lrec[l], l));
  }
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
  for (int64_t i = i0; i < v.nt; i += step) {
    v.alpha[i] = -INFINITY; v.beta[i] = -INFINITY;
    if (with_acc) { v.af[i] = 0.0; v.ab[i] = 0.0; }
  }
  for (int64_t t = i0; t < v.T; t += step) v.ref_post[t] = 0.0;
}

// linear value x at log scale r -> its log (r is NaN when x is a log already; -inf marks a token nothing reaches)
__device__ __forceinline__ double fb_log_of(double x, double r) { return (r == r && x != -INFINITY) ? r + log(x) : x; }

// alpha / beta of the frames the recursions left linear -> logs (sMBR / MPFE: the accuracy recursions read them in chains)
__global__ void __launch_bounds__(256) lat_fb_logs(FbParams p) {
  const int n = blockIdx.y;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) return;
  const FbView v = fb_view(p, n, U);
  for (int t = blockIdx.x; t <= v.T; t += gridDim.x) {
    const int i0 = v.ftok[t], i1 = v.ftok[t + 1];
    const double ra = v.sca[t], rb = v.scb[t];
    if (ra == ra) for (int i = i0 + threadIdx.x; i < i1; i += 256) v.alpha[i] = fb_log_of(v.alpha[i], ra);
    if (rb == rb) for (int i = i0 + threadIdx.x; i < i1; i += 256) v.beta[i] = fb_log_of(v.beta[i], rb);
    __syncthreads();
    if (threadIdx.x == 0) { v.sca[t] = __longlong_as_double(0x7ff8000000000000LL); v.scb[t] = v.sca[t]; }      // (logs now)
  }
}

// The recursions.  A frame has ~8400 token slots of which ~1000 are in the pruned lattice; every loop over the slots (zeroes,
// maximum, write-back: six per frame in the first version of this kernel) cost 0.5-1 us of the frame's 6.4-7.3.  The live tokens
// of frame t are exactly the ends of its kept links -- destinations of the emitting links t-1 -> t, both ends of the epsilon
// links inside t -- so those passes walk the LINK records the threads hold anyway (`touch`); duplicates are harmless (a zero,
// a maximum, the same value stored twice), and the normalisation is a factor in the next frame's gathers instead of a pass.
__global__ void __launch_bounds__(kLinThreads) lat_fb_alpha_beta_lin(FbParams p, int cap) {
  __shared__ double red[kLinWaves];
  __shared__ double redm[2][kLinWaves];
  const int n = blockIdx.y, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) { if (tid == 0 && blockIdx.x == 0) p.out[n] = NAN; return; }
  const FbView v = fb_view(p, n, U);
  const int T = v.T;
  const int fT0 = v.ftok[T], fT1 = v.ftok[T + 1];
  const bool fwd = blockIdx.x == 0;
  double* val = fwd ? v.alpha : v.beta;
  double* sc = fwd ? v.sca : v.scb;
  double* A = lat_fb_smem;
  double* P = lat_fb_smem + cap;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  int nmax = 0;
  auto load_sc = [&](int t, FbSc& c) {
    c.base = c.cnt = c.m0 = c.m1 = c.e0 = c.e1 = c.nlev = 0;
    if (t < 0 || t > T) return;
    c.base = v.ftok[t]; c.cnt = v.ftok[t + 1] - c.base;
    if (t > 0) { c.m0 = v.seg[2 * t - 1]; c.m1 = c.m0 + v.kept[2 * t - 1]; }
    c.e0 = v.seg[2 * t]; c.e1 = c.e0 + v.kept[2 * t]; c.nlev = v.maxlev[t];
  };
  auto load_emit = [&](const FbSc& c, FbEmitW& m) {
    m.m0 = c.m0; m.m1 = c.m1;
#pragma unroll
    for (int q = 0; q < kLinEmit; ++q) {
      const int l = m.m0 + tid + q * kLinThreads;
      m.src[q] = -1; m.dst[q] = 0; m.w[q] = 0.0;
      if (l < m.m1) { const int2 r = *reinterpret_cast<const int2*>(&v.lrec[l]); m.src[q] = r.x; m.dst[q] = r.y; m.w[q] = v.lw[l]; }
    }
  };
  // (the epsilon records of a frame are kept whatever its depth says: their ends are live tokens)
  auto load_eps = [&](const FbSc& c, FbEpsW& e) {
    e.e0 = c.e0; e.e1 = c.e1; e.nlev = c.nlev;
#pragma unroll
    for (int q = 0; q < kLinEps; ++q) {
      const int l = e.e0 + tid + q * kLinThreads;
      e.src[q] = -1; e.dst[q] = 0; e.lev[q] = -1; e.w[q] = 0.0;
      if (l < e.e1) { const int2 r = *reinterpret_cast<const int2*>(&v.lrec[l]); e.src[q] = r.x; e.dst[q] = r.y; e.w[q] = v.lw[l]; }
    }
  };
  auto load_eps_levels = [&](FbEpsW& e) {
#pragma unroll
    for (int q = 0; q < kLinEps; ++q)
      if (e.src[q] >= 0) e.lev[q] = v.tl[e.src[q]];
  };
  // f(token) for every live token of the frame whose emitting links (into it) are m and whose epsilon links are e
  auto touch = [&](const FbEmitW& m, const FbEpsW& e, auto f) {
#pragma unroll
    for (int q = 0; q < kLinEmit; ++q) if (m.src[q] >= 0) f(m.dst[q]);
    for (int l = m.m0 + tid + kLinEmit * kLinThreads; l < m.m1; l += kLinThreads) f(v.lrec[l].y);
#pragma unroll
    for (int q = 0; q < kLinEps; ++q) if (e.src[q] >= 0) { f(e.src[q]); f(e.dst[q]); }
    for (int l = e.e0 + tid + kLinEps * kLinThreads; l < e.e1; l += kLinThreads) { const int2 r = *reinterpret_cast<const int2*>(&v.lrec[l]); f(r.x); f(r.y); }
  };
  // Epsilon links of the frame in A (linear or log values) or in global memory (log), level by level.
  auto eps_levels = [&](const FbEpsW& e, int base, bool lds, bool lin) {
    if (e.nlev <= 0 || e.e1 <= e.e0) return;
    for (int k = 0; k < e.nlev; ++k) {
      const int lev = fwd ? k : e.nlev - 1 - k;
      if (lin) {
#pragma unroll
        for (int q = 0; q < kLinEps; ++q)
          if (e.src[q] >= 0 && e.lev[q] == lev) {
            const int from = fwd ? e.src[q] : e.dst[q], to = fwd ? e.dst[q] : e.src[q];
            const double x = A[from - base] * e.w[q];
            if (x != 0.0) atomicAdd(&A[to - base], x);
          }
        for (int l = e.e0 + tid + kLinEps * kLinThreads; l < e.e1; l += kLinThreads) {
          const int2 r = *reinterpret_cast<const int2*>(&v.lrec[l]);
          if (v.tl[r.x] == lev) {
            const int from = fwd ? r.x : r.y, to = fwd ? r.y : r.x;
            const double x = A[from - base] * v.lw[l];
            if (x != 0.0) atomicAdd(&A[to - base], x);
          }
        }
      } else {
        for (int l = e.e0 + tid; l < e.e1; l += kLinThreads) {
          const int4 r = v.lrec[l];
          if (v.tl[r.x] == lev) {
            const int from = fwd ? r.x : r.y, to = fwd ? r.y : r.x;
            const double like = link_like(p, v, r, l);
            if (lds) lds_log_add(&A[to - base], A[from - base] + like);
            else atomic_log_add(&val[to], ldc(&val[from]) + like);
          }
        }
      }
      fb_barrier(lds);
    }
  };
  // A (log values of a finished LDS frame, every slot) -> linear, maximum 1; returns the log scale
  auto to_linear = [&](int cnt) -> double {
    double mx = -INFINITY;
    for (int i = tid; i < cnt; i += kLinThreads) mx = fmax(mx, A[i]);
    mx = fb_block_max(mx, redm[nmax++ & 1]);
    if (mx == -INFINITY) mx = 0.0;
    for (int i = tid; i < cnt; i += kLinThreads) A[i] = exp(A[i] - mx);
    return mx;
  };
  FbEmitW cm, nm; FbEpsW ce, ne;
  FbSc s0, s1, s2;
#ifdef PK2_FB_PROFILE
  long long lph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, llast = wall_clock64();
#define FBL_T(k) do { const long long now_ = wall_clock64(); lph[k] += now_ - llast; llast = now_; } while (0)
#else
#define FBL_T(k) do { } while (0)
#endif
  if (fwd) {
    load_sc(0, s0); load_sc(1, s1);
    load_emit(s0, cm);
    load_eps(s0, ce);
    load_eps_levels(ce);
    int pbase = 0; bool plin = false;
    double r = 0.0, inv = 1.0;        // a linear value x of the frame in P stands for  r + log(x inv)  (inv: 1 / the frame's maximum)
    double rT = kNaN;                 // scale of the last frame as it lies in memory (for the total below)
    for (int t = 0; t <= T; ++t) {
      const int base = s0.base, cnt = s0.cnt;
      const bool lds = cnt <= cap, lin = lds && (t == 0 || plin);
      const bool whole = t == 0;      // (frame 0: the start token has no link into it -- every slot is walked, once)
      if (lin) {
        if (whole) { for (int i = tid; i < cnt; i += kLinThreads) A[i] = i == 0 ? 1.0 : 0.0; }
        else touch(cm, ce, [&](int tok) { A[tok - base] = 0.0; });
      } else if (lds) {
        for (int i = tid; i < cnt; i += kLinThreads) A[i] = (t == 0 && i == 0) ? 0.0 : -INFINITY;
      } else if (t == 0 && tid == 0) {
        val[0] = 0.0;
      }
      load_sc(t + 2, s2);
      load_emit(s1, nm);
      load_eps(s1, ne);
      if (lin) fb_barrier(true); else __syncthreads();      // (a log-domain frame gathers the frame before it from global memory)
      FBL_T(0);
      if (t > 0) {
        if (lin) {
#pragma unroll
          for (int q = 0; q < kLinEmit; ++q)
            if (cm.src[q] >= 0) { const double x = P[cm.src[q] - pbase] * inv * cm.w[q]; if (x != 0.0) atomicAdd(&A[cm.dst[q] - base], x); }
          for (int l = cm.m0 + tid + kLinEmit * kLinThreads; l < cm.m1; l += kLinThreads) {
            const int2 rr = *reinterpret_cast<const int2*>(&v.lrec[l]);
            const double x = P[rr.x - pbase] * inv * v.lw[l];
            if (x != 0.0) atomicAdd(&A[rr.y - base], x);
          }
        } else {
          // (the frame before it as LOGS from global memory: written back linear when it was an LDS frame)
          const double rp = sc[t - 1];
          for (int l = cm.m0 + tid; l < cm.m1; l += kLinThreads) {
            const int4 rr = v.lrec[l];
            const double x = fb_log_of(ldc(&val[rr.x]), rp) + link_like(p, v, rr, l);
            if (lds) lds_log_add(&A[rr.y - base], x); else atomic_log_add(&val[rr.y], x);
          }
        }
        fb_barrier(lds);
      }
      FBL_T(1);
      eps_levels(ce, base, lds, lin);
      FBL_T(2);
      load_eps_levels(ne);
      FBL_T(3);
      if (lin) {
        // finished: written back as it is, with its scale; its maximum becomes a factor of the next frame's gathers
        double mx = 0.0;
        if (whole) { for (int i = tid; i < cnt; i += kLinThreads) mx = fmax(mx, A[i]); }
        else touch(cm, ce, [&](int tok) { mx = fmax(mx, A[tok - base]); });
        mx = fb_block_max(mx, redm[nmax++ & 1]);
        FBL_T(4);
        if (whole) { for (int i = tid; i < cnt; i += kLinThreads) val[base + i] = A[i]; }
        else touch(cm, ce, [&](int tok) { val[tok] = A[tok - base]; });
        if (tid == 0) sc[t] = r;
        rT = r;
        inv = mx > 0.0 ? 1.0 / mx : 1.0;
        if (mx > 0.0) r += log(mx);
        plin = true;
      } else if (lds) {
        for (int i = tid; i < cnt; i += kLinThreads) val[base + i] = A[i];
        if (tid == 0) sc[t] = kNaN;
        rT = kNaN;
        r = to_linear(cnt); inv = 1.0;
        plin = true;
      } else {
        if (tid == 0) sc[t] = kNaN;
        rT = kNaN;
        __syncthreads();
        plin = false;
      }
      FBL_T(5);
      double* tmp = A; A = P; P = tmp;
      pbase = base;
      cm = nm; ce = ne; s0 = s1; s1 = s2;
    }
#ifdef PK2_FB_PROFILE
    if (tid == 0 && n == 1) printf("lat_fb_lin alpha utt %d, %d frames, 10 ns ticks per frame: zeroes+prefetch+barrier %lld | emitting %lld | eps levels %lld | level loads %lld | maximum %lld | write-back %lld\n", n, T,
                                lph[0] / (T + 1), lph[1] / (T + 1), lph[2] / (T + 1), lph[3] / (T + 1), lph[4] / (T + 1), lph[5] / (T + 1));
#endif
    __syncthreads();
    // total likelihood over the final tokens (stable log-sum-exp); the last frame may lie linear in memory
    double mx = -INFINITY;
    for (int i = fT0 + tid; i < fT1; i += kLinThreads)
      if (v.tf[i] < INFINITY) mx = fmax(mx, fb_log_of(ldc(&v.alpha[i]), rT) + final_like(p, v, i));
    mx = lin_block_max(mx, red);
    double sm = 0.0;
    for (int i = fT0 + tid; i < fT1; i += kLinThreads)
      if (v.tf[i] < INFINITY) sm += exp(fb_log_of(ldc(&v.alpha[i]), rT) + final_like(p, v, i) - mx);
    sm = lin_block_sum(sm, red);
    if (tid == 0) v.F->fb_tot = mx + log(sm);
  } else {
    int base = fT0, cnt = fT1 - fT0;
    bool lds = cnt <= cap, alin = false, whole = true;     // whole: every slot of A is meaningful (the last frame; a converted frame)
    double r = 0.0;
    for (int i = tid; i < cnt; i += kLinThreads) {
      const double b0 = v.tf[base + i] < INFINITY ? final_like(p, v, base + i) : -INFINITY;
      if (lds) A[i] = b0; else val[base + i] = b0;
    }
    load_sc(T, s0); load_sc(T - 1, s1);
    load_eps(s0, ce);
    load_eps_levels(ce);
    load_emit(s0, cm);
    __syncthreads();
    for (int t = T; t >= 0; --t) {
      if (lds && !alin) { r = to_linear(cnt); alin = true; whole = true; fb_barrier(true); }     // (filled in the log domain: the final costs, or from a frame in global memory)
      const bool lin = lds && alin;
      load_sc(t - 2, s2);
      load_eps(s1, ne);
      load_emit(s1, nm);
      FBL_T(0);
      eps_levels(ce, base, lds, lin);
      FBL_T(1);
      load_eps_levels(ne);
      const bool all = whole || t == 0;
      if (lin) {
        if (all) { for (int i = tid; i < cnt; i += kLinThreads) { const double a = A[i]; val[base + i] = a == 0.0 ? -INFINITY : a; } }
        else touch(cm, ce, [&](int tok) { val[tok] = A[tok - base]; });
      } else if (lds) {
        for (int i = tid; i < cnt; i += kLinThreads) val[base + i] = A[i];
      }
      if (tid == 0) sc[t] = lin ? r : kNaN;
      FBL_T(2);
      if (t > 0) {
        const int pbase = s1.base, pcnt = s1.cnt;
        const bool plds = pcnt <= cap, plin = lin && plds;
        if (plin) {
          // the live tokens of frame t-1: ends of ITS links, which are the records requested at the top of this iteration
          if (t - 1 == 0) { for (int i = tid; i < pcnt; i += kLinThreads) P[i] = 0.0; }
          else touch(nm, ne, [&](int tok) { P[tok - pbase] = 0.0; });
          double mx = 0.0;
          if (all) { for (int i = tid; i < cnt; i += kLinThreads) mx = fmax(mx, A[i]); }
          else touch(cm, ce, [&](int tok) { mx = fmax(mx, A[tok - base]); });
          mx = fb_block_max(mx, redm[nmax++ & 1]);              // (its barrier also orders P's zeroes)
          FBL_T(3);
          const double inv = mx > 0.0 ? 1.0 / mx : 1.0;
          if (mx > 0.0) r += log(mx);
#pragma unroll
          for (int q = 0; q < kLinEmit; ++q)
            if (cm.src[q] >= 0) { const double x = A[cm.dst[q] - base] * inv * cm.w[q]; if (x != 0.0) atomicAdd(&P[cm.src[q] - pbase], x); }
          for (int l = cm.m0 + tid + kLinEmit * kLinThreads; l < cm.m1; l += kLinThreads) {
            const int2 rr = *reinterpret_cast<const int2*>(&v.lrec[l]);
            const double x = A[rr.y - base] * inv * v.lw[l];
            if (x != 0.0) atomicAdd(&P[rr.x - pbase], x);
          }
        } else {
          if (plds)
            for (int i = tid; i < pcnt; i += kLinThreads) P[i] = -INFINITY;
          __syncthreads();           // (a log-domain push gathers this frame from global memory)
          for (int l = cm.m0 + tid; l < cm.m1; l += kLinThreads) {
            const int4 rr = v.lrec[l];
            const double x = fb_log_of(ldc(&val[rr.y]), lin ? r : kNaN) + link_like(p, v, rr, l);
            if (plds) lds_log_add(&P[rr.x - pbase], x); else atomic_log_add(&val[rr.x], x);
          }
        }
        fb_barrier(plds);
        FBL_T(4);
        double* tmp = A; A = P; P = tmp;
        base = pbase; cnt = pcnt; lds = plds; alin = plin; whole = false;
        cm = nm; ce = ne; s0 = s1; s1 = s2;
      }
    }
#ifdef PK2_FB_PROFILE
    if (tid == 0 && n == 1) printf("lat_fb_lin beta utt %d, %d frames, 10 ns ticks per frame: prefetch issue %lld | eps levels %lld | level loads + write-back %lld | zeroes + maximum %lld | emitting + barrier %lld\n", n, T,
                                lph[0] / (T + 1), lph[1] / (T + 1), lph[2] / (T + 1), lph[3] / (T + 1), lph[4] / (T + 1));
#endif
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
__global__ void __launch_bounds__(kFbThreads) lat_fb_posteriors(FbParams p, int lin) {
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
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
      // (the linear-domain recursions leave alpha of frame t / beta of frame t + 1 linear with their log scales: fb_log_of;
      // a scale of NaN = logs, which is also what the log-add kernel's fb_scale reads after lat_fb_logs or PK2_FB_LINEAR=0)
      const double ra = lin ? v.sca[t] : kNaN, rb = lin ? v.scb[t + 1] : kNaN;
      // denominator posterior of the reference transition-id (MergePosteriors' drop_frames test)
      for (int l = m0 + tid; l < m1; l += kFbThreads) {
        const int4 q = v.lrec[l];
        if (q.z == r) atomicAdd(&v.ref_post[t], exp(fb_log_of(v.alpha[q.x], ra) + link_like(p, v, q, l) + fb_log_of(v.beta[q.y], rb) - tot));
      }
      __syncthreads();
      const bool drop = p.drop_frames && ldc(&v.ref_post[t]) == 0.0;
      if (!drop) {
        for (int l = m0 + tid; l < m1; l += kFbThreads) {
          const int4 q = v.lrec[l];
          atomicAdd(&row[p.tid2pdf[q.z]], -(float)exp(fb_log_of(v.alpha[q.x], ra) + link_like(p, v, q, l) + fb_log_of(v.beta[q.y], rb) - tot));
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
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lat_fb_alpha_beta_lin), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kFbCap * (int)sizeof(double)));
    attr = true;
  }
  const char* cap_env = getenv("PK2_LAT_FIN_CAP");       // (test hook, shared with the pruning pass of the decoder)
  const int cap = cap_env ? std::max(0, std::min(kFbCap / 2, atoi(cap_env))) : kFbCap / 2;
  static const bool linear = [] { const char* e = getenv("PK2_FB_LINEAR"); return !(e && atoi(e) == 0); }();
  const dim3 wide(256, b->N);
  if (linear) {
    hipLaunchKernelGGL(lat_fb_prep, wide, dim3(256), 0, stream, p, mode == 0 ? 0 : 1);
    hipLaunchKernelGGL(lat_fb_alpha_beta_lin, two, dim3(kLinThreads), kFbCap * sizeof(double), stream, p, cap);
    if (mode != 0) hipLaunchKernelGGL(lat_fb_logs, wide, dim3(256), 0, stream, p);     // (MMI's posterior pass takes the logs of what it reads)
  } else {
    hipLaunchKernelGGL(lat_fb_alpha_beta, two, thr, kFbCap * sizeof(double), stream, p, cap);
  }
  if (mode == 0) {
    hipLaunchKernelGGL(lat_fb_posteriors<0>, many, thr, 0, stream, p, linear ? 1 : 0);
  } else {
    hipLaunchKernelGGL(lat_fb_accuracy, two, thr, 0, stream, p);
    hipLaunchKernelGGL(lat_fb_posteriors<1>, many, thr, 0, stream, p, 0);
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
