// Device helpers shared by the two lattice decoders (lattice_decode.hip: one workgroup per utterance, one launch;
// lattice_decode_frames.hip: a team of workgroups per utterance, a few launches per frame): cost encoding,
// workgroup reductions / scans, exact k-th smallest selection, arc walks with heavy-state hand-off, the
// per-utterance view of the workspace, link compaction, and the final-cost + lattice-beam pruning pass.
// Both translation units are compiled with -ffp-contract=off (pykaldi2_amd/build.py FILE_FLAGS).
#pragma once
#include <cmath>

#include "lattice_internal.h"

namespace pk2 {

constexpr int kLatThreads = 1024;
constexpr int kLatWaves = kLatThreads / 64;
constexpr int kMaxPdfsLds = 8192;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
// The thread's index.  The team decoder's persistent kernel (PK2_LAT_OPAQUE_TID) takes it through an opaque asm at every use:
// the compiler otherwise hoists every f(thread index) of every phase out of the frame loop -- shifts, masks, per-thread
// addresses -- keeps them alive across all phases and spills them (1024 threads: 128 registers each); a reload from scratch
// behind an L1 invalidation is a round trip to L2.
__device__ __forceinline__ int lat_tid() {
  int t = (int)threadIdx.x;
#ifdef PK2_LAT_OPAQUE_TID
  asm volatile("" : "+v"(t));
#endif
  return t;
}

constexpr int kMaxEpsRounds = 256;
constexpr int kHeavyDegree = 48;   // states with more arcs than this (word-loop / silence states: 10^4 and more)
constexpr int kMaxHeavy = 2048;    // are deferred and expanded by all 1024 threads, an arc per thread

__device__ __forceinline__ uint32_t enc_cost(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_cost(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
template <typename T>
__device__ __forceinline__ T ld_coherent(const T* p) {   // bypasses the vector L1 (values written by atomics)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct DecodeParams {
  DevDecodeGraph g;
  LatPtrs L;
  const float* loglikes; int64_t seq_stride, frame_stride; int32_t P;
  const int32_t* tid2pdf; int32_t num_tids;
  float beam, lattice_beam, beam_delta, ac_scale;
  int32_t max_active, min_active;
};

#ifdef PK2_LAT_PROFILE
#define LAT_T(k) do { __syncthreads(); if (lat_tid() == 0) { const long long now_ = wall_clock64(); sh.prof[k] += now_ - sh.prof_last; sh.prof_last = now_; } } while (0)
#else
#define LAT_T(k) do { } while (0)
#endif

struct Shared {
#ifdef PK2_LAT_PROFILE
  long long prof[16]; long long prof_last;
#endif
  float ll[kMaxPdfsLds];
  uint32_t hist[2048];
  float redf[kLatWaves];
  int redi[kLatWaves];
  int n_new;      // tokens appended to the frame being built
  int n_link;     // links appended to the segment being built
  int status;
  uint32_t sel_prefix; int sel_k;
  int n_elist;                       // tokens of the frame being built that have epsilon arcs
  int n_heavy;                       // tokens whose arcs the whole workgroup walks together
  int heavy_tok[kMaxHeavy];
  float heavy_cost[kMaxHeavy];
};

__device__ __forceinline__ float block_min(float v, Shared& sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((lat_tid() & 63) == 0) sh.redf[lat_tid() >> 6] = v;
  __syncthreads();
  float r = sh.redf[0];
#pragma unroll
  for (int k = 1; k < kLatWaves; ++k) r = fminf(r, sh.redf[k]);
  return r;
}
__device__ __forceinline__ int block_sum_i(int v, Shared& sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((lat_tid() & 63) == 0) sh.redi[lat_tid() >> 6] = v;
  __syncthreads();
  int r = 0;
#pragma unroll
  for (int k = 0; k < kLatWaves; ++k) r += sh.redi[k];
  return r;
}

// k-th smallest (0-based) of cost[0..n): radix select on the order-preserving keys, 11 + 11 + 10 bits.
__device__ float kth_smallest(const float* cost, int n, int k, Shared& sh) {
  const int tid = lat_tid();
  uint32_t prefix = 0, mask = 0;
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = tid; i < 2048; i += kLatThreads) sh.hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kLatThreads) {
      const uint32_t key = enc_cost(cost[i]);
      if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shifts[pass]) & ((1u << bits[pass]) - 1)], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0: lane q owns bins [q*nb/64, (q+1)*nb/64); wave prefix sum, then the owning lane scans its bins
      const int nb = 1 << bits[pass], per = nb / 64;
      int mine = 0;
      for (int b = 0; b < per; ++b) mine += (int)sh.hist[tid * per + b];
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (tid >= o) incl += y;
      }
      const int before = incl - mine;
      if (k >= before && k < incl) {
        int kk = k - before, b = 0;
        for (; b < per; ++b) {
          const int c = (int)sh.hist[tid * per + b];
          if (kk < c) break;
          kk -= c;
        }
        sh.sel_prefix = prefix | ((uint32_t)(tid * per + b) << shifts[pass]);
        sh.sel_k = kk;
      }
    }
    __syncthreads();
    prefix = sh.sel_prefix;
    k = sh.sel_k;
    mask |= ((1u << bits[pass]) - 1) << shifts[pass];
    __syncthreads();
  }
  return dec_cost(prefix);
}

// k-th smallest (0-based) of cost[0..n) when it is known to lie in [lo, hi): one pass over 2047 linear bins of
// [lo, hi) (the token costs of a frame share their exponent, so the radix select's first pass would pile every
// key into a few LDS counters), then the exact radix select among the members of the selected bin.
__device__ float kth_smallest_in_range(const float* cost, int n, int k, float lo, float hi, Shared& sh) {
  const int tid = lat_tid();
  const float scale = 2047.0f / (hi - lo);
  auto bin_of = [&](float c) { return c >= hi ? 2047 : min(2046, (int)((c - lo) * scale)); };
  for (int i = tid; i < 2048; i += kLatThreads) sh.hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += kLatThreads) atomicAdd(&sh.hist[bin_of(cost[i])], 1u);
  __syncthreads();
  if (tid < 64) {
    int mine = 0;
    for (int b = 0; b < 32; ++b) mine += (int)sh.hist[tid * 32 + b];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o, 64);
      if (tid >= o) incl += y;
    }
    const int before = incl - mine;
    if (k >= before && k < incl) {
      int kk = k - before, b = 0;
      for (; b < 32; ++b) {
        const int c = (int)sh.hist[tid * 32 + b];
        if (kk < c) break;
        kk -= c;
      }
      sh.sel_prefix = (uint32_t)(tid * 32 + b);
      sh.sel_k = kk;
    }
  }
  __syncthreads();
  const int sel_bin = (int)sh.sel_prefix;
  k = sh.sel_k;
  __syncthreads();
  // exact selection among the members of sel_bin
  uint32_t prefix = 0, mask = 0;
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = tid; i < 2048; i += kLatThreads) sh.hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kLatThreads) {
      const float c = cost[i];
      if (bin_of(c) != sel_bin) continue;
      const uint32_t key = enc_cost(c);
      if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shifts[pass]) & ((1u << bits[pass]) - 1)], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const int nb = 1 << bits[pass], per = nb / 64;
      int mine = 0;
      for (int b = 0; b < per; ++b) mine += (int)sh.hist[tid * per + b];
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (tid >= o) incl += y;
      }
      const int before = incl - mine;
      if (k >= before && k < incl) {
        int kk = k - before, b = 0;
        for (; b < per; ++b) {
          const int c = (int)sh.hist[tid * per + b];
          if (kk < c) break;
          kk -= c;
        }
        sh.sel_prefix = prefix | ((uint32_t)(tid * per + b) << shifts[pass]);
        sh.sel_k = kk;
      }
    }
    __syncthreads();
    prefix = sh.sel_prefix;
    k = sh.sel_k;
    mask |= ((1u << bits[pass]) - 1) << shifts[pass];
    __syncthreads();
  }
  return dec_cost(prefix);
}

// Applies body(i, cost, arc) to every arc (CSR `off`) of the tokens i = list[j], j in [0, n_list), that `active`
// accepts.  Light states are walked by the entry's thread; heavy ones are queued in LDS and then walked by the
// whole workgroup, an arc per thread.  Contains workgroup barriers: call from uniform control flow.
// A team of workgroups shares one list by passing first = workgroup * kLatThreads + thread, stride = team threads.
template <typename Active, typename Body>
__device__ __forceinline__ void for_each_arc(Shared& sh, const int32_t* list, int n_list, const int32_t* ts,
                                             const int32_t* off, Active active, Body body, int first = -1,
                                             int stride = kLatThreads) {
  const int tid = lat_tid();
  for (int j = first < 0 ? tid : first; j < n_list; j += stride) {
    const int i = list[j];
    float c;
    if (!active(i, &c)) continue;
    const int s = ts[i];
    const int a0 = off[s], a1 = off[s + 1];
    if (a1 - a0 > kHeavyDegree) {
      const int h = atomicAdd(&sh.n_heavy, 1);
      if (h < kMaxHeavy) { sh.heavy_tok[h] = i; sh.heavy_cost[h] = c; continue; }
    }
    for (int a = a0; a < a1; ++a) body(i, c, a);
  }
  __syncthreads();
  const int nh = min(sh.n_heavy, kMaxHeavy);
  for (int h = 0; h < nh; ++h) {
    const int i = sh.heavy_tok[h];
    const float c = sh.heavy_cost[h];
    const int s = ts[i];
    for (int a = off[s] + tid; a < off[s + 1]; a += kLatThreads) body(i, c, a);
  }
  __syncthreads();
  if (tid == 0) sh.n_heavy = 0;
  __syncthreads();
}

// Per-utterance views of the workspace.  During decoding the float64 arrays of the forward-backward are
// scratch: {first emitting arc, emitting degree} per token, the frame's arc work list, the list of the frame's
// tokens that have epsilon arcs.
struct UttView {
  uint32_t* stc; int32_t* stt;
  int32_t* ts; float* tc; float* te; float* tf; int32_t* tl;
  int2* tarc;                                   // [tok_cap] {e_off[state], emitting degree}        (alpha region)
  int2* work;                                   // [tok_cap] {token, emitting arc}                   (beta region)
  float* work_tot; int32_t* elist;              // [tok_cap] each                                    (acc_f / acc_b)
  int32_t* ftok; int32_t* seg; int32_t* kept; int32_t* maxlev;
  int4* lrec; float* lac;                       // links: {src token, dst token, transition-id, graph cost bits}, acoustic cost
  const int4* erec;                             // packed emitting arcs {dst state, transition-id, weight bits, pdf}
  int tok_cap, link_cap;
};

__device__ __forceinline__ UttView make_view(const DecodeParams& p, int n, const LatUtt& U) {
  UttView V;
  V.stc = p.L.st_cost + (size_t)n * p.g.S; V.stt = p.L.st_tok + (size_t)n * p.g.S;
  V.ts = p.L.tok_state + U.tok_base; V.tc = p.L.tok_cost + U.tok_base; V.te = p.L.tok_extra + U.tok_base;
  V.tf = p.L.tok_final + U.tok_base; V.tl = p.L.tok_level + U.tok_base;
  V.tarc = reinterpret_cast<int2*>(p.L.alpha + U.tok_base);
  V.work = reinterpret_cast<int2*>(p.L.beta + U.tok_base);
  V.work_tot = reinterpret_cast<float*>(p.L.acc_f + U.tok_base);
  V.elist = reinterpret_cast<int32_t*>(p.L.acc_b + U.tok_base);
  V.ftok = p.L.frame_tok + U.frame_base; V.seg = p.L.seg_off + U.frame_base;
  V.kept = p.L.seg_kept + U.frame_base; V.maxlev = p.L.frame_maxlev + U.frame_base;
  V.lrec = p.L.link_rec + U.link_base; V.lac = p.L.link_ac + U.link_base;
  V.erec = p.L.e_rec;
  V.tok_cap = U.tok_cap; V.link_cap = U.link_cap;
  return V;
}

// A token for state d was created at utterance-local index f0 + idx: record it; tokens with epsilon arcs join
// the frame's epsilon list.
__device__ __forceinline__ void register_token(const DecodeParams& p, const UttView& V, Shared& sh, int f0, int idx, int d) {
  if (f0 + idx < V.tok_cap) {
    V.ts[f0 + idx] = d;
    V.tc[f0 + idx] = INFINITY;     // "cost at the last epsilon expansion"; the final cost is written when the frame closes
    __hip_atomic_store(&V.stt[d], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.g.n_off[d + 1] > p.g.n_off[d]) {
      const int e = atomicAdd(&sh.n_elist, 1);
      if (e < V.tok_cap) V.elist[e] = f0 + idx; else sh.status = kLatTokenOverflow;
    }
  } else {
    sh.status = kLatTokenOverflow;
  }
}

// Epsilon closure of the frame whose tokens start at f0, the frame's epsilon links, final token costs, sparse
// reset of the state table.  On entry sh.n_new tokens exist with their costs in the table.
__device__ void close_frame(const DecodeParams& p, const UttView& V, Shared& sh, int f0, float cutoff, int* link_end,
                            int* tok_end, int seg_index) {
  const int tid = lat_tid();
  const uint32_t kcut = enc_cost(cutoff);
  int rounds = 0;
  while (true) {
    __syncthreads();
    const int ne = min(sh.n_elist, V.tok_cap);
    int changed = 0;
    for_each_arc(sh, V.elist, ne, V.ts, p.g.n_off,
                 [&](int i, float* c) {
                   const float cc = dec_cost(ld_coherent(&V.stc[V.ts[i]]));
                   if (!(cc < V.tc[i])) return false;      // not improved since its last expansion
                   V.tc[i] = cc;
                   *c = cc;
                   return cc < cutoff;
                 },
                 [&](int i, float c, int a) {
                   const float tot = c + p.g.n_w[a];
                   const uint32_t k = enc_cost(tot);
                   if (k < kcut) {
                     const int d = p.g.n_dst[a];
                     const uint32_t old = atomicMin(&V.stc[d], k);
                     if (k < old) {
                       changed = 1;
                       if (old == kEmpty) register_token(p, V, sh, f0, atomicAdd(&sh.n_new, 1), d);
                     }
                   }
                 });
    changed = __syncthreads_or(changed);
    if (!changed || sh.status != kLatOk) break;
    if (++rounds > kMaxEpsRounds) { if (tid == 0) sh.status = kLatEpsilonLoop; break; }
  }
  __syncthreads();
  if (sh.status != kLatOk) return;
  LAT_T(4);
  const int cnt = sh.n_new, ne = sh.n_elist;
  const int l0 = *link_end;
  // epsilon links from the final costs (every listed token was last expanded at its final cost)
  for_each_arc(sh, V.elist, ne, V.ts, p.g.n_off,
               [&](int i, float* c) { *c = V.tc[i]; return V.tc[i] < cutoff; },
               [&](int i, float c, int a) {
                 const float tot = c + p.g.n_w[a];
                 if (tot < cutoff) {
                   const int li = l0 + atomicAdd(&sh.n_link, 1);
                   if (li < V.link_cap) {
                     V.lrec[li] = make_int4(i, f0 + ld_coherent(&V.stt[p.g.n_dst[a]]), 0, __float_as_int(p.g.n_w[a]));
                     V.lac[li] = 0.f;
                   } else {
                     sh.status = kLatLinkOverflow;
                   }
                 }
               });
  LAT_T(5);
  // final costs, per-token arc ranges for the next frame's work list, sparse reset of the table
  for (int i0 = f0 + tid; i0 < f0 + cnt; i0 += 4 * kLatThreads) {
    int st[4]; uint32_t ck[4]; int a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i0 + q * kLatThreads < f0 + cnt) st[q] = V.ts[i0 + q * kLatThreads];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * kLatThreads < f0 + cnt) { ck[q] = ld_coherent(&V.stc[st[q]]); a0[q] = p.g.e_off[st[q]]; a1[q] = p.g.e_off[st[q] + 1]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * kLatThreads;
      if (i < f0 + cnt) {
        V.tc[i] = dec_cost(ck[q]);
        V.tarc[i] = make_int2(a0[q], a1[q] - a0[q]);
        V.te[i] = INFINITY;
        V.stc[st[q]] = kEmpty;
        V.stt[st[q]] = -1;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    *tok_end = f0 + cnt;
    *link_end = min(l0 + sh.n_link, V.link_cap);
    V.seg[seg_index + 1] = *link_end;
    sh.n_new = 0;
    sh.n_link = 0;
    sh.n_elist = 0;
  }
  __syncthreads();
  LAT_T(6);
}

// Exclusive prefix over the workgroup of one int per thread; *total receives the sum.
__device__ __forceinline__ int block_exclusive_scan(int v, Shared& sh, int* total) {
  const int lane = lat_tid() & 63, w = lat_tid() >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o, 64);
    if (lane >= o) incl += y;
  }
  __syncthreads();
  if (lane == 63) sh.redi[w] = incl;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < kLatWaves; ++q) {
    const int c = sh.redi[q];
    if (q < w) before += c;
    tot += c;
  }
  *total = tot;
  return before + incl - v;
}

// In-place compaction of the links [l0, l1), four consecutive links per thread and pass.  keep(s, d, g, a)
// decides and may update the source token's extra cost.  Returns the number kept (all threads).
template <typename Keep>
__device__ int compact_links(const UttView& V, Shared& sh, int l0, int l1, float ac_mul, Keep keep) {
  constexpr int KPT = 4;
  const int tid = lat_tid();
  int out = l0;
  for (int base = l0; base < l1; base += KPT * kLatThreads) {
    int4 r[KPT]; float a[KPT]; bool k[KPT];
    const int lb = base + tid * KPT;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const int l = lb + q;
      k[q] = false;
      if (l < l1) { r[q] = V.lrec[l]; a[q] = V.lac[l]; }
    }
    int mine = 0;
#pragma unroll
    for (int q = 0; q < KPT; ++q)
      if (lb + q < l1) { k[q] = keep(r[q].x, r[q].y, __int_as_float(r[q].w), a[q]); mine += k[q]; }
    int total;
    int o = out + block_exclusive_scan(mine, sh, &total);   // barriers inside: every link of the pass is in registers
#pragma unroll
    for (int q = 0; q < KPT; ++q)
      if (k[q]) { V.lrec[o] = r[q]; V.lac[o] = __fmul_rn(a[q], ac_mul); ++o; }
    out += total;
    __syncthreads();
  }
  return out - l0;
}

// Final costs (ComputeFinalCosts) and the token "extra costs" of lattice-beam pruning, last frame first
// (PruneForwardLinksFinal / PruneForwardLinks), of a decoded utterance with tok_end tokens and link_end links; one
// workgroup.  This is the only serial part of the pruning: whether a link survives is a function of the final extra
// costs alone, so the links are dropped, and the epsilon-DAG depths computed, afterwards by prune_segments, segment
// by segment in parallel.
__device__ __forceinline__ float block_min(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((lat_tid() & 63) == 0) red[lat_tid() >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int k = 1; k < kLatWaves; ++k) r = fminf(r, red[k]);
  return r;
}

// ---- final costs (ComputeFinalCosts): extra costs of the last frame's tokens; `red` = kLatWaves floats of LDS ----
__device__ void final_costs(const DecodeParams& p, const UttView& V, float* red, int T, int s_tok_end, int* any_final,
                            float* best) {
  const int tid = lat_tid();
  int32_t* ts = V.ts; float* tc = V.tc; float* te = V.te; float* tf = V.tf;
  const int fT0 = V.ftok[T], fT1 = s_tok_end;
  int anyf = 0;
  for (int i = fT0 + tid; i < fT1; i += kLatThreads) anyf |= (p.g.final_cost[ts[i]] < INFINITY);
  anyf = __syncthreads_or(anyf);
  float bmin = INFINITY;
  for (int i = fT0 + tid; i < fT1; i += kLatThreads) {
    const float fc = anyf ? p.g.final_cost[ts[i]] : 0.f;
    tf[i] = fc;
    if (fc < INFINITY) bmin = fminf(bmin, tc[i] + fc);
  }
  const float best_final = block_min(bmin, red);
  for (int i = fT0 + tid; i < fT1; i += kLatThreads) {
    const float fc = tf[i];
    te[i] = fc < INFINITY ? (tc[i] + fc) - best_final : INFINITY;
  }
  for (int i = tid; i < s_tok_end; i += kLatThreads) V.tl[i] = 0;
  __syncthreads();
  *any_final = anyf; *best = best_final;
}

// ---- lattice-beam pruning, last frame first (PruneForwardLinksFinal / PruneForwardLinks), everything in global memory ----
__device__ void extra_costs_global(const DecodeParams& p, const UttView& V, int T) {
  const int tid = lat_tid();
  float* tc = V.tc;
  int32_t* seg = V.seg;
  int4* lrec = V.lrec;
  uint32_t* teu = reinterpret_cast<uint32_t*>(V.te);   // extra costs are >= 0: their bit patterns order like the floats
  const float lbeam = p.lattice_beam;
  for (int t = T; t >= 0; --t) {
    // epsilon links inside frame t, to the fixed point
    const int e0 = seg[2 * t], e1 = seg[2 * t + 1];
    for (int rounds = 0; rounds < kMaxEpsRounds && e1 > e0; ++rounds) {
      int changed = 0;
      for (int l = e0 + tid; l < e1; l += kLatThreads) {
        const int4 r = lrec[l];
        const int s = r.x, d = r.y;
        const float ed = __uint_as_float(ld_coherent(&teu[d]));
        if (ed < INFINITY) {
          float le = ed + ((tc[s] + __int_as_float(r.w)) - tc[d]);
          if (le <= lbeam) {
            le = fmaxf(le, 0.f);
            const uint32_t k = __float_as_uint(le);
            if (k < atomicMin(&teu[s], k)) changed = 1;
          }
        }
      }
      if (!__syncthreads_or(changed)) break;
    }
    // emitting links t-1 -> t: the extra costs of frame t are final; they give the extra costs of their sources
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = seg[2 * t];
      for (int l0 = m0 + tid; l0 < m1; l0 += 4 * kLatThreads) {
        int4 r[4]; float a[4], ed[4], cs[4], cd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (l0 + q * kLatThreads < m1) { r[q] = lrec[l0 + q * kLatThreads]; a[q] = V.lac[l0 + q * kLatThreads]; }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (l0 + q * kLatThreads < m1) { ed[q] = __uint_as_float(ld_coherent(&teu[r[q].y])); cs[q] = tc[r[q].x]; cd[q] = tc[r[q].y]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (l0 + q * kLatThreads >= m1 || !(ed[q] < INFINITY)) continue;
          const float le = ed[q] + (__fadd_rn(__fadd_rn(cs[q], a[q]), __int_as_float(r[q].w)) - cd[q]);
          if (le <= lbeam) atomicMin(&teu[r[q].x], __float_as_uint(fmaxf(le, 0.f)));
        }
      }
    }
    __syncthreads();
  }
}

__device__ void finish_and_prune(const DecodeParams& p, const UttView& V, Shared& sh, int n, int T, int s_tok_end,
                                 int s_link_end) {
  int anyf; float best_final;
  final_costs(p, V, sh.redf, T, s_tok_end, &anyf, &best_final);
  extra_costs_global(p, V, T);
  if (lat_tid() == 0) {
    LatUtt* o = p.L.utt + n;
    o->status = kLatOk; o->n_tok = s_tok_end; o->n_link = s_link_end; o->any_final = anyf; o->best_cost = best_final;
  }
}

// Drops the links that lattice pruning rejects (in-place compaction of every segment; the acoustic scale is removed
// from the kept emitting links) and computes the depth of the epsilon DAG inside each frame, for the frames
// first, first + stride, ...: independent per frame once finish_and_prune has fixed the extra costs.
__device__ void prune_segments(const DecodeParams& p, const UttView& V, Shared& sh, int T, int first, int stride) {
  const int tid = lat_tid();
  const float* tc = V.tc;
  const uint32_t* teu = reinterpret_cast<const uint32_t*>(V.te);
  const int32_t* seg = V.seg;
  int4* lrec = V.lrec;
  int32_t* tl = V.tl;
  const float lbeam = p.lattice_beam;
  const float inv_scale = 1.0f / p.ac_scale;
  for (int t = first; t <= T; t += stride) {
    const int e0 = seg[2 * t], e1 = seg[2 * t + 1];
    int ke = 0;
    if (e1 > e0)
      ke = compact_links(V, sh, e0, e1, 1.0f, [&](int s, int d, float g, float a) {
        const float ed = __uint_as_float(teu[d]);
        return ed < INFINITY && (ed + ((tc[s] + g) - tc[d])) <= lbeam;
      });
    if (tid == 0) V.kept[2 * t] = ke;
    // epsilon DAG depth of the frame's tokens (order of the forward-backward inside the frame)
    int lev_max = 0;
    if (ke > 0) {
      for (int rounds = 0; rounds < kMaxEpsRounds; ++rounds) {
        int changed = 0;
        for (int l = e0 + tid; l < e0 + ke; l += kLatThreads) {
          const int4 r = lrec[l];
          const int lv = ld_coherent(&tl[r.x]) + 1;
          if (lv > atomicMax(&tl[r.y], lv)) changed = 1;
        }
        if (!__syncthreads_or(changed)) break;
      }
      float lm = 0.f;
      for (int l = e0 + tid; l < e0 + ke; l += kLatThreads) lm = fmaxf(lm, (float)ld_coherent(&tl[lrec[l].y]));
      lev_max = (int)(-block_min(-lm, sh));
    }
    if (tid == 0) V.maxlev[t] = lev_max;
    if (t > 0) {
      const int m0 = seg[2 * t - 1], m1 = seg[2 * t];
      const int km = compact_links(V, sh, m0, m1, inv_scale, [&](int s, int d, float g, float a) {
        const float ed = __uint_as_float(teu[d]);
        return ed < INFINITY && (ed + (__fadd_rn(__fadd_rn(tc[s], a), g) - tc[d])) <= lbeam;
      });
      if (tid == 0) V.kept[2 * t - 1] = km;
    }
    __syncthreads();
  }
}

// lattice_decode_frames.hip: `team` workgroups per utterance, a few launches per frame (replayed from hipGraphs).
int lattice_decode_frames(const DecodeParams& p, int N, int Tmax, int team, hipStream_t stream);
int lattice_persist_status(unsigned* abort_flag);

}  // namespace pk2
