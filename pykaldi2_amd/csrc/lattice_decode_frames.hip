// On-the-fly lattice generation by a TEAM of workgroups per utterance: the phases of a frame as device functions, run either
// inside ONE persistent launch per minibatch (lat_frames_persist, further down: the default on an MI355X) or as a few
// graph-replayed launches per frame (the lat_frames_* kernels right below: the fallback, PK2_LAT_DECODER=frames).
//
// lattice_decode.hip gives every utterance one workgroup for the whole decode: with the 8 utterances per GPU of
// the lattice-MMI configuration 248 of the 256 CUs idle while 8 run a serial chain of ~125 us frames.  Here a
// frame is cut into a few kernels with no barrier wider than a workgroup inside; between them the kernel boundary
// is the (cheapest) grid barrier, and the launches of 64 frames are replayed from one hipGraph:
//
//   cutoff     (1 workgroup / utterance)  GetCutoff: best cost (reduced by the previous frame's finalise), beam,
//                                         exact k-th smallest cost when max_active / min_active bind
//   list       (G workgroups / utterance) sparse reset of the state table; arc work list of the surviving tokens -- every workgroup reserves a
//                                         region with one atomicAdd, the order of a bag of arcs is irrelevant --
//                                         arc costs, atomicMin of the best new cost
//   expand                                atomicMin into the per-state table, new tokens, emitting links
//   fix+round0                            link destinations (state -> token), first epsilon relaxation round
//   round 1..R                            further rounds; each returns at once when the previous one changed nothing
//   tail       (1 workgroup / utterance)  finishes deeper epsilon chains to the exact fixed point (normally a no-op)
//   close                                 epsilon links from the final costs, final token costs, arc ranges, best
//                                         cost of the frame (the next frame's list launch clears the state table)
//
// The epsilon list is shared by index ownership (entry e belongs to team thread e mod team size) and a launch only
// walks the entries that existed at the launch boundary, so workgroups never read another workgroup's plain
// stores of the same launch; everything that crosses workgroups inside a launch is an agent-scope atomic (state
// table, counters).  Token costs and the kept link SET are those of the one-workgroup decoder and of the oracle
// (minima and sets do not depend on the order of the atomics); token and link numbering differ.
// Final costs and lattice pruning: one workgroup per utterance (lat_frames_finish below, extra costs in LDS).
#include <map>

#ifndef PK2_LAT_PLAIN_TID
#define PK2_LAT_OPAQUE_TID 1
#endif
#include "lattice_decode_common.h"
#include "persist_guard.h"
#include "step_graph.h"

namespace pk2 {

template <typename T>
__device__ __forceinline__ void st_coherent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Workgroup barrier for data exchanged through LDS only: __syncthreads() is a full fence that begins with s_waitcnt vmcnt(0),
// i.e. it waits for every global load, store and atomic the wave has in flight -- a round trip to L2 per barrier in a phase
// whose next loads do not depend on them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// The hot fields of the frame record as every workgroup of a PERSISTENT team holds them in its LDS (round 6): they are
// handed over inside the release words of the team barriers (lat_team_barrier_hand) or follow from values every workgroup
// already has, so a phase starts with its first data loads instead of a dependent round trip to the record in L2, and the
// barrier's last arriver releases the team without waiting for bookkeeping stores.  The launch-per-frame kernels (hot ==
// nullptr) read and write the record as before.
struct Hot {
  int f0, f1, l0;              // tokens [f0, f1) of the frame being expanded; links of all closed segments at the frame's start
  uint32_t best_key;
  float cur_cutoff, adaptive, build_cutoff;
  int n_arcs; uint32_t nmin_key;
  int n_link;                  // emitting links of the frame (raw counter behind expand)
  int ne, nh, n_new;           // raw counters behind the last relaxation round: epsilon list, heavy list, new tokens
  int more, status_bad;        // the last round lowered a cost; the utterance's status is no longer ok
  const float* ll_base; int64_t ll_stride;
};

struct TeamCtx {
  UttView V;
  LatFrame* F;
  int n, wg, G, t, T;
  bool live;
  const Hot* hot = nullptr;
};

// Common prologue: frame index t = base + local - 1 (step 0 is InitDecoding's closure, "frame -1").
__device__ __forceinline__ TeamCtx team_ctx(const DecodeParams& p, const StepCounter* cnt, int local) {
  TeamCtx c;
  c.n = blockIdx.y; c.wg = blockIdx.x; c.G = gridDim.x;
  c.t = cnt->base + local - 1;
  const LatUtt U = p.L.utt[c.n];
  c.T = U.T;
  c.V = make_view(p, c.n, U);
  c.F = p.L.frame + c.n;
  c.live = cnt->base + local < cnt->T && c.t < c.T && c.F->status == kLatOk;
  return c;
}

// The same for a team inside the persistent kernel: utterance n, rank wg of G, frame t (status is looked at by the caller).
__device__ __forceinline__ TeamCtx team_ctx_of(const DecodeParams& p, int n, int wg, int G, int t) {
  TeamCtx c;
  c.n = n; c.wg = wg; c.G = G; c.t = t;
  const LatUtt U = p.L.utt[n];
  c.T = U.T;
  c.V = make_view(p, n, U);
  c.F = p.L.frame + n;
  c.live = true;
  return c;
}

// The last workgroup of the team to get here returns true (all of its threads); the caller's thread 0 then does
// the bookkeeping of the launch.  Counters it reads were bumped by returning atomics that completed before the
// workgroup barrier.
__device__ __forceinline__ bool team_last(LatFrame* F, int G, int* s_flag) {
  __syncthreads();
  if (lat_tid() == 0) {
    const int old = atomicAdd(&F->arrive, 1);
    *s_flag = old == G - 1;
    if (old == G - 1) st_coherent(&F->arrive, 0);
  }
  __syncthreads();
  return *s_flag != 0;
}

// Slot allocation from a team-wide counter with ONE atomic per wave: the lanes that are active here and `want` a slot
// are counted by ballot, the first of them reserves the run.  (A per-lane atomicAdd on the one counter of an
// utterance serialises: 25k of them per frame cost 300 us.)  Call from any control flow; lanes of a wave always
// belong to one utterance.
__device__ __forceinline__ int wave_alloc(int32_t* counter, bool want) {
  const unsigned long long mask = __ballot(want);
  if (!want) return -1;
  const int lane = lat_tid() & 63, leader = __ffsll((long long)mask) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(mask));
  base = __shfl(base, leader, 64);
  return base + __popcll(mask & ((1ull << lane) - 1ull));
}

// Same for a run of `count` slots per lane; every lane of the wave must be active.  Returns the lane's first slot.
__device__ __forceinline__ int wave_alloc_n(int32_t* counter, int count) {
  const int lane = lat_tid() & 63;
  int incl = count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o, 64);
    if (lane >= o) incl += y;
  }
  const int total = __shfl(incl, 63, 64);
  int base = 0;
  if (lane == 63 && total > 0) base = atomicAdd(counter, total);
  return __shfl(base, 63, 64) + incl - count;
}

// Index ownership of the epsilon list: the team's waves take turns (wave w of workgroup g owns entries (w G + g) 64 .. + 63
// of every G * 1024) -- a frame has one or two thousand entries, which whole workgroups in a row would leave to two of them.
__device__ __forceinline__ int team_entry(int wg, int G) {
  return ((lat_tid() >> 6) * G + wg) * 64 + (lat_tid() & 63);
}

// A token whose state has more than kHeavyDegree epsilon arcs joins the team's heavy list (false: the list is full, the
// token stays on the ordinary epsilon list and its owner walks the arcs).
__device__ __forceinline__ bool team_register_heavy(LatFrame* F, int G, int tok, int state, int a0, int deg) {
  const int h = atomicAdd(&F->n_hlist, 1);
  if (h >= kLatTeamHeavy) return false;
  F->hlist[h] = tok;
  *reinterpret_cast<int4*>(&F->hrec[4 * h]) = make_int4(tok, state, a0, deg);
  for (int w = 0; w < G; ++w) F->hlast[h * kLatMaxTeam + w] = INFINITY;
  return true;
}

// Two such reservations with both atomics in flight together (one round trip to L2 instead of two).
__device__ __forceinline__ void wave_alloc_n2(int32_t* counter_a, int count_a, int32_t* counter_b, int count_b, int* first_a,
                                              int* first_b) {
  const int lane = lat_tid() & 63;
  int ia = count_a, ib = count_b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ya = __shfl_up(ia, o, 64), yb = __shfl_up(ib, o, 64);
    if (lane >= o) { ia += ya; ib += yb; }
  }
  const int ta = __shfl(ia, 63, 64), tb = __shfl(ib, 63, 64);
  int base_a = 0, base_b = 0;
  if (lane == 63) {
    if (ta > 0) base_a = atomicAdd(counter_a, ta);
    if (tb > 0) base_b = atomicAdd(counter_b, tb);
  }
  *first_a = __shfl(base_a, 63, 64) + ia - count_a;
  *first_b = __shfl(base_b, 63, 64) + ib - count_b;
}

// The same for a whole WORKGROUP: NC team-wide counters, one atomic per counter and workgroup.  (Round 6: the 224 waves of a
// team reserving token, link and epsilon-list slots with one atomic per wave and counter -- all on the cache line of the frame
// record -- queued behind each other in L2: 2.6 + 3.5 us of the expand phase's 7.)  scr: 64 ints of LDS; every thread of the
// workgroup calls (LDS-only barriers inside: requests in flight stay in flight).
template <int NC>
__device__ __forceinline__ void block_alloc(int* scr, int32_t* c0, int32_t* c1, const int (&count)[NC], int (&first)[NC]) {
  static_assert(NC == 1 || NC == 2, "one or two counters");
  const int lane = lat_tid() & 63, w = lat_tid() >> 6;
  int incl[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) incl[k] = count[k];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int y = __shfl_up(incl[k], o, 64);
      if (lane >= o) incl[k] += y;
    }
  }
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < NC; ++k) scr[k * kLatWaves + w] = incl[k];
  }
  lds_barrier();
  if (w == 0 && lane < NC) {
    int tot = 0;
    for (int q = 0; q < kLatWaves; ++q) tot += scr[lane * kLatWaves + q];
    scr[2 * kLatWaves + lane] = tot > 0 ? atomicAdd(lane == 0 ? c0 : c1, tot) : 0;
  }
  lds_barrier();
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    int before = scr[2 * kLatWaves + k];
    for (int q = 0; q < w; ++q) before += scr[k * kLatWaves + q];
    first[k] = before + incl[k] - count[k];
  }
  lds_barrier();
}

// The team's epsilon list holds RECORDS (round 6): {token, state} {first epsilon arc, epsilon degree} -- what a relaxation
// round needs to start on the state's cost and its first arcs at once, where a list of token indices put three dependent
// round trips in front of them (index -> state -> arc range).  16 bytes per entry in the lower half of the region whose upper
// half holds the work list's graph costs: room for tok_cap / 4 entries of ONE frame (more: the utterance reports a token
// overflow and is decoded again with larger pools).
__device__ __forceinline__ int elist_cap(const UttView& V) { return V.tok_cap >> 2; }
__device__ __forceinline__ void elist_put(const UttView& V, LatFrame* F, int e, int tok, int state, int a0, int deg) {
  if (e < elist_cap(V)) {
    int2* l2 = reinterpret_cast<int2*>(V.elist);
    l2[2 * e] = make_int2(tok, state);
    l2[2 * e + 1] = make_int2(a0, deg);
  } else {
    st_coherent(&F->status, (int32_t)kLatTokenOverflow);
  }
}

// Called by every lane that created the token of state d (old == kEmpty; its epsilon arcs are [a0, a0 + deg)); others pass
// create = false.
__device__ __forceinline__ void team_register_token(const DecodeParams& p, const UttView& V, LatFrame* F, int G, int fb, int d,
                                                    int a0, int deg, bool create = true) {
  // (both reservations are requested before either answer is used: one round trip; a state with very many arcs asks for the
  // team's heavy list first and for an epsilon-list slot only when that list is full)
  const bool light = create && deg > 0 && deg <= kHeavyDegree;
  const int idx = wave_alloc(&F->n_new, create);
  int e = wave_alloc(&F->n_elist, light);
  if (create) {
    if (fb + idx < V.tok_cap) {
      V.ts[fb + idx] = d;
      V.tc[fb + idx] = INFINITY;
      st_coherent(&V.stt[d], idx);
      if (light) {
        elist_put(V, F, e, fb + idx, d, a0, deg);
      } else if (deg > kHeavyDegree && !team_register_heavy(F, G, fb + idx, d, a0, deg)) {
        e = atomicAdd(&F->n_elist, 1);
        elist_put(V, F, e, fb + idx, d, a0, deg);
      }
    } else {
      st_coherent(&F->status, (int32_t)kLatTokenOverflow);
      if (light) elist_put(V, F, e, 0, d, a0, deg);      // (the reserved slot must hold valid indices; the utterance is decoded again)
    }
  }
}

// body(token, cost, arc, arc weight, arc destination) for the epsilon arcs of the list's records [first, first + stride, ..)
// that `active(token, state, &cost)` accepts.  The first two arcs of a record are requested TOGETHER with what `active`
// reads (every record has at least one arc); states with many arcs are queued in LDS and walked by the whole workgroup as in
// for_each_arc.  Contains workgroup barriers.
template <typename Active, typename Body>
__device__ __forceinline__ void for_each_eps_record(Shared& sh, const DecodeParams& p, const UttView& V, int n_list, Active active,
                                                    Body body, int first, int stride) {
  const int tid = lat_tid();
  const int2* l2 = reinterpret_cast<const int2*>(V.elist);
  for (int j = first; j < n_list; j += stride) {
    const int2 e0 = l2[2 * j], e1 = l2[2 * j + 1];
    const int a0 = e1.x, a1 = e1.x + e1.y, a0b = min(a0 + 1, a1 - 1);
    const float w0 = p.g.n_w[a0], w1 = p.g.n_w[a0b];
    const int d0 = p.g.n_dst[a0], d1 = p.g.n_dst[a0b];
    float c;
    if (!active(e0.x, e0.y, &c)) continue;
    if (e1.y > kHeavyDegree) {
      const int h = atomicAdd(&sh.n_heavy, 1);
      if (h < kMaxHeavy) { sh.heavy_tok[h] = e0.x; sh.heavy_cost[h] = c; continue; }
    }
    body(e0.x, c, a0, w0, d0);
    if (e1.y > 1) body(e0.x, c, a0 + 1, w1, d1);
    for (int a = a0 + 2; a < a1; ++a) body(e0.x, c, a, p.g.n_w[a], p.g.n_dst[a]);
  }
  lds_barrier();
  const int nh = min(sh.n_heavy, kMaxHeavy);
  for (int h = 0; h < nh; ++h) {
    const int i = sh.heavy_tok[h];
    const float c = sh.heavy_cost[h];
    const int s = V.ts[i];
    for (int a = p.g.n_off[s] + tid; a < p.g.n_off[s + 1]; a += kLatThreads) body(i, c, a, p.g.n_w[a], p.g.n_dst[a]);
  }
  lds_barrier();
  if (tid == 0) sh.n_heavy = 0;
  lds_barrier();
}

// The arcs of the team's heavy tokens [0, nh): workgroup wg of G walks arcs wg * 1024 + thread, + G * 1024, ... of each.
// track: a relaxation round -- the share is walked only if the token got cheaper since THIS workgroup last walked it
// (slots of the team of Gteam workgroups; the single workgroup of the tail pass, G = 1, walks everything when any share is
// behind and brings all slots up to date).  Contains workgroup barriers.
// Round 6: the tokens' records, costs and verdicts are fetched by nh lanes side by side (record -> cost: two dependent round
// trips for the whole list, where one thread took three per token with the workgroup waiting at a barrier), and handed to
// the workgroup through LDS.
template <typename Body>
__device__ __forceinline__ void team_heavy_arcs(const DecodeParams& p, const UttView& V, LatFrame* F, Shared& sh, int nh,
                                                int wg, int G, int Gteam, float cutoff, bool track, Body body) {
  if (nh <= 0) return;
  const int tid = lat_tid();
  // sh.heavy_tok[0..4 nh): {token, first arc, last arc + 1, walk it}, sh.heavy_cost[h]: the token's cost
  if (tid < nh) {
    const int h = tid;
    const int4 r = *reinterpret_cast<const int4*>(&F->hrec[4 * h]);
    const float cc = dec_cost(ld_coherent(&V.stc[r.y]));
    bool act = cc < cutoff;
    if (track && act) {
      float* last = F->hlast + h * kLatMaxTeam;
      if (G == Gteam) {
        act = cc < last[wg];
        if (act) last[wg] = cc;
      } else {
        float behind = last[0];
        for (int w = 1; w < Gteam; ++w) behind = fmaxf(behind, last[w]);
        act = cc < behind;
        if (act) for (int w = 0; w < Gteam; ++w) last[w] = cc;
      }
    }
    sh.heavy_cost[h] = cc;
    sh.heavy_tok[4 * h] = r.x; sh.heavy_tok[4 * h + 1] = r.z; sh.heavy_tok[4 * h + 2] = r.z + r.w; sh.heavy_tok[4 * h + 3] = act ? 1 : 0;
  }
  lds_barrier();
  for (int h = 0; h < nh; ++h) {
    if (!sh.heavy_tok[4 * h + 3]) continue;
    const int i = sh.heavy_tok[4 * h], a1 = sh.heavy_tok[4 * h + 2];
    const float cc = sh.heavy_cost[h];
    for (int a = sh.heavy_tok[4 * h + 1] + wg * kLatThreads + tid; a < a1; a += G * kLatThreads) body(i, cc, a, p.g.n_w[a], p.g.n_dst[a]);
  }
  lds_barrier();
}

// One relaxation round over the epsilon-list entries [0, ne) owned by this workgroup and its share of the heavy tokens
// [0, nh).  Returns (to all threads) whether this workgroup lowered a cost.
__device__ __forceinline__ int eps_round(const DecodeParams& p, const UttView& V, LatFrame* F, Shared& sh, int fb,
                                         float cutoff, int ne, int nh, int wg, int G, int Gteam) {
  const uint32_t kcut = enc_cost(cutoff);
  int changed = 0;
  auto relax = [&](int i, float c, int a, float w, int d) {
    const float tot = c + w;
    const uint32_t k = enc_cost(tot);
    if (k < kcut) {
      // (the destination's epsilon arc range is requested with the atomic, not behind its answer)
      const int o0 = p.g.n_off[d], o1 = p.g.n_off[d + 1];
      const uint32_t old = atomicMin(&V.stc[d], k);
      if (k < old) {
        // another round is needed only if the state that got cheaper has epsilon arcs of its own
        if (o1 > o0) changed = 1;
        if (old == kEmpty) team_register_token(p, V, F, Gteam, fb, d, o0, o1 - o0);
      }
    }
  };
  for_each_eps_record(sh, p, V, ne,
                      [&](int i, int s, float* c) {
                        const float cc = dec_cost(ld_coherent(&V.stc[s]));
                        if (!(cc < V.tc[i])) return false;      // not improved since its last expansion
                        V.tc[i] = cc;
                        *c = cc;
                        return cc < cutoff;
                      },
                      relax, team_entry(wg, G), G * kLatThreads);
  team_heavy_arcs(p, V, F, sh, nh, wg, G, Gteam, cutoff, true, relax);
  return __syncthreads_or(changed);
}

// ---- step 0 only: the start token ----
__device__ __forceinline__ void phase_init(const DecodeParams& p, int n, int G) {
  if (lat_tid() != 0) return;
  const LatUtt U = p.L.utt[n];
  const UttView V = make_view(p, n, U);
  LatFrame* F = p.L.frame + n;
  F->f0 = 0; F->f1 = 0; F->link_end = 0; F->n_new = 0; F->n_link = 0; F->n_elist = 0; F->n_arcs = 0; F->ne_snap = 0;
  F->n_hlist = 0; F->nh_snap = 0;
  F->best_key = kEmpty; F->best_next = kEmpty; F->nmin_key = kEmpty;
  F->cur_cutoff = INFINITY; F->adaptive = p.beam; F->build_cutoff = p.beam;   // InitDecoding: ProcessNonemitting(beam)
  F->status = kLatOk; F->arrive = 0;
  for (int r = 0; r <= kLatEpsRounds; ++r) F->changed[r] = 0;
  F->ll_base = p.loglikes + (int64_t)n * p.seq_stride; F->ll_stride = p.frame_stride;
  V.stc[p.g.start] = enc_cost(0.f);
  team_register_token(p, V, F, G, 0, p.g.start, p.g.n_off[p.g.start], p.g.n_off[p.g.start + 1] - p.g.n_off[p.g.start]);
  F->ne_snap = F->n_elist; F->nh_snap = min(F->n_hlist, kLatTeamHeavy);
  V.ftok[0] = 0; V.seg[0] = 0;
}
__global__ void lat_frames_init(const DecodeParams p, int G) { phase_init(p, blockIdx.x, G); }

// ---- GetCutoff ----
// Number of costs below hi = lo + beam and, when more than k of them are, the exact k-th smallest (0-based): one
// pass builds 2047 linear bins of [lo, hi) (bin 2047 = the rest), which gives the count and the bin of the k-th;
// a second pass collects that bin's members (a handful) in LDS, where each is ranked against the others.
// Returns false when at most k costs lie below hi.  Falls back to kth_smallest_in_range for a crowded bin.
#ifdef PK2_LATP_PROFILE
__device__ long long g_cut[8], g_exp[8];
#define CUT_T(k) do { if (lat_tid() == 0) { const long long n_ = wall_clock64(); g_cut[k] += n_ - cut_last; cut_last = n_; } } while (0)
#else
#define CUT_T(k) do { } while (0)
#endif
#if defined(PK2_LATP_PROFILE) && defined(PK2_LATP_EXPAND)
// (-DPK2_LATP_EXPAND: expand's stages on rank 1 of utterance 1, each behind a full wait: what a stage's requests take to come
// back; the waits lengthen the phase, so the phase table is taken without this)
#define EXP_T(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lat_tid() == 0 && c.wg == 1 && c.n == 1) { const long long n_ = wall_clock64(); g_exp[k] += n_ - exp_last; exp_last = n_; } } while (0)
#else
#define EXP_T(k) do { } while (0)
#endif
__device__ bool kth_below(const float* cost, int n, int k, float lo, float hi, Shared& sh, float* out) {
  const int tid = lat_tid(), lane = tid & 63, w = tid >> 6;
#ifdef PK2_LATP_PROFILE
  long long cut_last = wall_clock64();
#endif
  const float scale = 2047.0f / (hi - lo);
  auto bin_of = [&](float c) { return c >= hi ? 2047 : min(2046, (int)((c - lo) * scale)); };
  // The first kCutRegs * 1024 costs stay in registers for both passes: all their loads are in flight together (a
  // load -> LDS atomic loop waits for L2 once per iteration: 4.6 us for 8.5 k costs) and the second pass reads nothing.
  #ifndef PK2_LAT_LAUNDER
#define PK2_LAT_LAUNDER 1
#endif
#ifndef PK2_LAT_CUTREGS
#define PK2_LAT_CUTREGS 10
#endif
  constexpr int kCutRegs = PK2_LAT_CUTREGS;
  float cr[kCutRegs];
#pragma unroll
  for (int q = 0; q < kCutRegs; ++q) {
    const int i = tid + q * kLatThreads;
    cr[q] = i < n ? cost[i] : 0.f;
  }
  for (int i = tid; i < 2048; i += kLatThreads) sh.hist[i] = 0;
  if (tid == 0) { sh.sel_k = -1; sh.redi[0] = 0; }
  __syncthreads();
  CUT_T(0);
#pragma unroll
  for (int q = 0; q < kCutRegs; ++q)
    if (tid + q * kLatThreads < n) atomicAdd(&sh.hist[bin_of(cr[q])], 1u);
  for (int i = tid + kCutRegs * kLatThreads; i < n; i += kLatThreads) atomicAdd(&sh.hist[bin_of(cost[i])], 1u);
  __syncthreads();
  CUT_T(1);
  {
    // prefix sums of the 2048 bins by all threads: two bins per thread, a shuffle scan per wave, the waves' totals through LDS
    const int h0 = (int)sh.hist[2 * tid], h1 = (int)sh.hist[2 * tid + 1];
    const int mine = h0 + h1;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o, 64);
      if (lane >= o) incl += y;
    }
    if (lane == 63) sh.heavy_tok[w] = incl;            // (the heavy-token queue is idle in this phase)
    __syncthreads();
    int before = incl - mine;
    for (int v = 0; v < w; ++v) before += sh.heavy_tok[v];
    const int c_lt = n - (int)sh.hist[2047];
    if (c_lt > k && k >= before && k < before + mine) {
      int kk = k - before, b = 2 * tid;
      if (kk >= h0) { kk -= h0; ++b; }
      sh.sel_prefix = (uint32_t)b;
      sh.sel_k = kk;
    }
  }
  __syncthreads();
  CUT_T(2);
  if (sh.sel_k < 0) return false;
  const int sel_bin = (int)sh.sel_prefix, kk = sh.sel_k;
  __syncthreads();
  auto member = [&](float c) {
    if (bin_of(c) == sel_bin) {
      const int m = atomicAdd(&sh.redi[0], 1);
      if (m < 2048) sh.hist[m] = enc_cost(c);     // the bins are no longer needed
    }
  };
#pragma unroll
  for (int q = 0; q < kCutRegs; ++q)
    if (tid + q * kLatThreads < n) member(cr[q]);
  for (int i = tid + kCutRegs * kLatThreads; i < n; i += kLatThreads) member(cost[i]);
  __syncthreads();
  CUT_T(3);
  const int m = sh.redi[0];
  if (m > 2048) {             // many equal costs: the general selection
    __syncthreads();
    *out = kth_smallest_in_range(cost, n, k, lo, hi, sh);
    return true;
  }
  for (int i = tid; i < m; i += kLatThreads) {
    const uint32_t key = sh.hist[i];
    int lt = 0, le = 0;
    for (int j = 0; j < m; ++j) { const uint32_t o = sh.hist[j]; lt += o < key; le += o <= key; }
    if (lt <= kk && kk < le) sh.sel_prefix = key;
  }
  __syncthreads();
  *out = dec_cost(sh.sel_prefix);
  __syncthreads();
  CUT_T(4);
  return true;
}

// (cutoff_out: {cur_cutoff, adaptive, adaptive follows from cur_cutoff} for all threads of the calling workgroup)
__device__ __forceinline__ void phase_cutoff(const DecodeParams& p, const TeamCtx& c, Shared& sh, float* cutoff_out = nullptr) {
  if (!c.live || c.t < 0) return;
  const int tid = lat_tid();
  if (tid == 0) sh.n_heavy = 0;
  LatFrame* F = c.F;
  const float* tc = c.V.tc;
  const Hot* h = c.hot;
  const int f0 = h ? h->f0 : F->f0, f1 = h ? h->f1 : F->f1, nt = f1 - f0;
  const float best = dec_cost(h ? h->best_key : F->best_key);
  const float beam_cutoff = best + p.beam;
  float cur_cutoff = beam_cutoff, adaptive = p.beam;
  const bool chk_max = nt > p.max_active, chk_min = p.min_active > 0 && nt > p.min_active;
  bool bound = false, adapted = false;     // adapted: adaptive = (cur_cutoff - best) + beam_delta, else the beam
  if (chk_max) {          // c_lt > max_active  <=>  the max_active-th cost (0-based) exists below the beam cutoff
    float kth;
    bound = kth_below(tc + f0, nt, p.max_active, best, beam_cutoff, sh, &kth);
    if (bound) { cur_cutoff = kth; adaptive = (cur_cutoff - best) + p.beam_delta; adapted = true; }
  }
  if (!bound && chk_min) {
    int c_le = 0;
    for (int i = f0 + tid; i < f1; i += kLatThreads) c_le += tc[i] <= beam_cutoff;
    c_le = block_sum_i(c_le, sh);
    if (c_le <= p.min_active) {
      cur_cutoff = kth_smallest(tc + f0, nt, p.min_active, sh);
      adaptive = (cur_cutoff - best) + p.beam_delta;
      adapted = true;
    }
  }
  if (tid == 0) { F->cur_cutoff = cur_cutoff; F->adaptive = adaptive; F->n_arcs = 0; F->nmin_key = kEmpty; }
  if (cutoff_out) { cutoff_out[0] = cur_cutoff; cutoff_out[1] = adaptive; cutoff_out[2] = adapted ? 1.f : 0.f; }
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_cutoff(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ Shared sh;
  phase_cutoff(p, team_ctx(p, cnt, local), sh);
}

// ---- arc work list of the surviving tokens, arc costs, best new cost ----
// Round 6 (persistent decoder): what the list phase does that does NOT depend on the frame's cutoff -- the log-likelihood row
// into LDS, the sparse reset of the state table for the workgroup's slice of the old frame -- runs on ranks `first`.. while
// rank 0 selects the cutoff (7 us during which the other 15 workgroups used to wait at the barrier); rank 0 then takes no
// slice of the list phase.  `first` = 0, `pre` = false: the launch-per-frame kernels' behaviour.
__device__ __forceinline__ void phase_list_pre(const DecodeParams& p, const TeamCtx& c, Shared& sh, int first) {
  if (!c.live || c.t < 0 || c.wg < first) return;
  const int tid = lat_tid();
  const UttView& V = c.V;
  LatFrame* F = c.F;
  const Hot* h = c.hot;
  const float* row = (h ? h->ll_base : F->ll_base) + (int64_t)c.t * (h ? h->ll_stride : F->ll_stride);
  for (int i = tid; i < p.P; i += kLatThreads) sh.ll[i] = row[i];
  const int f0 = h ? h->f0 : F->f0, f1 = h ? h->f1 : F->f1, nr = c.G - first;
  const int per = (f1 - f0 + nr - 1) / nr;
  const int s0 = f0 + (c.wg - first) * per, s1 = min(s0 + per, f1);
  for (int i = s0 + tid; i < s1; i += kLatThreads) {
    const int st = V.ts[i];
    V.stc[st] = kEmpty;
    V.stt[st] = -1;
  }
}

__device__ __forceinline__ void phase_list(const DecodeParams& p, const TeamCtx& c, Shared& sh, int& s_base, int first = 0,
                                           bool pre = false) {
  if (!c.live || c.t < 0 || c.wg < first) return;
  const int tid = lat_tid();
  const UttView& V = c.V;
  LatFrame* F = c.F;
  const float* tc = V.tc;
  const Hot* h = c.hot;
  const int f0 = h ? h->f0 : F->f0, f1 = h ? h->f1 : F->f1;
  const float cur_cutoff = h ? h->cur_cutoff : F->cur_cutoff;
  // the frame's log-likelihood row staged in LDS (reading it per arc from L2 puts one more dependent round trip
  // into the cost pass: measured +6 us per launch)
  if (!pre) {
    const float* row = (h ? h->ll_base : F->ll_base) + (int64_t)c.t * (h ? h->ll_stride : F->ll_stride);
    for (int i = tid; i < p.P; i += kLatThreads) sh.ll[i] = row[i];
  }
  float2* wcost = reinterpret_cast<float2*>(V.work_tot);     // {total cost, acoustic cost} per listed arc
  // {destination state, transition-id, graph cost} of every listed arc, by work-list index: the expand phase reads them
  // together with the costs instead of going back to the arc records (one dependent round trip less in its chain).
  // Arrays that are idle while frames are decoded: the final costs and epsilon levels of the tokens (written by the passes
  // behind the last frame), the upper half of the epsilon list's region.
  int32_t* wdst = reinterpret_cast<int32_t*>(V.tf); int32_t* wtid = V.tl; int32_t* wgc = V.elist + V.tok_cap;
  float nmin = INFINITY;
  // the frame's tokens are cut into equal slices (a frame has a few thousand tokens: dealing them in chunks of
  // 4096 would leave most of the team without work and put several dependent passes of the cost loop on the rest)
  const int nr = c.G - first;
  const int per = (f1 - f0 + nr - 1) / nr;
  const int s0 = f0 + (c.wg - first) * per, s1 = min(s0 + per, f1);
  for (int base = s0; base < s1; base += 4 * kLatThreads) {
    int2 ar[4];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + tid * 4 + q;
      ar[q] = make_int2(0, 0);
      if (i < s1) {
        if (tc[i] <= cur_cutoff) ar[q] = V.tarc[i];
        // sparse reset of the state table: every token of the old frame passes here once, and nothing reads the
        // table between the launch that closed the frame and the next expand
        if (!pre) {
          const int st = V.ts[i];
          V.stc[st] = kEmpty;
          V.stt[st] = -1;
        }
      }
      mine += ar[q].y;
    }
    int total;
    const int rel = block_exclusive_scan(mine, sh, &total);
    if (tid == 0) s_base = total > 0 ? atomicAdd(&F->n_arcs, total) : 0;
    __syncthreads();
    const int b0 = s_base;
    int o = b0 + rel;
    if (b0 + total <= V.tok_cap) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + tid * 4 + q;
        for (int k = 0; k < ar[q].y; ++k) V.work[o + k] = make_int2(i, ar[q].x + k);
        o += ar[q].y;
      }
    } else {
      if (tid == 0) st_coherent(&F->status, (int32_t)kLatTokenOverflow);
      total = 0;
    }
    __syncthreads();
    for (int j0 = b0 + tid; j0 < b0 + total; j0 += 2 * kLatThreads) {     // two independent arcs in flight per thread
      const int j1 = j0 + kLatThreads;
      const bool two = j1 < b0 + total;
      const int2 wk0 = V.work[j0], wk1 = two ? V.work[j1] : wk0;
      const int4 er0 = V.erec[wk0.y], er1 = V.erec[wk1.y];
      const float c0 = tc[wk0.x], c1 = tc[wk1.x];
      const float ac0 = -__fmul_rn(p.ac_scale, sh.ll[er0.w]), ac1 = -__fmul_rn(p.ac_scale, sh.ll[er1.w]);
      const float tot0 = __fadd_rn(__fadd_rn(c0, ac0), __int_as_float(er0.z));
      const float tot1 = __fadd_rn(__fadd_rn(c1, ac1), __int_as_float(er1.z));
      wcost[j0] = make_float2(tot0, ac0);
      wdst[j0] = er0.x; wtid[j0] = er0.y; wgc[j0] = er0.z;
      nmin = fminf(nmin, tot0);
      if (two) {
        wcost[j1] = make_float2(tot1, ac1);
        wdst[j1] = er1.x; wtid[j1] = er1.y; wgc[j1] = er1.z;
        nmin = fminf(nmin, tot1);
      }
    }
    __syncthreads();
  }
  nmin = block_min(nmin, sh);
  if (tid == 0 && nmin < INFINITY) atomicMin(&F->nmin_key, enc_cost(nmin));
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_list(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ Shared sh;
  __shared__ int s_base;
  phase_list(p, team_ctx(p, cnt, local), sh, s_base);
}

// The bookkeeping behind a phase, done by ONE thread once every workgroup of the team has finished the phase: in the
// launch-per-frame kernels by the last workgroup to arrive (team_last: an atomic round trip at the end of the launch), in the
// persistent kernel INSIDE the team barrier by its last arriver (round 6: lat_team_barrier_last -- one round trip instead of
// two on every phase boundary).
__device__ __forceinline__ unsigned expand_last(const TeamCtx& c) {
  LatFrame* F = c.F;
  if (!c.live || c.t < 0) return 0u;
  const float nmin = dec_cost(F->nmin_key);
  if (!(nmin < INFINITY)) return 0u;
  F->ne_snap = min(ld_coherent(&F->n_elist), elist_cap(c.V));
  F->nh_snap = min(ld_coherent(&F->n_hlist), kLatTeamHeavy);
  F->build_cutoff = nmin + F->adaptive;
  return 0u;
}

// ---- tokens and emitting links of frame t+1 ----
__device__ __forceinline__ void phase_expand(const DecodeParams& p, const TeamCtx& c, int& s_flag, int* s_scr, bool defer = false) {
  if (!c.live || c.t < 0) return;
  const int tid = lat_tid();
  const UttView& V = c.V;
  LatFrame* F = c.F;
  const Hot* h = c.hot;
  const float nmin = dec_cost(h ? h->nmin_key : F->nmin_key);
  if (!(nmin < INFINITY)) {
    if (c.wg == 0 && tid == 0) st_coherent(&F->status, (int32_t)kLatNoSurvivor);     // seen by the next launch / at the frame's last barrier
    return;
  }
  const float next_cutoff = nmin + (h ? h->adaptive : F->adaptive);
  const int n_arcs = h ? h->n_arcs : F->n_arcs, l0 = h ? h->l0 : F->link_end, fb = h ? h->f1 : F->f1;
  const float2* wcost = reinterpret_cast<const float2*>(V.work_tot);
  const int32_t* wdst = reinterpret_cast<const int32_t*>(V.tf); const int32_t* wtid = V.tl; const int32_t* wgc = V.elist + V.tok_cap;
  const int stride = c.G * kLatThreads;
  // Waves stay whole (the loop bound is the wave's first lane) so that slots can be reserved with one atomic per
  // wave and counter, and the independent requests of a pass are all in flight together: costs + work items ->
  // arc records -> atomicMin on the table and the link slots -> token slots and the epsilon test -> epsilon slots.
#if defined(PK2_LATP_PROFILE) && defined(PK2_LATP_EXPAND)
  long long exp_last = wall_clock64();
#endif
  // (the workgroup's threads stay together: slots are reserved by the workgroup)
  for (int jb = c.wg * kLatThreads; jb < n_arcs; jb += 4 * stride) {
    const int j0 = jb + tid;
    float2 tc2[4]; int2 wk[4]; int4 er[4]; uint32_t old[4]; bool acc[4], made[4], eps[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + q * stride;
      acc[q] = j < n_arcs;
      if (acc[q]) { tc2[q] = wcost[j]; wk[q] = V.work[j]; er[q] = make_int4(wdst[j], wtid[j], wgc[j], 0); }
    }
    EXP_T(0);
    int n_acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = acc[q] && tc2[q].x < next_cutoff;
      n_acc += acc[q];
    }
    // one hop: atomicMin on the table, the link slots, the epsilon degree of the destinations
    int deg[4], ea0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      deg[q] = 0; ea0[q] = 0;
      if (acc[q]) {
        old[q] = atomicMin(&V.stc[er[q].x], enc_cost(tc2[q].x));
        ea0[q] = p.g.n_off[er[q].x];
        deg[q] = p.g.n_off[er[q].x + 1] - ea0[q];
      }
    }
    int li;
    {
      const int cnt[1] = {n_acc};
      int fst[1];
      block_alloc<1>(s_scr, &F->n_link, nullptr, cnt, fst);
      li = l0 + fst[0];
    }
    EXP_T(1);
    int n_made = 0, n_eps = 0;
    bool heavy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      made[q] = acc[q] && old[q] == kEmpty;
      n_made += made[q];
      heavy[q] = made[q] && deg[q] > kHeavyDegree;          // -> the team's heavy list
      eps[q] = made[q] && deg[q] > 0 && !heavy[q];
      n_eps += eps[q];
    }
    // one hop: token slots and epsilon-list slots together
    int ti, ei;
    {
      const int cnt[2] = {n_made, n_eps};
      int fst[2];
      block_alloc<2>(s_scr, &F->n_new, &F->n_elist, cnt, fst);
      ti = fst[0]; ei = fst[1];
    }
    EXP_T(2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!acc[q]) continue;
      if (made[q]) {
        if (fb + ti < V.tok_cap) {
          V.ts[fb + ti] = er[q].x;
          V.tc[fb + ti] = INFINITY;
          st_coherent(&V.stt[er[q].x], ti);
          if (eps[q]) {
            elist_put(V, F, ei, fb + ti, er[q].x, ea0[q], deg[q]);
            ++ei;
          }
          if (heavy[q] && !team_register_heavy(F, c.G, fb + ti, er[q].x, ea0[q], deg[q])) {      // (rare; the list is full: an ordinary entry)
            const int e = atomicAdd(&F->n_elist, 1);
            elist_put(V, F, e, fb + ti, er[q].x, ea0[q], deg[q]);
          }
        } else {
          st_coherent(&F->status, (int32_t)kLatTokenOverflow);
          if (eps[q]) { elist_put(V, F, ei, 0, er[q].x, ea0[q], deg[q]); ++ei; }   // (a reserved slot holds valid indices)
        }
        ++ti;
      }
      if (li < V.link_cap) {
        V.lrec[li] = make_int4(wk[q].x, er[q].x /* state for now */, er[q].y, er[q].z);
        V.lac[li] = tc2[q].y;
      } else {
        st_coherent(&F->status, (int32_t)kLatLinkOverflow);
      }
      ++li;
    }
    EXP_T(3);
  }
  if (!defer && team_last(F, c.G, &s_flag) && tid == 0) expand_last(c);
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_expand(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ int s_flag, s_scr[64];
  phase_expand(p, team_ctx(p, cnt, local), s_flag, s_scr);
}

// (round0_last / round_last return the round's "a cost was lowered" flag, close_last "the utterance's status is not ok")
__device__ __forceinline__ unsigned round0_last(const TeamCtx& c) {
  LatFrame* F = c.F;
  if (!c.live) return 0u;
  const int l0 = F->link_end, nl = min(ld_coherent(&F->n_link), c.V.link_cap - l0);
  F->link_end = l0 + nl;
  c.V.seg[2 * c.t + 2] = l0 + nl;
  st_coherent(&F->n_link, 0);
  F->ne_snap = min(ld_coherent(&F->n_elist), elist_cap(c.V));
  F->nh_snap = min(ld_coherent(&F->n_hlist), kLatTeamHeavy);
  return (unsigned)(ld_coherent(&F->changed[0]) != 0);
}
__device__ __forceinline__ unsigned round_last(const TeamCtx& c, int r) {
  LatFrame* F = c.F;
  if (!c.live) return 0u;
  F->ne_snap = min(ld_coherent(&F->n_elist), elist_cap(c.V));
  F->nh_snap = min(ld_coherent(&F->n_hlist), kLatTeamHeavy);
  return (unsigned)(ld_coherent(&F->changed[r]) != 0);
}

// ---- link destinations (state -> token index), epsilon relaxation round 0 ----
__device__ __forceinline__ void phase_round0(const DecodeParams& p, const TeamCtx& c, Shared& sh, int& s_flag, bool defer = false) {
  if (!c.live) return;
  const int tid = lat_tid();
  const UttView& V = c.V;
  LatFrame* F = c.F;
  if (tid == 0) sh.n_heavy = 0;
  const Hot* h = c.hot;
  // (the emitting links' destinations, state -> token index, are resolved in the close phase, inside a chain of loads that
  // is there anyway; here they were two dependent round trips in front of the relaxation)
  const int fb = h ? h->f1 : F->f1;
  lds_barrier();
  const int changed = eps_round(p, V, F, sh, fb, h ? h->build_cutoff : F->build_cutoff, h ? min(h->ne, elist_cap(V)) : F->ne_snap,
                                h ? min(h->nh, kLatTeamHeavy) : F->nh_snap, c.wg, c.G, c.G);
  if (changed && tid == 0) st_coherent(&F->changed[0], 1);
  if (!defer && team_last(F, c.G, &s_flag) && tid == 0) round0_last(c);
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_round0(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ Shared sh;
  __shared__ int s_flag;
  phase_round0(p, team_ctx(p, cnt, local), sh, s_flag);
}

// ---- epsilon relaxation round r >= 1 ----
__device__ __forceinline__ void phase_round(const DecodeParams& p, const TeamCtx& c, Shared& sh, int& s_flag, int r, bool defer = false) {
  if (!c.live) return;
  LatFrame* F = c.F;
  if (!defer && !F->changed[r - 1]) return;          // (the persistent kernel has the flag from the barrier's release word)
  const int tid = lat_tid();
  if (tid == 0) sh.n_heavy = 0;
  lds_barrier();
  const Hot* h = c.hot;
  const int changed = eps_round(p, c.V, F, sh, h ? h->f1 : F->f1, h ? h->build_cutoff : F->build_cutoff,
                                h ? min(h->ne, elist_cap(c.V)) : F->ne_snap, h ? min(h->nh, kLatTeamHeavy) : F->nh_snap, c.wg, c.G, c.G);
  if (changed && tid == 0) st_coherent(&F->changed[r], 1);
  if (!defer && team_last(F, c.G, &s_flag) && tid == 0) round_last(c, r);
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_round(const DecodeParams p, const StepCounter* cnt, int local, int r) {
  __shared__ Shared sh;
  __shared__ int s_flag;
  phase_round(p, team_ctx(p, cnt, local), sh, s_flag, r);
}

// ---- deeper epsilon chains: one workgroup relaxes to the fixed point ----
__device__ __forceinline__ void phase_tail(const DecodeParams& p, const TeamCtx& c, Shared& sh) {
  if (!c.live) return;
  LatFrame* F = c.F;
  const Hot* h = c.hot;
  if (!h && !F->changed[kLatEpsRounds]) return;           // (the persistent kernel has the flag from the barrier's release word)
  const int tid = lat_tid();
  if (tid == 0) sh.n_heavy = 0;
  __syncthreads();
  for (int rounds = 0; ; ++rounds) {
    const int ne = min(ld_coherent(&F->n_elist), elist_cap(c.V));   // single workgroup: its own appends, ordered by the barriers
    const int nh = min(ld_coherent(&F->n_hlist), kLatTeamHeavy);
    if (!eps_round(p, c.V, F, sh, h ? h->f1 : F->f1, h ? h->build_cutoff : F->build_cutoff, ne, nh, 0, 1, c.G)) break;
    if (ld_coherent(&F->status) != kLatOk) break;
    if (rounds > kMaxEpsRounds) { if (tid == 0) st_coherent(&F->status, (int32_t)kLatEpsilonLoop); break; }
  }
  __syncthreads();
  if (tid == 0) {
    F->ne_snap = min(ld_coherent(&F->n_elist), elist_cap(c.V));
    F->nh_snap = min(ld_coherent(&F->n_hlist), kLatTeamHeavy);
  }
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_tail(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ Shared sh;
  phase_tail(p, team_ctx(p, cnt, local), sh);
}

// ---- the frame is closed: epsilon links from the final costs, final token costs, arc ranges, best cost ----
__device__ __forceinline__ unsigned close_last(const TeamCtx& c) {
  LatFrame* F = c.F;
  if (!c.live) return 1u;
  const UttView& V = c.V;
  const int fb = F->f1, cnt_new = min(ld_coherent(&F->n_new), V.tok_cap - fb);
  const int tok_end = fb + cnt_new;
  const int link_end = min(F->link_end + ld_coherent(&F->n_link), V.link_cap);
  V.seg[2 * c.t + 3] = link_end;
  V.ftok[c.t + 2] = tok_end;
  F->f0 = fb; F->f1 = tok_end; F->link_end = link_end;
  F->best_key = ld_coherent(&F->best_next);
  st_coherent(&F->best_next, kEmpty);
  st_coherent(&F->n_new, 0); st_coherent(&F->n_link, 0); st_coherent(&F->n_elist, 0); st_coherent(&F->n_hlist, 0);
  F->ne_snap = 0; F->nh_snap = 0;
  for (int r = 0; r <= kLatEpsRounds; ++r) st_coherent(&F->changed[r], 0);
  return (unsigned)(ld_coherent(&F->status) != kLatOk);
}
__device__ __forceinline__ void phase_close(const DecodeParams& p, const TeamCtx& c, Shared& sh, int& s_flag, bool defer = false) {
  if (!c.live) return;
  const int tid = lat_tid();
  const UttView& V = c.V;
  LatFrame* F = c.F;
  if (tid == 0) sh.n_heavy = 0;
  lds_barrier();
  const Hot* h = c.hot;
  // (persistent kernel: the link counter runs on from the frame's emitting links, so the base stays the frame's first link)
  const int fb = h ? h->f1 : F->f1, l0 = h ? h->l0 : F->link_end;
  const float cutoff = h ? h->build_cutoff : F->build_cutoff;
  auto eps_link = [&](int i, float cc, int a, float w, int d) {
    const float tot = cc + w;
    const int li = l0 + wave_alloc(&F->n_link, tot < cutoff);
    if (tot < cutoff) {
      if (li < V.link_cap) {
        V.lrec[li] = make_int4(i, fb + V.stt[d], 0, __float_as_int(w));
        V.lac[li] = 0.f;
      } else {
        st_coherent(&F->status, (int32_t)kLatLinkOverflow);
      }
    }
  };
  for_each_eps_record(sh, p, V, h ? min(h->ne, elist_cap(V)) : F->ne_snap,
                      [&](int i, int s, float* cc) { *cc = dec_cost(V.stc[s]); return *cc < cutoff; },
                      eps_link, team_entry(c.wg, c.G), c.G * kLatThreads);
  team_heavy_arcs(p, V, F, sh, h ? min(h->nh, kLatTeamHeavy) : F->nh_snap, c.wg, c.G, c.G, cutoff, false, eps_link);
  // final costs and arc ranges of the new tokens (the state table is cleared by the next frame's list launch)
  const int cnt_new = min(h ? h->n_new : F->n_new, V.tok_cap - fb);
  uint32_t kmin = kEmpty;
  const int stride = c.G * kLatThreads;
  // The frame's emitting links [le0, le1) still name their destination STATE: state -> token index here, the first two links
  // of a thread inside the token loop's own chain of loads (state, then table entry, then stores).
  int le0 = 0, le1 = 0;
  if (c.t >= 0) {
    if (h) { le0 = h->l0; le1 = le0 + min(h->n_link, V.link_cap - le0); }
    else { le0 = V.seg[2 * c.t + 1]; le1 = V.seg[2 * c.t + 2]; }
  }
  const int la = le0 + c.wg * kLatThreads + tid, lb = la + stride;
  int ls0 = 0, ls1 = 0;
  if (la < le1) ls0 = V.lrec[la].y;
  if (lb < le1) ls1 = V.lrec[lb].y;
  bool links_pending = la < le1;
  for (int i0 = fb + c.wg * kLatThreads + tid; i0 < fb + cnt_new; i0 += 4 * stride) {
    int st[4]; uint32_t ck[4]; int a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i0 + q * stride < fb + cnt_new) st[q] = V.ts[i0 + q * stride];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * stride < fb + cnt_new) { ck[q] = V.stc[st[q]]; a0[q] = p.g.e_off[st[q]]; a1[q] = p.g.e_off[st[q] + 1]; }
    int lt0 = 0, lt1 = 0;
    if (links_pending) { lt0 = V.stt[ls0]; if (lb < le1) lt1 = V.stt[ls1]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * stride;
      if (i < fb + cnt_new) {
        V.tc[i] = dec_cost(ck[q]);
        V.tarc[i] = make_int2(a0[q], a1[q] - a0[q]);
        V.te[i] = INFINITY;
        kmin = min(kmin, ck[q]);
      }
    }
    if (links_pending) {
      V.lrec[la].y = fb + lt0;
      if (lb < le1) V.lrec[lb].y = fb + lt1;
      links_pending = false;
    }
  }
  if (links_pending) {                  // (no token of the loop above was this thread's)
    V.lrec[la].y = fb + V.stt[ls0];
    if (lb < le1) V.lrec[lb].y = fb + V.stt[ls1];
  }
  for (int l = la + 2 * stride; l < le1; l += stride) V.lrec[l].y = fb + V.stt[V.lrec[l].y];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
  // (one atomic per workgroup: 256 of them on the frame record's cache line took microseconds to drain)
  if ((tid & 63) == 0) sh.redi[tid >> 6] = (int)kmin;
  lds_barrier();
  if (tid == 0) {
    uint32_t m = kEmpty;
    for (int w = 0; w < kLatWaves; ++w) m = min(m, (uint32_t)sh.redi[w]);
    if (m != kEmpty) atomicMin(&F->best_next, m);
  }
  if (!defer && team_last(F, c.G, &s_flag) && tid == 0) close_last(c);
}
__global__ void __launch_bounds__(kLatThreads) lat_frames_close(const DecodeParams p, const StepCounter* cnt, int local) {
  __shared__ Shared sh;
  __shared__ int s_flag;
  phase_close(p, team_ctx(p, cnt, local), sh, s_flag);
}

// ---- the same frames inside ONE launch: a team of workgroups of one XCD per utterance ----
// A kernel boundary between two phases of a frame costs the graph-replayed launch chain ~4-5 us (dependent launch, cold
// L1s, parameters) around 2-4 us of work; eight of them per frame.  Here the phases of ALL frames of an utterance run
// inside one launch: the G workgroups of a team sit on the CUs of ONE XCD (teams form by arrival order under the XCC_ID
// register, as in the persistent recurrences; utterances are handed out from a queue), so everything they exchange lives
// in that XCD's L2.  Between two phases: every thread waits for its stores (the vector L1 writes through: they are in
// L2 then), the team meets at a counter in L2 (one agent-scope atomic per workgroup, one thread polls), and the L1 is
// invalidated (buffer_inv sc1) before the next phase reads what the others wrote.  The phase bodies are the launch-per-
// frame kernels' own; rounds that have nothing to do are skipped WITH their barrier (the flags they test are final
// behind the barrier of the round before), and an utterance stops at its own last frame, not at the longest one's.
// A poll that waits longer than 1 s raises the abort flag: everybody leaves, lat_persist_check marks the utterances "not
// decoded" and the caller's summary reports it; the first launch on a device is verified on the host and the launch-per-
// frame decoder takes over when it does not come back complete.
constexpr int kLatTeamsPerXcd = 4;
constexpr int kLatMaxIter = 64;
constexpr long long kLatSpinTicks = 1000LL * 1000 * 100;     // 1 s of the 100 MHz wall clock
struct LatTeamCtl {
  unsigned arrive[8];
  unsigned next_utt, abort, done, pad[5];
  // bar: arrivals (monotonic); rel: the release word of lat_team_barrier_last on a cache line of its own -- barrier number in
  // the low half, the last arriver's payload (a flag the next phase branches on) in the high half
  // (lat_team_barrier_hand: 7 bits of release number, 57 bits of payload.)  sig: the same for the cutoff, which rank 0 hands
  // to the workgroups that list the frame's arcs without a barrier (they owe rank 0 nothing at that point).
  struct Team {
    unsigned task[kLatMaxIter + 1]; unsigned bar; unsigned pad[30];
    unsigned long long rel; unsigned pad2[30];
    unsigned long long sig; unsigned pad3[30];
  } team[8][kLatTeamsPerXcd];
};
static_assert(sizeof(LatTeamCtl::Team) == 640 && offsetof(LatTeamCtl::Team, rel) == 384 && offsetof(LatTeamCtl::Team, sig) == 512,
              "team record: release words on cache lines of their own");

struct LatSpin {
  LatTeamCtl* ctl; long long t0; unsigned n;
  __device__ __forceinline__ explicit LatSpin(LatTeamCtl* c) : ctl(c), t0(0), n(0) {}
  __device__ __forceinline__ bool expired() {
    if ((++n & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > kLatSpinTicks || ld_coherent(&ctl->abort)) {
      st_coherent(&ctl->abort, 1u);
      return true;
    }
    return false;
  }
};

// Team barrier between two phases (see above).  Returns false after an abort.
__device__ __forceinline__ bool lat_team_barrier(LatTeamCtl* ctl, LatTeamCtl::Team* tm, int G, unsigned* nbar, int* s_abort, int inv_mode) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned target = (unsigned)G * ++*nbar;
  if (lat_tid() < 64) {
    if (lat_tid() == 0) {
      __hip_atomic_fetch_add(&tm->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      LatSpin spin(ctl);
      while (ld_coherent(&tm->bar) < target) {
        if (spin.expired()) { *s_abort = 1; break; }
      }
    }
    // what the other workgroups wrote before the barrier is in L2: drop the (now possibly stale) lines of this CU's L1
    // and of the scalar cache -- ONE wave does it for the workgroup (PK2_LAT_INV=0, every wave: the decode took 2.2x as long)
    if (inv_mode == 1) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (inv_mode == 0) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return *s_abort == 0;
}

// The team barrier whose LAST arriver runs `last_fn` (one thread) before it releases the others: arrival = a returning
// atomic on the team's monotonic counter, release = a second monotonic word the others poll.  Every workgroup has waited for
// its own stores and atomics (vmcnt) before it arrives, so the last arriver sees the phase's final counters in L2; its
// bookkeeping stores are acknowledged before the release word goes out.  `last_fn` returns a payload that travels IN the
// release word (the flag the next phase branches on: "did the round lower a cost", "is the utterance's status still ok"),
// which saves every workgroup the dependent load of that flag behind the barrier; *payload holds it on return.
template <class F_>
__device__ __forceinline__ bool lat_team_barrier_last(LatTeamCtl* ctl, LatTeamCtl::Team* tm, int G, unsigned* nbar, int* s_abort,
                                                      unsigned* s_pay, int inv_mode, unsigned* payload, F_ last_fn) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned gen = ++*nbar, target = (unsigned)G * gen;
  if (lat_tid() < 64) {
    if (lat_tid() == 0) {
      const unsigned old = __hip_atomic_fetch_add(&tm->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == target) {
        const unsigned pay = last_fn();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st_coherent(&tm->rel, ((unsigned long long)pay << 32) | gen);
        *s_pay = pay;
      } else {
        LatSpin spin(ctl);
        unsigned long long w;
        while ((unsigned)(w = ld_coherent(&tm->rel)) < gen) {
          if (spin.expired()) { *s_abort = 1; break; }
        }
        *s_pay = (unsigned)(w >> 32);
      }
    }
    if (inv_mode == 1) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (inv_mode == 0) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  *payload = *s_pay;
  return *s_abort == 0;
}

// Round 6, second form: the barrier hands over VALUES.  `pre` (last arriver, one thread) reads the counters the phase's
// atomics left in L2 and packs what the next phase needs into 57 bits; the release word carries them to every workgroup;
// `post` (same thread, AFTER the release) writes what only later phases or later kernels read (segment bounds, counter
// resets: they are acknowledged before that workgroup arrives at the next barrier).  A field that does not fit its bits is
// sent as all-ones and the receivers read the counter themselves (kField*: 2^23 - 1 tokens or links in ONE frame).
// Release numbers are compared for equality on 7 bits: a workgroup is never more than one release behind the word.
constexpr unsigned kRelMask = 127u;
constexpr unsigned long long kMask23 = (1ull << 23) - 1ull, kMask25 = (1ull << 25) - 1ull;
// (sender: a value that needs more bits goes into the record's hand[slot] -- acknowledged before the release -- and all-ones
// into the word; receiver: all-ones = read hand[slot], which the next sender touches only after every receiver has arrived at
// the next barrier)
__device__ __forceinline__ unsigned long long fld(int v, unsigned long long mx, int32_t* slot) {
  if ((unsigned long long)(unsigned)v < mx) return (unsigned long long)(unsigned)v;
  st_coherent(slot, v);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return mx;
}
__device__ __forceinline__ int unfld(unsigned long long f, unsigned long long mx, const int32_t* slot) {
  return f == mx ? ld_coherent(slot) : (int)f;
}
template <class Pre, class Post, class Recv>
__device__ __forceinline__ bool lat_team_barrier_hand(LatTeamCtl* ctl, LatTeamCtl::Team* tm, int G, unsigned* nbar, unsigned* nrel, int* s_abort,
                                                      int inv_mode, Pre pre, Post post, Recv recv) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned target = (unsigned)G * ++*nbar, gen = ++*nrel & kRelMask;
  if (lat_tid() < 64) {
    if (lat_tid() == 0) {
      const unsigned old = __hip_atomic_fetch_add(&tm->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == target) {
        const unsigned long long pay = pre();
        st_coherent(&tm->rel, (pay << 7) | gen);
        recv(pay);
        post(pay);
      } else {
        LatSpin spin(ctl);
        unsigned long long w;
        while (((unsigned)(w = ld_coherent(&tm->rel)) & kRelMask) != gen) {
          if (spin.expired()) { *s_abort = 1; break; }
        }
        recv(w >> 7);
      }
    }
    if (inv_mode == 1) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (inv_mode == 0) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return *s_abort == 0;
}

// The persistent kernel calls this at the top of every frame: the compiler otherwise hoists the per-thread addresses of all
// phases (pointer + f(thread)) out of the frame loop, keeps them alive across every phase and spills them (the workgroup has
// 128 registers per thread); scratch reloads behind an L1 invalidation are round trips to L2.  Behind an opaque asm the
// addresses are computed where they are used (a few VALU instructions).
__device__ __forceinline__ void launder_view(UttView& V, LatFrame*& F) {
#if PK2_LAT_LAUNDER
  asm volatile("" : "+s"(V.stc), "+s"(V.stt), "+s"(V.ts), "+s"(V.tc), "+s"(V.te), "+s"(V.tarc), "+s"(V.work), "+s"(V.work_tot));
  asm volatile("" : "+s"(V.elist), "+s"(V.ftok), "+s"(V.seg), "+s"(V.lrec), "+s"(V.lac), "+s"(V.erec), "+s"(F));
#endif
}

// One utterance's frames with the record's hot fields in the workgroup's LDS (see Hot; `h`: written by thread 0 inside the
// barriers, read by everybody behind them).  Returns false after an abort.
__device__ __forceinline__ bool persist_frames_hot(const DecodeParams& p, LatTeamCtl* ctl, LatTeamCtl::Team* tm, TeamCtx& c, Shared& sh,
                                                   Hot& h, int& s_flag, int& s_base, int& s_abort, unsigned& nbar, unsigned& nrel,
                                                   unsigned& nsig, int list_first, int inv_mode, int field_bits, int test_stall,
                                                   long long* lp_acc, int* lp_frames_p) {
  // (the largest count a 23-bit / 25-bit field carries itself; PK2_LAT_FIELD_BITS narrows them so that tests drive the spill path)
  const unsigned long long kF23 = (1ull << min(max(field_bits, 1), 23)) - 1ull, kF25 = (1ull << min(max(field_bits + 2, 1), 25)) - 1ull;
#ifdef PK2_LATP_PROFILE
  long long lp_last = 0;
#define LPH(k) do { const long long n_ = wall_clock64(); lp_acc[k] += n_ - lp_last; lp_last = n_; } while (0)
#else
#define LPH(k) do { } while (0)
#endif
  typedef unsigned long long u64;
  const int tid = lat_tid(), rank = c.wg, G = c.G, T = c.T;
  LatFrame*& F = c.F;
  const UttView& V = c.V;
  if (tid == 0) {
    // (behind the barrier that follows phase_init: the record as InitDecoding left it)
    h.f0 = F->f0; h.f1 = F->f1; h.l0 = F->link_end; h.best_key = F->best_key;
    h.cur_cutoff = F->cur_cutoff; h.adaptive = F->adaptive; h.build_cutoff = F->build_cutoff;
    h.n_arcs = 0; h.nmin_key = kEmpty; h.n_link = 0;
    h.ne = F->ne_snap; h.nh = F->nh_snap; h.n_new = ld_coherent(&F->n_new);
    h.ll_base = F->ll_base; h.ll_stride = F->ll_stride;
    h.more = 0; h.status_bad = 0;
  }
  __syncthreads();
  c.hot = &h;
  auto hand = [&](auto pre, auto post, auto recv) {
    return lat_team_barrier_hand(ctl, tm, G, &nbar, &nrel, &s_abort, inv_mode, pre, post, recv);
  };
  for (int t = -1; t < T; ++t) {
    c.t = t;
    launder_view(c.V, c.F);
    // (test hook, PK2_LAT_TEST_STALL=1: one workgroup of every team walks away in frame 2 -- its team mates must give up after
    // the poll's time-out, the check kernel behind the launch must report every utterance "not decoded" and raise the guard)
    if (test_stall && rank == G - 1 && t == 2) return false;
#ifdef PK2_LATP_PROFILE
    lp_last = wall_clock64(); ++*lp_frames_p;
#endif
    if (t >= 0) {
      if (h.status_bad) break;
      // ---- cutoff: rank 0 selects it and hands it over in a word of its own; nobody waits for anybody else here ----
      const unsigned sgen = ++nsig & kRelMask;
      if (rank == 0) {
        float co[3];
        phase_cutoff(p, c, sh, co);
        if (tid == 0) {
          h.cur_cutoff = co[0]; h.adaptive = co[1];
          // (the record's n_arcs / nmin_key were reset by this thread: acknowledged before the word goes out)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st_coherent(&tm->sig, ((u64)__float_as_uint(co[0]) << 8) | ((co[2] != 0.f ? 1ull : 0ull) << 7) | sgen);
        }
        if (!list_first) __syncthreads();
        LPH(0);
      } else {
        if (list_first) phase_list_pre(p, c, sh, list_first);
        LPH(0);
        if (tid == 0) {
          LatSpin spin(ctl);
          u64 w;
          while (((unsigned)(w = ld_coherent(&tm->sig)) & kRelMask) != sgen) {
            if (spin.expired()) { s_abort = 1; break; }
          }
          h.cur_cutoff = __uint_as_float((unsigned)(w >> 8));
          h.adaptive = ((w >> 7) & 1ull) ? (h.cur_cutoff - dec_cost(h.best_key)) + p.beam_delta : p.beam;
        }
        __syncthreads();
        if (s_abort) return false;
        LPH(1);
      }
      // ---- list ----
      phase_list(p, c, sh, s_base, list_first, list_first != 0);
      LPH(2);
      if (!hand([&]() -> u64 { return (fld(ld_coherent(&F->n_arcs), kF25, &F->hand[0]) << 32) | (u64)ld_coherent(&F->nmin_key); },
                [](u64) { },
                [&](u64 w) {
                  h.nmin_key = (uint32_t)w;
                  h.n_arcs = unfld(w >> 32, kF25, &F->hand[0]);
                  const float nmin = dec_cost((uint32_t)w);
                  if (nmin < INFINITY) h.build_cutoff = nmin + h.adaptive;
                })) return false;
      LPH(3);
      // ---- expand ----
      phase_expand(p, c, s_flag, reinterpret_cast<int*>(sh.hist), true);
      LPH(4);
#if defined(PK2_LATP_PROFILE) && defined(PK2_LATP_EXPAND)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); LPH(10);     // (tail slot: the phase's drain)
#endif
      if (!hand([&]() -> u64 { const int a = ld_coherent(&F->n_link), b = ld_coherent(&F->n_elist), d = ld_coherent(&F->n_hlist);
                               return (fld(a, kF23, &F->hand[0]) << 28) | (fld(b, kF23, &F->hand[1]) << 5) | (u64)min(d, kLatTeamHeavy); },
                [](u64) { },
                [&](u64 w) {
                  h.n_link = unfld(w >> 28, kF23, &F->hand[0]);
                  h.ne = unfld((w >> 5) & kMask23, kF23, &F->hand[1]);
                  h.nh = (int)(w & 31ull);
                })) return false;
      LPH(5);
    }
    // ---- link destinations, relaxation rounds ----
    phase_round0(p, c, sh, s_flag, true);
    LPH(6);
    for (int r = 0; ; ++r) {
      if (r > 0) { phase_round(p, c, sh, s_flag, r, true); LPH(8); }
      // {a cost was lowered, new tokens, epsilon list, heavy list}
      if (!hand([&]() -> u64 { const int fl = ld_coherent(&F->changed[r]), b = ld_coherent(&F->n_elist), d = ld_coherent(&F->n_hlist),
                                         e = ld_coherent(&F->n_new);
                               return ((u64)(fl != 0) << 51) | (fld(e, kF23, &F->hand[0]) << 28) | (fld(b, kF23, &F->hand[1]) << 5) |
                                      (u64)min(d, kLatTeamHeavy); },
                [&](u64) { if (r == 0) V.seg[2 * t + 2] = h.l0 + min(h.n_link, V.link_cap - h.l0); },
                [&](u64 w) {
                  h.more = (int)((w >> 51) & 1ull);
                  h.n_new = unfld((w >> 28) & kMask23, kF23, &F->hand[0]);
                  h.ne = unfld((w >> 5) & kMask23, kF23, &F->hand[1]);
                  h.nh = (int)(w & 31ull);
                })) return false;
      if (r == 0) LPH(7); else LPH(9);
      if (!h.more || r == kLatEpsRounds) break;
    }
    if (h.more) {                                            // (all rounds ran and the last one still lowered a cost)
      if (rank == 0) phase_tail(p, c, sh);
      if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return false;
      if (tid == 0) { h.ne = ld_coherent(&F->n_elist); h.nh = min(ld_coherent(&F->n_hlist), kLatTeamHeavy); h.n_new = ld_coherent(&F->n_new); }
      __syncthreads();
    }
    // ---- close ----
    phase_close(p, c, sh, s_flag, true);
    LPH(12);
    if (!hand([&]() -> u64 { const int a = ld_coherent(&F->n_link); const uint32_t bk = ld_coherent(&F->best_next);
                             const int st = ld_coherent(&F->status);
                             return ((u64)(st != kLatOk) << 55) | (fld(a, kF23, &F->hand[0]) << 32) | (u64)bk; },
              [&](u64) {  // (behind recv: h is the NEXT frame's) what later kernels read, and the counters for the next frame
                V.seg[2 * t + 3] = h.l0;
                V.ftok[t + 2] = h.f1;
                F->f0 = h.f0; F->f1 = h.f1; F->link_end = h.l0; F->best_key = h.best_key;
                st_coherent(&F->best_next, kEmpty);
                st_coherent(&F->n_new, 0); st_coherent(&F->n_link, 0); st_coherent(&F->n_elist, 0); st_coherent(&F->n_hlist, 0);
                for (int r = 0; r <= kLatEpsRounds; ++r) st_coherent(&F->changed[r], 0);
              },
              [&](u64 w) {
                const int fb = h.f1, a = unfld((w >> 32) & kMask23, kF23, &F->hand[0]);
                h.status_bad = (int)((w >> 55) & 1ull);
                h.l0 = min(h.l0 + a, V.link_cap);
                h.best_key = (uint32_t)w;
                h.f0 = fb; h.f1 = fb + min(h.n_new, V.tok_cap - fb);
                h.n_link = 0; h.ne = 0; h.nh = 0; h.n_new = 0;
              })) return false;
    LPH(13);
  }
#undef LPH
  c.hot = nullptr;
  return true;
}

__global__ void __launch_bounds__(kLatThreads) lat_frames_persist(const DecodeParams p, LatTeamCtl* ctl, int N, int G, int teams_per_xcd, int inv_mode,
                                                                  int field_bits, int test_stall) {
  __shared__ Shared sh;
  __shared__ int s_flag, s_base, s_abort, s_i[4];
  __shared__ unsigned s_pay;
  __shared__ Hot s_hot;
  const int tid = lat_tid();
  if (tid == 0) {
    unsigned xcd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
    xcd &= 7u;
    const unsigned slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_i[0] = (int)(slot % (unsigned)G); s_i[1] = (int)(slot / (unsigned)G); s_i[2] = (int)xcd;
    s_abort = 0;
  }
  __syncthreads();
  // (workgroup-uniform values read back from LDS: readfirstlane keeps them -- and every pointer derived from them -- in SGPRs)
  const int rank = __builtin_amdgcn_readfirstlane(s_i[0]);
  if (s_i[1] >= teams_per_xcd) return;                     // (whole teams only; the host sizes them for 32 CUs per XCD)
  LatTeamCtl::Team* tm = &ctl->team[__builtin_amdgcn_readfirstlane(s_i[2])][__builtin_amdgcn_readfirstlane(s_i[1])];
  unsigned nbar = 0, nrel = 0, nsig = 0;
#ifndef PK2_LAT_HOT
#define PK2_LAT_HOT 1
#endif
#ifndef PK2_LAT_LIST_PRE
#define PK2_LAT_LIST_PRE 1
#endif
  const int list_first = (PK2_LAT_LIST_PRE && G > 2) ? 1 : 0;      // ranks that take slices of the list phase: list_first .. G - 1
#ifndef PK2_LAT_MERGE_LAST
#define PK2_LAT_MERGE_LAST 1
#endif
  constexpr bool kMergeLast = PK2_LAT_MERGE_LAST != 0;
#ifdef PK2_LATP_PROFILE
  long long lp_acc[16], lp_last = 0; int lp_frames = 0;
  for (int k = 0; k < 16; ++k) lp_acc[k] = 0;
#define LP_T(k) do { const long long n_ = wall_clock64(); lp_acc[k] += n_ - lp_last; lp_last = n_; } while (0)
#else
#define LP_T(k) do { } while (0)
#endif
  for (int iter = 0; iter <= kLatMaxIter; ++iter) {
    // the team's next utterance
    if (tid == 0) {
      unsigned k;
      if (rank == 0) {
        k = __hip_atomic_fetch_add(&ctl->next_utt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_coherent(&tm->task[iter], k + 1u);
      } else {
        LatSpin spin(ctl);
        unsigned k1;
        while ((k1 = ld_coherent(&tm->task[iter])) == 0u) {
          if (spin.expired()) { s_abort = 1; k1 = 1u << 30; break; }
        }
        k = k1 - 1u;
      }
      s_i[3] = (int)k;
    }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(s_i[3]);
    if (s_abort || n >= N) return;
    if (rank == 0) phase_init(p, n, G);
    if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return;
    TeamCtx c = team_ctx_of(p, n, rank, G, -1);            // (the utterance's record and views: once, not per frame)
    const int T = c.T;
    LatFrame* F = c.F;
#if PK2_LAT_HOT
#ifdef PK2_LATP_PROFILE
    if (!persist_frames_hot(p, ctl, tm, c, sh, s_hot, s_flag, s_base, s_abort, nbar, nrel, nsig, list_first, inv_mode, field_bits, test_stall, lp_acc, &lp_frames)) return;
#else
    if (!persist_frames_hot(p, ctl, tm, c, sh, s_hot, s_flag, s_base, s_abort, nbar, nrel, nsig, list_first, inv_mode, field_bits, test_stall, nullptr, nullptr)) return;
#endif
#else
    // Barriers behind a phase with bookkeeping: merged form (the last arriver does the bookkeeping and hands the flag the
    // next phase branches on to everybody inside the release word) or, -DPK2_LAT_MERGE_LAST=0, round 5's form (team_last
    // inside the phase, plain barrier, every workgroup loads the flag).
    unsigned pay = 0u, status_bad = 0u;
#define LAT_BARRIER_LAST(FN)                                                                                           \
    do {                                                                                                                \
      if (kMergeLast) { if (!lat_team_barrier_last(ctl, tm, G, &nbar, &s_abort, &s_pay, inv_mode, &pay, [&] { return FN; })) return; } \
      else if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return;                                       \
    } while (0)
    for (int t = -1; t < T; ++t) {
      c.t = t;
#ifdef PK2_LATP_PROFILE
      lp_last = wall_clock64(); ++lp_frames;
#endif
      if (t >= 0) {
        // (nobody writes the status between the barrier that closed the previous frame and the one behind the cutoff)
        if (kMergeLast ? status_bad != 0u : ld_coherent(&F->status) != kLatOk) break;
        if (rank == 0) phase_cutoff(p, c, sh);
        else if (list_first) phase_list_pre(p, c, sh, list_first);
        LP_T(0);
        if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return;
        LP_T(1);
        phase_list(p, c, sh, s_base, list_first, list_first != 0);
        LP_T(2);
        if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return;
        LP_T(3);
        phase_expand(p, c, s_flag, reinterpret_cast<int*>(sh.hist), kMergeLast);
        LP_T(4);
        LAT_BARRIER_LAST(expand_last(c));
        LP_T(5);
      }
      phase_round0(p, c, sh, s_flag, kMergeLast);
      LP_T(6);
      LAT_BARRIER_LAST(round0_last(c));
      LP_T(7);
      // a round that lowered nothing ends the fixed point (the flags of later rounds stay clear); after the last round the
      // flag says whether the serial tail has to finish it
      bool more = kMergeLast ? pay != 0u : ld_coherent(&F->changed[0]) != 0;
      int r = 1;
      for (; r <= kLatEpsRounds && more; ++r) {
        phase_round(p, c, sh, s_flag, r, kMergeLast);
        LP_T(8);
        LAT_BARRIER_LAST(round_last(c, r));
        LP_T(9);
        more = kMergeLast ? pay != 0u : ld_coherent(&F->changed[r]) != 0;
      }
      if (more) {                                           // (all rounds ran and the last one still lowered a cost)
        if (rank == 0) phase_tail(p, c, sh);
        LP_T(10);
        if (!lat_team_barrier(ctl, tm, G, &nbar, &s_abort, inv_mode)) return;
        LP_T(11);
      }
      phase_close(p, c, sh, s_flag, kMergeLast);
      LP_T(12);
      LAT_BARRIER_LAST(close_last(c));
      status_bad = pay;
      LP_T(13);
    }
#undef LAT_BARRIER_LAST
#endif
#ifdef PK2_LATP_PROFILE
    if (tid == 0 && rank == 0 && n == 0)
      printf("expand stages, rank 1 of utterance 1, 10 ns ticks per frame: loads %lld | table atomics + link slots %lld | token slots %lld | stores %lld\n",
             g_exp[0] / 1613, g_exp[1] / 1613, g_exp[2] / 1613, g_exp[3] / 1613);
    if (tid == 0 && rank == 0 && n == 0)
      printf("kth_below, 10 ns ticks per frame: zero hist %lld | pass 1 %lld | scan %lld | pass 2 %lld | rank %lld\n", g_cut[0] / lp_frames,
             g_cut[1] / lp_frames, g_cut[2] / lp_frames, g_cut[3] / lp_frames, g_cut[4] / lp_frames);
    if (tid == 0 && n == 1)
      printf("lat_frames_persist rank %d utt %d, %d frames, 10 ns ticks per frame: cutoff %lld bar %lld | list %lld bar %lld | expand %lld bar %lld | round0 %lld bar %lld | rounds %lld bar %lld | tail %lld bar %lld | close %lld bar %lld\n",
             rank, n, lp_frames, lp_acc[0] / lp_frames, lp_acc[1] / lp_frames, lp_acc[2] / lp_frames, lp_acc[3] / lp_frames, lp_acc[4] / lp_frames,
             lp_acc[5] / lp_frames, lp_acc[6] / lp_frames, lp_acc[7] / lp_frames, lp_acc[8] / lp_frames, lp_acc[9] / lp_frames, lp_acc[10] / lp_frames,
             lp_acc[11] / lp_frames, lp_acc[12] / lp_frames, lp_acc[13] / lp_frames);
#endif
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Behind the persistent launch: an aborted or incomplete launch leaves every utterance "not decoded".
__global__ void lat_persist_check(const DecodeParams p, const LatTeamCtl* ctl, int N, unsigned* sticky, unsigned* guard_dev, unsigned* guard_host) {
  if (ctl->abort == 0u && ctl->done == (unsigned)N) return;
  *sticky = 1u;
  if (lat_tid() == 0) persist_guard_raise(guard_dev, guard_host);
  for (int n = lat_tid(); n < N; n += blockDim.x) p.L.frame[n].status = kLatNotDecoded;
}

// ---- after the last frame: final costs + lattice-beam pruning, or the failure report ----
// The pruning pass is a min-plus recursion over the frames, last frame first: extra[s] = min over the links s -> d of
// extra[d] + (cost[s] + link cost - cost[d]), with an epsilon fixed point inside every frame.  One workgroup per utterance
// walks it; with everything in global memory a frame cost ~150 us (every epsilon round = three dependent round trips to
// L2 plus returning atomics).  Here the bracketed part of every link is computed beforehand by a parallel pass
// (lat_link_delta: it depends on the token costs only), and the extra costs of the frame being finished and of the frame
// before it live in LDS (order-preserving bit patterns, ds_min): a round is LDS traffic and a barrier.  The arithmetic and
// its order are those of finish_and_prune (lattice_decode_common.h: the one-workgroup decoder's pass).  A frame with more
// tokens than the LDS arrays hold (frame 0 has one per word of the vocabulary) is worked on in global memory.
#ifndef PK2_FIN_EPS
#define PK2_FIN_EPS 6       // (8 / 12 before: 18.9 ms per call, 4 / 16: 18.65, 6 / 14: 18.4 -- tools/gpu_fin_ab.sh, twice in one job)
#endif
#ifndef PK2_FIN_EMIT
#define PK2_FIN_EMIT 14
#endif
constexpr int kFinEps = PK2_FIN_EPS;  // epsilon links per thread kept in registers over the rounds of a frame
constexpr int kFinEmit = PK2_FIN_EMIT;       // emitting links per thread fetched ahead of the epsilon rounds
extern __shared__ __attribute__((aligned(16))) uint32_t lat_fin_smem[];

__global__ void __launch_bounds__(256) lat_link_delta(const DecodeParams p) {
  const int n = blockIdx.y, tid = lat_tid();
  const LatUtt U = p.L.utt[n];
  if (p.L.frame[n].status != kLatOk) return;
  const UttView V = make_view(p, n, U);
  float* delta = p.L.link_delta + U.link_base;
  for (int k = blockIdx.x; k <= 2 * U.T; k += gridDim.x) {      // segment k: epsilon links of frame k/2 (even), k/2 -> k/2+1 (odd)
    const int l0 = V.seg[k], l1 = V.seg[k + 1];
    if (k & 1) {
      for (int l = l0 + tid; l < l1; l += 256) {
        const int4 r = V.lrec[l];
        delta[l] = __fadd_rn(__fadd_rn(V.tc[r.x], V.lac[l]), __int_as_float(r.w)) - V.tc[r.y];
      }
    } else {
      for (int l = l0 + tid; l < l1; l += 256) {
        const int4 r = V.lrec[l];
        delta[l] = (V.tc[r.x] + __int_as_float(r.w)) - V.tc[r.y];
      }
    }
  }
}

// Round 5: __syncthreads() / __syncthreads_or() are workgroup-scope release / acquire fences -- each waits for every
// outstanding global load of the wave (s_waitcnt vmcnt(0)), i.e. for the link records fetched ahead "to land while the
// rounds run": they were waited for at the first round's barrier instead.  Frames whose extra costs live in LDS exchange
// nothing through memory: their barriers order LDS accesses only, the "did any lane lower a value" vote goes through three
// rotating LDS words (one barrier per round), and the epsilon records of frame t-1 are requested as soon as the rounds of
// frame t are over (they land under the write-back, the initialisation and the emitting links).
__device__ __forceinline__ void fin_barrier(bool lds_only) {
  if (lds_only) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

__global__ void __launch_bounds__(kLatThreads) lat_frames_finish(const DecodeParams p, int cap) {
  __shared__ float s_redf[kLatWaves];
  __shared__ int s_vote[3];
  const int n = blockIdx.x, tid = lat_tid();
  const LatUtt U = p.L.utt[n];
  const UttView V = make_view(p, n, U);
  const LatFrame* F = p.L.frame + n;
  if (F->status != kLatOk) {
    if (tid == 0) { p.L.utt[n].status = F->status; p.L.utt[n].n_tok = F->f1; p.L.utt[n].n_link = F->link_end; }
    return;
  }
  const int T = U.T, s_tok_end = F->f1, s_link_end = F->link_end;
  int anyf; float best_final;
  final_costs(p, V, s_redf, T, s_tok_end, &anyf, &best_final);
  uint32_t* A = lat_fin_smem;            // extra costs of frame t (when it fits: at most `cap` tokens)
  uint32_t* B = lat_fin_smem + cap;      // ... of frame t-1, being collected
  uint32_t* teu = reinterpret_cast<uint32_t*>(V.te);
  const float* delta = p.L.link_delta + U.link_base;
  const float lbeam = p.lattice_beam;
  int base = V.ftok[T], cnt = s_tok_end - base;
  bool a_lds = cnt <= cap;               // a frame with more tokens (frame 0: one per word) is worked on in global memory
  if (a_lds)
    for (int i = tid; i < cnt; i += kLatThreads) A[i] = teu[base + i];
  if (tid < 3) s_vote[tid] = 0;
  __syncthreads();
  int vote_round = 0;
  // "did any thread of the workgroup change a value" with ONE LDS-only barrier: the word of round r + 1 was last read in
  // round r - 2, before every thread's arrival at the barrier of round r - 1 -- thread 0 may clear it in round r
  auto any_changed = [&](int changed) -> bool {
    const int r = vote_round % 3;
    if (changed) s_vote[r] = 1;
    if (tid == 0) s_vote[(vote_round + 1) % 3] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ++vote_round;
    return s_vote[r] != 0;
  };
#ifdef PK2_LAT_FIN_DEBUG
  long long ph[6] = {0, 0, 0, 0, 0, 0}, last = wall_clock64(); int nrounds = 0;
#define FIN_T(k) do { __syncthreads(); const long long now_ = wall_clock64(); ph[k] += now_ - last; last = now_; } while (0)
#else
#define FIN_T(k) do { } while (0)
#endif
  // The segment bounds of a frame are loaded one frame before its records are requested (the address of a record load
  // would otherwise wait for them on the spot): sc = frame t's {first emitting link t-1 -> t, first epsilon link, end of the
  // epsilon links, first token of frame t-1}.
  struct FinSc { int m0, e0, e1, pb; };
  auto load_sc = [&](int t, FinSc& c) {
    c.m0 = c.e0 = c.e1 = c.pb = 0;
    if (t < 0) return;
    if (t > 0) { c.m0 = V.seg[2 * t - 1]; c.pb = V.ftok[t - 1]; }
    c.e0 = V.seg[2 * t]; c.e1 = V.seg[2 * t + 1];
  };
  FinSc sc, scn;
  load_sc(T, sc);
  // epsilon links inside a frame, the first kFinEps per thread (relative to the frame's first token)
  int es[kFinEps], ed[kFinEps]; float el[kFinEps];
  int e0 = 0, e1 = 0;
  auto load_eps = [&](const FinSc& c, int tbase) {
    e0 = c.e0; e1 = c.e1;
#pragma unroll
    for (int q = 0; q < kFinEps; ++q) {
      const int l = e0 + tid + q * kLatThreads;
      es[q] = -1; ed[q] = 0; el[q] = 0.f;
      if (l < e1) { const int4 r = V.lrec[l]; es[q] = r.x - tbase; ed[q] = r.y - tbase; el[q] = delta[l]; }
    }
  };
  load_eps(sc, base);
  for (int t = T; t >= 0; --t) {
    load_sc(t - 1, scn);
    // the emitting links t-1 -> t do not depend on the epsilon rounds below: their loads are issued first and land
    // while the rounds run (one exposed round trip to memory per frame instead of two)
    const int m0 = t > 0 ? sc.m0 : 0, m1 = t > 0 ? sc.e0 : 0;
    int2 mr[kFinEmit]; float md[kFinEmit];
#pragma unroll
    for (int q = 0; q < kFinEmit; ++q) {
      const int l = m0 + tid + q * kLatThreads;
      mr[q] = make_int2(0, 0); md[q] = 0.f;
      if (l < m1) { mr[q] = *reinterpret_cast<const int2*>(&V.lrec[l]); md[q] = delta[l]; }
    }
    // epsilon links inside frame t, to the fixed point
    if (e1 > e0 && a_lds) {
      auto relax = [&](int s, int d, float dl) -> int {
        const float x = __uint_as_float(A[d]);
        if (x < INFINITY) {
          float le = x + dl;
          if (le <= lbeam) {
            le = fmaxf(le, 0.f);
            const uint32_t k = __float_as_uint(le);
            // (a token with thousands of links -- silence, word ends -- serialises its ds_min's: only those that would
            // lower the value are issued; the plain read of one address by many lanes is a broadcast)
            if (k < A[s] && k < atomicMin(&A[s], k)) return 1;
          }
        }
        return 0;
      };
      for (int rounds = 0; rounds < kMaxEpsRounds; ++rounds) {
        int changed = 0;
        // One dependent LDS chain per link, with its early exits: most links of a round fail the first test (their destination
        // has no finite extra cost yet, or the sum leaves the lattice beam).  Measured and dropped in round 5
        // (profiles/r05_lat.txt): the same links as straight-line batches -- all gathers, all candidates, all compares, then
        // the ds_min's -- 7.3 us per frame for the rounds against 5.5 (the batches pay every LDS read the exits skip);
        // and, on top, a wave whose 64 links all enter ONE token (the word-loop state: a link per word) reducing them
        // with DPP moves into a single ds_min: 8.0 us (the `k < A[s]` guard already keeps most of them from being issued).
#pragma unroll
        for (int q = 0; q < kFinEps; ++q)
          if (es[q] >= 0) changed |= relax(es[q], ed[q], el[q]);
        for (int l = e0 + tid + kFinEps * kLatThreads; l < e1; l += kLatThreads) {
          const int4 r = V.lrec[l];
          changed |= relax(r.x - base, r.y - base, delta[l]);
        }
#ifdef PK2_LAT_FIN_DEBUG
        ++nrounds;
#endif
        if (!any_changed(changed)) break;
      }
    } else if (e1 > e0) {
      for (int rounds = 0; rounds < kMaxEpsRounds; ++rounds) {
        int changed = 0;
        for (int l = e0 + tid; l < e1; l += kLatThreads) {
          const int4 r = V.lrec[l];
          const float x = __uint_as_float(ld_coherent(&teu[r.y]));
          if (x < INFINITY) {
            float le = x + delta[l];
            if (le <= lbeam) {
              le = fmaxf(le, 0.f);
              const uint32_t k = __float_as_uint(le);
              if (k < atomicMin(&teu[r.x], k)) changed = 1;
            }
          }
        }
        if (!__syncthreads_or(changed)) break;
      }
    }
    FIN_T(0);
    // the extra costs of frame t are final
    if (a_lds)
      for (int i = tid; i < cnt; i += kLatThreads) teu[base + i] = A[i];
    FIN_T(1);
    if (t > 0) {
      const int pbase = sc.pb, pcnt = base - pbase;
      const bool b_lds = pcnt <= cap;
      load_eps(scn, pbase);              // the epsilon records of the frame before: requested now, used after the emitting links
      if (b_lds)
        for (int i = tid; i < pcnt; i += kLatThreads) B[i] = 0x7f800000u;
      fin_barrier(a_lds && b_lds);
      FIN_T(2);
      // emitting links t-1 -> t
      auto emit = [&](int s, int d, float dl) {
        const float x = __uint_as_float(a_lds ? A[d - base] : ld_coherent(&teu[d]));
        if (!(x < INFINITY)) return;
        const float le = x + dl;
        if (le <= lbeam) {
          const uint32_t k = __float_as_uint(fmaxf(le, 0.f));
          if (b_lds) { if (k < B[s - pbase]) atomicMin(&B[s - pbase], k); } else atomicMin(&teu[s], k);
        }
      };
      if (a_lds && b_lds) {
        // (round 5) straight-line batches: all gathers of the destination tokens' extra costs, all candidates, all compares
        // against the source tokens' current values, then the ds_min's that would lower one
        uint32_t xd[kFinEmit], kq[kFinEmit], cur[kFinEmit];
#pragma unroll
        for (int q = 0; q < kFinEmit; ++q) {
          const bool have = m0 + tid + q * kLatThreads < m1;
          xd[q] = have ? A[mr[q].y - base] : 0x7f800000u;
        }
#pragma unroll
        for (int q = 0; q < kFinEmit; ++q) {
          const float x = __uint_as_float(xd[q]), le = x + md[q];
          kq[q] = (m0 + tid + q * kLatThreads < m1 && x < INFINITY && le <= lbeam) ? __float_as_uint(fmaxf(le, 0.f)) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int q = 0; q < kFinEmit; ++q) cur[q] = kq[q] != 0xFFFFFFFFu ? B[mr[q].x - pbase] : 0u;
#pragma unroll
        for (int q = 0; q < kFinEmit; ++q)
          if (kq[q] < cur[q]) atomicMin(&B[mr[q].x - pbase], kq[q]);
      } else {
#pragma unroll
        for (int q = 0; q < kFinEmit; ++q)
          if (m0 + tid + q * kLatThreads < m1) emit(mr[q].x, mr[q].y, md[q]);
      }
      for (int l = m0 + tid + kFinEmit * kLatThreads; l < m1; l += kLatThreads) {
        const int2 r = *reinterpret_cast<const int2*>(&V.lrec[l]);
        emit(r.x, r.y, delta[l]);
      }
      fin_barrier(a_lds && b_lds);
      FIN_T(3);
      uint32_t* tmp = A; A = B; B = tmp;
      base = pbase; cnt = pcnt; a_lds = b_lds;
      sc = scn;
    }
  }
#ifdef PK2_LAT_FIN_DEBUG
  if (tid == 0) printf("finish utt %d T %d: us per frame: eps (loads + %.1f rounds) %.2f | write back %.2f | init %.2f | emitting %.2f\n", n, T,
                       (double)nrounds / (T + 1), ph[0] * 0.01 / (T + 1), ph[1] * 0.01 / (T + 1), ph[2] * 0.01 / (T + 1), ph[3] * 0.01 / (T + 1));
#endif
  if (tid == 0) {
    LatUtt* o = p.L.utt + n;
    o->status = kLatOk; o->n_tok = s_tok_end; o->n_link = s_link_end; o->any_final = anyf; o->best_cost = best_final;
  }
}

static StepGraphs g_lat_graphs;
static std::map<DevStream, StepCounter*> g_lat_counters;

// The kernels take the call's parameters BY VALUE: a pointer to a parameter block would put one more dependent
// memory round trip (~1 us) in front of each of the ~11 launches of a frame.  The values are baked into the
// captured graph, so graphs are cached per parameter content (a training loop decodes into the same workspace
// with the same options step after step; the per-utterance geometry lives in the workspace, not in the
// parameters).  At most kMaxLatGraphs parameter sets are kept; the least recently used one is dropped (after a
// stream synchronisation: its graphs may still be running).
constexpr size_t kMaxLatGraphs = 8;
struct LatGraphSlot { int id; uint64_t last_use; };
static std::map<std::string, LatGraphSlot> g_lat_keys;
static uint64_t g_lat_clock = 0;

// ---- host side of the persistent decoder ----
struct LatPersistScratch { LatTeamCtl* ctl = nullptr; unsigned* sticky = nullptr; };
static std::map<DevStream, LatPersistScratch> g_lat_persist;
static PerDevice<int> g_lat_persist_state_pd(-1);      // -1: not tried on this device yet, 1: verified, 0: did not come back complete -> launch per frame

static bool lat_persist_wanted(int team) {
  static const int mode = [] {
    const char* e = getenv("PK2_LAT_DECODER");
    return (e && strcmp(e, "frames") == 0) ? 0 : 1;
  }();
  if (!mode || g_lat_persist_state_pd.ref() == 0 || (team != 8 && team != 16 && team != 32)) return false;
  static PerDevice<int> cus_pd(-1); int& cus = cus_pd.ref();
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  return cus == 256;                      // 8 XCDs x 32 CUs: what the team formation assumes
}

static int lat_persist_launch(const DecodeParams& p, int N, int team, hipStream_t stream, bool* ran) {
  *ran = false;
  LatPersistScratch& sc = g_lat_persist[dev_stream(stream)];
  if (!sc.ctl) {
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(LatTeamCtl)));
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.sticky), sizeof(unsigned)));
    PK2_HIP(hipMemsetAsync(sc.sticky, 0, sizeof(unsigned), stream));
  }
  PK2_HIP(hipMemsetAsync(sc.ctl, 0, sizeof(LatTeamCtl), stream));
  // one team per XCD while the utterances fit (an utterance then has its XCD's L2 to itself), more when there are more
  const int tpx = std::max(1, std::min({32 / team, kLatTeamsPerXcd, (N + 7) / 8}));
  static const int inv_mode = [] { const char* e = getenv("PK2_LAT_INV"); return e ? atoi(e) : 1; }();
  static const int field_bits = [] { const char* e = getenv("PK2_LAT_FIELD_BITS"); return e ? atoi(e) : 23; }();
  const char* stall_env = getenv("PK2_LAT_TEST_STALL");         // (read per launch: a test switches it on for one call)
  const int test_stall = stall_env && atoi(stall_env) != 0 ? 1 : 0;
  hipLaunchKernelGGL(lat_frames_persist, dim3(256), dim3(kLatThreads), 0, stream, p, sc.ctl, N, team, tpx, inv_mode, field_bits,
                     test_stall);
  PK2_LAUNCH_CHECK();
  if (g_lat_persist_state_pd.ref() < 0) {          // first use on this device: every utterance done, nobody timed out?
    LatTeamCtl* h = new LatTeamCtl;
    hipError_t e = hipMemcpyAsync(h, sc.ctl, sizeof(LatTeamCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)N;
    delete h;
    if (e != hipSuccess) { set_error("lattice decode (persistent): %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_lat_persist_state_pd.ref() = ok ? 1 : 0;
    if (!ok) {                            // the caller decodes again, a launch per frame: give it clean state tables
      PK2_HIP(hipMemsetAsync(p.L.st_cost, 0xFF, sizeof(uint32_t) * (size_t)N * p.g.S, stream));
      PK2_HIP(hipMemsetAsync(p.L.st_tok, 0xFF, sizeof(int32_t) * (size_t)N * p.g.S, stream));
      return PK2_OK;
    }
  }
  PersistGuard guard;
  (void)persist_guard(&guard);
  hipLaunchKernelGGL(lat_persist_check, dim3(1), dim3(64), 0, stream, p, sc.ctl, N, sc.sticky, guard.dev, guard.host_dev);
  PK2_LAUNCH_CHECK();
  *ran = true;
  return PK2_OK;
}

// 1: the persistent decoder is in use on this device, 0: it is not (disabled, or it failed its first launch), -1: not tried;
// *abort_flag: some launch since start-up was marked "not decoded" by its check kernel.
int lattice_persist_status(unsigned* abort_flag) {
  unsigned any = 0;
  for (auto& kv : g_lat_persist) {
    unsigned st = 0;
    if (kv.second.sticky && hipMemcpy(&st, kv.second.sticky, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) any |= st;
  }
  *abort_flag = any;
  return g_lat_persist_state_pd.ref();
}

static int lattice_decode_frames_graphs(const DecodeParams& p, int N, int Tmax, int team, hipStream_t stream);
static int lattice_finish_and_prune(const DecodeParams& p, int N, hipStream_t stream);

int lattice_decode_frames(const DecodeParams& p, int N, int Tmax, int team, hipStream_t stream) {
  bool ran = false;
  if (lat_persist_wanted(team)) {
    int rc = lat_persist_launch(p, N, team, stream, &ran);
    if (rc) return rc;
  }
  if (!ran) {
    int rc = lattice_decode_frames_graphs(p, N, Tmax, team, stream);
    if (rc) return rc;
  }
  return lattice_finish_and_prune(p, N, stream);
}

static int lattice_decode_frames_graphs(const DecodeParams& p, int N, int Tmax, int team, hipStream_t stream) {
  StepCounter*& counter = g_lat_counters[dev_stream(stream)];
  if (!counter) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&counter), sizeof(StepCounter)));
  const StepCounter* cnt = counter;
  hipLaunchKernelGGL(lat_frames_init, dim3(N), dim3(64), 0, stream, p, team);
  PK2_LAUNCH_CHECK();
  // the log-likelihood tensor is the one pointer that moves from call to call: the frame kernels take it from the
  // per-utterance state (written by lat_frames_init), not from the baked parameters
  DecodeParams pk = p;
  pk.loglikes = nullptr; pk.seq_stride = 0; pk.frame_stride = 0;
  std::string raw(reinterpret_cast<const char*>(&pk), sizeof(pk));
  raw += "|" + std::to_string(N) + "|" + std::to_string(team) + "|" + std::to_string((uintptr_t)stream);
  auto it = g_lat_keys.find(raw);
  char key[64];
  if (it == g_lat_keys.end()) {
    int id = (int)g_lat_keys.size();
    if (g_lat_keys.size() >= kMaxLatGraphs) {
      auto old = g_lat_keys.begin();
      for (auto q = g_lat_keys.begin(); q != g_lat_keys.end(); ++q)
        if (q->second.last_use < old->second.last_use) old = q;
      PK2_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>((uintptr_t)std::stoull(old->first.substr(old->first.rfind('|') + 1)))));
      snprintf(key, sizeof(key), "lat_frames_%d", old->second.id);
      g_lat_graphs.erase(key);
      id = old->second.id;
      g_lat_keys.erase(old);
    }
    it = g_lat_keys.emplace(raw, LatGraphSlot{id, 0}).first;
  }
  it->second.last_use = ++g_lat_clock;
  snprintf(key, sizeof(key), "lat_frames_%d", it->second.id);
  const dim3 one(1, N), all(team, N), thr(kLatThreads);
  int rc = g_lat_graphs.run(key, Tmax + 1, counter, stream, [&](hipStream_t s, int j) {
    hipLaunchKernelGGL(lat_frames_cutoff, one, thr, 0, s, pk, cnt, j);
    hipLaunchKernelGGL(lat_frames_list, all, thr, 0, s, pk, cnt, j);
    hipLaunchKernelGGL(lat_frames_expand, all, thr, 0, s, pk, cnt, j);
    hipLaunchKernelGGL(lat_frames_round0, all, thr, 0, s, pk, cnt, j);
    for (int r = 1; r <= kLatEpsRounds; ++r) hipLaunchKernelGGL(lat_frames_round, all, thr, 0, s, pk, cnt, j, r);
    hipLaunchKernelGGL(lat_frames_tail, one, thr, 0, s, pk, cnt, j);
    hipLaunchKernelGGL(lat_frames_close, all, thr, 0, s, pk, cnt, j);
  });
  return rc;
}

static int lattice_finish_and_prune(const DecodeParams& p, int N, hipStream_t stream) {
  const dim3 thr(kLatThreads);
  // lattice-beam pruning: the per-link constants in parallel, then the serial pass with two frames' extra costs in LDS
  constexpr int kFinCap = 19456;                   // tokens of a frame the LDS arrays hold (2 x 76 KB)
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lat_frames_finish), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kFinCap * (int)sizeof(uint32_t)));
    attr = true;
  }
  hipLaunchKernelGGL(lat_link_delta, dim3(256, N), dim3(256), 0, stream, p);
  const char* cap_env = getenv("PK2_LAT_FIN_CAP");       // (test hook: 0 sends every utterance down the global-memory pass)
  const int cap = cap_env ? std::max(0, std::min(kFinCap, atoi(cap_env))) : kFinCap;
  hipLaunchKernelGGL(lat_frames_finish, dim3(N), thr, 2 * kFinCap * sizeof(uint32_t), stream, p, cap);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

}  // namespace pk2
