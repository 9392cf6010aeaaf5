// Persistent denominator recursion: the alpha and beta chains of a whole minibatch in ONE launch (gfx950).
//
// The launch-per-frame kernels of chain_den.hip spend a frame's 18 us on a dependent launch, a cold re-stream of the
// 1 M arc records (the L2s are invalidated at kernel boundaries) and ~1 M random gathers from L2.  Here a recursion
// (one sequence, one direction) lives on ONE XCD for all of its frames:
//  * the 32 workgroups of a team (one per CU, 512 threads = 2 waves per SIMD with 256 VGPRs each) own contiguous row
//    ranges of the arc ordering; a thread keeps its 64 arc slots (probability + 16-bit gathered index: 96 registers)
//    for the whole call -- the arcs are read from memory once per call instead of once per frame;
//  * the frame's state vector (alpha[t, .], or the backward recursion's w[t+1, .] per virtual state) sits in LDS
//    (4 bytes per state: 120 KB for the 30 k states of the BASELINE graph), filled by LDS-DMA (global_load_lds_dwordx4,
//    no VGPR round trip); an arc then costs one ds_read_b32 and one FMA;
//  * row sums are the atomic-free lane-private / wave-scan / wave-carry scheme of den_step_sx with 64 slots per lane;
//    the host pads rows so that they end only at every 2nd / 4th / 8th slot (where the graph leaves room) and deals a
//    row's arcs to a thread's slots so that a half wave's gathers spread over the LDS banks (chain_graph.hip);
//  * a frame's results are exchanged through the XCD's own L2: two state-vector buffers written with agent-scope stores,
//    and four words per rank and frame (partial sums, "slice ready") that a rank publishes AFTER its slice has reached
//    L2 -- one wave per workgroup polls them, NaN-sentinel guarded, reset by their owner two frames ahead.  No grid
//    barrier, nothing crosses to another XCD.  Both recursions need ONE exchange per frame: the backward one's scaling
//    needs the frame's sums over all states before it can publish anything, but those sums are linear in the vector the
//    frame gathers, so the producers of that vector deliver them with it (see run_bwd).  The per-frame history (alpha,
//    per-occupancy-state alpha, btilde', partial sums) goes out with plain stores, after everything another workgroup
//    waits for, in the NG = 1 layouts of chain_den.hip, whose parallel passes (exp tables, scales, occupancies) run
//    unchanged around this kernel;
//  * 2 N recursions are handed to the teams from a queue, longest first (8 XCDs = 4 sequences x 2 directions at once).
// Teams form by arrival order per XCD (XCC_ID register); polls time out after 1 s and raise an abort flag (the first
// launch on a device is verified, a later failure poisons the objective with NaN instead of passing for a result).
// Measured on the bench graph (S = 30 k, A = 1.01 M, 4 sequences, Tmax = 589): 5.0 ms per denominator call against
// 11.0 ms for the frame kernels; a frame takes ~8 us in either direction.
// Replaces the same DenominatorComputation as chain_den.hip (reference ops/ops.py:265, bin/train_chain.py:202).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <vector>

#include "den_persist_dev.h"

namespace pk2 {

// The workgroup's LDS, carved out of the dynamic allocation (the recursions are separate functions: everything they
// share with the kernel lives here, and every pointer is derived from the LDS symbol so that the accesses stay ds_*).
extern __shared__ __attribute__((aligned(16))) float den_persist_smem[];
struct Lds {
  float* table;    // [rpad]  the frame's gather table
  float* acc;      // [cap]   row sums
  float* xown;     // [cap]   forward: x[t, own rows]
  float* leak;     // [cap]   forward: leaky-HMM constant of own rows; backward: weight of an own virtual state in lB
  float* aux;      // [cap]   backward: its weight in lU
  float* red;      // [2 * kPW]
  float* tot;      // [4]     a frame's partial sums added up (written by the polling wave)
  float* wcarry;   // [kPW]
  int* wcrow;      // [kPW]
  int* rb;         // [kPR+1] ring entries of a rank: [rb[r], rb[r+1])
  int* abort;      // + rank, team, xcd, task
};
constexpr int kLdsTailInts = kPW + (kPR + 1) + 8;
__device__ __forceinline__ Lds carve_lds(int rpad, int cap) {
  Lds L;
  L.table = den_persist_smem; L.acc = L.table + rpad; L.xown = L.acc + cap; L.leak = L.xown + cap; L.aux = L.leak + cap;
  L.red = L.aux + cap; L.tot = L.red + 2 * kPW; L.wcarry = L.tot + 4;
  L.wcrow = reinterpret_cast<int*>(L.wcarry + kPW); L.rb = L.wcrow + kPW; L.abort = L.rb + (kPR + 1);
  return L;
}

// Row sums of the thread's kPK register-resident arcs over the LDS table: complete rows are stored by the lane, the
// piece before the first row end gets the carry of the earlier lanes (segmented wave scan), the open tail of a wave goes
// to wcarry and is added to its row by row_sum.  ESTEP (chain_internal.h: HostPersist::estep): rows end only after slots
// ESTEP-1 (mod ESTEP), so the row-end code exists at 64/ESTEP places.  The gathers of the next 8 slots are in flight while
// 8 slots are accumulated.
template <int ESTEP>
__device__ __forceinline__ void arc_rows(const float (&prob)[kPK], const uint32_t (&idx2)[kPK / 2], uint64_t ends, int frow,
                                         const Lds& L) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float sum = 0.f;
  int c = frow;
  // the row-end bits of the possible places, most significant first: `m + m` shifts the next one into the carry (one
  // v_add_co per place instead of a 64-bit shift, a mask and a compare)
  uint64_t packed = 0;
#pragma unroll
  for (int k = 0; k < kPK / ESTEP; ++k) packed |= ((ends >> (k * ESTEP + ESTEP - 1)) & 1ull) << k;
  unsigned m[2] = {__builtin_bitreverse32((unsigned)packed), __builtin_bitreverse32((unsigned)(packed >> 32))};
  auto gather = [&](int j0, float (&a)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t pk = idx2[(j0 + j) >> 1];
      const uint32_t byte_off = (j & 1) ? (pk >> 16) << 2 : (pk & 0xffffu) << 2;
      a[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(L.table) + byte_off);
    }
  };
  float a[2][8];
  gather(0, a[0]);
#pragma unroll
  for (int g = 0; g < kPK / 8; ++g) {
    if (g + 1 < kPK / 8) gather(8 * (g + 1), a[(g + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jj = 8 * g + j;
      sum = fmaf(a[g & 1][j], prob[jj], sum);
      if ((jj + 1) % ESTEP == 0) {
        unsigned& mm = m[(jj / ESTEP) >> 5];
        if (__builtin_add_overflow(mm, mm, &mm)) { L.acc[c] = sum; ++c; sum = 0.f; }
      }
    }
  }
  float x[1] = {sum};
  int fl = ends != 0ull ? 1 : 0;
  seg_scan_step<1, 0x111, 0xf>(x, fl);
  seg_scan_step<1, 0x112, 0xf>(x, fl);
  seg_scan_step<1, 0x114, 0xf>(x, fl);
  seg_scan_step<1, 0x118, 0xf>(x, fl);
  seg_scan_step<1, 0x142, 0xa>(x, fl);
  seg_scan_step<1, 0x143, 0xc>(x, fl);
  const float cin = dpp_f<0x138, 0xf>(x[0]);     // wave_shr:1 -- lane 0 receives 0
  if (ends != 0ull) L.acc[frow] += cin;            // the lane's own first row end (stored above by this lane)
  if (lane == 63) L.wcarry[w] = x[0];
}
__device__ __forceinline__ void arc_rows_any(int estep, const float (&prob)[kPK], const uint32_t (&idx2)[kPK / 2], uint64_t ends,
                                             int frow, const Lds& L) {
  switch (estep) {
    case 8: arc_rows<8>(prob, idx2, ends, frow, L); break;
    case 4: arc_rows<4>(prob, idx2, ends, frow, L); break;
    case 2: arc_rows<2>(prob, idx2, ends, frow, L); break;
    default: arc_rows<1>(prob, idx2, ends, frow, L); break;
  }
}

// Value of rank-local row r after the arcs: the stored sum plus the carry-outs of the waves whose open tail belongs to
// it (a row longer than a wave's 4096 slots collects several).  The targets (wcr, at most kPW rows of the workgroup) are
// per-task constants in scalar registers; a frame's carries are read from LDS once per thread.
struct Carries { int row[kPW]; float val[kPW]; };
__device__ __forceinline__ void load_carries(const Lds& L, Carries& c) {
#pragma unroll
  for (int k = 0; k < kPW; ++k) c.val[k] = c.row[k] >= 0 ? L.wcarry[k] : 0.f;
}
__device__ __forceinline__ float row_sum(const Lds& L, const Carries& c, int r) {
  float v = L.acc[r];
#pragma unroll
  for (int k = 0; k < kPW; ++k)
    if (c.row[k] >= 0 && c.row[k] == r) v += c.val[k];
  return v;
}

template <int NW>
__device__ __forceinline__ void block_sum2_1(float& u, float& v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u = wave_sum_dpp(u); v = wave_sum_dpp(v);
  __syncthreads();
  if (lane == 0) { red[w] = u; red[NW + w] = v; }
  __syncthreads();
  float a = 0.f, b = 0.f;
  for (int k = 0; k < NW; ++k) { a += red[k]; b += red[NW + k]; }
  u = a; v = b;
}

// alpha recursion of sequence g (T frames): den_fwd_frame_sx<1> frame by frame.
__device__ __noinline__ void run_fwd(CParams* pp_, DenPersistCtl* ctl_, DenPersistCtl::Team* team_,
                                     unsigned* nbar, int g_, int T_, int rank_, float* ring_, float* pring_) {
  CParams* pp = uni(pp_);
  DenPersistCtl* ctl = uni(ctl_); DenPersistCtl::Team* team = uni(team_);
  const int g = uni(g_), T = uni(T_), rank = uni(rank_);
  ring_ = uni(ring_); pring_ = uni(pring_);
  CParams& p = *pp;
  CDenParams& d = p.d;
  const Lds L = carve_lds(p.rpad, p.cap);
  gfloat* ring = G(ring_); gfloat* pring = G(pring_);
  const int tid = threadIdx.x;
  const int S = d.S, V = d.V, Vo = d.Vo;
  const bool sep = d.alphav != d.alpha;
  float prob[kPK]; uint32_t idx2[kPK / 2];
#pragma unroll
  for (int j = 0; j < kPK; ++j) prob[j] = p.fwd.prob[((size_t)rank * kPK + j) * kPT + tid];
#pragma unroll
  for (int j = 0; j < kPK / 2; ++j) idx2[j] = p.fwd.idx2[((size_t)rank * (kPK / 2) + j) * kPT + tid];
  const uint64_t ends = p.fwd.ends[(size_t)rank * kPT + tid];
  const int frow = p.fwd.first_row[(size_t)rank * kPT + tid];
  const int row0 = p.fwd.row_begin[rank], nrows = p.fwd.row_begin[rank + 1] - row0;
  const int g0 = p.fwd.grp_begin[rank], ngrp = p.fwd.grp_begin[rank + 1] - g0;
  for (int r = tid; r < nrows; r += kPT) L.leak[r] = p.fwd.row_leak[row0 + r];
  Carries cr;
#pragma unroll
  for (int k = 0; k < kPW; ++k) cr.row[k] = p.fwd.wcrow[rank * kPW + k];
  int st_lo[kPSPT], st_hi[kPSPT], st_o[kPSPT]; float st_pl[kPSPT], st_pi[kPSPT]; bool st_ok[kPSPT];
#pragma unroll
  for (int i = 0; i < kPSPT; ++i) {
    const int r = tid + i * kPT;
    st_ok[i] = r < ngrp;
    st_lo[i] = st_hi[i] = st_o[i] = 0; st_pl[i] = st_pi[i] = 0.f;
    if (st_ok[i]) {
      const int dd = g0 + r;
      st_lo[i] = d.voff[dd] - row0; st_hi[i] = d.voff[dd + 1] - row0; st_o[i] = d.ooff[dd];
      st_pl[i] = d.loop_prob[dd]; st_pi[i] = d.pi[dd];
    }
  }
  // own words of the three frame slots start as "not yet written"
  if (tid < 3 * kPWords) st_agent(pring + ((tid / kPWords) * kPR + rank) * kPWords + tid % kPWords, __uint_as_float(kRingSentinel));
  if (!team_barrier(ctl, team, nbar, L.abort)) return;

  const size_t f0 = (size_t)g * (d.Tmax + 1);
  cgfloat* xv_g = G(p.xv) + (size_t)g * d.Tmax * V;
  cgfloat* xl_g = G(d.xl) + (size_t)g * d.Tmax * S;
  float xr[kPSPT], xlr[kPSPT];
  auto prefetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      const int r = tid + i * kPT;
      xr[i] = r < nrows ? xv_g[(size_t)t * V + row0 + r] : 0.f;
      xlr[i] = st_ok[i] ? xl_g[(size_t)t * S + g0 + r] : 0.f;
    }
  };
  prefetch(0);
  Spin spin(ctl);
  DP_T0();
  for (int t = 0; t < T; ++t) {
    float as;
    DP_TL(0, 0);
    if (t == 0) {
      for (int idx = tid; idx < S; idx += kPT) L.table[idx] = G(d.pi)[idx];
      as = d.pi_sum;
    } else {
      poll_words(pring + (size_t)(t % 3) * kPR * kPWords, 0, 1, spin, L);
      __syncthreads();
      DP_TL(0, 1);
      if (*L.abort) return;
      as = L.tot[0];
      dma_table(ring + (size_t)(t & 1) * p.rpad, S, L.table);
    }
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      const int r = tid + i * kPT;
      if (r < nrows) L.xown[r] = xr[i];
    }
    // the word this rank will publish two frames from now must read "not yet written" by then: reset here, long before
    // the stores it has to precede (the waits of the epilogue cover it)
    const bool publish = t + 1 < T;
    if (tid == 0 && publish) st_agent(word_of(pring, t + 2, rank, 0), __uint_as_float(kRingSentinel));
    dma_wait();
    DP_T(0);
    DP_TL(0, 2);
    __syncthreads();
    DP_T(1);
    DP_TL(0, 3);
    if (rank == 0 && tid == 0) G(d.asum)[f0 + t] = as;
    const float lk = d.leaky * as, inv_as = 1.0f / as;
    float own_a[kPSPT];
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) own_a[i] = st_ok[i] ? L.table[g0 + tid + i * kPT] : 0.f;
    arc_rows_any(p.fwd.estep, prob, idx2, ends, frow, L);
    DP_T(2);
    DP_TL(0, 4);
    __syncthreads();
    load_carries(L, cr);
    DP_T(3);
    // rows are virtual states (dst, pdf); a thread per real state sums its rows and adds the peeled self-loop.  First
    // only what the other workgroups wait for: the ring entries, then (once they are in L2) the partial sum.
    gfloat* ring_n = ring + (size_t)((t + 1) & 1) * p.rpad;
    float outv[kPSPT], loopv[kPSPT], loc = 0.f, unused = 0.f;
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      outv[i] = 0.f; loopv[i] = 0.f;
      if (!st_ok[i]) continue;
      float sum = 0.f;
      for (int q = st_lo[i]; q < st_hi[i]; ++q) {
        const float v = (row_sum(L, cr, q) + lk * L.leak[q]) * L.xown[q] * inv_as;
        L.acc[q] = v;          // (the row belongs to this thread alone: kept for the history stores below)
        sum += v;
      }
      if (st_pl[i] > 0.f) { loopv[i] = (own_a[i] + lk * st_pi[i]) * st_pl[i] * xlr[i] * inv_as; sum += loopv[i]; }
      if (publish) ring_store(ring_n + g0 + tid + i * kPT, sum);
      outv[i] = sum;
      loc += sum;
    }
    DP_T(4);
    DP_TL(0, 5);
    wait_stores();        // this rank's slice of frame t+1 (and the reset above) is in L2 before its partial sum says so
    block_sum2_1<kPW>(loc, unused, L.red);
    if (tid == 0 && publish) st_agent(word_of(pring, t + 1, rank, 0), loc);
    DP_T(5);
    DP_TL(0, 6);
    // the history the parallel passes read (nobody waits for these stores)
    gfloat* alpha_n = G(d.alpha) + (f0 + t + 1) * (size_t)S;
    gfloat* alphav_n = G(d.alphav) + (f0 + t + 1) * (size_t)Vo;
    if (tid == 0) G(d.apart)[(f0 + t + 1) * kPR + rank] = loc;
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      if (!st_ok[i]) continue;
      alpha_n[g0 + tid + i * kPT] = outv[i];
      if (sep) {
        for (int q = st_lo[i]; q < st_hi[i]; ++q) alphav_n[st_o[i] + q - st_lo[i]] = L.acc[q];
        if (st_pl[i] > 0.f) alphav_n[st_o[i] + st_hi[i] - st_lo[i]] = loopv[i];
      }
    }
    if (publish) prefetch(t + 1);
    DP_T(6);
  }
  DP_FLUSH(0);
}

// btilde' recursion of sequence g: den_beta_frame_sx<1> frame by frame, T-1 down to 0.  What a frame gathers is
//   w[t+1][v] = x[t, pdf(v)] * betahat[t+1, state(v)],  betahat[t+1, s] = btilde'[t+1, s] / c[t+1] + leaky term of t+1,
// and it is the PRODUCER of btilde'[t+1, .] that scales its slice (it needs x only for its own ~V/32 virtual states; a
// consumer-side transform would have every workgroup read all of x[t, .]: 120 KB more per frame and CU, and 64 more live
// registers per thread).  The scaling needs the frame's sums over ALL states, lB[t] = sum_s pi[s] btilde'[t, s] and
// lU[t] = sum_s btilde'[t, s] -- which would cost a second exchange per frame (partial sums out, totals back, only then
// the slices).  But btilde'[t, .] is linear in the vector the frame gathers:
//   lB[t] = sum_v w[t+1][v] * (sum of pi[src] * prob over the arcs entering v)  +  sum_s pi[s] loop_s x_loop[t, s] betahat[t+1, s]
// (lU alike without pi), and the per-virtual-state weights are constants of the graph (the forward layout's row_leak /
// row_psum).  So the producer of w[t+1] publishes, WITH its slice, its share of the sums frame t is going to compute: a
// frame knows c[t] before its first arc, scales its rows as soon as they are summed, and one exchange per frame is left.
__device__ __noinline__ void run_bwd(CParams* pp_, DenPersistCtl* ctl_, DenPersistCtl::Team* team_,
                                     unsigned* nbar, int g_, int T_, int rank_, float* ring_, float* pring_) {
  CParams* pp = uni(pp_);
  DenPersistCtl* ctl = uni(ctl_); DenPersistCtl::Team* team = uni(team_);
  const int g = uni(g_), T = uni(T_), rank = uni(rank_);
  ring_ = uni(ring_); pring_ = uni(pring_);
  CParams& p = *pp;
  CDenParams& d = p.d;
  const Lds L = carve_lds(p.rpad, p.cap);
  gfloat* ring = G(ring_); gfloat* pring = G(pring_);
  const int tid = threadIdx.x;
  const int S = d.S, V = d.V;
  float prob[kPK]; uint32_t idx2[kPK / 2];
#pragma unroll
  for (int j = 0; j < kPK; ++j) prob[j] = p.bwd.prob[((size_t)rank * kPK + j) * kPT + tid];
#pragma unroll
  for (int j = 0; j < kPK / 2; ++j) idx2[j] = p.bwd.idx2[((size_t)rank * (kPK / 2) + j) * kPT + tid];
  const uint64_t ends = p.bwd.ends[(size_t)rank * kPT + tid];
  const int frow = p.bwd.first_row[(size_t)rank * kPT + tid];
  const int row0 = p.bwd.row_begin[rank], nrows = p.bwd.row_begin[rank + 1] - row0;
  const int vfirst = d.voff[row0], nvirt = d.voff[row0 + nrows] - vfirst;     // own virtual states: a contiguous range
  Carries cr;
#pragma unroll
  for (int k = 0; k < kPW; ++k) cr.row[k] = p.bwd.wcrow[rank * kPW + k];
  int st_v0[kPSPT], st_v1[kPSPT]; float st_pl[kPSPT], st_pi[kPSPT]; bool st_ok[kPSPT];
#pragma unroll
  for (int i = 0; i < kPSPT; ++i) {
    const int r = tid + i * kPT;
    st_ok[i] = r < nrows;
    st_v0[i] = st_v1[i] = 0; st_pl[i] = st_pi[i] = 0.f;
    if (st_ok[i]) {
      const int s = row0 + r;
      st_v0[i] = d.voff[s] - vfirst; st_v1[i] = d.voff[s + 1] - vfirst; st_pl[i] = d.loop_prob[s]; st_pi[i] = d.pi[s];
    }
  }
  // what an own virtual state contributes to the next frame's sums (virtual states are the forward layout's rows)
  for (int r = tid; r < nvirt; r += kPT) { L.leak[r] = p.fwd.row_leak[vfirst + r]; L.aux[r] = p.fwd.row_psum[vfirst + r]; }
  if (tid < 3 * kPWords) st_agent(pring + ((tid / kPWords) * kPR + rank) * kPWords + tid % kPWords, __uint_as_float(kRingSentinel));
  if (!team_barrier(ctl, team, nbar, L.abort)) return;

  const size_t f0 = (size_t)g * (d.Tmax + 1);
  cgfloat* xv_g = G(p.xv) + (size_t)g * d.Tmax * V;
  cgfloat* xl_g = G(d.xl) + (size_t)g * d.Tmax * S;
  // x of own loops at the frame being computed (xl_cur) and at the one after it (xl_prev), x of own virtual states at the
  // latter (xw), beta-hat of own states at the frame computed last (bh)
  float xl_cur[kPSPT], xl_prev[kPSPT], xw[kPSPT], bh[kPSPT];
  const float cst_last = 1.0f / d.pi_sum + d.leaky;
#pragma unroll
  for (int i = 0; i < kPSPT; ++i) { bh[i] = cst_last; xl_cur[i] = 0.f; }
  auto prefetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      const int r = tid + i * kPT;
      xl_prev[i] = st_ok[i] ? xl_g[(size_t)t * S + row0 + r] : 0.f;
      xw[i] = r < nvirt ? xv_g[(size_t)t * V + vfirst + r] : 0.f;
    }
  };
  auto stage_x = [&]() {
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      const int r = tid + i * kPT;
      if (r < nvirt) L.xown[r] = xw[i];
    }
  };
  // Publishes this rank's slice of w[t] = x[t-1, .] * bh (xown = x[t-1, own virtual states], xl_prev = x_loop[t-1, own
  // states]) and, once it is in L2, its share of lB[t-1] / lU[t-1]: words 0 and 1 of frame slot t.
  auto emit = [&](int t) {
    gfloat* ring_n = ring + (size_t)(t & 1) * p.rpad + vfirst;
    float pB = 0.f, pU = 0.f;
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      if (!st_ok[i]) continue;
      for (int q = st_v0[i]; q < st_v1[i]; ++q) {
        const float w = L.xown[q] * bh[i];
        ring_store(ring_n + q, w);
        pB = fmaf(w, L.leak[q], pB); pU = fmaf(w, L.aux[q], pU);
      }
      if (st_pl[i] > 0.f) { const float lp = st_pl[i] * xl_prev[i] * bh[i]; pB = fmaf(st_pi[i], lp, pB); pU += lp; }
    }
    wait_stores();        // the slice (and the reset of the words two frames on) is in L2 before the sums say so
    block_sum2_1<kPW>(pB, pU, L.red);
    if (tid < 2) st_agent(word_of(pring, t, rank, tid), tid == 0 ? pB : pU);
    if (tid == 0) {       // the parallel passes recompute c[t-1] from the same shares
      G(d.bpart)[((f0 + t - 1) * kPR + rank) * 2] = pB;
      G(d.bpart)[((f0 + t - 1) * kPR + rank) * 2 + 1] = pU;
    }
  };
  // w[T] = x[T-1, .] * (1 / sum(pi) + leaky)
  prefetch(T - 1);
  stage_x();
  __syncthreads();
  emit(T);
#pragma unroll
  for (int i = 0; i < kPSPT; ++i) xl_cur[i] = xl_prev[i];
  if (T >= 2) prefetch(T - 2);
  Spin spin(ctl);
  DP_T0();
  for (int t = T - 1; t >= 0; --t) {
    DP_TL(1, 0);
    poll_words(pring + (size_t)((t + 1) % 3) * kPR * kPWords, 0, 2, spin, L);
    __syncthreads();
    DP_TL(1, 1);
    if (*L.abort) return;
    const float lB = L.tot[0], lU = L.tot[1];        // sums of btilde'[t, .], known before it is computed
    dma_table(ring + (size_t)((t + 1) & 1) * p.rpad, V, L.table);
    const bool publish = t > 0;
    if (publish) stage_x();
    // the words this rank will publish two frames from now must read "not yet written" by then: reset here, long before
    // the stores they have to precede (the waits of emit cover it)
    if (tid < 2 && publish) st_agent(word_of(pring, t + 2, rank, tid), __uint_as_float(kRingSentinel));   // (t-1) % 3 == (t+2) % 3
    dma_wait();
    DP_T(0);
    DP_TL(1, 2);
    __syncthreads();
    DP_T(1);
    DP_TL(1, 3);
    arc_rows_any(p.bwd.estep, prob, idx2, ends, frow, L);
    DP_T(2);
    DP_TL(1, 4);
    __syncthreads();
    load_carries(L, cr);
    DP_T(3);
    // rows are source states: btilde'[t, s] = row + peeled loop, then beta-hat[t, s] with the sums received above
    const float cu = lB + d.wu * lU;
    const float inv_c = cu > 0.f ? 1.0f / cu : 0.f;
    const float lkr = d.leaky * lB * inv_c;
    float vs[kPSPT];
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      vs[i] = 0.f;
      if (!st_ok[i]) continue;
      float v = row_sum(L, cr, tid + i * kPT);
      if (st_pl[i] > 0.f) v += st_pl[i] * xl_cur[i] * bh[i];
      vs[i] = v;
      bh[i] = v * inv_c + lkr;
    }
    DP_T(4);
    DP_TL(1, 5);
    if (publish) emit(t);
    DP_T(5);
    DP_TL(1, 6);
    // history for the parallel passes, after everything another workgroup waits for (the occupancy pass reads btilde' of
    // a state from its first virtual state: ovirt)
    gfloat* bx_t = G(d.beta) + (f0 + t) * (size_t)V * 2;
#pragma unroll
    for (int i = 0; i < kPSPT; ++i) {
      if (st_ok[i]) bx_t[(size_t)(vfirst + st_v0[i]) * 2] = vs[i];
      xl_cur[i] = xl_prev[i];
    }
    if (t >= 2) prefetch(t - 2);
    DP_T(6);
  }
  DP_FLUSH(1);
}

// (the parameter block lives in device memory: the two recursions are separate functions -- separate register
// allocations around their 128 arc registers -- and read it through scalar loads)
__global__ void __launch_bounds__(kPT) den_persist_kernel(const DenPersistParams* __restrict__ pp_, DenPersistCtl* ctl) {
  CParams* pp = (CParams*)pp_;
  CParams& p = *pp;
  const Lds L = carve_lds(p.rpad, p.cap);
  int& s_abort = L.abort[0];
  int& s_rank = L.abort[1]; int& s_team = L.abort[2]; int& s_xcd = L.abort[3]; int& s_task = L.abort[4];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_abort = 0;
    const unsigned xcd = den_xcc_id();
    const unsigned slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_rank = (int)(slot % kPR); s_team = (int)(slot / kPR); s_xcd = (int)xcd;
  }
  __syncthreads();
  if (s_team >= kMaxTeams) return;
  const int rank = s_rank;
  DenPersistCtl::Team* team = &ctl->team[s_xcd][s_team];
  const size_t ti = (size_t)s_xcd * kMaxTeams + s_team;
  float* ring = p.ring + ti * 2 * p.rpad;
  float* pring = p.pring + ti * 3 * kPR * kPWords;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kMaxTasks; ++iter) {
    if (tid == 0) {
      unsigned k;
      if (rank == 0) {
        k = __hip_atomic_fetch_add(&ctl->next_task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->task[iter], k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        Spin spin(ctl);
        unsigned k1;
        while ((k1 = ld_agent_u(G(&team->task[iter]))) == 0u) {
          if (spin.expired()) { s_abort = 1; k1 = 1u << 30; break; }
        }
        k = k1 - 1u;
      }
      s_task = (int)k;
    }
    __syncthreads();
    const int k = s_task;
    if (s_abort || k >= p.ntasks) return;
    const int g = p.task_seq[k], T = p.d.lengths[g];
    if (!team_barrier(ctl, team, &nbar, &s_abort)) return;      // everybody has left the previous recursion
#ifndef PK2_DP_NOFWD
    if (p.task_dir[k] == 0) run_fwd(pp, ctl, team, &nbar, g, T, rank, ring, pring);
#endif
#ifndef PK2_DP_NOBWD
    if (p.task_dir[k] != 0) run_bwd(pp, ctl, team, &nbar, g, T, rank, ring, pring);
#endif
    __syncthreads();
    if (s_abort) return;
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A launch that did not complete every recursion (a poll timed out) must not look like a result.
__global__ void den_persist_check(const DenPersistCtl* ctl, int ntasks, float* den_lp, int n, unsigned* guard_dev, unsigned* guard_host) {
  if (ctl->abort != 0u || ctl->done != (unsigned)ntasks) {
    if (threadIdx.x == 0) persist_guard_raise(guard_dev, guard_host);
    for (int i = threadIdx.x; i < n; i += blockDim.x) den_lp[i] = __uint_as_float(0x7fc00000u);
  }
}

// ----------------------------------------------------------------------------------------
// host
// ----------------------------------------------------------------------------------------
static PerDevice<int> g_den_persist_state_pd(-1);     // -1: not verified yet, 0: unusable on this device, 1: verified
struct DenPersistParams;
struct DenPersistScratch { DenPersistParams* params = nullptr; DenPersistCtl* ctl = nullptr; float* ring = nullptr; float* pring = nullptr; int rpad = 0; int ntasks = 0; };
static std::map<DevStream, DenPersistScratch> g_den_scratch;

static int den_rpad(const pk2_den_graph* g) { return (std::max(g->S, g->V) + 63) / 64 * 64; }
static int den_cap(const pk2_den_graph* g) {
  return (std::max({g->h_pfwd.max_rows, g->h_pfwd.max_groups, g->h_pbwd.max_rows, g->h_pbwd.max_groups, 1}) + 3) / 4 * 4;
}
size_t den_persist_lds_bytes(const pk2_den_graph* g) {
  return ((size_t)den_rpad(g) + 4 * (size_t)den_cap(g) + 3 * kPW + 4 + kLdsTailInts) * sizeof(float);
}

bool den_persist_fits(const pk2_den_graph* g) {
  return g->h_pfwd.ok && g->h_pbwd.ok && den_persist_lds_bytes(g) <= kDenPersistMaxLds;
}

bool den_persist_wanted(const pk2_den_graph* g, int N) {
  const char* env = getenv("PK2_DEN_PERSIST");
  if (env && atoi(env) == 0) return false;
  if (g_den_persist_state_pd.ref() == 0 || !den_persist_fits(g) || !den_use_sx(g)) return false;
  if (N < 1 || 2 * N > kMaxTasks) return false;
  static PerDevice<int> cus_pd(-1); int& cus = cus_pd.ref();
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  return cus == 8 * kPR;
}

int den_persist_launch(pk2_den_graph* g, const DenParams& dp, const float* xv, const int32_t* lengths_host, int N,
                       hipStream_t stream, bool* ran) {
  *ran = false;
  DenPersistScratch& sc = g_den_scratch[dev_stream(stream)];
  const int rpad = den_rpad(g);
  if (!sc.ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(DenPersistCtl)));
  if (!sc.params) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.params), sizeof(DenPersistParams)));
  if (!sc.pring) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.pring), (size_t)8 * kMaxTeams * 3 * kPR * kPWords * sizeof(float)));
  if (sc.rpad < rpad) {
    if (sc.ring) PK2_HIP(hipFree(sc.ring));
    sc.ring = nullptr; sc.rpad = 0;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ring), (size_t)8 * kMaxTeams * 2 * rpad * sizeof(float)));
    sc.rpad = rpad;
  }
  DenPersistParams p;
  p.d = dp;
  p.fwd = g->pfwd; p.bwd = g->pbwd;
  p.xv = xv; p.ring = sc.ring; p.pring = sc.pring;
  p.rpad = rpad; p.cap = den_cap(g);
  // recursions, longest first
  std::vector<std::pair<int, int>> order;    // (-T, task id = 2 n + dir)
  for (int n = 0; n < N; ++n)
    if (lengths_host[n] > 0) { order.push_back({-lengths_host[n], 2 * n}); order.push_back({-lengths_host[n], 2 * n + 1}); }
  std::sort(order.begin(), order.end());
  p.ntasks = (int)order.size();
  for (int k = 0; k < kMaxTasks; ++k) { p.task_seq[k] = 0; p.task_dir[k] = 0; }
  for (int k = 0; k < p.ntasks; ++k) { p.task_seq[k] = (short)(order[k].second >> 1); p.task_dir[k] = (unsigned char)(order[k].second & 1); }
  sc.ntasks = 0;
  if (p.ntasks == 0) { *ran = true; return PK2_OK; }
  const size_t lds = den_persist_lds_bytes(g);
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr = true;
  }
  PK2_HIP(hipMemsetAsync(sc.ctl, 0, sizeof(DenPersistCtl), stream));
  hipLaunchKernelGGL(param_block_store<DenPersistParams>, dim3(1), dim3(1), 0, stream, p, sc.params);
  hipLaunchKernelGGL(den_persist_kernel, dim3(8 * kPR), dim3(kPT), lds, stream, sc.params, sc.ctl);
#ifdef PK2_DP_PROFILE
  { int tot = 0; for (int n = 0; n < N; ++n) tot += lengths_host[n];
    hipLaunchKernelGGL(dp_prof_print, dim3(1), dim3(1), 0, stream, std::max(1, tot / 4), 1); }   // (rank 0 of every team adds up)
#endif
  PK2_LAUNCH_CHECK();
  if (g_den_persist_state_pd.ref() < 0) {     // first use on this device: every recursion done, nobody timed out?
    DenPersistCtl* h = new DenPersistCtl;
    hipError_t e = hipMemcpyAsync(h, sc.ctl, sizeof(DenPersistCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)p.ntasks;
    delete h;
    if (e != hipSuccess) { set_error("den_persist: %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_den_persist_state_pd.ref() = ok ? 1 : 0;
    if (!ok) return PK2_OK;
  }
  sc.ntasks = p.ntasks;
  *ran = true;
  return PK2_OK;
}

void den_persist_check_launch(float* den_lp, int N, hipStream_t stream) {
  const DenPersistScratch& sc = g_den_scratch[dev_stream(stream)];
  PersistGuard guard;
  (void)persist_guard(&guard);
  if (sc.ctl && sc.ntasks > 0) hipLaunchKernelGGL(den_persist_check, dim3(1), dim3(64), 0, stream, sc.ctl, sc.ntasks, den_lp, N, guard.dev, guard.host_dev);
}

}  // namespace pk2
