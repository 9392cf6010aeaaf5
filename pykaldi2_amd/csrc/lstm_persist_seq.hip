// Persistent LSTM recurrence, one (sequence, direction) per XCD (gfx950, H = 512).
//
// lstm_persist.hip runs a direction of a layer on one XCD for ALL sequences of the minibatch: at batch 4 that keeps 2 of
// the 8 XCDs busy, and a step pays for 4 batch rows on the 4x4x1 MFMA (128 instructions per wave: 0.85 of a step's 2.2 us).
// The recurrences of different sequences do not depend on each other, so here every (sequence, direction) pair is its own
// recurrence on its own XCD -- 4 sequences x 2 directions = the 8 XCDs of the part; more pairs queue up behind them:
//  * 32 workgroups (one per CU of the XCD; teams form by arrival order, pairs are handed out from a queue), workgroup r
//    owns hidden units 16r..16r+15, its 64 gate rows of W_hh (128 KB) stay in VGPRs;
//  * one batch row makes the recurrent product a matrix-VECTOR product: plain FMAs.  Forward: a lane owns a quarter of one
//    gate row (128 k's; the four quarters are added with two DPP moves), 128 FMAs per lane and step.  Backward: a lane
//    owns two columns k of the workgroup's 64 rows, 128 FMAs per lane, no cross-lane reduction at all;
//  * h_t (2 KB) / the partial products (a 2 KB mailbox per workgroup) are exchanged through the XCD's own L2 exactly as in
//    lstm_persist.hip: agent-scope stores over a NaN sentinel, polled with L1-bypassing 16-byte loads, no flags.
// Tensors and their layouts are those of lstm_persist.hip (y pre-filled with the sentinel, gates / cells kept for the
// backward pass), so the two implementations are interchangeable behind pk2_lstm_layer_fwd / _bwd.
// Replaces the same cuDNN RNN (reference models/lstm.py:49-58).
#include <algorithm>
#include <cstdlib>
#include <map>

#include "common.h"
#include "lstm_persist.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kSeqSentinel = 0x7fc0dead;
constexpr int kSH = 512;
constexpr int kSWgs = 32;                   // workgroups of a team = CUs of an XCD; 16 hidden units each
constexpr int kSeqTeams = 8;                // teams per XCD the control block has room for
constexpr int kSeqMaxTasks = 64;
constexpr long long kSeqSpinTicks = 1000LL * 1000 * 100;     // 1 s of the 100 MHz wall clock
constexpr int kSeqMailFloats = 3 * kSWgs * kSWgs * 16;       // per team: [step % 3][reader][writer][16 units]

struct SeqCtl {
  unsigned arrive[8];
  unsigned next_task;
  unsigned abort;
  unsigned done;
  unsigned pad[5];
  struct Team { unsigned task[kSeqMaxTasks + 1]; unsigned bar; unsigned pad[62]; } team[8][kSeqTeams];
};

__device__ __forceinline__ unsigned seq_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0x7;
}
__device__ __forceinline__ u32x4 seq_load16(const float* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// (the s_nop: ">8-byte VMEM store data followed by a VALU write of the same VGPRs" hazard, invisible to the compiler
// inside an asm statement)
__device__ __forceinline__ void seq_store16(float* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void seq_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned seq_load_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool seq_has_sentinel(u32x4 v) {
  return v.x == kSeqSentinel || v.y == kSeqSentinel || v.z == kSeqSentinel || v.w == kSeqSentinel;
}
__device__ __forceinline__ float seq_sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float seq_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
template <int CTRL>
__device__ __forceinline__ float seq_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Workgroup barrier that waits for the LDS traffic only: __syncthreads() also waits for every outstanding global store
// and load of the thread (s_waitcnt vmcnt(0)), i.e. for the HBM traffic of the previous step.
__device__ __forceinline__ void seq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A poll that keeps failing reads the wall clock every 256 rounds; after 1 s (or when somebody else gave up) it raises
// the abort flag.
struct SeqSpin {
  SeqCtl* ctl; long long t0; unsigned n;
  __device__ __forceinline__ explicit SeqSpin(SeqCtl* c) : ctl(c), t0(0), n(0) {}
  __device__ __forceinline__ bool expired() {
    if ((++n & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > kSeqSpinTicks || seq_load_u(&ctl->abort)) {
      __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    return false;
  }
};

// Team formation and the queue of (sequence, direction) pairs, shared by both kernels.
struct SeqRole { int rank; SeqCtl::Team* team; int team_index; };
__device__ __forceinline__ bool seq_register(SeqCtl* ctl, int* s_i, SeqRole* role) {
  if (threadIdx.x == 0) {
    const unsigned xcd = seq_xcc_id();
    const unsigned slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_i[0] = (int)(slot % kSWgs); s_i[1] = (int)(slot / kSWgs); s_i[2] = (int)xcd; s_i[3] = 0;
  }
  __syncthreads();
  if (s_i[1] >= kSeqTeams) return false;
  role->rank = s_i[0];
  role->team = &ctl->team[s_i[2]][s_i[1]];
  role->team_index = s_i[2] * kSeqTeams + s_i[1];
  return true;
}
// Next pair of this team (-1: none left / abort).  s_i[3] = abort flag of the workgroup, s_i[4] = the task.
__device__ __forceinline__ int seq_next_task(SeqCtl* ctl, const SeqRole& role, int iter, int ntasks, int* s_i) {
  if (threadIdx.x == 0) {
    unsigned k;
    if (role.rank == 0) {
      k = __hip_atomic_fetch_add(&ctl->next_task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&role.team->task[iter], k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      SeqSpin spin(ctl);
      unsigned k1;
      while ((k1 = seq_load_u(&role.team->task[iter])) == 0u) {
        if (spin.expired()) { s_i[3] = 1; k1 = 1u << 30; break; }
      }
      k = k1 - 1u;
    }
    s_i[4] = (int)k;
  }
  __syncthreads();
  const int k = s_i[4];
  return (s_i[3] || k >= ntasks) ? -1 : k;
}
__device__ __forceinline__ bool seq_team_barrier(SeqCtl* ctl, const SeqRole& role, unsigned* nbar, int* s_i) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned target = (unsigned)kSWgs * ++*nbar;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&role.team->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SeqSpin spin(ctl);
    while (seq_load_u(&role.team->bar) < target) {
      if (spin.expired()) { s_i[3] = 1; break; }
    }
  }
  __syncthreads();
  return s_i[3] == 0;
}

struct SeqFwdParams {
  const float* gx;    // [T][B][D*4H]
  const float* whh;   // [D][4H][H]
  const float* bhh;   // [D][4H] or null
  float* y;           // [T][B][D*H], pre-filled with the sentinel
  float* gates;       // [D][T][B][4H]
  float* cells;       // [D][T][B][H]
  int B, T, D;
};

__global__ void __launch_bounds__(256) lstm_fwd_seq(SeqFwdParams p, SeqCtl* ctl) {
  constexpr int H = kSH;
  constexpr int kPhasePitch = 132;          // 128 + 4: the four k-quarters the lanes of a quad read start 16 bytes apart mod the banks
  __shared__ __attribute__((aligned(16))) float hs[2][4 * kPhasePitch];
  __shared__ int s_i[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  SeqRole role;
  if (!seq_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  // lane roles: quad = one gate row, lane % 4 = its k-quarter; the 4 gates of a unit sit in one 16-lane DPP row
  const int ul = lane >> 4, g = (lane >> 2) & 3, q = lane & 3;
  const int gu = 16 * rank + 4 * w + ul;                 // hidden unit of this lane's row
  const bool unit_lane = (lane & 15) == 0;
  const bool poll_lane = (lane & 15) >= 1 && (lane & 15) <= 8;
  const int gran = w * 32 + (lane >> 4) * 8 + ((lane & 15) - 1);      // granule of h this lane polls (128 per step)
  const size_t yrow = (size_t)D * H;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kSeqMaxTasks; ++iter) {
    const int task = seq_next_task(ctl, role, iter, B * D, s_i);
    if (task < 0) return;
    const int b = task / D, d = task % D;
    float wa[128];                                       // W_hh[d][g*H + gu][128 q + i]
    {
      const float* wrow = p.whh + ((size_t)d * 4 * H + (size_t)g * H + gu) * H + 128 * q;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + i * 4);
        wa[i * 4] = v[0]; wa[i * 4 + 1] = v[1]; wa[i * 4 + 2] = v[2]; wa[i * 4 + 3] = v[3];
      }
    }
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (unit_lane && p.bhh) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bias[k] = p.bhh[(size_t)d * 4 * H + (size_t)k * H + gu];
    }
    float cstate = 0.f;
    bool timed_out = false;
    float gxn[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_gx = [&](int step_) {
      if (unit_lane && step_ < T) {
        const int t_ = d == 0 ? step_ : T - 1 - step_;
        const float* gxr = p.gx + ((size_t)t_ * B + b) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + gu;
#pragma unroll
        for (int k = 0; k < 4; ++k) gxn[k] = gxr[(size_t)k * H];
      }
    };
    load_gx(0);
    float gxm[4];                      // the input projections of step s+1; gxn: those of step s+2, in flight
#pragma unroll
    for (int k = 0; k < 4; ++k) gxm[k] = gxn[k];
    load_gx(1);
    SeqSpin spin(ctl);
    for (int step = 0; step < T; ++step) {
      const int t = d == 0 ? step : T - 1 - step;
      const int tp = d == 0 ? t - 1 : t + 1;
      const int buf = step & 1;
      float pre[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { pre[k] = gxm[k] + bias[k]; gxm[k] = gxn[k]; }
      // ---- gather h_{t-1} of this (sequence, direction): 512 floats = 128 granules of 16 bytes -----------------------
      // (by lanes 1..8 of every 16-lane row: a poll waits for ALL of the thread's outstanding memory operations, and the
      // unit lanes -- lane % 16 == 0 -- still have the previous step's gate / cell stores and the gx prefetch in flight)
      if (step > 0 && poll_lane) {
        const float* src = p.y + ((size_t)tp * B + b) * yrow + (size_t)d * H + 4 * gran;
        u32x4 v = seq_load16(src);
        while (seq_has_sentinel(v)) {
          if (spin.expired()) { timed_out = true; break; }
          v = seq_load16(src);
        }
        *reinterpret_cast<u32x4*>(&hs[buf][(gran >> 5) * kPhasePitch + 4 * (gran & 31)]) = v;
      }
      if (timed_out) s_i[3] = 1;
      seq_lds_barrier();
      if (s_i[3]) return;
      load_gx(step + 2);
      // ---- recurrent product: this lane's quarter of its gate row ------------------------------------------------------
      float s = 0.f;
      if (step > 0) {
        const float* hq = &hs[buf][q * kPhasePitch];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(hq + 4 * i);
          a0 = fmaf(wa[4 * i], hv[0], a0);
          a1 = fmaf(wa[4 * i + 1], hv[1], a1);
          a2 = fmaf(wa[4 * i + 2], hv[2], a2);
          a3 = fmaf(wa[4 * i + 3], hv[3], a3);
        }
        s = (a0 + a1) + (a2 + a3);
        s += seq_dpp<0xB1>(s);        // quad_perm [1,0,3,2]
        s += seq_dpp<0x4E>(s);        // quad_perm [2,3,0,1]
      }
      // the four gates of a unit: lanes +0, +4, +8, +12 of its 16-lane row
      const float sf = seq_dpp<0x104>(s), sg = seq_dpp<0x108>(s), so = seq_dpp<0x10C>(s);
      if (unit_lane) {
        const float ig = seq_sig(pre[0] + s), fg = seq_sig(pre[1] + sf), gg = seq_tanh(pre[2] + sg), og = seq_sig(pre[3] + so);
        cstate = fg * cstate + ig * gg;
        const float h = og * seq_tanh(cstate);
        seq_store(p.y + ((size_t)t * B + b) * yrow + (size_t)d * H + gu, h);
        p.cells[(((size_t)d * T + t) * B + b) * H + gu] = cstate;
        float* gr = p.gates + (((size_t)d * T + t) * B + b) * 4 * H + gu;
        gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
      }
    }
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!seq_team_barrier(ctl, role, &nbar, s_i)) return;     // (the LDS buffers and the team's pace are per pair)
  }
}

struct SeqBwdParams {
  const float* dy;     // [T][B][D*H]
  const float* whh;    // [D][4H][H]
  const float* gates;  // [D][T][B][4H]
  const float* cells;  // [D][T][B][H]
  float* dgx;          // [T][B][D*4H]
  float* mail;         // [teams][3][32 readers][32 writers][16 units], all sentinel
  int B, T, D;
};

__global__ void __launch_bounds__(256) lstm_bwd_seq(SeqBwdParams p, SeqCtl* ctl) {
  constexpr int H = kSH, G4 = 4 * kSH;
  __shared__ __attribute__((aligned(16))) float pl[kSWgs][16 + 4];    // gathered partials [writer][unit]
  __shared__ __attribute__((aligned(16))) float dgl[64];              // this workgroup's dgates [gate * 16 + unit]
  __shared__ int s_i[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  SeqRole role;
  if (!seq_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  float* mail0 = p.mail + (size_t)role.team_index * kSeqMailFloats;
  const u32x4 sent = {kSeqSentinel, kSeqSentinel, kSeqSentinel, kSeqSentinel};
  const int unit = 16 * rank + tid;                     // pointwise role of threads 0..15
  const bool pw = tid < 16;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kSeqMaxTasks; ++iter) {
    const int task = seq_next_task(ctl, role, iter, B * D, s_i);
    if (task < 0) return;
    const int b = task / D, d = task % D;
    // W_hh[d][(r/16)*H + 16*rank + r%16][k] for the 64 own rows r and the lane's two columns k = 128 w + lane (+ 64)
    float wb[128];
    {
      const float* wbase = p.whh + (size_t)d * G4 * H + 128 * w + lane;
#pragma unroll
      for (int r = 0; r < 64; ++r) {
        const float* row = wbase + ((size_t)(r >> 4) * H + 16 * rank + (r & 15)) * H;
        wb[2 * r] = row[0];
        wb[2 * r + 1] = row[64];
      }
    }
    float dcarry = 0.f;
    bool timed_out = false;
    float n_dy = 0.f, n_i = 0.f, n_f = 0.f, n_g = 0.f, n_o = 0.f, n_c = 0.f, n_cp = 0.f;          // operands of step s+2 (in flight)
    float m_dy = 0.f, m_i = 0.f, m_f = 0.f, m_g = 0.f, m_o = 0.f, m_c = 0.f, m_cp = 0.f;          // operands of step s+1
    auto load_pw = [&](int step_) {
      if (pw && step_ < T) {
        const int fs = T - 1 - step_;
        const int t_ = d == 0 ? fs : T - 1 - fs;
        const int tp_ = d == 0 ? t_ - 1 : t_ + 1;
        n_dy = p.dy[((size_t)t_ * B + b) * ((size_t)D * H) + (size_t)d * H + unit];
        const float* gr = p.gates + (((size_t)d * T + t_) * B + b) * G4 + unit;
        n_i = gr[0]; n_f = gr[(size_t)H]; n_g = gr[(size_t)2 * H]; n_o = gr[(size_t)3 * H];
        n_c = p.cells[(((size_t)d * T + t_) * B + b) * H + unit];
        n_cp = fs == 0 ? 0.f : p.cells[(((size_t)d * T + tp_) * B + b) * H + unit];
      }
    };
    load_pw(0);
    m_dy = n_dy; m_i = n_i; m_f = n_f; m_g = n_g; m_o = n_o; m_c = n_c; m_cp = n_cp;
    load_pw(1);
    SeqSpin spin(ctl);
    for (int step = 0; step < T; ++step) {
      const int fstep = T - 1 - step;
      const int t = d == 0 ? fstep : T - 1 - fstep;
      // (loaded two steps ahead: a step is shorter than a round trip to memory; the rotation below only touches values whose
      // loads were issued a whole step ago)
      const float c_dy = m_dy, c_i = m_i, c_f = m_f, c_g = m_g, c_o = m_o, c_c = m_c, c_cp = m_cp;
      m_dy = n_dy; m_i = n_i; m_f = n_f; m_g = n_g; m_o = n_o; m_c = n_c; m_cp = n_cp;
      // ---- gather the 32 partials of d h for the own 16 units (written by the peers during the previous step) ----------
      float rec = 0.f;
      if (step > 0) {
        // (threads 128..255 poll: the pointwise threads 0..15 have prefetches and gradient stores in flight, which a poll
        // -- it waits for all of a thread's outstanding memory operations -- would wait for)
        if (tid >= 128) {
          const int gt = tid - 128;
          float* src = mail0 + ((size_t)(step % 3) * kSWgs + rank) * (kSWgs * 16) + 4 * gt;      // [writer = gt/4][4 units]
          u32x4 v = seq_load16(src);
          while (seq_has_sentinel(v)) {
            if (spin.expired()) { timed_out = true; break; }
            v = seq_load16(src);
          }
          seq_store16(src, sent);               // free again (ordered before this step's own stores by the wait below)
          *reinterpret_cast<u32x4*>(&pl[gt >> 2][4 * (gt & 3)]) = v;
        }
        if (timed_out) s_i[3] = 1;
        __syncthreads();
        if (s_i[3]) {            // loud failure: the gradient of this layer turns NaN
          if (tid == 0) p.dgx[(size_t)d * G4 + 16 * rank] = __int_as_float(0x7fc00000);
          return;
        }
        if (pw) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int k = 0; k < kSWgs; k += 4) { s0 += pl[k][tid]; s1 += pl[k + 1][tid]; s2 += pl[k + 2][tid]; s3 += pl[k + 3][tid]; }
          rec = (s0 + s1) + (s2 + s3);
        }
      }
      // ---- gate derivatives of the own units ------------------------------------------------------------------------------
      float dgi = 0.f, dgf = 0.f, dgg = 0.f, dgo = 0.f;
      if (pw) {
        const float dh = c_dy + rec;
        const float tc = seq_tanh(c_c);
        const float dcv = dcarry + dh * c_o * (1.f - tc * tc);
        dcarry = dcv * c_f;
        dgi = dcv * c_g * c_i * (1.f - c_i);
        dgf = dcv * c_cp * c_f * (1.f - c_f);
        dgg = dcv * c_i * (1.f - c_g * c_g);
        dgo = dh * tc * c_o * (1.f - c_o);
        dgl[tid] = dgi; dgl[16 + tid] = dgf; dgl[32 + tid] = dgg; dgl[48 + tid] = dgo;
      }
      auto store_dgx = [&]() {        // (to HBM; nothing waits for these)
        if (pw) {
          float* o = p.dgx + ((size_t)t * B + b) * ((size_t)D * G4) + (size_t)d * G4 + unit;
          o[0] = dgi; o[(size_t)H] = dgf; o[(size_t)2 * H] = dgg; o[(size_t)3 * H] = dgo;
        }
      };
      if (step == T - 1) { store_dgx(); break; }
      // Three mailbox buffers: the one read (and reset) in step s is written again by the peers in their step s+2, after they
      // have read this workgroup's partials of step s+2 -- which go out behind the barrier of step s+1, and the polling
      // threads reach that barrier only after their reset stores have been acknowledged (a poll waits for the thread's
      // outstanding stores).  So no wait for the resets here (with two buffers it cost an L2 round trip per step).
      seq_lds_barrier();
      store_dgx();
      load_pw(step + 2);             // the pointwise operands of the step after next
      // ---- own 64 rows of dgates x own W_hh rows: the lane's two columns ---------------------------------------------------
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 16; ++r4) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(&dgl[4 * r4]);
        a0 = fmaf(dv[0], wb[8 * r4], a0);     b0 = fmaf(dv[0], wb[8 * r4 + 1], b0);
        a1 = fmaf(dv[1], wb[8 * r4 + 2], a1); b1 = fmaf(dv[1], wb[8 * r4 + 3], b1);
        a0 = fmaf(dv[2], wb[8 * r4 + 4], a0); b0 = fmaf(dv[2], wb[8 * r4 + 5], b0);
        a1 = fmaf(dv[3], wb[8 * r4 + 6], a1); b1 = fmaf(dv[3], wb[8 * r4 + 7], b1);
      }
      // peer k/16 reads [writer = rank][unit k%16] from its mailbox of the next step
      float* box = mail0 + (size_t)((step + 1) % 3) * kSWgs * (kSWgs * 16);
      const int k0 = 128 * w + lane, k1 = k0 + 64;
      seq_store(box + ((size_t)(k0 >> 4) * kSWgs + rank) * 16 + (k0 & 15), a0 + a1);
      seq_store(box + ((size_t)(k1 >> 4) * kSWgs + rank) * 16 + (k1 & 15), b0 + b1);
    }
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!seq_team_barrier(ctl, role, &nbar, s_i)) return;     // nobody writes a mailbox of the next pair before everybody has read the last of this one
  }
}

// A launch in which a poll timed out (or a pair was never finished) must not pass for a result: the forward
// pass keeps its NaN sentinels in y, but the backward pass would leave dgx -- freshly allocated memory -- partly
// unwritten.  One workgroup after every persistent launch turns the launch's output into NaN in that case and raises a
// sticky flag (the control block itself is cleared by the next launch) that pk2_lstm_persist_status reports.
__global__ void lstm_seq_check(const SeqCtl* ctl, unsigned pairs, float* out, size_t n, unsigned* sticky) {
  if (ctl->abort == 0u && ctl->done == pairs) return;
  if (threadIdx.x == 0) *sticky = 1u;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct SeqScratch { SeqCtl* ctl = nullptr; float* mail = nullptr; unsigned* sticky = nullptr; };
static std::map<hipStream_t, SeqScratch> g_seq_scratch;
static int g_seq_state = -1;             // -1 untested, 0 unusable, 1 verified on this device

static int seq_scratch(hipStream_t stream, SeqScratch** out) {
  SeqScratch& sc = g_seq_scratch[stream];
  if (!sc.ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(SeqCtl)));
  if (!sc.mail) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.mail), (size_t)8 * kSeqTeams * kSeqMailFloats * sizeof(float)));
  if (!sc.sticky) {
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.sticky), sizeof(unsigned)));
    PK2_HIP(hipMemsetAsync(sc.sticky, 0, sizeof(unsigned), stream));
  }
  *out = &sc;
  return PK2_OK;
}

// Teams per XCD of a launch: with more than 8 pairs a second team per XCD takes its own pairs from the queue at the same
// time (two workgroups per CU; a recurrence is latency-bound, two of them interleave almost for free).
static int seq_teams(int pairs) {
  const char* env = getenv("PK2_LSTM_SEQ_TEAMS");
  const int most = env ? std::max(1, std::min(atoi(env), kSeqTeams)) : 2;
  return std::max(1, std::min(most, (pairs + 7) / 8));
}

bool lstm_seq_wanted(int B, int H, int D) {
  const char* env = getenv("PK2_LSTM_SEQ");
  if (env && atoi(env) == 0) return false;
  // (up to 4 pairs per XCD one after the other; larger batches are better served by the batched step kernels)
  if (g_seq_state == 0 || H != kSH || B < 1 || B * D > 32 || (D != 1 && D != 2)) return false;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  return cus == 8 * kSWgs;
}

int lstm_fwd_seq_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                        float* gates, float* cells, hipStream_t stream, bool* ran) {
  *ran = false;
  SeqScratch* sc = nullptr;
  int rc = seq_scratch(stream, &sc);
  if (rc) return rc;
  PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(SeqCtl), stream));
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(y), (int)kSeqSentinel, (size_t)T * B * D * H, stream));
  SeqFwdParams p{gx, whh, bhh, y, gates, cells, B, T, D};
  hipLaunchKernelGGL(lstm_fwd_seq, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl);
  PK2_LAUNCH_CHECK();
  if (g_seq_state < 0) {                 // first use on this device: every pair done, nobody timed out?
    SeqCtl* h = new SeqCtl;
    hipError_t e = hipMemcpyAsync(h, sc->ctl, sizeof(SeqCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)(B * D);
    delete h;
    if (e != hipSuccess) { set_error("lstm_seq: %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_seq_state = ok ? 1 : 0;
    if (!ok) return PK2_OK;              // the caller falls back (and keeps doing so)
  }
  hipLaunchKernelGGL(lstm_seq_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)(B * D), y, (size_t)T * B * D * H, sc->sticky);
  *ran = true;
  return PK2_OK;
}

int lstm_bwd_seq_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                        int D, float* dgx, hipStream_t stream, bool* ran) {
  *ran = false;
  if (g_seq_state != 1) return PK2_OK;   // the forward pass verifies the device first
  SeqScratch* sc = nullptr;
  int rc = seq_scratch(stream, &sc);
  if (rc) return rc;
  PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(SeqCtl), stream));
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sc->mail), (int)kSeqSentinel, (size_t)8 * kSeqTeams * kSeqMailFloats, stream));
  SeqBwdParams p{dy, whh, gates, cells, dgx, sc->mail, B, T, D};
  hipLaunchKernelGGL(lstm_bwd_seq, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl);
  hipLaunchKernelGGL(lstm_seq_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)(B * D), dgx, (size_t)T * B * D * 4 * H, sc->sticky);
  PK2_LAUNCH_CHECK();
  *ran = true;
  return PK2_OK;
}

int lstm_seq_status(unsigned* abort_flag) {
  unsigned any = 0;
  for (auto& kv : g_seq_scratch) {
    if (!kv.second.ctl) continue;
    SeqCtl* h = new SeqCtl;
    hipError_t e = hipMemcpy(h, kv.second.ctl, sizeof(SeqCtl), hipMemcpyDeviceToHost);
    if (e == hipSuccess) any |= h->abort;
    delete h;
    unsigned st = 0;
    if (kv.second.sticky && hipMemcpy(&st, kv.second.sticky, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) any |= st;
  }
  *abort_flag = any;
  return PK2_OK;
}

}  // namespace pk2
