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
#include "persist_guard.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kSeqSentinel = 0x7fc0dead;
constexpr int kSH = 512;
constexpr int kSWgs = 32;                   // workgroups of a team = CUs of an XCD; 16 hidden units each
constexpr int kSeqTeams = 8;                // teams per XCD the control block has room for
constexpr int kSeqMaxTasks = 64;
constexpr long long kSeqSpinTicks = 1000LL * 1000 * 100;     // 1 s of the 100 MHz wall clock
// per team: [step % depth][reader][writer][16 units].  The round-2 kernel uses three slots; the round-4 kernel, whose
// hand-over stores are plain (XCD-local) ones, walks all eight: a slot's reset by its reader then has seven steps and as
// many dependent exchanges to land before a writer stores to the slot again.
constexpr int kSeqMailDepth = 8;
constexpr int kSeqMailFloats = kSeqMailDepth * kSWgs * kSWgs * 16;

struct SeqCtl {
  unsigned arrive[8];
  unsigned next_task;
  unsigned abort;
  unsigned done;
  unsigned exited;        // workgroups that have left the kernel (round 6: the last one out runs the check)
  unsigned pad[4];
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
__device__ __forceinline__ void seq_store16_mode(float* p, u32x4 v, int mode) {
  if (mode == 1) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void seq_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The hand-over stores of the round-4 kernels by cache policy (experiment switch; 0 = agent scope as everywhere else).
__device__ __forceinline__ void seq_store_mode(float* p, float v, int mode) {
  if (mode == 1) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else if (mode == 2) asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
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

// -DPK2_SEQ_PROFILE: phase timers of wave 0 of rank 0 of every team on its first pair (shader clock, s_memtime; the
// wall clock brackets the whole recurrence so that clocks convert to time), printed per team at the end of the pair.
#ifdef PK2_SEQ_PROFILE
#define SQ_NPH 8
#define SQ_T0() long long sq_acc_[SQ_NPH] = {0, 0, 0, 0, 0, 0, 0, 0}; long long sq_last_ = clock64(); const long long sq_w0_ = wall_clock64(); const long long sq_c0_ = sq_last_
#define SQ_T(k) do { const long long n_ = clock64(); sq_acc_[k] += n_ - sq_last_; sq_last_ = n_; } while (0)
#define SQ_PRINT(name, labels, steps) do { if (tid == 0 && rank == 0 && iter == 0) { const long long wt_ = wall_clock64() - sq_w0_, ct_ = clock64() - sq_c0_; \
    printf("%s xcd %d, %d steps, %.1f ns per step (wall), %.0f clocks per step; clocks per step %s: %lld | %lld | %lld | %lld | %lld | %lld | %lld | %lld\n", name, s_i[2], steps, \
           10.0 * (double)wt_ / (steps), (double)ct_ / (steps), labels, sq_acc_[0] / (steps), sq_acc_[1] / (steps), sq_acc_[2] / (steps), sq_acc_[3] / (steps), sq_acc_[4] / (steps), \
           sq_acc_[5] / (steps), sq_acc_[6] / (steps), sq_acc_[7] / (steps)); } } while (0)
#else
#define SQ_T0() do { } while (0)
#define SQ_T(k) do { } while (0)
#define SQ_PRINT(name, labels, steps) do { } while (0)
#endif

// A poll that keeps failing reads the wall clock every 256 rounds; after 1 s (or when somebody else gave up) it raises
// the abort flag.
struct SeqSpin {
  SeqCtl* ctl; long long t0; unsigned n; long long limit;
  __device__ __forceinline__ explicit SeqSpin(SeqCtl* c) : ctl(c), t0(0), n(0), limit(kSeqSpinTicks) {}
  __device__ __forceinline__ bool expired() {
    if ((++n & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > limit || seq_load_u(&ctl->abort)) {
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
  role->team_index = s_i[1] * 8 + s_i[2];       // (team-major: the teams a launch uses are the first 8 * teams mailboxes)
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
  int pre_sleep, loop_sleep;   // round-4 kernels: s_sleep units (64 clocks) before the first poll of a step / between polls
  int store_mode;
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
    SQ_T0();
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
      SQ_T(0);
      if (timed_out) s_i[3] = 1;
      seq_lds_barrier();
      SQ_T(1);
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
#ifdef PK2_SEQ_PROFILE
      asm volatile("" : "+v"(s));
#endif
      SQ_T(2);
      if (unit_lane) {
        const float ig = seq_sig(pre[0] + s), fg = seq_sig(pre[1] + sf), gg = seq_tanh(pre[2] + sg), og = seq_sig(pre[3] + so);
        cstate = fg * cstate + ig * gg;
        const float h = og * seq_tanh(cstate);
        seq_store(p.y + ((size_t)t * B + b) * yrow + (size_t)d * H + gu, h);
        p.cells[(((size_t)d * T + t) * B + b) * H + gu] = cstate;
        float* gr = p.gates + (((size_t)d * T + t) * B + b) * 4 * H + gu;
        gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
      }
      SQ_T(3);
    }
    SQ_PRINT("lstm_fwd_seq", "(poll | lds + barrier | product | gates + stores)", T);
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
  int pre_sleep, loop_sleep, store_mode;
  float* dbias_ih;     // round-4 kernel: [D][4H] += sum over frames and sequences of d gx (null: not wanted), likewise dbias_hh
  float* dbias_hh;
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
    SQ_T0();
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
        SQ_T(0);
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
      SQ_T(1);
      if (step == T - 1) { store_dgx(); break; }
      // Three mailbox buffers: the one read (and reset) in step s is written again by the peers in their step s+2, after they
      // have read this workgroup's partials of step s+2 -- which go out behind the barrier of step s+1, and the polling
      // threads reach that barrier only after their reset stores have been acknowledged (a poll waits for the thread's
      // outstanding stores).  So no wait for the resets here (with two buffers it cost an L2 round trip per step).
      seq_lds_barrier();
      SQ_T(2);
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
      SQ_T(3);
    }
    SQ_PRINT("lstm_bwd_seq", "(mailbox poll by waves 2, 3 + barrier | partial sums + gate derivatives | lds barrier | product + mailbox stores)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!seq_team_barrier(ctl, role, &nbar, s_i)) return;     // nobody writes a mailbox of the next pair before everybody has read the last of this one
  }
}

// ---- Round 4: the same recurrences with the step's critical path cut down (lstm_fwd_seq2 / lstm_bwd_seq2) ---------------
// What the phase timers (-DPK2_SEQ_PROFILE, profiles/r04_seq_phases.txt) charged a step of the kernels above besides the
// round trip through L2, and what changed (forward 1.47 -> 0.97 us, backward 1.79 -> 1.14 us per step, T = 589, B = 4):
//  * The hand-over stores were agent-scope (sc1) stores.  On this multi-XCD part an agent-scope store is written THROUGH
//    the XCD's L2 to the fabric; a team lives on one XCD, whose L2 is the coherence point of its 32 CUs, so plain stores
//    (which the write-through vector L1 forwards to L2, and which vmcnt acknowledges there) are enough as long as the
//    readers poll with L1-bypassing loads: backward 1.95 -> 1.22 us per step by that alone (its 2 KB of scattered partial
//    sums per workgroup and step), forward 1.09 -> 0.98.  (PK2_SEQ_STORE_MODE=0 restores sc1 for A/B.)  The data is
//    polled on itself over a sentinel, so no ordering between stores is assumed; the backward mailboxes are a ring of
//    eight step slots, so that a slot's reset has seven steps to land before the slot is written again.
//  * `s_waitcnt vmcnt(0)` counts the WAVE's operations, not the lane's: a poll waits for every vector-memory operation the
//    wave has in flight.  Forward: waves 0, 1 poll (a granule per lane) and have nothing else in flight but their h
//    stores; wave 2 prefetches the input projections of the whole workgroup two steps ahead into an LDS ring, wave 3
//    writes gate activations and cells (handed over in LDS) to HBM a step later -- both behind their h stores.
//    Backward: wave 0 polls; wave 1 prefetches the forward pass's values, turns them into the per-unit factors of the
//    gate derivatives (everything that does not depend on the incoming d h, tanh(c) included) two steps ahead in LDS,
//    and writes d gx out -- behind its share of the partial sums.  (Scalar loads for the prefetch were tried first:
//    slower, the scalar cache misses queue up behind each other and the LDS waits share their counter.)
//  * What the compiler does to a persistent loop: it waits for a load where the VALUE is first needed, so (i) the waits
//    for W_hh landed inside the step loop (an `s_waitcnt vmcnt(0)` in every product, which then waits for the step's own
//    stores and prefetches: explicit wait before the loop), (ii) a load under a branch is waited for at the join, i.e. at
//    once (unconditional loads from clamped pointers), (iii) a register rotation copies the values just requested (no
//    rotation: one set of registers in flight for a whole step).  Addresses walk pointers (the per-step 64-bit index
//    arithmetic was 150-300 clocks of a 2300-clock step: a lone wave issues an instruction every ~5 clocks).
//  * Forward product: a lane owns a 32-k slice of the FOUR gate rows of one unit (was: 128 k's of one row): 8
//    ds_read_b128 instead of 32 per lane and step, all in flight before the first FMA, packed FMAs in eight independent
//    chains (tools/ubench/valu_rate.hip: 5.1 clocks per v_pk_fma_f32 in independent chains, 8.4 in one chain); the 16
//    slices of a unit are the 16 lanes of a DPP row, added by a 4-level butterfly (v_add_f32_dpp in inline asm: left to
//    the compiler the four chains become v_pk_add_f32 with two v_mov_dpp each); lane j of the row applies gate j's
//    nonlinearity (tanh as a scaled sigmoid: one exp2 + one rcp for all four gates at once), lane 0 collects them.
//  * Backward: ONE wave gathers the mailbox (two 16-byte granules per lane: writers j and j + 16, a DPP row per group
//    of 4 units), adds the 32 partials with the same butterfly and finishes the gate derivatives in the same lanes: no
//    LDS round trip and no barrier between the poll and the pointwise work, no 32-deep serial sum on 16 threads.  A lane
//    PAIR owns a 16-byte granule of the partial d h (4 columns x half of the rows each), so the partials leave as one
//    16-byte store per pair instead of two scattered 4-byte stores per lane; the reader resets a slot behind its own
//    partials (issuing the two reset stores takes ~300 clocks).
//  * Tried and dropped: two polls in flight half a round trip apart (the arrival is noticed with half the granularity, the
//    polling traffic doubles): forward 1.02 -> 1.04 us per step; s_sleep before the first poll or between polls: slower
//    with plain stores (it paid only while the stores were agent-scope ones: backward 1.99 -> 1.56 us at 16 x 64 clocks,
//    i.e. polling a line that is being written through costs the writer).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PK2_CONSTANT_AS __attribute__((address_space(4)))

// Butterfly sum over the 16 lanes of a DPP row; every lane of the row ends with the same bits (each level adds a pair of
// values that both partners hold, and float addition is commutative).
__device__ __forceinline__ float seq_row_sum(float v) {
  v += seq_dpp<0xB1>(v);         // quad_perm [1,0,3,2]
  v += seq_dpp<0x4E>(v);         // quad_perm [2,3,0,1]
  v += seq_dpp<0x141>(v);        // row_half_mirror: the other quad of the half row (quads are uniform by now)
  v += seq_dpp<0x140>(v);        // row_mirror: the other half row
  return v;
}
// Four of them at once, level by level (left to the compiler the four chains are paired into v_pk_add_f32, which has no
// DPP form: two v_mov_dpp + a packed add per level and pair -- three times the instructions).  A DPP operand written by
// the previous VALU instruction needs two wait states: the s_nop covers the first level, the other three chains the rest.
__device__ __forceinline__ void seq_row_sum4(float& a, float& b, float& c, float& d) {
#define PK2_SEQ_DPP_LEVEL(ctrl)                                              \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" PK2_SEQ_DPP_LEVEL("quad_perm:[1,0,3,2]") PK2_SEQ_DPP_LEVEL("quad_perm:[2,3,0,1]")
               PK2_SEQ_DPP_LEVEL("row_half_mirror") PK2_SEQ_DPP_LEVEL("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef PK2_SEQ_DPP_LEVEL
}
__device__ __forceinline__ const float* seq_uniform_ptr(const float* ptr) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
// 16 contiguous bytes through the scalar cache (the address must be wave-uniform and 16-byte aligned).
__device__ __forceinline__ f32x4 seq_sload16(const float* ptr) {
  return *reinterpret_cast<const PK2_CONSTANT_AS f32x4*>(reinterpret_cast<unsigned long long>(ptr));
}
template <int LANE>
__device__ __forceinline__ float seq_writelane(float v, float old) {        // old with lane LANE replaced by the (uniform) v
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(LANE));
  return old;
}

__device__ __forceinline__ void lstm_fwd_seq2_body(const SeqFwdParams& p, SeqCtl* ctl) {
  constexpr int H = kSH;
  constexpr int kSlicePitch = 36;           // 32 + 4: the 16 slices a lane group reads side by side start on 16 different bank quads
  __shared__ __attribute__((aligned(16))) float hs[2][16 * kSlicePitch];
  __shared__ float gxs[3][64];              // input projections [step % 3][gate * 16 + unit], written two steps ahead by wave 2
  __shared__ float outs[2][80];             // [step & 1]: gate activations [gate * 16 + unit], cells [64 + unit]; wave 3 writes them out
  __shared__ int s_i[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  SeqRole role;
  if (!seq_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  // lane roles: a 16-lane DPP row = one hidden unit, lane % 16 = its k-slice; lane j < 4 of the row finishes gate j
  const int ul = lane >> 4, l16 = lane & 15;
  const int uw = 4 * w + ul;                             // unit within the workgroup
  const int gu = 16 * rank + uw;
  const bool unit_lane = l16 == 0, gate_lane = l16 < 4;
  const bool poller = w < 2;                             // waves 0, 1 poll (a granule per lane); waves 2, 3 do the HBM traffic
  const int gran = w * 64 + lane;                        // granule of h this lane polls (128 per step)
  // gate j's nonlinearity as a * rcp(1 + exp2(s * x)) + b: sigmoid (1, -log2 e, 0), tanh (2, -2 log2 e, -1)
  const float act_s = l16 == 2 ? -2.8853900817779268f : -1.4426950408889634f;
  const float act_a = l16 == 2 ? 2.0f : 1.0f, act_b = l16 == 2 ? -1.0f : 0.0f;
  const int pre_idx = 16 * (l16 & 3) + uw, out_idx = gate_lane ? 16 * l16 + uw : 0;
  const size_t yrow = (size_t)D * H;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kSeqMaxTasks; ++iter) {
    const int task = seq_next_task(ctl, role, iter, B * D, s_i);
    if (task < 0) return;
    const int b = task / D, d = task % D;
    f32x2 wa[4][16];                                     // W_hh[d][g*H + gu][32 l16 + 2 i, + 1]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* wrow = p.whh + ((size_t)d * 4 * H + (size_t)g * H + gu) * H + 32 * l16;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + i * 4);
        wa[g][2 * i] = f32x2{v[0], v[1]};
        wa[g][2 * i + 1] = f32x2{v[2], v[3]};
      }
    }
    const float bias = (gate_lane && p.bhh) ? p.bhh[(size_t)d * 4 * H + (size_t)l16 * H + gu] : 0.f;
    // Everything addressed by time walks a pointer: t = step (d = 0) or T - 1 - step (d = 1).
    const ptrdiff_t tdir = d == 0 ? 1 : -1;
    const size_t t0 = d == 0 ? 0 : (size_t)(T - 1);
    const ptrdiff_t y_stride = tdir * (ptrdiff_t)((size_t)B * yrow);
    float* y_out = p.y + (t0 * B + b) * yrow + (size_t)d * H + gu;                        // h_t of the lane's unit
    const float* y_poll = p.y + (t0 * B + b) * yrow + (size_t)d * H + 4 * gran - y_stride;  // the granule of h_{t-1}
    // wave 2: lane = gate * 16 + unit of the workgroup's input projections; wave 3: the same lanes store the activations,
    // its lanes 0..15 the cells
    const ptrdiff_t gx_stride = tdir * (ptrdiff_t)((size_t)B * D * 4 * H);
    const float* gx_in = p.gx + (t0 * B + b) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + (size_t)(lane >> 4) * H + 16 * rank + (lane & 15);
    const ptrdiff_t gates_stride = tdir * (ptrdiff_t)((size_t)B * 4 * H), cells_stride = tdir * (ptrdiff_t)((size_t)B * H);
    float* gates_out = p.gates + (((size_t)d * T + t0) * B + b) * 4 * H + (size_t)(lane >> 4) * H + 16 * rank + (lane & 15);
    float* cells_out = p.cells + (((size_t)d * T + t0) * B + b) * H + 16 * rank + (lane & 15);
    // (loads are unconditional and walk a clamped pointer: a load under a branch makes the compiler wait for it at the
    // join, i.e. at once)
    float gnext = 0.f;
    if (w == 2) {
      gxs[0][lane] = gx_in[0];
      gx_in += T > 1 ? gx_stride : 0;
      gxs[1][lane] = gx_in[0];
      gx_in += T > 2 ? gx_stride : 0;
      gnext = gx_in[0];
      gx_in += T > 3 ? gx_stride : 0;
    }
    // W_hh, bias and the first input projections have arrived BEFORE the loop: left to itself the compiler waits for them
    // at their first use, inside the loop -- an s_waitcnt vmcnt(0) in every step's product, which then also waits for the
    // step's own stores and prefetches.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();
    float cstate = 0.f;
    bool timed_out = false;
    SeqSpin spin(ctl);
    spin.limit = 10 * kSeqSpinTicks;      // until the first exchange has worked: a team mate dispatched late is not a deadlock
    SQ_T0();
    for (int step = 0; step < T; ++step) {
      if (step == 2) { spin.limit = kSeqSpinTicks; spin.t0 = 0; }
      const int buf = step & 1;
      const float pre = gxs[step % 3][pre_idx] + bias;
      // ---- gather h_{t-1} of this (sequence, direction): 512 floats = 128 granules of 16 bytes -----------------------
      if (step > 0 && poller) {
        for (int i = 0; i < p.pre_sleep; ++i) __builtin_amdgcn_s_sleep(1);
        u32x4 v = seq_load16(y_poll);
        while (seq_has_sentinel(v)) {
          if (spin.expired()) { timed_out = true; break; }
          for (int i = 0; i < p.loop_sleep; ++i) __builtin_amdgcn_s_sleep(1);
          v = seq_load16(y_poll);
        }
        *reinterpret_cast<u32x4*>(&hs[buf][(gran >> 3) * kSlicePitch + 4 * (gran & 7)]) = v;
      }
      y_poll += y_stride;
      SQ_T(0);
      if (timed_out) s_i[3] = 1;
      seq_lds_barrier();
      SQ_T(1);
      if (s_i[3]) return;
      // ---- recurrent product: the lane's 32-k slice of the unit's four gate rows ---------------------------------------
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (step > 0) {
        const float* hq = &hs[buf][l16 * kSlicePitch];
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, c0 = a0, c1 = a0, c2 = a0, c3 = a0;
        f32x4 hva[8];                                     // all eight reads in flight before the first FMA
#pragma unroll
        for (int i = 0; i < 8; ++i) hva[i] = *reinterpret_cast<const f32x4*>(hq + 4 * i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 hv = hva[i];
          const f32x2 h01 = {hv[0], hv[1]}, h23 = {hv[2], hv[3]};
          a0 = __builtin_elementwise_fma(wa[0][2 * i], h01, a0); c0 = __builtin_elementwise_fma(wa[0][2 * i + 1], h23, c0);
          a1 = __builtin_elementwise_fma(wa[1][2 * i], h01, a1); c1 = __builtin_elementwise_fma(wa[1][2 * i + 1], h23, c1);
          a2 = __builtin_elementwise_fma(wa[2][2 * i], h01, a2); c2 = __builtin_elementwise_fma(wa[2][2 * i + 1], h23, c2);
          a3 = __builtin_elementwise_fma(wa[3][2 * i], h01, a3); c3 = __builtin_elementwise_fma(wa[3][2 * i + 1], h23, c3);
        }
        s0 = (a0[0] + a0[1]) + (c0[0] + c0[1]);
        s1 = (a1[0] + a1[1]) + (c1[0] + c1[1]);
        s2 = (a2[0] + a2[1]) + (c2[0] + c2[1]);
        s3 = (a3[0] + a3[1]) + (c3[0] + c3[1]);
        seq_row_sum4(s0, s1, s2, s3);
      }
#ifdef PK2_SEQ_PROFILE
      asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
#endif
      SQ_T(2);
      // ---- gates: lane j of the row takes gate j --------------------------------------------------------------------------
      const float x = pre + (l16 == 0 ? s0 : l16 == 1 ? s1 : l16 == 2 ? s2 : s3);
      const float act = fmaf(act_a, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(act_s * x)), act_b);
      const float fg = seq_dpp<0x101>(act), gg = seq_dpp<0x102>(act), og = seq_dpp<0x103>(act);     // row_shl 1, 2, 3
      cstate = fg * cstate + act * gg;                    // (meaningful in the row's lane 0 only)
      const float h = og * seq_tanh(cstate);
      if (unit_lane) seq_store_mode(y_out, h, p.store_mode);
      y_out += y_stride;
      SQ_T(3);
      // ---- behind the hand-over: what the backward pass needs goes to HBM through wave 3, the input projections of the
      // step after next come in through wave 2 (the polling waves keep no other vector-memory operation in flight) ---------
      if (gate_lane) outs[buf][out_idx] = act;
      if (unit_lane) outs[buf][64 + uw] = cstate;
      if (w == 2) {
        gxs[(step + 2) % 3][lane] = gnext;                // (loaded a step ago)
        gnext = *gx_in;                                   // step + 3 (clamped to the last step)
        gx_in += step + 4 < T ? gx_stride : 0;
      }
      if (w == 3 && step > 0) {                           // the previous step's values (a barrier lies between)
        *gates_out = outs[buf ^ 1][lane];
        if (lane < 16) *cells_out = outs[buf ^ 1][64 + lane];
        gates_out += gates_stride; cells_out += cells_stride;
      }
      SQ_T(4);
    }
    __syncthreads();
    if (w == 3) {
      *gates_out = outs[(T - 1) & 1][lane];
      if (lane < 16) *cells_out = outs[(T - 1) & 1][64 + lane];
    }
    SQ_PRINT("lstm_fwd_seq2", "(poll | lds + barrier | product + row sums | gates + h store | lds hand-over, wave 2 / 3 traffic)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!seq_team_barrier(ctl, role, &nbar, s_i)) return;
  }
}

__device__ __forceinline__ void lstm_bwd_seq2_body(const SeqBwdParams& p, SeqCtl* ctl) {
  constexpr int H = kSH, G4 = 4 * kSH;
  __shared__ __attribute__((aligned(16))) float dgl[64];               // this workgroup's dgates [gate * 16 + unit]
  __shared__ __attribute__((aligned(16))) float cst[3][16][8];         // [step % 3][unit]{dy, A, F, Ci, Cf, Cg, Co, -}: written two steps ahead
  __shared__ int s_i[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  SeqRole role;
  if (!seq_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  float* mail0 = p.mail + (size_t)role.team_index * kSeqMailFloats;
  const u32x4 sent = {kSeqSentinel, kSeqSentinel, kSeqSentinel, kSeqSentinel};
  const int row = lane >> 4, l16 = lane & 15;
  const bool pw_lane = w == 0 && l16 < 4;               // wave 0, lane j of row r: unit 4 r + j of the workgroup
  const int pw_unit = (4 * row + l16) & 15;
  const bool io = w == 1;                               // wave 1: prefetches and prepares the factors, writes d gx
  const bool io16 = io && lane < 16;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kSeqMaxTasks; ++iter) {
    const int task = seq_next_task(ctl, role, iter, B * D, s_i);
    if (task < 0) return;
    const int b = task / D, d = task % D;
    // A lane pair owns one 16-byte granule of the partial d h -- columns k = 4 G .. 4 G + 3, G = 32 w + lane / 2, i.e. units
    // 4 (G % 4) .. of reader G / 4 -- and each lane of the pair half of the workgroup's 64 rows (lane % 2 = 0: gates i, f;
    // 1: gates g, o): W_hh[d][(gate)*H + 16*rank + unit][4 G + j] for its 32 rows.  The pair's sums meet by one DPP add per
    // value and the even lane stores the granule: one 16-byte store per lane pair and step.
    const int gidx = 32 * w + (lane >> 1), half = lane & 1;
    f32x2 wb[32][2];
    {
      const float* wbase = p.whh + (size_t)d * G4 * H + 4 * gidx;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int rr = 32 * half + r;
        const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + ((size_t)(rr >> 4) * H + 16 * rank + (rr & 15)) * H);
        wb[r][0] = f32x2{v[0], v[1]};
        wb[r][1] = f32x2{v[2], v[3]};
      }
    }
    float dcarry = 0.f;
    bool timed_out = false;
    // Everything addressed by time walks a pointer: the backward pass visits t = T - 1 - step (d = 0) or step (d = 1).
    const ptrdiff_t tdir = d == 0 ? -1 : 1;
    const size_t t0 = d == 0 ? (size_t)(T - 1) : 0;
    const ptrdiff_t dy_stride = tdir * (ptrdiff_t)((size_t)B * D * H), gates_stride = tdir * (ptrdiff_t)((size_t)B * G4);
    const ptrdiff_t cells_stride = tdir * (ptrdiff_t)((size_t)B * H), dgx_stride = tdir * (ptrdiff_t)((size_t)B * D * G4);
    const float* dy_in = p.dy + (t0 * B + b) * ((size_t)D * H) + (size_t)d * H + 16 * rank + (lane & 15);
    const float* gates_in = p.gates + (((size_t)d * T + t0) * B + b) * G4 + 16 * rank + (lane & 15);
    const float* cells_in = p.cells + (((size_t)d * T + t0) * B + b) * H + 16 * rank + (lane & 15);
    float* dgx_out = p.dgx + (t0 * B + b) * ((size_t)D * G4) + (size_t)d * G4 + (size_t)(lane >> 4) * H + 16 * rank + (lane & 15);
    // wave 1, lanes 0..15: the forward pass's values of step s + 4 (in flight) and of step s + 3
    // wave 1: the forward pass's values of the step after next but one, in flight for a whole step.  The loads are
    // unconditional, from pointers that stop at the last step (a load under a branch makes the compiler wait for it at the
    // join, i.e. at once; so does a register rotation: it copies the values just requested); lanes 16..63 of the wave
    // repeat the addresses of lanes 0..15.
    const float* cellsp_in = cells_in + (T > 1 ? cells_stride : 0);    // the cell of the time step before (forward order)
    float n_dy = 0.f, n_i = 0.f, n_f = 0.f, n_g = 0.f, n_o = 0.f, n_c = 0.f, n_cp = 0.f;
    auto load_pw = [&](int step_) {                      // (called for step_ = 0, 1, 2, ...: the pointers walk along)
      const bool more = step_ + 1 < T, more2 = step_ + 2 < T;
      n_dy = *dy_in;
      n_i = gates_in[0]; n_f = gates_in[H]; n_g = gates_in[2 * H]; n_o = gates_in[3 * H];
      n_c = *cells_in;
      n_cp = *cellsp_in;
      dy_in += more ? dy_stride : 0; gates_in += more ? gates_stride : 0; cells_in += more ? cells_stride : 0;
      cellsp_in += more2 ? cells_stride : 0;
    };
    auto put_factors = [&](int step_, int slot) {        // from the values in flight (step step_) into cst[slot]
      const float cp = step_ == T - 1 ? 0.f : n_cp;      // the last step of the backward pass has no previous cell
      if (io16) {
        const float tc = seq_tanh(n_c);
        float* o = &cst[slot][lane][0];
        *reinterpret_cast<f32x4*>(o) = f32x4{n_dy, n_o * (1.f - tc * tc), n_f, n_g * n_i * (1.f - n_i)};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{cp * n_f * (1.f - n_f), n_i * (1.f - n_g * n_g), tc * n_o * (1.f - n_o), 0.f};
      }
    };
    if (io) {
      load_pw(0); put_factors(0, 0);
      load_pw(1); put_factors(1, 1);
      load_pw(2);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): W_hh is there before the loop (no waits for it inside)
    __syncthreads();
    // the mailbox of the step: [step % 8][reader][writer][16 units]; wave 0 reads [.][rank][writer l16 (+ 16)][units 4 row ..]
    float* mail_rd = mail0 + (size_t)rank * (kSWgs * 16) + (size_t)l16 * 16 + 4 * row;
    float* mail_wr = mail0 + ((size_t)(gidx >> 2) * kSWgs + rank) * 16 + 4 * (gidx & 3);  // peer G/4 reads [writer = rank][units 4 (G%4) ..]
    constexpr int kBox = kSWgs * kSWgs * 16;
    int s3 = 0;                                          // step % 3
    float bias_sum = 0.f;                                // wave 1: sum over the frames of its lane's d gate (the bias gradient)
    SeqSpin spin(ctl);
    spin.limit = 10 * kSeqSpinTicks;      // (as in the forward kernel)
    SQ_T0();
    for (int step = 0; step < T; ++step) {
      if (step == 2) { spin.limit = kSeqSpinTicks; spin.t0 = 0; }
      const int s3n = s3 == 2 ? 0 : s3 + 1;
      const int m8 = step & (kSeqMailDepth - 1), m8n = (step + 1) & (kSeqMailDepth - 1);
      if (w == 0) {
        // the step's factors (written by wave 1 two steps ago: a barrier lies between)
        const f32x4 fa = *reinterpret_cast<const f32x4*>(&cst[s3][pw_unit][0]);
        const f32x4 fb = *reinterpret_cast<const f32x4*>(&cst[s3][pw_unit][4]);
        // ---- gather the 32 partials of d h for the own 16 units (written by the peers during the previous step) --------
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
          float* src = mail_rd + m8 * kBox;
          u32x4 v0, v1;
          for (int i = 0; i < p.pre_sleep; ++i) __builtin_amdgcn_s_sleep(1);
#ifdef PK2_SEQ_PROFILE
          { const long long c0_ = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sq_acc_[4] += clock64() - c0_; }   // own stores acknowledged
          bool first_ = true; const long long c1_ = clock64();
#endif
          for (;;) {
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1) : "v"(src) : "memory");
#ifdef PK2_SEQ_PROFILE
            if (first_) { sq_acc_[5] += clock64() - c1_; first_ = false; }
#endif
            if (!seq_has_sentinel(v0) && !seq_has_sentinel(v1)) break;
            if (spin.expired()) { timed_out = true; break; }
            for (int i = 0; i < p.loop_sleep; ++i) __builtin_amdgcn_s_sleep(1);
          }
#ifdef PK2_SEQ_PROFILE
          sq_acc_[6] += clock64() - c1_;           // the whole poll loop
#endif
          float q0 = __uint_as_float(v0.x) + __uint_as_float(v1.x), q1 = __uint_as_float(v0.y) + __uint_as_float(v1.y);
          float q2 = __uint_as_float(v0.z) + __uint_as_float(v1.z), q3 = __uint_as_float(v0.w) + __uint_as_float(v1.w);
          seq_row_sum4(q0, q1, q2, q3);
          r4 = f32x4{q0, q1, q2, q3};
        }
        SQ_T(0);
        if (timed_out) s_i[3] = 1;
        // ---- gate derivatives of the own units: lane j of row r, unit 4 r + j -------------------------------------------
        const float rec = l16 == 0 ? r4[0] : l16 == 1 ? r4[1] : l16 == 2 ? r4[2] : r4[3];
        const float dh = fa[0] + rec;
        const float dcv = fmaf(dh, fa[1], dcarry);
        dcarry = dcv * fa[2];
        if (pw_lane) {
          dgl[pw_unit] = dcv * fa[3]; dgl[16 + pw_unit] = dcv * fb[0]; dgl[32 + pw_unit] = dcv * fb[1]; dgl[48 + pw_unit] = dh * fb[2];
        }
        SQ_T(1);
      }
      seq_lds_barrier();
      SQ_T(2);
      if (s_i[3]) {              // loud failure: the gradient of this layer turns NaN
        if (tid == 0) p.dgx[(size_t)d * G4 + 16 * rank] = __int_as_float(0x7fc00000);
        return;
      }
      float dg_own = dgl[lane];                // (wave 1 writes it out behind the product; wave 0 may be a step ahead by then:
      asm volatile("" : "+v"(dg_own));         // the read must not sink below the mailbox stores)
      if (step < T - 1) {
        // ---- own 64 rows of dgates x own W_hh rows: the lane's two columns -------------------------------------------------
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, c0 = a0, c1 = a0, c2 = a0, c3 = a0;     // eight independent chains
        f32x4 dva[8];                                       // all eight reads in flight before the first FMA
#pragma unroll
        for (int r4i = 0; r4i < 8; ++r4i) dva[r4i] = *reinterpret_cast<const f32x4*>(&dgl[32 * half + 4 * r4i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r4i = 0; r4i < 8; ++r4i) {
          const f32x4 dv = dva[r4i];
          a0 = __builtin_elementwise_fma(f32x2{dv[0], dv[0]}, wb[4 * r4i][0], a0);     c0 = __builtin_elementwise_fma(f32x2{dv[0], dv[0]}, wb[4 * r4i][1], c0);
          a1 = __builtin_elementwise_fma(f32x2{dv[1], dv[1]}, wb[4 * r4i + 1][0], a1); c1 = __builtin_elementwise_fma(f32x2{dv[1], dv[1]}, wb[4 * r4i + 1][1], c1);
          a2 = __builtin_elementwise_fma(f32x2{dv[2], dv[2]}, wb[4 * r4i + 2][0], a2); c2 = __builtin_elementwise_fma(f32x2{dv[2], dv[2]}, wb[4 * r4i + 2][1], c2);
          a3 = __builtin_elementwise_fma(f32x2{dv[3], dv[3]}, wb[4 * r4i + 3][0], a3); c3 = __builtin_elementwise_fma(f32x2{dv[3], dv[3]}, wb[4 * r4i + 3][1], c3);
        }
        a0 = (a0 + a1) + (a2 + a3); c0 = (c0 + c1) + (c2 + c3);
        float g0 = a0[0], g1 = a0[1], g2 = c0[0], g3 = c0[1];
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));
        if (half == 0) {
          const f32x4 gv = {g0, g1, g2, g3};
          u32x4 gu4;
          __builtin_memcpy(&gu4, &gv, 16);
          seq_store16_mode(mail_wr + m8n * kBox, gu4, p.store_mode);
        }
      }
      // The mailbox slot read in this step is free again: its reset goes out behind the step's own partials (issuing the
      // two stores takes ~300 clocks, which the poll of the next step has to spare) and has until the slot's next use, seven
      // steps on, to land.
      if (w == 0 && step > 0) {
        seq_store16_mode(mail_rd + m8 * kBox, sent, p.store_mode);
        seq_store16_mode(mail_rd + m8 * kBox + 256, sent, p.store_mode);
      }
      SQ_T(3);
      if (io) {                  // behind the hand-over: d gx of the step to HBM, the factors two steps ahead, the prefetch
        *dgx_out = dg_own;
        dgx_out += dgx_stride;
        bias_sum += dg_own;
        put_factors(step + 2, s3 == 0 ? 2 : s3 - 1);                   // slot (step + 2) % 3, from the loads of a step ago
        load_pw(step + 3);
      }
      s3 = s3n;
    }
    // the bias gradients (b_ih and b_hh receive the same sum) of this (sequence, direction): one atomic per gate row
    if (io && p.dbias_ih) {
      const size_t at = (size_t)d * G4 + (size_t)(lane >> 4) * H + 16 * rank + (lane & 15);
      atomicAdd(p.dbias_ih + at, bias_sum);
      if (p.dbias_hh) atomicAdd(p.dbias_hh + at, bias_sum);
    }
    SQ_PRINT("lstm_bwd_seq2", "(wave 0: mailbox poll + row sums | gate derivatives | lds barrier | product + mailbox stores || inside the poll: own stores acknowledged | first poll round trip | poll loop)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!seq_team_barrier(ctl, role, &nbar, s_i)) return;     // nobody writes a mailbox of the next pair before everybody has read the last of this one
  }
}

// A launch in which a poll timed out (or a pair was never finished) must not pass for a result: the forward
// pass keeps its NaN sentinels in y, but the backward pass would leave dgx -- freshly allocated memory -- partly
// unwritten.  One workgroup after every persistent launch turns the launch's output into NaN in that case and raises a
// sticky flag (the control block itself is cleared by the next launch) that pk2_lstm_persist_status reports.
// `mail` (backward pass): an aborted launch leaves consumed and unconsumed hand-overs in the mailboxes, which the next
// launch would read as valid ones -- the same workgroup fills them with sentinels again, so that a caller who
// lowers the guard (pk2_persist_guard_clear) and carries on does not compute on stale words.
// (round 5: the check also leaves the control block zeroed for the next launch on this stream -- one hipMemsetAsync, i.e.
// one fill kernel, less per recurrence launch: six per LF-MMI step)
__global__ void lstm_seq_check(SeqCtl* ctl, unsigned pairs, float* out, size_t n, unsigned* sticky, unsigned* guard_dev, unsigned* guard_host,
                               float* mail, size_t mail_n) {
  const bool good = ctl->abort == 0u && ctl->done == pairs;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < sizeof(SeqCtl) / sizeof(unsigned); i += blockDim.x) reinterpret_cast<unsigned*>(ctl)[i] = 0u;
  if (good) return;
  if (threadIdx.x == 0) { *sticky = 1u; persist_guard_raise(guard_dev, guard_host); }
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
  for (size_t i = threadIdx.x; i < mail_n; i += blockDim.x) mail[i] = __uint_as_float(kSeqSentinel);
}

// Round 6: the check rides in the recurrence launch.  Every workgroup counts itself out; the last one to leave sees the
// control block in its final state (every other workgroup's atomics are in L2 before its exit count), does what
// lstm_seq_check does -- verdict, control block zeroed for the next launch, on failure the sticky flag, the guard and the NaN
// poison -- and the step has six launches less.  (first use on a device: `fold` = 0, the host reads the control block itself)
struct SeqExit { unsigned pairs; int fold; float* out; size_t n; unsigned* sticky; unsigned* guard_dev; unsigned* guard_host; float* mail; size_t mail_n; };
__device__ __forceinline__ void seq_exit_check(SeqCtl* ctl, const SeqExit& x) {
  if (!x.fold) return;
  __shared__ unsigned s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&ctl->exited, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  const bool good = seq_load_u(&ctl->abort) == 0u && seq_load_u(&ctl->done) == x.pairs && x.fold != 2;     // (fold = 2: test hook)
  __syncthreads();
  for (unsigned i = threadIdx.x; i < sizeof(SeqCtl) / sizeof(unsigned); i += blockDim.x) reinterpret_cast<unsigned*>(ctl)[i] = 0u;
  if (good) return;
  if (threadIdx.x == 0) { if (x.fold != 2) *x.sticky = 1u; persist_guard_raise(x.guard_dev, x.guard_host); }
  for (size_t i = threadIdx.x; i < x.n; i += blockDim.x) x.out[i] = __uint_as_float(0x7fc00000u);
  for (size_t i = threadIdx.x; i < x.mail_n; i += blockDim.x) x.mail[i] = __uint_as_float(kSeqSentinel);
}
__global__ void __launch_bounds__(256) lstm_fwd_seq2(SeqFwdParams p, SeqCtl* ctl, SeqExit x) {
  lstm_fwd_seq2_body(p, ctl);
  seq_exit_check(ctl, x);
}
__global__ void __launch_bounds__(256) lstm_bwd_seq2(SeqBwdParams p, SeqCtl* ctl, SeqExit x) {
  lstm_bwd_seq2_body(p, ctl);
  seq_exit_check(ctl, x);
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct SeqScratch { SeqCtl* ctl = nullptr; float* mail = nullptr; unsigned* sticky = nullptr; PersistGuard guard;
                    int mail_clean_teams = 0;         // mailboxes of that many teams per XCD are known to hold only sentinels
                    bool ctl_clean = false; };        // the last launch's check kernel has zeroed the control block
static std::map<DevStream, SeqScratch> g_seq_scratch;
static PerDevice<int> g_seq_state_pd(-1);             // -1 untested, 0 unusable, 1 verified on this device

// pk2_persist_guard_clear: whatever an aborted launch left behind, the next backward launch fills every mailbox again
// (belt and braces next to the refill by lstm_seq_check: a launch that was killed never reached its check kernel).
static void seq_mail_dirty() { for (auto& kv : g_seq_scratch) { kv.second.mail_clean_teams = 0; kv.second.ctl_clean = false; } }

static int seq_scratch(hipStream_t stream, SeqScratch** out) {
  static const bool hooked = (persist_guard_on_clear(seq_mail_dirty), true);
  (void)hooked;
  SeqScratch& sc = g_seq_scratch[dev_stream(stream)];
  if (!sc.ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(SeqCtl)));
  if (!sc.mail) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.mail), (size_t)8 * kSeqTeams * kSeqMailFloats * sizeof(float)));
  if (!sc.sticky) {
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.sticky), sizeof(unsigned)));
    PK2_HIP(hipMemsetAsync(sc.sticky, 0, sizeof(unsigned), stream));
  }
  if (!sc.guard.dev) { int rc = persist_guard(&sc.guard); if (rc) return rc; }
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

// PK2_LSTM_SEQ_FORM=1 keeps the round-2 kernels (lstm_fwd_seq / lstm_bwd_seq) for A/B runs; default: the round-4 forms.
static int seq_form() {
  static int form = -1;
  if (form < 0) {
    const char* env = getenv("PK2_LSTM_SEQ_FORM");
    form = (env && atoi(env) == 1) ? 1 : 2;
  }
  return form;
}

bool lstm_seq_wanted(int B, int H, int D) {
  const char* env = getenv("PK2_LSTM_SEQ");
  if (env && atoi(env) == 0) return false;
  // (up to 4 pairs per XCD one after the other; larger batches are better served by the batched step kernels)
  if (g_seq_state_pd.ref() == 0 || H != kSH || B < 1 || B * D > 32 || (D != 1 && D != 2)) return false;
  static PerDevice<int> cus_pd(-1); int& cus = cus_pd.ref();
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
  if (!sc->ctl_clean) PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(SeqCtl), stream));
  sc->ctl_clean = false;
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(y), (int)kSeqSentinel, (size_t)T * B * D * H, stream));
  static const int pre = getenv("PK2_SEQ_FWD_PRESLEEP") ? atoi(getenv("PK2_SEQ_FWD_PRESLEEP")) : 0;
  static const int lps = getenv("PK2_SEQ_FWD_LOOPSLEEP") ? atoi(getenv("PK2_SEQ_FWD_LOOPSLEEP")) : 0;
  static const int smode = getenv("PK2_SEQ_STORE_MODE") ? atoi(getenv("PK2_SEQ_STORE_MODE")) : 1;
  SeqFwdParams p{gx, whh, bhh, y, gates, cells, B, T, D, pre, lps, smode};
  static const bool fold_env = [] { const char* e = getenv("PK2_LSTM_SEQ_FOLD_CHECK"); return !(e && atoi(e) == 0); }();
  const bool fold = fold_env && seq_form() != 1 && g_seq_state_pd.ref() == 1;
  // (test hook, read per call: PK2_LSTM_SEQ_TEST_FAIL=1 makes the folded check of a forward launch behave as if a poll had timed out)
  const char* tf_e = getenv("PK2_LSTM_SEQ_TEST_FAIL");
  const int fold_mode = fold ? ((tf_e && atoi(tf_e) == 1) ? 2 : 1) : 0;
  const SeqExit ex{(unsigned)(B * D), fold_mode, y, (size_t)T * B * D * H, sc->sticky, sc->guard.dev, sc->guard.host_dev, nullptr, (size_t)0};
  if (seq_form() == 1) hipLaunchKernelGGL(lstm_fwd_seq, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl);
  else hipLaunchKernelGGL(lstm_fwd_seq2, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl, ex);
  PK2_LAUNCH_CHECK();
  if (g_seq_state_pd.ref() < 0) {                 // first use on this device: every pair done, nobody timed out?
    SeqCtl* h = new SeqCtl;
    hipError_t e = hipMemcpyAsync(h, sc->ctl, sizeof(SeqCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)(B * D);
    delete h;
    if (e != hipSuccess) { set_error("lstm_seq: %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_seq_state_pd.ref() = ok ? 1 : 0;
    if (!ok) return PK2_OK;              // the caller falls back (and keeps doing so)
  }
  if (!fold)
    hipLaunchKernelGGL(lstm_seq_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)(B * D), y, (size_t)T * B * D * H, sc->sticky, sc->guard.dev, sc->guard.host_dev,
                       nullptr, (size_t)0);
  sc->ctl_clean = true;
  *ran = true;
  return PK2_OK;
}

int lstm_bwd_seq_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                        int D, float* dgx, hipStream_t stream, bool* ran, float* dbias_ih, float* dbias_hh, bool* bias_done) {
  *ran = false;
  if (bias_done) *bias_done = false;
  if (g_seq_state_pd.ref() != 1) return PK2_OK;   // the forward pass verifies the device first
  SeqScratch* sc = nullptr;
  int rc = seq_scratch(stream, &sc);
  if (rc) return rc;
  if (!sc->ctl_clean) PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(SeqCtl), stream));
  sc->ctl_clean = false;
  // Every reader resets the slots it has read, so a launch that completes leaves the mailboxes as it found them: all
  // sentinels.  They are filled once (and again when a launch needs more teams than have been filled); a launch that
  // gave up raises the guard, which stops training anyway (persist_guard.h).
  if (sc->mail_clean_teams < seq_teams(B * D)) {
    PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sc->mail), (int)kSeqSentinel, (size_t)8 * seq_teams(B * D) * kSeqMailFloats, stream));
    sc->mail_clean_teams = seq_teams(B * D);
  }
  static const int pre = getenv("PK2_SEQ_BWD_PRESLEEP") ? atoi(getenv("PK2_SEQ_BWD_PRESLEEP")) : 0;
  static const int lps = getenv("PK2_SEQ_BWD_LOOPSLEEP") ? atoi(getenv("PK2_SEQ_BWD_LOOPSLEEP")) : 0;
  static const int smode = getenv("PK2_SEQ_STORE_MODE") ? atoi(getenv("PK2_SEQ_STORE_MODE")) : 1;
  const bool with_bias = seq_form() != 1 && dbias_ih != nullptr;
  SeqBwdParams p{dy, whh, gates, cells, dgx, sc->mail, B, T, D, pre, lps, smode, with_bias ? dbias_ih : nullptr, with_bias ? dbias_hh : nullptr};
  if (bias_done) *bias_done = with_bias;
  static const bool fold_env = [] { const char* e = getenv("PK2_LSTM_SEQ_FOLD_CHECK"); return !(e && atoi(e) == 0); }();
  const bool fold = fold_env && seq_form() != 1;
  const SeqExit ex{(unsigned)(B * D), fold ? 1 : 0, dgx, (size_t)T * B * D * 4 * H, sc->sticky, sc->guard.dev, sc->guard.host_dev, sc->mail,
                   (size_t)8 * kSeqTeams * kSeqMailFloats};
  if (seq_form() == 1) hipLaunchKernelGGL(lstm_bwd_seq, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl);
  else hipLaunchKernelGGL(lstm_bwd_seq2, dim3(8 * kSWgs * seq_teams(B * D)), dim3(256), 0, stream, p, sc->ctl, ex);
  if (!fold)
    hipLaunchKernelGGL(lstm_seq_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)(B * D), dgx, (size_t)T * B * D * 4 * H, sc->sticky, sc->guard.dev, sc->guard.host_dev,
                       sc->mail, (size_t)8 * kSeqTeams * kSeqMailFloats);
  PK2_LAUNCH_CHECK();
  sc->ctl_clean = true;
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
