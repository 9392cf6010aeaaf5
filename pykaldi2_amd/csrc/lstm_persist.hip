// Persistent recurrence of one (bi)directional LSTM layer for small batches (B <= 8, H = 512) on gfx950.
//
// The launch-per-step kernels of lstm.hip pay, every time step, a dependent kernel launch and -- because the L2s of
// the 8 XCDs are not coherent with each other and are invalidated at kernel boundaries -- a COLD round trip to memory
// for W_hh, h_{t-1}, the input projection and c_{t-1} (measured with in-kernel timers: 1.6 of the 3.9 us of a step is
// that one load phase; making every workgroup read the same W_hh slice changes nothing: latency, not bandwidth).
// Here a direction lives on ONE XCD for the whole sequence:
//  * 32 workgroups (one per CU of the XCD), workgroup r owns hidden units 16r..16r+15; its 64 gate rows of W_hh
//    (128 KB) stay in VGPRs, c_t in a register of the lane that owns (unit, batch row);
//  * the recurrent product runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction
//    = (4 units of one gate) x (4 batch rows) x one k -- no padding at batch 4, 128 instructions per wave and step;
//  * h_t is exchanged through the XCD's own L2: every workgroup stores its 16 x B values into y[t] (the layer output,
//    pre-filled with a NaN sentinel) and polls the 8 KB of the direction with L1-bypassing 16-byte loads until no
//    sentinel is left -- no flags, no atomics, no cross-XCD traffic;
//  * workgroups find their XCD from the XCC_ID register and take roles by arrival order, so nothing depends on how
//    the dispatcher deals workgroup ids to XCDs; a timeout raises an abort flag instead of hanging.
// Replaces the same cuDNN RNN as lstm.hip (reference models/lstm.py:49-58); the step kernels remain for other shapes.
#include <cstdlib>
#include <map>

#include "common.h"
#include "lstm_persist.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef PK2_PERSIST_SCALAR_STORES
#define PK2_PERSIST_SCALAR_STORES 0
#endif
#ifndef PK2_PERSIST_DUAL
#define PK2_PERSIST_DUAL 1
#endif
constexpr unsigned kSentinelBits = 0x7fc0dead;     // a NaN payload no LSTM output can take
constexpr int kPH = 512;                           // hidden size served
constexpr int kPUnits = 16;                        // hidden units per workgroup
constexpr int kPWgs = kPH / kPUnits;               // workgroups per direction = CUs of an XCD
constexpr long long kSpinTicks = 1000LL * 1000 * 100;  // 1 s of the 100 MHz wall clock before a poll gives up (peers whose
                                                       // dispatch is delayed by other kernels on the chip are simply waited for)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
// 16-byte load that bypasses the CU's L1 (agent scope): sees what other CUs of the XCD have stored into the shared L2
__device__ __forceinline__ u32x4 load16_agent(const float* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ bool has_sentinel(u32x4 v) {
  return v.x == kSentinelBits || v.y == kSentinelBits || v.z == kSentinelBits || v.w == kSentinelBits;
}

struct PersistFwdParams {
  const float* gx;    // [T][B][D*4H]
  const float* whh;   // [D][4H][H]
  const float* bhh;   // [D][4H] or null
  float* y;           // [T][B][D*H], pre-filled with the sentinel
  float* gates;       // [D][T][B][4H]
  float* cells;       // [D][T][B][H]
  int B, T, D;
};

#ifdef PK2_PERSIST_PROFILE
__device__ unsigned long long g_pp[8];
#define PP_T(k) do { const long long n_ = clock64(); pp_acc[k] += (unsigned long long)(n_ - pp_last); pp_last = n_; } while (0)
__device__ unsigned long long g_ppb[8];
__global__ void ppb_print(int steps) {
  printf("lstm_bwd_persist rank 0 thread 0, shader cycles per step over %d steps: loop top %llu | mailbox polled, reset, written to LDS %llu | barrier %llu | reduce 32 partials + gate derivatives %llu | store wait + barrier %llu | LDS reads + 128 MFMA + partial stores %llu\n",
         steps, g_ppb[0] / steps, g_ppb[1] / steps, g_ppb[2] / steps, g_ppb[3] / steps, g_ppb[4] / steps, g_ppb[5] / steps);
}
__global__ void pp_print(int steps) {
  printf("lstm_fwd_persist rank 0 thread 0, shader cycles per step over %d steps: loop top %llu | h polled and written to LDS %llu | barrier %llu | LDS reads + 128 MFMA + reduce %llu | transpose + gates + stores %llu\n",
         steps, g_pp[0] / steps, g_pp[1] / steps, g_pp[2] / steps, g_pp[3] / steps, g_pp[4] / steps);
  for (int k = 0; k < 8; ++k) g_pp[k] = 0;
}
#else
#define PP_T(k) do { } while (0)
#endif

// NBG = groups of 4 batch rows (1: B <= 4, 2: B <= 8): W_hh is read from the registers once per group.
template <int NBG>
__global__ void __launch_bounds__(256) lstm_fwd_persist(PersistFwdParams p, PersistCtl* ctl) {
  constexpr int H = kPH, NB = 4 * NBG;
  // h_{t-1} of the direction, double buffered, as [batch row][k-phase][128 + 4]: the 16 (batch row, k-phase) streams the
  // MFMA lanes read side by side start 16 bytes apart modulo the 64 banks (unpadded they all start on bank 0: measured
  // 1.9 us per step in 16-way conflicts)
  constexpr int kPhasePitch = 132, kRowPitch = 4 * kPhasePitch;
  __shared__ __attribute__((aligned(16))) float hs[2][NB * kRowPitch];
  __shared__ __attribute__((aligned(16))) float tr[4][NBG][16][4];   // per-wave transposition of the (gate, batch row) sums
  __shared__ int s_rank, s_dir, s_abort;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    s_abort = 0;
    const unsigned xcd = xcc_id();
    int r = -1;
    if ((int)xcd < p.D) {
      const unsigned slot = __hip_atomic_fetch_add(&ctl->reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < (unsigned)kPWgs) r = (int)slot;
    }
    s_rank = r; s_dir = (int)xcd;
  }
  __syncthreads();
  if (s_rank < 0) return;
  const int d = s_dir, B = p.B, T = p.T, D = p.D;
  const int u0 = s_rank * kPUnits + w * 4;       // this wave's 4 hidden units
  // MFMA roles of the lane: block b = lane/4 -> gate g = b%4 and k-phase kp = b/4 (128 k's each); A row / B column = lane%4
  const int blk = lane >> 2, q4 = lane & 3, g = blk & 3, kp = blk >> 2;
  float wa[128];                                 // W_hh[d][g*H + u0 + q4][kp*128 + i]
  {
    const float* wrow = p.whh + ((size_t)d * 4 * H + (size_t)g * H + u0 + q4) * H + kp * 128;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + i * 4);
      wa[i * 4] = v[0]; wa[i * 4 + 1] = v[1]; wa[i * 4 + 2] = v[2]; wa[i * 4 + 3] = v[3];
    }
  }
  // gate-math role (lanes 0 .. 16*NBG-1 of every wave): batch group lane/16, unit u0 + (lane%16)/4, row 4*group + lane%4
  const int gg_ = lane >> 4, gu = u0 + ((lane & 15) >> 2), gc = 4 * gg_ + (lane & 3);
  const bool gate_lane = lane < 16 * NBG && gc < B;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate_lane && p.bhh) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[k] = p.bhh[(size_t)d * 4 * H + (size_t)k * H + gu];
  }
  float cstate = 0.f;
  bool timed_out = false;
  const size_t yrow = (size_t)D * H;
  float gxn[4] = {0.f, 0.f, 0.f, 0.f};             // input projection of the step about to run
  auto load_gx = [&](int step_) {
    if (gate_lane && step_ < T) {
      const int t_ = d == 0 ? step_ : T - 1 - step_;
      const float* gxr = p.gx + ((size_t)t_ * B + gc) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + gu;
#pragma unroll
      for (int k = 0; k < 4; ++k) gxn[k] = gxr[(size_t)k * H];
    }
  };
  load_gx(0);
#ifdef PK2_PERSIST_PROFILE
  unsigned long long pp_acc[5] = {0, 0, 0, 0, 0};
  long long pp_last = clock64();
#endif
  for (int step = 0; step < T; ++step) {
    const int t = d == 0 ? step : T - 1 - step;
    const int tp = d == 0 ? t - 1 : t + 1;
    const int buf = step & 1;
    float pre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pre[k] = gxn[k] + bias[k];       // loaded one step ahead (does not depend on h)
    PP_T(0);
    // ---- gather h_{t-1} of this direction: [NB][H] floats = NB * 128 granules of 16 bytes, two per thread and group ---
#pragma unroll
    for (int grp = 0; grp < NBG; ++grp) {
      const int c0 = 4 * grp + (tid >> 7), c1 = c0 + 2, k4 = (tid & 127) * 4;
      u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
      const bool need0 = step > 0 && c0 < B, need1 = step > 0 && c1 < B;
      const float* src0 = p.y + ((size_t)tp * B + c0) * yrow + (size_t)d * H + k4;
      const float* src1 = p.y + ((size_t)tp * B + c1) * yrow + (size_t)d * H + k4;
      if (need0 || need1) {
        long long t0 = 0;                      // the wall clock is only read once a poll has been spinning for a while
        unsigned spins = 0;
        bool bad;
        do {
          // both loads in flight, one wait
          if (need0 && need1) {
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1) : "v"(src0), "v"(src1) : "memory");
          } else {
            if (need0 && (spins == 0 || has_sentinel(v0))) v0 = load16_agent(src0);
            if (need1 && (spins == 0 || has_sentinel(v1))) v1 = load16_agent(src1);
          }
          bad = (need0 && has_sentinel(v0)) || (need1 && has_sentinel(v1));
          ++spins;
          if (bad && (spins & 255u) == 0u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > kSpinTicks || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              timed_out = true;
              break;
            }
          }
        } while (bad);
      }
      *reinterpret_cast<u32x4*>(&hs[buf][c0 * kRowPitch + (k4 >> 7) * kPhasePitch + (k4 & 127)]) = v0;
      *reinterpret_cast<u32x4*>(&hs[buf][c1 * kRowPitch + (k4 >> 7) * kPhasePitch + (k4 & 127)]) = v1;
    }
    PP_T(1);
    // a poll that gave up (here or, seen through the flag inside the slow path, elsewhere) ends the kernel: y keeps its
    // sentinels, the host sees the flag.  No per-step read of the flag: that would be one more L2 round trip per step.
    if (timed_out) s_abort = 1;
    __syncthreads();
    if (s_abort) return;
    PP_T(2);
    load_gx(step + 1);
    // ---- recurrent product: D_blk[r][c] += W[g][u0+r][k] * h[c][k] over the lane block's 128 k's ---------------------
    f32x4 acc[NBG];
#pragma unroll
    for (int grp = 0; grp < NBG; ++grp) acc[grp] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
#pragma unroll
      for (int grp = 0; grp < NBG; ++grp) {
        // all of the lane's h values first (32 LDS reads in flight), then four independent accumulator chains
        const float* hrow = &hs[buf][(4 * grp + q4) * kRowPitch + kp * kPhasePitch];
        f32x4 hv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) hv[i] = *reinterpret_cast<const f32x4*>(hrow + i * 4);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[i * 4], hv[i][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[i * 4 + 1], hv[i][1], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[i * 4 + 2], hv[i][2], a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[i * 4 + 3], hv[i][3], a3, 0, 0, 0);
        }
        f32x4 sum = (a0 + a1) + (a2 + a3);
        // the four k-phases of a (gate, batch row) sit 16 lanes apart
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sum[r];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          sum[r] = v;
        }
        acc[grp] = sum;
      }
    }
    PP_T(3);
    // lane 4g + c (< 16) holds the sums of gate g, batch row c for units u0..u0+3: transpose to (unit, batch row) lanes
    if (lane < 16) {
#pragma unroll
      for (int grp = 0; grp < NBG; ++grp) *reinterpret_cast<f32x4*>(&tr[w][grp][lane][0]) = acc[grp];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (gate_lane) {
      const int ul = (lane & 15) >> 2, cl = lane & 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) pre[k] += tr[w][gg_][4 * k + cl][ul];
      const float ig = fsig(pre[0]), fg = fsig(pre[1]), gg = ftanh(pre[2]), og = fsig(pre[3]);
      cstate = fg * cstate + ig * gg;
      const float h = og * ftanh(cstate);
      store_agent(p.y + ((size_t)t * B + gc) * yrow + (size_t)d * H + gu, h);
      p.cells[(((size_t)d * T + t) * B + gc) * H + gu] = cstate;
      float* gr = p.gates + (((size_t)d * T + t) * B + gc) * 4 * H + gu;
      gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
    }
    PP_T(4);
  }
#ifdef PK2_PERSIST_PROFILE
  if (tid == 0 && s_rank == 0 && d == 0) for (int k = 0; k < 5; ++k) g_pp[k] = pp_acc[k];
#endif
}

// ---- backward through time ----------------------------------------------------------------------------------------------
// Same residency (direction on one XCD, workgroup r owns units 16r..16r+15, its 64 gate rows of W_hh in VGPRs), other
// dataflow: d h[b][k] = dy[b][k] + sum over ALL 2048 gate rows of dgates[b][row] * W_hh[row][k].  Gathering all of
// dgates would be 32 KB per workgroup and step; instead every workgroup multiplies ITS OWN 64 rows of dgates (which it has
// just produced, in LDS) with its W_hh rows into a partial [B][512] (again v_mfma_f32_4x4x1: four batch rows x four
// k's x one row per block, no cross-lane reduction at all), stores the 16 x B values each peer needs into that peer's
// mailbox (one 16-byte store per column and batch group: layout [writer][unit][batch row]) and gathers its own mailbox,
// 32 x 16 x B floats, summing the 32 partials.  Mailboxes are double buffered and reset to the sentinel by their reader: a
// peer can only write step s+2 after it has read what this workgroup produced in step s+1, which this workgroup stored
// after its resets of step s were acknowledged (wait for the outstanding stores + barrier).
struct PersistBwdParams {
  const float* dy;     // [T][B][D*H]
  const float* whh;    // [D][4H][H]
  const float* gates;  // [D][T][B][4H]
  const float* cells;  // [D][T][B][H]
  float* dgx;          // [T][B][D*4H]
  float* px;           // [D][2][32 reader][32 writer][16 units][4*NBG batch rows] mailboxes, pre-filled with the sentinel
  int B, T, D;
};

// The s_nop covers the ">8-byte VMEM store data followed by a VALU write of the same VGPRs" hazard: the compiler
// inserts that wait state for its own stores but does not see one inside an asm statement (without it the first
// two dwords were overwritten by the next VALU instruction before the store had read them).
__device__ __forceinline__ void store16_agent(float* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int NBG>
__global__ void __launch_bounds__(256) lstm_bwd_persist(PersistBwdParams p, PersistCtl* ctl) {
  constexpr int H = kPH, NB = 4 * NBG, BOX = 16 * NB;       // floats a writer leaves in a reader's mailbox
  __shared__ __attribute__((aligned(16))) float pl[kPWgs][BOX + 4];  // gathered partials [writer][unit * NB + batch row]
  __shared__ __attribute__((aligned(16))) float dgl[NB][64 + 4];     // this workgroup's dgates [batch row][gate * 16 + unit]
  __shared__ int s_rank, s_dir, s_abort;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    s_abort = 0;
    const unsigned xcd = xcc_id();
    int r = -1;
    if ((int)xcd < p.D) {
      const unsigned slot = __hip_atomic_fetch_add(&ctl->reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < (unsigned)kPWgs) r = (int)slot;
    }
    s_rank = r; s_dir = (int)xcd;
  }
  __syncthreads();
  if (s_rank < 0) return;
  const int d = s_dir, rank = s_rank, B = p.B, T = p.T, D = p.D;
  constexpr int G4 = 4 * H;
  // B operand of the MFMAs: W_hh[d][(r/16)*H + 16*rank + r%16][128*w + 64*cg + lane] for the 64 own rows r, cg = 0, 1
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
  // pointwise role (threads 0 .. 16*NB-1): batch row tid/16, unit 16*rank + tid%16
  const int pb = tid >> 4, pu = tid & 15, unit = 16 * rank + pu;
  const bool pw_thread = tid < 16 * NB, pw = pw_thread && pb < B;
  float dcarry = 0.f;
  bool timed_out = false;
  float* mail[2];
  mail[0] = p.px + ((size_t)(d * 2 + 0) * kPWgs + rank) * (kPWgs * BOX);
  mail[1] = p.px + ((size_t)(d * 2 + 1) * kPWgs + rank) * (kPWgs * BOX);
  const u32x4 sent = {kSentinelBits, kSentinelBits, kSentinelBits, kSentinelBits};
  // operands of the pointwise stage, one step ahead
  float n_dy = 0.f, n_i = 0.f, n_f = 0.f, n_g = 0.f, n_o = 0.f, n_c = 0.f, n_cp = 0.f;
  auto load_pw = [&](int step_) {
    if (pw && step_ < T) {
      const int fs = T - 1 - step_;
      const int t_ = d == 0 ? fs : T - 1 - fs;
      const int tp_ = d == 0 ? t_ - 1 : t_ + 1;
      n_dy = p.dy[((size_t)t_ * B + pb) * ((size_t)D * H) + (size_t)d * H + unit];
      const float* gr = p.gates + (((size_t)d * T + t_) * B + pb) * G4 + unit;
      n_i = gr[0]; n_f = gr[(size_t)H]; n_g = gr[(size_t)2 * H]; n_o = gr[(size_t)3 * H];
      n_c = p.cells[(((size_t)d * T + t_) * B + pb) * H + unit];
      n_cp = fs == 0 ? 0.f : p.cells[(((size_t)d * T + tp_) * B + pb) * H + unit];
    }
  };
  load_pw(0);
#ifdef PK2_PERSIST_PROFILE
  unsigned long long pp_acc[6] = {0, 0, 0, 0, 0, 0};
  long long pp_last = clock64();
#endif
  for (int step = 0; step < T; ++step) {
    const int fstep = T - 1 - step;
    const int t = d == 0 ? fstep : T - 1 - fstep;
    const float c_dy = n_dy, c_i = n_i, c_f = n_f, c_g = n_g, c_o = n_o, c_c = n_c, c_cp = n_cp;
    load_pw(step + 1);
    PP_T(0);
    // ---- gather the 32 partials of d h for the own 16 units (written by the peers during the previous step) ------------
    float rec = 0.f;
    if (step > 0) {
      float* box = mail[step & 1];
#pragma unroll
      for (int grp = 0; grp < NBG; ++grp) {
        // granule = writer * (4 * NBG * 4... ) : the mailbox is 32 * BOX floats = 512 * NBG granules, two per thread and group
        float* src0 = box + (size_t)(grp * 512 + tid) * 4;
        float* src1 = src0 + 1024;
        u32x4 v0, v1;
        long long t0 = 0;
        unsigned spins = 0;
        bool bad;
        do {
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(v0), "=&v"(v1) : "v"(src0), "v"(src1) : "memory");
          bad = has_sentinel(v0) || has_sentinel(v1);
          ++spins;
          if (bad && (spins & 255u) == 0u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > kSpinTicks || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              timed_out = true;
              break;
            }
          }
        } while (bad);
        store16_agent(src0, sent);               // the mailbox is free again (ordered before this step's own stores below)
        store16_agent(src1, sent);
        const int g0 = grp * 512 + tid, g1 = g0 + 256;      // float offset 4 * granule = writer * BOX + (unit * NB + row)
        *reinterpret_cast<u32x4*>(&pl[(g0 * 4) / BOX][(g0 * 4) % BOX]) = v0;
        *reinterpret_cast<u32x4*>(&pl[(g1 * 4) / BOX][(g1 * 4) % BOX]) = v1;
      }
      PP_T(1);
      if (timed_out) s_abort = 1;
      __syncthreads();
      PP_T(2);
      if (s_abort) {       // loud failure: the gradient of this layer turns NaN (nobody waits on a flag on the hot path)
        if (tid == 0) p.dgx[(size_t)d * G4 + 16 * rank] = __int_as_float(0x7fc00000);
        return;
      }
      if (pw_thread) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int col = pu * NB + pb;
#pragma unroll
        for (int q = 0; q < kPWgs; q += 4) { s0 += pl[q][col]; s1 += pl[q + 1][col]; s2 += pl[q + 2][col]; s3 += pl[q + 3][col]; }
        rec = (s0 + s1) + (s2 + s3);
      }
    }
    // ---- gate derivatives of (batch row, unit) ---------------------------------------------------------------------------
    if (pw_thread) {
      float dgi = 0.f, dgf = 0.f, dgg = 0.f, dgo = 0.f;
      if (pw) {
        const float dh = c_dy + rec;
        const float tc = ftanh(c_c);
        const float dcv = dcarry + dh * c_o * (1.f - tc * tc);
        dcarry = dcv * c_f;
        dgi = dcv * c_g * c_i * (1.f - c_i);
        dgf = dcv * c_cp * c_f * (1.f - c_f);
        dgg = dcv * c_i * (1.f - c_g * c_g);
        dgo = dh * tc * c_o * (1.f - c_o);
        float* o = p.dgx + ((size_t)t * B + pb) * ((size_t)D * G4) + (size_t)d * G4 + unit;
        o[0] = dgi; o[(size_t)H] = dgf; o[(size_t)2 * H] = dgg; o[(size_t)3 * H] = dgo;
      }
      dgl[pb][pu] = dgi; dgl[pb][16 + pu] = dgf; dgl[pb][32 + pu] = dgg; dgl[pb][48 + pu] = dgo;
    }
    PP_T(3);
    if (step == T - 1) break;
    // this thread's mailbox resets have been acknowledged by the L2 ... (a plain wait for the outstanding stores: an
    // agent-scope release FENCE would write the whole L2 back to memory on this multi-XCD part -- measured 12 us per step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // ... and so have everybody's, before any partial goes out
    PP_T(4);
    // ---- own 64 rows of dgates x own W_hh rows: partial[batch row][k] for k = 128 w + 64 cg + lane ---------------------
#pragma unroll
    for (int grp = 0; grp < NBG; ++grp) {
      const float* arow = &dgl[4 * grp + (lane & 3)][0];
      f32x4 av[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = *reinterpret_cast<const f32x4*>(arow + q * 4);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][0], wb[8 * q], a0, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][0], wb[8 * q + 1], b0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][1], wb[8 * q + 2], a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][1], wb[8 * q + 3], b1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][2], wb[8 * q + 4], a0, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][2], wb[8 * q + 5], b0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][3], wb[8 * q + 6], a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q][3], wb[8 * q + 7], b1, 0, 0, 0);
      }
      const f32x4 r0 = a0 + a1, r1 = b0 + b1;     // register i = batch row 4*grp + i; column k = 128 w + 64 cg + lane
      // peer k/16 reads [writer = rank][unit k%16][batch row] from its mailbox of the next step: one 16-byte store
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const int k = 128 * w + 64 * cg + lane;
        float* dst = p.px + (((size_t)(d * 2 + ((step + 1) & 1)) * kPWgs + (k >> 4)) * kPWgs + rank) * BOX + (k & 15) * NB + 4 * grp;
        const f32x4 rv = cg == 0 ? r0 : r1;
#if PK2_PERSIST_SCALAR_STORES
#pragma unroll
        for (int i = 0; i < 4; ++i) store_agent(dst + i, rv[i]);
#else
        u32x4 bits;
        bits.x = __float_as_uint(rv[0]); bits.y = __float_as_uint(rv[1]); bits.z = __float_as_uint(rv[2]); bits.w = __float_as_uint(rv[3]);
        store16_agent(dst, bits);
#endif
      }
    }
    PP_T(5);
  }
#ifdef PK2_PERSIST_PROFILE
  if (tid == 0 && s_rank == 0 && d == 0) for (int k = 0; k < 6; ++k) g_ppb[k] = pp_acc[k];
#endif
}

// ---- host -------------------------------------------------------------------------------------------------------------
static std::map<DevStream, PersistCtl*> g_ctls;   // device control blocks, one per caller stream (launches on different
                                                    // streams may overlap; each needs its own role counters)
static PersistCtl* g_ctl_last = nullptr;
static PerDevice<int> g_persist_state_pd(-1);         // -1 untested, 0 unusable, 1 verified on this device

static int get_ctl(hipStream_t stream, PersistCtl** out) {
  auto it = g_ctls.find(dev_stream(stream));
  if (it == g_ctls.end()) {
    PersistCtl* c = nullptr;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&c), sizeof(PersistCtl)));
    it = g_ctls.emplace(dev_stream(stream), c).first;
  }
  *out = g_ctl_last = it->second;
  return PK2_OK;
}

bool lstm_persist_wanted(int B, int H, int D) {
  const char* env = getenv("PK2_LSTM_PERSIST");
  if (env && atoi(env) == 0) return false;
  return g_persist_state_pd.ref() != 0 && H == kPH && B >= 1 && B <= 8 && (D == 1 || D == 2);
}

int lstm_fwd_persist_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                            float* gates, float* cells, hipStream_t stream, bool* ran) {
  *ran = false;
  PersistCtl* g_ctl = nullptr;
  int crc = get_ctl(stream, &g_ctl);
  if (crc) return crc;
  int dev = 0, cus = 256;
  PK2_HIP(hipGetDevice(&dev));
  PK2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  PK2_HIP(hipMemsetAsync(g_ctl, 0, sizeof(PersistCtl), stream));
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(y), (int)kSentinelBits, (size_t)T * B * D * H, stream));
  PersistFwdParams p{gx, whh, bhh, y, gates, cells, B, T, D};
  // one workgroup per CU: every XCD receives its 32, the ones on XCD 0 (and 1) take the roles, the rest return at once
  if (B <= 4) hipLaunchKernelGGL(lstm_fwd_persist<1>, dim3(std::max(cus, 8 * kPWgs)), dim3(256), 0, stream, p, g_ctl);
  else hipLaunchKernelGGL(lstm_fwd_persist<2>, dim3(std::max(cus, 8 * kPWgs)), dim3(256), 0, stream, p, g_ctl);
#ifdef PK2_PERSIST_PROFILE
  hipLaunchKernelGGL(pp_print, dim3(1), dim3(1), 0, stream, T);
#endif
  PK2_LAUNCH_CHECK();
  if (g_persist_state_pd.ref() < 0) {             // first use on this device: verify that the roles were filled and nobody timed out
    PersistCtl h;
    PK2_HIP(hipMemcpyAsync(&h, g_ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
    PK2_HIP(hipStreamSynchronize(stream));
    bool ok = h.abort == 0;
    for (int d = 0; d < D; ++d) ok = ok && h.reg[d] >= (unsigned)kPWgs;
    g_persist_state_pd.ref() = ok ? 1 : 0;
    if (!ok) return PK2_OK;              // caller falls back to the step kernels (and keeps doing so)
  }
  *ran = true;
  return PK2_OK;
}

int lstm_bwd_persist_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                            int D, float* dgx, float* mailboxes, hipStream_t stream, bool* ran) {
  *ran = false;
  if (g_persist_state_pd.ref() != 1) return PK2_OK;   // the forward pass verifies the device first
  PersistCtl* g_ctl = nullptr;
  int crc = get_ctl(stream, &g_ctl);
  if (crc) return crc;
  int dev = 0, cus = 256;
  PK2_HIP(hipGetDevice(&dev));
  PK2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  PK2_HIP(hipMemsetAsync(g_ctl, 0, sizeof(PersistCtl), stream));
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(mailboxes), (int)kSentinelBits, lstm_bwd_persist_mailbox_floats(D), stream));   // (sized for 8 batch rows)
  PersistBwdParams p{dy, whh, gates, cells, dgx, mailboxes, B, T, D};
  if (B <= 4) hipLaunchKernelGGL(lstm_bwd_persist<1>, dim3(std::max(cus, 8 * kPWgs)), dim3(256), 0, stream, p, g_ctl);
  else hipLaunchKernelGGL(lstm_bwd_persist<2>, dim3(std::max(cus, 8 * kPWgs)), dim3(256), 0, stream, p, g_ctl);
#ifdef PK2_PERSIST_PROFILE
  hipLaunchKernelGGL(ppb_print, dim3(1), dim3(1), 0, stream, T);
#endif
  PK2_LAUNCH_CHECK();
  *ran = true;
  return PK2_OK;
}

size_t lstm_bwd_persist_mailbox_floats(int D) { return (size_t)D * 2 * kPWgs * kPWgs * 128; }   // 16 units x 8 batch rows per writer

int lstm_persist_status(unsigned* abort_flag) {
  unsigned any = 0;
  for (auto& kv : g_ctls) {
    PersistCtl h{};
    PK2_HIP(hipMemcpy(&h, kv.second, sizeof(h), hipMemcpyDeviceToHost));
    any |= h.abort;
  }
  *abort_flag = any;
  return PK2_OK;
}

}  // namespace pk2
