// Persistent recurrence of one (bi)directional LSTM layer for small batches (B <= 4, H = 512) on gfx950.
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

#include "common.h"
#include "lstm_persist.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef PK2_PERSIST_DUAL
#define PK2_PERSIST_DUAL 1
#endif
constexpr unsigned kSentinelBits = 0x7fc0dead;     // a NaN payload no LSTM output can take
constexpr int kPH = 512;                           // hidden size served
constexpr int kPUnits = 16;                        // hidden units per workgroup
constexpr int kPWgs = kPH / kPUnits;               // workgroups per direction = CUs of an XCD
constexpr long long kSpinTicks = 20 * 1000 * 100;  // 20 ms of the 100 MHz wall clock before a poll gives up

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
#define PP_T(k) do { if (tid == 0 && s_rank == 0 && d == 0) { const long long n_ = clock64(); g_pp[k] += (unsigned long long)(n_ - pp_last); pp_last = n_; } } while (0)
__global__ void pp_print(int steps) {
  printf("lstm_fwd_persist rank 0, shader cycles per step over %d steps: gx prefetch issue %llu | h gathered into LDS %llu | barrier %llu | 128 MFMA + reduce %llu | transpose + gates + stores %llu\n",
         steps, g_pp[0] / steps, g_pp[1] / steps, g_pp[2] / steps, g_pp[3] / steps, g_pp[4] / steps);
  for (int k = 0; k < 8; ++k) g_pp[k] = 0;
}
#else
#define PP_T(k) do { } while (0)
#endif

__global__ void __launch_bounds__(256) lstm_fwd_persist(PersistFwdParams p, PersistCtl* ctl) {
  constexpr int H = kPH;
  // h_{t-1} of the direction, double buffered, as [batch row][k-phase][128 + 4]: the 16 (batch row, k-phase) streams the
  // MFMA lanes read side by side start 16 bytes apart modulo the 64 banks (unpadded they all start on bank 0: measured
  // 1.9 us per step in 16-way conflicts)
  constexpr int kPhasePitch = 132, kRowPitch = 4 * kPhasePitch;
  __shared__ __attribute__((aligned(16))) float hs[2][4 * kRowPitch];
  __shared__ __attribute__((aligned(16))) float tr[4][16][4];     // per-wave transposition of the 16 (gate, batch) sums
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
  // gate-math role (lanes 0..15 of every wave): unit u0 + lane/4, batch row lane%4
  const int gu = u0 + (lane >> 2), gc = lane & 3;
  const bool gate_lane = lane < 16 && gc < B;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (gate_lane && p.bhh) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[k] = p.bhh[(size_t)d * 4 * H + (size_t)k * H + gu];
  }
  float cstate = 0.f;
  bool timed_out = false;
  const size_t yrow = (size_t)D * H;
  float gxn[4] = {0.f, 0.f, 0.f, 0.f};             // input projection (+ bias) of the step about to run
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
    // ---- gather h_{t-1} of this direction: [4][H] floats = 512 granules of 16 bytes, two per thread -------------------
    {
      const int c0 = tid >> 7, c1 = c0 + 2, k4 = (tid & 127) * 4;          // granules tid and tid + 256
      u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
      bool timed_out_local = false;
      const bool need0 = step > 0 && c0 < B, need1 = step > 0 && c1 < B;
      const float* src0 = p.y + ((size_t)tp * B + c0) * yrow + (size_t)d * H + k4;
      const float* src1 = p.y + ((size_t)tp * B + c1) * yrow + (size_t)d * H + k4;
      if (need0 || need1) {
        long long t0 = 0;                      // the wall clock is only read once a poll has been spinning for a while
        unsigned spins = 0;
        bool bad;
        do {
          // both loads in flight, one wait
          if (need0 && need1 && PK2_PERSIST_DUAL) {
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
              timed_out = true; (void)timed_out_local;
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
    load_gx(step + 1);
    // ---- recurrent product: D_blk[r][c] += W[g][u0+r][k] * h[c][k] over the lane block's 128 k's ---------------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
      // all of the lane's h values first (32 LDS reads in flight), then four independent accumulator chains: a
      // dependent 4x4x1 MFMA waits ~4x its issue time for its predecessor (measured: 128 chained = 1.9 us)
      const float* hrow = &hs[buf][q4 * kRowPitch + kp * kPhasePitch];
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
      acc = (a0 + a1) + (a2 + a3);
      // the four k-phases of a (gate, batch row) sit 16 lanes apart
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[r] = v;
      }
    }
    PP_T(3);
    // lane 4g + c (< 16) holds the sums of gate g, batch row c for units u0..u0+3: transpose to (unit, batch row) lanes
    if (lane < 16) *reinterpret_cast<f32x4*>(&tr[w][lane][0]) = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (gate_lane) {
      const int ul = lane >> 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) pre[k] += tr[w][4 * k + gc][ul];
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
}

// ---- host -------------------------------------------------------------------------------------------------------------
static PersistCtl* g_ctl = nullptr;      // device
static int g_persist_state = -1;         // -1 untested, 0 unusable, 1 verified on this device

bool lstm_persist_wanted(int B, int H, int D) {
  const char* env = getenv("PK2_LSTM_PERSIST");
  if (env && atoi(env) == 0) return false;
  return g_persist_state != 0 && H == kPH && B >= 1 && B <= 4 && (D == 1 || D == 2);
}

int lstm_fwd_persist_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                            float* gates, float* cells, hipStream_t stream, bool* ran) {
  *ran = false;
  if (!g_ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&g_ctl), sizeof(PersistCtl)));
  int dev = 0, cus = 256;
  PK2_HIP(hipGetDevice(&dev));
  PK2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  PK2_HIP(hipMemsetAsync(g_ctl, 0, sizeof(PersistCtl), stream));
  PK2_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(y), (int)kSentinelBits, (size_t)T * B * D * H, stream));
  PersistFwdParams p{gx, whh, bhh, y, gates, cells, B, T, D};
  // one workgroup per CU: every XCD receives its 32, the ones on XCD 0 (and 1) take the roles, the rest return at once
  hipLaunchKernelGGL(lstm_fwd_persist, dim3(std::max(cus, 8 * kPWgs)), dim3(256), 0, stream, p, g_ctl);
#ifdef PK2_PERSIST_PROFILE
  hipLaunchKernelGGL(pp_print, dim3(1), dim3(1), 0, stream, T);
#endif
  PK2_LAUNCH_CHECK();
  if (g_persist_state < 0) {             // first use on this device: verify that the roles were filled and nobody timed out
    PersistCtl h;
    PK2_HIP(hipMemcpyAsync(&h, g_ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
    PK2_HIP(hipStreamSynchronize(stream));
    bool ok = h.abort == 0;
    for (int d = 0; d < D; ++d) ok = ok && h.reg[d] >= (unsigned)kPWgs;
    g_persist_state = ok ? 1 : 0;
    if (!ok) return PK2_OK;              // caller falls back to the step kernels (and keeps doing so)
  }
  *ran = true;
  return PK2_OK;
}

int lstm_persist_status(unsigned* abort_flag) {
  PersistCtl h{};
  if (g_ctl) PK2_HIP(hipMemcpy(&h, g_ctl, sizeof(h), hipMemcpyDeviceToHost));
  *abort_flag = h.abort;
  return PK2_OK;
}

}  // namespace pk2
