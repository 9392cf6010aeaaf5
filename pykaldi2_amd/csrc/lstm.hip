// Recurrent part of one (bi)directional LSTM layer on gfx950: forward and backward through time.
//
// Replaces the cuDNN RNN under nn.LSTM (reference models/lstm.py:49-58): gate order i,f,g,o,
// h0 = c0 = 0, every sequence runs over all T frames (the reference does not pack, so padding
// frames are processed too).  The input projections (x W_ih^T + b_ih for all frames) and all
// weight gradients are large GEMMs done by pk2_gemm_f32; this file is the serial part.
//
// One kernel launch per time step, both directions in the same launch (blockIdx.y); the T
// launches of a layer are replayed from cached hipGraphs (step_graph.h), so the host is not in
// the loop.  The recurrent product h_{t-1} W_hh^T runs on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact f32):
//  * forward: a workgroup owns 4 hidden units = 16 gate rows (one 16-wide MFMA N tile); its 4
//    wavefronts split K = H, each holding its slice of those 16 W_hh rows in 32 VGPRs; batch rows
//    are the MFMA M dimension (tiles of 16); partial tiles meet in LDS, then the gate
//    activations, c_t and h_t are computed in the same kernel (fused pointwise).  The pointwise
//    operands (input projection, bias, c_{t-1}) are fetched before the MFMA phase so their
//    latency hides behind it.
//  * backward: a workgroup owns 4 hidden units (4 live columns of the N tile) and splits K = 4H
//    over 16 wavefronts; d h_{t-1} = dgates_t W_hh is fused with the gate derivative of step t-1.
// W_hh slices are re-read from L2 every step (4 MB per direction stays L2 resident; each
// workgroup always reads the same slice, and consecutive launches place block b on XCD b%8).
// All activations are time-major ([T][B][...]) so "previous step" is a constant row offset.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gemm_tile.h"
#include "lstm_persist.h"
#include "step_graph.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFwdThreads = 256;    // 4 waves, K = H split 4 ways
constexpr int kFwdUnits = 4;        // hidden units per workgroup (x4 gates = 16 MFMA columns)
constexpr int kBwdWavesDefault = 16; // K = 4H split over this many wavefronts (PK2_LSTM_BWD_WAVES = 8 | 16)
constexpr int kBwdUnits = 4;        // hidden units per workgroup (4 of the 16 MFMA columns carry data:
                                    // the matrix work is negligible at these batch sizes, and 4x more
                                    // workgroups spread the W_hh^T read over the whole chip)

// Gate nonlinearities on v_exp_f32 / v_rcp_f32 (about 1 ulp each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)),
// tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)); absolute error ~1e-7 (libm's expf / tanhf cost 0.27 us of a 4.2 us step, measured).
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

struct LstmFwdParams {
  const float* gx;    // [T][B][D*4H]
  const float* whh;   // [D][4H][H]
  const float* bhh;   // [D][4H] recurrent bias (may be null)
  float* y;           // [T][B][D*H]
  float* gates;       // [D][T][B][4H]
  float* cells;       // [D][T][B][H]
  int B, T, H, D;
  const float* whh_units;  // [D][H][4][H] rows regrouped (unit, gate) for the large-batch kernel (or null)
};

// KS = number of 4-wide MFMA k-steps per wave (H / 4 waves / 4).
// Batch rows are processed in groups of kFwdTileGroup MFMA M-tiles (64 rows): all matrix work of a
// group, ONE barrier, then the gate math of the group's 64 x 4 (row, unit) pairs on all 256 threads.
constexpr int kFwdTileGroup = 4;

#ifdef PK2_LSTM_PROFILE
// Phase timers of the forward step kernel (shader clock cycles), thread 0 of workgroup (0,0,0), summed over the steps.
__device__ unsigned long long g_lstm_prof[8];
#define LSTM_T(k, wait) do { wait; if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { const long long now_ = clock64(); atomicAdd(&g_lstm_prof[k], (unsigned long long)(now_ - prof_last_)); prof_last_ = now_; } } while (0)
#define LSTM_WAIT_ALL asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define LSTM_WAIT_NONE asm volatile("" ::: "memory")
__global__ void lstm_prof_print(int steps) {
  printf("lstm_fwd_step wg0 avg shader cycles per step over %d steps: params+counter %llu | global loads returned %llu | mfma + partial tiles in LDS %llu | barrier %llu | gate math, stores issued %llu | stores drained %llu\n",
         steps, g_lstm_prof[0] / steps, g_lstm_prof[1] / steps, g_lstm_prof[2] / steps, g_lstm_prof[3] / steps, g_lstm_prof[4] / steps, g_lstm_prof[5] / steps);
  for (int k = 0; k < 8; ++k) g_lstm_prof[k] = 0;
}
#else
#define LSTM_T(k, wait) do { } while (0)
#endif

template <int KS>
__global__ void __launch_bounds__(kFwdThreads) lstm_fwd_step(const LstmFwdParams* __restrict__ pp,
                                                             const StepCounter* __restrict__ cnt, int local) {
  __shared__ float part[kFwdTileGroup][4][16][17];   // [tile][wave] partial 16x16 tiles (padded)
#ifdef PK2_LSTM_PROFILE
  long long prof_last_ = clock64();
#endif
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const LstmFwdParams p = *pp;
  LSTM_T(0, LSTM_WAIT_ALL);
  const int d = blockIdx.y;
  const int u0 = blockIdx.x * kFwdUnits;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int H = p.H, B = p.B, T = p.T, D = p.D;
  const int t = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? t - 1 : t + 1;      // previous step in processing order
  const bool first = step == 0;
  const int li = lane & 15, kq = lane >> 4;

  // B operand: column j = li -> gate g = j/4, unit u0 + j%4 -> row g*H + u0 + j%4 of W_hh[d]
  f32x4 wf[KS / 4];
  const int kbase = w * (KS * 4) + kq * KS;   // this lane's contiguous k-run of length KS
  if (!first) {
    const float* wrow = p.whh + ((size_t)d * 4 * H + (size_t)(li >> 2) * H + u0 + (li & 3)) * H + kbase;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + q * 4);
  }
  const size_t yrow = (size_t)D * H;
  const int ntiles = (B + 15) / 16;
  {
    const int mt0 = blockIdx.z * kFwdTileGroup;   // one group of M-tiles per workgroup (grid.z covers the batch)
    const int ng = min(kFwdTileGroup, ntiles - mt0);
    // gate-math operands of thread (tile tg, batch row i, unit u), fetched before the matrix phase
    const int tg = tid >> 6, pi_ = (tid >> 2) & 15, pu = tid & 3;
    const int pb = (mt0 + tg) * 16 + pi_;
    const bool pw_active = tg < ng && pb < B;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float cprev = 0.f;
    if (pw_active) {
      const float* gxr = p.gx + ((size_t)t * B + pb) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + u0 + pu;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pre[g] = gxr[(size_t)g * H];
        if (p.bhh) pre[g] += p.bhh[(size_t)d * 4 * H + (size_t)g * H + u0 + pu];
      }
      if (!first) cprev = p.cells[(((size_t)d * T + tp) * B + pb) * H + u0 + pu];
    }
    if (!first) {
      for (int g2 = 0; g2 < ng; ++g2) {
        const int b = (mt0 + g2) * 16 + li;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 af[KS / 4];
        if (b < B) {
          const float* hrow = p.y + ((size_t)tp * B + b) * yrow + (size_t)d * H + kbase;
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(hrow + q * 4);
        } else {
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) af[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        LSTM_T(1, LSTM_WAIT_ALL);
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][0], wf[q][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][1], wf[q][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][2], wf[q][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][3], wf[q][3], acc1, 0, 0, 0);
        }
        // C layout: row (batch) = (lane>>4)*4 + r, col = lane&15
#pragma unroll
        for (int r = 0; r < 4; ++r) part[g2][w][kq * 4 + r][li] = acc0[r] + acc1[r];
      }
    }
    LSTM_T(2, LSTM_WAIT_ALL);
    __syncthreads();
    LSTM_T(3, LSTM_WAIT_NONE);
    if (pw_active) {
      if (!first) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pre[g] += (part[tg][0][pi_][g * 4 + pu] + part[tg][1][pi_][g * 4 + pu]) +
                    (part[tg][2][pi_][g * 4 + pu] + part[tg][3][pi_][g * 4 + pu]);
      }
      const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]), gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
      const float c = fg * cprev + ig * gg;
      const float h = og * fast_tanh(c);
      p.cells[(((size_t)d * T + t) * B + pb) * H + u0 + pu] = c;
      p.y[((size_t)t * B + pb) * yrow + (size_t)d * H + u0 + pu] = h;
      float* gr = p.gates + (((size_t)d * T + t) * B + pb) * 4 * H + u0 + pu;
      gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
    }
    LSTM_T(4, LSTM_WAIT_NONE);
    LSTM_T(5, LSTM_WAIT_ALL);
  }
}

struct LstmBwdParams {
  const float* dy;     // [T][B][D*H]
  const float* whhT;   // [D][H][4H]   (transposed once per call)
  const float* gates;  // [D][T][B][4H]
  const float* cells;  // [D][T][B][H]
  float* dgx;          // [T][B][D*4H]
  float* dc;           // [D][B][H] running dL/dc carried between steps
  int B, T, H, D;
  const float* whh;    // [D][4H][H] untransposed (large-batch kernels)
  float* dh_part;      // [kBigSplitK][D][B][H] split-K partial sums of dgates W_hh (large-batch kernels)
};

// KS = 4-wide MFMA k-steps per wave (4H / 16 waves / 4).  Batch rows in groups of kBwdTileGroup M-tiles.
constexpr int kBwdTileGroup = 2;

template <int KS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) lstm_bwd_step(const LstmBwdParams* __restrict__ pp,
                                                            const StepCounter* __restrict__ cnt, int local) {
  __shared__ float part[kBwdTileGroup][WAVES][16][17];
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const LstmBwdParams p = *pp;
  const int d = blockIdx.y;
  const int k0 = blockIdx.x * kBwdUnits;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int H = p.H, B = p.B, T = p.T, D = p.D;
  // backward visits the forward steps in reverse order
  const int fstep = T - 1 - step;                  // forward step index being differentiated
  const int t = d == 0 ? fstep : T - 1 - fstep;    // its frame
  const int tn = d == 0 ? t + 1 : t - 1;           // frame of the step after it (already done)
  const int tp = d == 0 ? t - 1 : t + 1;           // frame of the step before it
  const bool last_fwd = step == 0;                 // no recurrent gradient flows in
  const bool first_fwd = fstep == 0;               // c_{prev} = 0
  const int li = lane & 15, kq = lane >> 4;
  const int G4 = 4 * H;

  f32x4 wf[KS / 4];
  const int rbase = w * (KS * 4) + kq * KS;        // run of gate rows r handled by this lane
#pragma unroll
  for (int q = 0; q < KS / 4; ++q) wf[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!last_fwd && li < kBwdUnits) {
    const float* wrow = p.whhT + ((size_t)d * H + k0 + li) * G4 + rbase;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + q * 4);
  }
  const int ntiles = (B + 15) / 16;
  {
    const int mt0 = blockIdx.z * kBwdTileGroup;   // one group of M-tiles per workgroup
    const int ng = min(kBwdTileGroup, ntiles - mt0);
    // pointwise operands of thread (tile tg, row i, unit j) fetched before the MFMA phase
    const int tg = tid >> 6, pi_ = (tid >> 2) & 15, pj = tid & 3;
    const int pb = (mt0 + tg) * 16 + pi_, pk = k0 + pj;
    const bool pw_active = tg < ng && pb < B;
    float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f, dcin = 0.f;
    if (pw_active) {
      dh = p.dy[((size_t)t * B + pb) * ((size_t)D * H) + (size_t)d * H + pk];
      const float* gr = p.gates + (((size_t)d * T + t) * B + pb) * G4 + pk;
      ig = gr[0]; fg = gr[(size_t)H]; gg = gr[(size_t)2 * H]; og = gr[(size_t)3 * H];
      c = p.cells[(((size_t)d * T + t) * B + pb) * H + pk];
      if (!first_fwd) cprev = p.cells[(((size_t)d * T + tp) * B + pb) * H + pk];
      if (!last_fwd) dcin = p.dc[((size_t)d * B + pb) * H + pk];
    }
    if (!last_fwd) {
      for (int g2 = 0; g2 < ng; ++g2) {
        const int b = (mt0 + g2) * 16 + li;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 af[KS / 4];
        if (b < B) {
          const float* grow = p.dgx + ((size_t)tn * B + b) * ((size_t)D * G4) + (size_t)d * G4 + rbase;
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(grow + q * 4);
        } else {
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) af[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][0], wf[q][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][1], wf[q][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][2], wf[q][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][3], wf[q][3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[g2][w][kq * 4 + r][li] = acc0[r] + acc1[r];
      }
    }
    __syncthreads();
    if (pw_active) {
      if (!last_fwd) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) s += part[tg][ww][pi_][pj];
        dh += s;
      }
      const float tc = fast_tanh(c);
      const float dcv = dcin + dh * og * (1.f - tc * tc);
      p.dc[((size_t)d * B + pb) * H + pk] = dcv * fg;
      float* o = p.dgx + ((size_t)t * B + pb) * ((size_t)D * G4) + (size_t)d * G4 + pk;
      o[0] = dcv * gg * ig * (1.f - ig);
      o[(size_t)H] = dcv * cprev * fg * (1.f - fg);
      o[(size_t)2 * H] = dcv * ig * (1.f - gg * gg);
      o[(size_t)3 * H] = dh * tc * og * (1.f - og);
    }
  }
}

// ----------------------------------------------------------------------------------------
// Small-batch backward step on v_mfma_f32_4x4x1_16b_f32.
// With a batch of 4 and 4 hidden units per workgroup the product dh[b][k] = sum_r dgates[b][r] W_hh[r][k] is a
// 4 x 2048 x 4 job: a 16x16x4 MFMA tile would be 15/16 padding (the 16-wave kernel above spends 1.7 us per step in
// the matrix pipe at 1/16 utilisation).  The 4x4x1 instruction computes 16 INDEPENDENT 4x4 outer products per
// issue (lane l feeds block l/4: row/column l%4; layout verified with tools/ubench/mfma4x4_layout.hip), so the 16
// blocks take 16 different gate rows r: every lane streams its own run of H/16 consecutive r's of one batch row
// and one unit (all 64 lanes load, 8 cycles per instruction, 4 waves instead of 16), the 16 block results are
// added with four shuffle rounds, the four waves meet in LDS.  Batches of 5..31 rows loop over groups of 4.
// ----------------------------------------------------------------------------------------
constexpr int kX4MaxGroups = 8;

template <int H>
__global__ void __launch_bounds__(256) lstm_bwd_step_x4(const LstmBwdParams* __restrict__ pp,
                                                        const StepCounter* __restrict__ cnt, int local) {
  constexpr int NR = H / 16;                       // gate rows per lane = MFMAs per wave and batch group
  constexpr int G4 = 4 * H;
  __shared__ float part[kX4MaxGroups][4][4][4];    // [batch group][wave][batch row i][unit j]
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const LstmBwdParams p = *pp;
  const int d = blockIdx.y;
  const int k0 = blockIdx.x * 4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int blk = lane >> 2, q = lane & 3;
  const int B = p.B, T = p.T, D = p.D;
  const int fstep = T - 1 - step;
  const int t = d == 0 ? fstep : T - 1 - fstep;
  const int tn = d == 0 ? t + 1 : t - 1;
  const int tp = d == 0 ? t - 1 : t + 1;
  const bool last_fwd = step == 0, first_fwd = fstep == 0;
  const int ngroups = (B + 3) >> 2;
  const int rbase = w * H + blk * NR;              // wave w covers gate rows [w*H, (w+1)*H), block blk a run of NR

  f32x4 wf[NR / 4];                                // W_hh^T[k0 + q][rbase ..]
  if (!last_fwd) {
    const float* wrow = p.whhT + ((size_t)d * H + k0 + q) * G4 + rbase;
#pragma unroll
    for (int n = 0; n < NR / 4; ++n) wf[n] = *reinterpret_cast<const f32x4*>(wrow + n * 4);
  }
  // pointwise operands of thread (group pg, batch row pi, unit pj), fetched before the matrix phase
  const int pg = tid >> 4, pi_ = (tid >> 2) & 3, pj = tid & 3;
  const int pb = pg * 4 + pi_, pk = k0 + pj;
  const bool pw_active = pg < ngroups && pb < B;
  float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f, dcin = 0.f;
  if (pw_active) {
    dh = p.dy[((size_t)t * B + pb) * ((size_t)D * H) + (size_t)d * H + pk];
    const float* gr = p.gates + (((size_t)d * T + t) * B + pb) * G4 + pk;
    ig = gr[0]; fg = gr[(size_t)H]; gg = gr[(size_t)2 * H]; og = gr[(size_t)3 * H];
    c = p.cells[(((size_t)d * T + t) * B + pb) * H + pk];
    if (!first_fwd) cprev = p.cells[(((size_t)d * T + tp) * B + pb) * H + pk];
    if (!last_fwd) dcin = p.dc[((size_t)d * B + pb) * H + pk];
  }
  if (!last_fwd) {
    for (int g = 0; g < ngroups; ++g) {
      const int b = g * 4 + q;
      f32x4 af[NR / 4];
      if (b < B) {
        const float* grow = p.dgx + ((size_t)tn * B + b) * ((size_t)D * G4) + (size_t)d * G4 + rbase;
#pragma unroll
        for (int n = 0; n < NR / 4; ++n) af[n] = *reinterpret_cast<const f32x4*>(grow + n * 4);
      } else {
#pragma unroll
        for (int n = 0; n < NR / 4; ++n) af[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < NR / 4; ++n) {
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[n][0], wf[n][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[n][1], wf[n][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[n][2], wf[n][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[n][3], wf[n][3], acc1, 0, 0, 0);
      }
      // register r of lane l = D_block(l/4)[row r][column l%4]: add the 16 blocks
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc0[r] + acc1[r];
        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 4) part[g][w][r][lane] = v;
      }
    }
  }
  __syncthreads();
  if (pw_active) {
    if (!last_fwd) dh += (part[pg][0][pi_][pj] + part[pg][1][pi_][pj]) + (part[pg][2][pi_][pj] + part[pg][3][pi_][pj]);
    const float tc = fast_tanh(c);
    const float dcv = dcin + dh * og * (1.f - tc * tc);
    p.dc[((size_t)d * B + pb) * H + pk] = dcv * fg;
    float* o = p.dgx + ((size_t)t * B + pb) * ((size_t)D * G4) + (size_t)d * G4 + pk;
    o[0] = dcv * gg * ig * (1.f - ig);
    o[(size_t)H] = dcv * cprev * fg * (1.f - fg);
    o[(size_t)2 * H] = dcv * ig * (1.f - gg * gg);
    o[(size_t)3 * H] = dh * tc * og * (1.f - og);
  }
}

// ----------------------------------------------------------------------------------------
// Large-batch step kernels (B >= kBigBatch, e.g. the CE configuration 256 x 80).  At these sizes the
// recurrence is a real GEMM ([B, H] x [H, 4H] per direction and step), so it is tiled like one: 64 x 64
// output tiles through the LDS staging of gemm_tile.h (v_mfma_f32_32x32x2_f32), W_hh rows regrouped so that
// a tile's 64 columns are the 4 gates of 16 hidden units, and the gate math fused into the epilogue.
// ----------------------------------------------------------------------------------------
constexpr int kBigBatch = 32;
constexpr int kBigSplitK = 4;

// whh_units[d][u][g][k] = whh[d][g*H + u][k]
__global__ void __launch_bounds__(256) regroup_whh_units(const float* __restrict__ whh, float* __restrict__ out, int H) {
  const int d = blockIdx.z, u = blockIdx.x, g = blockIdx.y;
  const float* src = whh + ((size_t)d * 4 * H + (size_t)g * H + u) * H;
  float* dst = out + (((size_t)d * H + u) * 4 + g) * H;
  for (int k = threadIdx.x; k < H; k += 256) dst[k] = src[k];
}

__global__ void __launch_bounds__(256) lstm_fwd_step_big(const LstmFwdParams* __restrict__ pp,
                                                         const StepCounter* __restrict__ cnt, int local) {
  constexpr int LD = Geo<1>::LDK;     // both operands are k-contiguous in memory
  __shared__ __attribute__((aligned(16))) float As[2][BK * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LD];
  __shared__ float Cs[64][65];
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
#ifdef PK2_BIGSTEP_PROFILE
  const long long bs_t0 = wall_clock64();
#endif
  const LstmFwdParams p = *pp;
  const int d = blockIdx.y, u0 = blockIdx.x * 16, m0 = blockIdx.z * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int H = p.H, B = p.B, T = p.T, D = p.D;
  const int t = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? t - 1 : t + 1;
  const bool first = step == 0;
  const size_t yrow = (size_t)D * H;
  // gate-math operands of the 4 (row, unit) items of this thread: ISSUED before the matrix phase, first USED behind it
  // (adding the bias here made the kernel wait for them before it had a single k-slab in flight: in-kernel timers showed
  // 8 us between entry and the main loop)
  float pre[4][4], bias[4][4], cprev[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + j * 256, i = idx >> 4, u = idx & 15, b = m0 + i;
    cprev[j] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { pre[j][g] = 0.f; bias[j][g] = 0.f; }
    if (b < B) {
      const float* gxr = p.gx + ((size_t)t * B + b) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + u0 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pre[j][g] = gxr[(size_t)g * H];
        if (p.bhh) bias[j][g] = p.bhh[(size_t)d * 4 * H + (size_t)g * H + u0 + u];
      }
      if (!first) cprev[j] = p.cells[(((size_t)d * T + tp) * B + b) * H + u0 + u];
    }
  }
  if (!first) {
    const float* A = p.y + (size_t)tp * B * yrow + (size_t)d * H;               // h_{prev}: row b, k contiguous
    const float* Bw = p.whh_units + ((size_t)d * H + u0) * 4 * H;              // 64 rows (unit, gate), k contiguous
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    // the step's matrix phase is one 64 x 64 x H block tile: the GEMM's pipelined loop (six k-slabs in flight, LDS-only
    // barriers -- one workgroup per CU has to cover the round trips to memory by itself; the weights come from memory
    // again in every launch).  (Two barriers and one 64-k stage in flight per iteration: 30 us per step.)
    f32x16 acc1[1][1];
#ifdef PK2_BIGSTEP_PROFILE
    const long long bs_t1 = wall_clock64();
#endif
    tile_mainloop<true, true, 1>(A, (int64_t)yrow, Bw, (int64_t)H, m0, 0, 0, H, B, 64, true, true, As, Bs, acc1);
#ifdef PK2_BIGSTEP_PROFILE
    if (tid == 0 && blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 1 && step == 40)
      printf("lstm_fwd_step_big step 40: entry -> main loop %lld, main loop %lld (10 ns ticks)\n", bs_t1 - bs_t0, wall_clock64() - bs_t1);
#endif
    const f32x16 acc = acc1[0][0];
    const int col = wn + (lane & 31), rh = 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[wm + (r & 3) + 8 * (r >> 2) + rh][col] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + j * 256, i = idx >> 4, u = idx & 15, b = m0 + i;
    if (b < B) {
      float v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = (pre[j][g] + bias[j][g]) + (first ? 0.f : Cs[i][u * 4 + g]);
      const float ig = fast_sigmoid(v[0]), fg = fast_sigmoid(v[1]), gg = fast_tanh(v[2]), og = fast_sigmoid(v[3]);
      const float c = fg * cprev[j] + ig * gg;
      const float h = og * fast_tanh(c);
      p.cells[(((size_t)d * T + t) * B + b) * H + u0 + u] = c;
      p.y[((size_t)t * B + b) * yrow + (size_t)d * H + u0 + u] = h;
      float* gr = p.gates + (((size_t)d * T + t) * B + b) * 4 * H + u0 + u;
      gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
    }
  }
#ifdef PK2_BIGSTEP_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 1 && step == 40)
    printf("lstm_fwd_step_big step 40: whole kernel %lld (10 ns ticks)\n", wall_clock64() - bs_t0);
#endif
}

// dh_part[s][d][b][k] = sum_{r in K-slice s} dgates[tn][b][r] * W_hh[d][r][k]   (64 x 64 tiles, split-K)
__global__ void __launch_bounds__(256) lstm_bwd_dh_big(const LstmBwdParams* __restrict__ pp,
                                                       const StepCounter* __restrict__ cnt, int local) {
  constexpr int LDA = Geo<1>::LDK, LDB = Geo<1>::LD;   // dgates rows are k-contiguous, W_hh is read [k][n]
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];
  const int step = cnt->base + local;
  if (step >= cnt->T || step == 0) return;     // the first backward step has no recurrent gradient
  const LstmBwdParams p = *pp;
  const int d = blockIdx.y / kBigSplitK, sk = blockIdx.y % kBigSplitK;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.z * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int H = p.H, B = p.B, T = p.T, D = p.D, G4 = 4 * H;
  const int fstep = T - 1 - step;
  const int t = d == 0 ? fstep : T - 1 - fstep;
  const int tn = d == 0 ? t + 1 : t - 1;
  const int klen = G4 / kBigSplitK, kbeg = sk * klen;
  const float* A = p.dgx + (size_t)tn * B * ((size_t)D * G4) + (size_t)d * G4 + kbeg;   // rows b, r contiguous
  const float* Bw = p.whh + (size_t)d * G4 * H + (size_t)kbeg * H;                      // [r][k]: (k=r, n) at r*H + n
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
  f32x16 acc1[1][1];
  tile_mainloop<true, false, 1>(A, (int64_t)D * G4, Bw, (int64_t)H, m0, n0, 0, klen, B, H, true, true, As, Bs, acc1);
  const f32x16 acc = acc1[0][0];
  float* out = p.dh_part + (((size_t)sk * D + d) * B) * H;
  const int col = n0 + wn + (lane & 31), rh = 4 * (lane >> 5);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = m0 + wm + (r & 3) + 8 * (r >> 2) + rh;
    if (b < B) out[(size_t)b * H + col] = acc[r];
  }
}

// Gate derivatives of forward step fstep from dh = dy + sum_s dh_part (one thread per (direction, row, unit)).
__global__ void __launch_bounds__(256) lstm_bwd_pointwise_big(const LstmBwdParams* __restrict__ pp,
                                                              const StepCounter* __restrict__ cnt, int local) {
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const LstmBwdParams p = *pp;
  const int H = p.H, B = p.B, T = p.T, D = p.D, G4 = 4 * H;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)D * B * H) return;
  const int k = (int)(idx % H), b = (int)((idx / H) % B), d = (int)(idx / ((int64_t)H * B));
  const int fstep = T - 1 - step;
  const int t = d == 0 ? fstep : T - 1 - fstep;
  const int tp = d == 0 ? t - 1 : t + 1;
  const bool last_fwd = step == 0, first_fwd = fstep == 0;
  float dh = p.dy[((size_t)t * B + b) * ((size_t)D * H) + (size_t)d * H + k];
  if (!last_fwd) {
#pragma unroll
    for (int s = 0; s < kBigSplitK; ++s) dh += p.dh_part[(((size_t)s * D + d) * B + b) * H + k];
  }
  const float* gr = p.gates + (((size_t)d * T + t) * B + b) * G4 + k;
  const float ig = gr[0], fg = gr[(size_t)H], gg = gr[(size_t)2 * H], og = gr[(size_t)3 * H];
  const float c = p.cells[(((size_t)d * T + t) * B + b) * H + k];
  const float cprev = first_fwd ? 0.f : p.cells[(((size_t)d * T + tp) * B + b) * H + k];
  float* dcp = p.dc + ((size_t)d * B + b) * H + k;
  const float tc = fast_tanh(c);
  const float dcv = (last_fwd ? 0.f : *dcp) + dh * og * (1.f - tc * tc);
  *dcp = dcv * fg;
  float* o = p.dgx + ((size_t)t * B + b) * ((size_t)D * G4) + (size_t)d * G4 + k;
  o[0] = dcv * gg * ig * (1.f - ig);
  o[(size_t)H] = dcv * cprev * fg * (1.f - fg);
  o[(size_t)2 * H] = dcv * ig * (1.f - gg * gg);
  o[(size_t)3 * H] = dh * tc * og * (1.f - og);
}

// whhT[d][k][r] = whh[d][r][k]
__global__ void __launch_bounds__(256) transpose_whh(const float* __restrict__ whh, float* __restrict__ whhT,
                                                     int H) {
  __shared__ float tile[32][33];
  const int d = blockIdx.z;
  const int r0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = whh + (size_t)d * 4 * H * H;
  float* dst = whhT + (size_t)d * 4 * H * H;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(size_t)(r0 + r) * H + k0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8) dst[(size_t)(k0 + k) * 4 * H + r0 + tx] = tile[tx][k];
}

static StepGraphs g_graphs;
static std::map<std::pair<int, hipStream_t>, ParamSlot<LstmFwdParams>> g_fwd_slots;
static std::map<std::pair<int, hipStream_t>, ParamSlot<LstmBwdParams>> g_bwd_slots;

template <typename K, typename P>
static void launch_step(K kernel, dim3 grid, dim3 block, hipStream_t s, const P* pb, const StepCounter* c, int j) {
  hipLaunchKernelGGL(kernel, grid, block, 0, s, pb, c, j);
}

}  // namespace pk2

using namespace pk2;

static bool lstm_h_ok(int H) { return H == 64 || H == 128 || H == 256 || H == 512 || H == 1024; }

extern "C" size_t pk2_lstm_fwd_workspace_floats(int32_t B, int32_t H, int32_t D) {
  return B >= kBigBatch ? (size_t)D * 4 * H * H : 0;
}

extern "C" int pk2_lstm_layer_fwd(const float* gx, const float* whh, const float* bhh, int32_t B, int32_t T,
                                  int32_t H, int32_t D, float* y, float* gates, float* cells, void* workspace,
                                  void* stream_) {
  PK2_REQUIRE(gx && whh && y && gates && cells && B > 0 && T > 0 && (D == 1 || D == 2), "lstm_fwd: bad args");
  PK2_REQUIRE(lstm_h_ok(H), "lstm_fwd: hidden size %d unsupported (64,128,256,512,1024)", H);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (lstm_seq_wanted(B, H, D)) {           // one launch, a (sequence, direction) pair per XCD (lstm_persist_seq.hip)
    bool ran = false;
    int prc = lstm_fwd_seq_launch(gx, whh, bhh, B, T, H, D, y, gates, cells, stream, &ran);
    if (prc) return prc;
    if (ran) return PK2_OK;
  }
  if (lstm_persist_wanted(B, H, D)) {       // one launch for the whole sequence (lstm_persist.hip)
    bool ran = false;
    int prc = lstm_fwd_persist_launch(gx, whh, bhh, B, T, H, D, y, gates, cells, stream, &ran);
    if (prc) return prc;
    if (ran) return PK2_OK;
  }
  if (lstm_big_wanted(B, H, D)) {           // large batches: one launch, W_hh slices resident in LDS (lstm_persist_big.hip)
    bool ran = false;
    int prc = lstm_fwd_big_launch(gx, whh, bhh, B, T, H, D, y, gates, cells, stream, &ran);
    if (prc) return prc;
    if (ran) return PK2_OK;
  }
  ParamSlot<LstmFwdParams>* slot;
  int rc = get_param_slot(g_fwd_slots, H * 4 + D, stream, &slot);
  if (rc) return rc;
  const bool big = B >= kBigBatch && workspace != nullptr && H % 64 == 0;
  float* wu = static_cast<float*>(workspace);
  if (big) hipLaunchKernelGGL(regroup_whh_units, dim3(H, 4, D), dim3(256), 0, stream, whh, wu, H);
  LstmFwdParams p{gx, whh, bhh, y, gates, cells, B, T, H, D, big ? wu : nullptr};
  hipLaunchKernelGGL(param_block_store<LstmFwdParams>, dim3(1), dim3(1), 0, stream, p, slot->params);
  if (big) {
    dim3 gridb(H / 16, D, (B + 63) / 64);
    const LstmFwdParams* pbb = slot->params;
    const StepCounter* cb = slot->counter;
    char keyb[64];
    snprintf(keyb, sizeof(keyb), "lstm_fwd_big_H%d_D%d_Z%d_%p", H, D, (int)gridb.z, (void*)stream);
    rc = g_graphs.run(keyb, T, slot->counter, stream, [&](hipStream_t s, int j) {
      hipLaunchKernelGGL(lstm_fwd_step_big, gridb, dim3(256), 0, s, pbb, cb, j);
    });
    if (rc) return rc;
    PK2_LAUNCH_CHECK();
    return PK2_OK;
  }
  const int zf = ((B + 15) / 16 + kFwdTileGroup - 1) / kFwdTileGroup;
  dim3 grid(H / kFwdUnits, D, zf), block(kFwdThreads);
  const LstmFwdParams* pb = slot->params;
  const StepCounter* c = slot->counter;
  char key[64];
  snprintf(key, sizeof(key), "lstm_fwd_H%d_D%d_Z%d_%p", H, D, zf, (void*)stream);
  rc = g_graphs.run(key, T, slot->counter, stream, [&](hipStream_t s, int j) {
    switch (H) {
      case 64: launch_step(lstm_fwd_step<4>, grid, block, s, pb, c, j); break;
      case 128: launch_step(lstm_fwd_step<8>, grid, block, s, pb, c, j); break;
      case 256: launch_step(lstm_fwd_step<16>, grid, block, s, pb, c, j); break;
      case 512: launch_step(lstm_fwd_step<32>, grid, block, s, pb, c, j); break;
      default: launch_step(lstm_fwd_step<64>, grid, block, s, pb, c, j); break;
    }
  });
  if (rc) return rc;
#ifdef PK2_LSTM_PROFILE
  hipLaunchKernelGGL(lstm_prof_print, dim3(1), dim3(1), 0, stream, T - 1);
#endif
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Test / monitoring hook: 1 in *abort_flag when a poll of the persistent recurrence (lstm_persist.hip) has timed out
// (synchronises the device).
extern "C" int pk2_lstm_persist_status(uint32_t* abort_flag) {
  PK2_REQUIRE(abort_flag, "lstm_persist_status: null pointer");
  unsigned f = 0;
  int rc = lstm_persist_status(&f);
  unsigned f2 = 0, f3 = 0;
  if (!rc) rc = lstm_seq_status(&f2);
  if (!rc) rc = lstm_big_status(&f3);
  f |= f2 | f3;
  *abort_flag = f;
  return rc;
}

extern "C" size_t pk2_lstm_bwd_scratch_floats(int32_t B, int32_t H, int32_t D) {
  return (size_t)D * H * 4 * H + (size_t)D * B * H + 64 + (B >= kBigBatch ? (size_t)kBigSplitK * D * B * H : 0);
}

static int lstm_layer_bwd_impl(const float* dy, const float* whh, const float* gates, const float* cells, int32_t B, int32_t T,
                               int32_t H, int32_t D, float* dgx, float* scratch, float* dbias_ih, float* dbias_hh,
                               int32_t* bias_done, void* stream_);

extern "C" int pk2_lstm_layer_bwd(const float* dy, const float* whh, const float* gates, const float* cells,
                                  int32_t B, int32_t T, int32_t H, int32_t D, float* dgx, float* scratch,
                                  void* stream_) {
  return lstm_layer_bwd_impl(dy, whh, gates, cells, B, T, H, D, dgx, scratch, nullptr, nullptr, nullptr, stream_);
}

// The same, with the bias gradients where the kernel can produce them on the way (the one-launch recurrence of a
// (sequence, direction) pair per XCD sums its d gates over the frames in registers): dbias_ih / dbias_hh [D][4H] are
// ACCUMULATED into (+=, the caller zeroes them); *bias_done = 0 when the path taken did not fill them -- the caller then
// takes the column sums of dgx itself (pk2_colsum_f32).
extern "C" int pk2_lstm_layer_bwd_bias(const float* dy, const float* whh, const float* gates, const float* cells,
                                       int32_t B, int32_t T, int32_t H, int32_t D, float* dgx, float* scratch,
                                       float* dbias_ih, float* dbias_hh, int32_t* bias_done, void* stream_) {
  PK2_REQUIRE(bias_done, "lstm_bwd_bias: null bias_done");
  return lstm_layer_bwd_impl(dy, whh, gates, cells, B, T, H, D, dgx, scratch, dbias_ih, dbias_hh, bias_done, stream_);
}

static int lstm_layer_bwd_impl(const float* dy, const float* whh, const float* gates, const float* cells, int32_t B, int32_t T,
                               int32_t H, int32_t D, float* dgx, float* scratch, float* dbias_ih, float* dbias_hh,
                               int32_t* bias_done, void* stream_) {
  PK2_REQUIRE(dy && whh && gates && cells && dgx && scratch && B > 0 && T > 0 && (D == 1 || D == 2),
              "lstm_bwd: bad args");
  PK2_REQUIRE(lstm_h_ok(H), "lstm_bwd: hidden size %d unsupported (64,128,256,512,1024)", H);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (bias_done) *bias_done = 0;
  if (lstm_seq_wanted(B, H, D) && !getenv("PK2_LSTM_PERSIST_FWD_ONLY")) {
    bool ran = false, bdone = false;
    int prc = lstm_bwd_seq_launch(dy, whh, gates, cells, B, T, H, D, dgx, stream, &ran, dbias_ih, dbias_hh, &bdone);
    if (prc) return prc;
    if (ran) { if (bias_done) *bias_done = bdone ? 1 : 0; return PK2_OK; }
  }
  if (lstm_persist_wanted(B, H, D) && !getenv("PK2_LSTM_PERSIST_FWD_ONLY")) {   // one launch for the whole sequence (lstm_persist.hip)
    bool ran = false;      // the mailboxes (1 MB) live where the step kernels keep W_hh^T
    int prc = lstm_bwd_persist_launch(dy, whh, gates, cells, B, T, H, D, dgx, scratch, stream, &ran);
    if (prc) return prc;
    if (ran) return PK2_OK;
  }
  if (lstm_big_wanted(B, H, D) && !getenv("PK2_LSTM_PERSIST_FWD_ONLY")) {   // large batches, one launch (lstm_persist_big.hip)
    bool ran = false;
    int prc = lstm_bwd_big_launch(dy, whh, gates, cells, B, T, H, D, dgx, stream, &ran);
    if (prc) return prc;
    if (ran) return PK2_OK;
  }
  float* whhT = scratch;
  float* dc = scratch + (size_t)D * H * 4 * H;
  hipLaunchKernelGGL(transpose_whh, dim3(H / 32, 4 * H / 32, D), dim3(256), 0, stream, whh, whhT, H);
  ParamSlot<LstmBwdParams>* slot;
  int rc = get_param_slot(g_bwd_slots, H * 4 + D, stream, &slot);
  if (rc) return rc;
  const bool big = B >= kBigBatch && H % 64 == 0;   // K-slice 4H / kBigSplitK = H: whole stages of 64
  float* dh_part = dc + (size_t)D * B * H + 64;
  LstmBwdParams p{dy, whhT, gates, cells, dgx, dc, B, T, H, D, whh, dh_part};
  hipLaunchKernelGGL(param_block_store<LstmBwdParams>, dim3(1), dim3(1), 0, stream, p, slot->params);
  if (big) {
    const LstmBwdParams* pbb = slot->params;
    const StepCounter* cb = slot->counter;
    dim3 gridg(H / 64, D * kBigSplitK, (B + 63) / 64);
    const int pw_blocks = (int)(((int64_t)D * B * H + 255) / 256);
    char keyb[64];
    snprintf(keyb, sizeof(keyb), "lstm_bwd_big_H%d_D%d_Z%d_%d_%p", H, D, (int)gridg.z, pw_blocks, (void*)stream);
    rc = g_graphs.run(keyb, T, slot->counter, stream, [&](hipStream_t s, int j) {
      hipLaunchKernelGGL(lstm_bwd_dh_big, gridg, dim3(256), 0, s, pbb, cb, j);
      hipLaunchKernelGGL(lstm_bwd_pointwise_big, dim3(pw_blocks), dim3(256), 0, s, pbb, cb, j);
    });
    if (rc) return rc;
    PK2_LAUNCH_CHECK();
    return PK2_OK;
  }
  // batches below 32 rows: the 4x4x1-MFMA kernel (PK2_LSTM_BWD_X4=0 keeps the 16x16x4 one)
  static const bool use_x4 = [] { const char* e = getenv("PK2_LSTM_BWD_X4"); return !(e && atoi(e) == 0); }();
  if (use_x4 && B <= 4 * kX4MaxGroups) {
    dim3 gridx(H / 4, D, 1);
    const LstmBwdParams* pbx = slot->params;
    const StepCounter* cx = slot->counter;
    char keyx[64];
    snprintf(keyx, sizeof(keyx), "lstm_bwd_x4_H%d_D%d_%p", H, D, (void*)stream);
    rc = g_graphs.run(keyx, T, slot->counter, stream, [&](hipStream_t s, int j) {
      switch (H) {
        case 64: launch_step(lstm_bwd_step_x4<64>, gridx, dim3(256), s, pbx, cx, j); break;
        case 128: launch_step(lstm_bwd_step_x4<128>, gridx, dim3(256), s, pbx, cx, j); break;
        case 256: launch_step(lstm_bwd_step_x4<256>, gridx, dim3(256), s, pbx, cx, j); break;
        case 512: launch_step(lstm_bwd_step_x4<512>, gridx, dim3(256), s, pbx, cx, j); break;
        default: launch_step(lstm_bwd_step_x4<1024>, gridx, dim3(256), s, pbx, cx, j); break;
      }
    });
    if (rc) return rc;
    PK2_LAUNCH_CHECK();
    return PK2_OK;
  }
  const int zb = ((B + 15) / 16 + kBwdTileGroup - 1) / kBwdTileGroup;
  const char* wenv = getenv("PK2_LSTM_BWD_WAVES");
  const int waves = (wenv && atoi(wenv) == 8) ? 8 : kBwdWavesDefault;
  dim3 grid(H / kBwdUnits, D, zb), block(waves * 64);
  const LstmBwdParams* pb = slot->params;
  const StepCounter* c = slot->counter;
  char key[64];
  snprintf(key, sizeof(key), "lstm_bwd_H%d_D%d_Z%d_W%d_%p", H, D, zb, waves, (void*)stream);
  rc = g_graphs.run(key, T, slot->counter, stream, [&](hipStream_t s, int j) {
#define PK2_BWD(HH)                                                                                  \
  case HH:                                                                                           \
    if (waves == 8) launch_step(lstm_bwd_step<HH / 8, 8>, grid, block, s, pb, c, j);                 \
    else launch_step(lstm_bwd_step<HH / 16, 16>, grid, block, s, pb, c, j);                          \
    break;
    switch (H) {
      PK2_BWD(64) PK2_BWD(128) PK2_BWD(256) PK2_BWD(512) PK2_BWD(1024)
    }
#undef PK2_BWD
  });
  if (rc) return rc;
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
