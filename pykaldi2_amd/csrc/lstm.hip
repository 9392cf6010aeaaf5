// Recurrent part of one (bi)directional LSTM layer on gfx950: forward and backward through time.
//
// Replaces the cuDNN RNN under nn.LSTM (reference models/lstm.py:49-58): gate order i,f,g,o,
// h0 = c0 = 0, every sequence runs over all T frames (the reference does not pack, so padding
// frames are processed too).  The input projections (x W_ih^T + b_ih + b_hh for all frames) and
// all weight gradients are large GEMMs done by pk2_gemm_f32; this file is the serial part.
//
// One launch per time step, both directions in the same launch (blockIdx.y).  The recurrent
// product h_{t-1} W_hh^T runs on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32):
//  * forward: a workgroup owns 4 hidden units = 16 gate rows (one 16-wide MFMA N tile); its 4
//    wavefronts split K = H, each holding its slice of those 16 W_hh rows in 32 VGPRs; batch rows
//    are the MFMA M dimension (tiles of 16); partial tiles meet in LDS, then the gate
//    activations, c_t and h_t are computed in the same kernel (fused pointwise).
//  * backward: a workgroup owns 16 hidden units (N tile) and splits K = 4H over 16 wavefronts;
//    d h_{t-1} = dgates_t W_hh is fused with the gate derivative of step t-1.
// W_hh slices are re-read from L2 every step (4 MB per direction stays L2 resident; each
// workgroup always reads the same slice, and consecutive launches place block b on XCD b%8).
// All activations are time-major ([T][B][...]) so "previous step" is a constant row offset.
#include <algorithm>

#include "common.h"

namespace pk2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFwdThreads = 256;    // 4 waves, K = H split 4 ways
constexpr int kFwdUnits = 4;        // hidden units per workgroup (x4 gates = 16 MFMA columns)
constexpr int kBwdThreads = 1024;   // 16 waves, K = 4H split 16 ways
constexpr int kBwdUnits = 16;       // hidden units per workgroup (16 MFMA columns)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct LstmFwdParams {
  const float* gx;    // [T][B][D*4H]
  const float* whh;   // [D][4H][H]
  const float* bhh;   // [D][4H] recurrent bias (may be null)
  float* y;           // [T][B][D*H]
  float* gates;       // [D][T][B][4H]
  float* cells;       // [D][T][B][H]
  int B, T, H, D;
};

// KS = number of 4-wide MFMA k-steps per wave (H / 4 waves / 4).
template <int KS>
__global__ void __launch_bounds__(kFwdThreads) lstm_fwd_step(LstmFwdParams p, int step) {
  __shared__ float part[4][16][17];   // per-wave partial 16x16 tiles (padded)
  const int d = blockIdx.y;
  const int u0 = blockIdx.x * kFwdUnits;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int H = p.H, B = p.B, T = p.T, D = p.D;
  const int t = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? t - 1 : t + 1;      // previous step in processing order
  const bool first = step == 0;
  const int li = lane & 15, kq = lane >> 4;

  // B operand: column j = li -> gate g = j/4, unit u0 + j%4 -> row g*H + u0 + j%4 of W_hh[d]
  f32x4 wf[KS / 4 > 0 ? KS / 4 : 1];
  const int kbase = w * (KS * 4) + kq * KS;   // this lane's contiguous k-run of length KS
  if (!first) {
    const float* wrow = p.whh + ((size_t)d * 4 * H + (size_t)(li >> 2) * H + u0 + (li & 3)) * H + kbase;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + q * 4);
  }
  const size_t yrow = (size_t)D * H;
  for (int mt = 0; mt < (B + 15) / 16; ++mt) {
    if (!first) {
      const int b = mt * 16 + li;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      f32x4 af[KS / 4 > 0 ? KS / 4 : 1];
      if (b < B) {
        const float* hrow = p.y + ((size_t)tp * B + b) * yrow + (size_t)d * H + kbase;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(hrow + q * 4);
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
      // C layout: row (batch) = (lane>>4)*4 + r, col = lane&15
#pragma unroll
      for (int r = 0; r < 4; ++r) part[w][kq * 4 + r][li] = acc0[r] + acc1[r];
    }
    __syncthreads();
    // fused gates: thread (i = batch row in tile, u = unit) for tid < 64
    if (tid < 64) {
      const int i = tid >> 2, u = tid & 3;
      const int b = mt * 16 + i;
      if (b < B) {
        float pre[4];
        const float* gxr = p.gx + ((size_t)t * B + b) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + u0 + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float s = gxr[(size_t)g * H];
          if (p.bhh) s += p.bhh[(size_t)d * 4 * H + (size_t)g * H + u0 + u];
          if (!first) s += (part[0][i][g * 4 + u] + part[1][i][g * 4 + u]) + (part[2][i][g * 4 + u] + part[3][i][g * 4 + u]);
          pre[g] = s;
        }
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
        const size_t cidx = (((size_t)d * T + t) * B + b) * H + u0 + u;
        const float cprev = first ? 0.f : p.cells[(((size_t)d * T + tp) * B + b) * H + u0 + u];
        const float c = fg * cprev + ig * gg;
        const float h = og * tanhf(c);
        p.cells[cidx] = c;
        p.y[((size_t)t * B + b) * yrow + (size_t)d * H + u0 + u] = h;
        float* gr = p.gates + (((size_t)d * T + t) * B + b) * 4 * H + u0 + u;
        gr[0] = ig; gr[(size_t)H] = fg; gr[(size_t)2 * H] = gg; gr[(size_t)3 * H] = og;
      }
    }
    __syncthreads();
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
};

// KS = 4-wide MFMA k-steps per wave (4H / 16 waves / 4).
template <int KS>
__global__ void __launch_bounds__(kBwdThreads) lstm_bwd_step(LstmBwdParams p, int step) {
  __shared__ float part[16][16][17];
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

  f32x4 wf[KS / 4 > 0 ? KS / 4 : 1];
  const int rbase = w * (KS * 4) + kq * KS;        // run of gate rows r handled by this lane
  if (!last_fwd) {
    const float* wrow = p.whhT + ((size_t)d * H + k0 + li) * G4 + rbase;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + q * 4);
  }
  for (int mt = 0; mt < (B + 15) / 16; ++mt) {
    if (!last_fwd) {
      const int b = mt * 16 + li;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      f32x4 af[KS / 4 > 0 ? KS / 4 : 1];
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
      for (int r = 0; r < 4; ++r) part[w][kq * 4 + r][li] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (tid < 256) {
      const int i = tid >> 4, j = tid & 15;
      const int b = mt * 16 + i, k = k0 + j;
      if (b < B) {
        float dh = p.dy[((size_t)t * B + b) * ((size_t)D * H) + (size_t)d * H + k];
        if (!last_fwd) {
          float s = 0.f;
#pragma unroll
          for (int ww = 0; ww < 16; ++ww) s += part[ww][i][j];
          dh += s;
        }
        const float* gr = p.gates + (((size_t)d * T + t) * B + b) * G4 + k;
        const float ig = gr[0], fg = gr[(size_t)H], gg = gr[(size_t)2 * H], og = gr[(size_t)3 * H];
        const float c = p.cells[(((size_t)d * T + t) * B + b) * H + k];
        const float cprev = first_fwd ? 0.f : p.cells[(((size_t)d * T + tp) * B + b) * H + k];
        const float tc = tanhf(c);
        float* dcp = p.dc + ((size_t)d * B + b) * H + k;
        const float dcv = (last_fwd ? 0.f : *dcp) + dh * og * (1.f - tc * tc);
        *dcp = dcv * fg;
        float* o = p.dgx + ((size_t)t * B + b) * ((size_t)D * G4) + (size_t)d * G4 + k;
        o[0] = dcv * gg * ig * (1.f - ig);
        o[(size_t)H] = dcv * cprev * fg * (1.f - fg);
        o[(size_t)2 * H] = dcv * ig * (1.f - gg * gg);
        o[(size_t)3 * H] = dh * tc * og * (1.f - og);
      }
    }
    __syncthreads();
  }
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

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_lstm_layer_fwd(const float* gx, const float* whh, const float* bhh, int32_t B, int32_t T,
                                  int32_t H, int32_t D, float* y, float* gates, float* cells, void* stream_) {
  PK2_REQUIRE(gx && whh && y && gates && cells && B > 0 && T > 0 && (D == 1 || D == 2), "lstm_fwd: bad args");
  PK2_REQUIRE(H % 64 == 0 && (H == 512 || H == 256 || H == 128 || H == 64 || H == 1024),
              "lstm_fwd: hidden size %d unsupported (64,128,256,512,1024)", H);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  LstmFwdParams p{gx, whh, bhh, y, gates, cells, B, T, H, D};
  dim3 grid(H / kFwdUnits, D), block(kFwdThreads);
  for (int s = 0; s < T; ++s) {
    switch (H) {
      case 64: hipLaunchKernelGGL(lstm_fwd_step<4>, grid, block, 0, stream, p, s); break;
      case 128: hipLaunchKernelGGL(lstm_fwd_step<8>, grid, block, 0, stream, p, s); break;
      case 256: hipLaunchKernelGGL(lstm_fwd_step<16>, grid, block, 0, stream, p, s); break;
      case 512: hipLaunchKernelGGL(lstm_fwd_step<32>, grid, block, 0, stream, p, s); break;
      default: hipLaunchKernelGGL(lstm_fwd_step<64>, grid, block, 0, stream, p, s); break;
    }
  }
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" size_t pk2_lstm_bwd_scratch_floats(int32_t B, int32_t H, int32_t D) {
  return (size_t)D * H * 4 * H + (size_t)D * B * H + 64;
}

extern "C" int pk2_lstm_layer_bwd(const float* dy, const float* whh, const float* gates, const float* cells,
                                  int32_t B, int32_t T, int32_t H, int32_t D, float* dgx, float* scratch,
                                  void* stream_) {
  PK2_REQUIRE(dy && whh && gates && cells && dgx && scratch && B > 0 && T > 0 && (D == 1 || D == 2),
              "lstm_bwd: bad args");
  PK2_REQUIRE(H % 64 == 0 && (H == 512 || H == 256 || H == 128 || H == 64 || H == 1024),
              "lstm_bwd: hidden size %d unsupported (64,128,256,512,1024)", H);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* whhT = scratch;
  float* dc = scratch + (size_t)D * H * 4 * H;
  hipLaunchKernelGGL(transpose_whh, dim3(H / 32, 4 * H / 32, D), dim3(256), 0, stream, whh, whhT, H);
  LstmBwdParams p{dy, whhT, gates, cells, dgx, dc, B, T, H, D};
  dim3 grid(H / kBwdUnits, D), block(kBwdThreads);
  for (int s = 0; s < T; ++s) {
    switch (H) {
      case 64: hipLaunchKernelGGL(lstm_bwd_step<4>, grid, block, 0, stream, p, s); break;
      case 128: hipLaunchKernelGGL(lstm_bwd_step<8>, grid, block, 0, stream, p, s); break;
      case 256: hipLaunchKernelGGL(lstm_bwd_step<16>, grid, block, 0, stream, p, s); break;
      case 512: hipLaunchKernelGGL(lstm_bwd_step<32>, grid, block, 0, stream, p, s); break;
      default: hipLaunchKernelGGL(lstm_bwd_step<64>, grid, block, 0, stream, p, s); break;
    }
  }
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
