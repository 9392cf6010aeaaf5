// Fused multi-head self-attention for gfx950 (f32 MFMA, head size 64): softmax(Q K^T / sqrt(d) + masks) V in one kernel,
// the scores never leave the CU; backward by recomputation from the saved log-sum-exp.
//
// Replaces nn.MultiheadAttention's scaled-dot-product core under nn.TransformerEncoderLayer (reference
// models/transformer.py:52-67): the unfused form (pykaldi2_amd/transformer.py, PK2_ATTN_FUSED=0) keeps B*H*T^2 scores
// per layer in HBM for the backward pass and launches 2 + 4 batched GEMMs and 2 row kernels per layer.
//
// Layout: the packed projection qkv[T*B][3C] (row = t*B + b; Q | K | V, head h at columns h*64) is read in place, the
// context ctx[T*B][C] is written head-interleaved -- no transposes.  A wave owns a 32 x 32 (query x key) tile at a time on
// v_mfma_f32_32x32x2_f32, everything TRANSPOSED so that a lane holds 16 keys of ONE query:
//     S^T[key][query] = K_tile Q^T          A = K rows (from LDS), B = Q^T (registers, pre-scaled)
//     O^T[dim][query] += V_tile^T P^T       B = the S^T accumulator registers AS THEY ARE: register j of the lanes
//                                           hi = 0 / 1 holds keys (j&3)+8(j>>2) and that + 4 -- exactly the two k's one
//                                           MFMA consumes, so P never moves between lanes or through LDS
// and the row statistics of the online softmax are 16 in-lane values + one exchange with lane ^ 32.
// A workgroup = 4 waves working on one (utterance, head, 32-query tile), each on every 4th key tile; their partial
// (max, sum, O) are merged through LDS.  The backward pass is two kernels of the same shape: dQ (owner: query tile, keys
// split over the waves) which also leaves D = rowsum(dO * O), then dK / dV (owner: key tile, queries split over the waves).
// Dropout on the attention probabilities uses the counter-based mask of dropout.hip indexed like the unfused [B*H][T][T]
// matrix, so both forms draw the same mask.
#include <algorithm>
#include <cmath>

#include <cstdlib>

#include "common.h"

namespace pk2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kAD = 64;            // head size served
constexpr int kAT = 32;            // tile edge (queries, keys)
constexpr int kALd = 65;           // LDS row pitch of a staged 32 x 64 tile: rows on different banks
constexpr int kAWaves = 4;
constexpr int kATileFloats = kAT * kALd;

struct AttnParams {
  const float* qkv; const float* ctx; const float* dctx; const float* lse_in;
  float* ctx_out; float* lse_out; float* dqkv; float* dsum;
  const float* src_mask; const uint8_t* key_pad; int skip_pad;
  int T, B, H;
  float scale;
  uint32_t keep_threshold; float keep_scale; uint64_t seed; int dropout;
};

__device__ __forceinline__ uint32_t attn_mix32(uint64_t z) {     // dropout.hip: splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float attn_keep(const AttnParams& p, int64_t idx) {
  const uint32_t r = attn_mix32(p.seed * 0xD1342543DE82EF95ull + (uint64_t)idx);
  return r < p.keep_threshold ? p.keep_scale : 0.f;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rows [r0, r0 + 32) of a matrix with row stride `rs` floats, 64 columns from `base`, into a wave-private LDS tile
// [32][kALd]; rows >= T read as zero.  Coalesced: an instruction covers 4 whole rows.
__device__ __forceinline__ void stage_tile(const float* __restrict__ base, int64_t rs, int r0, int T, float* tile) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), col = (lane & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < T) v = *reinterpret_cast<const float4*>(base + (int64_t)(r0 + row) * rs + col);
    float* o = tile + row * kALd + col;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
}
// The same in two halves, so that the global loads of the NEXT tile are in flight while a tile is multiplied.
struct TileRegs { float4 v[8]; };
__device__ __forceinline__ void fetch_tile(const float* __restrict__ base, int64_t rs, int r0, int T, TileRegs& t) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), col = (lane & 15) * 4;
    t.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < T) t.v[it] = *reinterpret_cast<const float4*>(base + (int64_t)(r0 + row) * rs + col);
  }
}
__device__ __forceinline__ void put_tile(const TileRegs& t, float* tile) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    float* o = tile + (it * 4 + (lane >> 4)) * kALd + (lane & 15) * 4;
    o[0] = t.v[it].x; o[1] = t.v[it].y; o[2] = t.v[it].z; o[3] = t.v[it].w;
  }
}
// Every element of the tile is (+-) zero: a tile of output gradients of frames no loss term reaches (padded frames more than a
// few frames behind an utterance's end -- the convolutions carry gradient one frame further per layer) multiplies into exact
// zeros in the backward kernels, so they leave it out (PK2_ATTN_SKIP_PAD=0 keeps it in).
__device__ __forceinline__ bool tile_is_zero(const TileRegs& t) {
  unsigned any = 0u;
#pragma unroll
  for (int it = 0; it < 8; ++it)
    any |= (__float_as_uint(t.v[it].x) | __float_as_uint(t.v[it].y) | __float_as_uint(t.v[it].z) | __float_as_uint(t.v[it].w)) << 1;
  return __ballot(any != 0u) == 0ull;
}
// "row operand" of a staged tile: lane (row = lane % 32, hi = lane / 32) -> tile[row][32 hi + i], i < 32
__device__ __forceinline__ void load_rows(const float* tile, float (&r)[32]) {
  const int lane = threadIdx.x & 63;
  const float* src = tile + (lane & 31) * kALd + 32 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = src[i];
}
// tile row the MFMA instruction j reads in the lanes' half hi (see the header)
__device__ __forceinline__ int row_of(int j, int hi) { return (j & 3) + 8 * (j >> 2) + 4 * hi; }

// acc[dim][n] += sum_rows tile[row][dim] * w[row][n] for the 64 dims (two 32-row blocks of the output), the weights w being
// an accumulator-layout register set (register j <-> tile rows row_of(j, hi)).
__device__ __forceinline__ void mfma_tile_t(const float* tile, const f32x16& w, f32x16& acc0, f32x16& acc1) {
  const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float* src = tile + row_of(j, hi) * kALd + m;
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(src[0], w[j], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(src[32], w[j], acc1, 0, 0, 0);
  }
}
// acc[row][n] = sum_dim tile[row][dim] * b[dim][n], b = a "row operand" register set of the n side
__device__ __forceinline__ f32x16 mfma_rows(const float* tile, const float (&b)[32]) {
  const int lane = threadIdx.x & 63;
  const float* src = tile + (lane & 31) * kALd + 32 * (lane >> 5);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(src[i], b[i], acc, 0, 0, 0);
  return acc;
}

// Additive mask of (query, key): -inf outside the sequence / on padded keys.
__device__ __forceinline__ float mask_of(const AttnParams& p, int b, int q, int k) {
  if (k >= p.T || (p.key_pad && p.key_pad[(int64_t)b * p.T + k])) return -INFINITY;
  return (p.src_mask && q < p.T) ? p.src_mask[(int64_t)q * p.T + k] : 0.f;
}

// Key tiles of utterance b that hold at least one key which is not padding (every wave computes it for itself: T bytes).  The
// tiles behind the last valid key contribute exactly nothing -- every probability is 0 -- and a minibatch of utterances of
// different lengths is mostly such tiles for its short ones (the bench minibatch, 146 / 539 / 569 / 159 frames: 45 of 72
// (utterance, key tile) pairs are valid): the forward and dQ loops end there, a dK / dV workgroup of such a tile stores zeros.
// PK2_ATTN_SKIP_PAD=0 (AttnParams::skip_pad) walks every tile as rounds 2-6a did.
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int valid_key_tiles(const AttnParams& p, int b) {
  const int all = (p.T + kAT - 1) / kAT;
  if (!p.key_pad || !(p.skip_pad & 1)) return all;
  const uint8_t* kp = p.key_pad + (int64_t)b * p.T;
  int last = -1;
  for (int k = threadIdx.x & 63; k < p.T; k += 64)
    if (!kp[k]) last = k;
  last = __builtin_amdgcn_readfirstlane(wave_max_i(last));
  return min(all, (last + kAT) / kAT);
}

// The 32 floats of dims {4 hi + (r&3) + 8 (r>>2)} (+32) of query/key row `row` held by a lane in the transposed
// accumulators acc0 / acc1 -> out[row][...] (8 float4 stores), scaled.
__device__ __forceinline__ void store_t(float* out_row, int hi, const f32x16& acc0, const f32x16& acc1, float s) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<float4*>(out_row + 8 * g + 4 * hi) =
        make_float4(acc0[4 * g] * s, acc0[4 * g + 1] * s, acc0[4 * g + 2] * s, acc0[4 * g + 3] * s);
    *reinterpret_cast<float4*>(out_row + 32 + 8 * g + 4 * hi) =
        make_float4(acc1[4 * g] * s, acc1[4 * g + 1] * s, acc1[4 * g + 2] * s, acc1[4 * g + 3] * s);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * kAWaves, 2) attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kAWaves][2 * kATileFloats];
  __shared__ float stat[kAWaves][2][kAT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, qi = lane & 31;
  const int z = blockIdx.y, b = z / p.H, h = z % p.H, q0 = blockIdx.x * kAT, T = p.T;
  const int C = p.H * kAD;
  const int64_t rs = (int64_t)p.B * 3 * C;                      // floats between consecutive frames of one utterance
  const float* Q = p.qkv + (int64_t)b * 3 * C + h * kAD;
  const float* K = Q + C;
  const float* V = Q + 2 * C;
  float* Ks = lds[w]; float* Vs = lds[w] + kATileFloats;
  float qreg[32];
  stage_tile(Q, rs, q0, T, Ks);
  wave_lds_sync();
  load_rows(Ks, qreg);
#pragma unroll
  for (int i = 0; i < 32; ++i) qreg[i] *= p.scale;
  wave_lds_sync();
  float m = -INFINITY, l = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  const int q = q0 + qi;
  const int nkt = valid_key_tiles(p, b);
  TileRegs kr, vr;
  if (w < nkt) { fetch_tile(K, rs, w * kAT, T, kr); fetch_tile(V, rs, w * kAT, T, vr); }
  for (int kt = w; kt < nkt; kt += kAWaves) {
    const int k0 = kt * kAT;
    put_tile(kr, Ks);
    put_tile(vr, Vs);
    wave_lds_sync();
    if (kt + kAWaves < nkt) { fetch_tile(K, rs, k0 + kAWaves * kAT, T, kr); fetch_tile(V, rs, k0 + kAWaves * kAT, T, vr); }
    f32x16 s = mfma_rows(Ks, qreg);                              // S^T: register r <-> key k0 + row_of(r, hi), this lane's query
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] += mask_of(p, b, q, k0 + row_of(r, hi));
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    if (__ballot(mn > -INFINITY) != 0ull) {                       // (a tile masked out for every query adds nothing)
      const float corr = (m == -INFINITY) ? 0.f : __expf(m - mn);
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = (mn == -INFINITY) ? 0.f : __expf(s[r] - mn);
        ls += s[r];
      }
      ls += __shfl_xor(ls, 32, 64);
      l = l * corr + ls;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
      if (p.dropout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= attn_keep(p, ((int64_t)z * T + q) * T + k0 + row_of(r, hi));
      }
      mfma_tile_t(Vs, s, o0, o1);
    }
    wave_lds_sync();
  }
  // merge the waves' partial results: the query of a lane is the same in every wave
  if (hi == 0) { stat[w][0][qi] = m; stat[w][1][qi] = l; }
  __syncthreads();
  float mstar = -INFINITY;
#pragma unroll
  for (int k = 0; k < kAWaves; ++k) mstar = fmaxf(mstar, stat[k][0][qi]);
  float lstar = 0.f;
#pragma unroll
  for (int k = 0; k < kAWaves; ++k) lstar += stat[k][0][qi] == -INFINITY ? 0.f : stat[k][1][qi] * __expf(stat[k][0][qi] - mstar);
  const float mine = (m == -INFINITY) ? 0.f : __expf(m - mstar);
  float* red = lds[w];                                           // [32 regs][64 lanes]
#pragma unroll
  for (int r = 0; r < 16; ++r) { red[r * 64 + lane] = o0[r] * mine; red[(16 + r) * 64 + lane] = o1[r] * mine; }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int k = 0; k < kAWaves; ++k) { a += lds[k][r * 64 + lane]; c += lds[k][(16 + r) * 64 + lane]; }
      o0[r] = a; o1[r] = c;
    }
    if (q < T) {
      const float inv = lstar > 0.f ? 1.0f / lstar : 0.f;
      store_t(p.ctx_out + ((int64_t)q * p.B + b) * C + h * kAD, hi, o0, o1, inv);
      if (hi == 0) p.lse_out[(int64_t)z * T + q] = lstar > 0.f ? mstar + __logf(lstar) : -INFINITY;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward 1: dQ (and D = rowsum(dO * O)), owner = query tile, the key tiles dealt to the waves
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * kAWaves, 2) attn_bwd_dq_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kAWaves][2 * kATileFloats];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, qi = lane & 31;
  const int z = blockIdx.y, b = z / p.H, h = z % p.H, q0 = blockIdx.x * kAT, T = p.T;
  const int C = p.H * kAD;
  const int64_t rs = (int64_t)p.B * 3 * C, rc = (int64_t)p.B * C;
  const float* Q = p.qkv + (int64_t)b * 3 * C + h * kAD;
  const float* K = Q + C;
  const float* V = Q + 2 * C;
  const float* O = p.ctx + (int64_t)b * C + h * kAD;
  const float* dO = p.dctx + (int64_t)b * C + h * kAD;
  float* Ks = lds[w]; float* Vs = lds[w] + kATileFloats;
  float qreg[32], doreg[32];
  stage_tile(Q, rs, q0, T, Ks);
  stage_tile(dO, rc, q0, T, Vs);
  wave_lds_sync();
  load_rows(Ks, qreg);
  load_rows(Vs, doreg);
  wave_lds_sync();
  if (p.skip_pad & 2) {      // dO of this query tile is all zero (every wave holds the same tile): D = 0, dQ = 0
    unsigned any = 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) any |= __float_as_uint(doreg[i]) << 1;
    if (__ballot(any != 0u) == 0ull) {
      const int q = q0 + qi;
      if (w == 0 && q < T) {
        if (hi == 0) p.dsum[(int64_t)z * T + q] = 0.f;
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        store_t(p.dqkv + ((int64_t)q * p.B + b) * 3 * C + h * kAD, hi, zero, zero, 1.0f);
      }
      return;
    }
  }
  stage_tile(O, rc, q0, T, Ks);
  wave_lds_sync();
  float dsum = 0.f;
  {
    float oreg[32];
    load_rows(Ks, oreg);
#pragma unroll
    for (int i = 0; i < 32; ++i) dsum += oreg[i] * doreg[i];
    dsum += __shfl_xor(dsum, 32, 64);
  }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < 32; ++i) qreg[i] *= p.scale;
  const int q = q0 + qi;
  const float lse = q < T ? p.lse_in[(int64_t)z * T + q] : -INFINITY;
  if (w == 0 && hi == 0 && q < T) p.dsum[(int64_t)z * T + q] = dsum;
  f32x16 g0, g1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  const int nkt = valid_key_tiles(p, b);
  TileRegs kr, vr;
  if (w < nkt) { fetch_tile(K, rs, w * kAT, T, kr); fetch_tile(V, rs, w * kAT, T, vr); }
  for (int kt = w; kt < nkt; kt += kAWaves) {
    const int k0 = kt * kAT;
    put_tile(kr, Ks);
    put_tile(vr, Vs);
    wave_lds_sync();
    if (kt + kAWaves < nkt) { fetch_tile(K, rs, k0 + kAWaves * kAT, T, kr); fetch_tile(V, rs, k0 + kAWaves * kAT, T, vr); }
    f32x16 s = mfma_rows(Ks, qreg);
    f32x16 dp = mfma_rows(Vs, doreg);                            // dP^T[key][query] = V dO^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + row_of(r, hi);
      const float sv = s[r] + mask_of(p, b, q, k);
      const float pr = (lse == -INFINITY || sv == -INFINITY) ? 0.f : __expf(sv - lse);
      float d = dp[r];
      if (p.dropout) d *= attn_keep(p, ((int64_t)z * T + q) * T + k);
      s[r] = pr * (d - dsum) * p.scale;                          // dS^T
    }
    mfma_tile_t(Ks, s, g0, g1);                                  // dQ^T[dim][query] += K^T dS^T
    wave_lds_sync();
  }
  float* red = lds[w];
#pragma unroll
  for (int r = 0; r < 16; ++r) { red[r * 64 + lane] = g0[r]; red[(16 + r) * 64 + lane] = g1[r]; }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int k = 0; k < kAWaves; ++k) { a += lds[k][r * 64 + lane]; c += lds[k][(16 + r) * 64 + lane]; }
      g0[r] = a; g1[r] = c;
    }
    if (q < T) store_t(p.dqkv + ((int64_t)q * p.B + b) * 3 * C + h * kAD, hi, g0, g1, 1.0f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward 2: dK, dV, owner = key tile, the query tiles dealt to the waves
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * kAWaves) attn_bwd_dkv_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kAWaves][2 * kATileFloats];
  __shared__ float qstat[kAWaves][2][kAT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, ki = lane & 31;
  const int z = blockIdx.y, b = z / p.H, h = z % p.H, k0 = blockIdx.x * kAT, T = p.T;
  const int C = p.H * kAD;
  const int64_t rs = (int64_t)p.B * 3 * C, rc = (int64_t)p.B * C;
  const float* Q = p.qkv + (int64_t)b * 3 * C + h * kAD;
  const float* K = Q + C;
  const float* V = Q + 2 * C;
  const float* dO = p.dctx + (int64_t)b * C + h * kAD;
  float* Qs = lds[w]; float* Ds = lds[w] + kATileFloats;
  float kreg[32], vreg[32];
  stage_tile(K, rs, k0, T, Qs);
  stage_tile(V, rs, k0, T, Ds);
  wave_lds_sync();
  load_rows(Qs, kreg);
  load_rows(Ds, vreg);
  wave_lds_sync();
  const int k = k0 + ki;
  f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { gk0[r] = 0.f; gk1[r] = 0.f; gv0[r] = 0.f; gv1[r] = 0.f; }
  const int nqt = blockIdx.x < valid_key_tiles(p, b) ? (T + kAT - 1) / kAT : 0;      // (a tile of padded keys: dK = dV = 0, stored below)
  // (the key side holds 64 registers here, the prefetched tiles 64 more: one workgroup per CU.  Measured: 135 us per layer
  // against 195 us without the prefetch at two workgroups per CU)
  TileRegs qr, dr;
  if (w < nqt) { fetch_tile(Q, rs, w * kAT, T, qr); fetch_tile(dO, rc, w * kAT, T, dr); }
  for (int qt = w; qt < nqt; qt += kAWaves) {
    const int q0 = qt * kAT;
    const bool dz = (p.skip_pad & 2) && tile_is_zero(dr);              // dO = 0 (and with it D = 0): dV += 0, dS = 0
    if (!dz) { put_tile(qr, Qs); put_tile(dr, Ds); }
    if (qt + kAWaves < nqt) { fetch_tile(Q, rs, q0 + kAWaves * kAT, T, qr); fetch_tile(dO, rc, q0 + kAWaves * kAT, T, dr); }
    if (dz) continue;
    if (lane < kAT) {
      const int q = q0 + lane;
      qstat[w][0][lane] = q < T ? p.lse_in[(int64_t)z * T + q] : -INFINITY;
      qstat[w][1][lane] = q < T ? p.dsum[(int64_t)z * T + q] : 0.f;
    }
    wave_lds_sync();
    f32x16 s = mfma_rows(Qs, kreg);                              // S[query][key]: register r <-> query q0 + row_of(r, hi), this lane's key
    f32x16 dp = mfma_rows(Ds, vreg);                             // dP[query][key] = dO V^T
    f32x16 pd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + row_of(r, hi);
      const float lse = qstat[w][0][row_of(r, hi)], dsum = qstat[w][1][row_of(r, hi)];
      const float sv = s[r] * p.scale + mask_of(p, b, q, k);
      const float pr = (q >= T || lse == -INFINITY || sv == -INFINITY) ? 0.f : __expf(sv - lse);
      const float keep = p.dropout ? attn_keep(p, ((int64_t)z * T + q) * T + k) : 1.0f;
      pd[r] = pr * keep;                                         // dropped-out probabilities: dV = P_d^T dO
      s[r] = pr * (dp[r] * keep - dsum) * p.scale;               // dS
    }
    mfma_tile_t(Ds, pd, gv0, gv1);                               // dV^T[dim][key] += dO^T P_d
    mfma_tile_t(Qs, s, gk0, gk1);                                // dK^T[dim][key] += Q^T dS
    wave_lds_sync();
  }
  float* red = lds[w];
  auto reduce_store = [&](const f32x16& a0, const f32x16& a1, int part) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[r * 64 + lane] = a0[r]; red[(16 + r) * 64 + lane] = a1[r]; }
    __syncthreads();
    if (w == 0) {
      f32x16 t0, t1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int kk = 0; kk < kAWaves; ++kk) { a += lds[kk][r * 64 + lane]; c += lds[kk][(16 + r) * 64 + lane]; }
        t0[r] = a; t1[r] = c;
      }
      if (k < T) store_t(p.dqkv + ((int64_t)k * p.B + b) * 3 * C + part * C + h * kAD, hi, t0, t1, 1.0f);
    }
    __syncthreads();
  };
  reduce_store(gk0, gk1, 1);
  reduce_store(gv0, gv1, 2);
}

}  // namespace pk2

using namespace pk2;

static int attn_fill(AttnParams* p, int32_t T, int32_t B, int32_t H, int32_t head_dim, float scale, const float* src_mask,
                     const uint8_t* key_pad, float dropout_p, uint64_t seed) {
  PK2_REQUIRE(T > 0 && B > 0 && H > 0 && (int64_t)B * H <= 65535, "attention: bad sizes");
  PK2_REQUIRE(head_dim == kAD, "attention: the fused kernel serves head size %d (got %d)", kAD, head_dim);
  PK2_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attention: bad dropout");
  p->T = T; p->B = B; p->H = H; p->scale = scale; p->src_mask = src_mask; p->key_pad = key_pad;
  // bit 0: loops end at the last valid key tile; bit 1: the backward kernels leave all-zero dO tiles out (default: both)
  static const int skip_pad = [] { const char* e = getenv("PK2_ATTN_SKIP_PAD"); return e ? (atoi(e) & 3) : 3; }();
  p->skip_pad = skip_pad;
  const double keep = 1.0 - (double)dropout_p;
  p->dropout = dropout_p > 0.f;
  p->keep_threshold = (uint32_t)std::min<double>(4294967295.0, keep * 4294967296.0);
  p->keep_scale = (float)(1.0 / keep);
  p->seed = seed;
  return PK2_OK;
}

extern "C" int pk2_attention_fwd(const float* qkv, int32_t T, int32_t B, int32_t H, int32_t head_dim, float scale,
                                 const float* src_mask, const uint8_t* key_padding, float dropout_p, uint64_t seed,
                                 float* ctx, float* lse, void* stream_) {
  PK2_REQUIRE(qkv && ctx && lse, "attention_fwd: null pointer");
  AttnParams p{};
  int rc = attn_fill(&p, T, B, H, head_dim, scale, src_mask, key_padding, dropout_p, seed);
  if (rc) return rc;
  p.qkv = qkv; p.ctx_out = ctx; p.lse_out = lse;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((T + kAT - 1) / kAT, B * H), dim3(64 * kAWaves), 0, static_cast<hipStream_t>(stream_), p);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_attention_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, int32_t T,
                                 int32_t B, int32_t H, int32_t head_dim, float scale, const float* src_mask,
                                 const uint8_t* key_padding, float dropout_p, uint64_t seed, float* dqkv, float* dsum,
                                 void* stream_) {
  PK2_REQUIRE(qkv && ctx && dctx && lse && dqkv && dsum, "attention_bwd: null pointer");
  AttnParams p{};
  int rc = attn_fill(&p, T, B, H, head_dim, scale, src_mask, key_padding, dropout_p, seed);
  if (rc) return rc;
  p.qkv = qkv; p.ctx = ctx; p.dctx = dctx; p.lse_in = lse; p.dqkv = dqkv; p.dsum = dsum;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid((T + kAT - 1) / kAT, B * H), block(64 * kAWaves);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, block, 0, stream, p);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, block, 0, stream, p);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
