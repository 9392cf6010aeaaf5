// Denominator forward-backward of the LF-MMI objective on gfx950.
//
// Replaces the DenominatorComputation half of kaldi.chain.compute_chain_objf_and_deriv
// (reference ops/ops.py:265); arithmetic per SURVEY.md Appendix A.2 (probability space, per-frame
// 1/sum(alpha) rescaling, leaky-HMM).  The graph layouts come from chain_graph.hip.
//
// MI355X design (not Kaldi's one-thread-per-state CUDA kernels, one launch per frame per op):
//  * up to 4 sequences are interleaved in every per-state / per-pdf vector, so one 16-byte
//    gather serves 4 sequences and the arc lists are streamed once per frame for all of them;
//  * arcs are pre-sorted three ways on the host (by destination for alpha, by source for beta,
//    by pdf for the occupancies), so every reduction is a segmented gather-reduce into an
//    LDS-private accumulator -- no global atomics on the common path;
//  * exp(logits) of the frame (P x 4 floats, 97 KB) is staged in LDS once per workgroup;
//  * arcs are stored lane-interleaved: each 64-lane wavefront loads 1 KiB per instruction;
//  * ONE launch per frame and direction: the global sums of a frame (sum alpha, sum pi*beta) are
//    never reduced in a separate launch -- each workgroup writes a partial and the NEXT frame's
//    workgroups all re-reduce the partials in a fixed order (bitwise identical everywhere); the
//    leaky-HMM term is applied on the fly through a precomputed pi[src]*prob per arc;
//  * the backward launch pairs a beta chunk (8 waves) and an occupancy chunk (8 waves) in one
//    1024-thread workgroup sharing the staged exp(logits);
//  * every global load a workgroup needs that does not depend on LDS (partials, exp(logits),
//    arc records, the alpha/beta gathers) is issued before the first barrier, so a frame costs
//    about two dependent memory round trips; the T launches are replayed from cached hipGraphs.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "chain_internal.h"
#include "chain_num.h"
#include "den_kernels.h"
#include "den_persist.h"
#include "persist_guard.h"
#include "step_graph.h"

namespace pk2 {

// ----------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------
template <int NG> struct Pack;
template <> struct Pack<4> { using type = float4; };
template <> struct Pack<2> { using type = float2; };
template <> struct Pack<1> { using type = float; };

template <int NG>
__device__ __forceinline__ void ldv(const float* p, float (&v)[NG]) {
  using T = typename Pack<NG>::type;
  T t = *reinterpret_cast<const T*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int n = 0; n < NG; ++n) v[n] = f[n];
}
template <int NG>
__device__ __forceinline__ void stv(float* p, const float (&v)[NG]) {
  using T = typename Pack<NG>::type;
  T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int n = 0; n < NG; ++n) f[n] = v[n];
  *reinterpret_cast<T*>(p) = t;
}

// Sum over the NW wavefronts of the workgroup; every thread receives the same bits.
template <int NG, int NW>
__device__ __forceinline__ void block_sum(float (&v)[NG], float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NG; ++n) v[n] = wave_sum(v[n]);
  __syncthreads();  // protects `red` against a previous use; publishes earlier LDS writes
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NG; ++n) red[w * NG + n] = v[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    float s = 0.f;
    for (int k = 0; k < NW; ++k) s += red[k * NG + n];
    v[n] = s;
  }
}

// Two vectors at once (one barrier pair); `red` holds 2 * NW * NG floats.
template <int NG, int NW>
__device__ __forceinline__ void block_sum2(float (&u)[NG], float (&v)[NG], float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NG; ++n) { u[n] = wave_sum(u[n]); v[n] = wave_sum(v[n]); }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NG; ++n) { red[w * NG + n] = u[n]; red[(NW + w) * NG + n] = v[n]; }
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < NW; ++k) { a += red[k * NG + n]; b += red[(NW + k) * NG + n]; }
    u[n] = a; v[n] = b;
  }
}

// Atomic-free row sums of the state-x kernels.  A lane owns kK consecutive arcs of the row-sorted list; rows that end
// inside the lane after its first row end are complete there (plain LDS store by the caller); the sum up to the first
// row end (`head`) still lacks what earlier lanes of the wave accumulated for that row, and the sum after the last row
// end (`tail`, the whole lane when no row ends in it) belongs to a row that ends in a later lane.  A segmented scan
// over the lanes (segments start at lanes with a row end) delivers the carry; the wave's own carry-out goes to
// `wcarry` and is added by the row epilogue (rows crossing a wave block).  LDS float atomics cost ~3 clocks per LANE on
// gfx950 (measured: 1.7 us of a 19 us frame), this costs 6 x (NG + 1) cross-lane moves per wave.
template <int NG>
__device__ __forceinline__ void seg_carry_store(const float (&tail)[NG], const float (&head)[NG], int first_row,
                                                float* acc, float* wcarry) {
  const int lane = threadIdx.x & 63;
  float x[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) x[n] = tail[n];
  int fl = first_row >= 0 ? 1 : 0;
  // segmented inclusive scan: row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast15 into rows 1 and 3 and
  // row_bcast31 into rows 2 and 3 (the gfx9 wave64 scan, with the segment flag travelling along)
  seg_scan_step<NG, 0x111, 0xf>(x, fl);
  seg_scan_step<NG, 0x112, 0xf>(x, fl);
  seg_scan_step<NG, 0x114, 0xf>(x, fl);
  seg_scan_step<NG, 0x118, 0xf>(x, fl);
  seg_scan_step<NG, 0x142, 0xa>(x, fl);
  seg_scan_step<NG, 0x143, 0xc>(x, fl);
  float v[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) v[n] = head[n] + dpp_f<0x138, 0xf>(x[n]);    // wave_shr:1 -- lane 0 receives 0
  if (first_row >= 0) stv<NG>(acc + (size_t)first_row * NG, v);
  if (lane == 63) stv<NG>(wcarry, x);
}

// Row r of the chunk: the stored sum plus the carry-outs of the wave blocks whose last row it is.
template <int NG>
__device__ __forceinline__ void row_value(const float* acc, const float* wcarry, const int (&crow)[kDenWaves], int r,
                                          float (&v)[NG]) {
  ldv<NG>(acc + (size_t)r * NG, v);
#pragma unroll
  for (int k = 0; k + 1 < kDenWaves; ++k) {       // the last block's carry-out is always zero (the chunk ends with a row end)
    if (crow[k] == r) {
#pragma unroll
      for (int n = 0; n < NG; ++n) v[n] += wcarry[k * NG + n];
    }
  }
}

struct IntPack { static constexpr int kN = 64; int32_t v[kN]; };
__global__ void store_ints(IntPack pack, int count, int32_t* out) {
  if ((int)threadIdx.x < count) out[threadIdx.x] = pack.v[threadIdx.x];
}

// Arc records are streamed once per launch; PK2_DEN_NT_ARCS loads them with the non-temporal hint so that they do
// not displace the state vectors the gathers want in L2.
#ifdef PK2_DEN_NT_ARCS
__device__ __forceinline__ int2 pk2_nt_load(const int2* p) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  const v2i v = __builtin_nontemporal_load(reinterpret_cast<const v2i*>(p));
  return make_int2(v.x, v.y);
}
#define PK2_ARC_LD(p) pk2_nt_load(p)
#else
#define PK2_ARC_LD(p) (*(p))
#endif

#ifdef PK2_DEN_PROFILE
// Phase timers of the state-x frame kernel (10 ns ticks of the constant-rate counter), workgroup 0 of each direction.
__device__ unsigned long long g_den_prof[2][8];
#define DEN_T(dir, k) do { if (threadIdx.x == 0 && chunk == 0) { const long long now_ = wall_clock64(); atomicAdd(&g_den_prof[dir][k], (unsigned long long)(now_ - prof_last_)); prof_last_ = now_; } } while (0)
#define DEN_T0() long long prof_last_ = wall_clock64()
__global__ void den_prof_print(int frames) {
  for (int dir = 0; dir < 2; ++dir) {
    printf("den_step_sx %s wg0, avg 10ns-ticks per frame over %d frames: prologue+issue %llu | wait partials(block_sum) %llu | accumulate (records+gathers arrive) %llu | barrier %llu | epilogue rows %llu | final block_sum+store %llu\n",
           dir ? "bwd" : "fwd", frames, g_den_prof[dir][0] / frames, g_den_prof[dir][1] / frames, g_den_prof[dir][2] / frames,
           g_den_prof[dir][3] / frames, g_den_prof[dir][4] / frames, g_den_prof[dir][5] / frames);
    for (int k = 0; k < 8; ++k) g_den_prof[dir][k] = 0;
  }
}
#else
#define DEN_T(dir, k) do { } while (0)
#define DEN_T0() do { } while (0)
#endif

// ----------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------
// xs[g][t][p][n] = exp(clamp(logits[seq][t][p], -30, 30)); 1 for finished / padding sequences.
template <int NG>
__global__ void __launch_bounds__(256) den_exp_transpose(const float* __restrict__ logits,
                                                         int64_t seq_stride, int64_t frame_stride,
                                                         const int32_t* __restrict__ lengths,
                                                         float* __restrict__ xs, int P, int Tmax) {
  const int t = blockIdx.x, g = blockIdx.y;
  const float* rows[NG]; bool live[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    int seq = g * NG + n;
    live[n] = t < lengths[seq];
    rows[n] = logits + (int64_t)seq * seq_stride + (int64_t)t * frame_stride;
  }
  float* out = xs + ((size_t)g * Tmax + t) * (size_t)P * NG;
  for (int p = threadIdx.x; p < P; p += 256) {
    float v[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      // explicit comparisons (not fminf/fmaxf) so that a NaN logit stays NaN and trips the guard,
      // as Kaldi's ApplyExpLimited does
      float x = live[n] ? rows[n][p] : 0.f;
      x = x < -30.f ? -30.f : (x > 30.f ? 30.f : x);
      v[n] = live[n] ? expf(x) : 1.0f;
    }
    stv<NG>(out + (size_t)p * NG, v);
  }
}

// alpha[g][0][s][:] = pi[s]; apart[g][0][0] = sum(pi), other partial slots 0.
template <int NG>
__global__ void __launch_bounds__(256) den_init(DenParams p) {
  const int g = blockIdx.y;
  float* a0 = p.alpha + (size_t)g * (p.Tmax + 1) * (size_t)p.S * NG;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < p.S; s += gridDim.x * 256) {
    float v[NG];
    float x = p.pi[s];
#pragma unroll
    for (int n = 0; n < NG; ++n) v[n] = x;
    stv<NG>(a0 + (size_t)s * NG, v);
  }
  if (blockIdx.x == 0) {
    float* ap = p.apart + (size_t)g * (p.Tmax + 1) * (size_t)p.fwd.n_chunks * NG;
    for (int c = threadIdx.x; c < p.fwd.n_chunks; c += 256) {
      float v[NG];
#pragma unroll
      for (int n = 0; n < NG; ++n) v[n] = (c == 0) ? p.pi_sum : 0.f;
      stv<NG>(ap + (size_t)c * NG, v);
    }
  }
}

// Copies n4 float4 of exp(logits) from global memory straight into LDS with the gfx950 LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, destination = wave-uniform base + lane*16,
// no VGPR round trip).  Asynchronous; the next __syncthreads() drains it.
__device__ __forceinline__ void glds16(const float* gsrc_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int NWAVES>
__device__ __forceinline__ void stage_x_async(const float* src, float* dst, int n4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = w; k * 64 < n4; k += NWAVES) {
    const int i = k * 64 + lane;
    if (i < n4) glds16(src + (size_t)i * 4, dst + (size_t)k * 256);
  }
}

// One frame of the alpha recursion: alpha[t+1] from alpha[t].
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_fwd_step(const DenParams* __restrict__ pp,
                                                            const StepCounter* __restrict__ cnt, int local) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = cnt->base + local;
  if (t >= cnt->T) return;
  const DenParams& p = *pp;  // fields are read with scalar loads; a by-value copy would live in scratch
  float* xs_l = smem;                              // P*NG
  float* acc = xs_l + (size_t)p.P * NG;            // kMaxRows*NG
  float* red = acc + (size_t)2 * kMaxRows * NG;    // 16*NG   (same carve as the backward kernel)
  const int g = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int nc = p.fwd.n_chunks;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  const float* alpha_t = p.alpha + frame * (size_t)p.S * NG;
  const int wb0 = p.fwd.wb_off[chunk], wb1 = p.fwd.wb_off[chunk + 1];
  const int nrows = p.fwd.nrows[chunk];

  // ---- every load that does not depend on LDS is issued here -----------------------------
  float as[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) as[n] = 0.f;
  for (int i = tid; i < nc; i += kDenThreads) {
    float v[NG];
    ldv<NG>(p.apart + (frame * nc + i) * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) as[n] += v[n];
  }
  const float* xsrc = p.xs + ((size_t)g * p.Tmax + t) * (size_t)p.P * NG;
  const int n4 = p.P * NG / 4;
  stage_x_async<kDenWaves>(xsrc, xs_l, n4);
  int wb = wb0 + w;
  int4 rec[kK];
  float a[kK][NG];
  uint2 meta = make_uint2(0u, 0u);
  if (wb < wb1) {
    meta = p.fwd.meta[(size_t)wb * 64 + lane];
#pragma unroll
    for (int j = 0; j < kK; ++j) rec[j] = p.fwd.arcs[((size_t)wb * kK + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < kK; ++j) ldv<NG>(alpha_t + (size_t)rec[j].x * NG, a[j]);
  }
  // ---- LDS: stage exp(logits), clear the accumulator, reduce the partials ------------------
  for (int i = n4 * 4 + tid; i < p.P * NG; i += kDenThreads) xs_l[i] = xsrc[i];
  for (int i = tid; i < nrows * NG; i += kDenThreads) acc[i] = 0.f;
  block_sum<NG, kDenWaves>(as, red);
  if (chunk == 0 && tid == 0) stv<NG>(p.asum + frame * NG, as);
  float lk[NG], inv_as[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { lk[n] = p.leaky * as[n]; inv_as[n] = 1.0f / as[n]; }

  // ---- arcs: acc[dst-row] += (alpha[src]*prob + leaky*asum*pi[src]*prob) * x[pdf] ---------
  while (wb < wb1) {
    int c = (int)meta.x;
    const uint32_t mask = meta.y;
    float sum[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) sum[n] = 0.f;
#pragma unroll
    for (int j = 0; j < kK; ++j) {
      const float prob = __int_as_float(rec[j].z), pr = __int_as_float(rec[j].w);
      float xv[NG];
      ldv<NG>(xs_l + (size_t)rec[j].y * NG, xv);
#pragma unroll
      for (int n = 0; n < NG; ++n) sum[n] += (a[j][n] * prob + lk[n] * pr) * xv[n];
      if ((mask >> j) & 1u) {
#pragma unroll
        for (int n = 0; n < NG; ++n) { atomicAdd(&acc[c * NG + n], sum[n]); sum[n] = 0.f; }
        ++c;
      }
    }
    wb += kDenWaves;
    if (wb < wb1) {
      meta = p.fwd.meta[(size_t)wb * 64 + lane];
#pragma unroll
      for (int j = 0; j < kK; ++j) rec[j] = p.fwd.arcs[((size_t)wb * kK + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < kK; ++j) ldv<NG>(alpha_t + (size_t)rec[j].x * NG, a[j]);
    }
  }
  __syncthreads();

  float* alpha_n = p.alpha + (frame + 1) * (size_t)p.S * NG;
  const int row0 = p.fwd.row0[chunk];
  const bool atomic = p.fwd.atomic[chunk] != 0;
  float loc[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) loc[n] = 0.f;
  for (int r = tid; r < nrows; r += kDenThreads) {
    float v[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) { v[n] = acc[r * NG + n] * inv_as[n]; loc[n] += v[n]; }
    float* o = alpha_n + (size_t)(row0 + r) * NG;
    if (atomic) {
#pragma unroll
      for (int n = 0; n < NG; ++n) atomicAdd(o + n, v[n]);
    } else {
      stv<NG>(o, v);
    }
  }
  block_sum<NG, kDenWaves>(loc, red);
  if (tid == 0) stv<NG>(p.apart + ((frame + 1) * nc + chunk) * NG, loc);
}

// log p_den, 1/tot, and the beta' initialisation constants.
template <int NG>
__global__ void __launch_bounds__(256) den_finalize(DenParams p, float* den_lp) {
  __shared__ double redd[4];
  __shared__ double reda[4];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int seq = g * NG + n;
  const int T = p.lengths[seq];
  const int nc = p.fwd.n_chunks;
  const size_t f0 = (size_t)g * (p.Tmax + 1);
  double acc = 0.0;
  for (int t = tid; t < T; t += 256) acc += log((double)p.asum[(f0 + t) * NG + n]);
  // asum[T] from the partials written by the forward step that produced frame T
  double at = 0.0;
  for (int c = tid; c < nc; c += 256) at += (double)p.apart[((f0 + T) * nc + c) * NG + n];
  acc = wave_sum_d(acc); at = wave_sum_d(at);
  if ((tid & 63) == 0) { redd[tid >> 6] = acc; reda[tid >> 6] = at; }
  __syncthreads();
  if (tid == 0) {
    double s = redd[0] + redd[1] + redd[2] + redd[3];
    double a = reda[0] + reda[1] + reda[2] + reda[3];
    double tot = a * (1.0 + (double)p.leaky * (double)p.pi_sum);
    den_lp[seq] = (T > 0) ? (float)(log(tot) + s) : 0.f;
    p.inv_tot[g * NG + n] = (T > 0) ? (float)(1.0 / tot) : 0.f;
  }
}

// One frame of the beta recursion plus the occupancies of frame t.  Threads 0..511 own beta chunk
// blockIdx.x (arcs by source), threads 512..1023 own occupancy chunk blockIdx.x (arcs by pdf).
template <int NG>
__global__ void __launch_bounds__(kDenBwdThreads) den_bwd_step(const DenParams* __restrict__ pp,
                                                               const StepCounter* __restrict__ cnt, int local) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const DenParams& p = *pp;  // fields are read with scalar loads; a by-value copy would live in scratch
  const int t = p.Tmax - 1 - step;
  float* xs_l = smem;
  float* accB = xs_l + (size_t)p.P * NG;
  float* accG = accB + (size_t)kMaxRows * NG;
  float* red = accG + (size_t)kMaxRows * NG;
  const int g = blockIdx.y, tid = threadIdx.x;
  const int half = __builtin_amdgcn_readfirstlane(tid / kDenThreads), th = tid % kDenThreads;
  const int lane = tid & 63, wh = th >> 6;
  const int ncb = p.bwd.n_chunks, ncg = p.gam.n_chunks;
  const bool beta_role = half == 0;
  const DevOrdering& ord = beta_role ? p.bwd : p.gam;
  const int chunk = blockIdx.x;
  const bool active = chunk < (beta_role ? ncb : ncg);
  float* acc = beta_role ? accB : accG;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  const float* alpha_t = p.alpha + frame * (size_t)p.S * NG;
  const float* beta_n = p.beta + (frame + 1) * (size_t)p.S * NG;
  const int wb0 = active ? ord.wb_off[chunk] : 0, wb1 = active ? ord.wb_off[chunk + 1] : 0;
  const int nrows = active ? ord.nrows[chunk] : 0;
  constexpr int KG = 4;   // arcs per lane fetched at a time (register budget of a 16-wave workgroup)

  // ---- loads that do not depend on LDS ---------------------------------------------------
  float lB[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) lB[n] = 0.f;
  for (int i = tid; i < ncb; i += kDenBwdThreads) {
    float v[NG];
    ldv<NG>(p.bpart + ((frame + 1) * ncb + i) * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) lB[n] += v[n];
  }
  const size_t xoff = ((size_t)g * p.Tmax + t) * (size_t)p.P * NG;
  const int n4 = p.P * NG / 4;
  const float* xsrc = p.xs + xoff;
  stage_x_async<2 * kDenWaves>(xsrc, xs_l, n4);
  int wb = wb0 + wh;
  int4 rec[KG];
  float av[KG][NG], bv[KG][NG];
  uint2 meta = make_uint2(0u, 0u);
  auto fetch = [&](int jg) {
#pragma unroll
    for (int j = 0; j < KG; ++j) rec[j] = ord.arcs[((size_t)wb * kK + jg * KG + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < KG; ++j) {
      if (beta_role) {
        ldv<NG>(beta_n + (size_t)rec[j].x * NG, bv[j]);
      } else {
        ldv<NG>(alpha_t + (size_t)rec[j].x * NG, av[j]);
        ldv<NG>(beta_n + (size_t)rec[j].y * NG, bv[j]);
      }
    }
  };
  if (wb < wb1) {
    meta = ord.meta[(size_t)wb * 64 + lane];
    fetch(0);
  }
  float as[NG];
  ldv<NG>(p.asum + frame * NG, as);

  // ---- LDS: stage exp(logits), clear accumulators, reduce sum pi*beta' of frame t+1 ---------
  for (int i = n4 * 4 + tid; i < p.P * NG; i += kDenBwdThreads) xs_l[i] = xsrc[i];
  for (int i = th; i < nrows * NG; i += kDenThreads) acc[i] = 0.f;
  block_sum<NG, 2 * kDenWaves>(lB, red);
  // beta[t+1][d] = beta'[t+1][d] + leaky * sum_k pi[k] beta'[t+1][k]   (t+1 <  T_n)
  //             = (1 + leaky*sum(pi)) / tot_n                            (t+1 == T_n)
  //             = 0                                                      (t+1 >  T_n)
  float cst[NG], inv_as[NG], lk[NG];
  bool gat[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int T = p.lengths[g * NG + n];
    gat[n] = (t + 1) < T;
    lB[n] = gat[n] ? p.leaky * lB[n] : 0.f;
    cst[n] = ((t + 1) == T) ? p.inv_tot[g * NG + n] * (1.0f + p.leaky * p.pi_sum) : 0.f;
    inv_as[n] = 1.0f / as[n];
    lk[n] = p.leaky * as[n];
  }

  // ---- arcs ---------------------------------------------------------------------------------
  while (wb < wb1) {
    int c = (int)meta.x;
    const uint32_t mask = meta.y;
    float sum[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) sum[n] = 0.f;
#pragma unroll
    for (int jg = 0; jg < kK / KG; ++jg) {
      if (jg > 0) fetch(jg);
#pragma unroll
      for (int j = 0; j < KG; ++j) {
        const float prob = __int_as_float(rec[j].z), pr = __int_as_float(rec[j].w);
        if (beta_role) {
          // rec = {dst, pdf, prob, -}: acc[src-row] += prob * x[pdf] * beta[t+1][dst]
          float xv[NG];
          ldv<NG>(xs_l + (size_t)rec[j].y * NG, xv);
#pragma unroll
          for (int n = 0; n < NG; ++n) sum[n] += prob * xv[n] * (gat[n] ? bv[j][n] + lB[n] : cst[n]);
        } else {
          // rec = {src, dst, prob, pi*prob}: acc[pdf-row] += alpha'[t][src] * prob * beta[t+1][dst]
#pragma unroll
          for (int n = 0; n < NG; ++n)
            sum[n] += (av[j][n] * prob + lk[n] * pr) * (gat[n] ? bv[j][n] + lB[n] : cst[n]);
        }
        if ((mask >> (jg * KG + j)) & 1u) {
#pragma unroll
          for (int n = 0; n < NG; ++n) { atomicAdd(&acc[c * NG + n], sum[n]); sum[n] = 0.f; }
          ++c;
        }
      }
    }
    wb += kDenWaves;
    if (wb < wb1) {
      meta = ord.meta[(size_t)wb * 64 + lane];
      fetch(0);
    }
  }
  __syncthreads();

  const int row0 = active ? ord.row0[chunk] : 0;
  const bool atomic = active && ord.atomic[chunk] != 0;
  float loc[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) loc[n] = 0.f;
  if (beta_role) {
    float* beta_t = p.beta + frame * (size_t)p.S * NG;
    for (int r = th; r < nrows; r += kDenThreads) {
      float v[NG];
      const float pis = p.pi[row0 + r];
#pragma unroll
      for (int n = 0; n < NG; ++n) { v[n] = acc[r * NG + n] * inv_as[n]; loc[n] += pis * v[n]; }
      float* o = beta_t + (size_t)(row0 + r) * NG;
      if (atomic) {
#pragma unroll
        for (int n = 0; n < NG; ++n) atomicAdd(o + n, v[n]);
      } else {
        stv<NG>(o, v);
      }
    }
  } else {
    float* gam_t = p.gamma + xoff;
    for (int r = th; r < nrows; r += kDenThreads) {
      float v[NG], xv[NG];
      ldv<NG>(xs_l + (size_t)(row0 + r) * NG, xv);
#pragma unroll
      for (int n = 0; n < NG; ++n) v[n] = acc[r * NG + n] * xv[n] * inv_as[n];
      float* o = gam_t + (size_t)(row0 + r) * NG;
      if (atomic) {
#pragma unroll
        for (int n = 0; n < NG; ++n) atomicAdd(o + n, v[n]);
      } else {
        stv<NG>(o, v);
      }
    }
  }
  block_sum<NG, 2 * kDenWaves>(loc, red);   // the occupancy half contributes zeros
  if (tid == 0 && chunk < ncb) stv<NG>(p.bpart + (frame * ncb + chunk) * NG, loc);
}

// csum[g][t][0][n] = cu[t] = sum_k (pi[k] + kBetaFloor*sum(pi)/S) btilde'[t,k], the normaliser of the backward recursion;
// csum[g][t][1][n] = sum_k pi[k] btilde'[t,k] / cu[t], the factor of its leaky-HMM term
// (fixed-order reduction of the backward partials {sum pi*btilde', sum btilde'} per chunk).
template <int NG>
__global__ void __launch_bounds__(256) den_csum(DenParams p, float* csum) {
  __shared__ float red[2 * 4 * NG];
  const int t = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int ncb = p.bwd.n_chunks;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  float c[NG], u[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { c[n] = 0.f; u[n] = 0.f; }
  for (int i = tid; i < ncb; i += 256) {
    float v[NG], w[NG];
    ldv<NG>(p.bpart + ((frame * ncb + i) * 2) * NG, v);
    ldv<NG>(p.bpart + ((frame * ncb + i) * 2 + 1) * NG, w);
#pragma unroll
    for (int n = 0; n < NG; ++n) { c[n] += v[n]; u[n] += w[n]; }
  }
  block_sum2<NG, 4>(c, u, red);
  if (tid == 0) {
    float cu[NG], ratio[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) { cu[n] = c[n] + p.wu * u[n]; ratio[n] = cu[n] > 0.f ? c[n] / cu[n] : 0.f; }
    stv<NG>(csum + (frame * 2) * NG, cu);
    stv<NG>(csum + (frame * 2 + 1) * NG, ratio);
  }
}

// K[g][t][n]: true beta[t] = K[t] * betahat[t].  K[T] = sum(pi)/tot, log K[t] = log K[t+1] + log cu[t]
// - log asum[t]: a suffix sum in double precision (block scan, one workgroup per sequence).
// check = sum_h alpha'[0,h] beta'[0,h] = (1 + leaky*sum(pi)) * K[0] * (sum pi*btilde'[0]) / cu[0].
template <int NG>
__global__ void __launch_bounds__(256) den_scales(DenParams p, const float* csum, float* Kf, float* check) {
  __shared__ double wsum[4];
  __shared__ double carry_s;
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = p.lengths[g * NG + n];
  const size_t f0 = (size_t)g * (p.Tmax + 1);
  for (int t = T + 1 + tid; t <= p.Tmax; t += 256) Kf[(f0 + t) * NG + n] = 0.f;
  if (T <= 0) { if (tid == 0) check[g * NG + n] = 1.f; return; }
  const double lkT = log((double)p.pi_sum) + log((double)p.inv_tot[g * NG + n]) + log((double)p.beta_seed);
  if (tid == 0) { Kf[(f0 + T) * NG + n] = (float)exp(lkT); carry_s = lkT; }
  __syncthreads();
  // walk t = T-1 .. 0 in tiles of 256 (thread i of a tile handles t = hi - i)
  for (int hi = T - 1; hi >= 0; hi -= 256) {
    const int t = hi - tid;
    double v = 0.0;
    if (t >= 0) v = log((double)csum[((f0 + t) * 2) * NG + n]) - log((double)p.asum[(f0 + t) * NG + n]);
    // inclusive scan over the tile
    double x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    double off = carry_s;
    for (int k = 0; k < w; ++k) off += wsum[k];
    const double lk = off + x;
    if (t >= 0) Kf[(f0 + t) * NG + n] = (float)exp(lk);
    __syncthreads();
    if (tid == 255 || t == 0) { carry_s = lk; if (t == 0) check[g * NG + n] = (float)((1.0 + (double)p.leaky * p.pi_sum) * exp(lk) * (double)csum[(f0 * 2 + 1) * NG + n]); }
    __syncthreads();
  }
}

// Round 6 (launches per step): den_csum + den_finalize + the persistent kernel's check + den_scales of the
// NG = 1 state-x path in ONE launch, one workgroup per sequence.  Each phase is the kernel above it, instruction for
// instruction in its arithmetic (same partial sums in the same order: the per-frame sums of den_csum take one wave, the other
// three contributed zeros there), with the frame normalisers handed from phase to phase in LDS.  `ctl` (optional): the
// control block of the persistent launch that has just run -- every workgroup reads its verdict first, the last one to have
// done so zeroes the block for the next launch (what den_persist2_check / den_persist_check did in a launch of their own).
__global__ void __launch_bounds__(256) den_tail1(DenParams p, float* csum, float* den_lp, float* Kf, float* check, DenTailCheck ck) {
  extern __shared__ float cs[];            // [(Tmax + 1)][2]: cu, ratio
  __shared__ double redd[4], reda[4], wsum[4];
  __shared__ double carry_s;
  __shared__ unsigned s_bad, s_last;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = p.lengths[g];
  const int ncb = p.bwd.n_chunks, nc = p.fwd.n_chunks;
  const size_t f0 = (size_t)g * (p.Tmax + 1);
  if (tid == 0) {
    s_bad = 0u; s_last = 0u;
    if (ck.ctl) {
      const unsigned ab = __hip_atomic_load(ck.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned dn = __hip_atomic_load(ck.done_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_bad = (ab != 0u || dn != (unsigned)ck.ntasks || ck.ntasks < 0) ? 1u : 0u;        // (ntasks < 0: test hook)
      s_last = __hip_atomic_fetch_add(ck.count_word, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    }
  }
  // ---- den_csum.  The persistent kernels leave 32 partial pairs per frame (256 contiguous bytes): a THREAD per frame, added
  // in the order of the 64-lane butterfly the kernel of its own used (offsets 16, 8, 4, 2, 1 over 32 values; the upper 32
  // lanes held zeros) -- the same bits, without a dependent round trip per frame (as a wave per frame this phase took 80 us)
  if (ncb == 32) {
    for (int t = tid; t <= p.Tmax; t += 256) {
      const float4* src = reinterpret_cast<const float4*>(p.bpart + (f0 + t) * 64);
      float c[32], u[32];
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float4 v = src[i]; c[2 * i] = v.x; u[2 * i] = v.y; c[2 * i + 1] = v.z; u[2 * i + 1] = v.w; }
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1)
#pragma unroll
        for (int i = 0; i < o; ++i) { c[i] += c[i + o]; u[i] += u[i + o]; }
      const float cu = c[0] + p.wu * u[0], ratio = cu > 0.f ? c[0] / cu : 0.f;
      cs[2 * t] = cu; cs[2 * t + 1] = ratio;
      csum[(f0 + t) * 2] = cu; csum[(f0 + t) * 2 + 1] = ratio;
    }
  } else
  for (int t = w; t <= p.Tmax; t += 4) {
    float c = 0.f, u = 0.f;
    for (int i = lane; i < ncb; i += 64) {
      c += p.bpart[((f0 + t) * ncb + i) * 2];
      u += p.bpart[((f0 + t) * ncb + i) * 2 + 1];
    }
    c = wave_sum(c); u = wave_sum(u);
    if (lane == 0) {
      const float cu = c + p.wu * u, ratio = cu > 0.f ? c / cu : 0.f;
      cs[2 * t] = cu; cs[2 * t + 1] = ratio;
      csum[(f0 + t) * 2] = cu; csum[(f0 + t) * 2 + 1] = ratio;
    }
  }
  // ---- den_finalize
  double acc = 0.0, at = 0.0;
  for (int t = tid; t < T; t += 256) acc += log((double)p.asum[f0 + t]);
  for (int c = tid; c < nc; c += 256) at += (double)p.apart[(f0 + T) * nc + c];
  acc = wave_sum_d(acc); at = wave_sum_d(at);
  if (lane == 0) { redd[w] = acc; reda[w] = at; }
  __syncthreads();
  const double s_log = redd[0] + redd[1] + redd[2] + redd[3];
  const double a_tot = reda[0] + reda[1] + reda[2] + reda[3];
  const double tot = a_tot * (1.0 + (double)p.leaky * (double)p.pi_sum);
  const float inv_tot = (T > 0) ? (float)(1.0 / tot) : 0.f;
  if (tid == 0) {
    float lp = (T > 0) ? (float)(log(tot) + s_log) : 0.f;
    if (s_bad) lp = __uint_as_float(0x7fc00000u);          // (den_persist2_check: a launch that gave up poisons its results)
    den_lp[g] = lp;
    p.inv_tot[g] = inv_tot;
    if (s_bad) persist_guard_raise(ck.guard_dev, ck.guard_host);
  }
  if (s_last) {          // every workgroup has read the verdict: the control block starts the next launch at zero
    for (int i = tid; i < ck.ctl_words; i += 256) ck.ctl[i] = 0u;
  }
  // ---- den_scales
  for (int t = T + 1 + tid; t <= p.Tmax; t += 256) Kf[f0 + t] = 0.f;
  if (T <= 0) { if (tid == 0) check[g] = 1.f; return; }
  const double lkT = log((double)p.pi_sum) + log((double)inv_tot) + log((double)p.beta_seed);
  if (tid == 0) { Kf[f0 + T] = (float)exp(lkT); carry_s = lkT; }
  __syncthreads();
  for (int hi = T - 1; hi >= 0; hi -= 256) {
    const int t = hi - tid;
    double v = 0.0;
    if (t >= 0) v = log((double)cs[2 * t]) - log((double)p.asum[f0 + t]);
    double x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    double off = carry_s;
    for (int k = 0; k < w; ++k) off += wsum[k];
    const double lk = off + x;
    if (t >= 0) Kf[f0 + t] = (float)exp(lk);
    __syncthreads();
    if (tid == 255 || t == 0) { carry_s = lk; if (t == 0) check[g] = (float)((1.0 + (double)p.leaky * p.pi_sum) * exp(lk) * (double)cs[1]); }
    __syncthreads();
  }
}

// Round 6: the passes in front of the recursions in ONE launch -- exp(logits) rows, the first alpha frame, the zeroed backward
// partial sums and (when the caller wants its gradient rows zeroed: the numerator adds into them) the gradient rows -- as
// independent jobs over ranges of blockIdx.x; each job is the kernel it replaces (den_exp_rows, den_init<1>, the memset of
// bpart, zero_rows).
struct DenPrepJobs { int exp_blocks, init_blocks, init_bx, bpart_blocks, zero_blocks; size_t bpart_floats;
                     const float* logits; int64_t lss, lfs; float* xp; float* grad; int64_t gss, gfs; int N; };
__global__ void __launch_bounds__(256) den_prep1(DenParams p, DenPrepJobs j) {
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b < j.exp_blocks) {
    const int t = b % p.Tmax, g = b / p.Tmax;
    const bool live = t < p.lengths[g];
    const float* row = j.logits + (int64_t)g * j.lss + (int64_t)t * j.lfs;
    float* out = j.xp + ((size_t)g * p.Tmax + t) * (size_t)p.P;
    for (int q = tid; q < p.P; q += 256) {
      float xx = live ? row[q] : 0.f;
      xx = xx < -30.f ? -30.f : (xx > 30.f ? 30.f : xx);
      out[q] = live ? expf(xx) : 1.0f;
    }
    return;
  }
  b -= j.exp_blocks;
  if (b < j.init_blocks) {
    const int bx = b % j.init_bx, g = b / j.init_bx;
    float* a0 = p.alpha + (size_t)g * (p.Tmax + 1) * (size_t)p.S;
    for (int s = bx * 256 + tid; s < p.S; s += j.init_bx * 256) a0[s] = p.pi[s];
    if (bx == 0) {
      float* ap = p.apart + (size_t)g * (p.Tmax + 1) * (size_t)p.fwd.n_chunks;
      for (int c = tid; c < p.fwd.n_chunks; c += 256) ap[c] = (c == 0) ? p.pi_sum : 0.f;
    }
    return;
  }
  b -= j.init_blocks;
  if (b < j.bpart_blocks) {
    for (size_t i = (size_t)b * 256 + tid; i < j.bpart_floats; i += (size_t)j.bpart_blocks * 256) p.bpart[i] = 0.f;
    return;
  }
  b -= j.bpart_blocks;
  {
    const int t = b % p.Tmax, n = b / p.Tmax;
    float* row = j.grad + (int64_t)n * j.gss + (int64_t)t * j.gfs;
    for (int q = tid; q < p.P; q += 256) row[q] = 0.f;
  }
}

// Occupancies without touching the arcs, for graphs whose pdf is a function of the destination
// state: the arcs entering state d at frame t carry total posterior alpha[t+1,d] * beta[t+1,d]
// (alpha before, beta after the leaky-HMM term), so
//   gamma[t,p] = sum_{d : pdf(d) = p} alpha[t+1,d] * beta[t+1,d],
//   beta[t+1,d] = K[t+1] * (btilde'[t+1,d]/c[t+1] + leaky)   (= K[T] * (1/sum(pi) + leaky) at t+1 = T).
// No serial dependence: one launch covers all frames.
template <int NG>
__device__ __forceinline__ void den_gamma_states_body(const DenParams& p, const float* csum, const float* Kf, int t, int g) {
  const int tid = threadIdx.x;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  float cv[NG], rv[NG], kv[NG], inv_c[NG], lkr[NG], cst[NG];
  bool gat[NG];
  ldv<NG>(csum + ((frame + 1) * 2) * NG, cv);
  ldv<NG>(csum + ((frame + 1) * 2 + 1) * NG, rv);
  ldv<NG>(Kf + (frame + 1) * NG, kv);
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int T = p.lengths[g * NG + n];
    gat[n] = (t + 1) < T;
    inv_c[n] = (gat[n] && cv[n] > 0.f) ? 1.0f / cv[n] : 0.f;
    lkr[n] = p.leaky * rv[n];
    cst[n] = ((t + 1) == T) ? (1.0f / p.pi_sum + p.leaky) : 0.f;
  }
  const float* alpha_n = p.alphav + (frame + 1) * (size_t)p.Vo * NG;       // per occupancy state
  const float* beta_n = p.beta + (frame + 1) * (size_t)p.V * (p.brec * NG);   // {btilde'[NG], xd[NG]} per virtual state (brec = 2)
  float* gam_t = p.gamma + ((size_t)g * p.Tmax + t) * (size_t)p.P * NG;
  for (int pdf = tid; pdf < p.P; pdf += 256) {
    float v[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) v[n] = 0.f;
    for (int k = p.ps_off[pdf]; k < p.ps_off[pdf + 1]; ++k) {
      const int s = p.ps_state[k];
      float a[NG], b[NG];
      ldv<NG>(alpha_n + (size_t)s * NG, a);
      ldv<NG>(beta_n + (size_t)p.ovirt[s] * (p.brec * NG), b);
#pragma unroll
      for (int n = 0; n < NG; ++n) v[n] += a[n] * (gat[n] ? b[n] * inv_c[n] + lkr[n] : cst[n]);
    }
#pragma unroll
    for (int n = 0; n < NG; ++n) v[n] *= kv[n];
    stv<NG>(gam_t + (size_t)pdf * NG, v);
  }
}

// The same sums with the states walked IN ORDER (coalesced alpha / beta rows at streaming speed instead of two
// dependent gathers per state through the pdf -> states lists) and one row of P x NG occupancies accumulated in LDS
// (97 KB for P = 6048, NG = 4); used whenever that row fits.  The row is FIXED POINT: a frame's occupancies are
// posteriors (each <= 1, summing to 1), so a term is scaled by the frame's K and 2^30 and added with ds_add_u32.
// LDS float atomics run at ~1 lane per 3 clocks on gfx950 (this pass: 0.92 ms with ds_add_f32, 0.23 ms with integer
// adds or with no atomics at all); the quantisation is 5e-10 per term against a 1e-4 tolerance, and the sums no longer
// depend on the order of the atomics.
constexpr int kGammaThreads = 1024;   // one workgroup per CU (the LDS row): 16 waves keep 4 x 3 loads each in flight
constexpr size_t kGammaMaxLds = 144 * 1024;
constexpr float kGammaFix = 1073741824.f;   // 2^30
template <int NG>
__device__ __forceinline__ void den_gamma_states_lds_body(const DenParams& p, const float* csum, const float* Kf, int t, int g,
                                                          unsigned* acc) {
  const int tid = threadIdx.x;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  float cv[NG], rv[NG], kv[NG], inv_c[NG], lkr[NG], cst[NG];
  bool gat[NG];
  ldv<NG>(csum + ((frame + 1) * 2) * NG, cv);
  ldv<NG>(csum + ((frame + 1) * 2 + 1) * NG, rv);
  ldv<NG>(Kf + (frame + 1) * NG, kv);
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int T = p.lengths[g * NG + n];
    gat[n] = (t + 1) < T;
    inv_c[n] = (gat[n] && cv[n] > 0.f) ? 1.0f / cv[n] : 0.f;
    lkr[n] = p.leaky * rv[n];
    cst[n] = ((t + 1) == T) ? (1.0f / p.pi_sum + p.leaky) : 0.f;
  }
  bool any = false;
#pragma unroll
  for (int n = 0; n < NG; ++n) any = any || t < p.lengths[g * NG + n];
  if (!any) return;      // nothing reads the occupancies of a frame past every sequence of the group (and the persistent
                         // kernel leaves alpha / beta of such frames unwritten)
  for (int i = tid; i < p.P * NG; i += kGammaThreads) acc[i] = 0u;

  __syncthreads();
  const float* alpha_n = p.alphav + (frame + 1) * (size_t)p.Vo * NG;      // per occupancy state
  const float* beta_n = p.beta + (frame + 1) * (size_t)p.V * (p.brec * NG);   // {btilde'[NG], xd[NG]} per virtual state (brec = 2)
  for (int s0 = tid; s0 < p.Vo; s0 += 4 * kGammaThreads) {
    float a[4][NG], b[4][NG]; int pdf[4], vi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = s0 + q * kGammaThreads;
      pdf[q] = -1; vi[q] = 0;
      if (s < p.Vo) {
        pdf[q] = p.opdf[s];
        vi[q] = p.ovirt[s];
        ldv<NG>(alpha_n + (size_t)s * NG, a[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (pdf[q] >= 0) ldv<NG>(beta_n + (size_t)vi[q] * (p.brec * NG), b[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pdf[q] < 0) continue;
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        // (alpha * beta-hat) * K <= 1 first, the fixed-point scale last: K alone can be near the float range
        const float w = a[q][n] * (gat[n] ? b[q][n] * inv_c[n] + lkr[n] : cst[n]) * kv[n] * kGammaFix;
        atomicAdd(&acc[n * p.P + pdf[q]], (unsigned)(w + 0.5f));
      }
    }
  }
  __syncthreads();
  float* gam_t = p.gamma + ((size_t)g * p.Tmax + t) * (size_t)p.P * NG;
  // the LDS row is sequence-major (acc[n][pdf]: the lanes of one ds_add_f32 then spread over all banks instead of
  // every fourth one); the occupancies go out pdf-major
  for (int i = tid; i < p.P * NG; i += kGammaThreads) gam_t[i] = (float)acc[(i % NG) * p.P + i / NG] * (1.0f / kGammaFix);
}

template <int NG>
__global__ void __launch_bounds__(256) den_gamma_states(DenParams p, const float* csum, const float* Kf) {
  den_gamma_states_body<NG>(p, csum, Kf, blockIdx.x, blockIdx.y);
}

template <int NG>
__global__ void __launch_bounds__(kGammaThreads) den_gamma_states_lds(DenParams p, const float* csum, const float* Kf) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  den_gamma_states_lds_body<NG>(p, csum, Kf, blockIdx.x, blockIdx.y, reinterpret_cast<unsigned*>(smem));
}

// The same launch also carries the numerator: workgroups x >= Tmax of group 0 run the forward-backward of one
// supervision FST each (chain_num.h).  The numerator is a ~1 ms latency-bound job of one workgroup per sequence;
// here it overlaps the occupancy pass instead of holding the stream alone.  LDS: the occupancy row, or (numerator
// workgroups) 2 * states + 8 floats.
template <int NG, bool LDS_ROW>
__global__ void __launch_bounds__(LDS_ROW ? kGammaThreads : 256) den_gamma_states_num(DenParams p, const float* csum, const float* Kf,
                                                                                     NumParams np, int n_seq, int stage) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < p.Tmax) {
    if (LDS_ROW) den_gamma_states_lds_body<NG>(p, csum, Kf, blockIdx.x, blockIdx.y, reinterpret_cast<unsigned*>(smem));
    else den_gamma_states_body<NG>(p, csum, Kf, blockIdx.x, blockIdx.y);
    return;
  }
  const int n = (int)blockIdx.x - p.Tmax;
  if (blockIdx.y != 0 || n >= n_seq) return;
  if (stage) {                         // alpha and beta chains on one wave each (chain_num.h)
    if (threadIdx.x >= 128) return;
    num_fwd_bwd_two_waves(np, n, smem);
  } else {
    if (threadIdx.x >= 64) return;     // one wave per supervision
    num_fwd_bwd_body<64, false>(np, n, smem);
  }
}

// ----------------------------------------------------------------------------------------
// "state-x" fast path, for graphs whose pdf is a function of the destination state.
// exp(logit) of the pdf a state emits is then a per-STATE quantity xd[t][d] = x[t, pdf(d)]:
//  * forward: it factors out of the arc sum, alpha[t+1,d] = xd[t][d]/asum * sum_arcs alpha'[t,s] prob;
//  * backward: it is stored right behind beta'[t+1][d] ({beta'[NG], xd[NG]} = 32 B per state), so the
//    one gather an arc needs anyway brings it along.
// Neither kernel stages the P x NG table of exp(logits) in LDS (saves 97 KB of LDS and 24 MB of
// L2 traffic per frame and direction); they keep only the 16 KB row accumulator.
// ----------------------------------------------------------------------------------------
// exp(clamp(logit[seq(g,n)][t][pdf(i)])) for a table of pdfs (1 where pdf < 0 and for finished sequences), written as
// records of `rec` floats per entry with the NG values at float offset `xo` (the floats before them are zeroed):
//   bx: entry = virtual state v, record {btilde' = 0, x} of frame t+1 (rec = 2*NG, xo = NG, frame_shift = 1);
//   xl: entry = state, x of its peeled self-loop's pdf, frame t (rec = NG, xo = 0, frame_shift = 0).
template <int NG>
__device__ __forceinline__ void den_exp_table(const float* __restrict__ logits, int64_t seq_stride,
                                              int64_t frame_stride, const int32_t* __restrict__ lengths,
                                              const int32_t* __restrict__ entry_pdf, float* __restrict__ out_base,
                                              int S /* entries */, int frames_per_group, int frame_shift, int rec, int xo,
                                              int slice, int num_slices) {
  const int t = blockIdx.x, g = blockIdx.y;
  const float* rows[NG]; bool live[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    int seq = g * NG + n;
    live[n] = t < lengths[seq];
    rows[n] = logits + (int64_t)seq * seq_stride + (int64_t)t * frame_stride;
  }
  // The whole {btilde', x} record of a virtual state is written (btilde' = 0 until the backward frame fills it): full
  // 32-byte records coalesce into whole lines, the x halves alone are partial-line writes.  Four entries per thread and
  // pass keep the 4 x NG row gathers of each in flight together.
  float* out = out_base + ((size_t)g * frames_per_group + t + frame_shift) * (size_t)S * rec;
  for (int d0 = slice * (4 * 256) + threadIdx.x; d0 < S; d0 += num_slices * (4 * 256)) {
    int pdf[4]; float x[4][NG];
#pragma unroll
    for (int q = 0; q < 4; ++q) pdf[q] = d0 + q * 256 < S ? entry_pdf[d0 + q * 256] : -1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < NG; ++n) x[q][n] = (live[n] && pdf[q] >= 0) ? rows[n][pdf[q]] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int d = d0 + q * 256;
      if (d >= S) continue;
      float v[NG], zero[NG];
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        float xx = x[q][n];
        xx = xx < -30.f ? -30.f : (xx > 30.f ? 30.f : xx);   // keeps NaN (see den_exp_transpose)
        v[n] = (live[n] && pdf[q] >= 0) ? expf(xx) : 1.0f;
        zero[n] = 0.f;
      }
      if (xo) stv<NG>(out + (size_t)d * rec, zero);
      stv<NG>(out + (size_t)d * rec + xo, v);
    }
  }
}

// Both tables in one launch: slices [0, zv) of grid.z write the virtual states' records, the rest the loop pdfs' x.
template <int NG>
__global__ void __launch_bounds__(256) den_exp_states(const float* __restrict__ logits, int64_t seq_stride, int64_t frame_stride,
                                                      const int32_t* __restrict__ lengths, const int32_t* __restrict__ vpdf,
                                                      float* __restrict__ bx, int V, const int32_t* __restrict__ loop_pdf,
                                                      float* __restrict__ xl, int S, int Tmax, int zv) {
  const int z = blockIdx.z;
  if (z < zv) den_exp_table<NG>(logits, seq_stride, frame_stride, lengths, vpdf, bx, V, Tmax + 1, 1, 2 * NG, NG, z, zv);
  else den_exp_table<NG>(logits, seq_stride, frame_stride, lengths, loop_pdf, xl, S, Tmax, 0, NG, 0, z - zv, (int)gridDim.z - zv);
}

// The same two tables through LDS: a workgroup first stages exp(clamp(logits)) of its frame for all pdfs (coalesced row
// reads, P x NG values, one expf per (pdf, sequence) instead of one per table entry), then every entry is an index load,
// one 16-byte LDS read and a store.  The direct version above is bound by its 4-byte row gathers (4 per entry: 212 M
// per call at ~0.5 lines/clk/CU = 0.75 ms); used whenever the staged row fits LDS.
constexpr int kExpThreads = 1024;
template <int NG>
__global__ void __launch_bounds__(kExpThreads) den_exp_states_lds(const float* __restrict__ logits, int64_t seq_stride,
                                                                  int64_t frame_stride, const int32_t* __restrict__ lengths,
                                                                  const int32_t* __restrict__ vpdf, float* __restrict__ bx, int V,
                                                                  const int32_t* __restrict__ loop_pdf, float* __restrict__ xl,
                                                                  int S, int P, int Tmax, float* __restrict__ xv) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [P][NG]
  const int t = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const float* rows[NG]; bool live[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int seq = g * NG + n;
    live[n] = t < lengths[seq];
    rows[n] = logits + (int64_t)seq * seq_stride + (int64_t)t * frame_stride;
  }
  for (int p = tid; p < P; p += kExpThreads) {
    float v[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      float xx = live[n] ? rows[n][p] : 0.f;
      xx = xx < -30.f ? -30.f : (xx > 30.f ? 30.f : xx);   // keeps NaN (see den_exp_transpose)
      v[n] = live[n] ? expf(xx) : 1.0f;
    }
    stv<NG>(xs + (size_t)p * NG, v);
  }
  __syncthreads();
  float one[NG], zero[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { one[n] = 1.f; zero[n] = 0.f; }
  float* out = bx + ((size_t)g * (Tmax + 1) + t + 1) * (size_t)V * (2 * NG);
  for (int d = blockIdx.z * kExpThreads + tid; d < V; d += gridDim.z * kExpThreads) {
    const int pdf = vpdf[d];
    float v[NG];
    if (pdf >= 0) ldv<NG>(xs + (size_t)pdf * NG, v);
    if (NG == 1 && xv) {      // the persistent kernel reads a compact copy and fills the btilde' halves itself: the records' x
                              // halves have no reader then (566 MB of stores per call on the bench graph)
      xv[((size_t)g * Tmax + t) * (size_t)V + d] = pdf >= 0 ? v[0] : 1.f;
      continue;
    }
    stv<NG>(out + (size_t)d * (2 * NG), zero);
    stv<NG>(out + (size_t)d * (2 * NG) + NG, pdf >= 0 ? v : one);
  }
  float* outl = xl + ((size_t)g * Tmax + t) * (size_t)S * NG;
  for (int d = blockIdx.z * kExpThreads + tid; d < S; d += gridDim.z * kExpThreads) {
    const int pdf = loop_pdf[d];
    float v[NG];
    if (pdf >= 0) ldv<NG>(xs + (size_t)pdf * NG, v);
    stv<NG>(outl + (size_t)d * NG, pdf >= 0 ? v : one);
  }
}

// exp(logits) as it is, one row of P entries per frame (1 beyond a sequence's length): what the second persistent kernel
// gathers its x from by pdf (round 4) -- 57 MB per call on the bench minibatch instead of the 2 x 283 MB of den_exp_states_lds'
// copies per virtual state and per state.
__global__ void __launch_bounds__(256) den_exp_rows(const float* __restrict__ logits, int64_t seq_stride, int64_t frame_stride,
                                                    const int32_t* __restrict__ lengths, float* __restrict__ xp, int P, int Tmax) {
  const int t = blockIdx.x, g = blockIdx.y;
  const bool live = t < lengths[g];
  const float* row = logits + (int64_t)g * seq_stride + (int64_t)t * frame_stride;
  float* out = xp + ((size_t)g * Tmax + t) * (size_t)P;
  for (int p = threadIdx.x; p < P; p += 256) {
    float xx = live ? row[p] : 0.f;
    xx = xx < -30.f ? -30.f : (xx > 30.f ? 30.f : xx);   // keeps NaN (see den_exp_transpose)
    out[p] = live ? expf(xx) : 1.0f;
  }
}

template <int NG>
__device__ __forceinline__ void den_fwd_frame_sx(const DenParams& p, int t, int chunk, float* acc, float* red, float* wcarry) {
  DEN_T0();
  const int g = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int nc = p.fwd.n_chunks;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  const float* alpha_t = p.alpha + frame * (size_t)p.S * NG;
  const int wb0 = p.fwd.wb_off[chunk], wb1 = p.fwd.wb_off[chunk + 1];
  float as[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) as[n] = 0.f;
  for (int i = tid; i < nc; i += kDenThreads) {
    float v[NG];
    ldv<NG>(p.apart + (frame * nc + i) * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) as[n] += v[n];
  }
  int wb = wb0 + w;
  int2 rec[kK];      // {gathered state, arc probability}
  float a[kK][NG];
  uint2 meta = make_uint2(0u, 0u);
  // what the row epilogue needs and that does not depend on this frame is fetched now, not after the arc loop
  const int row0 = p.fwd.row0[chunk];
  const int real0 = p.fwd.real0[chunk], nreal = p.fwd.nreal[chunk];
  const bool atomic = p.fwd.atomic[chunk] != 0;
  const float* leak = p.fwd.row_leak + p.fwd.slot0[chunk];     // sum of pi[src]*prob over this piece of the row
  // first state of this thread: virtual-row range, first occupancy slot, and the peeled self-loop's inputs
  // (alpha[t,d] was finished by the previous launch; x of the loop pdf and the probabilities are per-call constants)
  const float* xl_t = p.xl + ((size_t)g * p.Tmax + t) * (size_t)p.S * NG;
  int pf_lo = 0, pf_hi = 0, pf_o = 0; float pf_pl = 0.f, pf_pi = 0.f, pf_al[NG], pf_xl[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { pf_al[n] = 0.f; pf_xl[n] = 0.f; }
  if (tid < nreal) {
    const int d = real0 + tid;
    pf_lo = p.voff[d]; pf_hi = p.voff[d + 1]; pf_o = p.ooff[d];
    pf_pl = p.loop_prob[d]; pf_pi = p.pi[d];
    ldv<NG>(alpha_t + (size_t)d * NG, pf_al);
    ldv<NG>(xl_t + (size_t)d * NG, pf_xl);
  }
  if (wb < wb1) {
    meta = p.fwd.meta[(size_t)wb * 64 + lane];
#pragma unroll
    for (int j = 0; j < kK; ++j) rec[j] = PK2_ARC_LD(&p.fwd.arcs2[((size_t)wb * kK + ((p.debug & 4) ? (j & 3) : j)) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < kK; ++j) ldv<NG>(alpha_t + (size_t)((p.debug & 1) ? 0 : rec[j].x) * NG, a[j]);
  }
  // carry-outs of the wave blocks (chunk-local row they belong to: crow); every row of `acc` gets exactly one store
  int crow[kDenWaves];
#pragma unroll
  for (int k = 0; k < kDenWaves; ++k) crow[k] = wb0 + k < wb1 ? p.fwd.wb_crow[wb0 + k] : -1;
  if (tid < kDenWaves * NG) wcarry[tid] = 0.f;
  DEN_T(0, 0);
  block_sum<NG, kDenWaves>(as, red);
  DEN_T(0, 1);
  if (chunk == 0 && tid == 0) stv<NG>(p.asum + frame * NG, as);
  float lk[NG], inv_as[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { lk[n] = p.leaky * as[n]; inv_as[n] = 1.0f / as[n]; }
  if (p.debug & 2) wb = wb1;
  if (wb < wb1) {           // one wave block per wave (a chunk has at most kDenWaves of them)
    int c = (int)meta.x, first_row = -1;
    const uint32_t mask = meta.y;
    float sum[NG], head[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) { sum[n] = 0.f; head[n] = 0.f; }
#pragma unroll
    for (int j = 0; j < kK; ++j) {
      const float prob = __int_as_float(rec[j].y);
#pragma unroll
      for (int n = 0; n < NG; ++n) sum[n] += a[j][n] * prob;
      if ((mask >> j) & 1u) {
        if (first_row < 0) {
          first_row = c;
#pragma unroll
          for (int n = 0; n < NG; ++n) head[n] = sum[n];
        } else {
          stv<NG>(acc + (size_t)c * NG, sum);
        }
#pragma unroll
        for (int n = 0; n < NG; ++n) sum[n] = 0.f;
        ++c;
      }
    }
    seg_carry_store<NG>(sum, head, first_row, acc, wcarry + (size_t)(wb - wb0) * NG);
  }
  DEN_T(0, 2);
  __syncthreads();
  DEN_T(0, 3);
  // Rows are VIRTUAL states (dst, pdf): alpha_v[t+1,v] = x[t,pdf(v)]/asum * (sum over the row + leaky term); the alpha
  // of a state is the sum over its virtual states, which all sit in this chunk (chain_graph.hip), a thread per state.
  float* alpha_n = p.alpha + (frame + 1) * (size_t)p.S * NG;
  float* alphav_n = p.alphav + (frame + 1) * (size_t)p.Vo * NG;
  const bool sep = p.alphav != p.alpha;
  const float* xd = p.beta + (frame + 1) * (size_t)p.V * (2 * NG) + NG;   // x[t, pdf(v)]
  float loc[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) loc[n] = 0.f;
  if (atomic) {            // one (piece of a) virtual row of a state whose rows do not fit one chunk
    if (tid == 0) {
      float xv[NG];
      ldv<NG>(xd + (size_t)row0 * (2 * NG), xv);
      const int o = pf_o + (row0 - pf_lo);
      float a0[NG];
      row_value<NG>(acc, wcarry, crow, 0, a0);
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        const float v = (a0[n] + lk[n] * leak[0]) * xv[n] * inv_as[n];
        loc[n] = v;
        atomicAdd(alpha_n + (size_t)real0 * NG + n, v);
        if (sep) atomicAdd(alphav_n + (size_t)o * NG + n, v);
      }
      if (p.fwd.atomic[chunk] == 2 && pf_pl > 0.f) {   // the state's first piece also adds its peeled self-loop
        const int ol = pf_o + (pf_hi - pf_lo);
#pragma unroll
        for (int n = 0; n < NG; ++n) {
          const float v = (pf_al[n] + lk[n] * pf_pi) * pf_pl * pf_xl[n] * inv_as[n];
          loc[n] += v;
          atomicAdd(alpha_n + (size_t)real0 * NG + n, v);
          atomicAdd(alphav_n + (size_t)ol * NG + n, v);
        }
      }
    }
  } else {
    for (int r = tid; r < nreal; r += kDenThreads) {
      const int d = real0 + r;
      const bool pf = r == tid;
      const int lo = (pf ? pf_lo : p.voff[d]) - row0, hi = (pf ? pf_hi : p.voff[d + 1]) - row0;
      const int o0 = pf ? pf_o : p.ooff[d];
      const float pl = pf ? pf_pl : p.loop_prob[d];
      float sum[NG];
#pragma unroll
      for (int n = 0; n < NG; ++n) sum[n] = 0.f;
      for (int q = lo; q < hi; ++q) {
        float v[NG], xv[NG], aq[NG];
        ldv<NG>(xd + (size_t)(row0 + q) * (2 * NG), xv);
        const float lr = leak[q];
        row_value<NG>(acc, wcarry, crow, q, aq);
#pragma unroll
        for (int n = 0; n < NG; ++n) { v[n] = (aq[n] + lk[n] * lr) * xv[n] * inv_as[n]; sum[n] += v[n]; }
        if (sep) stv<NG>(alphav_n + (size_t)(o0 + q - lo) * NG, v);
      }
      if (pl > 0.f) {      // peeled self-loop: alpha'[t,d] * prob * x[t, loop pdf] / asum, no gather
        float v[NG], al[NG], xlv[NG];
        float pis = pf_pi;
        if (pf) {
#pragma unroll
          for (int n = 0; n < NG; ++n) { al[n] = pf_al[n]; xlv[n] = pf_xl[n]; }
        } else {
          ldv<NG>(alpha_t + (size_t)d * NG, al);
          ldv<NG>(xl_t + (size_t)d * NG, xlv);
          pis = p.pi[d];
        }
#pragma unroll
        for (int n = 0; n < NG; ++n) { v[n] = (al[n] + lk[n] * pis) * pl * xlv[n] * inv_as[n]; sum[n] += v[n]; }
        stv<NG>(alphav_n + (size_t)(o0 + hi - lo) * NG, v);
      }
#pragma unroll
      for (int n = 0; n < NG; ++n) loc[n] += sum[n];
      stv<NG>(alpha_n + (size_t)d * NG, sum);
    }
  }
  DEN_T(0, 4);
  block_sum<NG, kDenWaves>(loc, red);
  if (tid == 0) stv<NG>(p.apart + ((frame + 1) * nc + chunk) * NG, loc);
  DEN_T(0, 5);
}

// Backward recursion of the state-x path.  It does NOT use the forward pass: instead of dividing by
// the forward scale asum[t] it normalises itself -- frame t stores btilde'[t,s] = sum_arcs prob *
// x[t,pdf] * betahat[t+1,d] unnormalised, with betahat[t+1,d] = btilde'[t+1,d]/c[t+1] + leaky and
// c[t+1] = sum_k pi[k] btilde'[t+1,k] (reduced from the partials like the forward sums).  The true
// beta is K[t] * betahat[t] with log K[t] = log K[t+1] + log c[t] - log asum[t], K[T] = sum(pi)/tot
// (den_scales).  Forward and backward chains therefore run concurrently on two streams.
template <int NG>
__device__ __forceinline__ void den_beta_frame_sx(const DenParams& p, int t, int chunk, float* acc, float* red, float* wcarry) {
  DEN_T0();
  const int g = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int ncb = p.bwd.n_chunks;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;
  const float* bx_n = p.beta + (frame + 1) * (size_t)p.V * (2 * NG);   // gathered by virtual destination state
  const int wb0 = p.bwd.wb_off[chunk], wb1 = p.bwd.wb_off[chunk + 1];
  const int nrows = p.bwd.nrows[chunk];
  float lB[NG], lU[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { lB[n] = 0.f; lU[n] = 0.f; }
  for (int i = tid; i < ncb; i += kDenThreads) {
    float v[NG], u[NG];
    ldv<NG>(p.bpart + (((frame + 1) * ncb + i) * 2) * NG, v);
    ldv<NG>(p.bpart + (((frame + 1) * ncb + i) * 2 + 1) * NG, u);
#pragma unroll
    for (int n = 0; n < NG; ++n) { lB[n] += v[n]; lU[n] += u[n]; }
  }
  int wb = wb0 + w;
  int2 rec[kK];      // {gathered state, arc probability}
  float b[kK][NG], xv[kK][NG];
  uint2 meta = make_uint2(0u, 0u);
  // row-epilogue inputs that do not depend on this frame
  const int row0 = p.bwd.row0[chunk];
  const bool atomic = p.bwd.atomic[chunk] != 0;
  const float* xl_t = p.xl + ((size_t)g * p.Tmax + t) * (size_t)p.S * NG;
  int pf_v0 = 0, pf_v1 = 0; float pf_pi = 0.f, pf_pl = 0.f, pf_bl[NG], pf_xl[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { pf_bl[n] = 0.f; pf_xl[n] = 0.f; }
  if (tid < nrows) {
    const int s = row0 + tid;
    pf_v0 = p.voff[s]; pf_v1 = p.voff[s + 1]; pf_pi = p.pi[s]; pf_pl = p.loop_prob[s];
    ldv<NG>(xl_t + (size_t)s * NG, pf_xl);
  }
  if (wb < wb1) {
    meta = p.bwd.meta[(size_t)wb * 64 + lane];
#pragma unroll
    for (int j = 0; j < kK; ++j) rec[j] = PK2_ARC_LD(&p.bwd.arcs2[((size_t)wb * kK + ((p.debug & 4) ? (j & 3) : j)) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < kK; ++j) {
      const size_t gi = (p.debug & 1) ? 0 : rec[j].x;
      ldv<NG>(bx_n + gi * (2 * NG), b[j]);
      ldv<NG>(bx_n + gi * (2 * NG) + NG, xv[j]);
    }
  }
  if (tid < nrows && pf_pl > 0.f) ldv<NG>(bx_n + (size_t)pf_v0 * (2 * NG), pf_bl);   // btilde'[t+1,s] for the peeled self-loop
  int crow[kDenWaves];
#pragma unroll
  for (int k = 0; k < kDenWaves; ++k) crow[k] = wb0 + k < wb1 ? p.bwd.wb_crow[wb0 + k] : -1;
  if (tid < kDenWaves * NG) wcarry[tid] = 0.f;
  DEN_T(1, 0);
  block_sum2<NG, kDenWaves>(lB, lU, red);
  DEN_T(1, 1);
  // lB = sum_k pi[k] btilde'[t+1,k], lU = sum_k btilde'[t+1,k].  The recursion normalises by
  // cu = lB + wu * lU, i.e. with weights pi[k] + kBetaFloor * sum(pi)/S: pi is the cheap guess of which states matter
  // (the forward pass is not available: the two chains run side by side), the uniform floor bounds every normalised
  // value by S / (kBetaFloor * sum(pi)) whatever pi[k] is -- normalising by the pi-weighted sum alone lets a state
  // with pi ~ 1e-38 overflow (measured on a test graph).  The leaky-HMM term of the normalised beta-hat' is
  // leaky * lB / cu.  beta-hat[T_n] = 1/sum(pi) + leaky starts a sequence.
  float cst[NG], inv_c[NG], lkr[NG];
  bool gat[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int T = p.lengths[g * NG + n];
    gat[n] = (t + 1) < T;
    const float cu = lB[n] + p.wu * lU[n];
    inv_c[n] = (gat[n] && cu > 0.f) ? 1.0f / cu : 0.f;
    lkr[n] = p.leaky * lB[n] * inv_c[n];
    cst[n] = ((t + 1) == T) ? (1.0f / p.pi_sum + p.leaky) : 0.f;
  }
  if (wb < wb1) {           // one wave block per wave
    int c = (int)meta.x, first_row = -1;
    const uint32_t mask = meta.y;
    float sum[NG], head[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) { sum[n] = 0.f; head[n] = 0.f; }
#pragma unroll
    for (int j = 0; j < kK; ++j) {
      const float prob = __int_as_float(rec[j].y);
#pragma unroll
      for (int n = 0; n < NG; ++n) sum[n] += prob * xv[j][n] * (gat[n] ? b[j][n] * inv_c[n] + lkr[n] : cst[n]);
      if ((mask >> j) & 1u) {
        if (first_row < 0) {
          first_row = c;
#pragma unroll
          for (int n = 0; n < NG; ++n) head[n] = sum[n];
        } else {
          stv<NG>(acc + (size_t)c * NG, sum);
        }
#pragma unroll
        for (int n = 0; n < NG; ++n) sum[n] = 0.f;
        ++c;
      }
    }
    seg_carry_store<NG>(sum, head, first_row, acc, wcarry + (size_t)(wb - wb0) * NG);
  }
  DEN_T(1, 2);
  __syncthreads();
  DEN_T(1, 3);
  // rows are source states; btilde'[t,s] goes into the record of every virtual state of s
  float* bx_t = p.beta + frame * (size_t)p.V * (2 * NG);
  float loc[NG], locu[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) { loc[n] = 0.f; locu[n] = 0.f; }
  for (int r = tid; r < nrows; r += kDenThreads) {
    float v[NG];
    const int s = row0 + r;
    const bool pf = r == tid;
    const float pis = pf ? pf_pi : p.pi[s];
    const int v0 = pf ? pf_v0 : p.voff[s], v1 = pf ? pf_v1 : p.voff[s + 1];
    const float pl = pf ? pf_pl : p.loop_prob[s];
    row_value<NG>(acc, wcarry, crow, r, v);
    if (pl > 0.f && p.bwd.atomic[chunk] != 1) {   // peeled self-loop (a split row adds it in its first piece only)
      float bl[NG], xlv[NG];
      if (pf) {
#pragma unroll
        for (int n = 0; n < NG; ++n) { bl[n] = pf_bl[n]; xlv[n] = pf_xl[n]; }
      } else {
        ldv<NG>(bx_n + (size_t)v0 * (2 * NG), bl);
        ldv<NG>(xl_t + (size_t)s * NG, xlv);
      }
#pragma unroll
      for (int n = 0; n < NG; ++n) v[n] += pl * xlv[n] * (gat[n] ? bl[n] * inv_c[n] + lkr[n] : cst[n]);
    }
#pragma unroll
    for (int n = 0; n < NG; ++n) { loc[n] += pis * v[n]; locu[n] += v[n]; }
    for (int q = v0; q < v1; ++q) {
      float* o = bx_t + (size_t)q * (2 * NG);
      if (atomic) {
#pragma unroll
        for (int n = 0; n < NG; ++n) atomicAdd(o + n, v[n]);
      } else {
        stv<NG>(o, v);
      }
    }
  }
  DEN_T(1, 4);
  block_sum2<NG, kDenWaves>(loc, locu, red);
  if (tid == 0) {
    stv<NG>(p.bpart + ((frame * ncb + chunk) * 2) * NG, loc);
    stv<NG>(p.bpart + ((frame * ncb + chunk) * 2 + 1) * NG, locu);
  }
  DEN_T(1, 5);
}

// One launch = forward frame `step` (workgroups [0, nc_fwd)) AND backward frame Tmax-1-step
// (workgroups [nc_fwd, nc_fwd + nc_bwd)): the two recursions are independent (see den_beta_frame_sx),
// so the serial chain is Tmax launches long instead of 2*Tmax, and both halves share the chip.
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_step_sx(const DenParams* __restrict__ pp,
                                                           const StepCounter* __restrict__ cnt, int local) {
  __shared__ __attribute__((aligned(16))) float acc[kMaxRows * NG];
  __shared__ float red[2 * kDenWaves * NG];
  __shared__ __attribute__((aligned(16))) float wcarry[kDenWaves * NG];
  const int step = cnt->base + local;
  if (step >= cnt->T) return;
  const DenParams& p = *pp;
  const int ncf = p.fwd.n_chunks;
  if ((int)blockIdx.x < ncf) den_fwd_frame_sx<NG>(p, step, blockIdx.x, acc, red, wcarry);
  else den_beta_frame_sx<NG>(p, p.Tmax - 1 - step, blockIdx.x - ncf, acc, red, wcarry);
}

// Kaldi's consistency check: sum_h alpha'[0,h] beta'[0,h] (should be 1 per sequence).
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_check(DenParams p, float* check) {
  __shared__ float red[kDenWaves * NG];
  const int g = blockIdx.x;
  const int ncb = p.bwd.n_chunks;
  float B0[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) B0[n] = 0.f;
  for (int i = threadIdx.x; i < ncb; i += kDenThreads) {
    float v[NG];
    ldv<NG>(p.bpart + (((size_t)g * (p.Tmax + 1)) * ncb + i) * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) B0[n] += v[n];
  }
  block_sum<NG, kDenWaves>(B0, red);
  if (threadIdx.x == 0) {
    // alpha'[0,h] = pi[h] * (1 + leaky * sum(pi)); the product with beta'[0] must be 1
    const float k = 1.0f + p.leaky * p.pi_sum;
#pragma unroll
    for (int n = 0; n < NG; ++n) check[g * NG + n] = k * B0[n];
  }
}

// out[seq][t][p] = scale * gamma[g][t][p][n] for t < T_n, 0 for the padding frames.
template <int NG>
__global__ void __launch_bounds__(256) den_gamma_out(const float* __restrict__ gamma,
                                                     const int32_t* __restrict__ lengths, int N,
                                                     int P, int Tmax, float scale, float* out,
                                                     int64_t seq_stride, int64_t frame_stride) {
  const int t = blockIdx.x, g = blockIdx.y;
  const float* src = gamma + ((size_t)g * Tmax + t) * (size_t)P * NG;
  for (int p = threadIdx.x; p < P; p += 256) {
    float v[NG];
    ldv<NG>(src + (size_t)p * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      const int seq = g * NG + n;
      if (seq < N) {
        const bool live = t < lengths[seq];
        out[(int64_t)seq * seq_stride + (int64_t)t * frame_stride + p] = live ? scale * v[n] : 0.f;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// host: per-call driver
// ----------------------------------------------------------------------------------------
static size_t den_lds_bytes(int P, int NG) {
  return ((size_t)P * NG + (size_t)2 * kMaxRows * NG + (size_t)2 * kDenWaves * NG) * sizeof(float);
}

// The state-x kernels serve every graph; the general (per-arc pdf, LDS-staged exp(logits)) family remains for graphs
// whose virtual-state count explodes (more than 4 distinct entering pdfs per state on average) and for A/B runs:
// PK2_DEN_MODE=general | sx overrides the choice.
bool den_use_sx(const pk2_den_graph* g) {
  const char* mode = getenv("PK2_DEN_MODE");
  if (mode && strcmp(mode, "general") == 0) return false;
  if (mode && strcmp(mode, "sx") == 0) return true;
  return (int64_t)g->V <= 4 * (int64_t)g->S;
}

int den_choose_ng(const pk2_den_graph* g) {
  const size_t limit = 160 * 1024;
  for (int ng : {4, 2, 1})
    if (den_lds_bytes(g->P, ng) <= limit) return ng;
  return 0;
}

static size_t den_workspace_as(const pk2_den_graph* g, int N, int Tmax, bool persist, DenGeom* geom, DenBuffers* buf,
                               void* base) {
  DenGeom ge;
  ge.NG = persist ? 1 : den_choose_ng(g);
  if (ge.NG == 0) ge.NG = 1;
  ge.N = N; ge.Tmax = Tmax;
  ge.G = (N + ge.NG - 1) / ge.NG;
  ge.persist = persist;
  Carver c(base);
  DenBuffers b;
  const size_t GN = (size_t)ge.G * ge.NG;
  const bool sx = den_use_sx(g);
  const size_t V = sx ? (size_t)g->V : (size_t)g->S, Vo = sx ? (size_t)g->Vo : (size_t)g->S;
  b.alpha = c.take<float>(GN * (Tmax + 1) * (size_t)g->S);
  b.alphav = (Vo == (size_t)g->S) ? b.alpha : c.take<float>(GN * (Tmax + 1) * Vo);
  b.beta = c.take<float>(GN * (Tmax + 1) * V * 2);   // {beta', xd} per virtual state on the state-x path
  b.xl = sx ? c.take<float>(GN * (size_t)Tmax * g->S) : nullptr;
  b.xs = c.take<float>(GN * (size_t)Tmax * g->P);
  b.gamma = c.take<float>(GN * (size_t)Tmax * g->P);
  // (the persistent kernel leaves kPR partial sums per frame; its fallback the chunk counts of the frame kernels)
  const size_t ncf = std::max<size_t>((sx ? g->h_fwdv : g->h_fwd).n_chunks, persist ? kPR : 0);
  const size_t ncb = std::max<size_t>((sx ? g->h_bwdv : g->h_bwd).n_chunks, persist ? kPR : 0);
  b.apart = c.take<float>(GN * (Tmax + 1) * ncf);
  b.bpart = c.take<float>(GN * (Tmax + 1) * (sx ? 2 * ncb : ncb));   // state-x: two sums per chunk
  b.asum = c.take<float>(GN * (Tmax + 1));
  b.inv_tot = c.take<float>(GN);
  b.den_lp = c.take<float>(GN);
  b.check = c.take<float>(GN);
  b.lengths = c.take<int32_t>(GN);
  b.csum = c.take<float>(GN * (Tmax + 1) * 2);   // {cu, sum pi*btilde' / cu}
  b.kscale = c.take<float>(GN * (Tmax + 1));
  b.xv = persist ? c.take<float>(GN * (size_t)Tmax * V + 64) : nullptr;   // (read in whole 64-entry rows: up to 63 floats past the end)
  if (geom) *geom = ge;
  if (buf) *buf = b;
  return c.bytes();
}

// The persistent kernel works on one-sequence groups (NG = 1).  A sizing call (no base) answers for both geometries:
// the choice can still change before the compute call (the first launch on a device is verified).
size_t den_workspace(const pk2_den_graph* g, int N, int Tmax, DenGeom* geom, DenBuffers* buf,
                     void* base) {
  const bool persist = den_persist_version(g, N) != 0;
  if (base) return den_workspace_as(g, N, Tmax, persist, geom, buf, base);
  const size_t a = den_workspace_as(g, N, Tmax, false, nullptr, nullptr, nullptr);
  const size_t b = den_workspace_as(g, N, Tmax, true, nullptr, nullptr, nullptr);
  den_workspace_as(g, N, Tmax, persist, geom, buf, nullptr);
  return std::max(a, b);
}

static StepGraphs g_den_graphs;
static std::map<std::pair<int, hipStream_t>, ParamSlot<DenParams>> g_den_slots;

// One internal side stream (+ fork/join events) per caller stream, created on first use.
static std::map<DevStream, SideStream> g_side_streams;
int get_side_stream(hipStream_t main, SideStream** out) {
  auto it = g_side_streams.find(dev_stream(main));
  if (it == g_side_streams.end()) {
    SideStream s;
    PK2_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    PK2_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
    PK2_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
    it = g_side_streams.emplace(dev_stream(main), s).first;
  }
  *out = &it->second;
  return PK2_OK;
}

template <int NG>
static int den_compute_t(pk2_den_graph* g, const float* logits, int64_t seq_stride,
                         int64_t frame_stride, const int32_t* lengths_host, const DenGeom& ge,
                         const DenBuffers& b, float leaky, hipStream_t stream, const NumDeferred* tail, const DenZeroRows* zr) {
  const int Tmax = ge.Tmax, G = ge.G;
  const size_t GN = (size_t)G * NG;
  // Round 6: on the bench path (NG = 1, state-x kernels, second persistent form with the x gather) the passes in front of the
  // recursions are ONE launch (den_prep1) and the ones behind them one more (den_tail1); PK2_DEN_MERGE=0 keeps the round-5
  // launches.  The caller's gradient rows (zr) are zeroed by that launch, or by one of their own on every other path.
  static const bool merge_env = [] { const char* e = getenv("PK2_DEN_MERGE"); return !(e && atoi(e) == 0); }();
  bool zero_pending = zr != nullptr && zr->grad != nullptr;
  // lengths travel as kernel arguments (no dependence on the lifetime of the caller's host array);
  // padding sequences get 0 frames
  for (size_t base = 0; base < GN; base += IntPack::kN) {
    IntPack pack;
    for (int i = 0; i < IntPack::kN; ++i) {
      size_t n = base + i;
      pack.v[i] = (n < (size_t)ge.N) ? lengths_host[n] : 0;
    }
    int cnt = (int)std::min<size_t>(IntPack::kN, GN - base);
    hipLaunchKernelGGL(store_ints, dim3(1), dim3(64), 0, stream, pack, cnt, b.lengths + base);
  }
  // alpha / beta / gamma rows of split ("atomic") chunks accumulate with atomics -> start at zero
  // (on the state-x path every alpha / beta' row and every occupancy is written by a plain store unless a row is
  // split over several chunks, and values of frames beyond a sequence's end are only ever selected away, never used
  // in arithmetic: the 0.9 GB of fills per call are skipped then)
  const bool sx = den_use_sx(g);
  const HostOrdering& hf = sx ? g->h_fwdv : g->h_fwd;
  const HostOrdering& hb = sx ? g->h_bwdv : g->h_bwd;
  auto any_atomic = [](const HostOrdering& h) { for (int32_t a : h.atomic) if (a) return true; return false; };
  const bool need_fill = !sx || any_atomic(hf) || any_atomic(hb);
  if (need_fill) {
    PK2_HIP(hipMemsetAsync(b.alpha, 0, GN * (Tmax + 1) * (size_t)g->S * sizeof(float), stream));
    if (b.alphav != b.alpha) PK2_HIP(hipMemsetAsync(b.alphav, 0, GN * (Tmax + 1) * (size_t)g->Vo * sizeof(float), stream));
    PK2_HIP(hipMemsetAsync(b.beta, 0, GN * (Tmax + 1) * (size_t)(sx ? g->V : g->S) * 2 * sizeof(float), stream));
    PK2_HIP(hipMemsetAsync(b.gamma, 0, GN * (size_t)Tmax * g->P * sizeof(float), stream));
  }
  const bool persist = ge.persist && NG == 1 && sx;
  const size_t bpart_floats = GN * (Tmax + 1) * std::max<size_t>(hb.n_chunks, persist ? kPR : 0) * (sx ? 2 : 1);
  bool use_prep = false;
  if (merge_env && persist && NG == 1 && den_persist_version(g, ge.N) == 2 && !need_fill) {
    const char* xg_e0 = getenv("PK2_DEN_XGATHER");
    use_prep = !(xg_e0 && atoi(xg_e0) == 0) && g->P <= 32767 && g->V >= g->P && g->p2_rowarrays == kP2RowArrays;
  }
  if (!use_prep) PK2_HIP(hipMemsetAsync(b.bpart, 0, bpart_floats * sizeof(float), stream));

  DenParams p;
  p.fwd = sx ? g->fwdv : g->fwd; p.bwd = sx ? g->bwdv : g->bwd; p.gam = g->gam;
  p.pi = g->d_pi;
  p.alpha = b.alpha; p.beta = b.beta; p.xs = b.xs; p.gamma = b.gamma;
  p.apart = b.apart; p.bpart = b.bpart; p.asum = b.asum; p.inv_tot = b.inv_tot;
  p.lengths = b.lengths;
  p.ps_off = g->d_po_off; p.ps_state = g->d_po_occ;
  p.voff = g->d_voff; p.ooff = g->d_ooff; p.opdf = g->d_opdf; p.ovirt = g->d_ovirt; p.loop_prob = g->d_loop_prob;
  p.alphav = b.alphav; p.xl = b.xl; p.V = sx ? g->V : g->S; p.Vo = sx ? g->Vo : g->S;
  p.brec = 2;
  p.S = g->S; p.P = g->P; p.Tmax = Tmax;
  p.leaky = leaky; p.pi_sum = (float)g->pi_sum; p.wu = (float)(kBetaFloor * g->pi_sum / g->S);
  p.debug = getenv("PK2_DEN_DEBUG") ? atoi(getenv("PK2_DEN_DEBUG")) : 0;
  p.beta_seed = 1.0f;
  if (const char* env = getenv("PK2_DEN_DEBUG_BETA_SEED")) { const float v = (float)atof(env); if (v > 0.f) p.beta_seed = v; }
  if (persist) { p.fwd.n_chunks = kPR; p.bwd.n_chunks = kPR; }     // partial sums per frame: one per workgroup of a team

  const size_t lds = den_lds_bytes(g->P, NG);
  struct attr_set_t { bool f[8]; }; static PerDevice<attr_set_t> attr_set_pd(attr_set_t{}); bool (&attr_set)[8] = attr_set_pd.ref().f;
  if (!attr_set[NG]) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_fwd_step<NG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_bwd_step<NG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[NG] = true;
  }
  ParamSlot<DenParams>* slot;
  int rc = get_param_slot(g_den_slots, NG, stream, &slot);
  if (rc) return rc;
  // (the frame kernels read their parameters from this block; the persistent kernels carry their own)
  if (!(merge_env && persist)) hipLaunchKernelGGL(param_block_store<DenParams>, dim3(1), dim3(1), 0, stream, p, slot->params);
  const DenParams* pb = slot->params;
  const StepCounter* cnt = slot->counter;

  // Two kernel families: the "state-x" path (exp(logit) as a per-virtual-state factor; den_use_sx), else LDS-staged
  // exp(logits) + arc-based occupancies.
  char key[96];
  const int init_bx = std::min(256, (g->S + 255) / 256);
  if constexpr (NG == 1) {
    if (use_prep) {
      DenPrepJobs j{Tmax * G, init_bx * G, init_bx, 64, zero_pending ? Tmax * zr->N : 0, bpart_floats, logits, seq_stride, frame_stride,
                    b.xv, zero_pending ? zr->grad : nullptr, zero_pending ? zr->gss : 0, zero_pending ? zr->gfs : 0, zero_pending ? zr->N : 0};
      hipLaunchKernelGGL(den_prep1, dim3(j.exp_blocks + j.init_blocks + j.bpart_blocks + j.zero_blocks), dim3(256), 0, stream, p, j);
      zero_pending = false;
    } else if (zero_pending) {        // the gradient rows alone (same kernel, one job)
      DenPrepJobs j{0, 0, 1, 0, Tmax * zr->N, 0, logits, seq_stride, frame_stride, nullptr, zr->grad, zr->gss, zr->gfs, zr->N};
      hipLaunchKernelGGL(den_prep1, dim3(j.zero_blocks), dim3(256), 0, stream, p, j);
      zero_pending = false;
    }
  }
  if (zero_pending) {
    DenParams pz = p;
    DenPrepJobs j{0, 0, 1, 0, Tmax * zr->N, 0, logits, seq_stride, frame_stride, nullptr, zr->grad, zr->gss, zr->gfs, zr->N};
    hipLaunchKernelGGL(den_prep1, dim3(j.zero_blocks), dim3(256), 0, stream, pz, j);
    zero_pending = false;
  }
  if (!use_prep) hipLaunchKernelGGL(den_init<NG>, dim3(init_bx, G), dim3(256), 0, stream, p);
  if (sx) {
    // a frame's entries are cut into slices on grid.z so that short minibatches still fill the chip
    auto slices = [&](int n) { return std::max(1, std::min((n + 1023) / 1024, (4096 + Tmax * G - 1) / (Tmax * G))); };
    const size_t exp_lds = (size_t)g->P * NG * sizeof(float);
    const int form = persist ? den_persist_version(g, ge.N) : 0;
    // (the second persistent kernel gathers x by pdf from plain exp(logits) rows, which fit the xv buffer when V >= P;
    // PK2_DEN_XGATHER=0 keeps the expanded copies)
    const char* xg_e = getenv("PK2_DEN_XGATHER");      // (read per call: the tests switch it inside one process)
    const bool xg_env = !(xg_e && atoi(xg_e) == 0);
    const bool xgather = xg_env && form == 2 && NG == 1 && g->P <= 32767 && g->V >= g->P && g->p2_rowarrays == kP2RowArrays;
    if (form == 2 && NG == 1) p.brec = 1;       // (dense btilde' records: den_kernels.h; the frame kernels' copy of p keeps 2)
    if (xgather && use_prep) {
      // (den_prep1 has written the rows)
    } else if (xgather) {
      hipLaunchKernelGGL(den_exp_rows, dim3(Tmax, G), dim3(256), 0, stream, logits, seq_stride, frame_stride, b.lengths, b.xv,
                         g->P, Tmax);
    } else if (persist || (exp_lds <= kGammaMaxLds && !getenv("PK2_DEN_EXP_GATHER"))) {
      struct attr_e_t { bool f[8]; }; static PerDevice<attr_e_t> attr_e_pd(attr_e_t{}); bool (&attr_e)[8] = attr_e_pd.ref().f;
      if (!attr_e[NG]) {
        PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_exp_states_lds<NG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_e[NG] = true;
      }
      const int z = std::max(1, std::min((g->V + kExpThreads - 1) / kExpThreads, (2048 + Tmax * G - 1) / (Tmax * G)));
      hipLaunchKernelGGL(den_exp_states_lds<NG>, dim3(Tmax, G, z), dim3(kExpThreads), exp_lds, stream, logits, seq_stride,
                         frame_stride, b.lengths, g->d_vpdf, b.beta, g->V, g->d_loop_pdf, b.xl, g->S, g->P, Tmax,
                         persist ? b.xv : nullptr);
    } else {
      const int zv = slices(g->V), zl = slices(g->S);
      hipLaunchKernelGGL(den_exp_states<NG>, dim3(Tmax, G, zv + zl), dim3(256), 0, stream, logits, seq_stride, frame_stride,
                         b.lengths, g->d_vpdf, b.beta, g->V, g->d_loop_pdf, b.xl, g->S, Tmax, zv);
    }
    PK2_LAUNCH_CHECK();
    bool ran = false, num_rode = false;
    int persist_form = 0;
    if (persist) {      // both recursions of every sequence in one launch (chain_den_persist.hip / chain_den_persist2.hip)
      rc = form == 2 ? den_persist2_launch(g, p, b.xv, lengths_host, ge.N, stream, &ran, tail, &num_rode, xgather)
                     : den_persist_launch(g, p, b.xv, lengths_host, ge.N, stream, &ran);
      if (rc) return rc;
      persist_form = ran ? form : 0;
      if (!ran) {       // the device failed the first-use verification: the frame kernels, with their own chunk counts
        DenGeom ge2 = ge;
        ge2.persist = false;
        DenZeroRows none{};       // (the gradient rows have been zeroed above)
        return den_compute_t<NG>(g, logits, seq_stride, frame_stride, lengths_host, ge2, b, leaky, stream, tail, zr ? &none : nullptr);
      }
    } else {
      // the backward chain needs only exp(logits): forward frame `step` and backward frame Tmax-1-step
      // share one launch
      const dim3 gridS(g->fwdv.n_chunks + g->bwdv.n_chunks, G);
      snprintf(key, sizeof(key), "den_sx_%d_%u_%d_%p", NG, gridS.x, G, (void*)stream);
      rc = g_den_graphs.run(key, Tmax, slot->counter, stream, [&](hipStream_t s, int j) {
        hipLaunchKernelGGL(den_step_sx<NG>, gridS, dim3(kDenThreads), 0, s, pb, cnt, j);
      });
      if (rc) return rc;
    }
#ifdef PK2_DEN_PROFILE
    hipLaunchKernelGGL(den_prof_print, dim3(1), dim3(1), 0, stream, Tmax);
#endif
    bool tail_merged = false;
    if constexpr (NG == 1) {
      if (merge_env && (!ran || persist_form == 2) && (size_t)(Tmax + 1) * 2 * sizeof(float) <= 60 * 1024) {
        DenTailCheck ck{};
        if (ran && !den_persist2_tail_check(stream, &ck)) ck = DenTailCheck{};
        // (test hook, read per call: PK2_DEN_TEST_FAIL=1 makes the folded check behave as if the persistent launch had given up)
        if (ck.ctl) { const char* tf_e = getenv("PK2_DEN_TEST_FAIL"); if (tf_e && atoi(tf_e) == 1) ck.ntasks = -1; }
        hipLaunchKernelGGL(den_tail1, dim3(G), dim3(256), (size_t)(Tmax + 1) * 2 * sizeof(float), stream, p, b.csum, b.den_lp, b.kscale,
                           b.check, ck);
        tail_merged = true;
      }
    }
    if (!tail_merged) {
      hipLaunchKernelGGL(den_csum<NG>, dim3(Tmax + 1, G), dim3(256), 0, stream, p, b.csum);
      hipLaunchKernelGGL(den_finalize<NG>, dim3(G, NG), dim3(256), 0, stream, p, b.den_lp);
      if (ran && persist_form == 2) den_persist2_check_launch(b.den_lp, ge.N, stream);
      else if (ran) den_persist_check_launch(b.den_lp, ge.N, stream);
      hipLaunchKernelGGL(den_scales<NG>, dim3(G, NG), dim3(256), 0, stream, p, b.csum, b.kscale, b.check);
    }
    const size_t row_lds = (size_t)g->P * NG * sizeof(float);
    const bool lds_row = row_lds <= kGammaMaxLds && !getenv("PK2_DEN_GAMMA_GATHER");
    struct attr_n_t { bool f[8]; }; static PerDevice<attr_n_t> attr_n_pd(attr_n_t{}); bool (&attr_n)[8] = attr_n_pd.ref().f;
    if (!attr_n[NG]) {
      PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_gamma_states_num<NG, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_gamma_states_num<NG, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_gamma_states_lds<NG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_n[NG] = true;
    }
    if (tail && tail->valid && !num_rode) {
      if (lds_row && std::max(row_lds, (size_t)tail->lds) <= kGammaMaxLds)
        hipLaunchKernelGGL((den_gamma_states_num<NG, true>), dim3(Tmax + tail->N, G), dim3(kGammaThreads),
                           std::max(row_lds, (size_t)tail->lds), stream, p, b.csum, b.kscale, tail->p, tail->N, tail->stage ? 1 : 0);
      else
        hipLaunchKernelGGL((den_gamma_states_num<NG, false>), dim3(Tmax + tail->N, G), dim3(256), tail->lds, stream, p, b.csum,
                           b.kscale, tail->p, tail->N, tail->stage ? 1 : 0);
    } else if (lds_row) {
      hipLaunchKernelGGL(den_gamma_states_lds<NG>, dim3(Tmax, G), dim3(kGammaThreads), row_lds, stream, p, b.csum, b.kscale);
    } else {
      hipLaunchKernelGGL(den_gamma_states<NG>, dim3(Tmax, G), dim3(256), 0, stream, p, b.csum, b.kscale);
    }
    PK2_LAUNCH_CHECK();
    return PK2_OK;
  } else {
    hipLaunchKernelGGL(den_exp_transpose<NG>, dim3(Tmax, G), dim3(256), 0, stream, logits, seq_stride,
                       frame_stride, b.lengths, b.xs, g->P, Tmax);
    PK2_LAUNCH_CHECK();
    const dim3 gridF(g->fwd.n_chunks, G);
    snprintf(key, sizeof(key), "den_fwd_%d_%u_%d_%zu_%p", NG, gridF.x, G, lds, (void*)stream);
    rc = g_den_graphs.run(key, Tmax, slot->counter, stream, [&](hipStream_t s, int j) {
      hipLaunchKernelGGL(den_fwd_step<NG>, gridF, dim3(kDenThreads), lds, s, pb, cnt, j);
    });
    if (rc) return rc;
    hipLaunchKernelGGL(den_finalize<NG>, dim3(G, NG), dim3(256), 0, stream, p, b.den_lp);
    const dim3 gridB(std::max(g->bwd.n_chunks, g->gam.n_chunks), G);
    snprintf(key, sizeof(key), "den_bwd_%d_%u_%d_%zu_%p", NG, gridB.x, G, lds, (void*)stream);
    rc = g_den_graphs.run(key, Tmax, slot->counter, stream, [&](hipStream_t s, int j) {
      hipLaunchKernelGGL(den_bwd_step<NG>, gridB, dim3(kDenBwdThreads), lds, s, pb, cnt, j);
    });
    if (rc) return rc;
  }
  hipLaunchKernelGGL(den_check<NG>, dim3(G), dim3(kDenThreads), 0, stream, p, b.check);
  PK2_LAUNCH_CHECK();
  if (tail && tail->valid) return num_launch_deferred(*tail, stream);
  return PK2_OK;
}

int den_compute(pk2_den_graph* g, const float* logits, int64_t seq_stride, int64_t frame_stride,
                const int32_t* lengths_host, const DenGeom& ge, const DenBuffers& b, float leaky,
                hipStream_t stream, const NumDeferred* tail, const DenZeroRows* zr) {
  int rc = den_upload(g);
  if (rc) return rc;
  if (den_choose_ng(g) == 0) {
    set_error("den graph: num_pdfs %d too large for the LDS-staged kernel", g->P);
    return PK2_ERR_LIMIT;
  }
  switch (ge.NG) {
    case 4: return den_compute_t<4>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream, tail, zr);
    case 2: return den_compute_t<2>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream, tail, zr);
    default: return den_compute_t<1>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream, tail, zr);
  }
}

template <int NG>
static void launch_gamma_out(const DenGeom& ge, const DenBuffers& b, int P, float scale, float* out,
                             int64_t ss, int64_t fs, hipStream_t stream) {
  hipLaunchKernelGGL(den_gamma_out<NG>, dim3(ge.Tmax, ge.G), dim3(256), 0, stream, b.gamma, b.lengths,
                     ge.N, P, ge.Tmax, scale, out, ss, fs);
}

void den_gamma_out_launch(const DenGeom& ge, const DenBuffers& b, int P, float scale, float* out,
                          int64_t ss, int64_t fs, hipStream_t stream) {
  switch (ge.NG) {
    case 4: launch_gamma_out<4>(ge, b, P, scale, out, ss, fs, stream); break;
    case 2: launch_gamma_out<2>(ge, b, P, scale, out, ss, fs, stream); break;
    default: launch_gamma_out<1>(ge, b, P, scale, out, ss, fs, stream); break;
  }
}

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_chain_den_fwd_bwd(const pk2_den_graph* gc, const float* logits, int64_t seq_stride,
                                     int64_t frame_stride, const int32_t* lengths, int32_t num_seqs,
                                     float leaky, float* den_logprob, float* gamma,
                                     int64_t gamma_seq_stride, int64_t gamma_frame_stride,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  PK2_REQUIRE(gc && logits && lengths && num_seqs > 0 && workspace, "den_fwd_bwd: bad args");
  pk2_den_graph* g = const_cast<pk2_den_graph*>(gc);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int Tmax = 0;
  for (int n = 0; n < num_seqs; ++n) {
    PK2_REQUIRE(lengths[n] > 0, "den_fwd_bwd: sequence %d has no frames", n);
    Tmax = std::max(Tmax, lengths[n]);
  }
  DenGeom ge; DenBuffers b;
  size_t need = den_workspace(g, num_seqs, Tmax, &ge, &b, workspace);
  PK2_REQUIRE(workspace_bytes >= need, "den_fwd_bwd: workspace %zu < %zu", workspace_bytes, need);
  int rc = den_compute(g, logits, seq_stride, frame_stride, lengths, ge, b, leaky, stream);
  if (rc) return rc;
  if (gamma) den_gamma_out_launch(ge, b, g->P, 1.0f, gamma, gamma_seq_stride, gamma_frame_stride, stream);
  if (den_logprob)
    PK2_HIP(hipMemcpyAsync(den_logprob, b.den_lp, num_seqs * sizeof(float), hipMemcpyDeviceToDevice, stream));
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Which kernels a denominator call on `num_seqs` sequences takes: 0 = general (per-arc pdf), 1 = state-x, a launch per
// frame, 2 = state-x, persistent recursion kernel (chain_den_persist.hip).  Reporting hook (bench.py).
extern "C" int32_t pk2_den_graph_path(const pk2_den_graph* g, int32_t num_seqs) {
  if (!g) return -1;
  if (!pk2::den_use_sx(g)) return 0;
  return pk2::den_persist_version(g, num_seqs) != 0 ? 2 : 1;
}

// Which form of the persistent recursion kernel such a call takes: 0 = none (frame kernels), 1 = den_persist_kernel
// (everything resident), 2 = den_persist2_kernel (chunked table, two resident passes, streamed overflow).
extern "C" int32_t pk2_den_graph_persist_form(const pk2_den_graph* g, int32_t num_seqs) {
  if (!g) return -1;
  if (!pk2::den_use_sx(g)) return 0;
  return pk2::den_persist_version(g, num_seqs);
}
