// Denominator graph + denominator forward-backward of the LF-MMI objective on gfx950.
//
// Replaces kaldi.chain.DenominatorGraph (reference bin/train_chain.py:167,202) and the
// DenominatorComputation half of kaldi.chain.compute_chain_objf_and_deriv (reference
// ops/ops.py:265); arithmetic per SURVEY.md Appendix A.2 (probability space, per-frame
// 1/sum(alpha) rescaling, leaky-HMM).
//
// MI355X design (not Kaldi's one-thread-per-state CUDA kernels):
//  * up to 4 sequences are interleaved in every per-state / per-pdf vector, so one 16-byte
//    gather serves 4 sequences and the arc lists are streamed once per frame for all of them;
//  * arcs are pre-sorted three ways on the host (by destination for alpha, by source for beta,
//    by pdf for the occupancies), so every reduction is a segmented gather-reduce into an
//    LDS-private accumulator -- no global atomics on the common path;
//  * exp(logits) of the frame (P x 4 floats, 97 KB) is staged in LDS once per workgroup;
//  * arcs are stored lane-interleaved: each 64-lane wavefront loads 1 KiB per instruction;
//  * the global sums of a frame (sum alpha, sum pi*beta) are never reduced in a separate
//    launch: each workgroup writes a partial and the NEXT frame's workgroups all re-reduce the
//    partials in a fixed order (bitwise identical everywhere), so one launch per frame suffices
//    and the leaky-HMM term is applied on the fly through a precomputed pi[src]*prob per arc.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "chain_internal.h"

namespace pk2 {

// ----------------------------------------------------------------------------------------
// host: graph construction
// ----------------------------------------------------------------------------------------
static void build_ordering(int64_t A, int num_rows, const int32_t* key, const int32_t* a,
                           const int32_t* b, const float* prob, const float* piprob,
                           HostOrdering* out) {
  // counting sort by key (stable)
  std::vector<int64_t> ptr(num_rows + 1, 0);
  for (int64_t i = 0; i < A; ++i) ptr[key[i] + 1]++;
  for (int r = 0; r < num_rows; ++r) ptr[r + 1] += ptr[r];
  std::vector<int64_t> perm(A);
  {
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < A; ++i) perm[cur[key[i]]++] = i;
  }
  out->arcs.clear(); out->meta.clear(); out->wb_off.assign(1, 0);
  out->row0.clear(); out->nrows.clear(); out->atomic.clear();

  struct Piece { int row; int64_t lo, hi; };  // arcs [lo,hi) of sorted list belong to `row`
  auto emit_chunk = [&](const std::vector<Piece>& pieces, int row0, int nrows, int atomic) {
    // sorted arcs of this chunk, each tagged with its chunk-local row
    std::vector<int64_t> idx; std::vector<int> lrow;
    for (const Piece& p : pieces) {
      for (int64_t k = p.lo; k < p.hi; ++k) { idx.push_back(perm[k]); lrow.push_back(p.row - row0); }
      // a row without arcs gets one null arc so that chunk-local rows stay consecutive
      if (p.lo == p.hi) { idx.push_back(-1); lrow.push_back(p.row - row0); }
    }
    const int per_wb = 64 * kK;
    int64_t n = (int64_t)idx.size();
    int64_t padded = std::max<int64_t>(per_wb, (n + per_wb - 1) / per_wb * per_wb);
    int last_row = nrows - 1;
    int nwb = (int)(padded / per_wb);
    size_t base_arc = out->arcs.size(), base_meta = out->meta.size();
    out->arcs.resize(base_arc + padded);
    out->meta.resize(base_meta + (size_t)nwb * 64);
    for (int wb = 0; wb < nwb; ++wb) {
      for (int lane = 0; lane < 64; ++lane) {
        uint32_t mask = 0; int c0 = 0;
        for (int j = 0; j < kK; ++j) {
          int64_t s = (int64_t)wb * per_wb + (int64_t)lane * kK + j;
          int4 rec; int row_here, row_next;
          if (s < n && idx[s] < 0) {
            rec.x = 0; rec.y = 0; rec.z = 0; rec.w = 0;
            row_here = lrow[s];
          } else if (s < n) {
            int64_t i = idx[s];
            rec.x = a[i]; rec.y = b[i];
            rec.z = __builtin_bit_cast(int, prob[i]);
            rec.w = __builtin_bit_cast(int, piprob[i]);
            row_here = lrow[s];
          } else {
            rec.x = 0; rec.y = 0; rec.z = 0; rec.w = 0;  // null arc: contributes exactly 0
            row_here = last_row;
          }
          row_next = (s + 1 < n) ? lrow[s + 1] : last_row;
          if (j == 0) c0 = row_here;
          if (j == kK - 1 || row_next != row_here) mask |= (1u << j);
          out->arcs[base_arc + ((size_t)wb * kK + j) * 64 + lane] = rec;
        }
        out->meta[base_meta + (size_t)wb * 64 + lane] = (uint32_t)c0 | (mask << 16);
      }
    }
    out->wb_off.push_back(out->wb_off.back() + nwb);
    out->row0.push_back(row0);
    out->nrows.push_back(nrows);
    out->atomic.push_back(atomic);
  };

  std::vector<Piece> cur; int cur_row0 = 0; int64_t cur_arcs = 0;
  auto flush = [&](int next_row) {
    if (!cur.empty()) emit_chunk(cur, cur_row0, (int)cur.size(), 0);
    cur.clear(); cur_arcs = 0; cur_row0 = next_row;
  };
  for (int r = 0; r < num_rows; ++r) {
    int64_t lo = ptr[r], hi = ptr[r + 1], len = hi - lo;
    if (len > kChunkArcs) {
      flush(r);
      for (int64_t s = lo; s < hi; s += kChunkArcs) {
        std::vector<Piece> one{{r, s, std::min(hi, s + kChunkArcs)}};
        emit_chunk(one, r, 1, 1);
      }
      cur_row0 = r + 1;
      continue;
    }
    const int64_t len_eff = std::max<int64_t>(len, 1);  // empty rows carry one null arc
    if (cur_arcs + len_eff > kChunkArcs || (int)cur.size() + 1 > kMaxRows) flush(r);
    cur.push_back({r, lo, hi});
    cur_arcs += len_eff;
  }
  flush(num_rows);
  out->n_chunks = (int)out->row0.size();
}

static int build_graph(int32_t S, int32_t P, int64_t A, const int32_t* src, const int32_t* dst,
                       const int32_t* pdf, const float* prob, int32_t start, pk2_den_graph** out) {
  PK2_REQUIRE(S > 0 && P > 0 && A > 0 && start >= 0 && start < S, "den graph: bad sizes");
  PK2_REQUIRE(P <= 65536, "den graph: num_pdfs %d > 65536 unsupported", P);
  for (int64_t i = 0; i < A; ++i) {
    PK2_REQUIRE(src[i] >= 0 && src[i] < S && dst[i] >= 0 && dst[i] < S && pdf[i] >= 0 && pdf[i] < P,
                "den graph: arc %lld out of range", (long long)i);
  }
  auto* g = new pk2_den_graph();
  g->S = S; g->P = P; g->A = A; g->start = start;
  // initial_probs: Kaldi DenominatorGraph::SetInitialProbs: 100 iterations
  // of the normalised forward recursion from the start state, averaged (double precision).
  {
    std::vector<double> cur(S, 0.0), nxt(S), avg(S, 0.0);
    cur[start] = 1.0;
    const int iters = 100;
    for (int it = 0; it < iters; ++it) {
      for (int s = 0; s < S; ++s) avg[s] += cur[s] / iters;
      std::fill(nxt.begin(), nxt.end(), 0.0);
      for (int64_t i = 0; i < A; ++i) nxt[dst[i]] += cur[src[i]] * (double)prob[i];
      double tot = 0.0;
      for (int s = 0; s < S; ++s) tot += nxt[s];
      for (int s = 0; s < S; ++s) cur[s] = nxt[s] / tot;
    }
    g->pi.resize(S);
    double ps = 0.0;
    for (int s = 0; s < S; ++s) { g->pi[s] = (float)avg[s]; ps += (double)g->pi[s]; }
    g->pi_sum = ps;
  }
  std::vector<float> piprob(A);
  for (int64_t i = 0; i < A; ++i) piprob[i] = g->pi[src[i]] * prob[i];
  build_ordering(A, S, dst, src, pdf, prob, piprob.data(), &g->h_fwd);   // alpha: rows = dst
  build_ordering(A, S, src, dst, pdf, prob, piprob.data(), &g->h_bwd);   // beta : rows = src
  build_ordering(A, P, pdf, src, dst, prob, piprob.data(), &g->h_gam);   // gamma: rows = pdf
  *out = g;
  return PK2_OK;
}

template <typename T>
static int upload_vec(pk2_den_graph* g, const std::vector<T>& v, const T** dptr) {
  void* d = nullptr;
  PK2_HIP(hipMalloc(&d, std::max<size_t>(16, v.size() * sizeof(T))));
  g->allocs.push_back(d);
  if (!v.empty()) PK2_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dptr = static_cast<const T*>(d);
  return PK2_OK;
}

static int upload_ordering(pk2_den_graph* g, const HostOrdering& h, DevOrdering* d) {
  int rc;
  if ((rc = upload_vec(g, h.arcs, &d->arcs))) return rc;
  if ((rc = upload_vec(g, h.meta, &d->meta))) return rc;
  if ((rc = upload_vec(g, h.wb_off, &d->wb_off))) return rc;
  if ((rc = upload_vec(g, h.row0, &d->row0))) return rc;
  if ((rc = upload_vec(g, h.nrows, &d->nrows))) return rc;
  if ((rc = upload_vec(g, h.atomic, &d->atomic))) return rc;
  d->n_chunks = h.n_chunks;
  return PK2_OK;
}

int den_upload(pk2_den_graph* g) {
  if (g->uploaded) return PK2_OK;
  int rc;
  PK2_HIP(hipGetDevice(&g->device));
  if ((rc = upload_ordering(g, g->h_fwd, &g->fwd))) return rc;
  if ((rc = upload_ordering(g, g->h_bwd, &g->bwd))) return rc;
  if ((rc = upload_ordering(g, g->h_gam, &g->gam))) return rc;
  const float* dpi = nullptr;
  if ((rc = upload_vec(g, g->pi, &dpi))) return rc;
  g->d_pi = const_cast<float*>(dpi);
  g->uploaded = true;
  return PK2_OK;
}

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

// Sum over the workgroup; every thread receives the same bits.  `red` holds kDenWaves*NG floats.
template <int NG>
__device__ __forceinline__ void block_sum(float (&v)[NG], float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NG; ++n) v[n] = wave_sum(v[n]);
  __syncthreads();  // protects `red` against a previous use
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NG; ++n) red[w * NG + n] = v[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    float s = 0.f;
    for (int k = 0; k < kDenWaves; ++k) s += red[k * NG + n];
    v[n] = s;
  }
}

// Fixed-order reduction of `count` partial vectors (stride NG floats).
template <int NG>
__device__ __forceinline__ void reduce_partials(const float* part, int count, float (&out)[NG],
                                                float* red) {
  float acc[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) acc[n] = 0.f;
  for (int i = threadIdx.x; i < count; i += kDenThreads) {
    float v[NG];
    ldv<NG>(part + (size_t)i * NG, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) acc[n] += v[n];
  }
  block_sum<NG>(acc, red);
#pragma unroll
  for (int n = 0; n < NG; ++n) out[n] = acc[n];
}

struct IntPack { static constexpr int kN = 64; int32_t v[kN]; };
__global__ void store_ints(IntPack pack, int count, int32_t* out) {
  if ((int)threadIdx.x < count) out[threadIdx.x] = pack.v[threadIdx.x];
}

struct DenParams {
  DevOrdering fwd, bwd, gam;
  const float* pi;
  float* alpha; float* beta; float* xs; float* gamma;
  float* apart; float* bpart; float* asum; float* inv_tot;
  const int32_t* lengths;
  int S, P, Tmax;
  float leaky, pi_sum;
};

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
    for (int n = 0; n < NG; ++n)
      v[n] = live[n] ? expf(fminf(fmaxf(rows[n][p], -30.f), 30.f)) : 1.0f;
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

// One frame of the alpha recursion: alpha[t+1] from alpha[t].
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_fwd_step(DenParams p, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs_l = smem;                              // P*NG
  float* acc = xs_l + (size_t)p.P * NG;            // kMaxRows*NG
  float* red = acc + (size_t)kMaxRows * NG;        // kDenWaves*NG
  const int g = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int nc = p.fwd.n_chunks;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;

  float as[NG], lk[NG], inv_as[NG];
  reduce_partials<NG>(p.apart + frame * nc * NG, nc, as, red);
  if (chunk == 0 && tid == 0) stv<NG>(p.asum + frame * NG, as);
#pragma unroll
  for (int n = 0; n < NG; ++n) { lk[n] = p.leaky * as[n]; inv_as[n] = 1.0f / as[n]; }

  // stage exp(logits) of frame t
  {
    const float4* src = reinterpret_cast<const float4*>(p.xs + ((size_t)g * p.Tmax + t) * (size_t)p.P * NG);
    float4* dst = reinterpret_cast<float4*>(xs_l);
    const int n4 = p.P * NG / 4;
    for (int i = tid; i < n4; i += kDenThreads) dst[i] = src[i];
    for (int i = n4 * 4 + tid; i < p.P * NG; i += kDenThreads) xs_l[i] = p.xs[((size_t)g * p.Tmax + t) * (size_t)p.P * NG + i];
  }
  const int nrows = p.fwd.nrows[chunk];
  for (int i = tid; i < nrows * NG; i += kDenThreads) acc[i] = 0.f;
  __syncthreads();

  const float* alpha_t = p.alpha + frame * (size_t)p.S * NG;
  const int wb0 = p.fwd.wb_off[chunk], wb1 = p.fwd.wb_off[chunk + 1];
  for (int wb = wb0 + w; wb < wb1; wb += kDenWaves) {
    const uint32_t meta = p.fwd.meta[(size_t)wb * 64 + lane];
    int c = meta & 0xffffu;
    const uint32_t mask = meta >> 16;
    int4 rec[kK];
#pragma unroll
    for (int j = 0; j < kK; ++j) rec[j] = p.fwd.arcs[((size_t)wb * kK + j) * 64 + lane];
    float a[kK][NG];
#pragma unroll
    for (int j = 0; j < kK; ++j) ldv<NG>(alpha_t + (size_t)rec[j].x * NG, a[j]);
    float sum[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) sum[n] = 0.f;
#pragma unroll
    for (int j = 0; j < kK; ++j) {
      const float prob = __int_as_float(rec[j].z), pp = __int_as_float(rec[j].w);
      float xv[NG];
      ldv<NG>(xs_l + (size_t)rec[j].y * NG, xv);
#pragma unroll
      for (int n = 0; n < NG; ++n) sum[n] += (a[j][n] * prob + lk[n] * pp) * xv[n];
      if ((mask >> j) & 1u) {
#pragma unroll
        for (int n = 0; n < NG; ++n) { atomicAdd(&acc[c * NG + n], sum[n]); sum[n] = 0.f; }
        ++c;
      }
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
  block_sum<NG>(loc, red);
  if (tid == 0) stv<NG>(p.apart + ((frame + 1) * nc + chunk) * NG, loc);
}

// log p_den, 1/tot, and the beta' initialisation constants.
template <int NG>
__global__ void __launch_bounds__(256) den_finalize(DenParams p, float* den_lp) {
  __shared__ double redd[4];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int seq = g * NG + n;
  const int T = p.lengths[seq];
  const int nc = p.fwd.n_chunks;
  const size_t f0 = (size_t)g * (p.Tmax + 1);
  double acc = 0.0;
  for (int t = tid; t < T; t += 256) acc += log((double)p.asum[(f0 + t) * NG + n]);
  // asum[T] from the partials written by the last forward step that touched frame T
  double at = 0.0;
  for (int c = tid; c < nc; c += 256) at += (double)p.apart[((f0 + T) * nc + c) * NG + n];
  acc = wave_sum_d(acc); at = wave_sum_d(at);
  __shared__ double reda[4];
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

// One frame of the beta recursion plus the occupancies of frame t.
// blockIdx.x < bwd.n_chunks: beta'[t][src-rows]; otherwise gamma[t][pdf-rows].
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_bwd_step(DenParams p, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs_l = smem;
  float* acc = xs_l + (size_t)p.P * NG;
  float* red = acc + (size_t)kMaxRows * NG;
  const int g = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int ncb = p.bwd.n_chunks;
  const bool beta_role = (int)blockIdx.x < ncb;
  const int chunk = beta_role ? blockIdx.x : blockIdx.x - ncb;
  const DevOrdering& ord = beta_role ? p.bwd : p.gam;
  const size_t frame = (size_t)g * (p.Tmax + 1) + t;

  // beta[t+1][d] = beta'[t+1][d] + leaky * sum_k pi[k] beta'[t+1][k]   (t+1 <  T_n)
  //             = (1 + leaky*sum(pi)) / tot_n                            (t+1 == T_n)
  //             = 0                                                      (t+1 >  T_n)
  float lB[NG], cst[NG], inv_as[NG], lk[NG];
  bool gat[NG];
  reduce_partials<NG>(p.bpart + (frame + 1) * ncb * NG, ncb, lB, red);
  {
    float as[NG];
    ldv<NG>(p.asum + frame * NG, as);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      const int T = p.lengths[g * NG + n];
      gat[n] = (t + 1) < T;
      lB[n] = gat[n] ? p.leaky * lB[n] : 0.f;
      cst[n] = ((t + 1) == T) ? p.inv_tot[g * NG + n] * (1.0f + p.leaky * p.pi_sum) : 0.f;
      inv_as[n] = 1.0f / as[n];
      lk[n] = p.leaky * as[n];
    }
  }
  const size_t xoff = ((size_t)g * p.Tmax + t) * (size_t)p.P * NG;
  if (beta_role) {
    const float4* src = reinterpret_cast<const float4*>(p.xs + xoff);
    float4* dst = reinterpret_cast<float4*>(xs_l);
    const int n4 = p.P * NG / 4;
    for (int i = tid; i < n4; i += kDenThreads) dst[i] = src[i];
    for (int i = n4 * 4 + tid; i < p.P * NG; i += kDenThreads) xs_l[i] = p.xs[xoff + i];
  }
  const int nrows = ord.nrows[chunk];
  for (int i = tid; i < nrows * NG; i += kDenThreads) acc[i] = 0.f;
  __syncthreads();

  const float* alpha_t = p.alpha + frame * (size_t)p.S * NG;
  const float* beta_n = p.beta + (frame + 1) * (size_t)p.S * NG;
  const int wb0 = ord.wb_off[chunk], wb1 = ord.wb_off[chunk + 1];
  for (int wb = wb0 + w; wb < wb1; wb += kDenWaves) {
    const uint32_t meta = ord.meta[(size_t)wb * 64 + lane];
    int c = meta & 0xffffu;
    const uint32_t mask = meta >> 16;
    int4 rec[kK];
#pragma unroll
    for (int j = 0; j < kK; ++j) rec[j] = ord.arcs[((size_t)wb * kK + j) * 64 + lane];
    float sum[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) sum[n] = 0.f;
    if (beta_role) {
      // rec = {dst, pdf, prob, -}: acc[src-row] += prob * x[pdf] * beta[t+1][dst]
      float b[kK][NG];
#pragma unroll
      for (int j = 0; j < kK; ++j) ldv<NG>(beta_n + (size_t)rec[j].x * NG, b[j]);
#pragma unroll
      for (int j = 0; j < kK; ++j) {
        const float prob = __int_as_float(rec[j].z);
        float xv[NG];
        ldv<NG>(xs_l + (size_t)rec[j].y * NG, xv);
#pragma unroll
        for (int n = 0; n < NG; ++n) {
          const float bv = gat[n] ? b[j][n] + lB[n] : cst[n];
          sum[n] += prob * xv[n] * bv;
        }
        if ((mask >> j) & 1u) {
#pragma unroll
          for (int n = 0; n < NG; ++n) { atomicAdd(&acc[c * NG + n], sum[n]); sum[n] = 0.f; }
          ++c;
        }
      }
    } else {
      // rec = {src, dst, prob, pi*prob}: acc[pdf-row] += alpha'[t][src] * prob * beta[t+1][dst]
      float a[kK][NG], b[kK][NG];
#pragma unroll
      for (int j = 0; j < kK; ++j) {
        ldv<NG>(alpha_t + (size_t)rec[j].x * NG, a[j]);
        ldv<NG>(beta_n + (size_t)rec[j].y * NG, b[j]);
      }
#pragma unroll
      for (int j = 0; j < kK; ++j) {
        const float prob = __int_as_float(rec[j].z), pp = __int_as_float(rec[j].w);
#pragma unroll
        for (int n = 0; n < NG; ++n) {
          const float bv = gat[n] ? b[j][n] + lB[n] : cst[n];
          sum[n] += (a[j][n] * prob + lk[n] * pp) * bv;
        }
        if ((mask >> j) & 1u) {
#pragma unroll
          for (int n = 0; n < NG; ++n) { atomicAdd(&acc[c * NG + n], sum[n]); sum[n] = 0.f; }
          ++c;
        }
      }
    }
  }
  __syncthreads();

  const int row0 = ord.row0[chunk];
  const bool atomic = ord.atomic[chunk] != 0;
  if (beta_role) {
    float* beta_t = p.beta + frame * (size_t)p.S * NG;
    float loc[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) loc[n] = 0.f;
    for (int r = tid; r < nrows; r += kDenThreads) {
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
    block_sum<NG>(loc, red);
    if (tid == 0) stv<NG>(p.bpart + (frame * ncb + chunk) * NG, loc);
  } else {
    float* gam_t = p.gamma + xoff;
    for (int r = tid; r < nrows; r += kDenThreads) {
      float v[NG], xv[NG];
      ldv<NG>(p.xs + xoff + (size_t)(row0 + r) * NG, xv);
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
}

// Kaldi's consistency check: sum_h alpha'[0,h] beta'[0,h] (should be 1 per sequence).
template <int NG>
__global__ void __launch_bounds__(kDenThreads) den_check(DenParams p, float* check) {
  __shared__ float red[kDenWaves * NG];
  const int g = blockIdx.x;
  const int ncb = p.bwd.n_chunks;
  float B0[NG];
  reduce_partials<NG>(p.bpart + ((size_t)g * (p.Tmax + 1)) * ncb * NG, ncb, B0, red);
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
  return ((size_t)P * NG + (size_t)kMaxRows * NG + (size_t)kDenWaves * NG) * sizeof(float);
}

int den_choose_ng(const pk2_den_graph* g) {
  const size_t limit = 160 * 1024;
  if (den_lds_bytes(g->P, 4) <= limit) return 4;
  if (den_lds_bytes(g->P, 2) <= limit) return 2;
  if (den_lds_bytes(g->P, 1) <= limit) return 1;
  return 0;
}

size_t den_workspace(const pk2_den_graph* g, int N, int Tmax, DenGeom* geom, DenBuffers* buf,
                     void* base) {
  DenGeom ge;
  ge.NG = den_choose_ng(g);
  if (ge.NG == 0) ge.NG = 1;
  ge.N = N; ge.Tmax = Tmax;
  ge.G = (N + ge.NG - 1) / ge.NG;
  Carver c(base);
  DenBuffers b;
  const size_t GN = (size_t)ge.G * ge.NG;
  b.alpha = c.take<float>(GN * (Tmax + 1) * (size_t)g->S);
  b.beta = c.take<float>(GN * (Tmax + 1) * (size_t)g->S);
  b.xs = c.take<float>(GN * (size_t)Tmax * g->P);
  b.gamma = c.take<float>(GN * (size_t)Tmax * g->P);
  b.apart = c.take<float>(GN * (Tmax + 1) * (size_t)g->h_fwd.n_chunks);
  b.bpart = c.take<float>(GN * (Tmax + 1) * (size_t)g->h_bwd.n_chunks);
  b.asum = c.take<float>(GN * (Tmax + 1));
  b.inv_tot = c.take<float>(GN);
  b.den_lp = c.take<float>(GN);
  b.check = c.take<float>(GN);
  b.lengths = c.take<int32_t>(GN);
  if (geom) *geom = ge;
  if (buf) *buf = b;
  return c.bytes();
}

template <int NG>
static int den_compute_t(pk2_den_graph* g, const float* logits, int64_t seq_stride,
                         int64_t frame_stride, const int32_t* lengths_host, const DenGeom& ge,
                         const DenBuffers& b, float leaky, hipStream_t stream) {
  const int Tmax = ge.Tmax, G = ge.G;
  const size_t GN = (size_t)G * NG;
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
  PK2_HIP(hipMemsetAsync(b.alpha, 0, GN * (Tmax + 1) * (size_t)g->S * sizeof(float), stream));
  PK2_HIP(hipMemsetAsync(b.beta, 0, GN * (Tmax + 1) * (size_t)g->S * sizeof(float), stream));
  PK2_HIP(hipMemsetAsync(b.gamma, 0, GN * (size_t)Tmax * g->P * sizeof(float), stream));
  PK2_HIP(hipMemsetAsync(b.bpart, 0, GN * (Tmax + 1) * (size_t)g->h_bwd.n_chunks * sizeof(float), stream));

  DenParams p;
  p.fwd = g->fwd; p.bwd = g->bwd; p.gam = g->gam;
  p.pi = g->d_pi;
  p.alpha = b.alpha; p.beta = b.beta; p.xs = b.xs; p.gamma = b.gamma;
  p.apart = b.apart; p.bpart = b.bpart; p.asum = b.asum; p.inv_tot = b.inv_tot;
  p.lengths = b.lengths;
  p.S = g->S; p.P = g->P; p.Tmax = Tmax;
  p.leaky = leaky; p.pi_sum = (float)g->pi_sum;

  const size_t lds = den_lds_bytes(g->P, NG);
  static bool attr_set[8] = {false};
  if (!attr_set[NG]) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_fwd_step<NG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_bwd_step<NG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[NG] = true;
  }

  hipLaunchKernelGGL(den_exp_transpose<NG>, dim3(Tmax, G), dim3(256), 0, stream, logits, seq_stride,
                     frame_stride, b.lengths, b.xs, g->P, Tmax);
  hipLaunchKernelGGL(den_init<NG>, dim3(std::min(256, (g->S + 255) / 256), G), dim3(256), 0, stream, p);
  PK2_LAUNCH_CHECK();
  for (int t = 0; t < Tmax; ++t)
    hipLaunchKernelGGL(den_fwd_step<NG>, dim3(g->fwd.n_chunks, G), dim3(kDenThreads), lds, stream, p, t);
  PK2_LAUNCH_CHECK();
  hipLaunchKernelGGL(den_finalize<NG>, dim3(G, NG), dim3(256), 0, stream, p, b.den_lp);
  for (int t = Tmax - 1; t >= 0; --t)
    hipLaunchKernelGGL(den_bwd_step<NG>, dim3(g->bwd.n_chunks + g->gam.n_chunks, G), dim3(kDenThreads),
                       lds, stream, p, t);
  hipLaunchKernelGGL(den_check<NG>, dim3(G), dim3(kDenThreads), 0, stream, p, b.check);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

int den_compute(pk2_den_graph* g, const float* logits, int64_t seq_stride, int64_t frame_stride,
                const int32_t* lengths_host, const DenGeom& ge, const DenBuffers& b, float leaky,
                hipStream_t stream) {
  int rc = den_upload(g);
  if (rc) return rc;
  if (den_choose_ng(g) == 0) {
    set_error("den graph: num_pdfs %d too large for the LDS-staged kernel", g->P);
    return PK2_ERR_LIMIT;
  }
  switch (ge.NG) {
    case 4: return den_compute_t<4>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream);
    case 2: return den_compute_t<2>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream);
    default: return den_compute_t<1>(g, logits, seq_stride, frame_stride, lengths_host, ge, b, leaky, stream);
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

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
using namespace pk2;

extern "C" int pk2_den_graph_create(int32_t num_states, int32_t num_pdfs, int64_t num_arcs,
                                    const int32_t* arc_src, const int32_t* arc_dst,
                                    const int32_t* arc_pdf, const float* arc_prob,
                                    int32_t start_state, pk2_den_graph** out) {
  PK2_REQUIRE(arc_src && arc_dst && arc_pdf && arc_prob && out, "den graph: null pointer");
  return build_graph(num_states, num_pdfs, num_arcs, arc_src, arc_dst, arc_pdf, arc_prob,
                     start_state, out);
}

// OpenFst binary StdVectorFst: header, optional symbol tables are not
// supported (Kaldi writes den.fst without them), then per state {f32 final, i64 narcs,
// arcs {i32 ilabel, i32 olabel, f32 weight, i32 nextstate}}.
extern "C" int pk2_den_graph_from_openfst(const char* path, int32_t num_pdfs, pk2_den_graph** out) {
  PK2_REQUIRE(path && out, "den graph: null pointer");
  FILE* f = fopen(path, "rb");
  if (!f) { set_error("cannot open %s", path); return PK2_ERR_IO; }
  auto fail = [&](const char* why) { fclose(f); set_error("%s: %s", path, why); return (int)PK2_ERR_IO; };
  auto rd = [&](void* p, size_t n) { return fread(p, 1, n, f) == n; };
  int32_t magic;
  if (!rd(&magic, 4) || magic != 2125659606) return fail("bad magic");
  auto rdstr = [&](std::string* s) {
    int32_t n;
    if (!rd(&n, 4) || n < 0 || n > 4096) return false;
    s->resize(n);
    return n == 0 || rd(&(*s)[0], n);
  };
  std::string fst_type, arc_type;
  if (!rdstr(&fst_type) || !rdstr(&arc_type)) return fail("bad header");
  if (fst_type != "vector" || arc_type != "standard") return fail("not a vector/standard FST");
  int32_t version, flags; uint64_t props; int64_t start, nstates, narcs_hdr;
  if (!rd(&version, 4) || !rd(&flags, 4) || !rd(&props, 8) || !rd(&start, 8) || !rd(&nstates, 8) ||
      !rd(&narcs_hdr, 8))
    return fail("short header");
  if (flags & 3) return fail("embedded symbol tables are not supported");
  if (nstates <= 0 || nstates > (1 << 30)) return fail("bad state count");
  std::vector<int32_t> src, dst, pdf; std::vector<float> prob;
  for (int64_t s = 0; s < nstates; ++s) {
    float fin; int64_t na;
    if (!rd(&fin, 4) || !rd(&na, 8) || na < 0) return fail("truncated state");
    for (int64_t k = 0; k < na; ++k) {
      int32_t il, ol, ns; float w;
      if (!rd(&il, 4) || !rd(&ol, 4) || !rd(&w, 4) || !rd(&ns, 4)) return fail("truncated arc");
      if (il <= 0) return fail("epsilon / negative ilabel in den.fst");
      src.push_back((int32_t)s); dst.push_back(ns); pdf.push_back(il - 1); prob.push_back(expf(-w));
    }
  }
  fclose(f);
  return build_graph((int32_t)nstates, num_pdfs, (int64_t)src.size(), src.data(), dst.data(),
                     pdf.data(), prob.data(), (int32_t)start, out);
}

extern "C" int pk2_den_graph_destroy(pk2_den_graph* g) {
  if (!g) return PK2_OK;
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
  return PK2_OK;
}

extern "C" int pk2_den_graph_info(const pk2_den_graph* g, int32_t* num_states, int32_t* num_pdfs,
                                  int64_t* num_arcs) {
  PK2_REQUIRE(g, "den graph: null handle");
  if (num_states) *num_states = g->S;
  if (num_pdfs) *num_pdfs = g->P;
  if (num_arcs) *num_arcs = g->A;
  return PK2_OK;
}

extern "C" int pk2_den_graph_initial_probs(const pk2_den_graph* g, float* host_out) {
  PK2_REQUIRE(g && host_out, "den graph: null pointer");
  memcpy(host_out, g->pi.data(), g->pi.size() * sizeof(float));
  return PK2_OK;
}

// Test hook: copies one host-side ordering out (which: 0 = by dst, 1 = by src, 2 = by pdf).
// Sizes are queried by passing null buffers.
extern "C" int pk2_den_graph_debug_ordering(const pk2_den_graph* g, int which, int64_t* n_arcs_padded,
                                            int32_t* n_chunks, int32_t* arcs_out /* int4 */,
                                            uint32_t* meta_out, int32_t* wb_off_out,
                                            int32_t* row0_out, int32_t* nrows_out,
                                            int32_t* atomic_out) {
  PK2_REQUIRE(g && which >= 0 && which < 3, "debug ordering: bad args");
  const HostOrdering& h = which == 0 ? g->h_fwd : (which == 1 ? g->h_bwd : g->h_gam);
  if (n_arcs_padded) *n_arcs_padded = (int64_t)h.arcs.size();
  if (n_chunks) *n_chunks = h.n_chunks;
  if (arcs_out) memcpy(arcs_out, h.arcs.data(), h.arcs.size() * sizeof(int4));
  if (meta_out) memcpy(meta_out, h.meta.data(), h.meta.size() * sizeof(uint32_t));
  if (wb_off_out) memcpy(wb_off_out, h.wb_off.data(), h.wb_off.size() * sizeof(int32_t));
  if (row0_out) memcpy(row0_out, h.row0.data(), h.row0.size() * sizeof(int32_t));
  if (nrows_out) memcpy(nrows_out, h.nrows.data(), h.nrows.size() * sizeof(int32_t));
  if (atomic_out) memcpy(atomic_out, h.atomic.data(), h.atomic.size() * sizeof(int32_t));
  return PK2_OK;
}

namespace pk2 {
void den_gamma_out_launch(const DenGeom& ge, const DenBuffers& b, int P, float scale, float* out,
                          int64_t ss, int64_t fs, hipStream_t stream);
}

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
