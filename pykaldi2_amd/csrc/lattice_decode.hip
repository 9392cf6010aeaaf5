// On-the-fly lattice generation on gfx950: frame-synchronous token passing over HCLG with beam /
// max-active / min-active pruning, followed by lattice-beam pruning of the recorded links.
//
// Replaces `asr_decoder.decode(loglikes)` = Kaldi LatticeFasterDecoder with determinize_lattice = False
// (reference ops/ops.py:55,133; bin/train_se.py:172-183), restated in oracle/lattice_ref.py::decode
// (GetCutoff, ProcessEmitting, ProcessNonemitting, PruneForwardLinks[Final]).
//
// One workgroup (1024 threads) per utterance, one launch per minibatch -- see lattice_internal.h.  Inside a
// frame the workgroup
//   1. reduces the frame's token costs to the pruning cutoff (exact k-th smallest by a 3-pass radix select
//      in LDS when max_active / min_active bind),
//   2. stages the frame's log-likelihood row in LDS,
//   3. walks the emitting arcs of the surviving tokens twice: first for the best new cost (-> next cutoff =
//      best + adaptive beam), then to create tokens (atomicMin on a dense per-state table; the thread that
//      lowers an empty slot owns the new token) and links,
//   4. closes the new frame over epsilon arcs by rounds of relaxations to the exact fixed point, then
//      records the epsilon links from the final costs.
// All cost arithmetic is float32 in the oracle's order, (cur + acoustic) + graph, without FMA contraction,
// so token costs and the kept link set are bit-identical to the oracle's.
#include <cstdlib>
#include <cstring>

#include "lattice_decode_common.h"

// Costs must round exactly like the oracle's separate float32 multiply / add steps.  HIP's __fmul_rn /
// __fadd_rn are plain operators and hipcc's default -ffp-contract=fast ignores contraction pragmas, so this
// file is compiled with -ffp-contract=off (pykaldi2_amd/build.py FILE_FLAGS).

namespace pk2 {

// Packs the emitting arcs of HCLG with the pdf of their transition-id (one 16-byte gather per arc in the decoder).
__global__ void __launch_bounds__(256) lat_pack_arcs(DevDecodeGraph g, const int32_t* __restrict__ tid2pdf, int64_t n,
                                                     int4* __restrict__ out) {
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a < n) out[a] = make_int4(g.e_dst[a], g.e_tid[a], __float_as_int(g.e_w[a]), tid2pdf[g.e_tid[a]]);
}

__global__ void __launch_bounds__(kLatThreads) lat_decode_kernel(DecodeParams p) {
  __shared__ Shared sh;
  __shared__ int s_tok_end, s_link_end;
  const int n = blockIdx.x, tid = threadIdx.x;
  const LatUtt U = p.L.utt[n];
  const int T = U.T;
  const UttView V = make_view(p, n, U);
  float* tc = V.tc;
  int32_t* ftok = V.ftok; int32_t* seg = V.seg;
  int4* lrec = V.lrec; float* lac = V.lac;
  const int work_cap = U.tok_cap;

  if (tid == 0) {
    sh.status = kLatOk; sh.n_new = 1; sh.n_link = 0; sh.n_heavy = 0; sh.n_elist = 0;
#ifdef PK2_LAT_PROFILE
    for (int k = 0; k < 16; ++k) sh.prof[k] = 0;
    sh.prof_last = wall_clock64();
#endif
    s_tok_end = 0; s_link_end = 0;
    V.stc[p.g.start] = enc_cost(0.f);
    register_token(p, V, sh, 0, 0, p.g.start);
    ftok[0] = 0; seg[0] = 0;
  }
  __syncthreads();
  close_frame(p, V, sh, 0, p.beam, &s_link_end, &s_tok_end, 0);   // InitDecoding: ProcessNonemitting(beam)
  if (tid == 0) ftok[1] = s_tok_end;
  __syncthreads();

  for (int t = 0; t < T && sh.status == kLatOk; ++t) {
    const int f0 = ftok[t], f1 = s_tok_end, nt = f1 - f0;
    // ---- GetCutoff ----
    float lmin = INFINITY;
    for (int i = f0 + tid; i < f1; i += kLatThreads) lmin = fminf(lmin, tc[i]);
    const float best = block_min(lmin, sh);
    const float beam_cutoff = best + p.beam;
    float cur_cutoff = beam_cutoff, adaptive = p.beam;
    const bool chk_max = nt > p.max_active, chk_min = p.min_active > 0 && nt > p.min_active;
    if (chk_max || chk_min) {
      int c_lt = 0, c_le = 0;
      for (int i = f0 + tid; i < f1; i += kLatThreads) { c_lt += tc[i] < beam_cutoff; c_le += tc[i] <= beam_cutoff; }
      c_lt = block_sum_i(c_lt, sh);
      c_le = block_sum_i(c_le, sh);
      if (chk_max && c_lt > p.max_active) {
        cur_cutoff = kth_smallest_in_range(tc + f0, nt, p.max_active, best, beam_cutoff, sh);
        adaptive = (cur_cutoff - best) + p.beam_delta;
      } else if (chk_min && c_le <= p.min_active) {
        cur_cutoff = kth_smallest(tc + f0, nt, p.min_active, sh);
        adaptive = (cur_cutoff - best) + p.beam_delta;
      }
    }
    LAT_T(0);
    // ---- acoustic scores of the frame ----
    const float* row = p.loglikes + (int64_t)n * p.seq_stride + (int64_t)t * p.frame_stride;
    for (int i = tid; i < p.P; i += kLatThreads) sh.ll[i] = row[i];
    // ---- arc work list: one entry per emitting arc of the surviving tokens ----
    int run = 0;          // arcs listed so far (all threads)
    for (int base = f0; base < f1; base += 4 * kLatThreads) {
      int2 ar[4];
      int mine = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + tid * 4 + q;
        ar[q] = make_int2(0, 0);
        if (i < f1 && tc[i] <= cur_cutoff) ar[q] = V.tarc[i];
        mine += ar[q].y;
      }
      int total;
      int o = run + block_exclusive_scan(mine, sh, &total);
      if (o + mine <= work_cap) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = base + tid * 4 + q;
          for (int k = 0; k < ar[q].y; ++k) V.work[o + k] = make_int2(i, ar[q].x + k);
          o += ar[q].y;
        }
      } else if (mine > 0) {
        sh.status = kLatTokenOverflow;
      }
      run += total;
    }
    __syncthreads();
    if (sh.status != kLatOk) break;
    const int n_arcs = run;
    LAT_T(1);
    // ---- pass 1: cost of every listed arc, best new cost ----
    float nmin = INFINITY;
    {
      constexpr int U4 = 4;
      for (int j0 = tid; j0 < n_arcs; j0 += U4 * kLatThreads) {
        int2 wk[U4]; int4 er[U4]; float c[U4];
#pragma unroll
        for (int q = 0; q < U4; ++q)
          if (j0 + q * kLatThreads < n_arcs) wk[q] = V.work[j0 + q * kLatThreads];
#pragma unroll
        for (int q = 0; q < U4; ++q)
          if (j0 + q * kLatThreads < n_arcs) { er[q] = V.erec[wk[q].y]; c[q] = tc[wk[q].x]; }
#pragma unroll
        for (int q = 0; q < U4; ++q) {
          const int j = j0 + q * kLatThreads;
          if (j < n_arcs) {
            const float ac = -__fmul_rn(p.ac_scale, sh.ll[er[q].w]);
            const float tot = __fadd_rn(__fadd_rn(c[q], ac), __int_as_float(er[q].z));
            V.work_tot[j] = tot;
            nmin = fminf(nmin, tot);
          }
        }
      }
    }
    nmin = block_min(nmin, sh);
    if (!(nmin < INFINITY)) { if (tid == 0) sh.status = kLatNoSurvivor; __syncthreads(); break; }
    const float next_cutoff = nmin + adaptive;
    LAT_T(2);
    // ---- pass 2: tokens and links of frame t+1 ----
    const int l0 = s_link_end;
    for (int j0 = tid; j0 < n_arcs; j0 += 4 * kLatThreads) {
      float tot[4]; int2 wk[4]; int4 er[4]; uint32_t old[4]; bool acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = j0 + q * kLatThreads;
        acc[q] = false;
        if (j < n_arcs) { tot[q] = V.work_tot[j]; acc[q] = tot[q] < next_cutoff; }
        if (acc[q]) wk[q] = V.work[j];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (acc[q]) er[q] = V.erec[wk[q].y];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (acc[q]) old[q] = atomicMin(&V.stc[er[q].x], enc_cost(tot[q]));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!acc[q]) continue;
        if (old[q] == kEmpty) register_token(p, V, sh, f1, atomicAdd(&sh.n_new, 1), er[q].x);
        const int li = l0 + atomicAdd(&sh.n_link, 1);
        if (li < V.link_cap) {
          lrec[li] = make_int4(wk[q].x, er[q].x /* state for now */, er[q].y, er[q].z);
          lac[li] = -__fmul_rn(p.ac_scale, sh.ll[er[q].w]);
        } else {
          sh.status = kLatLinkOverflow;
        }
      }
    }
    __syncthreads();
    if (sh.status != kLatOk) break;
    const int l1 = l0 + sh.n_link;
    for (int l = l0 + tid; l < l1; l += kLatThreads) lrec[l].y = f1 + ld_coherent(&V.stt[lrec[l].y]);
    __syncthreads();
    if (tid == 0) { s_link_end = l1; seg[2 * t + 2] = l1; sh.n_link = 0; }
    __syncthreads();
    LAT_T(3);
    close_frame(p, V, sh, f1, next_cutoff, &s_link_end, &s_tok_end, 2 * t + 2);
    if (tid == 0) ftok[t + 2] = s_tok_end;
    __syncthreads();
  }
  __syncthreads();
  if (sh.status != kLatOk) {
    if (tid == 0) { p.L.utt[n].status = sh.status; p.L.utt[n].n_tok = s_tok_end; p.L.utt[n].n_link = s_link_end; }
    return;
  }
  LAT_T(7);

  finish_and_prune(p, V, sh, n, T, s_tok_end, s_link_end);
#ifdef PK2_LAT_PROFILE
  if (tid == 0 && n == 0)
    printf("lat_decode utt0 T=%d (10 ns ticks): cutoff %lld worklist %lld pass1 %lld pass2 %lld closure %lld epslinks %lld finalise %lld | final %lld prune_eps %lld levels %lld prune_em %lld\n",
           T, sh.prof[0], sh.prof[1], sh.prof[2], sh.prof[3], sh.prof[4], sh.prof[5], sh.prof[6], sh.prof[7], sh.prof[8], sh.prof[9], sh.prof[10]);
#endif
}

// Second half of the lattice pruning: link removal and epsilon-DAG depths, frames dealt round-robin to the
// workgroups of an utterance (prune_segments).
constexpr int kPruneTeam = 32;
__global__ void __launch_bounds__(kLatThreads) lat_prune_segments(DecodeParams p) {
  __shared__ Shared sh;
  const int n = blockIdx.y;
  const LatUtt U = p.L.utt[n];
  if (U.status != kLatOk) return;
  prune_segments(p, make_view(p, n, U), sh, U.T, blockIdx.x, gridDim.x);
}

}  // namespace pk2

using namespace pk2;

extern "C" int pk2_lattice_decode(pk2_lattice_batch* b, const float* loglikes, int64_t seq_stride,
                                  int64_t frame_stride, int32_t num_pdfs, const int32_t* tid2pdf, int32_t num_tids,
                                  void* workspace, void* stream_) {
  PK2_REQUIRE(b && loglikes && tid2pdf && workspace, "lattice decode: null pointer");
  PK2_REQUIRE(num_pdfs > 0 && num_pdfs <= kMaxPdfsLds, "lattice decode: num_pdfs %d exceeds the LDS row limit %d",
              num_pdfs, kMaxPdfsLds);
  PK2_REQUIRE(num_tids >= b->graph->max_ilabel, "lattice decode: HCLG uses transition-id %d but the model has %d",
              b->graph->max_ilabel, num_tids);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = decode_graph_upload(const_cast<pk2_decode_graph*>(b->graph));
  if (rc) return rc;
  LatPtrs L;
  lattice_carve(b, workspace, &L);
  PK2_HIP(hipMemcpyAsync(L.utt, b->utt.data(), sizeof(LatUtt) * b->N, hipMemcpyHostToDevice, stream));
  PK2_HIP(hipMemsetAsync(L.st_cost, 0xFF, sizeof(uint32_t) * (size_t)b->N * b->graph->S, stream));
  PK2_HIP(hipMemsetAsync(L.st_tok, 0xFF, sizeof(int32_t) * (size_t)b->N * b->graph->S, stream));
  PK2_HIP(hipMemsetAsync(L.seg_kept, 0, sizeof(int32_t) * (size_t)b->frame_total, stream));
  DecodeParams p;
  memset(&p, 0, sizeof(p));     // the frames decoder caches graphs by the bytes of p: no stray padding
  p.g = b->graph->dev; p.L = L;
  p.loglikes = loglikes; p.seq_stride = seq_stride; p.frame_stride = frame_stride; p.P = num_pdfs;
  p.tid2pdf = tid2pdf; p.num_tids = num_tids;
  p.beam = b->opts.beam; p.lattice_beam = b->opts.lattice_beam; p.beam_delta = b->opts.beam_delta;
  p.ac_scale = b->opts.acoustic_scale; p.max_active = b->opts.max_active; p.min_active = b->opts.min_active;
  const int64_t n_emit = (int64_t)b->graph->e_dst.size();
  if (n_emit > 0)
    hipLaunchKernelGGL(lat_pack_arcs, dim3((unsigned)((n_emit + 255) / 256)), dim3(256), 0, stream, p.g, tid2pdf, n_emit, L.e_rec);
  // Default: a team of PK2_LAT_TEAM workgroups per utterance, a few launches per frame (lattice_decode_frames.hip);
  // PK2_LAT_DECODER=wg: one workgroup per utterance, one launch (this file).
  static const bool frames = [] { const char* e = getenv("PK2_LAT_DECODER"); return !(e && strcmp(e, "wg") == 0); }();
  static const int team = [] { const char* e = getenv("PK2_LAT_TEAM"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  if (frames) {
    rc = lattice_decode_frames(p, b->N, b->Tmax, team, stream);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(lat_decode_kernel, dim3(b->N), dim3(kLatThreads), 0, stream, p);
    PK2_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(lat_prune_segments, dim3(kPruneTeam, b->N), dim3(kLatThreads), 0, stream, p);
  PK2_LAUNCH_CHECK();
  b->decoded = true;
  return PK2_OK;
}

extern "C" int pk2_lattice_persist_status(int32_t* state, uint32_t* abort_flag) {
  PK2_REQUIRE(state && abort_flag, "lattice_persist_status: null pointer");
  PK2_HIP(hipDeviceSynchronize());
  unsigned f = 0;
  *state = lattice_persist_status(&f);
  *abort_flag = f;
  return PK2_OK;
}
