// Numerator forward-backward of the LF-MMI objective on gfx950.
//
// Replaces Kaldi's CPU NumeratorComputation inside kaldi.chain.compute_chain_objf_and_deriv
// (reference ops/ops.py:265): forward-backward over the per-utterance
// supervision FST (acyclic, one frame per arc, label = pdf), giving log p_num and the
// numerator posteriors.
//
// One workgroup per sequence.  The FST of an utterance is tiny (a few arcs per frame), so the
// job is latency-bound: alpha/beta live in LDS, arc scores (-w + logit) and their per-frame
// maxima are computed up front in parallel, and the serial loop over frames only touches LDS.
// Instead of Kaldi's log-domain adds, the recursion runs in probability space with a
// per-frame max shift and per-frame renormalisation (the scaled forward algorithm); arc
// posteriors of a frame are normalised by their own sum, so no scale has to be carried into
// the backward pass.  log p_num is accumulated in double.
#include <algorithm>
#include <vector>

#include "chain_internal.h"
#include "chain_num.h"

namespace pk2 {

constexpr int kNumMaxStates = 18000;  // 2 * 18000 * 4 B = 144 KB of LDS

// score[a] = -w[a] + logit[n][t(a)][pdf[a]];  frame_max[t] = max over the frame's arcs.
__global__ void __launch_bounds__(kNumThreads) num_scores(NumParams p, int64_t frame_base_total) {
  const int n = blockIdx.y;
  const int32_t* info = p.seqinfo + n * 8;
  const int fbase = info[0], T = info[1];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kNumThreads / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kNumThreads / 64);
  for (int t = wave; t < T; t += nwaves) {
    const int lo = p.frame_off[fbase + t], hi = p.frame_off[fbase + t + 1];
    const float* row = p.logits + (int64_t)n * p.seq_stride + (int64_t)t * p.frame_stride;
    float m = -INFINITY;
    for (int a = lo + lane; a < hi; a += 64) {
      const float s = row[p.arc_pdf[a]] - p.arc_w[a];
      p.score[a] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    if (lane == 0) p.frame_max[fbase - n + t] = m;  // frame_off has length+1 entries per sequence
  }
}

__global__ void __launch_bounds__(kNumThreads) num_fwd_bwd(NumParams p, int stage) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (stage) num_fwd_bwd_body<kNumThreads, true>(p, blockIdx.x, smem);
  else num_fwd_bwd_body<kNumThreads, false>(p, blockIdx.x, smem);
}

size_t num_workspace(int N, int64_t total_arcs, int64_t total_frames, NumBuffers* buf, void* base) {
  Carver c(base);
  NumBuffers b;
  b.score = c.take<float>((size_t)std::max<int64_t>(1, total_arcs));
  b.frame_max = c.take<float>((size_t)std::max<int64_t>(1, total_frames));
  b.seqinfo = c.take<int32_t>((size_t)N * 8);
  b.num_lp = c.take<float>((size_t)N);
  if (buf) *buf = b;
  return c.bytes();
}

struct InfoPack { static constexpr int kSeqs = 8; int32_t v[kSeqs * 8]; };
__global__ void store_info(InfoPack pack, int count, int32_t* out) {
  if ((int)threadIdx.x < count * 8) out[threadIdx.x] = pack.v[threadIdx.x];
}

int num_compute(const pk2_num_batch* nb, const float* logits, int64_t seq_stride,
                int64_t frame_stride, const int32_t* lengths, int N, float scale, float* grad,
                int64_t gseq_stride, int64_t gframe_stride, const NumBuffers& buf,
                hipStream_t stream, NumDeferred* defer) {
  PK2_REQUIRE(nb && nb->arc_src && nb->arc_dst && nb->arc_pdf && nb->arc_weight && nb->frame_off &&
                  nb->state_off && nb->final_state && nb->final_weight && nb->final_off,
              "numerator: null pointer in pk2_num_batch");
  int max_states = 0, Tmax = 0;
  int64_t fbase = 0;
  for (int base = 0; base < N; base += InfoPack::kSeqs) {
    InfoPack pack;
    int cnt = std::min(InfoPack::kSeqs, N - base);
    for (int i = 0; i < cnt; ++i) {
      int n = base + i;
      int ns = nb->state_off[n + 1] - nb->state_off[n];
      if (ns > kNumMaxStates) {
        set_error("numerator: sequence %d has %d states (limit %d)", n, ns, kNumMaxStates);
        return PK2_ERR_LIMIT;
      }
      PK2_REQUIRE(ns > 0 && lengths[n] > 0, "numerator: empty supervision for sequence %d", n);
      max_states = std::max(max_states, ns);
      Tmax = std::max(Tmax, lengths[n]);
      int32_t* v = pack.v + i * 8;
      v[0] = (int32_t)fbase; v[1] = lengths[n]; v[2] = ns;
      v[3] = nb->final_off[n]; v[4] = nb->final_off[n + 1]; v[5] = v[6] = v[7] = 0;
      fbase += lengths[n] + 1;
    }
    hipLaunchKernelGGL(store_info, dim3(1), dim3(64), 0, stream, pack, cnt, buf.seqinfo + base * 8);
  }
  NumParams p;
  p.arc_src = nb->arc_src; p.arc_dst = nb->arc_dst; p.arc_pdf = nb->arc_pdf; p.arc_w = nb->arc_weight;
  p.frame_off = nb->frame_off; p.final_state = nb->final_state; p.final_w = nb->final_weight;
  p.seqinfo = buf.seqinfo;
  p.logits = logits; p.seq_stride = seq_stride; p.frame_stride = frame_stride;
  p.score = buf.score; p.frame_max = buf.frame_max; p.num_lp = buf.num_lp;
  p.grad = grad; p.gseq_stride = gseq_stride; p.gframe_stride = gframe_stride;
  p.scale = scale;
  // LDS: alpha, beta, reduction scratch; plus, when it fits, the sequence's frame table and arc arrays (bounded here
  // by the batch totals: the per-sequence counts live on the device)
  size_t lds = ((size_t)2 * max_states + 8) * sizeof(float);
  // (round 4: by the largest SEQUENCE when the caller says so -- the batch total kept every minibatch with more than ~6k
  // arcs, i.e. about 30 s of audio in four utterances, off the staged path and out of the persistent denominator launch:
  // its numerator then ran unstaged on four workgroups for up to 1.8 ms behind the occupancy pass)
  const size_t arc_bound = (size_t)(nb->max_seq_arcs > 0 ? std::min(nb->max_seq_arcs, nb->total_arcs) : nb->total_arcs);
  const size_t staged = lds + ((size_t)2 * Tmax + 2 + 5 * arc_bound) * sizeof(float);
  const bool stage = staged <= 128 * 1024;
  if (stage) lds = staged;
  static PerDevice<bool> attr_set_pd(false); bool& attr_set = attr_set_pd.ref();
  if (!attr_set) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&num_fwd_bwd),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int blocks_x = std::max(1, std::min(64, (Tmax + 3) / 4));
  hipLaunchKernelGGL(num_scores, dim3(blocks_x, N), dim3(kNumThreads), 0, stream, p, (int64_t)0);
  if (defer) {   // the caller launches the forward-backward together with the denominator's occupancy kernel
    defer->p = p; defer->lds = lds; defer->N = N; defer->valid = true; defer->stage = stage;
    PK2_LAUNCH_CHECK();
    return PK2_OK;
  }
  hipLaunchKernelGGL(num_fwd_bwd, dim3(N), dim3(kNumThreads), lds, stream, p, stage ? 1 : 0);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

int num_launch_deferred(const NumDeferred& d, hipStream_t stream) {
  hipLaunchKernelGGL(num_fwd_bwd, dim3(d.N), dim3(kNumThreads), d.lds, stream, d.p, d.stage ? 1 : 0);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

}  // namespace pk2
