// Decoding graph (HCLG) handle, minibatch lattice layout, summary / export of decoded lattices.
// See lattice_internal.h for the design.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "lattice_internal.h"
#include "openfst_io.h"

using namespace pk2;

static int build_decode_graph(int32_t S, int32_t start, int64_t A, const int32_t* src, const int32_t* dst,
                              const int32_t* ilabel, const float* weight, const float* final_cost,
                              pk2_decode_graph** out, const int32_t* olabel = nullptr) {
  PK2_REQUIRE(S > 0 && start >= 0 && start < S && A >= 0, "decode graph: bad sizes");
  auto* g = new pk2_decode_graph;
  g->S = S; g->start = start; g->A = A;
  g->e_off.assign(S + 1, 0);
  g->n_off.assign(S + 1, 0);
  for (int64_t a = 0; a < A; ++a) {
    if (src[a] < 0 || src[a] >= S || dst[a] < 0 || dst[a] >= S || ilabel[a] < 0) {
      delete g;
      set_error("decode graph: arc %lld out of range", (long long)a);
      return PK2_ERR_INVALID;
    }
    (ilabel[a] > 0 ? g->e_off : g->n_off)[src[a] + 1]++;
    g->max_ilabel = std::max(g->max_ilabel, ilabel[a]);
  }
  for (int32_t s = 0; s < S; ++s) { g->e_off[s + 1] += g->e_off[s]; g->n_off[s + 1] += g->n_off[s]; }
  g->e_dst.resize(g->e_off[S]); g->e_tid.resize(g->e_off[S]); g->e_w.resize(g->e_off[S]);
  g->n_dst.resize(g->n_off[S]); g->n_w.resize(g->n_off[S]);
  g->e_ol.assign(g->e_off[S], 0); g->n_ol.assign(g->n_off[S], 0);
  std::vector<int32_t> ep(g->e_off.begin(), g->e_off.end() - 1), np(g->n_off.begin(), g->n_off.end() - 1);
  for (int64_t a = 0; a < A; ++a) {
    if (ilabel[a] > 0) {
      const int32_t k = ep[src[a]]++;
      g->e_dst[k] = dst[a]; g->e_tid[k] = ilabel[a]; g->e_w[k] = weight[a];
      if (olabel) g->e_ol[k] = olabel[a];
    } else {
      const int32_t k = np[src[a]]++;
      g->n_dst[k] = dst[a]; g->n_w[k] = weight[a];
      if (olabel) g->n_ol[k] = olabel[a];
    }
  }
  g->final_cost.assign(final_cost, final_cost + S);
  *out = g;
  return PK2_OK;
}

extern "C" int pk2_decode_graph_create(int32_t num_states, int32_t start_state, int64_t num_arcs,
                                       const int32_t* arc_src, const int32_t* arc_dst, const int32_t* arc_ilabel,
                                       const float* arc_weight, const float* final_cost, pk2_decode_graph** out) {
  PK2_REQUIRE(arc_src && arc_dst && arc_ilabel && arc_weight && final_cost && out, "decode graph: null pointer");
  return build_decode_graph(num_states, start_state, num_arcs, arc_src, arc_dst, arc_ilabel, arc_weight, final_cost, out);
}

extern "C" int pk2_decode_graph_from_openfst(const char* path, pk2_decode_graph** out) {
  PK2_REQUIRE(path && out, "decode graph: null pointer");
  FstArrays fst;
  const std::string why = read_openfst(path, &fst);
  if (!why.empty()) { set_error("%s: %s", path, why.c_str()); return PK2_ERR_IO; }
  return build_decode_graph((int32_t)fst.num_states, (int32_t)fst.start, (int64_t)fst.src.size(), fst.src.data(),
                            fst.dst.data(), fst.ilabel.data(), fst.weight.data(), fst.final_cost.data(), out, fst.olabel.data());
}

extern "C" int pk2_decode_graph_create_words(int32_t num_states, int32_t start_state, int64_t num_arcs,
                                             const int32_t* arc_src, const int32_t* arc_dst, const int32_t* arc_ilabel,
                                             const int32_t* arc_olabel, const float* arc_weight, const float* final_cost,
                                             pk2_decode_graph** out) {
  PK2_REQUIRE(arc_src && arc_dst && arc_ilabel && arc_olabel && arc_weight && final_cost && out, "decode graph: null pointer");
  return build_decode_graph(num_states, start_state, num_arcs, arc_src, arc_dst, arc_ilabel, arc_weight, final_cost, out,
                            arc_olabel);
}

// Output label (word id) of the HCLG arc behind each lattice link (host arrays, e.g. from pk2_lattice_export): the arc
// src -> dst with that transition-id (0 = epsilon arc) whose weight is closest to the link's graph cost.  -1 when the
// graph has no such arc.
extern "C" int pk2_decode_graph_link_words(const pk2_decode_graph* g, int64_t num_links, const int32_t* src_state,
                                           const int32_t* dst_state, const int32_t* tid, const float* graph_cost,
                                           int32_t* word_out) {
  PK2_REQUIRE(g && src_state && dst_state && tid && graph_cost && word_out && num_links >= 0, "link_words: bad args");
  for (int64_t l = 0; l < num_links; ++l) {
    const int32_t s = src_state[l];
    PK2_REQUIRE(s >= 0 && s < g->S, "link_words: state %d out of range", s);
    int32_t best = -1; float bd = std::numeric_limits<float>::infinity();
    if (tid[l] > 0) {
      for (int32_t k = g->e_off[s]; k < g->e_off[s + 1]; ++k)
        if (g->e_dst[k] == dst_state[l] && g->e_tid[k] == tid[l]) {
          const float dd = std::fabs(g->e_w[k] - graph_cost[l]);
          if (dd < bd) { bd = dd; best = g->e_ol[k]; }
        }
    } else {
      for (int32_t k = g->n_off[s]; k < g->n_off[s + 1]; ++k)
        if (g->n_dst[k] == dst_state[l]) {
          const float dd = std::fabs(g->n_w[k] - graph_cost[l]);
          if (dd < bd) { bd = dd; best = g->n_ol[k]; }
        }
    }
    word_out[l] = best;
  }
  return PK2_OK;
}

extern "C" int pk2_decode_graph_destroy(pk2_decode_graph* g) {
  if (!g) return PK2_OK;
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
  return PK2_OK;
}

extern "C" int pk2_decode_graph_info(const pk2_decode_graph* g, int32_t* num_states, int64_t* num_arcs,
                                     int32_t* max_ilabel) {
  PK2_REQUIRE(g, "decode graph: null handle");
  if (num_states) *num_states = g->S;
  if (num_arcs) *num_arcs = g->A;
  if (max_ilabel) *max_ilabel = g->max_ilabel;
  return PK2_OK;
}

namespace pk2 {

template <typename T>
static int upload_vec(pk2_decode_graph* g, const std::vector<T>& v, const T** out) {
  void* d = nullptr;
  PK2_HIP(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  g->allocs.push_back(d);
  if (!v.empty()) PK2_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(d);
  return PK2_OK;
}

int decode_graph_upload(pk2_decode_graph* g) {
  int dev = 0;
  PK2_HIP(hipGetDevice(&dev));
  if (g->uploaded) {
    PK2_REQUIRE(g->device == dev, "decode graph was uploaded to another device");
    return PK2_OK;
  }
  int rc;
  if ((rc = upload_vec(g, g->e_off, &g->dev.e_off))) return rc;
  if ((rc = upload_vec(g, g->e_dst, &g->dev.e_dst))) return rc;
  if ((rc = upload_vec(g, g->e_tid, &g->dev.e_tid))) return rc;
  if ((rc = upload_vec(g, g->e_w, &g->dev.e_w))) return rc;
  if ((rc = upload_vec(g, g->n_off, &g->dev.n_off))) return rc;
  if ((rc = upload_vec(g, g->n_dst, &g->dev.n_dst))) return rc;
  if ((rc = upload_vec(g, g->n_w, &g->dev.n_w))) return rc;
  if ((rc = upload_vec(g, g->final_cost, &g->dev.final_cost))) return rc;
  g->dev.S = g->S; g->dev.start = g->start;
  g->uploaded = true; g->device = dev;
  return PK2_OK;
}

size_t lattice_carve(const pk2_lattice_batch* b, void* base, LatPtrs* out) {
  Carver c(base);
  LatPtrs L;
  const size_t N = b->N, S = b->graph->S;
  L.utt = c.take<LatUtt>(N);
  L.st_cost = c.take<uint32_t>(N * S);
  L.st_tok = c.take<int32_t>(N * S);
  L.tok_state = c.take<int32_t>(b->tok_total); L.tok_cost = c.take<float>(b->tok_total);
  L.tok_extra = c.take<float>(b->tok_total); L.tok_final = c.take<float>(b->tok_total);
  L.tok_level = c.take<int32_t>(b->tok_total);
  L.alpha = c.take<double>(b->tok_total); L.beta = c.take<double>(b->tok_total);
  L.acc_f = c.take<double>(b->tok_total); L.acc_b = c.take<double>(b->tok_total);
  L.link_rec = c.take<int4>(b->link_total);
  L.link_ac = c.take<float>(b->link_total);
  L.link_delta = c.take<float>(b->link_total);
  L.e_rec = c.take<int4>(b->graph->e_dst.size());
  L.frame_tok = c.take<int32_t>(b->frame_total); L.seg_off = c.take<int32_t>(b->frame_total);
  L.seg_kept = c.take<int32_t>(b->frame_total); L.frame_maxlev = c.take<int32_t>(b->frame_total);
  L.ref_post = c.take<double>(b->frame_total);
  L.frame = c.take<LatFrame>(N);
  L.link_w = c.take<double>(b->link_total);
  L.fb_scale = c.take<double>(2 * b->frame_total);
  L.frame_total = b->frame_total;
  if (out) *out = L;
  return c.bytes();
}

}  // namespace pk2

extern "C" int pk2_lattice_batch_create(const pk2_decode_graph* g, const int32_t* lengths_host, int32_t num_seq,
                                        const pk2_decoder_opts* opts, pk2_lattice_batch** out) {
  PK2_REQUIRE(g && lengths_host && opts && out && num_seq > 0, "lattice batch: bad args");
  PK2_REQUIRE(opts->beam > 0.f && opts->lattice_beam > 0.f && opts->max_active > 0 && opts->min_active >= 0 &&
                  opts->acoustic_scale > 0.f,
              "lattice batch: bad decoder options");
  auto* b = new pk2_lattice_batch;
  b->graph = g; b->N = num_seq; b->opts = *opts;
  const int64_t tpf = opts->tokens_per_frame > 0 ? opts->tokens_per_frame : opts->max_active;
  const int64_t lpf = opts->links_per_frame > 0 ? opts->links_per_frame : 3 * (int64_t)opts->max_active;
  for (int32_t n = 0; n < num_seq; ++n) {
    const int32_t T = lengths_host[n];
    if (T <= 0) { delete b; set_error("lattice batch: utterance %d has %d frames", n, T); return PK2_ERR_INVALID; }
    LatUtt u{};
    u.T = T;
    u.tok_base = b->tok_total; u.link_base = b->link_total; u.frame_base = b->frame_total;
    const int64_t tc = (int64_t)(T + 1) * tpf + 64, lc = (int64_t)(T + 1) * lpf + 64;
    if (tc > std::numeric_limits<int32_t>::max() || lc > std::numeric_limits<int32_t>::max()) {
      delete b; set_error("lattice batch: pool of utterance %d exceeds 2^31 entries", n); return PK2_ERR_LIMIT;
    }
    u.tok_cap = (int32_t)tc; u.link_cap = (int32_t)lc;
    u.status = kLatNotDecoded;
    b->tok_total += tc; b->link_total += lc; b->frame_total += 2 * (int64_t)T + 4;
    b->Tmax = std::max(b->Tmax, T);
    b->utt.push_back(u);
  }
  b->bytes = lattice_carve(b, nullptr, nullptr);
  *out = b;
  return PK2_OK;
}

extern "C" size_t pk2_lattice_batch_bytes(const pk2_lattice_batch* b) { return b ? b->bytes : 0; }

extern "C" int pk2_lattice_batch_destroy(pk2_lattice_batch* b) {
  delete b;
  return PK2_OK;
}

extern "C" int pk2_lattice_summary(const pk2_lattice_batch* b, const void* workspace, int32_t* status,
                                   int32_t* num_tokens, int32_t* num_links, float* best_cost, void* stream_) {
  PK2_REQUIRE(b && workspace, "lattice summary: null pointer");
  PK2_REQUIRE(b->decoded, "lattice summary: pk2_lattice_decode has not run on this batch");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  LatPtrs L;
  lattice_carve(b, const_cast<void*>(workspace), &L);
  std::vector<LatUtt> h(b->N);
  PK2_HIP(hipMemcpyAsync(h.data(), L.utt, sizeof(LatUtt) * b->N, hipMemcpyDeviceToHost, stream));
  PK2_HIP(hipStreamSynchronize(stream));
  int rc = PK2_OK;
  for (int32_t n = 0; n < b->N; ++n) {
    if (status) status[n] = h[n].status;
    if (num_tokens) num_tokens[n] = h[n].n_tok;
    if (num_links) num_links[n] = h[n].n_link;
    if (best_cost) best_cost[n] = h[n].best_cost;
    if (h[n].status != kLatOk && rc == PK2_OK) {
      static const char* what[] = {"ok", "token pool overflow (raise tokens_per_frame)",
                                   "link pool overflow (raise links_per_frame)", "no surviving token",
                                   "epsilon closure did not converge", "not decoded"};
      set_error("lattice decode: utterance %d: %s", n, what[std::min(h[n].status, 5)]);
      rc = (h[n].status == kLatTokenOverflow || h[n].status == kLatLinkOverflow) ? PK2_ERR_LIMIT : PK2_ERR_NUMERIC;
    }
  }
  return rc;
}

extern "C" int pk2_lattice_export(const pk2_lattice_batch* b, const void* workspace, int32_t n, int32_t* num_tokens,
                                  int32_t* num_links, int32_t* tok_frame, int32_t* tok_state, float* tok_cost,
                                  float* tok_final, int32_t* link_src, int32_t* link_dst, int32_t* link_tid,
                                  float* link_graph, float* link_ac, void* stream_) {
  PK2_REQUIRE(b && workspace && n >= 0 && n < b->N, "lattice export: bad args");
  PK2_REQUIRE(b->decoded, "lattice export: pk2_lattice_decode has not run on this batch");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  LatPtrs L;
  lattice_carve(b, const_cast<void*>(workspace), &L);
  LatUtt u;
  PK2_HIP(hipMemcpyAsync(&u, L.utt + n, sizeof(LatUtt), hipMemcpyDeviceToHost, stream));
  PK2_HIP(hipStreamSynchronize(stream));
  PK2_REQUIRE(u.status == kLatOk, "lattice export: utterance was not decoded successfully");
  const int32_t T = u.T, nt = u.n_tok, nseg = 2 * (T + 1);
  std::vector<int32_t> ftok(T + 2), seg(nseg + 1), kept(nseg), st(nt);
  std::vector<float> cost(nt), extra(nt), fin(nt);
  auto d2h = [&](void* dst, const void* src, size_t bytes) {
    return bytes == 0 ? hipSuccess : hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
  };
  PK2_HIP(d2h(ftok.data(), L.frame_tok + u.frame_base, sizeof(int32_t) * (T + 2)));
  PK2_HIP(d2h(seg.data(), L.seg_off + u.frame_base, sizeof(int32_t) * (nseg + 1)));
  PK2_HIP(d2h(kept.data(), L.seg_kept + u.frame_base, sizeof(int32_t) * nseg));
  PK2_HIP(d2h(st.data(), L.tok_state + u.tok_base, sizeof(int32_t) * nt));
  PK2_HIP(d2h(cost.data(), L.tok_cost + u.tok_base, sizeof(float) * nt));
  PK2_HIP(d2h(extra.data(), L.tok_extra + u.tok_base, sizeof(float) * nt));
  PK2_HIP(d2h(fin.data(), L.tok_final + u.tok_base, sizeof(float) * nt));
  PK2_HIP(hipStreamSynchronize(stream));
  // tokens that survive, frame by frame, by HCLG state inside a frame (the order the oracle uses)
  std::vector<int32_t> remap(nt, -1), order;
  for (int32_t t = 0; t <= T; ++t) {
    std::vector<int32_t> fr;
    for (int32_t k = ftok[t]; k < ftok[t + 1]; ++k)
      if (std::isfinite(extra[k])) fr.push_back(k);
    std::sort(fr.begin(), fr.end(), [&](int32_t a, int32_t c) { return st[a] < st[c]; });
    for (int32_t k : fr) { remap[k] = (int32_t)order.size(); order.push_back(k); }
  }
  int64_t nl = 0;
  for (int32_t s = 0; s < nseg; ++s) nl += kept[s];
  if (num_tokens) *num_tokens = (int32_t)order.size();
  if (num_links) *num_links = (int32_t)nl;
  if (tok_frame || tok_state || tok_cost || tok_final) {
    int32_t t = 0;
    for (size_t i = 0; i < order.size(); ++i) {
      const int32_t k = order[i];
      while (k >= ftok[t + 1]) ++t;
      if (tok_frame) tok_frame[i] = t;
      if (tok_state) tok_state[i] = st[k];
      if (tok_cost) tok_cost[i] = cost[k];
      if (tok_final) tok_final[i] = (t == T) ? fin[k] : std::numeric_limits<float>::infinity();
    }
  }
  if (link_src || link_dst || link_tid || link_graph || link_ac) {
    // emitting links into frame t, then the epsilon links of frame t (the oracle's order)
    int64_t w = 0;
    std::vector<int4> rec; std::vector<float> ac;
    auto seg_copy = [&](int32_t s) -> int {
      const int32_t k = kept[s];
      rec.resize(k); ac.resize(k);
      const int64_t o = u.link_base + seg[s];
      PK2_HIP(d2h(rec.data(), L.link_rec + o, sizeof(int4) * (size_t)k));
      PK2_HIP(d2h(ac.data(), L.link_ac + o, 4 * (size_t)k));
      PK2_HIP(hipStreamSynchronize(stream));
      for (int32_t i = 0; i < k; ++i, ++w) {
        if (remap[rec[i].x] < 0 || remap[rec[i].y] < 0) { set_error("lattice export: kept link touches a pruned token"); return PK2_ERR_NUMERIC; }
        if (link_src) link_src[w] = remap[rec[i].x];
        if (link_dst) link_dst[w] = remap[rec[i].y];
        if (link_tid) link_tid[w] = rec[i].z;
        if (link_graph) link_graph[w] = __builtin_bit_cast(float, rec[i].w);
        if (link_ac) link_ac[w] = ac[i];
      }
      return PK2_OK;
    };
    for (int32_t t = 0; t <= T; ++t) {
      int rc;
      if (t > 0 && (rc = seg_copy(2 * t - 1))) return rc;
      if ((rc = seg_copy(2 * t))) return rc;
    }
  }
  return PK2_OK;
}
