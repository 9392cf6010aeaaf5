// Chain supervision of one utterance from its alignment (reference bin/train_chain.py:262-272:
// aligner.to_phone_alignment -> kaldi.chain.alignment_to_proto_supervision -> proto_supervision_to_supervision).
// Host-side integer work; nothing here touches the device.
//
// What Kaldi does [upstream knowledge: chain-supervision.cc, hmm-utils.cc, context-fst.cc]:
//   AlignmentToProtoSupervision   linear phone acceptor + per subsampled frame the set of phones allowed there
//                                 (every phone instance widened by left/right tolerance);
//   ProtoSupervisionToSupervision compose with the context transducer and with H (built without self-loops,
//                                 transition_scale 0), AddSelfLoops(reorder = true, self_loop_scale 0), map
//                                 transition-ids to pdf-id + 1, compose with TimeEnforcerFst (state = frame,
//                                 a label passes when its phone is allowed at that frame), Connect.
// The acceptor those generic FST operations define is built directly.  With reordered self-loops a visit of HMM
// state s that lasts k frames emits  forward-pdf(s), self-loop-pdf(s) x (k-1)  -- the transition out of s is
// taken first, its self-loops follow -- so
//   node (t, i, s)  = t frames consumed, the last one by HMM state s of phone instance i,
//   (t,i,s) -> (t+1,i,s)     self-loop pdf of (i,s)      if s has a self-loop and phone_i is allowed at frame t,
//   (t,i,s) -> (t+1,i,s')    forward pdf of (i,s')       for every transition s -> s' (s' emitting), same test,
//   (t,i,s) -> (t+1,i+1,0)   forward pdf of (i+1,0)      if s reaches the final HMM state; phone_{i+1} allowed at t,
//   start   -> (1,0,0)       forward pdf of (0,0)        phone_0 allowed at frame 0,
//   final: (T', last phone, s) with s -> final HMM state.
// pdfs come from the tree: ContextDependency::Compute(window of N phones around instance i, 0 beyond the ends,
// pdf-class of the HMM state).  All weights are 0.  States that cannot reach a final state are removed and the
// rest numbered in time order, which is the arc order pk2_chain_objf_and_deriv wants.
#include <algorithm>
#include <array>
#include <set>
#include <vector>

#include "common.h"

struct pk2_sup_model {
  int32_t N = 0, P = 0;
  std::vector<int32_t> phone2entry, entry_off, fwd_class, loop_class, trans_off, trans_dst;
  std::vector<int32_t> kind, key, a, b, pool;
  std::set<std::array<int32_t, 4>> tuples;
};

struct pk2_supervision {
  int32_t frames = 0, num_states = 0;
  std::vector<int32_t> src, dst, pdf, frame_off, state_time, finals, allowed_off, allowed;
};

namespace {

// EventMap::Map on the flattened tree.
bool tree_answer(const pk2_sup_model& m, const int32_t* window, int32_t pdf_class, int32_t* ans) {
  int32_t node = 0;
  for (size_t guard = 0; guard <= m.kind.size(); ++guard) {
    if (node < 0 || node >= (int32_t)m.kind.size()) return false;
    if (m.kind[node] == 0) { *ans = m.a[node]; return true; }
    const int32_t k = m.key[node];
    if (k < -1 || k >= m.N) return false;
    const int32_t v = k == -1 ? pdf_class : window[k];
    const int32_t* p = m.pool.data() + m.a[node];
    if (m.kind[node] == 1) {
      if (v < 0 || v >= m.b[node]) return false;
      node = p[v];
    } else {
      node = p[m.b[node] + (std::binary_search(p, p + m.b[node], v) ? 0 : 1)];
    }
  }
  return false;   // cycle in a malformed tree
}

}  // namespace

extern "C" {

int pk2_split_to_phones(const int32_t* tid_tstate, const int32_t* tid_phone, const uint8_t* tid_flags,
                        int32_t num_tids, const int32_t* ali, int32_t T, int32_t* phones, int32_t* durations,
                        int32_t* num_phones, int32_t* ok) {
  PK2_REQUIRE(tid_tstate && tid_phone && tid_flags && phones && durations && num_phones && ok, "pk2_split_to_phones: null argument");
  for (int32_t i = 0; i < T; ++i)
    PK2_REQUIRE(ali[i] >= 1 && ali[i] <= num_tids, "pk2_split_to_phones: transition-id %d at frame %d is outside [1, %d]", ali[i], i, num_tids);
  auto loop = [&](int32_t i) { return (tid_flags[ali[i]] & 1) != 0; };
  auto fin = [&](int32_t i) { return (tid_flags[ali[i]] & 2) != 0; };
  // IsReordered: inside one transition-state, does the self-loop follow the forward transition?
  bool reordered = true;
  for (int32_t i = 0; i + 1 < T; ++i) {
    if (tid_tstate[ali[i]] != tid_tstate[ali[i + 1]]) continue;
    if (loop(i) && !loop(i + 1)) { reordered = false; break; }
    if (!loop(i) && loop(i + 1)) { reordered = true; break; }
  }
  bool good = true;
  int32_t n = 0, begin = 0;
  auto cut = [&](int32_t end) {
    phones[n] = tid_phone[ali[begin]];
    for (int32_t j = begin; j < end; ++j) good = good && tid_phone[ali[j]] == phones[n];
    durations[n++] = end - begin;
    begin = end;
  };
  for (int32_t i = 0; i < T; ++i) {
    if (fin(i)) {
      if (reordered)
        while (i + 1 < T && loop(i + 1) && tid_tstate[ali[i + 1]] == tid_tstate[ali[i]]) ++i;
      cut(i + 1);
    } else if (i + 1 == T) {
      good = false;   // ends inside a phone
      cut(i + 1);
    } else if (tid_tstate[ali[i]] != tid_tstate[ali[i + 1]] && tid_phone[ali[i]] != tid_phone[ali[i + 1]]) {
      good = false;   // phone changes without a transition to the final state
      cut(i + 1);
    }
  }
  *num_phones = n;
  *ok = good ? 1 : 0;
  return PK2_OK;
}

pk2_sup_model* pk2_sup_model_create(int32_t max_phone, const int32_t* phone2entry, int32_t num_entries,
                                    const int32_t* entry_state_off, const int32_t* state_fwd_class,
                                    const int32_t* state_loop_class, const int32_t* state_trans_off,
                                    const int32_t* trans_dst, int32_t num_tuples, const int32_t* tuples,
                                    int32_t context_width, int32_t central_position, int32_t num_nodes,
                                    const int32_t* node_kind, const int32_t* node_key, const int32_t* node_a,
                                    const int32_t* node_b, int32_t pool_size, const int32_t* pool) {
  auto fail = [](const char* msg) -> pk2_sup_model* { pk2::set_error("pk2_sup_model_create: %s", msg); return nullptr; };
  if (max_phone < 1 || num_entries < 1 || !phone2entry || !entry_state_off || !state_fwd_class || !state_loop_class ||
      !state_trans_off || !trans_dst || !node_kind || !node_key || !node_a || !node_b || (pool_size > 0 && !pool))
    return fail("null or empty argument");
  if (context_width < 1 || central_position < 0 || central_position >= context_width) return fail("bad context width / central position");
  if (num_nodes < 1) return fail("empty tree");
  auto* m = new pk2_sup_model;
  m->N = context_width; m->P = central_position;
  m->phone2entry.assign(phone2entry, phone2entry + max_phone + 1);
  m->entry_off.assign(entry_state_off, entry_state_off + num_entries + 1);
  const int32_t ns = m->entry_off[num_entries];
  m->fwd_class.assign(state_fwd_class, state_fwd_class + ns);
  m->loop_class.assign(state_loop_class, state_loop_class + ns);
  m->trans_off.assign(state_trans_off, state_trans_off + ns + 1);
  m->trans_dst.assign(trans_dst, trans_dst + m->trans_off[ns]);
  m->kind.assign(node_kind, node_kind + num_nodes);
  m->key.assign(node_key, node_key + num_nodes);
  m->a.assign(node_a, node_a + num_nodes);
  m->b.assign(node_b, node_b + num_nodes);
  if (pool_size > 0) m->pool.assign(pool, pool + pool_size);
  for (int32_t i = 0; i < num_tuples; ++i) m->tuples.insert({tuples[4 * i], tuples[4 * i + 1], tuples[4 * i + 2], tuples[4 * i + 3]});
  const char* bad = nullptr;
  for (int32_t p = 0; p <= max_phone && !bad; ++p)
    if (m->phone2entry[p] < -1 || m->phone2entry[p] >= num_entries) bad = "phone2entry out of range";
  for (int32_t e = 0; e < num_entries && !bad; ++e) {
    const int32_t s0 = m->entry_off[e], s1 = m->entry_off[e + 1];
    if (s1 - s0 < 2) { bad = "a topology entry needs an emitting state and a final state"; break; }
    for (int32_t s = s0; s < s1 && !bad; ++s) {
      const bool last = s + 1 == s1;
      // HmmTopology::Check: only the last state is non-emitting, and it has no transitions
      if ((m->fwd_class[s] < 0) != last) bad = "exactly the last HMM state of an entry must be non-emitting";
      if (last && m->trans_off[s + 1] != m->trans_off[s]) bad = "the final HMM state has transitions";
      for (int32_t k = m->trans_off[s]; k < m->trans_off[s + 1] && !bad; ++k)
        if (m->trans_dst[k] < 0 || m->trans_dst[k] >= s1 - s0) bad = "transition to a state outside its entry";
    }
  }
  for (int32_t n = 0; n < num_nodes && !bad; ++n) {
    const int32_t k = m->kind[n];
    if (k < 0 || k > 2) { bad = "unknown tree node kind"; break; }
    if (k == 0) continue;
    const int64_t need = (int64_t)m->a[n] + m->b[n] + (k == 2 ? 2 : 0);
    if (m->a[n] < 0 || m->b[n] < 0 || need > (int64_t)m->pool.size()) bad = "tree node points outside the pool";
  }
  if (bad) { delete m; return fail(bad); }
  return m;
}

void pk2_sup_model_destroy(pk2_sup_model* m) { delete m; }

int pk2_sup_model_pdf(const pk2_sup_model* m, const int32_t* window, int32_t pdf_class, int32_t* pdf) {
  PK2_REQUIRE(m && window && pdf, "pk2_sup_model_pdf: null argument");
  PK2_REQUIRE(tree_answer(*m, window, pdf_class, pdf), "pk2_sup_model_pdf: the tree has no pdf for this context (pdf-class %d)", pdf_class);
  return PK2_OK;
}

pk2_supervision* pk2_supervision_create(const pk2_sup_model* m, const int32_t* phones, const int32_t* durations,
                                        int32_t n, int32_t f, int32_t ltol, int32_t rtol) {
  auto fail = [](const char* fmt, int x = 0, int y = 0) -> pk2_supervision* {
    pk2::set_error(fmt, x, y);
    return nullptr;
  };
  if (!m || !phones || !durations || n < 1 || f < 1 || ltol < 0 || rtol < 0) return fail("pk2_supervision_create: bad argument");
  const int32_t max_phone = (int32_t)m->phone2entry.size() - 1;
  int64_t T64 = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (durations[i] < 1) return fail("pk2_supervision_create: phone %d has duration %d", i, durations[i]);
    if (phones[i] < 1 || phones[i] > max_phone || m->phone2entry[phones[i]] < 0)
      return fail("pk2_supervision_create: phone %d (position %d) has no topology", phones[i], i);
    T64 += durations[i];
  }
  if (T64 > (1 << 28)) return fail("pk2_supervision_create: alignment too long");
  const int32_t T = (int32_t)T64, Tp = (T + f - 1) / f;

  auto* sup = new pk2_supervision;
  sup->frames = Tp;
  // ---- AlignmentToProtoSupervision: allowed phones per subsampled frame
  std::vector<std::vector<int32_t>> allowed(Tp);
  for (int32_t i = 0, cur = 0; i < n; cur += durations[i], ++i) {
    const int32_t t0 = std::max(0, cur - ltol), t1 = std::min(T, cur + durations[i] + rtol);
    for (int32_t t = (t0 + f - 1) / f; t < (t1 + f - 1) / f; ++t) allowed[t].push_back(phones[i]);
  }
  sup->allowed_off.assign(1, 0);
  for (auto& a : allowed) {
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
    sup->allowed.insert(sup->allowed.end(), a.begin(), a.end());
    sup->allowed_off.push_back((int32_t)sup->allowed.size());
  }
  // ---- per phone instance: its emitting HMM states with their pdfs
  std::vector<int32_t> inst_off(n + 1, 0), inst_entry(n);
  for (int32_t i = 0; i < n; ++i) {
    inst_entry[i] = m->phone2entry[phones[i]];
    inst_off[i + 1] = inst_off[i] + (m->entry_off[inst_entry[i] + 1] - m->entry_off[inst_entry[i]] - 1);
  }
  const int32_t K = inst_off[n];
  std::vector<int32_t> node_inst(K), fwd_pdf(K), loop_pdf(K);
  std::vector<uint8_t> can_end(K, 0);
  struct Succ { int32_t node, pdf; };
  std::vector<int32_t> succ_off(K + 1, 0);
  std::vector<Succ> succ;
  std::vector<int32_t> window(m->N);
  for (int32_t i = 0; i < n; ++i) {
    for (int32_t j = 0; j < m->N; ++j) {
      const int32_t q = i - m->P + j;
      window[j] = (q >= 0 && q < n) ? phones[q] : 0;
    }
    const int32_t s0 = m->entry_off[inst_entry[i]], ns = inst_off[i + 1] - inst_off[i];
    for (int32_t s = 0; s < ns; ++s) {
      const int32_t k = inst_off[i] + s;
      node_inst[k] = i;
      if (!tree_answer(*m, window.data(), m->fwd_class[s0 + s], &fwd_pdf[k]) ||
          !tree_answer(*m, window.data(), m->loop_class[s0 + s], &loop_pdf[k])) {
        delete sup;
        return fail("pk2_supervision_create: the tree has no pdf for phone %d at position %d", phones[i], i);
      }
      if (!m->tuples.empty() && !m->tuples.count({phones[i], s, fwd_pdf[k], loop_pdf[k]})) {
        delete sup;
        return fail("pk2_supervision_create: no transition-model tuple for phone %d, HMM state %d with the pdfs the tree gives", phones[i], s);
      }
    }
  }
  for (int32_t k = 0; k < K; ++k) {
    const int32_t i = node_inst[k], s = k - inst_off[i], ns = inst_off[i + 1] - inst_off[i];
    const int32_t gs = m->entry_off[inst_entry[i]] + s;
    for (int32_t q = m->trans_off[gs]; q < m->trans_off[gs + 1]; ++q) {
      const int32_t d = m->trans_dst[q];
      if (d == s) succ.push_back({k, loop_pdf[k]});
      else if (d < ns) succ.push_back({inst_off[i] + d, fwd_pdf[inst_off[i] + d]});
      else if (i + 1 < n) succ.push_back({inst_off[i + 1], fwd_pdf[inst_off[i + 1]]});
      else can_end[k] = 1;
    }
    succ_off[k + 1] = (int32_t)succ.size();
  }
  // ---- TimeEnforcerFst + Connect: forward reachability, then co-reachability
  std::vector<uint8_t> ok((size_t)Tp * n, 0);        // phone instance i may emit frame t
  {
    std::vector<int32_t> stamp(max_phone + 1, -1);
    for (int32_t t = 0; t < Tp; ++t) {
      for (int32_t q = sup->allowed_off[t]; q < sup->allowed_off[t + 1]; ++q) stamp[sup->allowed[q]] = t;
      for (int32_t i = 0; i < n; ++i) ok[(size_t)t * n + i] = stamp[phones[i]] == t;
    }
  }
  std::vector<uint8_t> live((size_t)(Tp + 1) * K, 0);   // row t: nodes after t frames (row 0 unused: the start state)
  if (ok[0]) live[(size_t)1 * K + 0] = 1;
  for (int32_t t = 1; t < Tp; ++t) {
    const uint8_t* cur = &live[(size_t)t * K];
    uint8_t* nxt = &live[(size_t)(t + 1) * K];
    const uint8_t* okt = &ok[(size_t)t * n];
    for (int32_t k = 0; k < K; ++k) {
      if (!cur[k]) continue;
      for (int32_t q = succ_off[k]; q < succ_off[k + 1]; ++q)
        if (okt[node_inst[succ[q].node]]) nxt[succ[q].node] = 1;
    }
  }
  {
    uint8_t* last = &live[(size_t)Tp * K];
    for (int32_t k = 0; k < K; ++k) last[k] = last[k] && can_end[k];
  }
  for (int32_t t = Tp - 1; t >= 1; --t) {
    uint8_t* cur = &live[(size_t)t * K];
    const uint8_t* nxt = &live[(size_t)(t + 1) * K];
    const uint8_t* okt = &ok[(size_t)t * n];
    for (int32_t k = 0; k < K; ++k) {
      if (!cur[k]) continue;
      bool any = false;
      for (int32_t q = succ_off[k]; q < succ_off[k + 1] && !any; ++q)
        any = okt[node_inst[succ[q].node]] && nxt[succ[q].node];
      cur[k] = any;
    }
  }
  if (!live[(size_t)1 * K + 0]) {
    delete sup;
    return fail("pk2_supervision_create: no path through the %d phones satisfies the time constraints over %d frames", n, Tp);
  }
  // ---- number the states in time order and emit the arcs
  std::vector<int32_t> id((size_t)(Tp + 1) * K, -1);
  sup->state_time.push_back(0);
  int32_t next_id = 1;
  for (int32_t t = 1; t <= Tp; ++t)
    for (int32_t k = 0; k < K; ++k)
      if (live[(size_t)t * K + k]) { id[(size_t)t * K + k] = next_id++; sup->state_time.push_back(t); }
  sup->num_states = next_id;
  sup->frame_off.push_back(0);
  sup->src.push_back(0); sup->dst.push_back(id[(size_t)1 * K + 0]); sup->pdf.push_back(fwd_pdf[0]);
  for (int32_t t = 1; t < Tp; ++t) {
    sup->frame_off.push_back((int32_t)sup->src.size());
    const uint8_t* okt = &ok[(size_t)t * n];
    for (int32_t k = 0; k < K; ++k) {
      if (!live[(size_t)t * K + k]) continue;
      for (int32_t q = succ_off[k]; q < succ_off[k + 1]; ++q) {
        const int32_t d = succ[q].node;
        if (!okt[node_inst[d]] || !live[(size_t)(t + 1) * K + d]) continue;
        sup->src.push_back(id[(size_t)t * K + k]);
        sup->dst.push_back(id[(size_t)(t + 1) * K + d]);
        sup->pdf.push_back(succ[q].pdf);
      }
    }
  }
  sup->frame_off.push_back((int32_t)sup->src.size());
  for (int32_t k = 0; k < K; ++k)
    if (live[(size_t)Tp * K + k]) sup->finals.push_back(id[(size_t)Tp * K + k]);
  return sup;
}

void pk2_supervision_destroy(pk2_supervision* s) { delete s; }

void pk2_supervision_sizes(const pk2_supervision* s, int32_t* frames, int32_t* num_states, int32_t* num_arcs,
                           int32_t* num_final, int32_t* num_allowed) {
  if (frames) *frames = s->frames;
  if (num_states) *num_states = s->num_states;
  if (num_arcs) *num_arcs = (int32_t)s->src.size();
  if (num_final) *num_final = (int32_t)s->finals.size();
  if (num_allowed) *num_allowed = (int32_t)s->allowed.size();
}

int pk2_supervision_copy(const pk2_supervision* s, int32_t* arc_src, int32_t* arc_dst, int32_t* arc_pdf,
                         float* arc_weight, int32_t* frame_off, int32_t* state_time, int32_t* final_state,
                         float* final_weight, int32_t* allowed_off, int32_t* allowed_phones) {
  PK2_REQUIRE(s, "pk2_supervision_copy: null supervision");
  auto put = [](int32_t* out, const std::vector<int32_t>& v) { if (out) std::copy(v.begin(), v.end(), out); };
  put(arc_src, s->src); put(arc_dst, s->dst); put(arc_pdf, s->pdf); put(frame_off, s->frame_off);
  put(state_time, s->state_time); put(final_state, s->finals); put(allowed_off, s->allowed_off);
  put(allowed_phones, s->allowed);
  if (arc_weight) std::fill(arc_weight, arc_weight + s->src.size(), 0.f);
  if (final_weight) std::fill(final_weight, final_weight + s->finals.size(), 0.f);
  return PK2_OK;
}

}  // extern "C"
