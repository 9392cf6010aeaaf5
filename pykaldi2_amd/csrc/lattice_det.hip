// Lattice determinisation on word labels (host code; reference bin/latgen.py:149 asks PyKaldi's recogniser for
// `determinize_lattice = True`, i.e. Kaldi's DeterminizeLatticePhonePrunedWrapper over the raw state-level lattice).
//
// [upstream-knowledge: Kaldi's determinize-lattice-pruned.cc, restated from its published algorithm; no Kaldi source or
// binary exists in this environment.]  The lattice is an acyclic FST whose arcs carry a word (0 = epsilon), a pair of
// costs (graph, acoustic) and at most one transition-id.  Determinisation happens in the semiring of CompactLattice
// weights: a weight is (costs, transition-id string); "plus" keeps the operand with the smaller total cost (ties: the
// smaller graph cost, then the shorter / lexicographically smaller string), "times" adds costs and concatenates strings.
// The result accepts exactly the word sequences of the input, each ONCE, with the costs and the alignment of its best
// path:
//  * a state of the output is a set of (input state, residual weight) pairs, closed under epsilon arcs (the better
//    residual per state wins), NORMALISED: the smallest residual cost pair and the longest common prefix of the strings
//    are divided out and travel on the arc that enters the subset (so equal futures meet in one state);
//  * subsets are identified by their (state, residual) lists, costs compared after rounding to 2^-10 (Kaldi's delta);
//  * pruning is exact: input arcs that lie on no complete path within `beam` of the best one are removed first (forward /
//    backward best costs of the input lattice); the output states are then built best-first with that exact cost-to-go
//    (A* with a consistent heuristic, i.e. Dijkstra on reduced costs: the best cost INTO a state is final when the state
//    is expanded), so an output arc is created only if the best complete path through it is within the beam, and states
//    beyond the beam are never expanded -- without this the number of subsets explodes on decoder lattices (most arcs
//    carry no word, a closure runs through thousands of states); a last pass drops what the earlier decisions left
//    dangling.  A subset keeps only final states and states with a word arc leaving them (Kaldi's minimal
//    representation).  The construction stops with PK2_ERR_LIMIT once `max_states` subsets exist -- the caller retries
//    with a smaller beam, as Kaldi's wrapper does.
// No minimisation (Kaldi's default: minimize = false).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <map>
#include <queue>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace pk2 {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

struct Cost { double g, a; double total() const { return g + a; } };

// Transition-id strings live in a trie (Kaldi keeps them in a hash-consed repository for the same reason): a string is a
// node, appending an id is one hash lookup, equal strings are equal node ids.  During an epsilon closure the residual
// strings grow by one id per lattice arc along chains of hundreds of arcs; as vectors that copying made a 25 k-arc
// lattice take minutes.
struct StringTrie {
  std::vector<int32_t> parent{-1}, label{0}, depth{0};          // node 0 = the empty string
  std::unordered_map<uint64_t, int32_t> child;
  int32_t append(int32_t node, int32_t id) {
    const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)id;
    auto it = child.find(key);
    if (it != child.end()) return it->second;
    const int32_t n = (int32_t)parent.size();
    parent.push_back(node); label.push_back(id); depth.push_back(depth[node] + 1);
    child.emplace(key, n);
    return n;
  }
  int32_t lca(int32_t a, int32_t b) const {
    while (depth[a] > depth[b]) a = parent[a];
    while (depth[b] > depth[a]) b = parent[b];
    while (a != b) { a = parent[a]; b = parent[b]; }
    return a;
  }
  // the ids of `node` below its ancestor `anc`, in order
  void suffix(int32_t anc, int32_t node, std::vector<int32_t>* out) const {
    out->resize((size_t)(depth[node] - depth[anc]));
    for (size_t k = out->size(); node != anc; node = parent[node]) (*out)[--k] = label[node];
  }
  int cmp(int32_t a, int32_t b) const {          // shorter first, then lexicographic
    if (a == b) return 0;
    if (depth[a] != depth[b]) return depth[a] < depth[b] ? -1 : 1;
    std::vector<int32_t> x, y;
    suffix(0, a, &x); suffix(0, b, &y);
    return x < y ? -1 : 1;
  }
};

struct Element { int32_t state; Cost w; int32_t str; };           // str: node of the trie

struct DetArc { int32_t src, dst, word; Cost w; std::vector<int32_t> str; };

struct InArc { int32_t dst, word, tid; Cost w; };

}  // namespace

struct DetLattice {
  int32_t start = 0;
  std::vector<DetArc> arcs;
  std::vector<Cost> final_w;                       // +inf = not final
  std::vector<std::vector<int32_t>> final_str;
};

namespace {

struct Determinizer {
  int32_t ns;
  std::vector<std::vector<InArc>> out;             // arcs by source state
  std::vector<double> final_cost;
  std::vector<double> beta;                        // best cost from a state to the end (+inf: dead)
  std::vector<int32_t> topo, pos;                  // states in topological order, and each state's position in it
  double beam, best;
  int64_t max_states;
  StringTrie trie;

  // -1 / 0 / +1: which of two CompactLattice weights "plus" keeps (smaller total, then smaller graph cost, then string)
  int compare(const Cost& x, int32_t xs, const Cost& y, int32_t ys) const {
    const double tx = x.total(), ty = y.total();
    if (tx < ty) return -1;
    if (tx > ty) return 1;
    if (x.g < y.g) return -1;
    if (x.g > y.g) return 1;
    return trie.cmp(xs, ys);
  }

  typedef std::vector<Element> Subset;             // sorted by state, epsilon-closed, normalised
  struct KeyLess {
    static long long q(double v) { return (long long)std::llround(v * 1024.0); }
    bool operator()(const Subset& a, const Subset& b) const {
      if (a.size() != b.size()) return a.size() < b.size();
      for (size_t i = 0; i < a.size(); ++i) {
        if (a[i].state != b[i].state) return a[i].state < b[i].state;
        const long long ag = q(a[i].w.g), bg = q(b[i].w.g), aa = q(a[i].w.a), ba = q(b[i].w.a);
        if (ag != bg) return ag < bg;
        if (aa != ba) return aa < ba;
        if (a[i].str != b[i].str) return a[i].str < b[i].str;        // equal strings are equal trie nodes
      }
      return false;
    }
  };
  std::map<Subset, int32_t, KeyLess> ids;
  std::vector<Subset> subsets;
  std::deque<int32_t> queue;
  DetLattice* res;
  // scratch of close(): per input state the index of its element in `cl` (-1: none), reset through `touched`
  std::vector<int32_t> slot, touched;
  std::vector<char> queued;
  std::vector<Element> cl;

  bool topo_sort() {
    std::vector<int32_t> indeg(ns, 0);
    for (int32_t s = 0; s < ns; ++s) for (const InArc& a : out[s]) ++indeg[a.dst];
    topo.clear();
    std::vector<int32_t> st;
    for (int32_t s = 0; s < ns; ++s) if (!indeg[s]) st.push_back(s);
    while (!st.empty()) {
      const int32_t s = st.back(); st.pop_back();
      topo.push_back(s);
      for (const InArc& a : out[s]) if (--indeg[a.dst] == 0) st.push_back(a.dst);
    }
    pos.assign(ns, -1);
    for (size_t i = 0; i < topo.size(); ++i) pos[topo[i]] = (int32_t)i;
    slot.assign(ns, -1);
    queued.assign(ns, 0);
    return (int32_t)topo.size() == ns;
  }

  // Epsilon closure of `els` (any order, states possibly repeated): best residual per state, then sorted by state.
  void close(std::vector<Element>* els) {
    cl.clear(); touched.clear();
    std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t>> work;      // topological positions
    auto offer = [&](const Element& e) {
      int32_t& k = slot[e.state];
      if (k < 0) { k = (int32_t)cl.size(); cl.push_back(e); touched.push_back(e.state); }
      else if (compare(e.w, e.str, cl[k].w, cl[k].str) < 0) cl[k] = e;
      else return;
      if (!queued[e.state]) { queued[e.state] = 1; work.push(pos[e.state]); }
    };
    for (const Element& e : *els) offer(e);
    // the lattice is acyclic: a state is final once every state before it in topological order has been relaxed
    while (!work.empty()) {
      const int32_t s = topo[work.top()];
      work.pop();
      queued[s] = 0;
      const Element cur = cl[slot[s]];
      for (const InArc& a : out[s]) {
        if (a.word != 0) continue;
        offer(Element{a.dst, Cost{cur.w.g + a.w.g, cur.w.a + a.w.a}, a.tid > 0 ? trie.append(cur.str, a.tid) : cur.str});
      }
    }
    els->clear();
    for (const Element& e : cl) if (std::isfinite(beta[e.state])) els->push_back(e);
    std::sort(els->begin(), els->end(), [](const Element& x, const Element& y) { return x.state < y.state; });
    for (int32_t s : touched) slot[s] = -1;
  }

  // Divides the smallest residual and the common string prefix out of a closed subset; they are returned.
  void normalise(std::vector<Element>* els, Cost* cw, std::vector<int32_t>* cs) {
    size_t bi = 0;
    int32_t anc = (*els)[0].str;
    for (size_t i = 1; i < els->size(); ++i) {
      if (compare((*els)[i].w, (*els)[i].str, (*els)[bi].w, (*els)[bi].str) < 0) bi = i;
      anc = trie.lca(anc, (*els)[i].str);
    }
    *cw = (*els)[bi].w;
    trie.suffix(0, anc, cs);
    std::vector<int32_t> tail;
    for (Element& e : *els) {
      e.w.g -= cw->g; e.w.a -= cw->a;
      if (anc != 0) {                  // re-root the residual string at the empty string
        trie.suffix(anc, e.str, &tail);
        int32_t n = 0;
        for (int32_t id : tail) n = trie.append(n, id);
        e.str = n;
      }
    }
  }

  int32_t subset_id(Subset&& sub) {
    auto it = ids.find(sub);
    if (it != ids.end()) return it->second;
    const int32_t id = (int32_t)subsets.size();
    ids.emplace(sub, id);
    subsets.push_back(std::move(sub));
    res->final_w.push_back(Cost{kInf, kInf});
    res->final_str.emplace_back();
    queue.push_back(id);
    return id;
  }

  // h of a subset: the best cost from it to the end (exact: backward costs of the input lattice)
  double to_go(const Subset& sub) const {
    double h = kInf;
    for (const Element& e : sub) h = std::min(h, e.w.total() + beta[e.state]);
    return h;
  }

  // Kaldi's "minimal representation": a state that is not final and has only epsilon arcs leaving it contributes nothing
  // the closure has not already followed -- dropping those keeps subsets small and lets equal futures meet.
  void minimal(std::vector<Element>* els) const {
    els->erase(std::remove_if(els->begin(), els->end(), [&](const Element& e) { return !keep[e.state]; }), els->end());
  }
  std::vector<char> keep;

  int run(int32_t start) {
    keep.assign(ns, 0);
    for (int32_t s = 0; s < ns; ++s) {
      keep[s] = std::isfinite(final_cost[s]) ? 1 : 0;
      for (const InArc& a : out[s]) if (a.word != 0) keep[s] = 1;
    }
    std::vector<Element> init{Element{start, Cost{0.0, 0.0}, 0}};
    close(&init);
    minimal(&init);
    if (init.empty()) { set_error("lattice_determinize: no path from the start state to a final state"); return PK2_ERR_INVALID; }
    // (the start subset is NOT normalised: what would be divided out of it has no arc to travel on)
    res->start = subset_id(std::move(init));
    // Best-first over the output states with an exact cost-to-go (A* with a consistent heuristic = Dijkstra on reduced
    // costs, negative arc costs included): when a state is taken off the queue the cost of the best path into it is
    // final, so "the best complete path through this arc is within the beam" is decided exactly while the lattice is
    // built, and states beyond the beam are never expanded.
    std::vector<double> alpha(1, 0.0);
    std::vector<char> done(1, 0);
    typedef std::pair<double, int32_t> QE;
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
    pq.push(QE(to_go(subsets[0]), 0));
    queue.clear();
    const double limit = best + beam + 1e-6;
    std::map<int32_t, std::vector<Element>> by_word;
    while (!pq.empty()) {
      const int32_t id = pq.top().second;
      pq.pop();
      if (done[id]) continue;
      done[id] = 1;
      if ((int64_t)subsets.size() > max_states) {
        set_error("lattice_determinize: more than %lld states at beam %g", (long long)max_states, beam);
        return PK2_ERR_LIMIT;
      }
      const Subset sub = subsets[id];          // (copy: `subsets` grows below)
      const double a0 = alpha[id];
      // final weight: the best (residual * final cost) among the final elements
      {
        bool any = false; Cost fw{kInf, kInf}; int32_t fs = 0;
        for (const Element& e : sub) {
          if (!std::isfinite(final_cost[e.state])) continue;
          const Cost w{e.w.g + final_cost[e.state], e.w.a};
          if (!any || compare(w, e.str, fw, fs) < 0) { fw = w; fs = e.str; any = true; }
        }
        if (any && a0 + fw.total() <= limit) { res->final_w[id] = fw; trie.suffix(0, fs, &res->final_str[id]); }
      }
      // successors by word
      by_word.clear();
      for (const Element& e : sub)
        for (const InArc& a : out[e.state]) {
          if (a.word == 0) continue;
          if (a0 + e.w.total() + a.w.total() + beta[a.dst] > limit) continue;      // no complete path through it survives
          by_word[a.word].push_back(Element{a.dst, Cost{e.w.g + a.w.g, e.w.a + a.w.a}, a.tid > 0 ? trie.append(e.str, a.tid) : e.str});
        }
      for (auto& kv : by_word) {
        std::vector<Element>& els = kv.second;
        close(&els);
        minimal(&els);
        if (els.empty()) continue;
        Cost w; std::vector<int32_t> s;
        normalise(&els, &w, &s);
        const double f = a0 + w.total() + to_go(els);
        if (f > limit) continue;
        const size_t before = subsets.size();
        const int32_t dst = subset_id(std::move(els));
        if (subsets.size() > before) { alpha.push_back(a0 + w.total()); done.push_back(0); pq.push(QE(f, dst)); }
        else if (a0 + w.total() < alpha[dst] && !done[dst]) { alpha[dst] = a0 + w.total(); pq.push(QE(f, dst)); }
        res->arcs.push_back(DetArc{id, dst, kv.first, w, std::move(s)});
      }
    }
    return PK2_OK;
  }
};

// Beam pruning of an acyclic lattice given as arcs + final costs: keeps what lies on a complete path within `beam` of the
// best one (exact forward / backward best costs); states are renumbered, dead ones dropped.
void prune_det(DetLattice* l, double beam) {
  const int32_t n = (int32_t)l->final_w.size();
  std::vector<std::vector<int32_t>> outa(n);
  std::vector<int32_t> indeg(n, 0);
  for (size_t i = 0; i < l->arcs.size(); ++i) { outa[l->arcs[i].src].push_back((int32_t)i); ++indeg[l->arcs[i].dst]; }
  std::vector<int32_t> topo, st;
  for (int32_t s = 0; s < n; ++s) if (!indeg[s]) st.push_back(s);
  while (!st.empty()) {
    const int32_t s = st.back(); st.pop_back();
    topo.push_back(s);
    for (int32_t i : outa[s]) if (--indeg[l->arcs[i].dst] == 0) st.push_back(l->arcs[i].dst);
  }
  std::vector<double> al(n, kInf), be(n, kInf);
  al[l->start] = 0.0;
  for (int32_t s : topo)
    if (std::isfinite(al[s])) for (int32_t i : outa[s]) al[l->arcs[i].dst] = std::min(al[l->arcs[i].dst], al[s] + l->arcs[i].w.total());
  for (auto it = topo.rbegin(); it != topo.rend(); ++it) {
    const int32_t s = *it;
    double b = std::isfinite(l->final_w[s].g) ? l->final_w[s].total() : kInf;
    for (int32_t i : outa[s]) b = std::min(b, l->arcs[i].w.total() + be[l->arcs[i].dst]);
    be[s] = b;
  }
  const double limit = be[l->start] + beam + 1e-6;
  std::vector<DetArc> kept;
  std::vector<char> used(n, 0);
  used[l->start] = 1;
  for (DetArc& a : l->arcs)
    if (al[a.src] + a.w.total() + be[a.dst] <= limit) { used[a.src] = used[a.dst] = 1; kept.push_back(std::move(a)); }
  std::vector<int32_t> renum(n, -1);
  int32_t m = 0;
  for (int32_t s = 0; s < n; ++s) if (used[s]) renum[s] = m++;
  std::vector<Cost> fw(m, Cost{kInf, kInf});
  std::vector<std::vector<int32_t>> fs(m);
  for (int32_t s = 0; s < n; ++s)
    if (used[s] && std::isfinite(l->final_w[s].g) && al[s] + l->final_w[s].total() <= limit) { fw[renum[s]] = l->final_w[s]; fs[renum[s]] = std::move(l->final_str[s]); }
  for (DetArc& a : kept) { a.src = renum[a.src]; a.dst = renum[a.dst]; }
  l->start = renum[l->start];
  l->arcs = std::move(kept); l->final_w = std::move(fw); l->final_str = std::move(fs);
}

}  // namespace
}  // namespace pk2

using namespace pk2;

typedef struct pk2_det_lattice pk2_det_lattice;

// Determinises the lattice given by its arcs (host arrays; arc i: src[i] -> dst[i], word[i] (0 = epsilon), tid[i] (0 =
// none), costs graph[i], acoustic[i]) and final costs (final_cost[s], +inf = not final), start state `start`.
// beam: arcs whose best complete path costs more than the best path + beam are dropped (the raw lattice is already
// lattice-beam pruned; pass that beam).  max_states: PK2_ERR_LIMIT beyond that many output states (retry with a smaller
// beam).  The result is read with pk2_det_lattice_sizes / _export and released with pk2_det_lattice_destroy.
extern "C" int pk2_lattice_determinize(int32_t num_states, int32_t start, int64_t num_arcs, const int32_t* src,
                                       const int32_t* dst, const int32_t* word, const int32_t* tid, const float* graph,
                                       const float* acoustic, const float* final_cost, double beam, int64_t max_states,
                                       pk2_det_lattice** out) {
  PK2_REQUIRE(out && num_states > 0 && start >= 0 && start < num_states && num_arcs >= 0 && final_cost, "lattice_determinize: bad args");
  PK2_REQUIRE(num_arcs == 0 || (src && dst && word && tid && graph && acoustic), "lattice_determinize: null arc arrays");
  PK2_REQUIRE(beam > 0 && max_states > 0, "lattice_determinize: beam and max_states must be positive");
  *out = nullptr;
  Determinizer d;
  d.ns = num_states; d.beam = beam; d.max_states = max_states;
  d.out.resize(num_states);
  for (int64_t i = 0; i < num_arcs; ++i) {
    PK2_REQUIRE(src[i] >= 0 && src[i] < num_states && dst[i] >= 0 && dst[i] < num_states && word[i] >= 0 && tid[i] >= 0,
                "lattice_determinize: arc %lld out of range", (long long)i);
    d.out[src[i]].push_back(InArc{dst[i], word[i], tid[i], Cost{(double)graph[i], (double)acoustic[i]}});
  }
  d.final_cost.resize(num_states);
  for (int32_t s = 0; s < num_states; ++s) d.final_cost[s] = std::isfinite(final_cost[s]) ? (double)final_cost[s] : kInf;
  PK2_REQUIRE(d.topo_sort(), "lattice_determinize: the lattice has a cycle");
  d.beta.assign(num_states, kInf);
  for (int32_t i = num_states - 1; i >= 0; --i) {
    const int32_t s = d.topo[i];
    double b = d.final_cost[s];
    for (const InArc& a : d.out[s]) b = std::min(b, a.w.total() + d.beta[a.dst]);
    d.beta[s] = b;
  }
  d.best = d.beta[start];
  PK2_REQUIRE(std::isfinite(d.best), "lattice_determinize: no path from the start state to a final state");
  // input arcs that lie on no complete path within the beam go first (exact forward costs of the input lattice) ...
  {
    std::vector<double> al(num_states, kInf);
    al[start] = 0.0;
    for (int32_t s : d.topo)
      if (std::isfinite(al[s])) for (const InArc& a : d.out[s]) al[a.dst] = std::min(al[a.dst], al[s] + a.w.total());
    const double limit = d.best + beam + 1e-6;
    for (int32_t s = 0; s < num_states; ++s) {
      std::vector<InArc>& v = d.out[s];
      v.erase(std::remove_if(v.begin(), v.end(), [&](const InArc& a) { return !(al[s] + a.w.total() + d.beta[a.dst] <= limit); }), v.end());
      if (!(al[s] + d.final_cost[s] <= limit)) d.final_cost[s] = kInf;
    }
    for (int32_t i = num_states - 1; i >= 0; --i) {      // backward costs of what is left (dead states: +inf)
      const int32_t s = d.topo[i];
      double b = d.final_cost[s];
      for (const InArc& a : d.out[s]) b = std::min(b, a.w.total() + d.beta[a.dst]);
      d.beta[s] = b;
    }
  }
  DetLattice* res = new DetLattice;
  d.res = res;
  const int rc = d.run(start);
  if (rc) { delete res; return rc; }
  // ... then the determinised lattice is pruned with the same beam: a word sequence whose best path is a patchwork of
  // arcs from different good paths can cost more than the beam allows
  prune_det(res, beam);
  *out = reinterpret_cast<pk2_det_lattice*>(res);
  return PK2_OK;
}

extern "C" int pk2_det_lattice_sizes(const pk2_det_lattice* h, int32_t* num_states, int32_t* start, int64_t* num_arcs,
                                     int64_t* arc_tids, int64_t* final_tids) {
  PK2_REQUIRE(h, "det_lattice_sizes: null handle");
  const DetLattice* l = reinterpret_cast<const DetLattice*>(h);
  int64_t at = 0, ft = 0;
  for (const DetArc& a : l->arcs) at += (int64_t)a.str.size();
  for (const auto& s : l->final_str) ft += (int64_t)s.size();
  if (num_states) *num_states = (int32_t)l->final_w.size();
  if (start) *start = l->start;
  if (num_arcs) *num_arcs = (int64_t)l->arcs.size();
  if (arc_tids) *arc_tids = at;
  if (final_tids) *final_tids = ft;
  return PK2_OK;
}

// Arcs (sorted by source state, then word): src, dst, word, graph, acoustic [num_arcs], tid_off [num_arcs + 1] into tids;
// per state final_graph / final_acoustic (+inf = not final), final_tid_off [num_states + 1] into final_tids.
extern "C" int pk2_det_lattice_export(const pk2_det_lattice* h, int32_t* src, int32_t* dst, int32_t* word, float* graph,
                                      float* acoustic, int64_t* tid_off, int32_t* tids, float* final_graph,
                                      float* final_acoustic, int64_t* final_tid_off, int32_t* final_tids) {
  PK2_REQUIRE(h && src && dst && word && graph && acoustic && tid_off && final_graph && final_acoustic && final_tid_off,
              "det_lattice_export: null pointer");
  const DetLattice* l = reinterpret_cast<const DetLattice*>(h);
  std::vector<int32_t> order(l->arcs.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
    return l->arcs[x].src != l->arcs[y].src ? l->arcs[x].src < l->arcs[y].src : l->arcs[x].word < l->arcs[y].word;
  });
  int64_t t = 0;
  for (size_t i = 0; i < order.size(); ++i) {
    const DetArc& a = l->arcs[order[i]];
    src[i] = a.src; dst[i] = a.dst; word[i] = a.word; graph[i] = (float)a.w.g; acoustic[i] = (float)a.w.a;
    tid_off[i] = t;
    for (int32_t v : a.str) { PK2_REQUIRE(tids, "det_lattice_export: null tids"); tids[t++] = v; }
  }
  tid_off[order.size()] = t;
  int64_t f = 0;
  for (size_t s = 0; s < l->final_w.size(); ++s) {
    const bool fin = std::isfinite(l->final_w[s].g);
    final_graph[s] = fin ? (float)l->final_w[s].g : std::numeric_limits<float>::infinity();
    final_acoustic[s] = fin ? (float)l->final_w[s].a : std::numeric_limits<float>::infinity();
    final_tid_off[s] = f;
    for (int32_t v : l->final_str[s]) { PK2_REQUIRE(final_tids, "det_lattice_export: null final_tids"); final_tids[f++] = v; }
  }
  final_tid_off[l->final_w.size()] = f;
  return PK2_OK;
}

extern "C" void pk2_det_lattice_destroy(pk2_det_lattice* h) { delete reinterpret_cast<DetLattice*>(h); }
