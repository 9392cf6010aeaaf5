// Internal declarations of the lattice path (MMIFunction / sMBRFunction):
// lattice_graph.hip (decoding graph, batch layout, export), lattice_decode.hip (token passing + lattice
// pruning), lattice_fb.hip (lattice forward-backward: MMI, sMBR, MPFE).
//
// MI355X design: the reference decodes every utterance on the CPU with Kaldi's LatticeFasterDecoder, copies
// the lattice posteriors to the GPU and loops over utterances in Python (reference ops/ops.py:55-66,
// bin/train_se.py:237-249).  Here the whole minibatch is decoded on the device and the lattices never leave HBM:
// by default a team of workgroups of one XCD per utterance, all frames inside one persistent launch (or a few
// graph-replayed launches per frame: lattice_decode_frames.hip), or one workgroup per utterance inside one launch
// (lattice_decode.hip: utterances are independent, so that recursion needs workgroup barriers only).  Per utterance the workspace holds
//   * a dense per-state table {best cost, token index} for the frame being built (reset sparsely),
//   * frame-layered token arrays and link arrays (pools sized from the utterance length: 288 GB of HBM make
//     Kaldi's periodic pruning unnecessary; one backward pruning pass runs after the last frame),
//   * the forward/backward quantities of the lattice forward-backward in float64 (as Kaldi).
#pragma once
#include <vector>

#include "common.h"

namespace pk2 {

struct DevDecodeGraph {
  const int32_t* e_off = nullptr;   // [S+1] emitting arcs (ilabel > 0) in CSR by source state
  const int32_t* e_dst = nullptr;
  const int32_t* e_tid = nullptr;
  const float* e_w = nullptr;
  const int32_t* n_off = nullptr;   // [S+1] non-emitting (epsilon-input) arcs
  const int32_t* n_dst = nullptr;
  const float* n_w = nullptr;
  const float* final_cost = nullptr;  // [S], +inf = not final
  int32_t S = 0, start = 0;
};

// Per-utterance record (device copy in the workspace; the result fields are written by the decoder).
struct LatUtt {
  int64_t tok_base, link_base, frame_base;   // element offsets of this utterance in the pools / frame tables
  int32_t T, tok_cap, link_cap;
  int32_t n_tok, n_link, status, any_final;
  float best_cost;
};

enum LatStatus : int32_t {
  kLatOk = 0,
  kLatTokenOverflow = 1,
  kLatLinkOverflow = 2,
  kLatNoSurvivor = 3,       // every token was pruned at some frame
  kLatEpsilonLoop = 4,      // epsilon closure did not converge (epsilon cycle with non-positive cost?)
  kLatNotDecoded = 5,
};

// Decoding state of one utterance between the launches of a frame (lattice_decode_frames.hip: a team of workgroups
// per utterance, several launches per frame).  Counters are bumped with agent-scope atomics; the plain fields are
// written by the last workgroup of a launch to finish and read by the next launch.
constexpr int kLatTeamHeavy = 16;  // heavy tokens of a frame the team list holds (further ones stay with their owner)
constexpr int kLatMaxTeam = 64;    // workgroups per team (PK2_LAT_TEAM) the per-workgroup bookkeeping has room for
constexpr int kLatEpsRounds = 2;   // epsilon relaxation launches per frame after round 0; a one-workgroup tail finishes deeper chains
struct LatFrame {
  int32_t f0, f1;              // tokens of the frame being expanded (utterance-local); new tokens are appended at f1
  int32_t link_end;            // links of all closed segments
  int32_t n_new, n_link, n_elist, n_arcs;
  int32_t ne_snap;             // epsilon-list entries present at the launch boundary
  uint32_t best_key;           // best final cost of [f0, f1), order-preserving encoding
  uint32_t best_next;          // same for the frame being built
  uint32_t nmin_key;           // best cost over the frame's arcs
  float cur_cutoff, adaptive, build_cutoff;
  int32_t status, arrive;
  int32_t hand[2];             // persistent decoder: counts too large for their field of a barrier's release word
  int32_t changed[kLatEpsRounds + 1];
  // Tokens of the frame being built whose state has very many epsilon arcs (the word-loop state: one per word): their arcs
  // are dealt to ALL workgroups of the team (left to the token's owner, that workgroup walked 20 k arcs while the others
  // waited: 6 of the 9 us of every relaxation round).  hlast[h][w] = the token's cost when workgroup w last walked its share.
  int32_t n_hlist, nh_snap;
  int32_t hlist[kLatTeamHeavy];
  int32_t hrec[kLatTeamHeavy * 4];   // {token, state, first epsilon arc, epsilon degree} of the list's tokens (team decoder)
  float hlast[kLatTeamHeavy * kLatMaxTeam];
  const float* ll_base;        // the utterance's log-likelihood rows (kept out of the graph-baked parameters)
  int64_t ll_stride;
  double fb_tot, fb_score;     // lattice forward-backward: total log-likelihood, expected accuracy
};

// Arrays of the workspace (device pointers).
struct LatPtrs {
  LatUtt* utt;
  uint32_t* st_cost;   // [N][S] order-preserving encoding of the float cost, 0xFFFFFFFF = no token
  int32_t* st_tok;     // [N][S] frame-local token index
  int32_t* tok_state; float* tok_cost; float* tok_extra; float* tok_final; int32_t* tok_level;
  double* alpha; double* beta; double* acc_f; double* acc_b;
  int4* link_rec;        // {source token, destination token, transition-id (0 = epsilon), graph cost bits}: one 16-byte access per link
  float* link_ac;        // acoustic cost
  float* link_delta;     // scratch of the lattice-beam pruning: the part of a link's extra cost that its ends' costs fix
  int4* e_rec;           // per call: emitting arcs of HCLG packed as {dst state, transition-id, weight bits, pdf}
  int32_t* frame_tok;    // per utterance [T+2]: first token of each frame (utterance-local), [T+1] = end
  int32_t* seg_off;      // per utterance [2(T+1)+1]: link segments: 2t = epsilon links inside frame t, 2t+1 = t -> t+1
  int32_t* seg_kept;     // per utterance [2(T+1)]: links of the segment that survive lattice pruning (compacted to its front)
  int32_t* frame_maxlev; // per utterance [T+1]: depth of the epsilon DAG inside the frame
  double* ref_post;      // per utterance [T]
  LatFrame* frame;       // [N]
  // forward-backward in the linear domain (round 6, lattice_fb.hip): exp(scaled log-likelihood) of every kept link, and per
  // frame the log scale of the frame's alpha values ([0, frame_total)) and beta values ([frame_total, 2 frame_total)); NaN = the
  // frame's values are logs already
  double* link_w;
  double* fb_scale;
  int64_t frame_total;
};

}  // namespace pk2

struct pk2_decode_graph {
  int32_t S = 0, start = 0, max_ilabel = 0;
  int64_t A = 0;
  std::vector<int32_t> e_off, e_dst, e_tid, n_off, n_dst;
  std::vector<int32_t> e_ol, n_ol;   // output labels (word ids) of the emitting / epsilon arcs, host only (0 when not given)
  std::vector<float> e_w, n_w, final_cost;
  bool uploaded = false;
  int device = -1;
  pk2::DevDecodeGraph dev;
  std::vector<void*> allocs;
};

struct pk2_lattice_batch {
  const pk2_decode_graph* graph = nullptr;
  int32_t N = 0, Tmax = 0;
  pk2_decoder_opts opts;
  std::vector<pk2::LatUtt> utt;
  int64_t tok_total = 0, link_total = 0, frame_total = 0;
  size_t bytes = 0;
  bool decoded = false;
};

namespace pk2 {
int decode_graph_upload(pk2_decode_graph* g);
// Carves the workspace (base may be null: returns the size only).
size_t lattice_carve(const pk2_lattice_batch* b, void* base, LatPtrs* out);
}  // namespace pk2
