// Internal declarations shared by chain_den.hip / chain_num.hip / chain_objf.hip.
#pragma once
#include <vector>

#include "common.h"

namespace pk2 {

// Work decomposition of one arc ordering (arcs sorted by a row key).
//  * a chunk = the arcs of a contiguous range of whole rows, processed by one
//    workgroup; <= kChunkArcs arcs and <= kMaxRows rows.  A row longer than
//    kChunkArcs is split over several single-row chunks flagged `atomic`
//    (their partial results are combined with global atomics).
//  * inside a chunk, arcs are padded to wave blocks of 64*kK arcs; lane i of a
//    wave block owns kK consecutive sorted arcs, stored interleaved so that
//    iteration j of all 64 lanes is one coalesced 1 KiB load:
//        arcs[(wb*kK + j)*64 + lane]  <->  sorted arc  wb*64*kK + lane*kK + j
//  * meta[wb*64+lane] = {c0, mask}: c0 = chunk-local row of the lane's
//    first arc, bit j of mask = "flush the running sum into row c after arc j,
//    then c++" (set where a row ends and at j = kK-1); kK <= 32.
// (the three tuning constants can be overridden at build time: -DPK2_DEN_K=.. -DPK2_DEN_CHUNK=.. for experiments)
#ifndef PK2_DEN_K
#define PK2_DEN_K 16     // 8-byte arc records leave the registers for 16 gathers in flight per lane
#endif
#ifndef PK2_DEN_CHUNK
#define PK2_DEN_CHUNK 4096
#endif
constexpr int kK = PK2_DEN_K;
static_assert(kK >= 1 && kK <= 32, "the flush mask of a lane has 32 bits");
constexpr int kChunkArcs = PK2_DEN_CHUNK;
constexpr int kMaxRows = 1024;
constexpr int kDenWaves = kChunkArcs / (64 * kK);  // wavefronts working on one chunk (one wave block each)
constexpr int kDenThreads = kDenWaves * 64;        // forward: one chunk per workgroup
constexpr int kDenBwdThreads = 2 * kDenThreads;    // backward: a beta chunk and a gamma chunk per workgroup

struct HostOrdering {
  std::vector<int4> arcs;        // {a, b, prob bits, pi*prob bits}
  std::vector<int2> arcs2;       // {a, prob bits}: the 8-byte records of the state-x kernels, same positions as `arcs`
  std::vector<float> row_leak;   // [sum of nrows]: sum of pi[src]*prob over the arcs of a (chunk, row) piece
  std::vector<int32_t> slot0;    // [n_chunks]: first row_leak slot of the chunk
  std::vector<uint2> meta;       // per lane of each wave block
  std::vector<int32_t> wb_off;   // [n_chunks+1]
  std::vector<int32_t> row0;     // [n_chunks]
  std::vector<int32_t> nrows;    // [n_chunks]
  std::vector<int32_t> atomic;   // [n_chunks]: 0 = whole rows; else a single-row chunk combined by atomics (2 = the row's / group's first piece)
  std::vector<int32_t> real0;    // [n_chunks]: group (real state) of the chunk's first row   (grouped orderings only)
  std::vector<int32_t> nreal;    // [n_chunks]: number of groups the chunk's rows belong to    (grouped orderings only)
  std::vector<int32_t> wb_crow;  // [wave blocks]: chunk-local row of the block's last slot (the row its carry-out belongs to)
  int n_chunks = 0;
};

struct DevOrdering {
  const int4* arcs = nullptr;
  const int2* arcs2 = nullptr;
  const float* row_leak = nullptr;
  const int32_t* slot0 = nullptr;
  const uint2* meta = nullptr;
  const int32_t* wb_off = nullptr;
  const int32_t* row0 = nullptr;
  const int32_t* nrows = nullptr;
  const int32_t* atomic = nullptr;
  const int32_t* real0 = nullptr;
  const int32_t* nreal = nullptr;
  const int32_t* wb_crow = nullptr;
  int n_chunks = 0;
};

// Layout of one arc ordering for the persistent denominator kernel (chain_den_persist.hip): the rows are dealt to the
// kPR workgroups of a team (one per CU of an XCD) in contiguous ranges of whole groups (a group = the rows of one real
// state), every workgroup keeps its <= kPSlots arc slots in registers for the whole call:
//   thread `tid` of rank r owns the consecutive sorted slots tid*kPK .. tid*kPK + kPK-1 of the rank's list,
//   arcs[(r*kPK + j)*kPT + tid] = {gathered index, probability bits}; a row without arcs has one null slot; on the device
//   the probabilities are prob[(r*kPK + j)*kPT + tid] and the indices (< 32768: the LDS table) are packed two to a word,
//   idx2[(r*kPK/2 + j/2)*kPT + tid] = idx(j even) | idx(j odd) << 16 -- 96 registers per thread instead of 128;
//   bit j of ends[r*kPT + tid]: a row ends after the thread's slot j; first_row = rank-local row of its slot 0;
//   wcrow[r*kPW + w]: rank-local row still open after the last slot of wave w (-1: none) -- the wave's carry-out is its.
constexpr int kPR = 32;                 // workgroups of a team = CUs of an XCD
constexpr int kPT = 512;                // threads of a workgroup: 2 waves per SIMD, 256 VGPRs each (the arcs take 128)
constexpr int kPW = kPT / 64;
constexpr int kPK = 64;                 // arc slots per thread
constexpr int kPSlots = kPT * kPK;      // arc slots per workgroup
constexpr int kPSPT = 4;                // states per thread in the row epilogues
constexpr int kPMaxRows = kPSPT * kPT;  // rows, and groups, per workgroup
struct HostPersist {
  bool ok = false;                 // the graph fits (slots, rows per rank)
  int max_rows = 0, max_groups = 0;
  int estep = 1;                   // rows end only at slots j = estep-1 (mod estep) of a thread
  std::vector<int2> arcs;
  std::vector<float> prob;         // device form of `arcs`
  std::vector<uint32_t> idx2;
  std::vector<uint64_t> ends;
  std::vector<int32_t> first_row;
  std::vector<int32_t> wcrow;
  std::vector<int32_t> row_begin;  // [kPR+1] first row of a rank
  std::vector<int32_t> grp_begin;  // [kPR+1] first group (real state) of a rank
  std::vector<float> row_leak;     // [rows]   sum of pi[src]*prob over the arcs of a row
  std::vector<float> row_psum;     // [rows]   sum of prob over the arcs of a row
};
struct DevPersist {
  const float* prob = nullptr;
  const uint32_t* idx2 = nullptr;
  const uint64_t* ends = nullptr;
  const int32_t* first_row = nullptr;
  const int32_t* wcrow = nullptr;
  const int32_t* row_begin = nullptr;
  const int32_t* grp_begin = nullptr;
  const float* row_leak = nullptr;
  const float* row_psum = nullptr;
  int max_rows = 0, max_groups = 0;
  int estep = 1;
};

// Second layout of the persistent kernel (chain_den_persist2.hip): no capacity limits.
//  * The frame's state vector is cut into K table CHUNKS, contiguous index ranges [cbeg[c], cbeg[c+1]) (whole 1 KB LDS-DMA
//    rows).  A vector that fits the LDS table has two, laid out back to back: the kernel starts both DMAs at once and runs the
//    arcs that gather from chunk 0 while chunk 1 is still arriving.  A longer vector goes through two LDS buffers of half the
//    table each, chunk c in buffer c % 2.
//  * Per rank and chunk the arcs form their own row-sorted slot list (rows padded to `estep` slots).  A thread keeps kQ
//    register slots of list 0 ("pass A") and kQ of list 1 ("pass B"); these two lists hold only the rows that have an arc in
//    them, numbered compactly (rmap).  With back-to-back chunks pass B may gather from chunk 0 as well, so arcs move from
//    list 0 to list 1 until the two are equally long.  What does not fit the register slots -- the rest of lists 0 / 1 and the
//    whole lists of chunks >= 2 -- is STREAMED: read again in every frame from the XCD's L2, kSP slots per thread at a time
//    ("piece"); a thread owns `pieces * kSP` consecutive sorted slots of a streamed segment, which holds EVERY row from its
//    first one on (a null slot where a row has no arc).
//  * Pass A and pass B sum into their own compact LDS row arrays (plain stores, one per row), the streamed segments add into
//    a third, row-indexed one; the row epilogue adds them up together with the wave carry-outs of every segment.
constexpr int kQ = kPK / 2;             // register slots per thread of one resident pass
constexpr int kSP = 8;                  // slots per thread of a streamed piece
constexpr int kMaxChunks = 6;
constexpr int kSegs = 2 + kMaxChunks;   // row-sum segments of a frame: pass A, pass B, streamed segment of chunk 0 .. K-1
struct HostPersist2 {
  bool ok = false;
  int estep = 1;
  int K = 0;                             // table chunks
  int R = 0;                             // table entries
  int cbeg[kMaxChunks + 1] = {0};        // first table index of a chunk (multiples of 256 but the last end = R)
  int lds_off[kMaxChunks] = {0};         // where the chunk sits in the LDS table (floats)
  int tfloats = 0;                       // LDS table floats
  bool flexible = false;                 // chunks back to back: list 1 (pass B and its segment) may gather from chunk 0 too
  int max_rows = 0, max_groups = 0;
  int64_t resident_slots = 0, streamed_slots = 0;   // (statistics)
  int max_pieces = 0;                    // most streamed pieces of one rank
  std::vector<float> prob;               // [(r*kPK + j)*kPT + tid]   j < kQ: pass A, else pass B
  std::vector<uint32_t> idx2;            // [(r*kPK/2 + j/2)*kPT + tid] LDS float offsets, two to a word
  std::vector<uint32_t> ends;            // [(r*2 + pass)*kPT + tid]  bit j: a row ends after the pass's slot j
  std::vector<int32_t> first_row;        // [(r*2 + pass)*kPT + tid]  COMPACT row of the thread's first slot
  std::vector<int32_t> uncovered;        // [r*2 + pass] first compact row whose end is not in the resident pass (ncomp: none)
  std::vector<int32_t> ncomp;            // [r*2 + pass] rows of the rank that have a slot in list `pass` (its compact rows)
  std::vector<int16_t> rmap;             // [pass*rows + row] compact index of a row in list `pass` of its rank, -1 = absent
  std::vector<int32_t> pbeg;             // [r*(kMaxChunks+1) + c] pieces of rank r, chunk c: [pbeg[c], pbeg[c+1]) (global ids)
  std::vector<float> sprob;              // [(piece*kPT + tid)*kSP + j]      (a thread reads its piece with two 16-byte loads)
  std::vector<uint32_t> sidx2;           // [(piece*kPT + tid)*kSP/2 + j/2]
  std::vector<uint32_t> sends;           // [piece*kPT + tid] bit j (< kSP): a row ends after slot j of the piece
  std::vector<int32_t> sfirst_row;       // [(r*kMaxChunks + c)*kPT + tid]
  std::vector<int32_t> wcrow;            // [(r*kSegs + seg)*kPW + w] rank-local row open after the last slot of wave w (-1: none)
  std::vector<int32_t> row_begin, grp_begin;
  std::vector<float> row_leak, row_psum;
};
struct DevPersist2 {
  const float* prob = nullptr; const uint32_t* idx2 = nullptr; const uint32_t* ends = nullptr; const int32_t* first_row = nullptr;
  const int32_t* uncovered = nullptr; const int32_t* ncomp = nullptr; const int16_t* rmap = nullptr; const int32_t* pbeg = nullptr;
  int num_rows = 0;
  const float* sprob = nullptr; const uint32_t* sidx2 = nullptr; const uint32_t* sends = nullptr; const int32_t* sfirst_row = nullptr;
  const int32_t* wcrow = nullptr; const int32_t* row_begin = nullptr; const int32_t* grp_begin = nullptr;
  const float* row_leak = nullptr; const float* row_psum = nullptr;
  int estep = 1, K = 0, R = 0;
  int cbeg[kMaxChunks + 1] = {0};
  int lds_off[kMaxChunks] = {0};
};

}  // namespace pk2

struct pk2_den_graph {
  int32_t S = 0, P = 0, start = 0;
  int64_t A = 0;
  std::vector<float> pi;
  std::vector<int32_t> orig_of;   // [S] caller's id of internal state k (identity unless PK2_DEN_ORDER reorders)
  double pi_sum = 0.0;
  pk2::HostOrdering h_fwd, h_bwd, h_gam;  // keyed by dst / src / pdf (general kernels)
  // State-x kernels (chain_den.hip).  One self-loop per state is PEELED off the arc lists: its contribution is a
  // per-state term of the row epilogues (loop_pdf / loop_prob; -1 / 0 = none) -- every state of a Kaldi chain graph
  // has one, carrying the self-loop pdf.  Over the remaining arcs, "virtual states" v = distinct (destination state,
  // pdf) pairs numbered by (state, pdf): the virtual states of state d are voff[d] .. voff[d+1]-1 (a state nobody
  // enters has one with pdf -1).  exp(logit) is a per-VIRTUAL-state factor for any graph; V == S when the arcs
  // entering a state from elsewhere carry one pdf (the forward pdf of a chain graph).  Occupancies are kept per
  // "occupancy state" o: the virtual states of d followed by its peeled loop, ooff[d] .. ooff[d+1]-1, Vo = V + loops.
  int32_t V = 0, Vo = 0;
  std::vector<int32_t> voff, vpdf;        // [S+1], [V]
  std::vector<int32_t> loop_pdf;          // [S]
  std::vector<float> loop_prob;           // [S]
  std::vector<int32_t> ooff, opdf, ovirt; // [S+1], [Vo] pdf, [Vo] first virtual state of the occupancy state's state
  std::vector<int32_t> po_off, po_occ;    // occupancy states grouped by pdf (CSR over P)
  pk2::HostOrdering h_fwdv, h_bwdv;       // rows = virtual dst gathering src | rows = src gathering virtual dst
  pk2::HostPersist h_pfwd, h_pbwd;        // the same two orderings for the persistent kernel
  pk2::DevPersist pfwd, pbwd;
  pk2::HostPersist2 h_p2fwd, h_p2bwd;     // ... and for its second form (chain_den_persist2.hip: chunked table, streamed overflow)
  pk2::DevPersist2 p2fwd, p2bwd;
  int p2_cap = 0;                         // LDS row buffers of that kernel (rows / groups / own virtual states of a rank)
  int p2_rowarrays = 8;                   // LDS arrays of p2_cap floats: 8 with the pdfs the x gather needs, 7 without them
  const int32_t* d_voff = nullptr;
  const int32_t* d_vpdf = nullptr;
  const int32_t* d_loop_pdf = nullptr;
  const float* d_loop_prob = nullptr;
  const int32_t* d_ooff = nullptr;
  const int32_t* d_opdf = nullptr;
  const int32_t* d_ovirt = nullptr;
  const int32_t* d_po_off = nullptr;
  const int32_t* d_po_occ = nullptr;
  // device copies, created lazily on the first compute call
  bool uploaded = false;
  int device = -1;
  pk2::DevOrdering fwd, bwd, gam, fwdv, bwdv;
  float* d_pi = nullptr;
  std::vector<void*> allocs;
};

namespace pk2 {

// Per-call geometry of the denominator computation.
struct DenGeom {
  int NG;      // sequences interleaved per group (4, 2 or 1)
  int G;       // number of groups
  int N;       // sequences
  int Tmax;
  bool persist = false;   // NG = 1, one group per sequence, recursions by the persistent kernel (chain_den_persist.hip)
};

struct DenBuffers {
  float* alpha;   // [G][Tmax+1][S][NG]  alpha (before the leaky term)
  float* alphav;  // [G][Tmax+1][Vo][NG] alpha per occupancy state (state-x path; == alpha when Vo == S)
  float* xl;      // [G][Tmax][S][NG]    exp(logit) of the peeled self-loop's pdf (state-x path)
  float* beta;    // [G][Tmax+1][S][NG]  beta' (before the leaky term); state-x path: [G][Tmax+1][V][2*NG] {btilde', x}
  float* xs;      // [G][Tmax][P][NG]    exp(clamp(logits)), sequences interleaved
  float* gamma;   // [G][Tmax][P][NG]    occupancies
  float* apart;   // [G][Tmax+1][nc_fwd][NG]
  float* bpart;   // [G][Tmax+1][nc_bwd][NG]
  float* asum;    // [G][Tmax+1][NG]
  float* inv_tot; // [G][NG]
  float* den_lp;  // [G*NG]
  float* check;   // [G*NG]
  int32_t* lengths;  // [G*NG] device copy (0 for the padding sequences)
  float* csum;    // [G][Tmax+1][2][NG]  {cu[t], sum_k pi[k] btilde'[t,k] / cu[t]}   (state-x path, chain_den.hip)
  float* kscale;  // [G][Tmax+1][NG]  beta[t] = kscale[t] * betahat[t] (state-x path)
  float* xv;      // [G][Tmax][V]     exp(logit) per virtual state, compact (persistent kernel only)
};

int den_choose_ng(const pk2_den_graph* g);
bool den_use_sx(const pk2_den_graph* g);
size_t den_workspace(const pk2_den_graph* g, int N, int Tmax, DenGeom* geom, DenBuffers* buf,
                     void* base);
struct NumDeferred;
int den_upload(pk2_den_graph* g);
// Runs exp-transpose, T forward steps, finalize, T backward steps.  Leaves
// gamma / den_lp / check in `buf`.
// `zero` (optional): gradient rows [N][Tmax][P] of the caller that must hold zeros before the numerator adds into them --
// zeroed by the launch that prepares the recursions (round 6: one launch less per step) or by one of their own.
struct DenZeroRows { float* grad = nullptr; int64_t gss = 0, gfs = 0; int N = 0; };
int den_compute(pk2_den_graph* g, const float* logits, int64_t seq_stride, int64_t frame_stride,
                const int32_t* lengths_host, const DenGeom& geom, const DenBuffers& buf,
                float leaky, hipStream_t stream, const NumDeferred* tail = nullptr, const DenZeroRows* zero = nullptr);

// Internal side stream (+ fork/join events) paired with a caller stream: the numerator runs there while
// the denominator occupies the caller's stream.
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
int get_side_stream(hipStream_t main, SideStream** out);

// Numerator.
// seqinfo[n] = {frame_off base index, length, state count, final lo, final hi, -, -, -}
struct NumParams {
  const int32_t* arc_src; const int32_t* arc_dst; const int32_t* arc_pdf; const float* arc_w;
  const int32_t* frame_off; const int32_t* final_state; const float* final_w;
  const int32_t* seqinfo;
  const float* logits; int64_t seq_stride, frame_stride;
  float* score; float* frame_max; float* num_lp;
  float* grad; int64_t gseq_stride, gframe_stride;
  float scale;
};

// A numerator whose forward-backward launch has been handed to the denominator (see chain_num.h).
struct NumDeferred { NumParams p; size_t lds = 0; int N = 0; bool valid = false; bool stage = false; };
int num_launch_deferred(const NumDeferred& d, hipStream_t stream);

struct NumBuffers {
  float* score;     // [total_arcs]
  float* frame_max; // [sum lengths]
  int32_t* seqinfo; // [N][8]
  float* num_lp;    // [N]
};
size_t num_workspace(int N, int64_t total_arcs, int64_t total_frames, NumBuffers* buf, void* base);
// Adds scale * posterior into grad (which must already hold zeros) and writes num_lp.
int num_compute(const pk2_num_batch* nb, const float* logits, int64_t seq_stride,
                int64_t frame_stride, const int32_t* lengths_host, int N, float scale, float* grad,
                int64_t grad_seq_stride, int64_t grad_frame_stride, const NumBuffers& buf,
                hipStream_t stream, NumDeferred* defer = nullptr);

}  // namespace pk2
