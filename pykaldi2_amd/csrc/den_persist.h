// Persistent denominator recursion (chain_den_persist.hip): internal interface used by chain_den.hip.
#pragma once
#include "den_kernels.h"

namespace pk2 {

// True when the alpha / beta recursions of this graph can run as one launch: the persistent layouts exist, a state
// vector fits the LDS of a CU next to the row buffers, the device is an 8 x 32-CU part, PK2_DEN_PERSIST != 0 and an
// earlier launch has not failed its verification.
bool den_persist_wanted(const pk2_den_graph* g, int N);
size_t den_persist_lds_bytes(const pk2_den_graph* g);
// The layouts exist and a state vector fits the LDS table and a thread's registers (a property of the graph alone).
bool den_persist_fits(const pk2_den_graph* g);
constexpr size_t kDenPersistMaxLds = 160 * 1024 - 256;

// Runs the forward and backward recursions of all N sequences (NG = 1 layouts: one group per sequence; `p` as the
// launch-per-frame kernels would receive it, `xv` = [G][Tmax][V] exp(logit) per virtual state).  Leaves alpha, alphav,
// apart, asum, the btilde' slots of `beta` and bpart where den_step_sx<1> leaves them, with kPR partial sums per frame
// (bpart: shares whose sum over the ranks is the frame's sum, not sums over a rank's own states).  *ran = false when the device failed the first-use verification (the caller then replays the frame launches).
int den_persist_launch(pk2_den_graph* g, const DenParams& p, const float* xv, const int32_t* lengths_host, int N,
                       hipStream_t stream, bool* ran);
// After den_finalize: a launch in which a poll timed out turns den_lp into NaN instead of passing for a result.
void den_persist_check_launch(float* den_lp, int N, hipStream_t stream);

// ---- second form (chain_den_persist2.hip): chunked LDS table, two resident passes, streamed overflow -- any graph whose
// ranks hold at most kPMaxRows rows and whose vector needs at most kMaxChunks table chunks.
// LDS of a workgroup: the table, seven row arrays of `cap` floats (sums of pass A, of pass B, of the streamed segments, x
// of own rows, two per-row constants, and the two 16-bit compact-row maps), and the small fixed part (reduction scratch,
// the segments' wave carries and their rows, flags).
constexpr int kP2RowArrays = 8;      // (the eighth: the pdfs of own rows and own states as shorts, round 4 -- left out, with
                                     // the x gather, for a graph it would cost a table chunk: pk2_den_graph::p2_rowarrays)
#ifndef PK2_DP2_POLLCOPY
#define PK2_DP2_POLLCOPY 0          // chain_den_persist2.hip: the copy of chunk 0 following the poll (round 5: measured, off)
#endif
constexpr int kP2NeedRows = PK2_DP2_POLLCOPY ? 128 : 0;     // 1 KB rows of table chunk 0: which ranks' slices each one holds
constexpr int kP2FixedFloats = 2 * kPW + 4 + 2 * kSegs * kPW + 16 + kP2NeedRows;
inline size_t den_persist2_lds_bytes(int tfloats, int cap, int arrays = kP2RowArrays) {
  return ((size_t)tfloats + arrays * (size_t)cap + kP2FixedFloats) * sizeof(float);
}
bool den_persist2_fits(const pk2_den_graph* g);
// `tail`: the minibatch's deferred numerator forward-backward (may be null); *num_ran = true when it rode in this launch
// (as tasks behind the recursions) and must not be launched again.
struct NumDeferred;
int den_persist2_launch(pk2_den_graph* g, const DenParams& p, const float* xv, const int32_t* lengths_host, int N,
                        hipStream_t stream, bool* ran, const NumDeferred* tail = nullptr, bool* num_ran = nullptr,
                        bool xgather = false);      // xgather: `xv` is exp(logits) [G][Tmax][P], gathered by pdf
void den_persist2_check_launch(float* den_lp, int N, hipStream_t stream);
// Round 6: the check as a phase of den_tail1 (chain_den.hip) instead of a launch of its own.  Fills `ck` from the scratch of the
// persistent launch that has just run on `stream` (false: there was none) and books the control block as zeroed.
struct DenTailCheck { unsigned* ctl; int ctl_words; int ntasks; unsigned* abort_word; unsigned* done_word; unsigned* count_word;
                      unsigned* guard_dev; unsigned* guard_host; };
bool den_persist2_tail_check(hipStream_t stream, DenTailCheck* ck);
// Which recursion kernel a call of N sequences takes: 0 = the launch-per-frame kernels, 1 = den_persist_kernel (everything
// resident: graphs up to ~1.05 M arc slots and ~36 k states), 2 = den_persist2_kernel.  PK2_DEN_PERSIST = 0 | 1 | 2 forces one
// (a forced form that does not fit the graph falls back to the frame kernels).
int den_persist_version(const pk2_den_graph* g, int N);

}  // namespace pk2
