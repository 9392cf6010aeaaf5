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

}  // namespace pk2
