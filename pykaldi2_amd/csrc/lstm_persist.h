// Persistent small-batch LSTM recurrence (lstm_persist.hip): internal interface used by lstm.hip.
#pragma once
#include "common.h"

namespace pk2 {

struct PersistCtl {
  unsigned reg[8];     // workgroups registered per XCD (arrival order = role)
  unsigned abort;      // set when a poll timed out: every workgroup leaves, outputs keep their NaN sentinels
  unsigned pad[7];
};

bool lstm_persist_wanted(int B, int H, int D);
// Runs the whole forward recurrence of a layer in one launch.  *ran = false when the persistent path is unusable on this
// device (checked once): the caller then uses the step kernels.
int lstm_fwd_persist_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                            float* gates, float* cells, hipStream_t stream, bool* ran);
// Backward recurrence; `mailboxes`: device scratch of lstm_bwd_persist_mailbox_floats(D) floats.
size_t lstm_bwd_persist_mailbox_floats(int D);
int lstm_bwd_persist_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                            int D, float* dgx, float* mailboxes, hipStream_t stream, bool* ran);
int lstm_persist_status(unsigned* abort_flag);

// One (sequence, direction) per XCD (lstm_persist_seq.hip): same tensors, same layouts; preferred when available.
bool lstm_seq_wanted(int B, int H, int D);
int lstm_fwd_seq_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                        float* gates, float* cells, hipStream_t stream, bool* ran);
// dbias_ih / dbias_hh (may be null): [D][4H] accumulators (+=) of the bias gradient, filled by the round-4 kernel inside its
// launch; *bias_done says whether it was.
int lstm_bwd_seq_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                        int D, float* dgx, hipStream_t stream, bool* ran, float* dbias_ih = nullptr, float* dbias_hh = nullptr,
                        bool* bias_done = nullptr);
int lstm_seq_status(unsigned* abort_flag);

// Large batches (B >= 32, H = 512; lstm_persist_big.hip): a (direction, 64-row batch tile) task per XCD team, the rank's
// W_hh slice resident in LDS, one launch per layer.
bool lstm_big_wanted(int B, int H, int D);
int lstm_fwd_big_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                        float* gates, float* cells, hipStream_t stream, bool* ran);
int lstm_bwd_big_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                        int D, float* dgx, hipStream_t stream, bool* ran);
int lstm_big_status(unsigned* abort_flag);

}  // namespace pk2
