// What happens after a persistent kernel gave up: every persistent launch is followed by a one-workgroup check
// kernel that, when a poll of the launch timed out (or a task never finished), poisons the launch's output with NaN and
// raises a flag of its own subsystem.  That alone leaves the damage to the caller: a CE step has no Kaldi-style NaN guard,
// and Adam would write the NaN into every weight.  So the check kernels also raise this process-wide guard:
//  * a word in device memory that the optimiser kernels (optim.hip) read -- while it is set they leave the parameters and
//    their moments untouched, whatever the gradients hold;
//  * a word in host-mapped memory that the host reads WITHOUT synchronising (pk2_persist_guard_status): the Python
//    optimisers raise at the first step() that sees it.
// One guard per device (a process that drives two GPUs gets two).
#pragma once
#include "common.h"

namespace pk2 {

struct PersistGuard {
  unsigned* dev = nullptr;        // device memory: read by kernels
  unsigned* host_dev = nullptr;   // device address of the host-mapped word: written by the check kernels
  volatile unsigned* host = nullptr;
  // Verdicts of the COLLECTIVE guard (several ranks, round 5): ring of kGuardRing host-mapped words, word[stamp % ring] =
  // (stamp << 1) | raised, written by pk2_persist_guard_import after the ranks exchanged their guard words, so that every
  // rank reads the same verdict for the same step and stops at the same step (pykaldi2_amd/hvd.py).
  unsigned* ring_dev = nullptr;
  volatile unsigned* ring = nullptr;
};
constexpr unsigned kGuardRing = 8;

// The guard of the current device (created on first use); PK2_OK or an error code.
int persist_guard(PersistGuard* out);
// `fn` runs whenever pk2_persist_guard_clear lowers a guard: scratch state an aborted launch may have left behind is
// marked for re-initialisation (lstm_persist_seq.hip: the backward mailboxes).
void persist_guard_on_clear(void (*fn)());

__device__ __forceinline__ void persist_guard_raise(unsigned* dev, unsigned* host_dev) {
  if (dev) *dev = 1u;
  if (host_dev) __hip_atomic_store(host_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace pk2
