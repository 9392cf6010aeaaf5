// Process-wide guard raised by the check kernels of the persistent launches (persist_guard.h); no reference counterpart:
// cuDNN / Kaldi kernels cannot time out, the persistent kernels here can (1 s polls), and a time-out must stop training
// instead of reaching the weights.
#include <map>
#include <mutex>
#include <vector>

#include "persist_guard.h"

namespace pk2 {

static std::mutex g_guard_mu;
static std::map<int, PersistGuard> g_guards;

// Subsystems whose scratch memory an aborted launch leaves in an undefined state register a callback that marks it dirty;
// pk2_persist_guard_clear ("the caller has dealt with the failure") runs them.
static std::vector<void (*)()>& clear_hooks() { static std::vector<void (*)()> v; return v; }
void persist_guard_on_clear(void (*fn)()) { std::lock_guard<std::mutex> lock(g_guard_mu); clear_hooks().push_back(fn); }
static void persist_guard_cleared() {
  std::vector<void (*)()> hooks;
  { std::lock_guard<std::mutex> lock(g_guard_mu); hooks = clear_hooks(); }
  for (auto fn : hooks) fn();
}

int persist_guard(PersistGuard* out) {
  int dev = 0;
  PK2_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_guard_mu);
  PersistGuard& g = g_guards[dev];
  if (!g.dev) {
    unsigned* h = nullptr;
    PK2_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), (1 + kGuardRing) * sizeof(unsigned), hipHostMallocMapped));
    for (unsigned i = 0; i <= kGuardRing; ++i) h[i] = 0u;
    unsigned* hd = nullptr;
    PK2_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0));
    unsigned* d = nullptr;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned)));
    PK2_HIP(hipMemset(d, 0, sizeof(unsigned)));
    g.host = h; g.host_dev = hd; g.dev = d;
    g.ring = h + 1; g.ring_dev = hd + 1;
  }
  *out = g;
  return PK2_OK;
}

__global__ void persist_guard_raise_kernel(unsigned* dev, unsigned* host_dev) { persist_guard_raise(dev, host_dev); }

__global__ void persist_guard_export_kernel(const unsigned* dev, float* slot) { *slot = *dev ? 1.f : 0.f; }

// After the ranks have combined their slots (max / sum): any rank's raised guard (or a NaN that got into the slot) raises
// this rank's guard too -- its optimiser kernels, queued behind this one, then leave the weights alone like the failing
// rank's -- and the verdict of step `stamp` is published for the host.
__global__ void persist_guard_import_kernel(const float* slot, unsigned stamp, unsigned* dev, unsigned* host_dev, unsigned* ring_dev) {
  const float v = *slot;
  const bool raised = !(v == 0.f);
  if (raised) persist_guard_raise(dev, host_dev);
  __hip_atomic_store(ring_dev + stamp % kGuardRing, (stamp << 1) | (raised ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace pk2

using namespace pk2;

// 1 in *raised when a check kernel of a persistent launch on the current device has raised the guard since start-up (or
// the last clear).  Does not synchronise: the word lives in host-mapped memory.
extern "C" int pk2_persist_guard_status(uint32_t* raised) {
  PK2_REQUIRE(raised, "persist_guard_status: null pointer");
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  *raised = *g.host ? 1u : 0u;
  return PK2_OK;
}

// Lowers the guard (synchronises the device).  For tests and for a caller that has dealt with the failure.
extern "C" int pk2_persist_guard_clear(void) {
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  PK2_HIP(hipDeviceSynchronize());
  PK2_HIP(hipMemset(g.dev, 0, sizeof(unsigned)));
  *g.host = 0u;
  for (unsigned i = 0; i < kGuardRing; ++i) g.ring[i] &= ~1u;
  persist_guard_cleared();
  return PK2_OK;
}

// Test hook: raises the guard from the device, the way a check kernel does.
extern "C" int pk2_persist_guard_raise(void* stream_) {
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  hipLaunchKernelGGL(persist_guard_raise_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), g.dev, g.host_dev);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Collective guard (several ranks; pykaldi2_amd/hvd.py): *slot = 1.0f when this device's guard is raised, else 0.0f.  The
// caller combines the slots of all ranks (all-reduce, max or sum) on the same stream and hands the result to
// pk2_persist_guard_import.
extern "C" int pk2_persist_guard_export(float* slot, void* stream_) {
  PK2_REQUIRE(slot, "persist_guard_export: null pointer");
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  hipLaunchKernelGGL(persist_guard_export_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), g.dev, slot);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Raises this device's guard when the combined slot is non-zero and publishes the verdict of step `stamp` (>= 1) in
// host-mapped memory.
extern "C" int pk2_persist_guard_import(const float* slot, uint32_t stamp, void* stream_) {
  PK2_REQUIRE(slot && stamp >= 1 && stamp < (1u << 30), "persist_guard_import: bad args");
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  hipLaunchKernelGGL(persist_guard_import_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), slot, stamp, g.dev, g.host_dev, g.ring_dev);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Verdict of step `stamp` without synchronising: *ready = 1 once pk2_persist_guard_import(stamp) has run on the device,
// *raised = its verdict.  (A ring of kGuardRing steps: ask for a step at most kGuardRing - 1 imports back.)
extern "C" int pk2_persist_guard_verdict(uint32_t stamp, uint32_t* ready, uint32_t* raised) {
  PK2_REQUIRE(ready && raised && stamp >= 1, "persist_guard_verdict: bad args");
  PersistGuard g;
  int rc = persist_guard(&g);
  if (rc) return rc;
  const unsigned w = g.ring[stamp % kGuardRing];
  *ready = (w >> 1) == stamp ? 1u : 0u;
  *raised = *ready ? (w & 1u) : 0u;
  return PK2_OK;
}
