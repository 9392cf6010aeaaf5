// Process-wide guard raised by the check kernels of the persistent launches (persist_guard.h); no reference counterpart:
// cuDNN / Kaldi kernels cannot time out, the persistent kernels here can (1 s polls), and a time-out must stop training
// instead of reaching the weights.
#include <map>
#include <mutex>

#include "persist_guard.h"

namespace pk2 {

static std::mutex g_guard_mu;
static std::map<int, PersistGuard> g_guards;

int persist_guard(PersistGuard* out) {
  int dev = 0;
  PK2_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_guard_mu);
  PersistGuard& g = g_guards[dev];
  if (!g.dev) {
    unsigned* h = nullptr;
    PK2_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(unsigned), hipHostMallocMapped));
    *h = 0u;
    unsigned* hd = nullptr;
    PK2_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0));
    unsigned* d = nullptr;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned)));
    PK2_HIP(hipMemset(d, 0, sizeof(unsigned)));
    g.host = h; g.host_dev = hd; g.dev = d;
  }
  *out = g;
  return PK2_OK;
}

__global__ void persist_guard_raise_kernel(unsigned* dev, unsigned* host_dev) { persist_guard_raise(dev, host_dev); }

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
