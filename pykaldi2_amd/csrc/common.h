// Shared helpers for libpk2hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "pk2hip.h"

namespace pk2 {

void set_error(const char* fmt, ...);

#define PK2_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      pk2::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return PK2_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define PK2_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      pk2::set_error(__VA_ARGS__);    \
      return PK2_ERR_INVALID;         \
    }                                 \
  } while (0)

#define PK2_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      pk2::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,            \
                     hipGetErrorString(e_));                                      \
      return PK2_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Host-side state of the library is kept PER DEVICE: "verified on this device", "this kernel's LDS attribute is
// set", the CU count and the per-stream scratch blocks belong to the device that is current when they are used -- a process
// that drives a second GPU must verify, configure and allocate there again (the null stream is the same handle on every
// device, so a stream alone does not identify one).
inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
template <typename T>
struct PerDevice {
  static constexpr int kMax = 64;
  T v[kMax];
  explicit PerDevice(const T& init) { for (int i = 0; i < kMax; ++i) v[i] = init; }
  T& ref() { return v[current_device() & (kMax - 1)]; }
};
struct DevStream {
  int dev; hipStream_t stream;
  bool operator<(const DevStream& o) const { return dev != o.dev ? dev < o.dev : stream < o.stream; }
};
inline DevStream dev_stream(hipStream_t s) { return DevStream{current_device(), s}; }

// Carves aligned sub-buffers out of a caller-owned workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* r = reinterpret_cast<T*>(base ? base + off : nullptr);
    off += count * sizeof(T);
    return r;
  }
  size_t bytes() const { return align_up(off, 256); }
};

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace pk2
