// Error reporting and version of the C ABI (include/pk2hip.h).
#include "common.h"

namespace pk2 {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pk2

extern "C" const char* pk2_last_error(void) { return pk2::g_err; }
extern "C" int pk2_version(void) { return 1; }

// ---- side streams confined to a part of the chip (include/pk2hip.h) ----
extern "C" int pk2_stream_create_cu_mask(int32_t cus_per_xcd, void** stream_out) {
  PK2_REQUIRE(stream_out && cus_per_xcd >= 1 && cus_per_xcd <= 32, "stream_create_cu_mask: bad args");
  // bit i of the mask = CU (i / 8) of XCD (i % 8): the first 8 * k bits are the first k CUs of every XCD
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8 * cus_per_xcd; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  PK2_HIP(hipExtStreamCreateWithCUMask(&st, 8, mask));
  *stream_out = st;
  return PK2_OK;
}

extern "C" int pk2_stream_destroy(void* stream) {
  if (stream) PK2_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return PK2_OK;
}

namespace pk2 {
__global__ void debug_where_kernel(int32_t* out) {
  if (threadIdx.x != 0) return;
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  out[3 * blockIdx.x + 0] = (int32_t)(xcc & 0xf);
  out[3 * blockIdx.x + 1] = (int32_t)((hw >> 13) & 0x7);     // SE_ID
  out[3 * blockIdx.x + 2] = (int32_t)((hw >> 8) & 0xf);      // CU_ID
  // (a little work so that the workgroups of one launch spread over the CUs instead of reusing the first one)
  for (volatile int i = 0; i < 20000; ++i) { }
}
}  // namespace pk2

// What a ring / direct all-reduce kernel of the collective library does to the chip while a step runs, without a second GPU:
// `blocks` workgroups of 256 threads stream dst[i] += src[i] over the bucket, `passes` times (pre-pricing the
// co-residency of the persistent kernels with the exchange; tools/gpu_r05_peer.sh).  With src = zeros the bucket keeps its values.
namespace pk2 {
__global__ void __launch_bounds__(256) debug_peer_reduce_kernel(float* dst, const float* src, int64_t n, int passes) {
  const int64_t n4 = n / 4;
  float4* d4 = reinterpret_cast<float4*>(dst);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int it = 0; it < passes; ++it) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      float4 a = d4[i];
      const float4 b = s4[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      d4[i] = a;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] += src[i];
  }
}
}  // namespace pk2

extern "C" int pk2_debug_peer_reduce(float* dst, const float* src, int64_t n, int32_t blocks, int32_t passes, void* stream) {
  PK2_REQUIRE(dst && src && n >= 0 && blocks > 0 && passes > 0, "debug_peer_reduce: bad args");
  PK2_REQUIRE(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0, "debug_peer_reduce: 16-byte alignment");
  if (n == 0) return PK2_OK;
  hipLaunchKernelGGL(pk2::debug_peer_reduce_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, n, passes);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_debug_where(int32_t* out, int32_t blocks, void* stream) {
  PK2_REQUIRE(out && blocks > 0, "debug_where: bad args");
  hipLaunchKernelGGL(pk2::debug_where_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
