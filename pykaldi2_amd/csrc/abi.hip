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

extern "C" int pk2_debug_where(int32_t* out, int32_t blocks, void* stream) {
  PK2_REQUIRE(out && blocks > 0, "debug_where: bad args");
  hipLaunchKernelGGL(pk2::debug_where_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}
