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
