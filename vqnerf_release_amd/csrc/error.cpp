// Thread-local error string + version for the C ABI (include/vqnerf_hip.h).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vqn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vqn_last_error(void) { return g_err; }
extern "C" int vqn_version(void) { return 100; /* 0.1.0 */ }
