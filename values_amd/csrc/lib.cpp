// libvalues_amd.so: version + thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/values_amd.h"

static thread_local char g_err[512] = "";

void vx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int vx_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* vx_last_error_string(void) { return g_err; }
