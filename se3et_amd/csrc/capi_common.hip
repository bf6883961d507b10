// Error reporting and version entry points of the C ABI.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void se3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* se3_last_error(void) { return g_err; }
extern "C" const char* se3_version(void) { return "se3et_hip 0.1 (gfx950)"; }
