// Error reporting + version for libcatseg_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "catseg.h"

static thread_local char g_err[512] = "";

void catseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* catseg_last_error(void) { return g_err; }
extern "C" int catseg_version(void) { return 1; }
