// Error reporting and identification entry points of libmtam_hip.so.
#include <stdarg.h>
#include "common.h"

namespace {
thread_local char g_last_error[512] = "";
}

void mtam_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" const char *mtam_last_error(void) { return g_last_error; }
extern "C" int mtam_version(void) { return 1000 * 0 + 1; }
extern "C" const char *mtam_arch(void) { return "gfx950"; }
