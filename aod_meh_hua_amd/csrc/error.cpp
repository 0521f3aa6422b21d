#include <stdarg.h>
#include <stdio.h>
#include "../../include/aod_hip.h"
thread_local char g_aod_err[512] = "";
int aod_set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_aod_err, sizeof(g_aod_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* aod_last_error(void) { return g_aod_err; }
// 2: aod_conv_desc_t.x3 (round 4) -- the field fills what was padding in front of seg[]; descriptors must be zero-initialised
extern "C" int aod_version(void) { return 2; }
