// Error plumbing and version entry points of the C ABI (include/unidisc_hip.h).
#include "common.h"
#include "../../include/unidisc_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_err[512] = "";
}

void udm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* udm_last_error(void) { return g_err; }
extern "C" int udm_abi_version(void) { return UDM_ABI_VERSION; }
