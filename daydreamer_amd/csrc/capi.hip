// Error reporting and version for libdaydreamer_hip.so.
#include "dd_common.h"
#include "../../include/daydreamer_hip.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void dd_set_error(const char* where, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
}

void dd_set_error_msg(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" int dd_version(void) { return DD_ABI_VERSION; }
extern "C" const char* dd_last_error(void) { return g_err; }
