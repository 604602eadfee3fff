// Host-side ABI plumbing of libpp_hip.so: version + last-error text.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "pp_hip.h"

static thread_local char g_err[256] = "";

void pp_set_last_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

#ifndef PP_BUILD_ID
#define PP_BUILD_ID "unknown"
#endif
extern "C" int pp_abi_version(void) { return PP_ABI_VERSION; }
extern "C" const char* pp_build_id(void) { return PP_BUILD_ID; }
extern "C" const char* pp_last_error(void) { return g_err; }
