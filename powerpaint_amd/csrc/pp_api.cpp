// Host-side ABI plumbing of libpp_hip.so: version + last-error text.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <utility>

#include "pp_hip.h"

static thread_local char g_err[256] = "";

void pp_set_last_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(function, device) property: the largest size granted so far is
// remembered per (kernel, device), so a process that drives a second device never launches there with the default limit.
int pp_func_lds(const void* kern, int bytes, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> granted;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> g(mu);
  int& have = granted[std::make_pair(kern, dev)];
  if (bytes <= have) return PP_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    pp_set_last_error(what, hipGetLastError());
    return PP_ERR_LAUNCH;
  }
  have = bytes;
  return PP_OK;
}

#ifndef PP_BUILD_ID
#define PP_BUILD_ID "unknown"
#endif
extern "C" int pp_abi_version(void) { return PP_ABI_VERSION; }
extern "C" const char* pp_build_id(void) { return PP_BUILD_ID; }
extern "C" const char* pp_last_error(void) { return g_err; }
