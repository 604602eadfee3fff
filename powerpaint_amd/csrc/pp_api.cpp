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

// compute units of the current device (cached per device; 256 if the query fails: the part this library is written for)
int pp_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> g(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev] = n;
  return n;
}

// ---- workgroup -> XCD placement probe (ABI v21; what the in-kernel split-K combine's co-location rests on, gemm_combine.h)
__global__ void __launch_bounds__(64) pp_xcc_probe_kernel(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.y * gridDim.x + blockIdx.x] = v & 7u;
  }
}

extern "C" int pp_xcd_placement_ok(void) {
  static std::mutex mu;
  static std::map<int, int> verdict;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> g(mu);
  auto it = verdict.find(dev);
  if (it != verdict.end()) return it->second;
  // two grid shapes of the split-K launches (tiles % 8 == 0): residue (linear id % 8) -> one XCC each, all eight distinct
  int ok = 1;
  const int shapes[2][2] = {{64, 4}, {32, 8}};
  unsigned* d = nullptr;
  if (hipMalloc(&d, 256 * sizeof(unsigned)) != hipSuccess) return 0;
  for (int sh = 0; sh < 2 && ok; ++sh) {
    const int X = shapes[sh][0], Y = shapes[sh][1];
    unsigned h[256];
    if (hipMemset(d, 0xff, sizeof(h)) != hipSuccess) { ok = 0; break; }
    hipLaunchKernelGGL(pp_xcc_probe_kernel, dim3(X, Y), dim3(64), 0, 0, d);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { ok = 0; break; }
    int map[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (int f = 0; f < X * Y; ++f) {
      if (h[f] > 7u) ok = 0;
      else if (map[f & 7] < 0) map[f & 7] = (int)h[f];
      else if (map[f & 7] != (int)h[f]) ok = 0;
    }
    for (int i = 0; i < 8 && ok; ++i)
      for (int j = i + 1; j < 8; ++j)
        if (map[i] == map[j]) ok = 0;
  }
  (void)hipFree(d);
  verdict[dev] = ok;
  return ok;
}

#ifndef PP_BUILD_ID
#define PP_BUILD_ID "unknown"
#endif
extern "C" int pp_abi_version(void) { return PP_ABI_VERSION; }
extern "C" const char* pp_build_id(void) { return PP_BUILD_ID; }
extern "C" const char* pp_last_error(void) { return g_err; }
