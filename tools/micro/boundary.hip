// Round 6: what does a dependent kernel boundary inside a replayed hipGraph cost, by launch configuration?  The step is a
// serial chain of ~200 kernels; profiles/r05_step_timeline.txt shows 4-6 us for a ONE-block kernel inside the graph, the
// guide's "boundary" row says 1.45-1.9 us.  Variants: block size, dynamic LDS, kernel-argument bytes, what the kernel writes.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/boundary.hip -o tools/micro/boundary && tools/micro/boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Big { int v[100]; float* p; };   // ~408 bytes of kernel arguments, like PPGemmArgs

__global__ void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void __launch_bounds__(512) k_lds(float* p) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) lds[0] = p[0];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = lds[0] + 1.f;
}
__global__ void __launch_bounds__(512) k_big(const Big b) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) lds[0] = b.p[0] + b.v[7];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = lds[0] + 1.f;
}
// writes `bytes` per launch (dirty lines the boundary must write back), 256 workgroups
__global__ void __launch_bounds__(512) k_write(float4* dst, int n16) {
  extern __shared__ float lds[];
  for (int i = blockIdx.x * 512 + threadIdx.x; i < n16; i += gridDim.x * 512) dst[i] = float4{1.f, 2.f, 3.f, 4.f};
  if (lds[0] == 77.f) dst[0].x = 1.f;
}

template <class F>
static void run(const char* name, int n, F launch) {
  hipStream_t st;
  hipStreamCreate(&st);
  launch(st);
  hipStreamSynchronize(st);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(st);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  std::vector<float> ts;
  for (int it = 0; it < 9; ++it) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1e3f / n);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-86s %6.2f us per kernel (median of 9 replays of %d nodes)\n", name, ts[4], n);
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
}

int main() {
  float* p;
  float4* big;
  hipMalloc(&p, 4096);
  hipMemset(p, 0, 4096);
  hipMalloc(&big, 64 << 20);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_big), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_write), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  Big b;
  for (int i = 0; i < 100; ++i) b.v[i] = i;
  b.p = p;
  const int N = 200;
  run("1 workgroup x 64 threads, no LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p); });
  run("256 workgroups x 64 threads, no LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_small, dim3(256), dim3(64), 0, s, p); });
  run("256 x 512 threads, no LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 64, s, p); });
  run("256 x 512 threads, 64 KB dynamic LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 64 * 1024, s, p); });
  run("256 x 512 threads, 160 KB dynamic LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 160 * 1024, s, p); });
  run("256 x 512 threads, 160 KB LDS, 408-byte kernel arguments", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 160 * 1024, s, b); });
  run("1 x 512 threads, 160 KB LDS, 408-byte kernel arguments", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(1), dim3(512), 160 * 1024, s, b); });
  run("1024 x 512 threads, 160 KB LDS (four residency rounds)", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(1024), dim3(512), 160 * 1024, s, p); });
  for (int mb : {1, 4, 16, 42}) {
    char nm[128];
    snprintf(nm, sizeof nm, "256 x 512 threads, 160 KB LDS, writes %d MB per launch", mb);
    const int n16 = mb * (1 << 20) / 16;
    run(nm, 50, [&](hipStream_t s) { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), 160 * 1024, s, big, n16); });
  }
  return 0;
}
