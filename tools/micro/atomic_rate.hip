// Can GroupNorm statistics be accumulated by the producing GEMM's epilogue with 64-bit integer (fixed-point,
// order-independent => deterministic) device-scope atomics?  Pattern of a 64x64-latent conv epilogue: 512 blocks x 2
// passes, each pass adds (sum, sumsq) for 16 groups of its batch item: 8 items x 32 groups x 2 = 512 addresses, 64-128
// adds per address, spread over the kernel's lifetime.  Measures the kernel time with and without the atomics.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned long long* acc, float* sink, int do_atomic, int spin) {
  const int tid = threadIdx.x, blk = blockIdx.x;
  float v = tid * 1e-3f;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;     // stands in for the block's main loop
    if (do_atomic && tid < 32) {
      const int b = blk / 64;                                  // 64 blocks per batch item
      const int g = tid & 15, w = tid >> 4;                    // 16 groups x (sum, sumsq)
      const unsigned long long q = (unsigned long long)(long long)(v * 1048576.0f);
      __hip_atomic_fetch_add(acc + ((b * 32 + (blk & 1) * 16 + g) * 2 + w), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (v == 123.f) sink[0] = v;
}

int main() {
  unsigned long long* acc;
  float* sink;
  hipMalloc(&acc, 8 * 32 * 2 * 8);
  hipMalloc(&sink, 4);
  hipMemset(acc, 0, 8 * 32 * 2 * 8);
  for (int spin : {2000, 20000}) {
    for (int at = 0; at < 2; ++at) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, acc, sink, at, spin);
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, acc, sink, at, spin);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("spin %5d  atomics %d : %.2f us per launch\n", spin, at, ms * 1e3 / 20);
    }
  }
  unsigned long long h[4];
  hipMemcpy(h, acc, 32, hipMemcpyDeviceToHost);
  printf("acc[0..1] = %llu %llu\n", h[0], h[1]);
  return 0;
}
