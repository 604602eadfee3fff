// Does a consumer kernel find the producer kernel's output in its XCD's L2 ACROSS a kernel boundary?
// Block b runs on XCD b % 8 (observed).  writer: block b writes chunk b.  reader: block b reads chunk (b + shift) % G:
// shift 0 = the chunk its own XCD wrote, shift 1 = the next XCD's, shift 8 = another block's of the SAME XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_reuse.hip -o tools/micro/xcd_reuse && tools/micro/xcd_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__global__ void __launch_bounds__(256) writer(u4* buf, int chunk16, unsigned v) {
  u4* p = buf + (size_t)blockIdx.x * chunk16;
  for (int i = threadIdx.x; i < chunk16; i += 256) p[i] = u4{v, v + 1, v + 2, (unsigned)i};
}
__global__ void __launch_bounds__(256) reader(const u4* buf, int chunk16, int shift, unsigned* out) {
  const int c = (blockIdx.x + shift) % gridDim.x;
  const u4* p = buf + (size_t)c * chunk16;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < chunk16; i += 256) { const u4 v = p[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main() {
  for (int total_mb : {8, 16, 24, 48, 128}) {
    const int G = 2048;
    const size_t bytes = (size_t)total_mb << 20;
    const int chunk16 = (int)(bytes / G / 16);
    u4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, G * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shift : {0, 8, 1, 4, 1027}) {
      std::vector<float> ts;
      for (int it = 0; it < 15; ++it) {
        hipLaunchKernelGGL(writer, dim3(G), dim3(256), 0, 0, buf, chunk16, (unsigned)it);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, buf, chunk16, shift, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      printf("total %3d MB (%.1f MB per XCD)  reader shift %4d: %7.2f us  -> %6.2f TB/s\n", total_mb, total_mb / 8.0, shift,
             ts[7], bytes / (ts[7] * 1e-6) / 1e12);
    }
    hipFree(buf); hipFree(out);
  }
  return 0;
}
