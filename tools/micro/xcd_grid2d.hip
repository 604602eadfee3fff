// Round 6: is the workgroup -> XCD placement of a TWO-dimensional grid the round-robin of the
// linearised id (y * gridDim.x + x) % 8?  The split-K launches are dim3(tiles, splits): with tiles % 8 == 0 every split of
// a tile then lands on the XCD x % 8 and the fused combine's hand-off stays inside one L2.  Also checked: the placement
// behind a long-running predecessor kernel, and with two co-resident workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_grid2d.hip -o tools/micro/xcd_grid2d && tools/micro/xcd_grid2d
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

__global__ void __launch_bounds__(512) record(unsigned* out, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = xcc_id();
  // keep the workgroup resident for a while so that later ones cannot simply take its slot
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (lds[threadIdx.x] == 77) out[0] = 99;
}

__global__ void busy(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

static int check(const char* name, int X, int Y, int lds, int spin, bool behind, unsigned* d, float* scratch) {
  hipMemset(d, 0xff, (size_t)X * Y * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(record), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (behind) hipLaunchKernelGGL(busy, dim3(256 * 3 + 5), dim3(256), 0, 0, scratch, 20000);
  hipLaunchKernelGGL(record, dim3(X, Y), dim3(512), lds, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h((size_t)X * Y);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int map[8];
  for (int i = 0; i < 8; ++i) map[i] = -1;
  int bad = 0, split_bad = 0;
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      const int f = y * X + x;
      const int xc = (int)h[f];
      if (map[f & 7] < 0) map[f & 7] = xc;
      if (map[f & 7] != xc) ++bad;
      if (X % 8 == 0 && xc != (int)h[x]) ++split_bad;      // every split of tile x on the XCD of (x, 0)
    }
  bool perm = true;
  for (int i = 0; i < 8; ++i)
    for (int j = i + 1; j < 8; ++j)
      if (map[i] == map[j]) perm = false;
  printf("%-64s grid (%3d, %d): residue -> XCC %d %d %d %d %d %d %d %d  %s  off-residue WGs %d  splits off their tile's XCD %d\n", name, X, Y,
         map[0], map[1], map[2], map[3], map[4], map[5], map[6], map[7], perm ? "permutation" : "NOT a permutation", bad, split_bad);
  return bad + split_bad + (perm ? 0 : 1);
}

int main() {
  unsigned* d;
  float* scratch;
  hipMalloc(&d, 4096 * 4);
  hipMalloc(&scratch, (256 * 3 + 5) * 256 * 4);
  hipMemset(scratch, 0, (256 * 3 + 5) * 256 * 4);
  int bad = 0;
  for (int rep = 0; rep < 3; ++rep) {
    bad += check("64 tiles x 4 splits, one WG per CU (150 KB LDS)", 64, 4, 150 * 1024, 20000, false, d, scratch);
    bad += check("32 tiles x 8 splits, one WG per CU", 32, 8, 150 * 1024, 20000, false, d, scratch);
    bad += check("128 tiles x 2 splits, two WGs per CU (72 KB LDS)", 128, 2, 72 * 1024, 20000, false, d, scratch);
    bad += check("64 x 4 behind a kernel that still occupies the chip", 64, 4, 150 * 1024, 20000, true, d, scratch);
    bad += check("32 x 8 behind a kernel that still occupies the chip", 32, 8, 150 * 1024, 2000, true, d, scratch);
    bad += check("128 x 4 = 512 WGs (two waves of residency)", 128, 4, 150 * 1024, 20000, true, d, scratch);
    bad += check("20 tiles x 4 splits (tiles % 8 != 0: splits NOT co-located, expected)", 20, 4, 150 * 1024, 2000, false, d, scratch);
  }
  printf("total deviations from the round-robin-of-the-linear-id model (excluding nothing): %d\n", bad);
  return 0;
}
