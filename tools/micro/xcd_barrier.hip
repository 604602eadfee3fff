// VERDICT round 4, item 3 (go / no-go for a persistent "sample per XCD" section over the latency-bound middle of the UNet):
// what does a hand-off between workgroups OF ONE XCD cost when nothing leaves that XCD's L2 -- a 32-arrival XCD-local
// counter barrier plus the cheapest visibility primitive that is actually sufficient -- against the agent-scope
// release / acquire form that is correct across XCDs (MI355X_MICROARCH.md, barrier-xcd: 4.8 - 7.2 us)?
//
// 256 workgroups (one per CU: 64 KB of LDS each keeps a second one out), each reads its XCC id and takes a rank inside its
// XCD.  Per phase: every workgroup writes a stamped 8 KB slot (double buffered), arrives at its XCD's counter, waits for
// the XCD's other 31, then reads the slot of its ring neighbour IN THE SAME XCD and counts stale words.  Variants:
//   0  no synchronisation at all (the floor: write + read)
//   1  plain stores -> s_waitcnt vmcnt(0) -> L2 atomic (no sc bits) ; poll with sc1 loads ; payload with sc1 loads
//   2  as 1, payload with PLAIN loads (is the L1 the only thing in the way?)
//   3  as 1, payload with plain loads behind `buffer_inv sc1` (one lane + barrier)
//   4  agent-scope release fence -> agent atomic ; poll ; agent-scope acquire fence ; plain payload loads (the cross-XCD form)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier.hip -o tools/micro/xcd_barrier && tools/micro/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned u4;
constexpr int SLOT16 = 512;     // 16-byte words per slot (8 KB)
constexpr int PH = 200;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
__device__ __forceinline__ u4 load_sc1(const u4* p) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned load_u32_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void atomic_inc_l2(unsigned* p) {      // no sc0 / sc1: performed in this XCD's L2
  const unsigned one = 1u;
  asm volatile("global_atomic_add %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(one) : "memory");
}

template <int VAR>
__global__ void __launch_bounds__(256) phases(u4* slots, unsigned* ranks, unsigned* arrive, unsigned* stale, unsigned* xcc_of) {
  extern __shared__ char lds[];      // (occupancy limiter only)
  __shared__ unsigned s_rank, s_xcc;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_xcc = xcc_id();
    s_rank = atomicAdd(&ranks[s_xcc], 1u);
    xcc_of[blockIdx.x] = s_xcc;
  }
  __syncthreads();
  const unsigned xcc = s_xcc, rank = s_rank;
  // slots: [xcc][rank 0..63][parity][SLOT16]
  u4* mine = slots + (((size_t)xcc * 64 + rank) * 2) * SLOT16;
  unsigned bad = 0;
  __shared__ unsigned s_dead;        // sticky: a wait timed out (placement was not 32 per XCD / a CU missing): never hang the box
  if (tid == 0) s_dead = 0;
  __syncthreads();
  for (int ph = 0; ph < PH; ++ph) {
    u4* w = mine + (ph & 1) * SLOT16;
    for (int i = tid; i < SLOT16; i += 256) w[i] = u4{(unsigned)ph, rank, xcc, (unsigned)i};
    unsigned nranks = 32;
    if (VAR != 0) {
      if (VAR == 4) {
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_fetch_add(&arrive[xcc * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int spin = 0; !s_dead && __hip_atomic_load(&arrive[xcc * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                                              nranks * (unsigned)(ph + 1); ++spin) {
            __builtin_amdgcn_s_sleep(1);
            if (spin > 400000) { s_dead = 1; atomicAdd(&stale[255], 1u << 30); }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's stores have reached the L2
        __syncthreads();
        if (tid == 0) {
          atomic_inc_l2(&arrive[xcc * 64]);
          for (int spin = 0; !s_dead && load_u32_sc1(&arrive[xcc * 64]) < nranks * (unsigned)(ph + 1); ++spin) {
            __builtin_amdgcn_s_sleep(1);
            if (spin > 400000) { s_dead = 1; atomicAdd(&stale[255], 1u << 30); }
          }
          if (VAR == 3) asm volatile("buffer_inv sc1" ::: "memory");
        }
        __syncthreads();
      }
    }
    const u4* r = slots + (((size_t)xcc * 64 + ((rank + 1) & 31)) * 2 + (ph & 1)) * SLOT16;
    for (int i = tid; i < SLOT16; i += 256) {
      const u4 v = (VAR == 1) ? load_sc1(r + i) : r[i];
      if (v[0] != (unsigned)ph || v[3] != (unsigned)i) ++bad;
    }
  }
  if (bad) atomicAdd(&stale[blockIdx.x], bad);
  if (lds[0] == 77) stale[0] = 1;
}

template <int VAR>
void run(const char* name, u4* slots, unsigned* ranks, unsigned* arrive, unsigned* stale, unsigned* xcc_of) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(phases<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  std::vector<float> ts;
  unsigned long long tot = 0;
  std::vector<unsigned> h(256), hx(256);
  for (int it = 0; it < 7; ++it) {
    hipMemset(ranks, 0, 64); hipMemset(arrive, 0, 8 * 64 * 4); hipMemset(stale, 0, 256 * 4);
    hipMemset(slots, 0xff, (size_t)8 * 64 * 2 * SLOT16 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(phases<VAR>, dim3(256), dim3(256), 96 * 1024, 0, slots, ranks, arrive, stale, xcc_of);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f / PH);
    hipMemcpy(h.data(), stale, 256 * 4, hipMemcpyDeviceToHost);
    for (unsigned v : h) tot += v;
  }
  hipMemcpy(hx.data(), xcc_of, 256 * 4, hipMemcpyDeviceToHost);
  int per[8] = {0};
  for (unsigned v : hx) per[v & 7]++;
  std::sort(ts.begin(), ts.end());
  printf("%-78s %6.2f us / phase (median of 7 x %d phases)   stale words %llu   WGs per XCD %d %d %d %d %d %d %d %d\n", name, ts[3],
         PH, tot, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
}

int main() {
  u4* slots; unsigned *ranks, *arrive, *stale, *xcc_of;
  hipMalloc(&slots, (size_t)8 * 64 * 2 * SLOT16 * 16); hipMalloc(&ranks, 64); hipMalloc(&arrive, 8 * 64 * 4);
  hipMalloc(&stale, 256 * 4); hipMalloc(&xcc_of, 256 * 4);
  run<0>("0 no synchronisation (write 8 KB + read the neighbour's 8 KB; stale expected)", slots, ranks, arrive, stale, xcc_of);
  run<1>("1 plain stores, vmcnt(0), L2 atomic; sc1 poll; sc1 payload loads", slots, ranks, arrive, stale, xcc_of);
  run<2>("2 ... payload with PLAIN loads (no L1 invalidate)", slots, ranks, arrive, stale, xcc_of);
  run<3>("3 ... payload with plain loads behind one buffer_inv sc1", slots, ranks, arrive, stale, xcc_of);
  run<4>("4 agent release fence, agent atomic, agent acquire fence, plain payload (cross-XCD form)", slots, ranks, arrive, stale, xcc_of);
  return 0;
}
