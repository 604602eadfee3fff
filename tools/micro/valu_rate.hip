// Issue-rate microbenchmark for the VALU ops that bound the attention softmax on gfx950: cycles per wave64 instruction
// with 1, 2, 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
template <int OP>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) {   // v_exp_f32
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 1) {   // v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 2) {   // v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                     "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      } else if (OP == 3) {   // v_max3_f32
        asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                     "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 4) {   // v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                     "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 5) {   // v_mov_b64
        asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                     "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      } else if (OP == 6) {   // v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n"
                     "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      } else if (OP == 7) {   // exp interleaved with pk_fma (do they co-issue?)
        asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %5, %5\n"
                     "v_exp_f32 %2, %2\n v_pk_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_pk_fma_f32 %7, %7, %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[0] + p2[1] + p3[1];
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(256), block(256 * wps);   // 4 SIMDs x wps waves per CU (one block per CU)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_instr = ms * 1e6 / ((double)iters * REP * wps);   // per wave-instruction per SIMD
    printf("%-22s waves/SIMD=%d  %.2f ns per wave-instr per SIMD (= %.1f cycles at 2.4 GHz)\n", name, wps, ns_per_instr,
           ns_per_instr * 2.4);
  }
}

// ---- does a wave's VALU work overlap with its own in-flight MFMA?  loop body = 1 MFMA 32x32x16 (32 cycles of matrix
// pipe) + NV independent v_fma_f32 (~4.3 cycles each).  Overlap => max(32, 4.3 NV) cycles; none => the sum.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, int NM>
__global__ void km(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a0 + i); y[i] = (__bf16)(a1 - i); }
  f16v acc0 = {0}, acc1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (NM >= 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y));
      if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (NM >= 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y));
      if (NV >= 8) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                                : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      if (NV >= 12) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s;
}

// 2 independent MFMAs + 8 VALU ops of kind OP in the body (OP as in k<>): which VALU classes run under an MFMA?
template <int OP, int NM>
__global__ void kmo(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a0 + i); y[i] = (__bf16)(a1 - i); }
  f16v acc0 = {0}, acc1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (NM) {
          if (hh == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y));
        }
        if (OP == 0) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[0] + p2[1] + p3[1] + s;
}

// The same body as kmo with the MFMA accumulators in ACCUMULATION registers (AGPR C / D operands): does the softmax-type
// VALU work overlap better with an in-flight 32x32 MFMA when the 16-register accumulator stays off the VGPR ports?
// NV8 = number of 4-instruction VALU groups per MFMA (1: 4 ops per MFMA as in kmo; 2: 8 ops, the attention slot).
template <int OP, int NM, bool AG, int NV8>
__global__ void kma(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a0 + i); y[i] = (__bf16)(a1 - i); }
  f16v acc0 = {0}, acc1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (NM) {
          if (AG) {
            if (hh == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(x), "v"(y));
          } else {
            if (hh == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y));
          }
        }
#pragma unroll
        for (int g = 0; g < NV8; ++g) {
          if (OP == 0) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
          if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
          if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
          if (OP == 8) asm volatile("v_fma_f32 %0, %4, %5, %0\n v_exp_f32 %1, %6\n v_max3_f32 %2, %2, %7, %4\n v_cvt_pk_bf16_f32 %3, %5, %6"   // the attention mix, 8 distinct sources
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + s;
}

template <int OP, int NV8>
void runa(const char* name, float* d) {
  const int iters = 20000;
  for (int cfg = 0; cfg < 3; ++cfg)          // 0: VALU alone, 1: + MFMA (VGPR accumulators), 2: + MFMA (AGPR accumulators)
    for (int wps = 1; wps <= 2; wps *= 2) {
      dim3 grid(256), block(256 * wps);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto go = [&](int it) {
        if (cfg == 0) hipLaunchKernelGGL((kma<OP, 0, false, NV8>), grid, block, 0, 0, d, it);
        else if (cfg == 1) hipLaunchKernelGGL((kma<OP, 1, false, NV8>), grid, block, 0, 0, d, it);
        else hipLaunchKernelGGL((kma<OP, 1, true, NV8>), grid, block, 0, 0, d, it);
      };
      go(100);
      hipEventRecord(e0);
      go(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ns = ms * 1e6 / ((double)iters * 4 * wps);
      printf("2 x (mfma + %d x %-14s) %-22s waves/SIMD=%d  %.1f ns per body per wave slot (2 mfma alone = 33)\n", 4 * NV8, name,
             cfg == 0 ? "VALU alone" : cfg == 1 ? "+ mfma, VGPR acc" : "+ mfma, AGPR acc", wps, ns);
    }
}

template <int OP>
void runo(const char* name, float* d) {
  const int iters = 20000;
  for (int nm = 0; nm < 2; ++nm)
    for (int wps = 1; wps <= 2; wps *= 2) {
      dim3 grid(256), block(256 * wps);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      if (nm) hipLaunchKernelGGL((kmo<OP, 1>), grid, block, 0, 0, d, 100); else hipLaunchKernelGGL((kmo<OP, 0>), grid, block, 0, 0, d, 100);
      hipEventRecord(e0);
      if (nm) hipLaunchKernelGGL((kmo<OP, 1>), grid, block, 0, 0, d, iters); else hipLaunchKernelGGL((kmo<OP, 0>), grid, block, 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ns = ms * 1e6 / ((double)iters * 4 * wps);
      printf("8 x %-18s %s waves/SIMD=%d  %.1f ns per body per wave slot (2 mfma alone = 33)\n", name, nm ? "+ 2 mfma" : "        ", wps, ns);
    }
}

template <int NV, int NM>
void runm(const char* name, float* d) {
  const int iters = 20000;
  for (int wps = 1; wps <= 2; wps *= 2) {
    dim3 grid(256), block(256 * wps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((km<NV, NM>), grid, block, 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((km<NV, NM>), grid, block, 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 4 * wps);   // per unrolled body per wave slot
    printf("%-34s waves/SIMD=%d  %.1f ns per body per wave (= %.0f cycles at 2.4 GHz)\n", name, wps, ns, ns * 2.4);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_exp_f32", d);
  run<1>("v_fma_f32", d);
  run<2>("v_pk_fma_f32", d);
  run<3>("v_max3_f32", d);
  run<4>("v_cvt_pk_bf16_f32", d);
  run<5>("v_mov_b64", d);
  run<6>("v_pk_mul_f32", d);
  run<7>("exp + pk_fma pairs", d);
  runm<0, 1>("1 mfma", d);
  runm<0, 2>("2 mfma (independent acc)", d);
  runm<8, 0>("8 fma", d);
  runm<4, 1>("1 mfma + 4 fma", d);
  runm<8, 2>("2 mfma + 8 fma", d);
  runm<12, 2>("2 mfma + 8 fma + 4 exp", d);
  runm<12, 0>("8 fma + 4 exp", d);
  runo<0>("v_exp_f32", d);
  runo<1>("v_fma_f32", d);
  runo<2>("v_pk_fma_f32", d);
  runo<3>("v_max3_f32", d);
  runo<4>("v_cvt_pk_bf16_f32", d);
  runa<1, 1>("v_fma_f32", d);
  runa<0, 1>("v_exp_f32", d);
  runa<8, 1>("softmax mix", d);
  runa<8, 2>("softmax mix", d);
  runa<1, 2>("v_fma_f32", d);
  return 0;
}
