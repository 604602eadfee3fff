// Shared device helpers for the gfx950 kernels (wave = 64 lanes, bf16 MFMA, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pp_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define PP_DEVINL __device__ __forceinline__

PP_DEVINL float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
PP_DEVINL uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // round-to-nearest-even, v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(uint16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
PP_DEVINL uint32_t pack2bf(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
PP_DEVINL float bflo(uint32_t u) { return __uint_as_float(u << 16); }
PP_DEVINL float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

PP_DEVINL float silu_f(float x) { return x / (1.0f + __expf(-x)); }
PP_DEVINL float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Exact-erf GELU, x * Phi(x), with erfc from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below bf16 output
// rounding): Phi(-|x|) = 0.5 * erfc(|x|/sqrt 2) = 0.5 * poly(t) * exp(-x^2/2), t = 1/(1 + p|x|/sqrt 2).  ~15 VALU
// instructions, branch-free, against ~45 for the library erff -- the GEGLU epilogue is VALU-bound at K = 320.
PP_DEVINL float gelu_fast_f(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  p = __builtin_fmaf(t, p, 0.5f * 1.421413741f);
  p = __builtin_fmaf(t, p, 0.5f * -0.284496736f);
  p = __builtin_fmaf(t, p, 0.5f * 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));   // Phi(-|x|)
  return x * (x >= 0.f ? 1.0f - q : q);
}

// Buffer resource over [base, base+bytes): out-of-range voffset reads return 0 -> free zero padding for im2col.
PP_DEVINL __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
#define PP_OOB 0x80000000u

PP_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
PP_DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Host-side error plumbing (pp_api.cpp)
void pp_set_last_error(const char* what, hipError_t e);
#define PP_CHECK_LAUNCH(what)                         \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) {                          \
      pp_set_last_error(what, e__);                   \
      return PP_ERR_LAUNCH;                           \
    }                                                 \
  } while (0)
