// Shared device helpers for the gfx950 kernels (wave = 64 lanes, bf16 MFMA, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// ---- the two 16-bit storage formats of the path.  DT follows the dtype codes of pp_hip.h: 1 = bf16, 2 = fp16 (the
//      reference's default torch_dtype, /root/reference/app.py:548,559).  Accumulation is fp32 in both.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <int DT>
struct E16;
template <>
struct E16<PP_DT_BF16> {
  typedef bf16x8_t v8;
  static PP_DEVINL float to_f(uint16_t v) { return bf2f(v); }
  static PP_DEVINL uint16_t from_f(float f) { return f2bf(f); }
  static PP_DEVINL float lo(uint32_t u) { return bflo(u); }
  static PP_DEVINL float hi(uint32_t u) { return bfhi(u); }
  static PP_DEVINL uint32_t pack2(float lo, float hi) { return pack2bf(lo, hi); }
  static PP_DEVINL f32x4_t mfma16(v8 a, v8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static PP_DEVINL f32x16_t mfma32(v8 a, v8 b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static PP_DEVINL f32x16_t mfma32(v8 a, v8 b, f32x16_t c, int, int, int) { return mfma32(a, b, c); }
  static PP_DEVINL f32x4_t mfma16(v8 a, v8 b, f32x4_t c, int, int, int) { return mfma16(a, b, c); }
};
template <>
struct E16<PP_DT_F16> {
  typedef f16x8_t v8;
  static PP_DEVINL float to_f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  static PP_DEVINL uint16_t from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // RNE (v_cvt_f16_f32)
  static PP_DEVINL float lo(uint32_t u) { return to_f((uint16_t)(u & 0xffffu)); }
  static PP_DEVINL float hi(uint32_t u) { return to_f((uint16_t)(u >> 16)); }
  static PP_DEVINL uint32_t pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
  static PP_DEVINL f32x4_t mfma16(v8 a, v8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static PP_DEVINL f32x16_t mfma32(v8 a, v8 b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static PP_DEVINL f32x16_t mfma32(v8 a, v8 b, f32x16_t c, int, int, int) { return mfma32(a, b, c); }
  static PP_DEVINL f32x4_t mfma16(v8 a, v8 b, f32x4_t c, int, int, int) { return mfma16(a, b, c); }
};
// host-side dispatch on the runtime dtype code: PP_DT_SWITCH(dt, kernel<..., EDT>(...)) with EDT a constant in BODY
#define PP_DT_SWITCH(DTV, ...)                    \
  do {                                            \
    if ((DTV) == PP_DT_F16) {                     \
      constexpr int EDT = PP_DT_F16;               \
      __VA_ARGS__;                                \
    } else {                                      \
      constexpr int EDT = PP_DT_BF16;             \
      __VA_ARGS__;                                \
    }                                             \
  } while (0)
inline bool pp_dt_ok(int dt) { return dt == PP_DT_BF16 || dt == PP_DT_F16; }

PP_DEVINL float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// SiLU of a GroupNorm apply (gn_apply_kernel, the combine + apply kernels, gemm_combine.h: ONE function, they are compared
// bit for bit): the hardware reciprocal (1 ulp) instead of the IEEE division -- ~10 of the ~25 VALU instructions per element
// of a kernel that is otherwise a copy (10.5 M elements at the 64x64 level: 7.6 us of VALU in a 14 us launch); the 16-bit
// rounding that follows hides the difference.  The formula of conv_gn.hip's loader.
PP_DEVINL float silu_fast_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }
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
  // x Phi(x) = x - x q (x >= 0) | x q (x < 0)  =  max(x, 0) - |x| q : two instructions instead of compare/select/sub/mul
  return __builtin_fmaf(-ax, q, __builtin_fmaxf(x, 0.f));
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

// Lab switches.  The shipping library has ONE code path per op: environment-variable A/B switches, the ablation
// instantiations of the kernels and the `dbg` fields exist only in a -DPP_LAB build (`make EXTRA=-DPP_LAB`, what the
// measurement scripts under tools/ use).  pp_lab_env() is a compile-time constant otherwise.
#ifdef PP_LAB
inline int pp_lab_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
constexpr int pp_lab_env(const char*, int dflt) { return dflt; }
#endif

// Host-side error plumbing (pp_api.cpp)
void pp_set_last_error(const char* what, hipError_t e);
// raise a kernel's dynamic-LDS limit once per (kernel, device, size); PP_OK or PP_ERR_LAUNCH (last error set)
int pp_func_lds(const void* kern, int bytes, const char* what);
// compute units of the current device (pp_api.cpp; cached)
int pp_cu_count();
#define PP_CHECK_LAUNCH(what)                         \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) {                          \
      pp_set_last_error(what, e__);                   \
      return PP_ERR_LAUNCH;                           \
    }                                                 \
  } while (0)
