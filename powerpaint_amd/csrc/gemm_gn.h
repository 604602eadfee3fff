// GroupNorm statistics accumulated by a GEMM / conv epilogue (shared by gemm.hip and conv_gn.hip).
#pragma once
#include "pp_common.h"

namespace {

// GroupNorm statistics from an epilogue.  A pass / tile = up to 64 output rows of ONE batch item x the <= 160 columns
// [n_blk, n_blk + ncols).  Stage 1 (gn_column): the thread that owns column `col` adds its column's (sum, sumsq), in
// fixed point, to the LDS slot of the group the column belongs to (64-bit integer LDS atomics: order-independent).
// Stage 2 (gn_flush, after a barrier): one thread per (subscription, group) moves the slot to the global accumulator
// with a 64-bit integer atomic and clears it.  Integer arithmetic end to end => bit-reproducible.
constexpr int GN_SLOTS = 24;   // >= groups one 160-column tile can touch (160 / 10 + 2)
PP_DEVINL void gn_column(const PPGemmArgs& a, unsigned long long* slots, int n_blk, int col, float sm, float sq) {
  const unsigned long long fs = (unsigned long long)(long long)__float2ll_rn(sm * PP_GN_SUM_SCALE);
  const unsigned long long fq = (unsigned long long)(long long)__float2ll_rn(sq * PP_GN_SQ_SCALE);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!a.gn_acc[k]) continue;
    const int cg = a.gn_cg[k], cbase = a.gn_c0[k] + n_blk;
    const int gl = (cbase + col) / cg - cbase / cg;          // group index relative to the first group of the tile
    __hip_atomic_fetch_add(slots + (k * GN_SLOTS + gl) * 2, fs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(slots + (k * GN_SLOTS + gl) * 2 + 1, fq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
PP_DEVINL void gn_flush(const PPGemmArgs& a, unsigned long long* slots, int m0, int n_blk, int ncols, int tid) {
  const int k = tid / GN_SLOTS, gl = tid - k * GN_SLOTS;
  if (k < 2 && a.gn_acc[k]) {
    const int cg = a.gn_cg[k], cbase = a.gn_c0[k] + n_blk;
    const int g_first = cbase / cg, g_last = (cbase + ncols - 1) / cg;
    if (g_first + gl <= g_last) {
      const int b = m0 / a.rows_per_batch;
      unsigned long long* dst =
          reinterpret_cast<unsigned long long*>(a.gn_acc[k]) + ((size_t)b * a.gn_groups[k] + g_first + gl) * 2;
      unsigned long long* sl = slots + (k * GN_SLOTS + gl) * 2;
      __hip_atomic_fetch_add(dst, sl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(dst + 1, sl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.gn_dup_mask >> k & 1) {      // the twin half of a CFG batch holds the same values (PPGemmArgs.out_dup_rows)
        unsigned long long* dup = dst + (size_t)a.gn_dup_batch * a.gn_groups[k] * 2;
        __hip_atomic_fetch_add(dup, sl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(dup + 1, sl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sl[0] = 0ull;
      sl[1] = 0ull;
    }
  }
}

}  // namespace
