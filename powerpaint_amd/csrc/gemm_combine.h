// Split-K combine INSIDE the producing kernel, XCD-local (round 6; shared by gemm.hip and conv_gn.hip).
//
// A split-K launch is dim3(tiles, splits): the dispatcher hands workgroups to the eight XCDs round-robin in the order of
// the linear id  y * gridDim.x + x  (observed, tools/micro/xcd_grid2d.hip; pp_xcd_placement_ok() re-checks it on the box
// before any plan relies on it), so with  tiles % 8 == 0  every split of tile x runs on XCD x % 8 and the fp32 slabs of a
// tile all live in ONE L2.  Inside one XCD the L2 is the coherence point (tools/micro/xcd_barrier.hip, variant 1):
//   producer : plain slab stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> ONE returning atomic on the tile's counter
//   consumer : the workgroup that draws the last ticket reads the slabs with `sc1` loads (they bypass its CU's L1 and
//              are served by that L2), sums them IN SLAB ORDER (fixed order => bit-reproducible, and bit-identical to the
//              separate combine kernels below it replaces) and runs their epilogue: bias / time-embedding row /
//              residuals / 16-bit store, the fixed-point GroupNorm statistics of the output, and -- where the tile holds
//              whole (batch item, group) populations -- the consumer GroupNorm (+ SiLU) apply (PPGemmArgs.gn_next_*).
// No spinning, no agent-scope fence: a workgroup that is not last simply exits.  The counter is a 64-bit word:
//   bits 0..7 arrivals, bits 8 + 4 j .. 11 + 4 j arrivals from XCC j -- device atomics are coherent across XCDs whatever
//   the placement, so the last arriver can PROVE the co-location it relies on (all arrivals in its own XCC's field); a
//   violation is counted in PPGemmArgs.combine_fault (the host raises on it) -- never a silently stale sum.
// Replaces pp_splitk_reduce_kernel<true> / pp_splitk_reduce_gn_kernel / pp_splitk_reduce_gn_apply_kernel (gemm.hip) for
// the launches pp_gemm_fused_combine() admits; reference ops: the ResnetBlock2D convs and FeedForward / proj_out Linears of
// the 16x16 and 8x8 levels, /root/reference/powerpaint/models/unet_2d_blocks.py:1457-1500, 850-899, 2696-2770.
#pragma once
#include <type_traits>
#include <utility>

#include "pp_common.h"
#include "gemm_gn.h"

namespace {

PP_DEVINL unsigned pp_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

constexpr int FC_BN = 160;
constexpr int FC_VLD = FC_BN * 2 + 16;              // row stride of the finished 16-bit tile in LDS (bytes; +16: bank spread)
constexpr int fc_flag_off(int bm) { return bm * FC_VLD + 2 * GN_SLOTS * 2 * 8 + 2 * FC_BN * 4; }   // the arrival verdict word
constexpr int fc_lds_bytes(int bm) { return fc_flag_off(bm) + 64; }

// Arrival at the tile's counter.  Call with every slab store of this workgroup issued.  Returns 0 = not last (exit),
// 1 = last, 2 = last but the splits did NOT share an XCD (fault recorded; the caller still combines so that `out` is
// written -- the host refuses the result).
PP_DEVINL int splitk_arrive(const PPGemmArgs& a, int tile_id, int splits, int tid, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // every thread's slab stores have reached the XCD's L2
  if (tid == 0) {
    const unsigned xcc = pp_xcc_id();
    const unsigned long long inc = 1ull + (1ull << (8 + 4 * xcc));
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(a.tile_ctr) + tile_id;
    const unsigned long long old = __hip_atomic_fetch_add(ctr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int r = 0;
    if ((int)(old & 0xffull) == splits - 1) {
      const unsigned long long now = old + inc;
      r = ((now >> (8 + 4 * xcc)) & 0xfull) == (unsigned long long)splits ? 1 : 2;
      __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
      if (r == 2 && a.combine_fault) atomicAdd(a.combine_fault, 1u);
    }
    *lds_flag = r;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  return *lds_flag;
}

template <int... I, class F>
PP_DEVINL void fc_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void fc_static_for(F&& f) { fc_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// The combine of one BM x 160 tile by the 512 threads of the last-arriving workgroup.  `smem`: fc_lds_bytes(BM) of LDS that
// nothing else uses any more.  Arithmetic and summation order are those of the lean combine kernels in gemm.hip.
//
// One workgroup pulls S x BM x 640 bytes (0.16 .. 1.3 MB) through ONE CU's 64 B/clk path while the rest of the chip idles, so
// the loop is built around that path (first version: a thread owned 8 columns = two half-dense 16-byte loads per slab and a
// dependent rowvec load per row: ~40 GB/s, 13 .. 36 us SLOWER per launch than the separate combine; profiles/r06_fused_combine.txt):
//   * thread = (row slot, 4-column strip): every slab load instruction of a wave is 1 KB of whole 640-byte tile rows;
//   * a thread keeps ONE strip for the whole tile: bias / gamma / beta once, the time-embedding row once per batch item;
//   * the tile is a flat sequence of groups of G rows per thread; the loads of group i + 1 (slabs, residuals) are in
//     flight while group i is summed -- across the per-pass statistics phases as well.
// SMAX = 4 | 8 slabs at most (register arrays), G = 3 | 2 rows per group.
template <int BM, int EDT, int SMAX, int G>
PP_DEVINL void splitk_fused_combine_run(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int splits, int tid) {
  using E = E16<EDT>;
  constexpr int T = 512, BN = FC_BN, SC = BN / 4, RS = 12, ROWS = 64, IPP = 6, VLD = FC_VLD;   // 40 strips x 12 row slots
  constexpr int NG = IPP / G, NPASS = BM / ROWS, NTOT = NPASS * NG;
  static_assert(IPP % G == 0 && RS * IPP >= ROWS && SC * RS <= T, "row slots x items cover a 64-row pass");
  constexpr int SL_OFF = BM * VLD, SC_OFF = SL_OFF + 2 * GN_SLOTS * 2 * 8, SH_OFF = SC_OFF + BN * 4;
  const int strip = tid % SC, rslot = tid / SC;
  const int n = n_blk + strip * 4;
  const bool active = rslot < RS && n < a.N;
  const bool gns = a.gn_acc[0] != nullptr || a.gn_acc[1] != nullptr;
  const bool apply = a.gn_next_out != nullptr;
  const int hw = a.rows_per_batch;
  const int ncols = min(BN, a.N - n_blk);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + SL_OFF);
  float* sc_s = reinterpret_cast<float*>(smem + SC_OFF);
  float* sh_s = reinterpret_cast<float*>(smem + SH_OFF);
  if (tid < 2 * GN_SLOTS * 2) slots[tid] = 0ull;
  const uint32_t slab_b = (uint32_t)a.M * (uint32_t)a.N * 4u;                  // (host-checked: splits * slab_b < 2^31)
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.workspace, slab_b * (uint32_t)splits);
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t bs = zero4, gam = zero4, bet = zero4;
  if (active && a.bias) bs = *reinterpret_cast<const f32x4_t*>(a.bias + n);
  if (active && apply) {
    gam = *reinterpret_cast<const f32x4_t*>(a.gn_next_gamma + n);
    bet = *reinterpret_cast<const f32x4_t*>(a.gn_next_beta + n);
  }
  f32x4_t p[2][G][SMAX], rv[2];
  u32x2_t q1[2][G], q2[2][G];
  rv[0] = rv[1] = zero4;

  auto load = [&](auto GI_) __attribute__((always_inline)) {
    constexpr int gi = decltype(GI_)::value, buf = gi & 1, pass = gi / NG, g = gi % NG;
    const int m0 = m_blk + pass * ROWS;
    if constexpr (g == 0) {
      if (active && a.rowvec) rv[pass & 1] = *reinterpret_cast<const f32x4_t*>(a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + n);
    }
#pragma unroll
    for (int jj = 0; jj < G; ++jj) {
      const int row = rslot + RS * (g * G + jj), m = m0 + row;
      const bool ok = active && row < ROWS && m < a.M;
      const int voff = ok ? (m * a.N + n) * 4 : (int)PP_OOB;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s < splits) p[buf][jj][s] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(s * slab_b), 16));
        else p[buf][jj][s] = zero4;
      }
      q1[buf][jj] = u32x2_t{0u, 0u};
      q2[buf][jj] = u32x2_t{0u, 0u};
      if (ok && a.res1)
        q1[buf][jj] = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res1 +
                                                        (size_t)((a.res1_wrap_rows > 0 && m >= a.res1_wrap_rows) ? m - a.res1_wrap_rows : m) * a.ldres1 + n);
      if (ok && a.res2) q2[buf][jj] = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
    }
  };
  auto process = [&](auto GI_) __attribute__((always_inline)) {
    constexpr int gi = decltype(GI_)::value, buf = gi & 1, pass = gi / NG, g = gi % NG;
    const int m0 = m_blk + pass * ROWS;
#pragma unroll
    for (int jj = 0; jj < G; ++jj) {
      const int row = rslot + RS * (g * G + jj), m = m0 + row;
      if (!(active && row < ROWS && m < a.M)) continue;
      f32x4_t v = p[buf][jj][0];
#pragma unroll
      for (int s = 1; s < SMAX; ++s) v += p[buf][jj][s];
      if (a.bias) v += bs;                           // (conditional like the lean kernels: -0 + 0 would flip a sign bit)
      if (a.rowvec) v += rv[pass & 1];
      v *= a.scale;
      const u32x2_t r1 = q1[buf][jj], r2 = q2[buf][jj];
      v[0] += E::lo(r1[0]) + E::lo(r2[0]); v[1] += E::hi(r1[0]) + E::hi(r2[0]);
      v[2] += E::lo(r1[1]) + E::lo(r2[1]); v[3] += E::hi(r1[1]) + E::hi(r2[1]);
      u32x2_t o;
      o[0] = E::pack2(v[0], v[1]);
      o[1] = E::pack2(v[2], v[3]);
      *reinterpret_cast<u32x2_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
      if (gns) *reinterpret_cast<u32x2_t*>(smem + (pass * ROWS + row) * VLD + strip * 8) = o;
    }
  };
  // statistics (+ the consumer norm) of pass `pass`, whose finished rows are in LDS
  auto gn_phase = [&](int pass) __attribute__((always_inline)) {
    const int m0 = m_blk + pass * ROWS;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (16-row block, column) moments of the values as stored, rows in order: the partial sums of pp_splitk_reduce_gn_kernel,
    // so the integers the accumulators receive are the same
    for (int q = tid; q < (ROWS / 16) * BN; q += T) {
      const int blk = q / BN, col = q - blk * BN;
      if (col < ncols && m0 + blk * 16 < a.M) {
        const char* vp = smem + (pass * ROWS + blk * 16) * VLD + col * 2;
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = E::to_f(*reinterpret_cast<const uint16_t*>(vp + r * VLD));
          sm += v;
          sq += v * v;
        }
        gn_column(a, slots, n_blk, col, sm, sq);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (apply && ((m0 + ROWS) % hw) != 0) return;            // (apply: the statistics of a batch item span several passes)
    if (apply && tid < ncols) {
      // the consumer's (scale, shift) of the tile's columns from its complete slots: the arithmetic of gn_fold_acc
      const int k = a.gn_next_sub, cg = a.gn_cg[k], cbase = a.gn_c0[k] + n_blk;
      const int gl = (cbase + tid) / cg - cbase / cg;
      const double s = (double)(long long)slots[(k * GN_SLOTS + gl) * 2] * (1.0 / (double)PP_GN_SUM_SCALE);
      const double q = (double)(long long)slots[(k * GN_SLOTS + gl) * 2 + 1] * (1.0 / (double)PP_GN_SQ_SCALE);
      const double cnt = (double)hw * (double)cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      sc_s[tid] = (float)(1.0 / sqrt(var + (double)a.gn_next_eps));        // rstd; gamma is applied by the strip's thread
      sh_s[tid] = (float)mean;
    }
    if (apply) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    gn_flush(a, slots, m0, n_blk, ncols, tid);       // (batch item of row m0; clears the slots it moved)
    if (apply && active) {
      const f32x4_t rstd4 = *reinterpret_cast<const f32x4_t*>(sc_s + strip * 4), mean4 = *reinterpret_cast<const f32x4_t*>(sh_s + strip * 4);
      f32x4_t a0, b0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sc = rstd4[e] * gam[e];
        a0[e] = sc;
        b0[e] = bet[e] - mean4[e] * sc;
      }
      const bool silu = a.gn_next_silu != 0;
      const int trow0 = pass * ROWS + ROWS - hw;     // first tile row of the finished batch item
      for (int rr = rslot; rr < hw; rr += RS) {
        const int m = m_blk + trow0 + rr;
        const u32x2_t v = *reinterpret_cast<const u32x2_t*>(smem + (trow0 + rr) * VLD + strip * 8);
        float r[4];
        r[0] = E::lo(v[0]) * a0[0] + b0[0]; r[1] = E::hi(v[0]) * a0[1] + b0[1];
        r[2] = E::lo(v[1]) * a0[2] + b0[2]; r[3] = E::hi(v[1]) * a0[3] + b0[3];
        if (silu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = silu_f(r[j]);
        }
        u32x2_t o;
        o[0] = E::pack2(r[0], r[1]);
        o[1] = E::pack2(r[2], r[3]);
        *reinterpret_cast<u32x2_t*>((uint16_t*)a.gn_next_out + (size_t)m * a.N + n) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slots / (rstd, mean) free for the next item
  };

  load(std::integral_constant<int, 0>{});
  fc_static_for<NTOT>([&](auto GI_) __attribute__((always_inline)) {
    constexpr int gi = decltype(GI_)::value;
    if constexpr (gi + 1 < NTOT) load(std::integral_constant<int, gi + 1>{});
    process(GI_);
    if constexpr (gi % NG == NG - 1) {
      if (gns) gn_phase(gi / NG);
    }
  });
}

template <int BM, int T, int EDT>
PP_DEVINL void splitk_fused_combine(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int splits, int tid) {
  static_assert(T == 512, "the in-kernel combine is written for the 8-wave tiles");
  if (splits <= 4) splitk_fused_combine_run<BM, EDT, 4, 3>(a, smem, m_blk, n_blk, splits, tid);
  else splitk_fused_combine_run<BM, EDT, 8, 2>(a, smem, m_blk, n_blk, splits, tid);
}

}  // namespace
