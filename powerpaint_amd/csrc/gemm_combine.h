// Split-K combine INSIDE the producing kernel, XCD-local, every split combining its own share (round 6; shared by
// gemm.hip and conv_gn.hip).
//
// A split-K launch is dim3(tiles, splits): the dispatcher hands workgroups to the eight XCDs round-robin in the order of
// the linear id  y * gridDim.x + x  (observed, tools/micro/xcd_grid2d.hip; pp_xcd_placement_ok() re-checks it on the box
// before any plan relies on it), so with  tiles % 8 == 0  every split of tile x runs on XCD x % 8 and the fp32 slabs of a
// tile all live in ONE L2.  Inside one XCD the L2 is the coherence point (tools/micro/xcd_barrier.hip, variant 1: plain
// stores -> s_waitcnt vmcnt(0) -> an atomic on a word of that L2 -> `sc1` loads on the other side; 0.93 us per hand-off).
//
// First form (profiles/r06_fused_combine.txt): the workgroup that arrives LAST at its tile combines the whole tile -- one
// CU pulls splits x rows x 640 bytes while the rest of the chip idles: 4 .. 20 us SLOWER per launch than the separate
// combine kernel.  This form: the S = 2 | 4 | 8 splits of a tile each combine 1 / S of its rows (a SHARE: BM / S = 16 .. 128
// rows x 160 columns), so the tail of the launch is S times shorter and runs on every CU.
//   arrive   : slab stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> ONE returning atomic on the tile's counter.
//   wait     : a workgroup that is not last polls the counter (`sc1` load + s_sleep) until the S-th arrival.  A launch is
//              sized to one workgroup per CU, so its splits are co-resident and the wait is about the skew between them;
//              it is BOUNDED all the same (FC_SPIN_TICKS): a workgroup that runs out of patience marks its share
//              abandoned (compare-and-swap, which fails once the last split has arrived) and exits; the last arriver -- it
//              never waits -- sees the mark in the value its own arrival returned and combines that share as well.  No
//              placement, occupancy or profiler serialisation can turn the wait into a hang.
//   combine  : slabs read with `sc1` loads, summed IN SLAB ORDER (bit-reproducible, and bit-identical to the separate
//              combine kernels in gemm.hip this replaces), bias / time-embedding row / residuals / 16-bit store, the
//              fixed-point GroupNorm statistics of the output (integer atomics: order-free).
//   apply    : where the tile holds whole (batch item, group) populations the consumer GroupNorm (+ SiLU) of
//              PPGemmArgs.gn_next_* follows: every share stores its (16-row block, group) integer sums to the tile's
//              scratch (plain stores, overwritten by every launch: nothing to re-arm), a second arrival / wait on the
//              tile's second counter, then every share reads the sums of its batch item, and normalises its own rows.
// The counter is a 64-bit word: bits 40..63 arrivals, MONOTONIC (a launch adds S; round and target follow from the value an
// arrival returns, so nothing has to be reset while others poll); bits 0..31 arrivals per XCC of this round and bits
// 32..39 the abandoned shares -- the last arriver takes both out again with one non-returning atomic.  Every access to it is
// performed in the XCD's L2 (no sc1; an agent-scope atomic is forwarded to the memory side and costs microseconds), so a
// count of S is S arrivals IN THIS L2: a workgroup that finds itself last has, by that alone, the proof of the co-location
// its `sc1` loads rely on.  Splits of a tile on different XCDs -- ruled out by the grid shape and pp_xcd_placement_ok() --
// would never count to S anywhere: every one of them gives up after FC_SPIN_TICKS, and an abandoned share that no last
// arriver takes back stays counted in PPGemmArgs.combine_fault (the host raises on it) -- never a silently stale sum.
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
constexpr int fc_flag_off(int bm) { return bm * FC_VLD + 2 * GN_SLOTS * 2 * 8 + 2 * FC_BN * 4; }   // verdict words of an arrival
constexpr int fc_lds_bytes(int bm) { return fc_flag_off(bm) + 64; }
constexpr int FC_CTR_BYTES = 16;                    // per tile: arrival counter, second (statistics) counter
#ifdef PP_LAB
// (lab) + 8 x 64 bytes of s_memrealtime stamps per split, tools/fc_time.py: 0 entry, 1 slab stores acknowledged, 2 arrival
// atomic back, 3 last split seen, 7 own inputs (bias, residuals) back, 4 first slab loads back, 5 share combined (stores
// issued), 6 stores acknowledged.  fc_stp: where this split's stamps go (nullptr: none)
constexpr int FC_SCR_BYTES = 16 * GN_SLOTS * 16 + 512;
#define FC_STAMP(i)                                                            \
  do {                                                                         \
    if (tid == 0 && fc_stp) fc_stp[i] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#define FC_STAMP_SYNC(i)                                                       \
  do {                                                                         \
    if (fc_stp) {                                                              \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         \
      FC_STAMP(i);                                                             \
    }                                                                          \
  } while (0)
#else
constexpr int FC_SCR_BYTES = 16 * GN_SLOTS * 16;    // per tile, behind the slabs: [16-row block][group slot] x (sum, sumsq)
#define FC_STAMP(i) do { } while (0)
#define FC_STAMP_SYNC(i) do { } while (0)
#endif
constexpr unsigned long long FC_SPIN_TICKS = 20000; // 200 us of s_memrealtime (100 MHz) before a waiting split gives up
constexpr unsigned FC_FAULT = 1u << 16;
// Cache policy of the slab loads.  A slab line is read ONCE per launch by the CU that combines it and was never in that CU's
// L1 before (the L1 is invalidated when a kernel starts; a CU only ever wrote its OWN slab), so a plain load misses the L1
// and is served by the XCD's L2, where the line is still dirty.  `sc1` loads (agent scope, aux 16) are forwarded to the
// memory side: 2.3 us and fabric-bound for the 21 MB of an 8x8 launch (lab stamps, profiles/r06_fused_combine.txt).
#ifndef FC_SLAB_AUX
#define FC_SLAB_AUX 0
#endif

// Atomics performed IN THE XCD'S L2 (no sc1: not forwarded to the memory side) and `sc1` loads served by that L2 -- the
// hand-off of tools/micro/xcd_barrier.hip, variant 1.  Every access to a tile's counter words is one of these.
PP_DEVINL unsigned long long fc_l2_add_ret(unsigned long long* p, unsigned long long v) {
  unsigned long long old;
  asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(v) : "memory");
  return old;
}
PP_DEVINL unsigned long long fc_l2_or_ret(unsigned long long* p, unsigned long long v) {
  unsigned long long old;
  asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(v) : "memory");
  return old;
}
PP_DEVINL void fc_l2_add(unsigned long long* p, unsigned long long v) {
  asm volatile("global_atomic_add_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
PP_DEVINL void fc_l2_and(unsigned long long* p, unsigned long long v) {
  asm volatile("global_atomic_and_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
PP_DEVINL unsigned long long fc_load_sc1(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// Arrival of split `share` at its tile.  Call with every slab store of this workgroup issued.  Returns the mask of the
// shares this workgroup combines (0: none -- abandoned after FC_SPIN_TICKS -- exit).
// `between`: run by every thread once the arrival is on its way -- the loads of what the share needs that does not depend on
// the other splits (fc_preload): their ~2 us of HBM latency pass while thread 0 waits for the last split.
template <int S, class F>
PP_DEVINL unsigned fc_arrive(const PPGemmArgs& a, int tile_id, int share, int tid, unsigned* lds, unsigned long long* fc_stp, F&& between) {
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // every thread's slab stores have reached the XCD's L2
  const unsigned xcc = pp_xcc_id();
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(a.tile_ctr) + 2 * tile_id;
  const unsigned long long inc = (1ull << 40) + (1ull << (4 * xcc));
  unsigned long long old = 0ull;
  if (tid == 0) {
    FC_STAMP(1);
    old = fc_l2_add_ret(ctr, inc);
    FC_STAMP(2);
  }
  between();
  if (tid == 0) {
    const unsigned c_old = (unsigned)(old >> 40);
    const unsigned target = ((c_old & ~(unsigned)(S - 1)) + S) & 0xffffffu;
    unsigned mask = 0;
    if ((c_old & (S - 1)) == S - 1) {
      const unsigned long long low = (old + inc) & ((1ull << 40) - 1);
      const unsigned extra = (unsigned)(low >> 32) & ((1u << S) - 1u) & ~(1u << share);
      mask = (1u << share) | extra;
      fc_l2_add(ctr, 0ull - low);                                   // the round-local fields out again (no reply awaited)
      if ((unsigned)low != ((unsigned)S << (4 * xcc))) {            // (cannot happen: a count of S in ONE L2 is S arrivals there)
        mask |= FC_FAULT;
        if (a.combine_fault) atomicAdd(a.combine_fault, 1u);
      }
      if (extra && a.combine_fault) atomicSub(a.combine_fault, (unsigned)__builtin_popcount(extra));   // abandoned shares recovered
    } else {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned long long v = fc_load_sc1(ctr);
#ifdef PP_LAB
        if (a.dbg & 0x400) { mask = 1u << share; break; }
#endif
        if ((unsigned)(v >> 40) == target) {
          mask = 1u << share;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > FC_SPIN_TICKS) {
          const unsigned long long bit = 1ull << (32 + share);
          const unsigned long long was = fc_l2_or_ret(ctr, bit);
          if ((unsigned)(was >> 40) == target) {                    // the last split arrived in between: not abandoned after all
            fc_l2_and(ctr, ~bit);
            mask = 1u << share;
          } else if (a.combine_fault) {
            // abandoned: the last arriver combines this share and takes this count back.  Splits of one tile on DIFFERENT
            // XCDs -- which pp_xcd_placement_ok() and the grid shape rule out -- would count in different L2s, never reach S
            // anywhere, and leave this count standing: the host refuses the result.
            atomicAdd(a.combine_fault, 1u);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    FC_STAMP(3);
    lds[0] = mask;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  return lds[0];
}

// Second arrival: this workgroup's `nshares` shares have their statistics in the tile's scratch; returns when all S have.
// (The second counter is monotonic as well and S per launch: the arrivals of this round so far are its count mod S.)
template <int S>
PP_DEVINL void fc_arrive_stats(const PPGemmArgs& a, int tile_id, int nshares, int tid) {
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (tid == 0) {
    unsigned long long* c2 = reinterpret_cast<unsigned long long*>(a.tile_ctr) + 2 * tile_id + 1;
    const unsigned long long inc = (unsigned long long)nshares << 40;
    unsigned long long v = fc_l2_add_ret(c2, inc);
    const unsigned target = (((unsigned)(v >> 40) & ~(unsigned)(S - 1)) + S) & 0xffffffu;
    v += inc;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((unsigned)(v >> 40) != target) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 8 * FC_SPIN_TICKS) {   // (every share is held by a running workgroup that
        if (a.combine_fault) atomicAdd(a.combine_fault, 1u);             //  does not wait before it arrives here: unreachable)
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      v = fc_load_sc1(c2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int... I, class F>
PP_DEVINL void fc_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void fc_static_for(F&& f) { fc_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// Geometry of a share: R = BM / S rows in passes of RP <= 64 rows; thread = (row slot of 12, 4-column strip of 40) -- every
// slab load instruction of a wave is 1 KB of whole 640-byte tile rows; a thread keeps ONE strip (bias / gamma / beta once,
// the time-embedding row once per pass); the loads of group i + 1 (G rows per thread: slabs, residuals) are in flight while
// group i is summed.
template <int BM, int S>
struct FcGeom {
  static constexpr int R = BM / S, RP = R < 64 ? R : 64, NPASS = R / RP;
  static constexpr int G = S == 8 ? 2 : 3;                              // rows per thread in flight: G x S 16-byte loads
  static constexpr int RS = 12, SC = FC_BN / 4;
  static constexpr int IPP = ((RP + RS - 1) / RS + G - 1) / G * G, NG = IPP / G, NTOT = NPASS * NG;
  static_assert(BM % S == 0 && R % 16 == 0 && R >= 16 && RP * NPASS == R, "a share is whole 16-row blocks");
  static_assert(SC * RS <= 512 && RS * IPP >= RP, "row slots x items cover a pass");
};

// What a share needs that does NOT depend on the other splits: the bias strip, the time-embedding row of its first pass and the
// residual rows of its first group -- cold HBM lines, ~1.9 us, and loads return in order, so slab loads issued behind them wait
// for them as well.  They are issued right BEHIND the arrival atomic (fc_arrive's `between`) and pass while thread 0 waits for
// the last split.  (In FRONT of the arrival they queue behind the 80 KB of slab stores and the arrival's own
// `s_waitcnt vmcnt(0)` waits for them too: +2 us per launch, measured.)
template <int G>
struct FcPre {
  f32x4_t bs, rv0;
  u32x2_t q1[G], q2[G];
};
template <int BM, int S>
PP_DEVINL void fc_preload(const PPGemmArgs& a, int m_blk, int n_blk, int share, int tid, FcPre<FcGeom<BM, S>::G>& pre) {
  using Gm = FcGeom<BM, S>;
  const int strip = tid % Gm::SC, rslot = tid / Gm::SC;
  const int n = n_blk + strip * 4;
  const bool active = rslot < Gm::RS && n < a.N;
  const int m0 = m_blk + share * Gm::R;
  pre.bs = pre.rv0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (active && a.bias) pre.bs = *reinterpret_cast<const f32x4_t*>(a.bias + n);
  if (active && a.rowvec) pre.rv0 = *reinterpret_cast<const f32x4_t*>(a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + n);
#pragma unroll
  for (int jj = 0; jj < Gm::G; ++jj) {
    const int row = rslot + Gm::RS * jj, m = m0 + row;
    const bool ok = active && row < Gm::RP && m < a.M;
    pre.q1[jj] = u32x2_t{0u, 0u};
    pre.q2[jj] = u32x2_t{0u, 0u};
    if (ok && a.res1)
      pre.q1[jj] = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res1 +
                                                     (size_t)((a.res1_wrap_rows > 0 && m >= a.res1_wrap_rows) ? m - a.res1_wrap_rows : m) * a.ldres1 + n);
    if (ok && a.res2) pre.q2[jj] = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
  }
}

// Share `share` of the tile: rows [share * R, + R).  `smem`: fc_lds_bytes(BM) of LDS that nothing else uses any more; `scr`:
// the tile's FC_SCR_BYTES of scratch (apply only).  Arithmetic and summation order are those of the lean combine kernels.
template <int BM, int EDT, int S>
PP_DEVINL void fc_share_combine(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int share, int tid, unsigned long long* scr,
                                const FcPre<FcGeom<BM, S>::G>& pre, unsigned long long* fc_stp) {
  using E = E16<EDT>;
  using Gm = FcGeom<BM, S>;
  constexpr int T = 512, BN = FC_BN, SC = Gm::SC, RS = Gm::RS, RP = Gm::RP, G = Gm::G, NG = Gm::NG, NTOT = Gm::NTOT, VLD = FC_VLD;
  constexpr int SL_OFF = BM * VLD;
  const int strip = tid % SC, rslot = tid / SC;
  const int n = n_blk + strip * 4;
  const bool active = rslot < RS && n < a.N;
  const bool gns = a.gn_acc[0] != nullptr || a.gn_acc[1] != nullptr;
  const bool apply = a.gn_next_out != nullptr;
  const int ncols = min(BN, a.N - n_blk);
  const int r0 = share * Gm::R;
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + SL_OFF);
  const uint32_t slab_b = (uint32_t)a.M * (uint32_t)a.N * 4u;                  // (host-checked: splits * slab_b < 2^31)
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.workspace, slab_b * (uint32_t)S);
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4_t bs = pre.bs;
  f32x4_t p[2][G][S], rv[2];
  u32x2_t q1[2][G], q2[2][G];
  rv[0] = pre.rv0;
  rv[1] = zero4;

  auto load = [&](auto GI_) __attribute__((always_inline)) {
    constexpr int gi = decltype(GI_)::value, buf = gi & 1, pass = gi / NG, g = gi % NG;
    const int m0 = m_blk + r0 + pass * RP;
    if constexpr (g == 0 && gi > 0) {
      if (active && a.rowvec) rv[pass & 1] = *reinterpret_cast<const f32x4_t*>(a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + n);
    }
#pragma unroll
    for (int jj = 0; jj < G; ++jj) {
      const int row = rslot + RS * (g * G + jj), m = m0 + row;
      const bool ok = active && row < RP && m < a.M;
      const int voff = ok ? (m * a.N + n) * 4 : (int)PP_OOB;
#pragma unroll
      for (int s = 0; s < S; ++s)
        p[buf][jj][s] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(s * slab_b), FC_SLAB_AUX));
      if constexpr (gi == 0) {                      // (group 0: loaded before the arrival, fc_preload)
        q1[buf][jj] = pre.q1[jj];
        q2[buf][jj] = pre.q2[jj];
        continue;
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
    const int m0 = m_blk + r0 + pass * RP;
#pragma unroll
    for (int jj = 0; jj < G; ++jj) {
      const int row = rslot + RS * (g * G + jj), m = m0 + row;
      if (!(active && row < RP && m < a.M)) continue;
      f32x4_t v = p[buf][jj][0];
#pragma unroll
      for (int s = 1; s < S; ++s) v += p[buf][jj][s];
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
      if (gns) *reinterpret_cast<u32x2_t*>(smem + (r0 + pass * RP + row) * VLD + strip * 8) = o;
    }
  };
  // statistics of pass `pass`, whose finished rows are in LDS
  auto gn_phase = [&](int pass) __attribute__((always_inline)) {
    const int trow0 = r0 + pass * RP, m0 = m_blk + trow0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (16-row block, column) moments of the values as stored, rows in order: the partial sums of pp_splitk_reduce_gn_kernel,
    // so the integers the accumulators receive are the same
    for (int q = tid; q < (RP / 16) * BN; q += T) {
      const int blk = q / BN, col = q - blk * BN;
      if (col < ncols && m0 + blk * 16 < a.M) {
        const char* vp = smem + (trow0 + blk * 16) * VLD + col * 2;
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
    if (apply) {
      // the consumer's subscription: this pass's integer sums per group to the tile's scratch (the thread that moves the
      // slot to the global accumulator next, in gn_flush, writes it here first)
      const int k = tid / GN_SLOTS, gl = tid - k * GN_SLOTS;
      if (k == a.gn_next_sub) {
        const unsigned long long* sl = slots + (k * GN_SLOTS + gl) * 2;
        unsigned long long* dst = scr + ((size_t)(trow0 >> 4) * GN_SLOTS + gl) * 2;
        dst[0] = sl[0];
        dst[1] = sl[1];
      }
    }
    gn_flush(a, slots, m0, n_blk, ncols, tid);       // (batch item of row m0; clears the slots it moved)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slots free for the next pass / share
  };

  load(std::integral_constant<int, 0>{});
  FC_STAMP_SYNC(4);
  fc_static_for<NTOT>([&](auto GI_) __attribute__((always_inline)) {
    constexpr int gi = decltype(GI_)::value;
    if constexpr (gi + 1 < NTOT) load(std::integral_constant<int, gi + 1>{});
    process(GI_);
    if constexpr (gi % NG == NG - 1) {
      if (gns) gn_phase(gi / NG);
    }
  });
}

// The consumer GroupNorm (+ SiLU) of share `share`, once every share of the tile has its sums in `scr`: (scale, shift) of the
// tile's columns from the complete integer sums of the batch item -- the arithmetic of gn_fold_acc and of
// pp_splitk_reduce_gn_apply_kernel -- then this share's finished rows (still in LDS) -> gn_next_out.
template <int BM, int EDT, int S>
PP_DEVINL void fc_share_apply(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int share, int tid, const unsigned long long* scr) {
  using E = E16<EDT>;
  using Gm = FcGeom<BM, S>;
  constexpr int BN = FC_BN, SC = Gm::SC, RS = Gm::RS, RP = Gm::RP, VLD = FC_VLD;
  constexpr int SC_OFF = BM * VLD + 2 * GN_SLOTS * 2 * 8, SH_OFF = SC_OFF + BN * 4;
  float* sc_s = reinterpret_cast<float*>(smem + SC_OFF);
  float* sh_s = reinterpret_cast<float*>(smem + SH_OFF);
  const int strip = tid % SC, rslot = tid / SC;
  const int n = n_blk + strip * 4;
  const bool active = rslot < RS && n < a.N;
  const int ncols = min(BN, a.N - n_blk);
  const int hw = a.rows_per_batch;
  const bool silu = a.gn_next_silu != 0;
  for (int pass = 0; pass < Gm::NPASS; ++pass) {
    const int trow0 = share * Gm::R + pass * RP;
    const int it0 = trow0 / hw * hw;                 // first tile row of the batch item (tiles start on item boundaries)
    if (tid < ncols) {
      const int k = a.gn_next_sub, cg = a.gn_cg[k], cbase = a.gn_c0[k] + n_blk;
      const int gl = (cbase + tid) / cg - cbase / cg;
      unsigned long long si = 0ull, qi = 0ull;
      for (int e = it0 >> 4; e < (it0 + hw) >> 4; e += RP >> 4) {
        const unsigned long long* src = scr + ((size_t)e * GN_SLOTS + gl) * 2;
        si += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qi += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const double s = (double)(long long)si * (1.0 / (double)PP_GN_SUM_SCALE);
      const double q = (double)(long long)qi * (1.0 / (double)PP_GN_SQ_SCALE);
      const double cnt = (double)hw * (double)cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float meanf = (float)mean, rstdf = (float)(1.0 / sqrt(var + (double)a.gn_next_eps));
      const float sc = rstdf * a.gn_next_gamma[n_blk + tid];
      sc_s[tid] = sc;
      sh_s[tid] = a.gn_next_beta[n_blk + tid] - meanf * sc;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (active) {
      const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(sc_s + strip * 4), b0 = *reinterpret_cast<const f32x4_t*>(sh_s + strip * 4);
      for (int rr = rslot; rr < RP; rr += RS) {
        const int m = m_blk + trow0 + rr;
        if (m >= a.M) break;
        const u32x2_t v = *reinterpret_cast<const u32x2_t*>(smem + (trow0 + rr) * VLD + strip * 8);
        float r[4];
        r[0] = E::lo(v[0]) * a0[0] + b0[0]; r[1] = E::hi(v[0]) * a0[1] + b0[1];
        r[2] = E::lo(v[1]) * a0[2] + b0[2]; r[3] = E::hi(v[1]) * a0[3] + b0[3];
        if (silu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = silu_fast_f(r[j]);
        }
        u32x2_t o;
        o[0] = E::pack2(r[0], r[1]);
        o[1] = E::pack2(r[2], r[3]);
        *reinterpret_cast<u32x2_t*>((uint16_t*)a.gn_next_out + (size_t)m * a.N + n) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (scale, shift) free for the next pass / share
  }
}

template <int BM, int EDT, int S>
PP_DEVINL void fc_run(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int tile_id, int share, int tid) {
  unsigned* flag = reinterpret_cast<unsigned*>(smem + fc_flag_off(BM));
  FcPre<FcGeom<BM, S>::G> pre;
  unsigned long long* fc_stp = nullptr;
#ifdef PP_LAB
  if (a.dbg & 0x2000)
    fc_stp = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + (size_t)S * a.M * a.N * 4 +
                                                   (size_t)tile_id * FC_SCR_BYTES + 16 * GN_SLOTS * 16) + share * 8;
#endif
  FC_STAMP(0);
#ifdef PP_LAB
  // phase ablations of tools/fc_time.py (wrong results, timing only): 0x100 nothing after the slab stores, 0x200 arrive and
  // exit, 0x400 no wait for the other splits, 0x800 no combine body, 0x1000 no second arrival / apply
  if (a.dbg & 0x100) return;
#endif
  const unsigned mask = fc_arrive<S>(a, tile_id, share, tid, flag, fc_stp, [&]() __attribute__((always_inline)) {
    fc_preload<BM, S>(a, m_blk, n_blk, share, tid, pre);
  });
  if ((mask & 0xffu) == 0) return;
#ifdef PP_LAB
  if (a.dbg & 0x200) return;
#endif
  unsigned long long* scr = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + (size_t)S * a.M * a.N * 4 +
                                                                  (size_t)tile_id * FC_SCR_BYTES);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + BM * FC_VLD);
  if (tid < 2 * GN_SLOTS * 2) slots[tid] = 0ull;
#ifdef PP_LAB
  if (!(a.dbg & 0x800))
#endif
  for (int j = 0; j < S; ++j) {                      // (its own share; the last arriver: the abandoned ones as well)
    if (!(mask >> j & 1u)) continue;
    if (mask != (1u << share)) fc_preload<BM, S>(a, m_blk, n_blk, j, tid, pre);   // (the last arriver with abandoned shares to combine)
    FC_STAMP_SYNC(7);
    fc_share_combine<BM, EDT, S>(a, smem, m_blk, n_blk, j, tid, scr, pre, fc_stp);
  }
  FC_STAMP(5);
#ifdef PP_LAB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FC_STAMP(6);
#endif
  if (a.gn_next_out == nullptr) return;
#ifdef PP_LAB
  if (a.dbg & 0x1000) return;
#endif
  fc_arrive_stats<S>(a, tile_id, __builtin_popcount(mask & 0xffu), tid);
  for (int j = 0; j < S; ++j)
    if (mask >> j & 1u) fc_share_apply<BM, EDT, S>(a, smem, m_blk, n_blk, j, tid, scr);
}

// tile_id = blockIdx.x, share = blockIdx.y of the dim3(tiles, splits) launch
template <int BM, int T, int EDT>
PP_DEVINL void splitk_fused_combine(const PPGemmArgs& a, char* smem, int m_blk, int n_blk, int tile_id, int share, int splits, int tid) {
  static_assert(T == 512, "the in-kernel combine is written for the 8-wave tiles");
  // (the host admits splits = 2 | 4 | 8 with BM / splits >= 16 only: fused_combine_shape_ok)
  if (splits == 2) fc_run<BM, EDT, 2>(a, smem, m_blk, n_blk, tile_id, share, tid);
  if constexpr (BM >= 64) {
    if (splits == 4) fc_run<BM, EDT, 4>(a, smem, m_blk, n_blk, tile_id, share, tid);
  }
  if constexpr (BM >= 128) {
    if (splits == 8) fc_run<BM, EDT, 8>(a, smem, m_blk, n_blk, tile_id, share, tid);
  }
}

}  // namespace
