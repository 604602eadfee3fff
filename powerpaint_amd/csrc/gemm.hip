// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )          X: activations (rows = pixels/tokens), W: weights
//
// Design (CDNA4-first, see DESIGN.md "GEMM core"):
//   * block tile BM x 160 x 64 : every GEMM width in SD-1.5 is a multiple of 320 = 2*160, so BN = 160 wastes nothing
//     (a 128/256-wide power-of-two tile would waste 17-37 % of the MFMA work of every 320-channel layer);
//   * 4 (or 8) waves of 64 lanes, wave tile (MI*16) x 80, v_mfma_f32_16x16x32_bf16, fp32 accumulators in registers;
//   * operands are SWAPPED at the MFMA (A-operand = W rows, B-operand = X rows) so each lane ends up with 4
//     CONSECUTIVE output channels of one row -> 8-byte bf16 stores / residual loads along the contiguous axis and a
//     per-lane GEGLU pair without any cross-lane traffic;
//   * both operand tiles live in LDS as [rows][64] bf16 (128-B rows) with the 16-B slots XOR-swizzled by (row & 7):
//     ds_write_b128 / ds_read_b128 are bank-conflict free;
//   * global -> register -> LDS staging with the next tile's loads in flight during the MFMAs of the current one,
//     double-buffered LDS, ONE barrier per 64-deep K step;
//   * buffer loads with out-of-range offsets return 0: the 3x3 halo, nearest-2x upsample, stride-2 and the two-source
//     channel concat of the up blocks are pure address arithmetic in the loader -- no im2col / concat / upsample
//     tensor ever touches HBM;
//   * workgroup id -> tile mapping is XCD-aware (8 XCDs, private 4 MiB L2 each): the blocks resident on one XCD walk
//     consecutive N tiles of the same M panel, so the activation panel is fetched from HBM once per XCD.
#include "pp_common.h"
#include "gemm_gn.h"
#include "gemm_combine.h"

// Epilogue: accumulators staged through LDS in 64-row passes, full-row 16-byte stores.  (A register-direct epilogue was
// built and measured in round 3 -- parity-green, +1 % on the UNet step: profiles/r03_epilogue_ab.txt, DESIGN.md section 8;
// the code left the tree in round 4, `git log -S PP_EPI_DIRECT` finds it.)
namespace {

struct GemmDerived {
  int tiles_m, tiles_n, kt_total, kt_per_split, ctiles;
  int n_major;   // 1: consecutive tile ids walk the M tiles of one N tile (the blocks of an XCD share the W strip)
};

// Folded LayerNorm: per-row (mean, rstd) from the producer's per-N-tile (sum, sum of squares) partials.
PP_DEVINL void ln_row_moments(const PPGemmArgs& a, int m, float& mean, float& rstd) {
  const f32x2_t* p = reinterpret_cast<const f32x2_t*>(a.ln_stats) + (size_t)m * a.ln_tiles;
  float s = 0.f, q = 0.f;
  for (int t = 0; t < a.ln_tiles; ++t) {
    const f32x2_t v = p[t];
    s += v[0];
    q += v[1];
  }
  const float inv = 1.0f / (float)a.ln_dim;
  mean = s * inv;
  rstd = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + a.ln_eps);
}

// res1 of a launch whose residual holds only the first half of the batch (PPGemmArgs.res1_wrap_rows): row m reads row m mod wrap
PP_DEVINL size_t res1_row(const PPGemmArgs& a, int m) {
  return (size_t)((a.res1_wrap_rows > 0 && m >= a.res1_wrap_rows) ? m - a.res1_wrap_rows : m);
}

template <int BN, int EDT>
PP_DEVINL void epilogue4(const PPGemmArgs& a, int m, int n, f32x4_t v) {
  // v holds columns n..n+3 of row m (fp32 accumulators)
  if (a.ln_stats) {
    float mean, rstd;
    ln_row_moments(a, m, mean, rstd);
    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(a.ln_colsum + n);
    v = (v - cs * mean) * rstd;
  }
  if (a.bias) {
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(a.bias + n);
    v += b;
  }
  int bidx = 0, rin = m;
  if (a.rowvec || a.out_vt) {
    bidx = m / a.rows_per_batch;
    rin = m - bidx * a.rows_per_batch;
  }
  if (a.rowvec) {
    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(a.rowvec + (size_t)bidx * a.ld_rowvec + n);
    v += t;
  }
  v *= a.scale;
  if (a.res1) {
    const u32x2_t r = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res1 + res1_row(a, m) * a.ldres1 + n);
    v[0] += E16<EDT>::lo(r[0]); v[1] += E16<EDT>::hi(r[0]); v[2] += E16<EDT>::lo(r[1]); v[3] += E16<EDT>::hi(r[1]);
  }
  if (a.res2) {
    const u32x2_t r = *reinterpret_cast<const u32x2_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
    v[0] += E16<EDT>::lo(r[0]); v[1] += E16<EDT>::hi(r[0]); v[2] += E16<EDT>::lo(r[1]); v[3] += E16<EDT>::hi(r[1]);
  }
  if (a.act == PP_ACT_GEGLU) {
    const float o0 = v[0] * gelu_fast_f(v[2]);
    const float o1 = v[1] * gelu_fast_f(v[3]);
    *reinterpret_cast<uint32_t*>((uint16_t*)a.out + (size_t)m * a.ldo + (n >> 1)) = E16<EDT>::pack2(o0, o1);
    return;
  }
  if (a.act == PP_ACT_SILU) {
    v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
  }
  if (a.out_vt && n >= a.vt_col0) {
    const int ncols = a.N - a.vt_col0;
    uint16_t* dst = (uint16_t*)a.out_vt + ((size_t)bidx * ncols + (n - a.vt_col0)) * a.vt_ld + rin;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(size_t)j * a.vt_ld] = E16<EDT>::from_f(v[j]);
    return;
  }
  if (a.out_f32) {
    *reinterpret_cast<f32x4_t*>((float*)a.out + (size_t)m * a.ldo + n) = v;
  } else {
    u32x2_t o;
    o[0] = E16<EDT>::pack2(v[0], v[1]);
    o[1] = E16<EDT>::pack2(v[2], v[3]);
    *reinterpret_cast<u32x2_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
  }
}

template <int BM, int BN, int WM, int WN, int XMODE, int EDT>
__global__ void __launch_bounds__(WM* WN * 64, 2) pp_gemm_kernel(const PPGemmArgs a, const GemmDerived d) {
  typedef typename E16<EDT>::v8 v8_t;
  constexpr int T = WM * WN * 64;
  constexpr int MI = BM / WM / 16;
  constexpr int NI = BN / WN / 16;
  constexpr int RPP = T / 8;                      // tile rows covered per load pass
  constexpr int XP = (BM + RPP - 1) / RPP;        // 16-B pieces per thread, X tile
  constexpr int WP = (BN + RPP - 1) / RPP;        // 16-B pieces per thread, W tile
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUFBYTES = XBYTES + WBYTES;
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile/wave mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware bijective block remap (block b runs on XCD b % 8)
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // the blocks resident on one XCD take consecutive tile ids: M-major ids walk the N tiles of one M panel (the activation
  // panel is fetched into that XCD's L2 once), N-major ids the M tiles of one N strip (small-M launches, where the
  // weight strip is the big operand: every XCD would otherwise pull the whole matrix through the fabric)
  int tile_m, tile_n;
  if (d.n_major) {
    tile_n = lid / d.tiles_m;
    tile_m = lid - tile_n * d.tiles_m;
  } else {
    tile_m = lid / d.tiles_n;
    tile_n = lid - tile_m * d.tiles_n;
  }
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;
  const int split = blockIdx.y;
  const int kt_begin = split * d.kt_per_split;
  int kt_end = kt_begin + d.kt_per_split;
  if (kt_end > d.kt_total) kt_end = d.kt_total;

  // ---- loader geometry
  const int slot = tid & 7;
  const int prow = tid >> 3;                                   // row within a pass
  const int lds_piece = prow * 128 + ((slot ^ (prow & 7)) << 4);  // RPP % 8 == 0 -> same swizzle every pass

  const int ctot = a.c1 + a.c2;
  const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(
      a.x1, XMODE == PP_X_PLAIN ? (uint32_t)a.M * (uint32_t)a.ldx1 * 2u
                                : (uint32_t)a.batch * (uint32_t)a.hin * (uint32_t)a.win * (uint32_t)a.c1 * 2u);
  const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(
      a.x2 ? a.x2 : a.x1, !a.x2 ? 0u
                          : (XMODE == PP_X_PLAIN ? (uint32_t)a.M * (uint32_t)a.ldx2 * 2u
                                                 : (uint32_t)a.batch * (uint32_t)a.hin * (uint32_t)a.win * (uint32_t)a.c2 * 2u));
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.w, (uint32_t)a.N * (uint32_t)a.K * 2u);

  bool xvalid[XP];
  int xa[XP], xb[XP], xc[XP];   // PLAIN: xa = m*ldx1, xb = m*ldx2 ; CONV: xa = b*hin*win, xb = oy*stride-1, xc = ox*stride-1
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = prow + i * RPP;
    const int m = m_blk + row;
    xvalid[i] = (row < BM) && (m < a.M);
    if (XMODE == PP_X_PLAIN) {
      xa[i] = m * a.ldx1;
      xb[i] = m * a.ldx2;
      xc[i] = 0;
    } else {
      const int hw = a.hout * a.wout;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / a.wout;
      const int ox = rem - oy * a.wout;
      xa[i] = b * a.hin * a.win;
      xb[i] = oy * a.stride - 1;
      xc[i] = ox * a.stride - 1;
    }
  }
  bool wvalid[WP];
  int wo[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = prow + i * RPP;
    const int n = n_blk + row;
    wvalid[i] = (row < BN) && (n < a.N);
    wo[i] = n * a.K;
  }

  // conv tap bookkeeping (block-uniform)
  int tap = 0, cc = 0;
  if (XMODE == PP_X_CONV3X3) {
    tap = kt_begin / d.ctiles;
    cc = (kt_begin - tap * d.ctiles) * 64;
  }
  const int hv = a.up ? a.hin * 2 : a.hin;
  const int wv = a.up ? a.win * 2 : a.win;

  u32x4_t xr[XP], wr[WP];

  auto load_tile = [&](int kt) {
    const int k0 = kt * 64;
    if (XMODE == PP_X_PLAIN) {
      const bool first = k0 < a.c1;
      const __amdgpu_buffer_rsrc_t rs = first ? rs_x1 : rs_x2;
      const int kk = (first ? k0 : k0 - a.c1) + slot * 8;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const uint32_t off = xvalid[i] ? (uint32_t)((first ? xa[i] : xb[i]) + kk) * 2u : PP_OOB;
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      const bool first = cc < a.c1;
      const __amdgpu_buffer_rsrc_t rs = first ? rs_x1 : rs_x2;
      const int csrc = first ? a.c1 : a.c2;
      const int kk = (first ? cc : cc - a.c1) + slot * 8;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const int iy = xb[i] + ky, ix = xc[i] + kx;
        const bool ok = xvalid[i] && (unsigned)iy < (unsigned)hv && (unsigned)ix < (unsigned)wv;
        const int sy = a.up ? (iy >> 1) : iy;
        const int sx = a.up ? (ix >> 1) : ix;
        const uint32_t off = ok ? (uint32_t)((xa[i] + sy * a.win + sx) * csrc + kk) * 2u : PP_OOB;
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
      cc += 64;
      if (cc == ctot) { cc = 0; ++tap; }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const uint32_t off = wvalid[i] ? (uint32_t)(wo[i] + k0 + slot * 8) * 2u : PP_OOB;
      wr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    char* xs = smem + buf * BUFBYTES;
    char* ws = xs + XBYTES;
#pragma unroll
    for (int i = 0; i < XP; ++i)
      if ((XP * RPP == BM) || (prow + i * RPP < BM)) *reinterpret_cast<u32x4_t*>(xs + lds_piece + i * RPP * 128) = xr[i];
#pragma unroll
    for (int i = 0; i < WP; ++i)
      if ((WP * RPP == BN) || (prow + i * RPP < BN)) *reinterpret_cast<u32x4_t*>(ws + lds_piece + i * RPP * 128) = wr[i];
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = base + (lane & 15), k-slot = ks*4 + (lane >> 4), swizzled by row & 7
  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int xrow0 = wm * (MI * 16) + frow;
  const int wrow0 = wn * (NI * 16) + frow;
  // (row & 7) is invariant to +16*i, so the swizzle term is the same for every fragment of this lane
  const int fsw = frow & 7;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) load_tile(kt + 1);

    const char* xs = smem + cur * BUFBYTES;
    const char* ws = xs + XBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 4 + fk) ^ fsw) << 4;
      v8_t xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        xf[mi] = *reinterpret_cast<const v8_t*>(xs + (xrow0 + mi * 16) * 128 + so);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        wf[ni] = *reinterpret_cast<const v8_t*>(ws + (wrow0 + ni * 16) * 128 + so);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = E16<EDT>::mfma16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }

    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds rows n = .. + 4*(lane>>4) + {0..3} (A-operand = W), col m = .. + (lane & 15)
  const int m_base = m_blk + wm * (MI * 16) + (lane & 15);
  const int n_base = n_blk + wn * (NI * 16) + 4 * (lane >> 4);
  if (gridDim.y > 1) {
    float* ws = a.workspace + (size_t)split * a.M * a.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m_base + mi * 16;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n_base + ni * 16;
        if (m < a.M && n < a.N) *reinterpret_cast<f32x4_t*>(ws + (size_t)m * a.N + n) = acc[ni][mi];
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_base + mi * 16;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n_base + ni * 16;
      if (m < a.M && n < a.N) epilogue4<BN, EDT>(a, m, n, acc[ni][mi]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: multi-stage LDS-direct pipeline + LDS-staged coalesced epilogue.
//   * operand tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write pass); the LDS
//     image of a wave-instruction is lane-linear (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE
//     k-slot and undone by the same XOR on the fragment read;
//   * NS stages: the loads of tiles t+1 .. t+NS-1 stay in flight across the (raw) barrier while tile t is consumed;
//     counted `s_waitcnt vmcnt(P*(NS-2))`, ONE s_barrier per 64-deep K step;
//   * addressing: per-lane byte offsets are computed once per (3x3 tap, source tensor) -- NOT per K tile; the walk
//     along K is a wave-uniform SGPR offset.  Invalid lanes (halo, row tail) carry an out-of-range voffset and dead
//     tiles a zero-sized descriptor: both make the DMA write zeros, and every wave always issues exactly P loads per
//     tile so the vmcnt arithmetic is uniform;
//   * epilogue: accumulators are staged through LDS (fp32, 64-row passes, row stride padded by 16 B => conflict-free
//     ds_write_b128) and read back row-major, so bias / residual loads and the output stores are 16 B per lane along
//     the contiguous axis: 320-byte full-row segments instead of 8-byte pieces scattered over 16 rows.
//   * EPI selects the epilogue family, as separate instantiations so that each keeps its register budget (64x160 and the
//     8-wave 128x160 must stay <= 128 VGPRs for 4 waves/SIMD):  0 = standard;  1 = standard + folded LayerNorm (row
//     moments out / mean-rstd correction in);  2 = GEGLU in registers (+ optional folded LayerNorm);  4 = standard +
//     GroupNorm statistics of the output (gn_acc).  1 and 2 are PLAIN-only and prefetch their epilogue operands
//     into LDS.
//   * PP ("ping-pong", 8-wave tiles, NS >= 3): the two waves of every SIMD run half a K step apart -- waves 0-3 issue the
//     MFMAs of tile t while waves 4-7 read their fragments of tile t and issue the refill DMAs, then the roles swap
//     (two raw barriers per K step, group 1 enters the loop one barrier late).  In the lock-step loop both waves of a
//     SIMD sit in barrier / bookkeeping / LDS-latency at the same time and the matrix pipe idles ~1/3 of every K step.
template <int BM, int BN, int WM, int WN, int XMODE, int NS, int EPI, bool DMAI, bool PP, int EDT>
__global__ void __launch_bounds__(WM* WN * 64, (PP ? 2 : (WM * WN == 8 && (BM / WM / 16) * (BN / WN / 16) <= 10 ? 4 : 2)))
pp_gemm_kernel_v2(const PPGemmArgs a, const GemmDerived d) {   // 2nd bound = waves / SIMD the register budget must allow:
  // four only for the 8-wave 128x160x2 tile (two co-resident workgroups = 16 waves per CU); the 4-wave 64x160 and
  // 128x160 tiles are limited to two workgroups per CU by their LDS, i.e. two waves per SIMD whatever the registers
  typedef typename E16<EDT>::v8 v8_t;
  constexpr bool LNF = EPI == 1 || EPI == 2 || EPI == 5;   // (5: softmax over 80-column groups, PP_ACT_SOFTMAX80)
  constexpr bool GNS = EPI == 4;
  constexpr int T = WM * WN * 64;
  constexpr int MI = BM / WM / 16;
  constexpr int NI = BN / WN / 16;
  constexpr int RPP = T / 8;
  constexpr int XP = BM / RPP;
  constexpr int WP = (BN + RPP - 1) / RPP;
  constexpr int P = XP + WP;
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
  constexpr int EPI_ROWS = 64;                    // rows per epilogue pass
  constexpr int EPI_LD = BN * 4 + 16;             // fp32 row stride in bytes (+16: bank spread for ds_write_b128)
  static_assert(BM % RPP == 0, "X tile must be whole passes");
  static_assert(P * (NS - 1) < 64, "vmcnt field");
  constexpr int LN_OFF = EPI_ROWS * EPI_LD;       // folded-LN (mean, rstd) per block row
  constexpr int RS_OFF = LN_OFF + BM * 8;         // producer row-moment partials [EPI_ROWS][BN/8] float2
  static_assert(RS_OFF + EPI_ROWS * (BN / 8) * 8 <= NS * STAGE, "epilogue staging must fit in the pipeline stages");
  static_assert(MI * 16 <= EPI_ROWS && EPI_ROWS % (MI * 16) == 0, "wave rows vs epilogue pass");
  // LNF: epilogue operands (bias, LN column sums, LN row-moment partials of this block's rows) are DMA'd into LDS behind
  // the pipeline stages at kernel start, so the short-K epilogue never waits on a global load.
  constexpr int LN_TMAX = 4;                      // row-moment partials per row kept in LDS (C <= 640); more -> global
  constexpr int PRE_B = NS * STAGE, PRE_C = PRE_B + 1024, PRE_M = PRE_C + 1024;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // the blocks resident on one XCD take consecutive tile ids: M-major ids walk the N tiles of one M panel (the activation
  // panel is fetched into that XCD's L2 once), N-major ids the M tiles of one N strip (small-M launches, where the
  // weight strip is the big operand: every XCD would otherwise pull the whole matrix through the fabric)
  int tile_m, tile_n;
  if (d.n_major) {
    tile_n = lid / d.tiles_m;
    tile_m = lid - tile_n * d.tiles_m;
  } else {
    tile_m = lid / d.tiles_n;
    tile_n = lid - tile_m * d.tiles_n;
  }
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;
  const int split = blockIdx.y;
  const int kt_begin = split * d.kt_per_split;
  int kt_end = kt_begin + d.kt_per_split;
  if (kt_end > d.kt_total) kt_end = d.kt_total;
  const int nkt = kt_end - kt_begin;
  // ablation switches (tools/gemm_ablate.py): 1 no refill, 2 no MFMA, 4 no epilogue, 8 no s_setprio.  The ping-pong loop
  // honours them only in a -DPP_GEMM_DBG build (seven branches per K step otherwise ride in its read phase).
#if defined(PP_LAB) && defined(PP_GEMM_DBG)
  const int dbg = a.dbg;
#elif defined(PP_LAB)
  const int dbg = PP ? 0 : a.dbg;
#else
  constexpr int dbg = 0;      // shipping build: one code path (PPGemmArgs.dbg is ignored)
#endif

  // lane -> (row within the wave's 8-row strip, k-slot it must FETCH so that its lane-linear LDS position is swizzled)
  const int lrow = lane >> 3;
  const int kslot = (lane & 7) ^ lrow;          // (row & 7) == lrow because every strip starts at a multiple of 8
  const int prow = wave * 8 + lrow;             // tile row of pass 0

  const int ctot = a.c1 + a.c2;
  const uint32_t xbytes1 = XMODE == PP_X_PLAIN ? (uint32_t)a.M * (uint32_t)a.ldx1 * 2u
                                               : (uint32_t)a.batch * (uint32_t)a.hin * (uint32_t)a.win * (uint32_t)a.c1 * 2u;
  const uint32_t xbytes2 = !a.x2 ? 0u
                           : (XMODE == PP_X_PLAIN ? (uint32_t)a.M * (uint32_t)a.ldx2 * 2u
                                                  : (uint32_t)a.batch * (uint32_t)a.hin * (uint32_t)a.win * (uint32_t)a.c2 * 2u);
  const uint32_t wbytes = (uint32_t)a.N * (uint32_t)a.K * 2u;
  // (ABI v20) one weight matrix per batch item (the cross-attention operands folded per prompt): a tile lies inside ONE
  // item (host-checked: rows_per_batch % BM == 0)
  const int bitem = a.w_batch_stride > 0 ? m_blk / a.rows_per_batch : 0;
  const void* const w_base = reinterpret_cast<const uint16_t*>(a.w) + (size_t)bitem * (size_t)a.w_batch_stride;

  // ---- per-lane offsets.  PLAIN: vx1/vx2 = byte offset of (row m, k-slot) in source 1 / 2 (OOB if m >= M), fixed.
  //      CONV : pixel coordinates kept in (xa, xb, xc); vx1 recomputed when the (tap, source) pair changes.
  int vx1[XP], vx2[XP];   // (int: the LDS-DMA builtin takes a signed voffset)
  int xa[XP], xb[XP], xc[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int m = m_blk + prow + i * RPP;
    const bool ok = m < a.M;
    if (XMODE == PP_X_PLAIN) {
      vx1[i] = ok ? (m * a.ldx1 + kslot * 8) * 2 : (int)PP_OOB;
      vx2[i] = ok ? (m * a.ldx2 + kslot * 8) * 2 : (int)PP_OOB;
      xa[i] = xb[i] = xc[i] = 0;
    } else {
      const int hw = a.hout * a.wout;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / a.wout;
      const int ox = rem - oy * a.wout;
      xa[i] = ok ? b * a.hin * a.win : -1;      // -1 marks a dead row
      xb[i] = oy * a.stride - 1;
      xc[i] = ox * a.stride - 1;
      vx1[i] = (int)PP_OOB;
      vx2[i] = (int)PP_OOB;
    }
  }
  int vw[WP];
  int wlds[WP];   // wave-uniform LDS byte offset of the strip inside the W tile
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    // a wave whose strip falls beyond BN in the last pass re-issues its previous strip (identical bytes to the
    // identical LDS address) so that every wave has exactly WP loads in flight per tile
    const int strip = (wave * 8 + i * RPP < BN) ? i : i - 1;
    const int n = n_blk + prow + strip * RPP;
    vw[i] = (n < a.N) ? (n * a.K + kslot * 8) * 2 : (int)PP_OOB;
    wlds[i] = (wave * 8 + strip * RPP) * 128;
  }

  // conv K walk: taps 0..8 over the c1 + c2 channels of the (concatenated) input, then tap 9 = the optional 1x1 tail
  // over c3 + c4 channels of (x3, x4) at the output pixel (ResnetBlock2D.conv_shortcut merged into conv2)
  int tap = 0, cc = 0;
  bool retap = true;
  if (XMODE == PP_X_CONV3X3) {
    tap = kt_begin / d.ctiles;
    if (tap > 9) tap = 9;
    cc = (kt_begin - tap * d.ctiles) * 64;
  }
  const uint64_t px1 = reinterpret_cast<uint64_t>(a.x1), px2 = reinterpret_cast<uint64_t>(a.x2);
  const uint64_t px3 = XMODE == PP_X_CONV3X3 ? reinterpret_cast<uint64_t>(a.x3) : 0ull;
  const uint64_t px4 = XMODE == PP_X_CONV3X3 ? reinterpret_cast<uint64_t>(a.x4) : 0ull;
  const int pc1 = a.c1, pc2 = a.c2, pc3 = XMODE == PP_X_CONV3X3 ? a.c3 : 0, pc4 = XMODE == PP_X_CONV3X3 ? a.c4 : 0;
  uint64_t cur_src = px1;            // source of the current (tap, channel range): updated only where it changes
  uint32_t cur_bytes = 0u;
  int cur_cA = pc1, cur_c0 = 0;
  const uint32_t xbytes3 = (XMODE == PP_X_CONV3X3 && a.x3) ? (uint32_t)a.M * (uint32_t)a.c3 * 2u : 0u;
  const uint32_t xbytes4 = (XMODE == PP_X_CONV3X3 && a.x4) ? (uint32_t)a.M * (uint32_t)a.c4 * 2u : 0u;
  const int hv = a.up ? a.hin * 2 : a.hin;
  const int wv = a.up ? a.win * 2 : a.win;

  // (always_inline: once this lambda is outlined its by-reference captures force the whole kernel-argument struct into
  //  scratch memory -- 700 B per lane and a 2.7x slower kernel)
  // One tile refill = issue_begin (descriptors, the conv (tap, source) refresh, K walk) + P single-instruction pieces
  // (XP of the X tile, WP of the W tile), so that the main loop can spread the pieces over its MFMA burst.
  __amdgpu_buffer_rsrc_t is_rsx, is_rsw;
  int is_sox = 0, is_sow = 0, is_second = 0;
  char* is_xs = smem;
  char* is_ws = smem;
  auto issue_begin = [&](int kt, int stage) __attribute__((always_inline)) {
    is_xs = smem + stage * STAGE;
    is_ws = is_xs + XBYTES;
    const bool live = kt < kt_end;
    const int k0 = kt * 64;
    if (XMODE == PP_X_PLAIN) {
      const bool first = k0 < a.c1;
      is_rsx = make_rsrc(first ? a.x1 : a.x2, live ? (first ? xbytes1 : xbytes2) : 0u);
      is_sox = (first ? k0 : k0 - a.c1) * 2;
      is_second = first ? 0 : 1;
    } else {
      const bool tail = tap >= 9;                          // 1x1 phase over (x3, x4) at the output pixel
      if (live && (retap || cc == 0 || cc == cur_cA)) {    // (tap, source) changed: refresh source + per-lane offsets
        // plain if / else chains on purpose: a 4-way select over (x1..x4) is turned into a private-memory lookup table
        // by LLVM (scratch traffic in the hot loop, 2.7x slower conv)
        const bool first = cc < (tail ? pc3 : pc1);   // (a split-K slice may start in the middle of a source)
        int csrc;
        if (!tail) {
          cur_cA = pc1;
          if (first) { cur_src = px1; cur_bytes = xbytes1; csrc = pc1; cur_c0 = 0; }
          else { cur_src = px2; cur_bytes = xbytes2; csrc = pc2; cur_c0 = pc1; }
        } else {
          cur_cA = pc3;
          if (first) { cur_src = px3; cur_bytes = xbytes3; csrc = pc3; cur_c0 = 0; }
          else { cur_src = px4; cur_bytes = xbytes4; csrc = pc4; cur_c0 = pc3; }
        }
        const int ky = tail ? 1 : tap / 3, kx = tail ? 1 : tap - (tap / 3) * 3;      // tail = centre tap geometry
#pragma unroll
        for (int i = 0; i < XP; ++i) {
          const int iy = xb[i] + ky, ix = xc[i] + kx;
          const bool ok = xa[i] >= 0 && (unsigned)iy < (unsigned)hv && (unsigned)ix < (unsigned)wv;
          const int sy = a.up ? (iy >> 1) : iy;
          const int sx = a.up ? (ix >> 1) : ix;
          vx1[i] = ok ? ((xa[i] + sy * a.win + sx) * csrc + kslot * 8) * 2 : (int)PP_OOB;
        }
        retap = false;
      }
      is_rsx = make_rsrc(reinterpret_cast<const void*>(cur_src), live ? cur_bytes : 0u);
      is_sox = (cc - cur_c0) * 2;
      if (live) {
        cc += 64;
        if (!tail && cc == ctot) { cc = 0; ++tap; retap = true; }
      }
    }
    is_rsw = make_rsrc(w_base, live ? wbytes : 0u);
    is_sow = k0 * 2;
  };
  auto issue_piece = [&](int i) __attribute__((always_inline)) {      // i is a compile-time constant at every call site
    if (i < XP) {
      // (local copy: passing the captured array element straight to the builtin makes clang drop the HOST stub)
      const int ii = i < XP ? i : 0;
      const int vo = (XMODE == PP_X_PLAIN && is_second) ? vx2[ii] : vx1[ii];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_rsx, (lds_ptr_t)(is_xs + (wave * 8 + ii * RPP) * 128), 16, vo, is_sox,
                                               0, 0);
    } else {
      const int j = i - XP < WP ? i - XP : 0;
      const int vo = vw[j], lo = wlds[j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_rsw, (lds_ptr_t)(is_ws + lo), 16, vo, is_sow, 0, 0);
    }
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    issue_begin(kt, stage);
#pragma unroll
    for (int i = 0; i < P; ++i) issue_piece(i);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int xrow0 = wm * (MI * 16) + frow;
  const int wrow0 = wn * (NI * 16) + frow;
  const int fsw = frow & 7;

  const bool ln = LNF && a.ln_stats != nullptr;
  const bool ln_lds = ln && a.ln_tiles <= LN_TMAX;
  if (LNF) {   // older than every tile load -> covered by the counted vmcnt waits below
    const size_t voff = (size_t)bitem * (size_t)a.vec_batch_stride;     // (per-item bias / column sums: PP_ACT_SOFTMAX80)
    if (wave == 0) {
      const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bias ? (const void*)(a.bias + voff) : (const void*)a.w, a.bias ? (uint32_t)a.N * 4u : 0u);
      const int vo = n_blk * 4 + lane * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(smem + PRE_B), 16, vo, 0, 0, 0);
    }
    if (wave == 1) {
      const __amdgpu_buffer_rsrc_t rc = make_rsrc(ln ? (const void*)(a.ln_colsum + voff) : (const void*)a.w, ln ? (uint32_t)a.N * 4u : 0u);
      const int vo = n_blk * 4 + lane * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (lds_ptr_t)(smem + PRE_C), 16, vo, 0, 0, 0);
    }
    if (ln_lds) {
      const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.ln_stats, (uint32_t)a.M * (uint32_t)a.ln_tiles * 8u);
      const int chunks = (BM * a.ln_tiles * 8 + 1023) >> 10;
      for (int c = wave; c < chunks; c += WM * WN) {
        const int vo = m_blk * a.ln_tiles * 8 + c * 1024 + lane * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_ptr_t)(smem + PRE_M + c * 1024), 16, vo, 0, 0, 0);
      }
    }
  }
  if constexpr (PP) {
    static_assert(WM * WN == 8 && NS >= 3, "ping-pong: two groups of four waves (one wave of each per SIMD), >= 3 stages");
    const int grp = wave >> 2;
    // ---- issue side: the K walk as SEGMENTS of whole 64-deep tiles (one source tensor at one tap).  The per-tile work
    //      is P DMA pieces + three scalar adds; descriptors and per-lane offsets are rebuilt only on a segment change
    //      (a wave-uniform branch), and tiles past the end of this block's K range use zero-sized descriptors (the DMA
    //      writes nothing useful but every wave keeps exactly P loads per tile in flight: uniform vmcnt arithmetic).
    // segment = (tap, second source?): taps 0..8 over (x1, x2), tap 9 = the 1x1 tail over (x3, x4), tap 10 = past the end;
    // PLAIN walks (x1, x2) as "tap 9".  Plain scalar state + two-way selects only: an index-driven formulation is turned
    // into a private-memory lookup table by LLVM (scratch traffic in the loop).
    const int n1 = pc1 >> 6, n2 = pc2 >> 6, n3 = pc3 >> 6, n4 = pc4 >> 6;
    int tap_i = 0, seg_left = 0, tiles_left = nkt;
    bool second = false;
    int sox = 0, sow = kt_begin * 128;
    const void* const qx1 = reinterpret_cast<const void*>(px1);
    const void* const qw = w_base;
    const int up_ = a.up, win_ = a.win;
    __amdgpu_buffer_rsrc_t rsx = make_rsrc(qx1, 0u), rsw = make_rsrc(qw, wbytes);
    int vx[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) vx[i] = (int)PP_OOB;
    // The walk is written as MACROS over the kernel's locals, not lambdas: in a closure the by-reference captures are
    // pointers, `second ? n2 : n1` becomes a load through a SELECTED closure field, the closure (and with it every
    // captured variable) can no longer be promoted to registers, and the loop runs out of private memory.
#define PP_SEG_LEN() (XMODE == PP_X_PLAIN ? (tap_i > 9 ? 0 : (second ? n2 : n1)) \
                                          : (tap_i < 9 ? (second ? n2 : n1) : (tap_i == 9 ? (second ? n4 : n3) : 0)))
#define PP_SEG_STEP()                     \
  do {                                    \
    if (!second) second = true;           \
    else { second = false; ++tap_i; }     \
  } while (0)
    // enter the segment (tap_i, second) at its tile CHUNK0: skip absent sources (at most x2 -> x3 -> x4 in a row), then
    // rebuild the X descriptor and the per-lane offsets; past the end: zero-sized descriptors
#define PP_SEG_ENTER(CHUNK0)                                                                                         \
  do {                                                                                                               \
    int chunk0_ = (CHUNK0);                                                                                          \
    if (tap_i <= 9 && PP_SEG_LEN() == 0) { PP_SEG_STEP(); chunk0_ = 0; }                                             \
    if (tap_i <= 9 && PP_SEG_LEN() == 0) { PP_SEG_STEP(); chunk0_ = 0; }                                             \
    if (tap_i <= 9 && PP_SEG_LEN() == 0) { PP_SEG_STEP(); chunk0_ = 0; }                                             \
    if (tap_i > 9 || tiles_left <= 0) {                                                                              \
      rsx = make_rsrc(qx1, 0u);                                                                                      \
      rsw = make_rsrc(qw, 0u);                                                                                       \
      seg_left = 1 << 30;                                                                                            \
    } else {                                                                                                         \
      int len_ = PP_SEG_LEN() - chunk0_;                                                                             \
      if (len_ > tiles_left) len_ = tiles_left;                                                                      \
      seg_left = len_;                                                                                               \
      tiles_left -= len_;                                                                                            \
      sox = chunk0_ * 128;                                                                                           \
      uint64_t srcv_;                                                                                                \
      uint32_t srcb_;                                                                                                \
      if (XMODE == PP_X_PLAIN) {                                                                                     \
        if (!second) { srcv_ = px1; srcb_ = xbytes1; }                                                               \
        else { srcv_ = px2; srcb_ = xbytes2; }                                                                       \
        rsx = make_rsrc(reinterpret_cast<const void*>(srcv_), srcb_);                                                \
        _Pragma("unroll") for (int i = 0; i < XP; ++i) vx[i] = second ? vx2[i] : vx1[i];                             \
      } else {                                                                                                       \
        const bool tail_ = tap_i >= 9;                                                                               \
        int csrc_;                                                                                                   \
        if (!tail_) {                                                                                                \
          if (!second) { srcv_ = px1; srcb_ = xbytes1; csrc_ = pc1; }                                                \
          else { srcv_ = px2; srcb_ = xbytes2; csrc_ = pc2; }                                                        \
        } else {                                                                                                     \
          if (!second) { srcv_ = px3; srcb_ = xbytes3; csrc_ = pc3; }                                                \
          else { srcv_ = px4; srcb_ = xbytes4; csrc_ = pc4; }                                                        \
        }                                                                                                            \
        rsx = make_rsrc(reinterpret_cast<const void*>(srcv_), srcb_);                                                \
        const int ky_ = tail_ ? 1 : tap_i / 3, kx_ = tail_ ? 1 : tap_i - (tap_i / 3) * 3; /* tail = centre tap */    \
        _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                             \
          const int iy = xb[i] + ky_, ix = xc[i] + kx_;                                                              \
          const bool ok = xa[i] >= 0 && (unsigned)iy < (unsigned)hv && (unsigned)ix < (unsigned)wv;                  \
          const int sy = up_ ? (iy >> 1) : iy;                                                                       \
          const int sx = up_ ? (ix >> 1) : ix;                                                                       \
          vx[i] = ok ? ((xa[i] + sy * win_ + sx) * csrc_ + kslot * 8) * 2 : (int)PP_OOB;                             \
        }                                                                                                            \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)
    // one tile refill into stage STG: (segment change, rarely) + P DMA pieces + three scalar adds; split so that the
    // main loop can interleave the pieces with its fragment reads
#define PP_ISSUE_BEGIN(STG)                                                                                          \
  do {                                                                                                               \
    if (seg_left == 0) {                                                                                             \
      PP_SEG_STEP();                                                                                                 \
      PP_SEG_ENTER(0);                                                                                               \
    }                                                                                                                \
    is_xs = smem + (STG) * STAGE;                                                                                    \
    is_ws = is_xs + XBYTES;                                                                                          \
  } while (0)
#define PP_PIECE(I)                                                                                                  \
  do {                                                                                                               \
    if ((I) < XP) {                                                                                                  \
      const int ii_ = (I) < XP ? (I) : 0;                                                                            \
      const int vo = vx[ii_];                                                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(is_xs + (wave * 8 + ii_ * RPP) * 128), 16, vo, sox, 0, 0); \
    } else {                                                                                                         \
      const int jj_ = (I) - XP < WP ? (I) - XP : 0;                                                                  \
      const int vo = vw[jj_], lo = wlds[jj_];                                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(is_ws + lo), 16, vo, sow, 0, 0);                     \
    }                                                                                                                \
  } while (0)
#define PP_ISSUE_END()                                                                                               \
  do {                                                                                                               \
    sox += 128;                                                                                                      \
    sow += 128;                                                                                                      \
    --seg_left;                                                                                                      \
  } while (0)
#define PP_ISSUE(STG)                                                                                                \
  do {                                                                                                               \
    PP_ISSUE_BEGIN(STG);                                                                                             \
    _Pragma("unroll") for (int i = 0; i < P; ++i) PP_PIECE(i);                                                       \
    PP_ISSUE_END();                                                                                                  \
  } while (0)
    {   // position of this block's first K tile
      int k = kt_begin;
      if (XMODE == PP_X_PLAIN) {
        tap_i = 9;
        if (k >= n1) { second = true; k -= n1; }
      } else {
        tap_i = k / d.ctiles;
        if (tap_i > 9) tap_i = 9;
        k -= tap_i * d.ctiles;
        const int nf = tap_i < 9 ? n1 : n3;
        if (k >= nf) { second = true; k -= nf; }
      }
      PP_SEG_ENTER(k);
    }
#pragma nounroll
    for (int s = 0; s < NS - 1; ++s) PP_ISSUE(s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (NS - 2)) : "memory");   // this wave's pieces of tile 0 have landed
    if (grp == 1) asm volatile("s_barrier" ::: "memory");                 // group 1 runs one phase behind group 0
    int stage = 0;
    for (int t = 0; t < nkt; ++t) {
      asm volatile("s_barrier" ::: "memory");   // X: (everyone's) tile t is in LDS; the partner group left its read phase
      // ---- read phase (the partner wave on this SIMD is in its MFMA phase).  The P refill DMAs of the stage tile t-1
      //      occupied (both groups have left it) are spread between the fragment reads: issued as one burst behind
      //      the reads they queue on the CU's address unit (~16 cycles each, 4 waves x P) AFTER the LDS has served the
      //      reads, and the read phase (reads 425 + DMA issue 425 cycles measured) outlasts the partner's 640 MFMA
      //      cycles; interleaved, the address unit and the LDS work side by side.
      const char* xs = smem + stage * STAGE;
      const char* ws = xs + XBYTES;
      int nstage = stage + (NS - 1);
      if (nstage >= NS) nstage -= NS;
      const bool refill = !(dbg & 1);
      if (refill) PP_ISSUE_BEGIN(nstage);
      v8_t xf[2][MI], wf[2][NI];
      constexpr int NR = 2 * (MI + NI);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int ks = r / (MI + NI), j = r % (MI + NI);
        const int so = ((ks * 4 + fk) ^ fsw) << 4;
        if (j < NI) wf[ks][j < NI ? j : 0] = *reinterpret_cast<const v8_t*>(ws + (wrow0 + j * 16) * 128 + so);
        else xf[ks][j >= NI ? j - NI : 0] = *reinterpret_cast<const v8_t*>(xs + (xrow0 + (j - NI) * 16) * 128 + so);
        // piece k goes after read number ceil((k + 1) * NR / (P + 1))
        if (refill) {
#pragma unroll
          for (int k = 0; k < P; ++k)
            if (((k + 1) * NR + P) / (P + 1) == r + 1) {
              __builtin_amdgcn_sched_barrier(0);
              PP_PIECE(k);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
      if (refill) PP_ISSUE_END();
      // Y: this wave's pieces of tile t+1 have landed, its fragments of tile t are in registers
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P * (NS - 2)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA phase (the partner wave reads / refills)
      if (!(dbg & 2)) {
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[ni][mi] = E16<EDT>::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
#undef PP_ISSUE
#undef PP_ISSUE_BEGIN
#undef PP_PIECE
#undef PP_ISSUE_END
#undef PP_SEG_ENTER
#undef PP_SEG_LEN
#undef PP_SEG_STEP
  } else {
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(kt_begin + s, s);

  int stage = 0;
  for (int t = 0; t < nkt; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (NS - 2)) : "memory");   // this wave's loads of tile t have landed
    asm volatile("s_barrier" ::: "memory");                               // everyone's have; everyone left tile t-1
    int nstage = stage + (NS - 1);
    if (nstage >= NS) nstage -= NS;
    if constexpr (DMAI) issue_begin(kt_begin + t + NS - 1, nstage);       // the P DMA pieces ride inside the MFMA burst
    else issue((dbg & 1) ? kt_end : kt_begin + t + NS - 1, nstage);       // refill the stage consumed last iteration

    const char* xs = smem + stage * STAGE;
    const char* ws = xs + XBYTES;
    if (!(dbg & 2)) {
      // all 2*(MI+NI) fragment reads of the tile are issued up front (the LDS latency of k-step 1 hides behind the MFMAs
      // of k-step 0 instead of serialising read -> wait -> 4 MFMAs -> read ...), then 2*MI*NI MFMAs back to back
      v8_t xf[2][MI], wf[2][NI];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = ((ks * 4 + fk) ^ fsw) << 4;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wf[ks][ni] = *reinterpret_cast<const v8_t*>(ws + (wrow0 + ni * 16) * 128 + so);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xf[ks][mi] = *reinterpret_cast<const v8_t*>(xs + (xrow0 + mi * 16) * 128 + so);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);      // co-resident waves in their load / epilogue phase yield the issue slots
      if constexpr (DMAI) {
        // All waves of the block leave the barrier together; when each then fires its P DMA instructions back to back,
        // 8 x P of them queue on the CU's one texture-address unit (~16 cycles per wave-wide 16-byte load) and every
        // wave sits in VMEM issue for up to ~900 cycles before its first MFMA.  Spread over the burst -- one piece
        // every `per` MFMAs -- the address unit keeps its pace and the matrix pipe starts right after the barrier.
        constexpr int NM = 2 * MI * NI, per = NM / P > 0 ? NM / P : 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              acc[ni][mi] = E16<EDT>::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
              const int idx = (ks * NI + ni) * MI + mi + 1;
              if (idx % per == 0 && idx / per <= P) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(idx / per - 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
        for (int i = NM / per; i < P; ++i) issue_piece(i);
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[ni][mi] = E16<EDT>::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  }   // !PP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (zero-sized) tail prefetches before LDS is reused
  if (dbg & 4) {
    if (acc[0][0][0] == 123.456f) ((float*)a.out)[0] = 1.f;   // keep the accumulators live
    return;
  }

  const bool splitk = gridDim.y > 1;
  // (V^T output exists for plain GEMMs without GroupNorm statistics only: elsewhere the staged epilogue is dead code)
  constexpr bool VT_POSSIBLE = XMODE == PP_X_PLAIN && !GNS;
  const bool vt_blk = VT_POSSIBLE && !splitk && a.out_vt && n_blk >= a.vt_col0;
  const bool geglu = EPI == 2;
  f32x2_t ln_mr = {0.f, 0.f};
  if (ln && tid < BM && m_blk + tid < a.M) {
    float mean, rstd;
    if (ln_lds) {
      const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(smem + PRE_M) + tid * a.ln_tiles;
      float sm = 0.f, sq = 0.f;
      for (int t = 0; t < a.ln_tiles; ++t) { sm += pm[t][0]; sq += pm[t][1]; }
      const float inv = 1.0f / (float)a.ln_dim;
      mean = sm * inv;
      rstd = rsqrtf(fmaxf(sq * inv - mean * mean, 0.f) + a.ln_eps);
    } else {
      ln_row_moments(a, m_blk + tid, mean, rstd);
    }
    ln_mr = f32x2_t{mean, rstd};
  }
  // ================= epilogue: 64-row passes through LDS (V^T blocks; every block in a -DPP_EPI_STAGED lab build) ====
  const int my_pass = (wm * (MI * 16)) / EPI_ROWS;
  const int my_row0 = (wm * (MI * 16)) % EPI_ROWS;
  const bool rs_out = LNF && a.row_stats_out != nullptr && !splitk && !vt_blk && !geglu;
  if constexpr (EPI == 2) {
    // GEGLU in the MFMA register layout: the weight rows are interleaved (h0,h1,g0,g1), so each lane's accumulator quad is
    // two complete (value, gate) pairs -> bias / LN correction / GELU happen in registers and only the bf16 result (a
    // quarter of the fp32 tile) is staged through LDS for 16-byte row-contiguous stores.  One pass, one barrier pair.
    constexpr int GLD = BN + 16;                       // staged row: BN/2 bf16 + 16 B pad
    static_assert(BM * GLD + BM * 8 <= NS * STAGE, "GEGLU staging must fit in the pipeline stages");
    constexpr int GLN = BM * GLD;                      // (mean, rstd) table behind the staged tile
    asm volatile("s_barrier" ::: "memory");            // main loop done with LDS
    if (ln) {
      if (tid < BM) *reinterpret_cast<f32x2_t*>(smem + GLN + tid * 8) = ln_mr;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int q4 = 4 * (lane >> 4);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nl = wn * (NI * 16) + ni * 16 + q4;    // column inside the block tile
      f32x4_t bs, cs;
      bs = *reinterpret_cast<const f32x4_t*>(smem + PRE_B + nl * 4);   // (prefetched; zero where absent / n >= N)
      cs = *reinterpret_cast<const f32x4_t*>(smem + PRE_C + nl * 4);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (MI * 16) + mi * 16 + (lane & 15);
        f32x4_t v = acc[ni][mi];
        if (ln) {
          const f32x2_t mr = *reinterpret_cast<const f32x2_t*>(smem + GLN + row * 8);
          v = (v - cs * mr[0]) * mr[1];
        }
        v += bs;
        *reinterpret_cast<uint32_t*>(smem + row * GLD + nl) = E16<EDT>::pack2(v[0] * gelu_fast_f(v[2]), v[1] * gelu_fast_f(v[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    constexpr int CH = BN / 16;                        // 16-byte chunks per staged row
    for (int q = tid; q < BM * CH; q += T) {
      const int row = q / CH, c = q - row * CH;
      const int m = m_blk + row, n = n_blk + c * 16;
      if (m < a.M && n < a.N)
        *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + (n >> 1)) =
            *reinterpret_cast<const u32x4_t*>(smem + row * GLD + c * 16);
    }
    return;
  } else if constexpr (EPI == 5) {
    // PP_ACT_SOFTMAX80: out[m][80 g .. 80 g + 79] = softmax over the group's 80 logits in the exp2 domain (the caller's
    // weights carry log2 e; -inf bias entries mask padding columns).  The fp32 tile (bias / folded-LayerNorm correction
    // applied in the accumulator layout) is staged once; one thread per (row, group) then runs the three passes.
    static_assert(BN % 80 == 0, "whole groups per tile");
    constexpr int SLN = BM * EPI_LD;                   // (mean, rstd) table behind the staged tile
    static_assert(SLN + BM * 8 <= NS * STAGE, "softmax staging must fit in the pipeline stages");
    asm volatile("s_barrier" ::: "memory");            // main loop done with LDS
    if (ln) {
      if (tid < BM) *reinterpret_cast<f32x2_t*>(smem + SLN + tid * 8) = ln_mr;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int q4 = 4 * (lane >> 4);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nl = wn * (NI * 16) + ni * 16 + q4;
      const f32x4_t bs = *reinterpret_cast<const f32x4_t*>(smem + PRE_B + nl * 4);   // (prefetched; zero where absent / n >= N)
      const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(smem + PRE_C + nl * 4);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (MI * 16) + mi * 16 + (lane & 15);
        f32x4_t v = acc[ni][mi];
        if (ln) {
          const f32x2_t mr = *reinterpret_cast<const f32x2_t*>(smem + SLN + row * 8);
          v = (v - cs * mr[0]) * mr[1];
        }
        v += bs;
        *reinterpret_cast<f32x4_t*>(smem + row * EPI_LD + nl * 4) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    constexpr int G = BN / 80;
    for (int q = tid; q < BM * G; q += T) {
      const int row = q % BM, gi = q / BM;
      const int m = m_blk + row, n0 = n_blk + gi * 80;
      if (m >= a.M || n0 >= a.N) continue;
      const char* src = smem + row * EPI_LD + gi * 320;
      float mx = -INFINITY;
#pragma unroll 5
      for (int j = 0; j < 20; ++j) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + j * 16);
        mx = fmaxf(fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], v[2])), v[3]);
      }
      if (mx == -INFINITY) mx = 0.f;                   // (a fully masked group stays all zero instead of NaN)
      float sum = 0.f;
#pragma unroll 5
      for (int j = 0; j < 20; ++j) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + j * 16);
        sum += (__builtin_amdgcn_exp2f(v[0] - mx) + __builtin_amdgcn_exp2f(v[1] - mx)) +
               (__builtin_amdgcn_exp2f(v[2] - mx) + __builtin_amdgcn_exp2f(v[3] - mx));
      }
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
      uint16_t* dst = (uint16_t*)a.out + (size_t)m * a.ldo + n0;
#pragma unroll 2
      for (int j = 0; j < 10; ++j) {
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src + j * 32), v1 = *reinterpret_cast<const f32x4_t*>(src + j * 32 + 16);
        u32x4_t o;
        o[0] = E16<EDT>::pack2(__builtin_amdgcn_exp2f(v0[0] - mx) * inv, __builtin_amdgcn_exp2f(v0[1] - mx) * inv);
        o[1] = E16<EDT>::pack2(__builtin_amdgcn_exp2f(v0[2] - mx) * inv, __builtin_amdgcn_exp2f(v0[3] - mx) * inv);
        o[2] = E16<EDT>::pack2(__builtin_amdgcn_exp2f(v1[0] - mx) * inv, __builtin_amdgcn_exp2f(v1[1] - mx) * inv);
        o[3] = E16<EDT>::pack2(__builtin_amdgcn_exp2f(v1[2] - mx) * inv, __builtin_amdgcn_exp2f(v1[3] - mx) * inv);
        *reinterpret_cast<u32x4_t*>(dst + j * 8) = o;
      }
    }
    return;
  } else {
  // GroupNorm statistics (EPI 4): per-thread column moments of the stored values, folded into the groups' accumulators
  // once per batch item the tile touches -- once per workgroup when a batch item spans whole tiles, else per 64-row pass
  float gcs[8], gcq[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { gcs[jj] = 0.f; gcq[jj] = 0.f; }
  const bool gn_per_pass = GNS && (a.rows_per_batch % BM) != 0;
  auto gn_fold_block = [&](int m_first, bool first_call) {
    constexpr int EC = BN / 8, ER = T / EC;
    const int c8 = tid % EC, r0 = tid / EC;
    const int n = n_blk + c8 * 8;
    // per-thread column moments -> LDS [row-thread][column] (over the staged tile, which everyone has finished reading)
    // -> the 160 column threads fold the ER row-threads in fixed order and add into the groups' integer slots
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + RS_OFF);
    if (first_call && tid < 2 * GN_SLOTS * 2) slots[tid] = 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (r0 < ER && n < a.N) {
      float* dstp = reinterpret_cast<float*>(smem) + ((size_t)r0 * BN + c8 * 8) * 2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        *reinterpret_cast<f32x4_t*>(dstp + 4 * jj) = f32x4_t{gcs[2 * jj], gcq[2 * jj], gcs[2 * jj + 1], gcq[2 * jj + 1]};
        gcs[2 * jj] = gcq[2 * jj] = gcs[2 * jj + 1] = gcq[2 * jj + 1] = 0.f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int ncols = min(BN, a.N - n_blk);
    if (tid < ncols) {
      float sm = 0.f, sq = 0.f;
#pragma unroll 5
      for (int r = 0; r < ER; ++r) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(smem + ((size_t)r * BN + tid) * 8);
        sm += v[0];
        sq += v[1];
      }
      gn_column(a, slots, n_blk, tid, sm, sq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (m_first < a.M) gn_flush(a, slots, m_first, n_blk, ncols, tid);
  };
#pragma unroll 1
  for (int pass = 0; pass < BM / EPI_ROWS; ++pass) {
    asm volatile("s_barrier" ::: "memory");            // LDS free: main loop (pass 0) / previous read-out finished
    if (ln && pass == 0 && tid < BM) *reinterpret_cast<f32x2_t*>(smem + LN_OFF + tid * 8) = ln_mr;
    if (my_pass == pass) {
      // lane holds rows n = wn*NI*16 + ni*16 + 4*(lane>>4) + {0..3} of column m = my_row0 + mi*16 + (lane & 15)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          *reinterpret_cast<f32x4_t*>(smem + (my_row0 + mi * 16 + (lane & 15)) * EPI_LD +
                                      (wn * (NI * 16) + ni * 16 + 4 * (lane >> 4)) * 4) = acc[ni][mi];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int m0 = m_blk + pass * EPI_ROWS;
    if (vt_blk) {
      // transposed (V^T) output: thread = (column n, 8 consecutive rows)
      const int ncols = a.N - a.vt_col0;
      for (int q = tid; q < BN * (EPI_ROWS / 8); q += T) {
        const int col = q % BN, rg = q / BN;
        const int n = n_blk + col, m = m0 + rg * 8;
        if (n >= a.N || m >= a.M) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(smem + (rg * 8 + j) * EPI_LD + col * 4);
        if (ln) {
          const float cs = *reinterpret_cast<const float*>(smem + PRE_C + col * 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x2_t mr = *reinterpret_cast<const f32x2_t*>(smem + LN_OFF + (pass * EPI_ROWS + rg * 8 + j) * 8);
            v[j] = (v[j] - mr[0] * cs) * mr[1];
          }
        }
        const float bsv = LNF ? *reinterpret_cast<const float*>(smem + PRE_B + col * 4) : (a.bias ? a.bias[n] : 0.f);
        const int bidx = m / a.rows_per_batch, rin = m - bidx * a.rows_per_batch;
        uint16_t* dst = (uint16_t*)a.out_vt + ((size_t)bidx * ncols + (n - a.vt_col0)) * a.vt_ld;
        if ((a.rows_per_batch & 7) == 0 && (a.vt_ld & 7) == 0 && m + 8 <= a.M) {
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = E16<EDT>::pack2((v[2 * j] + bsv) * a.scale, (v[2 * j + 1] + bsv) * a.scale);
          *reinterpret_cast<u32x4_t*>(dst + rin) = o;
        } else {
          for (int j = 0; j < 8 && m + j < a.M; ++j) {
            const int bj = (m + j) / a.rows_per_batch, rj = (m + j) - bj * a.rows_per_batch;
            ((uint16_t*)a.out_vt)[((size_t)bj * ncols + (n - a.vt_col0)) * a.vt_ld + rj] = E16<EDT>::from_f((v[j] + bsv) * a.scale);
          }
        }
      }
    } else {
      // thread = fixed 8-column strip, rows tid/EC + j*ER.  All residual loads of the pass are issued BEFORE the math so
      // their latencies overlap (a load -> use -> store loop would serialise ~1 us of memory latency per piece).
      constexpr int EC = BN / 8, ER = T / EC, EP = (EPI_ROWS + ER - 1) / ER;
      const int c8 = tid % EC, r0 = tid / EC;
      const int n = n_blk + c8 * 8;
      if (r0 < ER && n < a.N) {
        if (splitk) {
#pragma unroll
          for (int j = 0; j < EP; ++j) {
            const int row = r0 + j * ER, m = m0 + row;
            if (row < EPI_ROWS && m < a.M) {
              float* wsp = a.workspace + ((size_t)split * a.M + m) * a.N + n;
              *reinterpret_cast<f32x4_t*>(wsp) = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32);
              *reinterpret_cast<f32x4_t*>(wsp + 4) = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32 + 16);
            }
          }
        } else {
          u32x4_t r1[EP], r2[EP];
          // (PPGemmArgs.res1_wrap_rows, a multiple of the 64-row pass: one wave-uniform shift per pass)
          const int r1shift = (a.res1_wrap_rows > 0 && m0 >= a.res1_wrap_rows) ? a.res1_wrap_rows : 0;
#pragma unroll
          for (int j = 0; j < EP; ++j) {
            const int row = r0 + j * ER, m = m0 + row;
            const bool ok = row < EPI_ROWS && m < a.M;
            r1[j] = (ok && a.res1) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + (size_t)(m - r1shift) * a.ldres1 + n)
                                   : u32x4_t{0u, 0u, 0u, 0u};
            r2[j] = (ok && a.res2) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n)
                                   : u32x4_t{0u, 0u, 0u, 0u};
          }
          f32x4_t bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f};
          f32x4_t cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
          if (LNF) {
            bs0 = *reinterpret_cast<const f32x4_t*>(smem + PRE_B + c8 * 32);
            bs1 = *reinterpret_cast<const f32x4_t*>(smem + PRE_B + c8 * 32 + 16);
            cs0 = *reinterpret_cast<const f32x4_t*>(smem + PRE_C + c8 * 32);
            cs1 = *reinterpret_cast<const f32x4_t*>(smem + PRE_C + c8 * 32 + 16);
          } else if (a.bias) {
            bs0 = *reinterpret_cast<const f32x4_t*>(a.bias + n);
            bs1 = *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
          }
#pragma unroll
          for (int j = 0; j < EP; ++j) {
            const int row = r0 + j * ER, m = m0 + row;
            if (row < EPI_ROWS && m < a.M) {
              f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32);
              f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32 + 16);
              if (ln) {
                const f32x2_t mr = *reinterpret_cast<const f32x2_t*>(smem + LN_OFF + (pass * EPI_ROWS + row) * 8);
                v0 = (v0 - cs0 * mr[0]) * mr[1];
                v1 = (v1 - cs1 * mr[0]) * mr[1];
              }
              v0 += bs0;
              v1 += bs1;
              if (a.rowvec) {
                const float* rv = a.rowvec + (size_t)(m / a.rows_per_batch) * a.ld_rowvec + n;
                v0 += *reinterpret_cast<const f32x4_t*>(rv);
                v1 += *reinterpret_cast<const f32x4_t*>(rv + 4);
              }
              v0 *= a.scale;
              v1 *= a.scale;
              v0[0] += E16<EDT>::lo(r1[j][0]) + E16<EDT>::lo(r2[j][0]); v0[1] += E16<EDT>::hi(r1[j][0]) + E16<EDT>::hi(r2[j][0]);
              v0[2] += E16<EDT>::lo(r1[j][1]) + E16<EDT>::lo(r2[j][1]); v0[3] += E16<EDT>::hi(r1[j][1]) + E16<EDT>::hi(r2[j][1]);
              v1[0] += E16<EDT>::lo(r1[j][2]) + E16<EDT>::lo(r2[j][2]); v1[1] += E16<EDT>::hi(r1[j][2]) + E16<EDT>::hi(r2[j][2]);
              v1[2] += E16<EDT>::lo(r1[j][3]) + E16<EDT>::lo(r2[j][3]); v1[3] += E16<EDT>::hi(r1[j][3]) + E16<EDT>::hi(r2[j][3]);
              if (a.act == PP_ACT_SILU) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { v0[jj] = silu_f(v0[jj]); v1[jj] = silu_f(v1[jj]); }
              }
              if (a.out_f32) {
                float* op = (float*)a.out + (size_t)m * a.ldo + n;
                *reinterpret_cast<f32x4_t*>(op) = v0;
                *reinterpret_cast<f32x4_t*>(op + 4) = v1;
              } else {
                u32x4_t o;
                o[0] = E16<EDT>::pack2(v0[0], v0[1]); o[1] = E16<EDT>::pack2(v0[2], v0[3]);
                o[2] = E16<EDT>::pack2(v1[0], v1[1]); o[3] = E16<EDT>::pack2(v1[2], v1[3]);
                *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
                if constexpr (!LNF) {        // (conv_in: never a folded-LayerNorm launch; those kernels have no registers to spare)
                  if (a.out_dup_rows > 0)    // the twin half of a CFG batch (PPGemmArgs.out_dup_rows): same values, second copy
                    *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + ((size_t)m + a.out_dup_rows) * a.ldo + n) = o;
                }
                if (GNS) {   // per-column moments of the values as stored, over this thread's rows of the pass
#pragma unroll
                  for (int jj = 0; jj < 4; ++jj) {
                    const float lo = E16<EDT>::lo(o[jj]), hi = E16<EDT>::hi(o[jj]);
                    gcs[2 * jj] += lo; gcq[2 * jj] += lo * lo;
                    gcs[2 * jj + 1] += hi; gcq[2 * jj + 1] += hi * hi;
                  }
                }
                if (rs_out) {   // moments of the values as stored (bf16-rounded), for the LayerNorm folded downstream
                  float sm = 0.f, sq = 0.f;
#pragma unroll
                  for (int jj = 0; jj < 4; ++jj) {
                    const float lo = E16<EDT>::lo(o[jj]), hi = E16<EDT>::hi(o[jj]);
                    sm += lo + hi;
                    sq += lo * lo + hi * hi;
                  }
                  *reinterpret_cast<f32x2_t*>(smem + RS_OFF + (row * EC + c8) * 8) = f32x2_t{sm, sq};
                }
              }
            }
          }
        }
      }
      if (gn_per_pass && !splitk) gn_fold_block(m0, pass == 0);
      if (rs_out) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < EPI_ROWS && m0 + tid < a.M) {
          const int nc = min(EC, (a.N - n_blk) >> 3);
          float sm = 0.f, sq = 0.f;
          for (int c = 0; c < nc; ++c) {
            const f32x2_t v = *reinterpret_cast<const f32x2_t*>(smem + RS_OFF + (tid * EC + c) * 8);
            sm += v[0];
            sq += v[1];
          }
          const int tiles_n = (a.N + BN - 1) / BN;
          *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)(m0 + tid) * tiles_n + n_blk / BN) * 2) = f32x2_t{sm, sq};
        }
      }
    }
  }
  if (GNS && !splitk && !gn_per_pass) gn_fold_block(m_blk, true);
  if constexpr (EPI == 0 && PP) {
    // (ABI v21) the split-K combine by the workgroup that arrives last at its tile (gemm_combine.h); a.tile_ctr is set by
    // the host only where every split of a tile runs on one XCD and the epilogue is the lean combine's
    if (splitk && a.tile_ctr) {
      static_assert(fc_lds_bytes(BM) <= NS * STAGE, "fused combine staging must fit in the pipeline stages");
      splitk_fused_combine<BM, T, EDT>(a, smem, m_blk, n_blk, blockIdx.x, blockIdx.y, gridDim.y, tid);
    }
  }
  }   // EPI != 2
}

// Deterministic split-K combine + epilogue: thread = (row, 8 columns); the <= 8 partial slabs are read with all loads
// in flight at once (a `for s: v += load` loop would serialise one memory latency per split), summed in slab order.
// LEAN = the common conv / linear epilogue (bias, per-batch row vector, scale, two residuals, bf16 store) as straight,
// short code: this kernel runs ~38 times per UNet forward with a cold instruction cache, where its duration (~30 us in
// the rocprof trace against ~8 us back to back) is dominated by fetching its own instructions.
template <bool LEAN, int EDT>
__global__ void __launch_bounds__(256) pp_splitk_reduce_kernel(const PPGemmArgs a, int splits) {
  const int n8 = a.N >> 3;
  const long long total = (long long)a.M * n8;
  const size_t slab = (size_t)a.M * a.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / n8);
    const int n = (int)(i - (long long)m * n8) * 8;
    const float* src = a.workspace + (size_t)m * a.N + n;
    f32x4_t p0[8], p1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < splits) {
        p0[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab);
        p1[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab + 4);
      } else {
        p0[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        p1[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (LEAN) {
      u32x4_t r1 = {0u, 0u, 0u, 0u}, r2 = {0u, 0u, 0u, 0u};
      if (a.res1) r1 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + res1_row(a, m) * a.ldres1 + n);
      if (a.res2) r2 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
      f32x4_t v0 = p0[0], v1 = p1[0];
#pragma unroll
      for (int s = 1; s < 8; ++s) { v0 += p0[s]; v1 += p1[s]; }
      if (a.bias) {
        v0 += *reinterpret_cast<const f32x4_t*>(a.bias + n);
        v1 += *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
      }
      if (a.rowvec) {
        const float* rv = a.rowvec + (size_t)(m / a.rows_per_batch) * a.ld_rowvec + n;
        v0 += *reinterpret_cast<const f32x4_t*>(rv);
        v1 += *reinterpret_cast<const f32x4_t*>(rv + 4);
      }
      v0 *= a.scale;
      v1 *= a.scale;
      v0[0] += E16<EDT>::lo(r1[0]) + E16<EDT>::lo(r2[0]); v0[1] += E16<EDT>::hi(r1[0]) + E16<EDT>::hi(r2[0]);
      v0[2] += E16<EDT>::lo(r1[1]) + E16<EDT>::lo(r2[1]); v0[3] += E16<EDT>::hi(r1[1]) + E16<EDT>::hi(r2[1]);
      v1[0] += E16<EDT>::lo(r1[2]) + E16<EDT>::lo(r2[2]); v1[1] += E16<EDT>::hi(r1[2]) + E16<EDT>::hi(r2[2]);
      v1[2] += E16<EDT>::lo(r1[3]) + E16<EDT>::lo(r2[3]); v1[3] += E16<EDT>::hi(r1[3]) + E16<EDT>::hi(r2[3]);
      u32x4_t o;
      o[0] = E16<EDT>::pack2(v0[0], v0[1]); o[1] = E16<EDT>::pack2(v0[2], v0[3]);
      o[2] = E16<EDT>::pack2(v1[0], v1[1]); o[3] = E16<EDT>::pack2(v1[2], v1[3]);
      *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
    } else {
      f32x4_t v0 = p0[0], v1 = p1[0];
#pragma unroll
      for (int s = 1; s < 8; ++s) { v0 += p0[s]; v1 += p1[s]; }
      epilogue4<160, EDT>(a, m, n, v0);
      epilogue4<160, EDT>(a, m, n + 4, v1);
    }
  }
}

// Lean combine that also accumulates GroupNorm statistics: one block = 16 rows x 160 columns, 320 threads, thread =
// (row, 8-column strip) -> every slab load of the block is in flight at once; the finished values go through an LDS
// tile, the 160 column threads fold the 16 rows and feed the groups' integer slots (gn_column / gn_flush).
template <int EDT>
__global__ void __launch_bounds__(320) pp_splitk_reduce_gn_kernel(const PPGemmArgs a, int splits, int tiles_n) {
  constexpr int BN = 160, ROWS = 16, EC = BN / 8;
  __shared__ __attribute__((aligned(16))) float tile[ROWS][BN + 4];
  __shared__ unsigned long long slots[2 * GN_SLOTS * 2];
  const int tid = threadIdx.x;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * ROWS, n_blk = tile_n * BN;
  const int c8 = tid % EC, row = tid / EC;
  const int m = m0 + row, n = n_blk + c8 * 8;
  if (tid < 2 * GN_SLOTS * 2) slots[tid] = 0ull;
  f32x4_t w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
  if (m < a.M && n < a.N) {
    const size_t slab = (size_t)a.M * a.N;
    const float* src = a.workspace + (size_t)m * a.N + n;
    f32x4_t p0[8], p1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < splits) {
        p0[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab);
        p1[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab + 4);
      } else {
        p0[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        p1[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    u32x4_t r1 = {0u, 0u, 0u, 0u}, r2 = {0u, 0u, 0u, 0u};
    if (a.res1) r1 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + res1_row(a, m) * a.ldres1 + n);
    if (a.res2) r2 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
    f32x4_t v0 = p0[0], v1 = p1[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) { v0 += p0[s]; v1 += p1[s]; }
    if (a.bias) {
      v0 += *reinterpret_cast<const f32x4_t*>(a.bias + n);
      v1 += *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
    }
    if (a.rowvec) {
      const float* rv = a.rowvec + (size_t)(m / a.rows_per_batch) * a.ld_rowvec + n;
      v0 += *reinterpret_cast<const f32x4_t*>(rv);
      v1 += *reinterpret_cast<const f32x4_t*>(rv + 4);
    }
    v0 *= a.scale;
    v1 *= a.scale;
    v0[0] += E16<EDT>::lo(r1[0]) + E16<EDT>::lo(r2[0]); v0[1] += E16<EDT>::hi(r1[0]) + E16<EDT>::hi(r2[0]);
    v0[2] += E16<EDT>::lo(r1[1]) + E16<EDT>::lo(r2[1]); v0[3] += E16<EDT>::hi(r1[1]) + E16<EDT>::hi(r2[1]);
    v1[0] += E16<EDT>::lo(r1[2]) + E16<EDT>::lo(r2[2]); v1[1] += E16<EDT>::hi(r1[2]) + E16<EDT>::hi(r2[2]);
    v1[2] += E16<EDT>::lo(r1[3]) + E16<EDT>::lo(r2[3]); v1[3] += E16<EDT>::hi(r1[3]) + E16<EDT>::hi(r2[3]);
    u32x4_t o;
    o[0] = E16<EDT>::pack2(v0[0], v0[1]); o[1] = E16<EDT>::pack2(v0[2], v0[3]);
    o[2] = E16<EDT>::pack2(v1[0], v1[1]); o[3] = E16<EDT>::pack2(v1[2], v1[3]);
    *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
    w0 = f32x4_t{E16<EDT>::lo(o[0]), E16<EDT>::hi(o[0]), E16<EDT>::lo(o[1]), E16<EDT>::hi(o[1])};
    w1 = f32x4_t{E16<EDT>::lo(o[2]), E16<EDT>::hi(o[2]), E16<EDT>::lo(o[3]), E16<EDT>::hi(o[3])};
  }
  *reinterpret_cast<f32x4_t*>(&tile[row][c8 * 8]) = w0;      // rows / columns past the edge contribute zeros
  *reinterpret_cast<f32x4_t*>(&tile[row][c8 * 8 + 4]) = w1;
  __syncthreads();
  const int ncols = min(BN, a.N - n_blk);
  if (tid < ncols) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float v = tile[r][tid];
      sm += v;
      sq += v * v;
    }
    gn_column(a, slots, n_blk, tid, sm, sq);
  }
  __syncthreads();
  gn_flush(a, slots, m0, n_blk, ncols, tid);
}

// The lean combine of a launch whose CONSUMER is a GroupNorm (+ SiLU) apply, at the levels where one workgroup can own a
// whole (batch item, group) population: hw = rows_per_batch <= 256 rows (16 x 16 and 8 x 8 latents), single-tensor norm.
// Tile = all hw rows of one batch item x 40 columns -- 40 = lcm(8, channels per group) for SD-1.5's 10 / 20 / 40-channel
// groups, so a tile holds whole groups and B x N / 40 workgroups fill the chip (8 x 32 at the 8x8 level; a 160-column tile
// version with 64 workgroups looping over the rows was 0.9 % SLOWER on the step than the two launches it replaced).
// The tile's fixed-point (sum, sum of squares) ARE the groups' statistics -- folded per 16-row block in the order of
// pp_splitk_reduce_gn_kernel, so the integers the accumulators receive are the same -- and mean / rstd / scale / shift
// follow gn_fold_acc bit for bit; the workgroup then normalises its own finished values (kept as the 16-bit words it
// stored to `out`) into gn_next_out.  One launch less per such norm (6.5 .. 12 us each).
template <int EDT>
__global__ void __launch_bounds__(640) pp_splitk_reduce_gn_apply_kernel(const PPGemmArgs a, int splits, int tiles_n, int rows_pass) {
  using E = E16<EDT>;
  constexpr int BN = 40, EC = BN / 8, RMAX = 128;
  __shared__ __attribute__((aligned(16))) char vals[256 * BN * 2];       // [hw][BN] 16-bit words of the finished tile
  __shared__ __attribute__((aligned(16))) float tile[RMAX][BN + 4];
  __shared__ unsigned long long slots[2 * GN_SLOTS * 2];
  __shared__ __attribute__((aligned(16))) float sc_s[BN], sh_s[BN];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / tiles_n, tile_n = blockIdx.x - b * tiles_n;
  const int hw = a.rows_per_batch, n_blk = tile_n * BN;
  const int c8 = tid % EC, row = tid / EC;                 // blockDim.x = rows_pass * EC
  const int n = n_blk + c8 * 8;
  for (int i = tid; i < 2 * GN_SLOTS * 2; i += blockDim.x) slots[i] = 0ull;
  const size_t slab = (size_t)a.M * a.N;
  for (int r0 = 0; r0 < hw; r0 += rows_pass) {
    const int m = b * hw + r0 + row;
    const float* src = a.workspace + (size_t)m * a.N + n;
    f32x4_t p0[8], p1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < splits) {
        p0[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab);
        p1[s] = *reinterpret_cast<const f32x4_t*>(src + s * slab + 4);
      } else {
        p0[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        p1[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    u32x4_t r1 = {0u, 0u, 0u, 0u}, r2 = {0u, 0u, 0u, 0u};
    if (a.res1) r1 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + res1_row(a, m) * a.ldres1 + n);
    if (a.res2) r2 = *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n);
    f32x4_t v0 = p0[0], v1 = p1[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) { v0 += p0[s]; v1 += p1[s]; }
    if (a.bias) {
      v0 += *reinterpret_cast<const f32x4_t*>(a.bias + n);
      v1 += *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
    }
    if (a.rowvec) {
      const float* rv = a.rowvec + (size_t)b * a.ld_rowvec + n;
      v0 += *reinterpret_cast<const f32x4_t*>(rv);
      v1 += *reinterpret_cast<const f32x4_t*>(rv + 4);
    }
    v0 *= a.scale;
    v1 *= a.scale;
    v0[0] += E::lo(r1[0]) + E::lo(r2[0]); v0[1] += E::hi(r1[0]) + E::hi(r2[0]);
    v0[2] += E::lo(r1[1]) + E::lo(r2[1]); v0[3] += E::hi(r1[1]) + E::hi(r2[1]);
    v1[0] += E::lo(r1[2]) + E::lo(r2[2]); v1[1] += E::hi(r1[2]) + E::hi(r2[2]);
    v1[2] += E::lo(r1[3]) + E::lo(r2[3]); v1[3] += E::hi(r1[3]) + E::hi(r2[3]);
    u32x4_t o;
    o[0] = E::pack2(v0[0], v0[1]); o[1] = E::pack2(v0[2], v0[3]);
    o[2] = E::pack2(v1[0], v1[1]); o[3] = E::pack2(v1[2], v1[3]);
    *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
    *reinterpret_cast<u32x4_t*>(vals + ((size_t)(r0 + row) * BN + c8 * 8) * 2) = o;
    *reinterpret_cast<f32x4_t*>(&tile[row][c8 * 8]) = f32x4_t{E::lo(o[0]), E::hi(o[0]), E::lo(o[1]), E::hi(o[1])};
    *reinterpret_cast<f32x4_t*>(&tile[row][c8 * 8 + 4]) = f32x4_t{E::lo(o[2]), E::hi(o[2]), E::lo(o[3]), E::hi(o[3])};
    __syncthreads();
    // (16-row block, column) pairs: the partial sums of pp_splitk_reduce_gn_kernel, in its order
    if (tid < (rows_pass >> 4) * BN) {
      const int blk = tid / BN, col = tid - blk * BN;
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = tile[blk * 16 + r][col];
        sm += v;
        sq += v * v;
      }
      gn_column(a, slots, n_blk, col, sm, sq);
    }
    __syncthreads();
  }
  // the consumer's (scale, shift) of the tile's 40 columns from its complete slots: the arithmetic of gn_fold_acc
  if (tid < BN) {
    const int k = a.gn_next_sub, cg = a.gn_cg[k], cbase = a.gn_c0[k] + n_blk;
    const int gl = (cbase + tid) / cg - cbase / cg;
    const double s = (double)(long long)slots[(k * GN_SLOTS + gl) * 2] * (1.0 / (double)PP_GN_SUM_SCALE);
    const double q = (double)(long long)slots[(k * GN_SLOTS + gl) * 2 + 1] * (1.0 / (double)PP_GN_SQ_SCALE);
    const double cnt = (double)hw * (double)cg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstdf = (float)(1.0 / sqrt(var + (double)a.gn_next_eps));
    const float sc = rstdf * a.gn_next_gamma[n_blk + tid];
    sc_s[tid] = sc;
    sh_s[tid] = a.gn_next_beta[n_blk + tid] - meanf * sc;
  }
  __syncthreads();
  gn_flush(a, slots, b * hw, n_blk, BN, tid);               // (the global accumulators: any second consumer reads them)
  const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(sc_s + c8 * 8), a1 = *reinterpret_cast<const f32x4_t*>(sc_s + c8 * 8 + 4);
  const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sh_s + c8 * 8), b1 = *reinterpret_cast<const f32x4_t*>(sh_s + c8 * 8 + 4);
  const bool silu = a.gn_next_silu != 0;
  for (int r0 = 0; r0 < hw; r0 += rows_pass) {
    const int m = b * hw + r0 + row;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(vals + ((size_t)(r0 + row) * BN + c8 * 8) * 2);
    float r[8];
    r[0] = E::lo(v[0]) * a0[0] + b0[0]; r[1] = E::hi(v[0]) * a0[1] + b0[1];
    r[2] = E::lo(v[1]) * a0[2] + b0[2]; r[3] = E::hi(v[1]) * a0[3] + b0[3];
    r[4] = E::lo(v[2]) * a1[0] + b1[0]; r[5] = E::hi(v[2]) * a1[1] + b1[1];
    r[6] = E::lo(v[3]) * a1[2] + b1[2]; r[7] = E::hi(v[3]) * a1[3] + b1[3];
    if (silu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = silu_fast_f(r[j]);
    }
    u32x4_t o;
    o[0] = E::pack2(r[0], r[1]); o[1] = E::pack2(r[2], r[3]);
    o[2] = E::pack2(r[4], r[5]); o[3] = E::pack2(r[6], r[7]);
    *reinterpret_cast<u32x4_t*>((uint16_t*)a.gn_next_out + (size_t)m * a.N + n) = o;
  }
}

// the LEAN combine handles: no activation, bf16 output, no transposed / GEGLU / folded-LN epilogue, 16-byte aligned rows
bool reduce_lean_ok(const PPGemmArgs& a) {
  return a.act == PP_ACT_NONE && !a.out_f32 && !a.out_vt && !a.ln_stats && a.ldo % 8 == 0 &&
         (!a.res1 || a.ldres1 % 8 == 0) && (!a.res2 || a.ldres2 % 8 == 0);
}

struct Choice {
  int tile, splitk;
};
bool gemm_pingpong();
bool gemm_n_major();

// v2 (LDS-direct pipeline + staged 16-byte epilogue) needs 16-byte aligned rows on every tensor the epilogue touches
bool v2_ok(const PPGemmArgs& a) {
  if (a.N % 8) return false;
  if (a.act == PP_ACT_GEGLU && (a.N % 16 || a.ldo % 8)) return false;
  if (a.out_f32 ? (a.ldo % 4) : (a.ldo % 8)) return false;
  if (a.res1 && a.ldres1 % 8) return false;
  if (a.res2 && a.ldres2 % 8) return false;
  if (a.out_vt && (a.vt_col0 % 160 || a.res1 || a.res2 || a.rowvec)) return false;
  return true;
}

Choice choose(const PPGemmArgs& a) {
  // Heuristic distilled from tools/gemm_sweep.py --cold on MI355X (profiles/r01_gemm_sweep_cold.txt): every candidate
  // is timed behind a 1 GiB memset, i.e. with L2 / MALL / instruction cache as cold as inside the real ~350-launch step
  // (hot back-to-back sweeps overrate split-K: its combine kernel costs ~8 us hot and ~31 us in the pipeline, and
  // underrate the 3-stage pipelines, which ride out HBM latency better).  Unit = blocks per CU of the 256-CU chip:
  //   >= 2 blocks of 128 rows per CU : 8-wave 128x160 (two co-resident blocks), or 256x160 x 3 stages for long K;
  //   ~ 1 block per CU               : 128x160 x 3 stages, split-K 2 only for very long conv K;
  //   less                           : 64x160 x 3 stages when that fills the chip, else split-K over 256-row tiles.
  // Tensors the 16-byte staged epilogue cannot address fall back to the register-staged v1 kernel.
  // per-item weight matrices / the group softmax: 64-row tiles (a tile inside one batch item down to the 8x8 level), one pass
  if (a.w_batch_stride > 0 || a.act == PP_ACT_SOFTMAX80) return Choice{32, 1};
  Choice c{a.tile, a.splitk};
  const int tn = (a.N + 159) / 160;
  auto blocks = [&](int bm) { return ((a.M + bm - 1) / bm) * tn; };
  const bool conv = a.x_mode == PP_X_CONV3X3;
  const int nb256 = blocks(256), nb128 = blocks(128), nb64 = blocks(64), kt = a.K / 64;
  if (c.tile == PP_TILE_AUTO) {
    int sk = 1;
    if (!v2_ok(a)) {
      c.tile = (nb128 >= 256) ? PP_TILE_128x160 : PP_TILE_64x160;
      while (blocks(c.tile == PP_TILE_128x160 ? 128 : 64) * sk < 224 && kt / (sk * 2) >= 12 && sk < 8) sk *= 2;
    } else if (nb128 >= 512) {
      c.tile = (kt >= 20 && nb256 >= 256) ? 33 : 24;
    } else if (nb128 >= 256) {
      if (conv && kt > 140) { c.tile = 33; sk = 2; }
      else c.tile = a.out_vt ? 21 : 31;
    } else if (a.K <= 2880) {
      c.tile = 32;
      while (nb64 * sk < 128 && kt / (sk * 2) >= 24 && sk < 8) sk *= 2;
    } else if (nb64 >= 256 && kt <= 100) {
      // (round 5, tools/tile_probe.py, cold in a graph) FF2 . proj_out of the 16x16 level, M = 2048 x N = 1280 x K = 6400:
      // 128-row tiles x 2 K splits 54 us incl. the combine against 60 - 66 for one pass of 64-row tiles
      if (kt >= 80 && nb128 * 2 >= 256 && !conv && !a.row_stats_out && !a.out_vt && a.act == PP_ACT_NONE) { c.tile = 31; sk = 2; }
      else c.tile = 32;
    } else if (nb256 >= 32) {
      c.tile = 33;
      while (nb256 * sk < 256 && sk < 8) sk *= 2;
    } else {
      c.tile = 31;
      while (nb128 * sk < 256 && kt / (sk * 2) >= 4 && sk < 8) sk *= 2;
    }
    if (c.splitk <= 0) c.splitk = sk;
    // Ping-pong forms of the one-block-per-CU tiles (profiles/r02_pp_bench.txt: 256x160 +6..10 %, 128x160 +10..14 % on
    // the UNet's conv shapes, never slower): 128x160 x 3 stages (4 waves) -> 8 waves x 4 stages; 256x160 x 3 stages ->
    // ping-pong unless the folded-LayerNorm prefetch (which does not fit beside three 256-row stages) sends the launch
    // to the 2-stage lock-step kernel anyway.
    if (gemm_pingpong()) {
      const bool lnf = a.x_mode == PP_X_PLAIN && (a.ln_stats || a.row_stats_out || a.act == PP_ACT_GEGLU);
      if (c.tile == 31) c.tile = 54;
      else if (c.tile == 33 && !lnf) c.tile = 53;
    }
  }
  if (c.splitk <= 0) {
    const int tb = c.tile % 10;
    const int bm = tb == PP_TILE_128x160 ? 128 : tb == PP_TILE_64x160 ? 64 : 256;
    int sk = 1;
    while (blocks(bm) * sk < 192 && kt / (sk * 2) >= 16 && sk < 8) sk *= 2;
    c.splitk = sk;
  }
  if (a.N % 8) c.splitk = 1;   // the split-K combine works on 8-column strips
  if (a.row_stats_out) c.splitk = 1;   // row moments come out of the fused (single-pass) epilogue only
  return c;
}

// the deterministic split-K combine (+ the GroupNorm statistics of the output, if subscribed) behind a GEMM / conv launch
// that wrote `splitk` fp32 slabs to a.workspace
// PPGemmArgs.gn_next_*: can the combine of this (split-K) launch apply the consumer GroupNorm of subscription `sub`?
bool gn_next_shape_ok(const PPGemmArgs& a, int sub) {
  if (sub < 0 || sub > 1 || !a.gn_acc[sub] || a.gn_c0[sub] != 0 || a.gn_cg[sub] < 8 || 40 % a.gn_cg[sub]) return false;
  if (a.gn_cg[sub] * a.gn_groups[sub] != a.N || a.N % 40 || !reduce_lean_ok(a)) return false;
  const int o = 1 - sub;      // a second subscription: its groups per 40-column tile must fit the LDS slots as well
  if (a.gn_acc[o] && (a.gn_cg[o] < 8)) return false;
  const int hw = a.rows_per_batch;
  return hw >= 16 && hw <= 256 && hw % 16 == 0 && (hw <= 128 || hw % 128 == 0) && a.M % hw == 0 && a.ldo == a.N;
}

// (ABI v21) can the launch combine its split-K slabs in-kernel (gemm_combine.h)?  bm / tiles / pp_tile describe the form
// pp_gemm_bf16 runs it in.  `with_consumers`: also honour what a consumer may have patched into the request (gn_next_*).
bool fused_combine_shape_ok(const PPGemmArgs& a, int bm, int tiles, int splitk, bool pp_tile, bool with_consumers, bool advice) {
  if (!(splitk == 2 || splitk == 4 || splitk == 8) || !pp_tile || !reduce_lean_ok(a)) return false;
  if (tiles % 8) return false;                                   // every split of a tile on the XCD (tile id % 8)
  if (bm % splitk || bm / splitk < 16) return false;             // a share = whole 16-row blocks
  // ADVICE (what pp_gemm_combine_ctr_bytes() tells a planner; counters handed over anyway are honoured
  // wherever the result is right).  Measured inside the headline step's hipGraph, kernel by kernel against the separate combine
  // (rocprofv3, profiles/r06_fused_combine.txt): 2 and 4 splits win 0.5 .. 8 us per launch (shares of 32 .. 128 rows: the
  // 16x16 and 32x32 levels, FF2 . proj_out at 16x16); 8 splits LOSE 2 .. 3 us (shares of 16 / 32 rows at the 8x8 level and the
  // 32 -> 16 downsample: the tail is a chain of six ~1 us memory round trips -- drain, arrival, poll, slabs, statistics,
  // second arrival -- against one 11 us kernel).  And only launches whose splits are all resident at once (one workgroup
  // per CU): beyond that a split waits FC_SPIN_TICKS for a partner that cannot start.
  // (lab) PP_FUSED_COMBINE_SPLITS = the largest split count advised
  static const int max_splits = pp_lab_env("PP_FUSED_COMBINE_SPLITS", 4);
  if (advice && (splitk > max_splits || tiles * splitk > pp_cu_count())) return false;
  if (a.M % bm || a.N % 8) return false;
  if ((uint64_t)splitk * (uint64_t)a.M * (uint64_t)a.N * 4u >= 0x80000000ull) return false;
  if (a.rowvec && (a.rows_per_batch <= 0 || a.rows_per_batch % 64)) return false;   // a pass of <= 64 rows inside one batch item
  if (!with_consumers) return true;
  if ((a.gn_acc[0] || a.gn_acc[1]) && (a.rows_per_batch <= 0 || a.rows_per_batch % 64)) return false;
  if (a.gn_next_out) {
    if (!gn_next_shape_ok(a, a.gn_next_sub)) return false;
    const int hw = a.rows_per_batch, cg = a.gn_cg[a.gn_next_sub];
    if (hw % 64 || bm % hw || 160 % cg) return false;            // whole (batch item, group) populations inside a tile
  }
  return true;
}

template <int EDT>
int launch_combine(const PPGemmArgs& a, int splitk, hipStream_t st) {
  const long long total = (long long)a.M * (a.N / 8);
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (a.gn_next_out && gn_next_shape_ok(a, a.gn_next_sub)) {
    if (!a.gn_next_gamma || !a.gn_next_beta) return PP_ERR_BAD_ARG;
    const int tn = a.N / 40;
    const int rows_pass = a.rows_per_batch < 128 ? a.rows_per_batch : 128;
    hipLaunchKernelGGL(pp_splitk_reduce_gn_apply_kernel<EDT>, dim3((a.M / a.rows_per_batch) * tn), dim3(rows_pass * 5), 0, st, a,
                       splitk, tn, rows_pass);
    PP_CHECK_LAUNCH("pp_splitk_reduce_gn_apply_kernel");
    return PP_OK;
  }
  if (a.gn_next_out) return PP_ERR_UNSUPPORTED;      // (the caller asked pp_gemm_gn_next_ok() first: never a silent skip)
  if (a.gn_acc[0] || a.gn_acc[1]) {
    const int tn = (a.N + 159) / 160;
    hipLaunchKernelGGL(pp_splitk_reduce_gn_kernel<EDT>, dim3(((a.M + 15) / 16) * tn), dim3(320), 0, st, a, splitk, tn);
  } else if (reduce_lean_ok(a)) hipLaunchKernelGGL((pp_splitk_reduce_kernel<true, EDT>), dim3(nb), dim3(256), 0, st, a, splitk);
  else hipLaunchKernelGGL((pp_splitk_reduce_kernel<false, EDT>), dim3(nb), dim3(256), 0, st, a, splitk);
  PP_CHECK_LAUNCH("pp_splitk_reduce_kernel");
  return PP_OK;
}

template <int BM, int BN, int WM, int WN, int XMODE, int EDT>
int launch(const PPGemmArgs& a, int splitk, hipStream_t st) {
  constexpr int T = WM * WN * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = pp_gemm_kernel<BM, BN, WM, WN, XMODE, EDT>;
  if (pp_func_lds(reinterpret_cast<const void*>(kern), LDS, "hipFuncSetAttribute(gemm)") != PP_OK) return PP_ERR_LAUNCH;
  GemmDerived d;
  d.tiles_m = (a.M + BM - 1) / BM;
  d.tiles_n = (a.N + BN - 1) / BN;
  d.kt_total = a.K / 64;
  d.kt_per_split = (d.kt_total + splitk - 1) / splitk;
  d.ctiles = (a.c1 + a.c2) / 64;
  d.n_major = (gemm_n_major() && d.tiles_m < d.tiles_n && d.tiles_m <= 8) ? 1 : 0;
  dim3 grid(d.tiles_m * d.tiles_n, splitk, 1);
  hipLaunchKernelGGL(kern, grid, dim3(T), LDS, st, a, d);
  PP_CHECK_LAUNCH("pp_gemm_kernel");
  if (splitk > 1) return launch_combine<EDT>(a, splitk, st);
  return PP_OK;
}

// (lab build) PP_GEMM_NMAJOR=0|1: A/B switch for the N-major tile order of small-M launches (default 1)
bool gemm_n_major() {
  static const int v = pp_lab_env("PP_GEMM_NMAJOR", 1);
  return v != 0;
}

// (lab build) PP_GEMM_PP=0|1: A/B switch for the ping-pong tiles in the automatic choice (default 1)
bool gemm_pingpong() {
  static const int v = pp_lab_env("PP_GEMM_PP", 1);
  return v != 0;
}

// (lab build) PP_GEMM_DMAI=0|1: tile refills as one DMA burst after the barrier (0) or spread over the MFMA burst (1, default)
bool gemm_dma_interleave() {
  static const int v = pp_lab_env("PP_GEMM_DMAI", 1);
  return v != 0;
}

template <int BM, int BN, int WM, int WN, int XMODE, int NS, int EPI, bool PP, int EDT>
int launch2(const PPGemmArgs& a, int splitk, hipStream_t st) {
  if constexpr (EPI == 0) {
    if ((a.gn_acc[0] || a.gn_acc[1]) && splitk == 1) return launch2<BM, BN, WM, WN, XMODE, NS, 4, PP, EDT>(a, splitk, st);
  }
  if constexpr (XMODE == PP_X_PLAIN && EPI == 0) {
    if (a.act == PP_ACT_SOFTMAX80) {                   // (one instantiation: the 64 x 160 x 3-stage tile choose() sends it to)
      if constexpr (BM == 64 && NS == 3 && !PP) return launch2<BM, BN, WM, WN, XMODE, NS, 5, PP, EDT>(a, splitk, st);
      else return PP_ERR_UNSUPPORTED;
    }
    if (a.act == PP_ACT_GEGLU && splitk == 1) return launch2<BM, BN, WM, WN, XMODE, NS, 2, PP, EDT>(a, splitk, st);
    if (a.ln_stats || a.row_stats_out) return launch2<BM, BN, WM, WN, XMODE, NS, 1, PP, EDT>(a, splitk, st);
  }
  constexpr bool LNF = EPI == 1 || EPI == 2 || EPI == 5;
  constexpr int T = WM * WN * 64;
  // + epilogue-operand prefetch (LNF)
  constexpr int LDS = NS * (BM + BN) * 128 + (LNF ? 2048 + BM * 32 : 0);
  static_assert(LDS <= 160 * 1024 || (LNF && NS > 2), "LDS budget");
  if constexpr (LDS > 160 * 1024) {   // 256x160 x 3 stages has no room for the prefetch: drop to 2 stages (lock-step)
    return launch2<BM, BN, WM, WN, XMODE, 2, EPI, false, EDT>(a, splitk, st);
  } else {
  // (2-stage pipelines need their single in-flight refill as early as possible: the spread costs them time)
#ifdef PP_LAB
  auto kern = pp_gemm_kernel_v2<BM, BN, WM, WN, XMODE, NS, EPI, false, PP, EDT>;
  if constexpr (NS >= 3 && !PP) {
    if (gemm_dma_interleave()) kern = pp_gemm_kernel_v2<BM, BN, WM, WN, XMODE, NS, EPI, true, false, EDT>;
  }
#else
  auto kern = pp_gemm_kernel_v2<BM, BN, WM, WN, XMODE, NS, EPI, (NS >= 3 && !PP), PP, EDT>;
#endif
  if (pp_func_lds(reinterpret_cast<const void*>(kern), LDS, "hipFuncSetAttribute(gemm v2)") != PP_OK) return PP_ERR_LAUNCH;
  GemmDerived d;
  d.tiles_m = (a.M + BM - 1) / BM;
  d.tiles_n = (a.N + BN - 1) / BN;
  d.kt_total = a.K / 64;
  d.kt_per_split = (d.kt_total + splitk - 1) / splitk;
  d.ctiles = (a.c1 + a.c2) / 64;
  d.n_major = (gemm_n_major() && d.tiles_m < d.tiles_n && d.tiles_m <= 8) ? 1 : 0;
  dim3 grid(d.tiles_m * d.tiles_n, splitk, 1);
  hipLaunchKernelGGL(kern, grid, dim3(T), LDS, st, a, d);
  PP_CHECK_LAUNCH("pp_gemm_kernel_v2");
  if (splitk > 1 && !a.tile_ctr) return launch_combine<EDT>(a, splitk, st);   // (tile_ctr: combined by the last arriver)
  return PP_OK;
  }
}

// GroupNorm statistics in the epilogue: plain bf16 output through the v2 staged epilogue or the lean split-K combine,
// 64-row passes inside one batch item
bool gn_stats_supported(const PPGemmArgs& a) {
  if (a.act != PP_ACT_NONE || a.out_f32 || a.out_vt || a.ln_stats || a.row_stats_out) return false;
  if (!v2_ok(a) || !reduce_lean_ok(a)) return false;
  if (a.rows_per_batch <= 0 || a.rows_per_batch % 64) return false;
  for (int k = 0; k < 2; ++k)
    if (a.gn_acc[k] && (a.gn_cg[k] < 8 || a.gn_groups[k] <= 0 || a.gn_c0[k] < 0)) return false;   // <= GN_SLOTS groups / tile
  return true;
}

int validate(const PPGemmArgs& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || !pp_dt_ok(a.dtype)) return PP_ERR_BAD_ARG;
  if (a.K % 64 != 0 || a.N % 4 != 0) return PP_ERR_BAD_ARG;
  if (!a.x1 || !a.w || !a.out) return PP_ERR_BAD_ARG;
  if (a.x_mode == PP_X_PLAIN) {
    if (a.c1 + a.c2 != a.K) return PP_ERR_BAD_ARG;
    if (a.c2 > 0 && (!a.x2 || a.c1 % 64 != 0)) return PP_ERR_BAD_ARG;
    if (a.ldx1 % 8 != 0 || (a.c2 > 0 && a.ldx2 % 8 != 0)) return PP_ERR_BAD_ARG;
    if ((uint64_t)a.M * (uint64_t)a.ldx1 * 2u >= 0x80000000ull) return PP_ERR_UNSUPPORTED;
  } else if (a.x_mode == PP_X_CONV3X3) {
    if (a.c1 % 64 != 0 || a.c2 % 64 != 0 || a.c1 <= 0) return PP_ERR_BAD_ARG;
    if (a.c2 > 0 && !a.x2) return PP_ERR_BAD_ARG;
    if (a.c3 < 0 || a.c4 < 0 || a.c3 % 64 || a.c4 % 64 || (a.c3 > 0 && !a.x3) || (a.c4 > 0 && (!a.x4 || a.c3 == 0)))
      return PP_ERR_BAD_ARG;
    if ((a.c3 > 0) && (a.stride != 1 || a.up)) return PP_ERR_UNSUPPORTED;
    if (a.K != 9 * (a.c1 + a.c2) + a.c3 + a.c4) return PP_ERR_BAD_ARG;
    if (a.stride != 1 && a.stride != 2) return PP_ERR_BAD_ARG;
    if (a.M != a.batch * a.hout * a.wout) return PP_ERR_BAD_ARG;
    const int hv = a.up ? 2 * a.hin : a.hin, wv = a.up ? 2 * a.win : a.win;
    if (a.hout != (hv + 2 - 3) / a.stride + 1 || a.wout != (wv + 2 - 3) / a.stride + 1) return PP_ERR_BAD_ARG;
    if ((uint64_t)a.batch * a.hin * a.win * (uint64_t)(a.c1 > a.c2 ? a.c1 : a.c2) * 2u >= 0x80000000ull)
      return PP_ERR_UNSUPPORTED;
  } else {
    return PP_ERR_BAD_ARG;
  }
  if ((uint64_t)a.N * (uint64_t)a.K * 2u >= 0x80000000ull) return PP_ERR_UNSUPPORTED;
  if ((a.rowvec || a.out_vt) && a.rows_per_batch <= 0) return PP_ERR_BAD_ARG;
  if (a.act == PP_ACT_GEGLU && (a.out_f32 || a.out_vt || a.x_mode != PP_X_PLAIN)) return PP_ERR_BAD_ARG;
  // (ABI v20) per-item weights and the 80-column group softmax: the folded cross-attention as two GEMMs
  if (a.w_batch_stride < 0 || a.vec_batch_stride < 0) return PP_ERR_BAD_ARG;
  if (a.w_batch_stride > 0) {
    if (a.x_mode != PP_X_PLAIN || a.rows_per_batch <= 0 || a.rows_per_batch % 64 || a.M % a.rows_per_batch || (a.w_batch_stride & 7))
      return PP_ERR_BAD_ARG;
    if (a.out_vt || a.out_f32 || a.act == PP_ACT_GEGLU || a.splitk > 1 || !v2_ok(a)) return PP_ERR_UNSUPPORTED;
  }
  if (a.vec_batch_stride > 0 && (a.act != PP_ACT_SOFTMAX80 || a.w_batch_stride <= 0 || (a.vec_batch_stride & 3))) return PP_ERR_UNSUPPORTED;
  if (a.act == PP_ACT_SOFTMAX80) {
    if (a.x_mode != PP_X_PLAIN || a.N % 80 || a.ldo % 8 || a.ldo < a.N) return PP_ERR_BAD_ARG;
    if (a.out_f32 || a.out_vt || a.res1 || a.res2 || a.rowvec || a.row_stats_out || a.gn_acc[0] || a.gn_acc[1] ||
        a.gn_next_out || a.out_dup_rows > 0 || a.splitk > 1 || a.scale != 1.0f || !v2_ok(a))
      return PP_ERR_UNSUPPORTED;
  }
  if (a.out_vt && a.vt_col0 % 4 != 0) return PP_ERR_BAD_ARG;
  if (a.ln_stats && (!a.ln_colsum || a.ln_tiles <= 0 || a.ln_dim <= 0)) return PP_ERR_BAD_ARG;
  if (a.row_stats_out && (a.out_f32 || a.out_vt || a.act == PP_ACT_GEGLU || !v2_ok(a))) return PP_ERR_UNSUPPORTED;
  if ((a.gn_acc[0] || a.gn_acc[1]) && !gn_stats_supported(a)) return PP_ERR_UNSUPPORTED;
  // GroupNorm in the conv loader: a conv3x3 feature with both operands or none -- never silently ignored
  if ((a.gn_in_acc || a.gn_in_gb) && a.x_mode != PP_X_CONV3X3) return PP_ERR_UNSUPPORTED;
  if ((a.gn_in_acc != nullptr) != (a.gn_in_gb != nullptr)) return PP_ERR_BAD_ARG;
  // the CFG-twin prefix (ABI v19): half-batch residual read with wrap, output written for both halves
  if (a.res1_wrap_rows < 0 || a.out_dup_rows < 0 || a.gn_dup_batch < 0 || (a.gn_dup_mask & ~3)) return PP_ERR_BAD_ARG;
  if (a.res1_wrap_rows > 0 && (!a.res1 || a.M > 2 * a.res1_wrap_rows || a.res1_wrap_rows % 64)) return PP_ERR_BAD_ARG;
  if (a.out_dup_rows > 0) {
    if (a.out_dup_rows < a.M) return PP_ERR_BAD_ARG;                 // (the copy must not overlap the rows this launch writes)
    if (a.out_f32 || a.out_vt || a.act == PP_ACT_GEGLU || a.row_stats_out || a.ln_stats || a.gn_next_out || a.gn_in_acc ||
        !v2_ok(a))
      return PP_ERR_UNSUPPORTED;
  }
  if (a.gn_dup_mask) {
    if (a.out_dup_rows <= 0 || a.gn_dup_batch <= 0 || a.rows_per_batch <= 0 ||
        a.gn_dup_batch * a.rows_per_batch != a.out_dup_rows)
      return PP_ERR_BAD_ARG;
    for (int k = 0; k < 2; ++k)
      if ((a.gn_dup_mask >> k & 1) && !a.gn_acc[k]) return PP_ERR_BAD_ARG;
  }
  return PP_OK;
}

template <int EDT>
int dispatch(const PPGemmArgs& a, const Choice& c, hipStream_t st) {
  const bool conv = a.x_mode == PP_X_CONV3X3;
  switch (c.tile) {
    case PP_TILE_128x160:
      return conv ? launch<128, 160, 2, 2, PP_X_CONV3X3, EDT>(a, c.splitk, st) : launch<128, 160, 2, 2, PP_X_PLAIN, EDT>(a, c.splitk, st);
    case PP_TILE_64x160:
      return conv ? launch<64, 160, 2, 2, PP_X_CONV3X3, EDT>(a, c.splitk, st) : launch<64, 160, 2, 2, PP_X_PLAIN, EDT>(a, c.splitk, st);
    case PP_TILE_256x160:
      return conv ? launch<256, 160, 4, 2, PP_X_CONV3X3, EDT>(a, c.splitk, st) : launch<256, 160, 4, 2, PP_X_PLAIN, EDT>(a, c.splitk, st);
#define PP_V2(ID, BM_, WM_, NS_, PP_)                                                                     \
    case ID:                                                                                             \
      return conv ? launch2<BM_, 160, WM_, 2, PP_X_CONV3X3, NS_, 0, PP_, EDT>(a, c.splitk, st)           \
                  : launch2<BM_, 160, WM_, 2, PP_X_PLAIN, NS_, 0, PP_, EDT>(a, c.splitk, st);
      PP_V2(21, 128, 2, 2, false)
      PP_V2(31, 128, 2, 3, false)
      PP_V2(22, 64, 2, 2, false)
      PP_V2(32, 64, 2, 3, false)
      PP_V2(42, 64, 2, 4, false)
      PP_V2(62, 64, 2, 5, false)    // (five stages: 140 KB, one block per CU)
      PP_V2(23, 256, 4, 2, false)
      PP_V2(33, 256, 4, 3, false)
      PP_V2(24, 128, 4, 2, false)   // 8-wave 128x160 (wave tile 32x80): 4 waves / SIMD with two co-resident blocks
      PP_V2(53, 256, 4, 3, true)    // ping-pong 8-wave tiles (one block per CU)
      PP_V2(44, 128, 4, 3, true)
      PP_V2(54, 128, 4, 4, true)
#undef PP_V2
    default:
      return PP_ERR_BAD_ARG;
  }
}

}  // namespace

// conv_gn.hip: GroupNorm + SiLU of the conv input fused into the loader (PPGemmArgs.gn_in_acc)
bool pp_conv_gn_wanted(const PPGemmArgs& a);
int pp_conv_gn_splitk(const PPGemmArgs& a);       // 0 = shape not supported by the fused kernel
int pp_conv_gn_bm(const PPGemmArgs& a);           // rows of the tile it runs on (0 = not supported)
int pp_conv_gn_run(const PPGemmArgs& a, hipStream_t st);

extern "C" int pp_gemm_gn_stats_ok(const PPGemmArgs* args) {
  if (!args) return 0;
  PPGemmArgs a = *args;
  a.gn_acc[0] = a.gn_acc[1] = nullptr;
  a.gn_dup_mask = 0;
  if (validate(a) != PP_OK || !gn_stats_supported(a)) return 0;
  if (pp_conv_gn_wanted(a)) return pp_conv_gn_splitk(a) > 0 ? 1 : 0;
  const Choice c = choose(a);
  return c.tile > 10 ? 1 : 0;
}

// the split-K factor pp_gemm_bf16 will run this launch with (1 = no combine launch behind it)
static int planned_splitk(const PPGemmArgs& a) {
  if (pp_conv_gn_wanted(a)) return pp_conv_gn_splitk(a);
  const Choice c = choose(a);
  return c.tile > 10 ? c.splitk : 1;
}

extern "C" int pp_gemm_gn_next_ok(const PPGemmArgs* args, int sub) {
  if (!args || validate(*args) != PP_OK) return 0;
  return (planned_splitk(*args) > 1 && gn_next_shape_ok(*args, sub)) ? 1 : 0;
}

// the form pp_gemm_bf16 runs this request in, as the fused-combine predicate sees it
struct PlannedForm {
  int bm, tiles, splitk;
  bool pp_tile;
};
static PlannedForm planned_form(const PPGemmArgs& a) {
  PlannedForm f{0, 0, 1, false};
  const int tn = (a.N + 159) / 160;
  if (pp_conv_gn_wanted(a)) {
    f.splitk = pp_conv_gn_splitk(a);
    f.bm = pp_conv_gn_bm(a);
    f.pp_tile = f.bm > 0;                        // (the shipping fused-norm kernels are all 8-wave ping-pong tiles)
  } else {
    const Choice c = choose(a);
    f.splitk = c.tile > 10 ? c.splitk : 1;
    f.pp_tile = c.tile == 53 || c.tile == 44 || c.tile == 54;
    f.bm = c.tile == 53 ? 256 : 128;
  }
  if (f.bm > 0) f.tiles = ((a.M + f.bm - 1) / f.bm) * tn;
  return f;
}

extern "C" size_t pp_gemm_combine_ctr_bytes(const PPGemmArgs* args) {
  if (!args || validate(*args) != PP_OK) return 0;
  const PlannedForm f = planned_form(*args);
  if (!fused_combine_shape_ok(*args, f.bm, f.tiles, f.splitk, f.pp_tile, false, true)) return 0;
  if (!pp_xcd_placement_ok()) return 0;
  return (size_t)f.tiles * FC_CTR_BYTES;
}

static bool combine_fused(const PPGemmArgs& a) {
  if (!a.tile_ctr) return false;
  const PlannedForm f = planned_form(a);
  return fused_combine_shape_ok(a, f.bm, f.tiles, f.splitk, f.pp_tile, true, false);
}

extern "C" int pp_gemm_combine_fused(const PPGemmArgs* args) {
  return (args && validate(*args) == PP_OK && combine_fused(*args)) ? 1 : 0;
}

// the fp32 slabs of a split-K launch + (behind them) the per-tile scratch of the in-kernel combine (gemm_combine.h)
extern "C" size_t pp_gemm_workspace_bytes(const PPGemmArgs* args) {
  if (!args || validate(*args) != PP_OK) return 0;
  const PlannedForm f = planned_form(*args);
  const int sk = pp_conv_gn_wanted(*args) ? f.splitk : choose(*args).splitk;
  if (sk <= 1) return 0;
  const size_t scratch = fused_combine_shape_ok(*args, f.bm, f.tiles, f.splitk, f.pp_tile, false, false) ? (size_t)f.tiles * FC_SCR_BYTES : 0;
  return (size_t)sk * args->M * args->N * sizeof(float) + scratch;
}


extern "C" int pp_gemm_bf16(const PPGemmArgs* args, void* stream) {
  if (!args) return PP_ERR_BAD_ARG;
  const int v = validate(*args);
  if (v != PP_OK) return v;
  // tile_ctr is a permission: the kernels see it only where the launch is eligible for the in-kernel combine
  PPGemmArgs a_ = *args;
  const bool fused = combine_fused(a_);
  if (!fused) a_.tile_ctr = nullptr;
  const PPGemmArgs& a = a_;
  if (a.gn_next_out && !(planned_splitk(a) > 1 && gn_next_shape_ok(a, a.gn_next_sub))) return PP_ERR_UNSUPPORTED;
  if (pp_conv_gn_wanted(a)) {      // norm -> SiLU -> conv3x3 as one launch (no silent fallback: pp_conv_gn_supported() tells)
    const int sk = pp_conv_gn_splitk(a);
    if (sk <= 0) return PP_ERR_UNSUPPORTED;
    if ((a.gn_acc[0] || a.gn_acc[1]) && !gn_stats_supported(a)) return PP_ERR_UNSUPPORTED;
    const int rc = pp_conv_gn_run(a, (hipStream_t)stream);
    if (rc != PP_OK || sk == 1 || fused) return rc;
    return a.dtype == PP_DT_F16 ? launch_combine<PP_DT_F16>(a, sk, (hipStream_t)stream)
                                : launch_combine<PP_DT_BF16>(a, sk, (hipStream_t)stream);
  }
  const Choice c = choose(a);
  if (c.splitk > 1 && !a.workspace) return PP_ERR_WORKSPACE;
  if (c.tile > 10 && !v2_ok(a)) return PP_ERR_BAD_ARG;
  if (a.out_dup_rows > 0 && (c.tile < 10 || c.splitk > 1)) return PP_ERR_UNSUPPORTED;   // single-pass staged epilogue only
  if ((a.row_stats_out || a.gn_acc[0] || a.gn_acc[1] || (a.x_mode == PP_X_CONV3X3 && a.c3 > 0)) &&
      c.tile < 10)
    return PP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (a.dtype == PP_DT_F16) return dispatch<PP_DT_F16>(a, c, st);
  return dispatch<PP_DT_BF16>(a, c, st);
}
