// ResnetBlock2D's `GroupNorm -> SiLU -> Conv2d(3x3)` as ONE launch for gfx950 (MI355X): a halo-tile implicit GEMM whose
// loader applies the normalisation.
//
// Reference: diffusers 0.27 ResnetBlock2D.forward (norm1 -> nonlinearity -> conv1, norm2 -> nonlinearity -> dropout ->
// conv2; ctor sites /root/reference/powerpaint/models/unet_2d_blocks.py:1274-1285, 1457-1500, 2696-2770).  The two-launch
// form of this repository is pp_groupnorm_apply_acc (norm.hip) + pp_gemm_bf16(PP_X_CONV3X3) (gemm.hip): the apply launch
// reads and rewrites the whole activation (~10 us of launch floor 44 times per UNet step), and the tap-major implicit
// GEMM then pulls every activation row through the CU's 64 B/clk global->LDS path nine times (52 KB per K step for a
// 256 x 160 tile: 0.65 of the MFMA time on the address unit, 4.2x the algorithmic fabric traffic).
//
// Here the K walk is CHANNEL-CHUNK major: for each 64-channel chunk the workgroup brings the tile's rows PLUS one image
// row above and below (the "halo tile", (BM / W + 2) x W pixels x 64 channels, <= 48 KB) into LDS ONCE, normalises it in
// place (x * scale[c] + shift[c], SiLU, rounded to the 16-bit format exactly as the two-launch path stores it) and then
// runs the nine taps against it: a tap is a constant pixel offset (ky * W + kx - 1) on the fragment-read address.  The
// horizontal halo needs no storage: tiles span full image rows, so the out-of-row lanes of the kx = 0 / 2 taps are
// redirected to a 128-byte block of zeros.  Per K step only the 20 KB weight tile moves (LDS-DMA, 3-stage ring); the
// halo tile of the NEXT chunk is DMA'd raw into the second halo buffer during taps 0 .. 2 and normalised by the lanes
// that fetched it, one 8-pixel strip per tap, in the READ phase of the ping-pong loop -- VALU + LDS work that runs
// beside the partner wave's MFMA phase on the same SIMD (MI355X_MICROARCH.md "Two waves per SIMD": the complementary
// pairing).  The optional 1x1 tail over (x3, x4) (ResnetBlock2D.conv_shortcut merged into conv2) follows as a plain
// 3-stage GEMM phase on the same accumulators.  Epilogue = the staged 64-row passes of gemm.hip (bias, time-embedding
// row vector, residuals, GroupNorm statistics of the OUTPUT for the next norm, split-K slabs).
#include <type_traits>
#include <utility>

#include "pp_common.h"
#include "gemm_gn.h"
#include "gemm_combine.h"

namespace {

typedef __attribute__((address_space(3))) void* cg_lds_t;

constexpr int CG_T = 512, CG_BN = 160, CG_NI = 5;
constexpr int CG_WP = 3;                        // weight-tile DMA pieces per wave and K step (20 strips of 8 rows, 8 waves)
constexpr int CG_HPW_MAX = 6;                   // halo DMA pieces per wave and chunk at most (48 strips of 8 pixels)
constexpr int CG_HALO_PX = 384;
constexpr int CG_HALO = CG_HALO_PX * 128;       // one halo buffer
constexpr int CG_WST = CG_BN * 128;             // one weight stage
constexpr int CG_WOFF = 2 * CG_HALO;
constexpr int CG_TAB = CG_WOFF + 3 * CG_WST;
constexpr int CG_T_STATS = CG_TAB;              // (mean, rstd) of the batch item's groups (<= 32)
constexpr int CG_T_ZERO = CG_TAB + 256;         // 128 bytes of zeros: where the out-of-row lanes of the kx = 0 / 2 taps read
constexpr int CG_T_GB = CG_TAB + 1024;          // 2 x 1 KB: (gamma, beta) of a chunk's 64 channels
constexpr int CG_T_SCSH = CG_TAB + 3072;        // 512 B: (scale, shift) of the chunk being normalised, [k-slot][8 channels][2]
constexpr int CG_LDS = CG_TAB + 3584;           // 163,328 B of the CU's 163,840
static_assert(CG_LDS <= 160 * 1024, "LDS budget");

struct CGDerived {
  int tiles_m, tiles_n, n_major;
  int nch1, nch;        // 64-channel chunks of x1, of concat(x1, x2)
  int ntail3, ntail;    // 64-deep K tiles of x3, of concat(x3, x4) (the 1x1 tail)
  int hp;               // pixels of the halo tile = BM + 2 W
  int cg;               // channels per group of the input norm
  float inv_cg;
};

template <int... I, class F>
PP_DEVINL void cg_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void cg_static_for(F&& f) { cg_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// PP: ping-pong main loop (waves 0-3 / 4-7 half a K step apart, two barriers per K step); else lock-step (one barrier).
// HPW: halo strips per wave = ceil((BM + 2 W) / 64): a wave DMAs and normalises strips wave + 8 j, j < HPW, of the next chunk.
// NMODE: where the normalisation of the next halo tile runs -- 1 inside the MFMA burst, 0 in the read phase; lab timing
// probes: 2 = no normalisation at all (the loop skeleton on raw data), 3 = read phase and no s_setprio around the burst.
template <int BM, int HPW, bool PP, int NMODE, int EDT>
__global__ void __launch_bounds__(CG_T, 2) pp_conv_gn_kernel(const PPGemmArgs a, const CGDerived d) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  constexpr int T = CG_T, WN = 2, MI = BM / 64, NI = CG_NI, BN = CG_BN;
  constexpr int CG_HPW = HPW;
  constexpr bool NORM = NMODE != 2;
  constexpr bool NREAD = NMODE == 0 || NMODE == 3;      // normalisation in the read phase
  static_assert(HPW >= 2 && HPW <= CG_HPW_MAX, "halo strips per wave");
  constexpr int XP2 = BM / 64;                    // phase-2 (tail) X-tile pieces per wave and K step
  constexpr int P2 = XP2 + CG_WP;
  constexpr int ST2 = BM * 128;                   // phase-2 X stage (three of them over the two halo buffers)
  constexpr int EPI_ROWS = 64, EPI_LD = BN * 4 + 16;
  constexpr int RS_OFF = EPI_ROWS * EPI_LD + BM * 8;
  static_assert(3 * ST2 <= 2 * CG_HALO, "tail stages live in the halo buffers");
  static_assert(MI * 16 <= EPI_ROWS && EPI_ROWS % (MI * 16) == 0, "wave rows vs epilogue pass");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;

  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  if (d.n_major) {
    tile_n = lid / d.tiles_m;
    tile_m = lid - tile_n * d.tiles_m;
  } else {
    tile_m = lid / d.tiles_n;
    tile_n = lid - tile_m * d.tiles_n;
  }
  const int m_blk = tile_m * BM, n_blk = tile_n * BN;
  const int split = blockIdx.y, splits = gridDim.y;
  // every split takes an equal share of the chunks AND of the tail tiles
  const int ch_b = split * d.nch / splits, ch_e = (split + 1) * d.nch / splits;
  const int tk_b = split * d.ntail / splits, tk_e = (split + 1) * d.ntail / splits;

  // tile / halo geometry in OUTPUT pixels; (round 6) a.up: the input is the nearest-2x upsampling of a [hin][win] source
  // (Upsample2D: F.interpolate(scale_factor=2, mode="nearest") -> conv), i.e. halo pixel (iy, hx) is source pixel (iy / 2, hx / 2)
  const int Wd = a.wout, Hd = a.hout, HW = Hd * Wd;
  const int HWs = a.hin * a.win;                      // source pixels per image
  const int bimg = m_blk / HW;                        // the tile lies inside ONE image (host-checked: HW % BM == 0)
  const int y0 = (m_blk - bimg * HW) / Wd;            // its first image row (BM % W == 0)
  const int ctot = a.c1 + a.c2;
  const uint32_t wbytes = (uint32_t)a.N * (uint32_t)a.K * 2u;

  // lane -> (row of an 8-row strip, the k-slot it FETCHES so that its lane-linear LDS slot is swizzled by the row)
  const int lrow = lane >> 3;
  const int kslot = (lane & 7) ^ lrow;

  // ---- weight tile: strips of 8 rows, three per wave (a wave whose third strip falls beyond 160 re-issues its second)
  int vw[CG_WP], wlds[CG_WP];
#pragma unroll
  for (int i = 0; i < CG_WP; ++i) {
    const int strip = (wave * 8 + i * 64 < BN) ? i : i - 1;
    const int n = n_blk + wave * 8 + lrow + strip * 64;
    vw[i] = (n < a.N) ? (n * a.K + kslot * 8) * 2 : (int)PP_OOB;
    wlds[i] = (wave * 8 + strip * 64) * 128;
  }
  // ---- halo tile: strip s = wave + 8 j covers halo pixels 8 s .. 8 s + 7; halo pixel hp = (row hp / W, column hp % W), halo
  //      row 0 = the image row above the tile.  hpix = linear pixel index in the source tensors, -1 outside the image
  int hpix[CG_HPW];
#pragma unroll
  for (int j = 0; j < CG_HPW; ++j) {
    const int hp = (wave + 8 * j) * 8 + lrow;
    const int hr = hp / Wd, hx = hp - hr * Wd;
    const int iy = y0 + hr - 1;
    hpix[j] = (hp < d.hp && iy >= 0 && iy < Hd) ? (a.up ? bimg * HWs + (iy >> 1) * a.win + (hx >> 1) : bimg * HW + iy * Wd + hx) : -1;
  }
  // ---- fragments: lane (r16, g) reads row (.. + r16), k-slot (ks * 4 + g) of a 16 x 32 fragment
  const int r16 = lane & 15, g = lane >> 4;
  const int fsw = r16 & 7;
  const int wrow0 = wn * (NI * 16) + r16;
  const int rt0 = wm * (MI * 16) + r16;             // tile row of fragment 0 (fragment mi: + 16 mi); halo pixel of tap
                                                    // (ky, kx) = tile row + ky W + kx - 1
  unsigned edge = 0u;                               // bit mi: first pixel of an image row, bit 8 + mi: last one
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int lx = (rt0 + mi * 16) % Wd;
    if (lx == 0) edge |= 1u << mi;
    if (lx == Wd - 1) edge |= 1u << (8 + mi);
  }

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto mfma_all = [&](const v8_t (&wf)[2][NI], const v8_t (&xf)[2][MI]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = E::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
  };

  // =====================================================================================================================
  // phase 1: the nine taps over the normalised halo tiles of chunks ch_b .. ch_e - 1
  // =====================================================================================================================
  if (ch_b < ch_e) {
    // (gamma, beta) of chunk c -> table buffer gbuf (every wave issues the same 1 KB piece: identical bytes)
    auto issue_gb = [&](int c, int gbuf, bool live) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.gn_in_gb, (NORM && live) ? (uint32_t)ctot * 8u : 0u);   // (raw input: nothing is read)
      const int vo = lane < 32 ? lane * 16 : (int)PP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cg_lds_t)(smem + CG_T_GB + gbuf * 1024), 16, vo, c * 512, 0, 0);
    };
    // (scale, shift) of chunk c -> LDS table [k-slot][8][2], from the (gamma, beta) buffer (landed: the caller's vmcnt) and
    // the groups' (mean, rstd).  Every wave holds all eight k-slots and writes the whole table (identical bytes), then
    // reads only behind its own writes: no cross-wave ordering needed.  (In LDS, not in 16 registers per lane: the
    // 256 x 160 tile has 152 accumulator + fragment registers live in the MFMA phase that applies them.)
    auto compute_scsh = [&](int c, int gbuf) __attribute__((always_inline)) {
      const char* gp = smem + CG_T_GB + gbuf * 1024 + kslot * 64;
      const f32x2_t* st = reinterpret_cast<const f32x2_t*>(smem + CG_T_STATS);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t gbv = *reinterpret_cast<const f32x4_t*>(gp + q * 16);    // (gamma, beta) of channels 2 q, 2 q + 1
        f32x4_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = c * 64 + kslot * 8 + 2 * q + h;                        // channel of the (concatenated) norm input
          const int gi = min((int)(((float)ch + 0.5f) * d.inv_cg), 31);      // (clamp: the dead chunk behind the last one)
          const f32x2_t mr = st[gi];
          const float sv = mr[1] * gbv[2 * h];
          o[2 * h] = sv;
          o[2 * h + 1] = gbv[2 * h + 1] - mr[0] * sv;
        }
        // One lane per k-slot stores (lrow == 0): the eight lanes that share a k-slot hold the same bytes for the same
        // address, an 8-way serialised LDS store (2.4 M bank-conflict cycles per launch in profiles/r04_gemm_pmc.txt).
        // That makes the table a CROSS-LANE hand-off inside the wave, so the stores are fenced from the table reads that
        // follow by the asm statement behind this function's loop: without it hipcc sank the computation into the storing
        // lanes' branch and ran the other lanes' reads first (stale coefficients; the op tests caught it).
        if (lrow == 0) *reinterpret_cast<f32x4_t*>(smem + CG_T_SCSH + kslot * 64 + q * 16) = o;
      }
      // every lane, after the storing lanes' branch: the wave's LDS operations execute in order, the compiler may not move
      // a later table read above this statement
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // in-place normalisation of this lane's 16 bytes of halo strip (wave + 8 j) in buffer hbuf: what the lane's own DMA
    // wrote (ordered by the wave's vmcnt), rounded exactly as pp_groupnorm_apply_acc stores it; pixels outside the image
    // stay zero (the convolution pads the NORMALISED tensor)
    auto norm_piece = [&](int hbuf, int j) __attribute__((always_inline)) {
      char* p = smem + hbuf * CG_HALO + (wave + 8 * j) * 1024 + lane * 16;
      if (hpix[j] >= 0) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
        float r[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t ss = *reinterpret_cast<const f32x4_t*>(smem + CG_T_SCSH + kslot * 64 + q * 16);
          r[2 * q] = E::lo(v[q]) * ss[0] + ss[1];
          r[2 * q + 1] = E::hi(v[q]) * ss[2] + ss[3];
        }
        // SiLU with the hardware reciprocal (1 ulp) instead of silu_f's IEEE division (~10 instructions per element in a
        // phase that must not outlast the partner's 40 MFMAs); the 16-bit rounding that follows hides the difference
#pragma unroll
        for (int e = 0; e < 8; ++e)
          r[e] = r[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r[e] * -1.44269504088896340736f));
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = E::pack2(r[2 * q], r[2 * q + 1]);
        *reinterpret_cast<u32x4_t*>(p) = o;
      }
    };

    // the same arithmetic as a pure register function (for the MFMA-burst placement): every lane computes, the lanes whose
    // pixel lies outside the image (vmask = 0) return zeros
    auto norm_math = [&](const u32x4_t v, uint32_t vmask) __attribute__((always_inline)) -> u32x4_t {
      u32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // pair by pair: beside the MFMAs the matrix instruction in between hides the chain
        const f32x4_t ss = *reinterpret_cast<const f32x4_t*>(smem + CG_T_SCSH + kslot * 64 + q * 16);
        float r0 = E::lo(v[q]) * ss[0] + ss[1];
        float r1 = E::hi(v[q]) * ss[2] + ss[3];
        const float e0 = __builtin_amdgcn_exp2f(r0 * -1.44269504088896340736f);
        const float e1 = __builtin_amdgcn_exp2f(r1 * -1.44269504088896340736f);
        r0 *= __builtin_amdgcn_rcpf(1.0f + e0);
        r1 *= __builtin_amdgcn_rcpf(1.0f + e1);
        o[q] = E::pack2(r0, r1) & vmask;        // (a mask, not a select: hipcc turns the select into a branch around the math)
      }
      return o;
    };

    // next chunk's halo source (refreshed per chunk): descriptor, per-lane offsets, channel offset inside the source
    __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.x1, 0u);
    int hsoff = 0, hcsrc = 0;
    auto halo_source = [&](int c, bool live) __attribute__((always_inline)) {
      const bool first = c < d.nch1;
      int csrc;
      if (first) {
        csrc = a.c1;
        rs_h = make_rsrc(a.x1, live ? (uint32_t)a.batch * (uint32_t)HWs * (uint32_t)a.c1 * 2u : 0u);
        hsoff = c * 128;
      } else {
        csrc = a.c2;
        rs_h = make_rsrc(a.x2 ? a.x2 : a.x1, (live && a.x2) ? (uint32_t)a.batch * (uint32_t)HWs * (uint32_t)a.c2 * 2u : 0u);
        hsoff = (c - d.nch1) * 128;
      }
      hcsrc = csrc;
    };
    auto issue_halo_piece = [&](int hbuf, int j) __attribute__((always_inline)) {
      const int vo = hpix[j] >= 0 ? (hpix[j] * hcsrc + kslot * 8) * 2 : (int)PP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (cg_lds_t)(smem + hbuf * CG_HALO + (wave + 8 * j) * 1024), 16, vo,
                                               hsoff, 0, 0);
    };
    auto issue_w_piece = [&](const __amdgpu_buffer_rsrc_t rs, int stage, int i, int sow) __attribute__((always_inline)) {
      const int vo = vw[i], lo = wlds[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cg_lds_t)(smem + CG_WOFF + stage * CG_WST + lo), 16, vo, sow, 0, 0);
    };

    // ---- prologue: weight tiles of K steps 0 and 1, the first halo tile and its table; the groups' (mean, rstd)
    {
      const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.w, wbytes);
#pragma unroll
      for (int i = 0; i < CG_WP; ++i) issue_w_piece(rsw, 0, i, (0 * ctot + ch_b * 64) * 2);
#pragma unroll
      for (int i = 0; i < CG_WP; ++i) issue_w_piece(rsw, 1, i, (1 * ctot + ch_b * 64) * 2);
      halo_source(ch_b, true);
#pragma unroll
      for (int j = 0; j < CG_HPW; ++j) issue_halo_piece(0, j);
      issue_gb(ch_b, 0, true);
    }
    if (NORM && tid < a.gn_in_groups) {     // (the arithmetic of gn_fold_acc, norm.hip)
      const long long* ap = reinterpret_cast<const long long*>(a.gn_in_acc) + ((size_t)bimg * a.gn_in_groups + tid) * 2;
      const double s = (double)ap[0] * (1.0 / (double)PP_GN_SUM_SCALE);
      const double q = (double)ap[1] * (1.0 / (double)PP_GN_SQ_SCALE);
      const double n = (double)HW * (double)d.cg;
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      reinterpret_cast<f32x2_t*>(smem + CG_T_STATS)[tid] = f32x2_t{(float)mean, (float)(1.0 / sqrt(var + (double)a.gn_in_eps))};
    }
    if (tid < 32) reinterpret_cast<float*>(smem + CG_T_ZERO)[tid] = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (NORM) {
      compute_scsh(ch_b, 0);
#pragma unroll
      for (int j = 0; j < CG_HPW; ++j) norm_piece(0, j);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (PP && grp == 1) asm volatile("s_barrier" ::: "memory");      // group 1 runs one phase behind group 0

    int hb = 0;
#pragma unroll 1
    for (int c = ch_b; c < ch_e; ++c) {
      const bool nxt = c + 1 < ch_e;
      halo_source(c + 1, nxt);
      const __amdgpu_buffer_rsrc_t rsw_same = make_rsrc(a.w, wbytes);
      const __amdgpu_buffer_rsrc_t rsw_next = make_rsrc(a.w, nxt ? wbytes : 0u);
      const int hoff = hb * CG_HALO;
      const int hbn = hb ^ 1;
      int Wl = Wd;                                  // (opaque per iteration: keeps the nine taps' fragment addresses -- 2 x 9
      asm volatile("" : "+s"(Wl));                  //  registers, hoisted out of the chunk loop otherwise -- inside it)

      cg_static_for<9>([&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        constexpr int st_r = t % 3, st_w = (t + 2) % 3;             // (9 % 3 == 0: the stage of tap t is t % 3 in every chunk)
        // extra DMA pieces of a tap, after its three weight pieces: the next chunk's halo strips two per tap over taps
        // 0 .. 2 (strip j is needed from tap 3 + j on) and, in tap 0, the chunk's (gamma, beta) table
        constexpr auto NXT = [](int tt) constexpr {
          const int nh = tt > 2 ? 0 : (2 * tt + 2 <= CG_HPW ? 2 : 2 * tt + 1 <= CG_HPW ? 1 : 0);   // strips 2 tt, 2 tt + 1
          return nh + (tt == 0 ? 1 : 0);
        };
        constexpr int NX = NXT(t);
        // weight tile of K step + 2
        const __amdgpu_buffer_rsrc_t rsw = t + 2 <= 8 ? rsw_same : rsw_next;
        const int sow = t + 2 <= 8 ? ((t + 2) * ctot + c * 64) * 2 : ((t + 2 - 9) * ctot + (c + 1) * 64) * 2;

        if constexpr (PP) {
          asm volatile("s_barrier" ::: "memory");   // X: everyone's weight tile of this step is in LDS; the partner left its read phase
        } else {
          // this wave's weight pieces of this K step (issued two taps ago) have landed; younger pieces may be in flight
          constexpr int V = NXT((t + 7) % 9) + CG_WP + NXT((t + 8) % 9);
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(V) : "memory");
        }
        // ---- the next chunk's halo tile.  Strip j of this wave (DMA'd in tap j / 2) has landed when tap 3 + j begins: the
        //      counted waits of taps 2 .. 4 leave only younger pieces in flight.  Its normalisation -- ~70 VALU instructions --
        //      rides INSIDE this wave's MFMA burst (NMODE 1: two per MFMA, in the issue slots the 16-cycle matrix
        //      instruction leaves free) or sits in the read phase (NMODE 0, where the partner wave's s_setprio 1 starves
        //      its transcendentals: MI355X_MICROARCH.md "Two waves per SIMD", item 2).
        constexpr bool NT = NORM && t >= 3 && t - 3 < CG_HPW;        // a tap that carries one strip
        char* const np = smem + hbn * CG_HALO + (wave + 8 * (NT ? t - 3 : 0)) * 1024 + lane * 16;
        u32x4_t nv = {0u, 0u, 0u, 0u};
        f32x4_t ss0 = {0.f, 0.f, 0.f, 0.f}, ss1 = {0.f, 0.f, 0.f, 0.f};     // first half of this lane's (scale, shift) row
        if constexpr (NT) {
          // (unconditional: behind the last chunk this works on the zeros of the dead DMAs in the unused buffer -- a branch
          //  around the MFMA burst would put the 80 accumulator registers through a phi and double them)
          if constexpr (t == 3) compute_scsh(c + 1, hbn);
          if constexpr (NREAD) norm_piece(hbn, t - 3);
          else {
            nv = *reinterpret_cast<const u32x4_t*>(np);
            ss0 = *reinterpret_cast<const f32x4_t*>(smem + CG_T_SCSH + kslot * 64);
            ss1 = *reinterpret_cast<const f32x4_t*>(smem + CG_T_SCSH + kslot * 64 + 16);
          }
          if constexpr (NREAD) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- read phase: fragment reads of this K step, the DMA pieces spread between them
        const char* ws = smem + CG_WOFF + st_r * CG_WST;
        const int hp0 = rt0 + (t / 3) * Wl + (t % 3) - 1;
        const int xbase = hoff + (hp0 << 7) + ((g ^ (hp0 & 7)) << 4);
        v8_t wf[2][NI], xf[2][MI];
        constexpr int NR = NI + MI, NPC = CG_WP + NX;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (r < NI) {
            const int ni = r < NI ? r : 0;
            const char* pw = ws + (wrow0 + ni * 16) * 128;
            wf[0][ni] = *reinterpret_cast<const v8_t*>(pw + ((g ^ fsw) << 4));
            wf[1][ni] = *reinterpret_cast<const v8_t*>(pw + (((4 + g) ^ fsw) << 4));
          } else {
            const int mi = r >= NI ? r - NI : 0;
            // k-slot g of halo pixel hp0 + 16 mi (same swizzle for every mi: 16 = 0 mod 8; ks = 1: slot g + 4 = ^ 64 B)
            int o0 = xbase + mi * 2048;
            if constexpr (t % 3 == 0) o0 = (edge >> mi) & 1u ? CG_T_ZERO : o0;
            if constexpr (t % 3 == 2) o0 = (edge >> (8 + mi)) & 1u ? CG_T_ZERO : o0;
            xf[0][mi] = *reinterpret_cast<const v8_t*>(smem + o0);
            xf[1][mi] = *reinterpret_cast<const v8_t*>(smem + (o0 ^ 64));
          }
          // piece k goes after read number ceil((k + 1) * NR / (NPC + 1))
#pragma unroll
          for (int k = 0; k < NPC; ++k)
            if (((k + 1) * NR + NPC) / (NPC + 1) == r + 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (k < CG_WP) issue_w_piece(rsw, st_w, k < CG_WP ? k : 0, sow);
              else if (t == 0 && k == NPC - 1) issue_gb(c + 1, hbn, nxt);
              else issue_halo_piece(hbn, 2 * t + (k - CG_WP < 2 ? k - CG_WP : 0));
              __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (PP) {
          // Y: this wave's weight pieces of the NEXT K step (issued in the previous tap) have landed, its fragments of this
          // one are in registers
          constexpr int V = NXT((t + 8) % 9) + CG_WP + NX;
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(V) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NMODE != 3) __builtin_amdgcn_s_setprio(1);
        if constexpr (NT && NMODE == 1) {
          // One matrix instruction per slot, and behind each a fixed slice of the strip's normalisation, pinned by
          // sched_barrier(0) -- left to the scheduler (sched_group_barrier) the 64 VALU instructions ended up BEHIND the
          // burst as one dependent chain of LDS round trips and transcendental latencies (~600 cycles per strip, nothing
          // hidden; profiles/r04_conv_gn_variants.txt).  The program works on two channel pairs at a time (four
          // independent chains, a dependent instruction at least two slots = 32 cycles behind its producer, 12 live
          // registers besides the strip itself), one transcendental per slot at most:
          //   U unpack, F x * scale + shift, M * -log2(e), X exp2 (one per slot), A 1 +, R rcp (one per slot), S *, C pack
          // for pairs (0, 1), then (2, 3) with the second half of the (scale, shift) row loaded meanwhile; LDS store last.
          constexpr int NM = 2 * MI * NI, NG = 41;
          const uint32_t vmask = ~(uint32_t)(hpix[NT ? t - 3 : 0] >> 31);   // all ones inside the image, else 0
          const char* const sp = smem + CG_T_SCSH + kslot * 64;
          float xa[4], ea[4];
          u32x4_t no;
          auto micro = [&](auto GG) __attribute__((always_inline)) {
            constexpr int g = decltype(GG)::value;
            if constexpr (g == 40) {
              *reinterpret_cast<u32x4_t*>(np) = no;
            } else {
              constexpr int h = g / 20, gg = g % 20, p0 = 2 * h, p1 = 2 * h + 1;
              if constexpr (gg == 0) { xa[0] = E::lo(nv[p0]); xa[1] = E::hi(nv[p0]); }
              if constexpr (gg == 1) { xa[2] = E::lo(nv[p1]); xa[3] = E::hi(nv[p1]); }
              if constexpr (gg == 2) { xa[0] = __builtin_fmaf(xa[0], ss0[0], ss0[1]); xa[1] = __builtin_fmaf(xa[1], ss0[2], ss0[3]); }
              if constexpr (gg == 3) { xa[2] = __builtin_fmaf(xa[2], ss1[0], ss1[1]); xa[3] = __builtin_fmaf(xa[3], ss1[2], ss1[3]); }
              if constexpr (gg == 4) { ea[0] = xa[0] * -1.44269504088896340736f; ea[1] = xa[1] * -1.44269504088896340736f; }
              if constexpr (gg == 5) { ea[2] = xa[2] * -1.44269504088896340736f; ea[3] = xa[3] * -1.44269504088896340736f; }
              if constexpr (gg == 6) {
                ea[0] = __builtin_amdgcn_exp2f(ea[0]);
                if constexpr (h == 0) ss0 = *reinterpret_cast<const f32x4_t*>(sp + 32);
              }
              if constexpr (gg == 7) {
                ea[1] = __builtin_amdgcn_exp2f(ea[1]);
                if constexpr (h == 0) ss1 = *reinterpret_cast<const f32x4_t*>(sp + 48);
              }
              if constexpr (gg == 8) ea[2] = __builtin_amdgcn_exp2f(ea[2]);
              if constexpr (gg == 9) ea[3] = __builtin_amdgcn_exp2f(ea[3]);
              if constexpr (gg == 10) { ea[0] += 1.0f; ea[1] += 1.0f; }
              if constexpr (gg == 11) { ea[2] += 1.0f; ea[3] += 1.0f; }
              if constexpr (gg == 12) ea[0] = __builtin_amdgcn_rcpf(ea[0]);
              if constexpr (gg == 13) ea[1] = __builtin_amdgcn_rcpf(ea[1]);
              if constexpr (gg == 14) ea[2] = __builtin_amdgcn_rcpf(ea[2]);
              if constexpr (gg == 15) ea[3] = __builtin_amdgcn_rcpf(ea[3]);
              if constexpr (gg == 16) { xa[0] *= ea[0]; xa[1] *= ea[1]; }
              if constexpr (gg == 17) { xa[2] *= ea[2]; xa[3] *= ea[3]; }
              if constexpr (gg == 18) no[p0] = E::pack2(xa[0], xa[1]) & vmask;   // (a mask, not a select: hipcc turns the
              if constexpr (gg == 19) no[p1] = E::pack2(xa[2], xa[3]) & vmask;   //  select into a branch around the math)
            }
          };
          cg_static_for<NM>([&](auto II) __attribute__((always_inline)) {
            constexpr int i = decltype(II)::value;
            constexpr int ks = i / (NI * MI), ni = (i / MI) % NI, mi = i % MI;
            acc[ni][mi] = E::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int G0 = i * NG / NM, G1 = (i + 1) * NG / NM;
            cg_static_for<G1 - G0>([&](auto JJ) __attribute__((always_inline)) {
              micro(std::integral_constant<int, G0 + decltype(JJ)::value>{});
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          mfma_all(wf, xf);
        }
        if constexpr (NMODE != 3) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      });
      hb ^= 1;
    }
    if (PP && grp == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // =====================================================================================================================
  // phase 2: the 1x1 tail over concat(x3, x4) at the output pixel -- a plain 3-stage LDS-DMA GEMM on the same accumulators
  // =====================================================================================================================
  if (tk_b < tk_e) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone left phase 1's LDS
    const int mrow = m_blk + wave * 8 + lrow;        // this lane's row of X strip 0 (strip i: + 64 i)
    const uint32_t xb3 = (uint32_t)a.M * (uint32_t)a.c3 * 2u, xb4 = a.x4 ? (uint32_t)a.M * (uint32_t)a.c4 * 2u : 0u;
    auto issue2 = [&](int kt, int stage) __attribute__((always_inline)) {
      const bool live = kt < tk_e;
      const bool first = kt < d.ntail3;
      const __amdgpu_buffer_rsrc_t rsx = first ? make_rsrc(a.x3, live ? xb3 : 0u) : make_rsrc(a.x4 ? a.x4 : a.x3, live ? xb4 : 0u);
      const int sox = (first ? kt : kt - d.ntail3) * 128;
      const int csrc = first ? a.c3 : a.c4;          // (per-lane offsets from the scalar: a select between two per-lane
                                                     //  arrays becomes a private-memory table)
      const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.w, live ? wbytes : 0u);
      const int sow = (9 * ctot + kt * 64) * 2;
#pragma unroll
      for (int i = 0; i < XP2; ++i) {
        const int m = mrow + i * 64;
        const int vo = m < a.M ? (m * csrc + kslot * 8) * 2 : (int)PP_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (cg_lds_t)(smem + stage * ST2 + (wave * 8 + i * 64) * 128), 16, vo, sox,
                                                 0, 0);
      }
#pragma unroll
      for (int i = 0; i < CG_WP; ++i) {
        const int vo = vw[i], lo = wlds[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (cg_lds_t)(smem + CG_WOFF + stage * CG_WST + lo), 16, vo, sow, 0, 0);
      }
    };
    issue2(tk_b, 0);
    issue2(tk_b + 1, 1);
    const int xrow0 = wm * (MI * 16) + r16;
    int stage = 0;
#pragma unroll 1
    for (int kt = tk_b; kt < tk_e; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P2) : "memory");
      int nstage = stage + 2;
      if (nstage >= 3) nstage -= 3;
      issue2(kt + 2, nstage);
      const char* xs = smem + stage * ST2;
      const char* ws = smem + CG_WOFF + stage * CG_WST;
      v8_t wf[2][NI], xf[2][MI];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = ((ks * 4 + g) ^ fsw) << 4;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ks][ni] = *reinterpret_cast<const v8_t*>(ws + (wrow0 + ni * 16) * 128 + so);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf[ks][mi] = *reinterpret_cast<const v8_t*>(xs + (xrow0 + mi * 16) * 128 + so);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mfma_all(wf, xf);
      __builtin_amdgcn_s_setprio(0);
      stage = stage + 1 == 3 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // =====================================================================================================================
  // epilogue: 64-row passes through LDS (the staged form of gemm.hip): lane holds columns n = .. + 4 (lane >> 4) + {0..3}
  // of row m = .. + (lane & 15); thread = fixed 8-column strip on the read-back side
  // =====================================================================================================================
  const bool splitk = splits > 1;
  const bool gns = !splitk && (a.gn_acc[0] || a.gn_acc[1]);
  const int my_pass = (wm * (MI * 16)) / EPI_ROWS;
  const int my_row0 = (wm * (MI * 16)) % EPI_ROWS;
  float gcs[8], gcq[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { gcs[jj] = 0.f; gcq[jj] = 0.f; }
  constexpr int EC = BN / 8, ER = T / EC, EP = (EPI_ROWS + ER - 1) / ER;
  const int c8 = tid % EC, r0 = tid / EC;
  const int n = n_blk + c8 * 8;
#pragma unroll 1
  for (int pass = 0; pass < BM / EPI_ROWS; ++pass) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS free: main loop (pass 0) / previous read-out finished
    if (my_pass == pass) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          *reinterpret_cast<f32x4_t*>(smem + (my_row0 + mi * 16 + r16) * EPI_LD + (wn * (NI * 16) + ni * 16 + 4 * g) * 4) =
              acc[ni][mi];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int m0 = m_blk + pass * EPI_ROWS;
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
        // all residual loads of the pass are issued BEFORE the math so that their latencies overlap
        u32x4_t r1[EP], r2[EP];
#pragma unroll
        for (int j = 0; j < EP; ++j) {
          const int row = r0 + j * ER, m = m0 + row;
          const bool ok = row < EPI_ROWS && m < a.M;
          r1[j] = (ok && a.res1) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + (size_t)((a.res1_wrap_rows > 0 && m >= a.res1_wrap_rows) ? m - a.res1_wrap_rows : m) * a.ldres1 + n)
                                 : u32x4_t{0u, 0u, 0u, 0u};
          r2[j] = (ok && a.res2) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n)
                                 : u32x4_t{0u, 0u, 0u, 0u};
        }
        f32x4_t bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          bs0 = *reinterpret_cast<const f32x4_t*>(a.bias + n);
          bs1 = *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
        }
        if (a.rowvec) {                                 // (one batch item per tile)
          const float* rv = a.rowvec + (size_t)(m_blk / a.rows_per_batch) * a.ld_rowvec + n;
          bs0 += *reinterpret_cast<const f32x4_t*>(rv);
          bs1 += *reinterpret_cast<const f32x4_t*>(rv + 4);
        }
#pragma unroll
        for (int j = 0; j < EP; ++j) {
          const int row = r0 + j * ER, m = m0 + row;
          if (row < EPI_ROWS && m < a.M) {
            f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32);
            f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + row * EPI_LD + c8 * 32 + 16);
            v0 += bs0;
            v1 += bs1;
            v0 *= a.scale;
            v1 *= a.scale;
            v0[0] += E::lo(r1[j][0]) + E::lo(r2[j][0]); v0[1] += E::hi(r1[j][0]) + E::hi(r2[j][0]);
            v0[2] += E::lo(r1[j][1]) + E::lo(r2[j][1]); v0[3] += E::hi(r1[j][1]) + E::hi(r2[j][1]);
            v1[0] += E::lo(r1[j][2]) + E::lo(r2[j][2]); v1[1] += E::hi(r1[j][2]) + E::hi(r2[j][2]);
            v1[2] += E::lo(r1[j][3]) + E::lo(r2[j][3]); v1[3] += E::hi(r1[j][3]) + E::hi(r2[j][3]);
            u32x4_t o;
            o[0] = E::pack2(v0[0], v0[1]); o[1] = E::pack2(v0[2], v0[3]);
            o[2] = E::pack2(v1[0], v1[1]); o[3] = E::pack2(v1[2], v1[3]);
            *reinterpret_cast<u32x4_t*>((uint16_t*)a.out + (size_t)m * a.ldo + n) = o;
            if (gns) {   // per-column moments of the values as stored, over this thread's rows of the tile
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const float lo = E::lo(o[jj]), hi = E::hi(o[jj]);
                gcs[2 * jj] += lo; gcq[2 * jj] += lo * lo;
                gcs[2 * jj + 1] += hi; gcq[2 * jj + 1] += hi * hi;
              }
            }
          }
        }
      }
    }
  }
  if (gns) {
    // per-thread column moments -> LDS [row-thread][column] -> the 160 column threads fold the ER row-threads in fixed order
    // and add into the groups' integer slots -> one 64-bit atomic per (consumer, group) to the global accumulators
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + RS_OFF);
    if (tid < 2 * GN_SLOTS * 2) slots[tid] = 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (r0 < ER && n < a.N) {
      float* dstp = reinterpret_cast<float*>(smem) + ((size_t)r0 * BN + c8 * 8) * 2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        *reinterpret_cast<f32x4_t*>(dstp + 4 * jj) = f32x4_t{gcs[2 * jj], gcq[2 * jj], gcs[2 * jj + 1], gcq[2 * jj + 1]};
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
    gn_flush(a, slots, m_blk, n_blk, ncols, tid);
  }
  // (ABI v21) the split-K combine by the workgroup that arrives last at its tile (gemm_combine.h)
  if (splitk && a.tile_ctr) {
    static_assert(fc_lds_bytes(BM) <= CG_LDS, "fused combine staging must fit in the kernel's LDS");
    splitk_fused_combine<BM, T, EDT>(a, smem, m_blk, n_blk, blockIdx.x, blockIdx.y, splits, tid);
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct CGChoice {
  int bm, splitk;
};

// the loader's geometry: stride 1, no upsample, tiles of whole image rows inside one image, halo tile <= 384 pixels
bool cg_shape_ok(const PPGemmArgs& a, int bm) {
  const int hw = a.hout * a.wout;
  return bm % a.wout == 0 && hw % bm == 0 && bm + 2 * a.wout <= CG_HALO_PX;
}

// what the halo-tile kernel needs of a conv whatever stands in front of it
bool cg_geometry_ok(const PPGemmArgs& a) {
  if (a.x_mode != PP_X_CONV3X3 || !pp_dt_ok(a.dtype)) return false;
  if (a.stride != 1 || a.wout < 8 || (a.wout & 7)) return false;
  if (a.up ? (a.hout != 2 * a.hin || a.wout != 2 * a.win) : (a.hin != a.hout || a.win != a.wout)) return false;
  const int ctot = a.c1 + a.c2;
  if (a.c1 <= 0 || a.c1 % 64 || a.c2 % 64 || (a.c2 > 0 && !a.x2)) return false;
  if (a.c3 < 0 || a.c4 < 0 || a.c3 % 64 || a.c4 % 64 || (a.c3 > 0 && !a.x3) || (a.c4 > 0 && (!a.x4 || a.c3 == 0))) return false;
  if (a.K != 9 * ctot + a.c3 + a.c4 || a.M != a.batch * a.hout * a.wout || a.M <= 0 || a.N <= 0) return false;
  if (a.N % 8 || a.ldo % 8 || a.out_f32 || a.out_vt || a.act != PP_ACT_NONE || a.ln_stats || a.row_stats_out) return false;
  if ((a.res1 && a.ldres1 % 8) || (a.res2 && a.ldres2 % 8)) return false;
  if (a.rows_per_batch != a.hout * a.wout || a.out_dup_rows > 0 || a.w_batch_stride > 0 || a.vec_batch_stride > 0) return false;
  if ((uint64_t)a.batch * a.hin * a.win * (uint64_t)(a.c1 > a.c2 ? a.c1 : a.c2) * 2u >= 0x80000000ull) return false;
  if ((uint64_t)a.N * (uint64_t)a.K * 2u >= 0x80000000ull) return false;
  if ((uint64_t)a.M * (uint64_t)(a.c3 > a.c4 ? a.c3 : a.c4) * 2u >= 0x80000000ull) return false;
  for (int k = 0; k < 2; ++k)
    if (a.gn_acc[k] && (a.gn_cg[k] < 8 || a.gn_groups[k] <= 0 || a.gn_c0[k] < 0)) return false;
  return cg_shape_ok(a, 256) || cg_shape_ok(a, 128) || cg_shape_ok(a, 64);
}

// norm -> SiLU -> conv3x3 in one launch (the request carries the producers' statistics of its input)
bool cg_fused_ok(const PPGemmArgs& a) {
  if (!a.gn_in_acc || !a.gn_in_gb || a.up || !cg_geometry_ok(a)) return false;
  if (a.gn_in_silu != 1 || a.gn_in_groups <= 0 || a.gn_in_groups > 32) return false;
  return (a.c1 + a.c2) % a.gn_in_groups == 0;
}

// A plain conv3x3 (input already normalised, or none: conv behind an apply launch) on the SAME halo-tile loop without the
// normalisation (NMODE = 2): every input pixel crosses the 64 B/clk global -> LDS path once per tile (+ halo rows) instead
// of once per tap, 52 against 58 us at 64x64 (K = 2880), 58 against 65 at 32x32 (K = 5760) beside the tap-major kernel
// (tools/conv_gn_shapes.py, profiles/r06_conv_raw.txt).  Upsample2D's conv (a.up: nearest 2x in front of the conv) as well:
// the halo tile is gathered from the half-resolution source.  Taken where the automatic choice is asked for (tile = AUTO) and
// the image is at least 16 wide: at 8x8 the launches are split-K weight streams on 64-row tiles and the tap-major kernel's
// 128-row tiles in N-major order win (+1.7 % on the step with this loop there).
// (lab) PP_CONV_RAW = the smallest image width routed here (0 = never)
bool cg_raw_ok(const PPGemmArgs& a) {
  static const int min_w = pp_lab_env("PP_CONV_RAW", 16);
  if (a.gn_in_acc || a.gn_in_gb || a.tile != PP_TILE_AUTO || min_w <= 0 || a.wout < min_w) return false;
  return cg_geometry_ok(a);
}
bool cg_supported(const PPGemmArgs& a) { return a.gn_in_acc ? cg_fused_ok(a) : cg_raw_ok(a); }

CGChoice cg_choose(const PPGemmArgs& a) {
  const int tn = (a.N + 159) / 160;
  auto tiles = [&](int bm) { return (a.M / bm) * tn; };
  CGChoice c{0, 1};
  // explicit tile request (PP_TILE_256x160 / 128x160 / 64x160), else: the largest tile that still fills the chip;
  // failing that the largest tile, split over the channel chunks
  const int want = a.tile == PP_TILE_256x160 ? 256 : a.tile == PP_TILE_128x160 ? 128 : a.tile == PP_TILE_64x160 ? 64 : 0;
  const int nch0 = (a.c1 + a.c2) / 64;
  // long K on half a chip's worth of 256-row tiles: two K splits of the 256-row tile (40 MFMAs per wave and K step, one
  // halo strip in three normalised per output row less) beat one pass of 128-row tiles from ~13 chunks on, combine
  // included -- 179 against 211 us at K = 17280, 135 / 147 at 11520, 105 / 116 at 8640, a draw at 5760 (32x32 level,
  // profiles/r04_rejected_experiments.txt item 8).  (lab) PP_CONV_GN_SK2 = the chunk count from which, 0 = never
  static const int sk2_from = pp_lab_env("PP_CONV_GN_SK2", 12);
#ifdef PP_LAB
  // (lab) PP_CONV_GN_W16 = 10 * BM + splits: the tile / split-K form of the 16x16-level launches (ships: 256 rows x 4 splits)
  static const int w16 = pp_lab_env("PP_CONV_GN_W16", 0);
  if (w16 > 0 && a.wout == 16 && !want && a.splitk <= 0 && cg_shape_ok(a, w16 / 10)) {
    c.bm = w16 / 10;
    c.splitk = w16 % 10;
    if (c.splitk > nch0) c.splitk = nch0;
    return c;
  }
#endif
  if (want && cg_shape_ok(a, want)) c.bm = want;
  else if (cg_shape_ok(a, 256) && tiles(256) >= 224) c.bm = 256;
  else if (sk2_from > 0 && nch0 >= sk2_from && a.splitk <= 0 && cg_shape_ok(a, 256) && tiles(256) * 2 >= 224 &&
           tiles(256) < 224 && cg_shape_ok(a, 128) && tiles(128) >= 224) {
    c.bm = 256;
    c.splitk = 2;
    return c;
  }
  else if (cg_shape_ok(a, 128) && tiles(128) >= 224) c.bm = 128;
  // a quarter of the chip's worth of 256-row tiles or less (the 16x16 level at batch 8: 64 tiles x 4 splits) and a SHORT K:
  // 128-row tiles with half the splits (half the fp32 slabs to write and combine, the same 256 workgroups) -- 53 against
  // 57 us at K = 5760.  From K = 11520 on the 256-row tile wins on a fast box of the pool (77 / 83, 123 / 143 us at
  // K = 23040; step 8.73 against 8.83 ms) and LOSES on a slow one (10.76 against 10.62 ms): profiles/r05_w16_tiles.txt.
  // (lab) PP_CONV_GN_W16 = 10 * rows + splits forces a form at W = 16
  else if (cg_shape_ok(a, 256) && cg_shape_ok(a, 128) && tiles(256) * 4 <= 256 && nch0 <= 10 && a.splitk <= 0) c.bm = 128;
  else if (cg_shape_ok(a, 256)) c.bm = 256;
  else if (cg_shape_ok(a, 128)) c.bm = 128;
  else c.bm = 64;
  const int nch = (a.c1 + a.c2) / 64;
  int sk = 1;
  while (tiles(c.bm) * sk * 2 <= 256 && nch / (sk * 2) >= 2 && sk < 8) sk *= 2;
  c.splitk = a.splitk > 0 ? a.splitk : sk;
  if (c.splitk > 8) c.splitk = 8;
  if (c.splitk > nch) c.splitk = nch;
  return c;
}

template <int BM, int HPW, bool PP, int NMODE, int EDT>
int cg_launch(const PPGemmArgs& a, int splitk, hipStream_t st) {
  auto kern = pp_conv_gn_kernel<BM, HPW, PP, NMODE, EDT>;
  if (pp_func_lds(reinterpret_cast<const void*>(kern), CG_LDS, "hipFuncSetAttribute(conv_gn)") != PP_OK) return PP_ERR_LAUNCH;
  CGDerived d;
  d.tiles_m = a.M / BM;
  d.tiles_n = (a.N + CG_BN - 1) / CG_BN;
  d.n_major = (d.tiles_m < d.tiles_n && d.tiles_m <= 8) ? 1 : 0;
  d.nch1 = a.c1 / 64;
  d.nch = (a.c1 + a.c2) / 64;
  d.ntail3 = a.c3 / 64;
  d.ntail = (a.c3 + a.c4) / 64;
  d.hp = BM + 2 * a.wout;
  d.cg = a.gn_in_groups > 0 ? (a.c1 + a.c2) / a.gn_in_groups : 1;
  d.inv_cg = 1.0f / (float)d.cg;
  hipLaunchKernelGGL(kern, dim3(d.tiles_m * d.tiles_n, splitk, 1), dim3(CG_T), CG_LDS, st, a, d);
  PP_CHECK_LAUNCH("pp_conv_gn_kernel");
  return PP_OK;
}

// halo strips per wave for a tile: ceil((BM + 2 W) / 64), rounded up to an instantiated count
//   BM = 256: W = 64 -> 6, W = 32 / 16 -> 5;  BM = 128: W = 128 -> 6, 64 -> 4, <= 32 -> 3;  BM = 64: W = 64 -> 3, <= 32 -> 2
template <int BM, bool PP, int NMODE, int EDT>
int cg_launch_hpw(const PPGemmArgs& a, int splitk, hipStream_t st) {
  const int need = (BM + 2 * a.wout + 63) / 64;
  if constexpr (BM == 256) {
    if (need <= 5) return cg_launch<256, 5, PP, NMODE, EDT>(a, splitk, st);
    return cg_launch<256, 6, PP, NMODE, EDT>(a, splitk, st);
  } else if constexpr (BM == 128) {
    if (need <= 3) return cg_launch<128, 3, PP, NMODE, EDT>(a, splitk, st);
    if (need <= 4) return cg_launch<128, 4, PP, NMODE, EDT>(a, splitk, st);
    return cg_launch<128, 6, PP, NMODE, EDT>(a, splitk, st);
  } else {
    if (need <= 2) return cg_launch<64, 2, PP, NMODE, EDT>(a, splitk, st);
    return cg_launch<64, 3, PP, NMODE, EDT>(a, splitk, st);
  }
}

// (lab build) PP_CONV_GN_PP=0: the lock-step main loop instead of the ping-pong one; PP_CONV_GN_NMODE=0: the halo
// normalisation in the read phase instead of inside the MFMA burst; 2 / 3: timing probes (see the kernel)
template <int EDT>
int cg_dispatch(const PPGemmArgs& a, const CGChoice& c, hipStream_t st) {
#ifdef PP_LAB
  static const int pp = pp_lab_env("PP_CONV_GN_PP", 1), nm = pp_lab_env("PP_CONV_GN_NMODE", 1);
  if (!pp || nm != 1) {
#define CG_CASE(BM_)                                                                        \
    case BM_:                                                                               \
      if (!pp) return cg_launch_hpw<BM_, false, 1, EDT>(a, c.splitk, st);                   \
      if (nm == 0) return cg_launch_hpw<BM_, true, 0, EDT>(a, c.splitk, st);                \
      if (nm == 2) return cg_launch_hpw<BM_, true, 2, EDT>(a, c.splitk, st);                \
      return cg_launch_hpw<BM_, true, 3, EDT>(a, c.splitk, st);
    switch (c.bm) {
      CG_CASE(256)
      CG_CASE(128)
      default:
      CG_CASE(64)
    }
#undef CG_CASE
  }
#endif
  if (!a.gn_in_acc) {                 // plain conv: the loop without the normalisation (cg_raw_ok)
    switch (c.bm) {
      case 256: return cg_launch_hpw<256, true, 2, EDT>(a, c.splitk, st);
      case 128: return cg_launch_hpw<128, true, 2, EDT>(a, c.splitk, st);
      default: return cg_launch_hpw<64, true, 2, EDT>(a, c.splitk, st);
    }
  }
  switch (c.bm) {
    case 256: return cg_launch_hpw<256, true, 1, EDT>(a, c.splitk, st);
    case 128: return cg_launch_hpw<128, true, 1, EDT>(a, c.splitk, st);
    default: return cg_launch_hpw<64, true, 1, EDT>(a, c.splitk, st);
  }
}

}  // namespace

// entry points used by gemm.hip's pp_gemm_bf16 / pp_gemm_workspace_bytes (one C-ABI call per conv, whichever kernel runs)
// does this conv3x3 request run on the halo-tile kernel: the fused norm (asked for by gn_in_*; pp_gemm_bf16 refuses what the
// kernel cannot do, never a silent fallback) or a plain conv the library routes there (cg_raw_ok)
bool pp_conv_gn_wanted(const PPGemmArgs& a) { return a.x_mode == PP_X_CONV3X3 && (a.gn_in_acc != nullptr || cg_raw_ok(a)); }
int pp_conv_gn_splitk(const PPGemmArgs& a) { return cg_supported(a) ? cg_choose(a).splitk : 0; }
int pp_conv_gn_bm(const PPGemmArgs& a) { return cg_supported(a) ? cg_choose(a).bm : 0; }
int pp_conv_gn_run(const PPGemmArgs& a, hipStream_t st) {
  if (!cg_supported(a)) return PP_ERR_UNSUPPORTED;
  const CGChoice c = cg_choose(a);
  if (c.splitk > 1 && !a.workspace) return PP_ERR_WORKSPACE;
  return a.dtype == PP_DT_F16 ? cg_dispatch<PP_DT_F16>(a, c, st) : cg_dispatch<PP_DT_BF16>(a, c, st);
}

// 1: the fused norm -> SiLU -> conv launch (gn_in_* set); 2: a plain conv3x3 that pp_gemm_bf16 routes to the same halo-tile
// loop without the normalisation; 0: neither (the tap-major implicit GEMM runs it)
extern "C" int pp_conv_gn_supported(const PPGemmArgs* args) {
  if (!args) return 0;
  return cg_fused_ok(*args) ? 1 : cg_raw_ok(*args) ? 2 : 0;
}

// Where the fused launch beats pp_groupnorm_apply_acc + plain conv inside the UNet step (same-box A/Bs of the headline
// benchmark and per-launch timings, profiles/r04_conv_gn_variants.txt): the normalisation is ~600 wave cycles per strip
// that no placement hides behind the matrix pipe, repeated per N tile (N / 160) and per halo row ((BM + 2 W) / BM):
//   class 1  W >= 64 (256-row tiles of 4 image rows, N = 320: 3x redundancy): every shape, -10 .. -15 us per conv
//   class 2  W = 32, at most five 64-channel chunks, one source: short K, the removed apply launch outweighs it
//   class 4  W = 16 (N = 1280): draws per launch, wins the launch boundary
//   class 8  W = 32 otherwise (concatenated inputs / long tails, 128-row tiles): +10 .. +40 us per launch back to back,
//            a draw inside the step (9.61 against 9.62 ms) -- fused for the nine launches it removes
//   Round 6: the plain conv behind an apply launch runs on this file's loop WITHOUT the normalisation (cg_raw_ok) instead of
//   the tap-major kernel -- 52 against 58 us at 64x64 (K = 2880), 58 against 65 at 32x32 (K = 5760) -- and the apply kernel
//   uses the hardware reciprocal in its SiLU (14.3 -> 11.4 us on the 21 MB tensors of the 64x64 level).  Classes 1, 2 and 8
//   flip: back to back the 64x64 level costs 707 + 148 us per forward against 947 fused, the 32x32 level 769 + 114 against
//   973; headline step, same box: everything fused 8.474 ms, classes 2 and 8 unfused 8.404, class 1 as well 8.380 (-1.1 %,
//   twenty launches MORE), class 4 too -- at 16x16 most applies ride in the producers' split-K combines -- 8.854 against
//   8.921 (-0.75 %; the tap-major kernel there: -0.32 %); the raw loop at 8x8: +1.7 % (profiles/r06_conv_raw.txt).
//   So NO class is fused in the plans any more: the fused launch stays an operator for a caller that asks for it
//   (gn_in_*), checked by tests/test_conv_gn_gpu.py as before.
//   class 16 W <= 8: 64-row tiles, one workgroup per CU at 10 MFMAs per wave and K step: +10 .. +26 us, NOT fused
//            (9.76 against 9.61 ms per step with it)
// same-box step times, masks 0 / 1 / 5 / 7 / 15 / 31: 9.75 / 9.64 / 9.62 / 9.62 / 9.61 / 9.76 ms (251 .. 207 launches)
// (lab build: PP_CONV_GN_ROUTE = bit mask of the classes to fuse)
extern "C" int pp_conv_gn_preferred(const PPGemmArgs* args) {
  if (!args || !cg_fused_ok(*args)) return 0;
  const PPGemmArgs& a = *args;
  const int nch = (a.c1 + a.c2) / 64;
  int cls;
  if (a.win >= 64) cls = 1;
  else if (a.win == 32) cls = (nch <= 5 && a.c2 == 0) ? 2 : 8;
  else if (a.win == 16) cls = 4;
  else cls = a.win < 16 ? 16 : 8;
  static const int mask = pp_lab_env("PP_CONV_GN_ROUTE", 0);
  return (mask & cls) ? 1 : 0;
}
