// FeedForward of BasicTransformerBlock + Transformer2DModel.proj_out at C = 320 (the 64x64 level of SD-1.5; 128x128 on
// config 5) as ONE launch:
//
//     out = proj_out( hs + FF2( h (.) gelu(g) ) ) + x_in (+ side residual),   h | g = FF1(LayerNorm3(hs))
//
// (reference ctor site /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300 -> diffusers 0.27 FeedForward
// (GEGLU) / BasicTransformerBlock.forward `ff(norm3(h)) + h` and Transformer2DModel.proj_out + residual.  The two-launch
// plan is engine.py `_transformer`: `linear_geglu` (M x 2560 x 320, GEGLU in the epilogue: 81 us at 64x64 x 8) and `linear`
// (FF2 . proj_out composed at pack time, K = 1280 + 320: 46 us), which write and re-read the [M][1280] GEGLU tensor --
// 84 MB each way per block -- and whose first GEMM has only five K steps per tile in front of a VALU-heavy epilogue.)
//
// Here the hidden dimension is STREAMED: a workgroup owns 128 rows and walks the 1280 hidden units in 40 chunks of 32;
// per chunk   S = x . W1[chunk]^T   (64 interleaved (h0, h1, g0, g1) columns, K = 320),   act = GEGLU(S)  in the MFMA
// register layout (the epilogue arithmetic of pp_gemm_bf16's EPI = 2),   out += act . W2'[:, chunk]^T   on persistent
// fp32 accumulators; after the last chunk the K tail `hs . W_po^T` (the residual inside the transformer, composed into the
// same GEMM at pack time) and the epilogue of the second GEMM (bias, residuals, 16-bit store, GroupNorm statistics of the
// consumer).  The [M][1280] tensor never exists.
//
// Tiling (the lesson of tfront.hip / xattn_fused.hip: a 16-row wave tile feeds ONE MFMA per LDS fragment read and runs at
// ~20 % matrix-pipe occupancy): FOUR waves per workgroup, ONE per SIMD, each owning 32 rows x ALL columns with the whole
// register file of its SIMD (256 VGPRs + 256 accumulation registers) --
//   * the wave's 32 x 320 input rows are 20 MFMA B fragments held in registers (80 VGPRs) for the whole kernel: they feed
//     every chunk's first GEMM and the K tail; nothing but weights moves through LDS;
//   * out[32][320] = 160 accumulator registers, S double buffered (2 x 32) so that the GEGLU of chunk c - 1 (VALU: ~41
//     instructions per accumulator quad) is hand-interleaved, one slice per MFMA slot, with the first GEMM of chunk c;
//   * every weight fragment read from LDS feeds two MFMAs (both 16-row blocks of the wave);
//   * the accumulator layout of the first GEMM (lane = row, quad = (h0, h1, g0, g1) of two hidden units) IS the B-operand
//     layout of the second GEMM up to a fixed permutation of the hidden index inside every group of 32, applied to W2' when
//     it is packed (engine._kperm_geglu): storage position 8 kg + 2 q + e holds unit 8 q + 2 kg + e.  No exchange between
//     waves, no barrier for the activation.
// Weights stream through a six-stage LDS-DMA ring of 20 KB pieces with ONE s_barrier per piece and a counted vmcnt.  Every
// piece is five sub-tiles of [64 rows][32 k] or one tile of [320 rows][32 k] -- 64-byte rows, 16-byte slots XOR-swizzled by
// -(row >> 2) & 3 (ff_swz), conflict-free ds_read_b128 -- i.e. exactly five DMA instructions per wave and 40 MFMA slots per wave:
//   T1(c, h) = W1 rows [64 c, +64) x k [160 h, +160)         (two per chunk),
//   T2(c)    = W2' rows [0, 320) x hidden units [32 c, +32)  (one per chunk; the K tail: ten more of the same shape),
// issued five pieces ahead.  130 pieces, 5200 MFMAs (16x16x32) per wave.
#include <type_traits>
#include <utility>

#include "pp_common.h"
#include "gemm_gn.h"

namespace {

constexpr int FF_C = 320, FF_BM = 128, FF_HID = 4 * FF_C, FF_HC = 32, FF_NCH = FF_HID / FF_HC;   // 40 chunks of 32 units
constexpr int FF_K2 = FF_HID + FF_C;                       // row length of W2' = [W_po W_ff2 | W_po]
constexpr int FF_STAGE = 20 * 1024, FF_NS = 6, FF_PD = FF_NS - 1;
constexpr int FF_TAB = FF_NS * FF_STAGE;                   // fp32 tables: bias1[2560] | colsum1[2560]
constexpr int FF_LDS = FF_TAB + 2 * 2 * FF_HID * 4;
constexpr int FF_EPI_ROWS = 64, FF_EPI_LD = FF_C * 4 + 16; // epilogue staging: fp32 rows, +16 B bank spread
constexpr int FF_FD = 3;                                   // weight fragments read ahead of the MFMA pair that uses them
constexpr bool FF_DSP = false;
static_assert(FF_EPI_ROWS * FF_EPI_LD <= FF_TAB, "epilogue staging fits in the ring");
static_assert(FF_LDS <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) void* ff_lds_ptr_t;

// XOR swizzle of the 16-byte slot of a [16 rows][64 B] tile, by the row's block of four t = row >> 2.  gfx950 serves a
// ds_read_b128 in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md,
// LDS): a group holds rows 0-3 and 12-15 of k-slot g and rows 4-11 of k-slot g ^ 1, so the four rows r, r + 4, r + 8, r + 12
// (same 16-byte slot column of the 256-byte bank row) need {f(0), f(1) ^ 1, f(2) ^ 1, f(3)} distinct.  Round 5's f(t) = t --
// derived for contiguous groups -- gives {0, 0, 3, 3}: every fragment read 2-way conflicted (8 LDS cycles instead of 4; the
// "unexplained" 12.4 M conflict cycles = 45 % of LDS-active cycles of profiles/r05_ff_fused_pmc.txt).  f(t) = -t mod 4 gives
// {0, 2, 3, 1} and keeps the ds_write_b32 of the activation tile at its unavoidable 2 ways (tools/lds_conflicts.py).
PP_DEVINL int ff_swz(int t) { return (-t) & 3; }

struct FFArgs {
  PPGemmArgs g;            // the second GEMM as pp_gemm_bf16 takes it: x2 = hs [M][c2 = 320], w = W2' (hidden index permuted),
  //                          bias, res1 (+ res1_wrap_rows), res2, scale, out / ldo, gn_acc subscriptions; x1 is NOT read
  const uint16_t* w1;      // [2560][320], rows interleaved (h0, h1, g0, g1), LayerNorm gamma folded in
  const float* b1;         // [2560]
  const float* cs1;        // [2560] column sums of w1 (folded LayerNorm) or NULL
  const float* ln_stats;   // [M][ln_tiles][2] row moments of hs or NULL
  int ln_tiles;
  float ln_eps;
};

template <int... I, class F>
PP_DEVINL void ff_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void ff_static_for(F&& f) { ff_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// DBG (lab build only, timing experiments: results are garbage): 1 no weight DMA, 2 no MFMAs, 4 no fragment reads,
// 8 no GEGLU slices, 16 no barriers
// FD: weight fragments read ahead of the MFMA pair that uses them; DSP: the five DMA instructions of a step spread over its
// MFMA slots (one every eight) instead of a burst behind the barrier
template <int EDT, int DBG = 0, int FD = FF_FD, bool DSP = FF_DSP>
__global__ void __launch_bounds__(256, 1) ff_fused_kernel(const FFArgs fa) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PPGemmArgs& a = fa.g;
  float* tabs = reinterpret_cast<float*>(smem + FF_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m_blk = lid * FF_BM;

  // ---- the wave's 32 input rows as B fragments (row block mi, k block s: eight consecutive channels per lane)
  v8_t xf[2][10];
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m_blk + wave * 32 + mi * 16 + r16;
    const uint16_t* xr = reinterpret_cast<const uint16_t*>(a.x2) + (size_t)m * a.ldx2 + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) xf[mi][s] = *reinterpret_cast<const v8_t*>(xr + 32 * s);
    if (fa.ln_stats) {
      const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(fa.ln_stats) + (size_t)m * fa.ln_tiles;
      float sm = 0.f, sq = 0.f;
      for (int t = 0; t < fa.ln_tiles; ++t) { const f32x2_t v = pm[t]; sm += v[0]; sq += v[1]; }
      mean[mi] = sm * (1.0f / FF_C);
      rstd[mi] = rsqrtf(fmaxf(sq * (1.0f / FF_C) - mean[mi] * mean[mi], 0.f) + fa.ln_eps);
    }
  }
  for (int i = tid; i < 2 * FF_HID / 4; i += 256) {
    *reinterpret_cast<f32x4_t*>(tabs + 4 * i) = reinterpret_cast<const f32x4_t*>(fa.b1)[i];
    *reinterpret_cast<f32x4_t*>(tabs + 2 * FF_HID + 4 * i) =
        fa.cs1 ? reinterpret_cast<const f32x4_t*>(fa.cs1)[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  // ---- weight pieces: a DMA instruction moves 16 rows x 64 B; lane -> (row of the strip, k-slot it FETCHES so that its
  //      lane-linear LDS slot is swizzled by ff_swz(row >> 2))
  const int ksl = ((lane & 3) ^ ff_swz(lane >> 4)) * 8;
  const int vA = ((16 * wave + (lane >> 2)) * FF_C + ksl) * 2;       // T1: instruction i = sub-tile (k block) i, strip = wave
  const int vB = ((16 * wave + (lane >> 2)) * FF_K2 + ksl) * 2;      // T2: instruction i = rows 64 i + 16 wave ..
  // producer side.  The piece sequence is
  //     T1(0,0) T1(0,1) | T1(c,0) T1(c,1) T2(c-1)  for c = 1 .. 39 | tail(0) .. tail(9) | T2(39) | dead pieces ...
  // and every consumer step issues the piece five positions ahead.  WHICH kind that is is a compile-time property of the
  // code position (the last two loop iterations are peeled for it): no branch, no select -- a branch would split the main
  // loop into basic blocks, and LLVM then sinks the hand-interleaved GEGLU slices out of the MFMA slots into the block that
  // uses their results.  Dead pieces (zero-sized descriptor) write zeros into a stage nobody reads again: every wave keeps
  // exactly five loads per piece in flight and the counted wait is a constant.
  const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(fa.w1, 2u * FF_HID * FF_C * 2u);
  const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(a.w, (uint32_t)FF_C * FF_K2 * 2u);
  const __amdgpu_buffer_rsrc_t rs_0 = make_rsrc(fa.w1, 0u);
  // ISS: 0 = T1 piece at byte offset `so` = (64 c * 320 + 160 h) * 2 of W1, 1 = T2 piece at column offset `so` = 2 k of W2',
  //      2 = dead
  auto issue_one = [&](auto ISS_, int so, int stage, int i) __attribute__((always_inline)) {
    constexpr int ISS = decltype(ISS_)::value;
    if constexpr (DBG & 1) return;
    char* st = smem + stage * FF_STAGE + wave * 1024;
    if constexpr (ISS == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (ff_lds_ptr_t)(st + i * 4096), 16, vA, so + i * 64, 0, 0);
    else if constexpr (ISS == 1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (ff_lds_ptr_t)(st + i * 4096), 16, vB, so + i * (64 * FF_K2 * 2), 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_0, (ff_lds_ptr_t)(st + i * 4096), 16, vA, i * 64, 0, 0);
  };
  auto issue = [&](auto ISS_, int so, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_one(ISS_, so, stage, i);
  };
  auto so_t1 = [](int c, int h) __attribute__((always_inline)) { return (c * 64 * FF_C + h * 160) * 2; };
  auto so_t2 = [](int c) __attribute__((always_inline)) { return c * FF_HC * 2; };
  auto so_tail = [](int t) __attribute__((always_inline)) { return (FF_HID + 32 * t) * 2; };

  f32x4_t S[2][4][2];        // [parity of the chunk][16-column block][16-row block]
  f32x4_t out[20][2];
  uint32_t pf[2][4];         // GEGLU of a chunk as the B fragment of its second-GEMM piece: [row block][register]
#pragma unroll
  for (int nj = 0; nj < 20; ++nj)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) out[nj][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- GEGLU of chunk `c` (accumulators Sq) in 80 slices, one per MFMA slot: quad qd = slice / 10 = (column block ni,
  //      row block mi), stage = slice % 10.  Arithmetic = the EPI = 2 epilogue of pp_gemm_kernel_v2 + gelu_fast_f.
  f32x4_t g_cs = {0.f, 0.f, 0.f, 0.f}, g_b = {0.f, 0.f, 0.f, 0.f}, g_v = {0.f, 0.f, 0.f, 0.f};
  float g_t2 = 0.f, g_t3 = 0.f, g_p2 = 0.f, g_p3 = 0.f, g_e2 = 0.f, g_e3 = 0.f;
  auto gelu_slice = [&](auto GS, const f32x4_t (&Sq)[4][2], int c) __attribute__((always_inline)) {
    constexpr int gs = decltype(GS)::value;
    if constexpr (gs < 80) {
      constexpr int qd = gs / 10, stg = gs % 10, ni = qd >> 1, mi = qd & 1;
      if constexpr (stg == 0) {
        if constexpr (mi == 0) {
          const int n = c * 64 + ni * 16 + 4 * g;
          g_b = *reinterpret_cast<const f32x4_t*>(tabs + n);
          g_cs = *reinterpret_cast<const f32x4_t*>(tabs + 2 * FF_HID + n);
        }
        g_v = Sq[ni][mi] - g_cs * mean[mi];
      } else if constexpr (stg == 1) {
        g_v = g_v * rstd[mi] + g_b;
      } else if constexpr (stg == 2) {
        g_t2 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g_v[2]), 0.3275911f * 0.70710678118654752440f, 1.0f));
        g_t3 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g_v[3]), 0.3275911f * 0.70710678118654752440f, 1.0f));
      } else if constexpr (stg == 3) {
        g_p2 = __builtin_fmaf(g_t2, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
        g_p3 = __builtin_fmaf(g_t3, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * 1.421413741f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * 1.421413741f);
      } else if constexpr (stg == 4) {
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * -0.284496736f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * -0.284496736f);
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * 0.254829592f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * 0.254829592f);
      } else if constexpr (stg == 5) {
        g_e2 = __builtin_amdgcn_exp2f(g_v[2] * g_v[2] * (-0.5f * 1.44269504088896340736f));
        g_e3 = __builtin_amdgcn_exp2f(g_v[3] * g_v[3] * (-0.5f * 1.44269504088896340736f));
      } else if constexpr (stg == 6) {
        g_p2 = g_p2 * g_t2 * g_e2;
        g_p3 = g_p3 * g_t3 * g_e3;
      } else if constexpr (stg == 7) {
        g_p2 = __builtin_fmaf(-__builtin_fabsf(g_v[2]), g_p2, __builtin_fmaxf(g_v[2], 0.f));
        g_p3 = __builtin_fmaf(-__builtin_fabsf(g_v[3]), g_p3, __builtin_fmaxf(g_v[3], 0.f));
      } else if constexpr (stg == 8) {
        pf[mi][ni] = E::pack2(g_v[0] * g_p2, g_v[1] * g_p3);
      }
    }
  };

  // ---- consumer side.  One step = counted wait + barrier (everyone's piece has landed, everyone left the stage that is
  //      refilled now), refill, then the piece's MFMA slots.
  int rd = 0, wr = 0;                                      // ring stage of the piece consumed next / refilled by this step
  auto step_begin = [&](auto ISS_, int so) __attribute__((always_inline)) -> const char* {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (FF_PD - 1)) : "memory");
    if constexpr (!(DBG & 16)) asm volatile("s_barrier" ::: "memory");
    wr = rd == 0 ? FF_NS - 1 : rd - 1;                     // piece p + PD goes where piece p - 1 was read
    if constexpr (!DSP) issue(ISS_, so, wr);
    const char* st = smem + rd * FF_STAGE;
    rd = rd + 1 == FF_NS ? 0 : rd + 1;
    return st;
  };
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const int soB = r16 * 64 + ((g ^ ff_swz(r16 >> 2)) << 4);

  // one piece = 20 weight fragments (16 rows x 32 k each, at k * 1024 in the stage) x the wave's 2 row blocks = 40 MFMA slots.
  //   KIND 0: first GEMM, half H of chunk c (parity P): fragment k = (k block kq = k >> 2, column block ni = k & 3), B = the
  //           input rows' k block 5 H + kq; GELU: the slots carry the GEGLU slices of chunk c - 1 (accumulators S[1 - P]);
  //   KIND 1: second GEMM piece: fragment k = output column block, B = (b0, b1); TAIL >= 0: K-tail piece number TAIL, whose
  //           slots carry the GEGLU slices of the last chunk.
  //   (ISS_, so): the piece issued five positions ahead (see `issue`).
  auto piece = [&](auto KIND_, auto H_, auto P_, auto GELU_, auto TAIL_, const v8_t b0, const v8_t b1, int c, auto ISS_, int so)
                   __attribute__((always_inline)) {
    constexpr int KIND = decltype(KIND_)::value, H = decltype(H_)::value, P = decltype(P_)::value, TAIL = decltype(TAIL_)::value;
    constexpr bool GELU = decltype(GELU_)::value;
    const char* st = step_begin(ISS_, so);
    v8_t frag[20];
    auto fetch = [&](int k) __attribute__((always_inline)) {
      if constexpr (DBG & 4) frag[k] = xf[0][k % 10];
      else frag[k] = *reinterpret_cast<const v8_t*>(st + k * 1024 + soB);
    };
#pragma unroll
    for (int k = 0; k < FD; ++k) fetch(k);
    ff_static_for<40>([&](auto F_) __attribute__((always_inline)) {
      constexpr int f = decltype(F_)::value, k = f >> 1, mi = f & 1;
      if constexpr (mi == 0 && k + FD < 20) fetch(k + FD);
      if constexpr (DSP && f % 8 == 3) issue_one(ISS_, so, wr, f / 8);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KIND == 0) {
        constexpr int kq = k >> 2, ni = k & 3;
        if constexpr (DBG & 2) S[P][ni][mi] += __builtin_bit_cast(f32x4_t, frag[k]);
        else S[P][ni][mi] = E::mfma16(frag[k], xf[mi][5 * H + kq], (H == 0 && kq == 0) ? zero4 : S[P][ni][mi]);
        if constexpr (GELU && !(DBG & 8)) gelu_slice(std::integral_constant<int, H * 40 + f>{}, S[1 - P], c - 1);
      } else {
        if constexpr (DBG & 2) out[k][mi] += __builtin_bit_cast(f32x4_t, frag[k]);
        else out[k][mi] = E::mfma16(frag[k], mi ? b1 : b0, out[k][mi]);
        if constexpr (TAIL >= 0 && !(DBG & 8))
          gelu_slice(std::integral_constant<int, TAIL * 40 + f>{}, S[(FF_NCH - 1) & 1], FF_NCH - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto pfrag = [&](int mi) __attribute__((always_inline)) -> v8_t {
    return __builtin_bit_cast(v8_t, u32x4_t{pf[mi][0], pf[mi][1], pf[mi][2], pf[mi][3]});
  };
  constexpr std::integral_constant<int, -1> NOTAIL{};
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::integral_constant<bool, true> YES{};
  constexpr std::integral_constant<bool, false> NO{};
  constexpr std::integral_constant<int, 2> I2{};
  const v8_t vnone = xf[0][0];                             // (placeholder operand of the KIND 0 instantiations)
  // two chunks (c odd, c + 1) = six pieces; MODE 0: steady state (look-ahead stays inside the chunk loop, c <= 35),
  // MODE 1: c = 37 (positions 4, 5 look ahead to the K tail), MODE 2: c = 39 (one chunk; everything ahead is K tail)
  auto body = [&](auto MODE_, int c) __attribute__((always_inline)) {
    constexpr int MODE = decltype(MODE_)::value;
    if constexpr (MODE < 2) {
      piece(I0, I0, I1, YES, NOTAIL, vnone, vnone, c, I1, so_t2(c));
      piece(I0, I1, I1, YES, NOTAIL, vnone, vnone, c, I0, so_t1(c + 2, 0));
      piece(I1, I0, I0, NO, NOTAIL, pfrag(0), pfrag(1), c, I0, so_t1(c + 2, 1));
      piece(I0, I0, I0, YES, NOTAIL, vnone, vnone, c + 1, I1, so_t2(c + 1));
      if constexpr (MODE == 0) {
        piece(I0, I1, I0, YES, NOTAIL, vnone, vnone, c + 1, I0, so_t1(c + 3, 0));
        piece(I1, I0, I0, NO, NOTAIL, pfrag(0), pfrag(1), c + 1, I0, so_t1(c + 3, 1));
      } else {
        piece(I0, I1, I0, YES, NOTAIL, vnone, vnone, c + 1, I1, so_tail(0));
        piece(I1, I0, I0, NO, NOTAIL, pfrag(0), pfrag(1), c + 1, I1, so_tail(1));
      }
    } else {
      piece(I0, I0, I1, YES, NOTAIL, vnone, vnone, c, I1, so_tail(2));
      piece(I0, I1, I1, YES, NOTAIL, vnone, vnone, c, I1, so_tail(3));
      piece(I1, I0, I0, NO, NOTAIL, pfrag(0), pfrag(1), c, I1, so_tail(4));
    }
  };

  // prologue: pieces 0 .. 4 = T1(0,0) T1(0,1) T1(1,0) T1(1,1) T2(0)
  issue(I0, so_t1(0, 0), 0);
  issue(I0, so_t1(0, 1), 1);
  issue(I0, so_t1(1, 0), 2);
  issue(I0, so_t1(1, 1), 3);
  issue(I1, so_t2(0), 4);
  piece(I0, I0, I0, NO, NOTAIL, vnone, vnone, 0, I0, so_t1(2, 0));
  piece(I0, I1, I0, NO, NOTAIL, vnone, vnone, 0, I0, so_t1(2, 1));
#pragma unroll 1
  for (int c = 1; c + 4 < FF_NCH; c += 2) body(I0, c);     // c = 1, 3, .., 35
  body(I1, FF_NCH - 3);
  body(I2, FF_NCH - 1);
  // K tail: out += hs . W_po^T from the resident input fragments, carrying the GEGLU of the last chunk; then its second GEMM
  ff_static_for<10>([&](auto T_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value;
    if constexpr (t + 5 < 10) piece(I1, I0, I0, NO, T_, xf[0][t], xf[1][t], 0, I1, so_tail(t + 5));
    else if constexpr (t + 5 == 10) piece(I1, I0, I0, NO, T_, xf[0][t], xf[1][t], 0, I1, so_t2(FF_NCH - 1));
    else piece(I1, I0, I0, NO, T_, xf[0][t], xf[1][t], 0, I2, 0);
  });
  piece(I1, I0, I0, NO, NOTAIL, pfrag(0), pfrag(1), 0, I2, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the dead tail prefetches before LDS is reused

  // ================= epilogue of the second GEMM: two 64-row passes through LDS (the staged epilogue of gemm.hip) =========
  constexpr int EC = FF_C / 8, ER = 256 / EC, EP = (FF_EPI_ROWS + ER - 1) / ER;       // 40 strips, 6 rows per sweep, 11 sweeps
  const int c8 = tid % EC, r0 = tid / EC;
  const int n = c8 * 8;
  const bool gns = a.gn_acc[0] || a.gn_acc[1];
  float gcs[8], gcq[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { gcs[jj] = 0.f; gcq[jj] = 0.f; }
#pragma unroll 1
  for (int pass = 0; pass < FF_BM / FF_EPI_ROWS; ++pass) {
    asm volatile("s_barrier" ::: "memory");              // LDS free: main loop (pass 0) / previous read-out finished
    if ((wave >> 1) == pass) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 20; ++nj)
          *reinterpret_cast<f32x4_t*>(smem + ((wave & 1) * 32 + mi * 16 + r16) * FF_EPI_LD + (nj * 16 + 4 * g) * 4) = out[nj][mi];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int m0 = m_blk + pass * FF_EPI_ROWS;
    if (r0 < ER) {
      const int r1shift = (a.res1_wrap_rows > 0 && m0 >= a.res1_wrap_rows) ? a.res1_wrap_rows : 0;
      u32x4_t r1[EP], r2[EP];
#pragma unroll
      for (int j = 0; j < EP; ++j) {
        const int row = r0 + j * ER, m = m0 + row;
        const bool ok = row < FF_EPI_ROWS;
        r1[j] = (ok && a.res1) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + (size_t)(m - r1shift) * a.ldres1 + n)
                               : u32x4_t{0u, 0u, 0u, 0u};
        r2[j] = (ok && a.res2) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n)
                               : u32x4_t{0u, 0u, 0u, 0u};
      }
      f32x4_t bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        bs0 = *reinterpret_cast<const f32x4_t*>(a.bias + n);
        bs1 = *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
      }
#pragma unroll
      for (int j = 0; j < EP; ++j) {
        const int row = r0 + j * ER, m = m0 + row;
        if (row < FF_EPI_ROWS) {
          f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + row * FF_EPI_LD + c8 * 32);
          f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + row * FF_EPI_LD + c8 * 32 + 16);
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
          if (gns) {       // per-column moments of the values as stored, over this thread's rows
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
  if (gns) {
    // GroupNorm statistics of the consumer: per-thread column moments -> LDS [row thread][column] -> the column threads fold
    // the ER row threads in fixed order and feed the groups' integer slots (gemm_gn.h), one slot set per 160-column half
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + FF_TAB);
    constexpr int SL = 2 * GN_SLOTS * 2;
    for (int i = tid; i < 2 * SL; i += 256) slots[i] = 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (... and everyone finished reading the staged tile)
    if (r0 < ER) {
      float* dstp = reinterpret_cast<float*>(smem) + ((size_t)r0 * FF_C + c8 * 8) * 2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        *reinterpret_cast<f32x4_t*>(dstp + 4 * jj) = f32x4_t{gcs[2 * jj], gcq[2 * jj], gcs[2 * jj + 1], gcq[2 * jj + 1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int col = tid; col < FF_C; col += 256) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int r = 0; r < ER; ++r) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(smem + ((size_t)r * FF_C + col) * 8);
        sm += v[0];
        sq += v[1];
      }
      const int h = col >= 160 ? 1 : 0;
      gn_column(a, slots + h * SL, h * 160, col - h * 160, sm, sq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    gn_flush(a, slots, m_blk, 0, 160, tid);
    gn_flush(a, slots + SL, m_blk, 160, 160, tid);
  }
}


// ---- the same launch with EIGHT waves (two per SIMD, 256 registers each): wave = (row block wm = wave & 3: 32 rows,
// column half wn = wave >> 2).  A 4-wave workgroup is one in-order instruction stream per SIMD: matrix instructions, the
// GEGLU's vector instructions, fragment reads and DMA issue are served one after the other and their times ADD (measured:
// 30 us of MFMA + 40 GEGLU + 26 DMA issue + 6 reads + 8 launch = 112 us; profiles/r05_ff_fused.txt).  Two waves per SIMD
// split every one of those streams in two and let one wave's vector / DMA work sit under the other's matrix work:
//   * first GEMM: the wave computes its HALF of the chunk's 64 interleaved columns (2 column blocks x 2 row blocks: 32
//     accumulator registers with the double buffer), every weight fragment still feeds two MFMAs;
//   * the GEGLU values (16 hidden units per wave) are exchanged with the partner wave through a 1 KB LDS tile per 16-row
//     block -- natural unit order, so W2' needs NO index permutation here -- behind the barrier the next piece starts with
//     anyway;
//   * second GEMM: the wave owns 160 of the 320 output columns (80 accumulator registers), B operand = the two activation
//     fragments of its rows read back from LDS; the K tail takes the resident input fragments as before;
//   * a piece is 20 DMA instructions: three per wave (waves 4-7: two + a dead one), vmcnt(3 (PD - 1)).
constexpr int FF8_ACT = FF_TAB + 2 * 2 * FF_HID * 4;      // [wm][mi] 16 rows x 64 B (32 hidden units), slots swizzled like the weights
constexpr int FF8_DUMMY = FF8_ACT + 8 * 1024;             // landing zone of the dead DMA instruction of waves 4-7
constexpr int FF8_LDS = FF8_DUMMY + 4 * 1024;
static_assert(FF8_LDS <= 160 * 1024, "LDS budget");

template <int EDT, int DBG = 0, int FD = 1>
__global__ void __launch_bounds__(512, 2) ff_fused8_kernel(const FFArgs fa) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PPGemmArgs& a = fa.g;
  float* tabs = reinterpret_cast<float*>(smem + FF_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int r16 = lane & 15, g = lane >> 4;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m_blk = lid * FF_BM;

  v8_t xf[2][10];
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m_blk + wm * 32 + mi * 16 + r16;
    const uint16_t* xr = reinterpret_cast<const uint16_t*>(a.x2) + (size_t)m * a.ldx2 + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) xf[mi][s] = *reinterpret_cast<const v8_t*>(xr + 32 * s);
    if (fa.ln_stats) {
      const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(fa.ln_stats) + (size_t)m * fa.ln_tiles;
      float sm = 0.f, sq = 0.f;
      for (int t = 0; t < fa.ln_tiles; ++t) { const f32x2_t v = pm[t]; sm += v[0]; sq += v[1]; }
      mean[mi] = sm * (1.0f / FF_C);
      rstd[mi] = rsqrtf(fmaxf(sq * (1.0f / FF_C) - mean[mi] * mean[mi], 0.f) + fa.ln_eps);
    }
  }
  for (int i = tid; i < 2 * FF_HID / 4; i += 512) {
    *reinterpret_cast<f32x4_t*>(tabs + 4 * i) = reinterpret_cast<const f32x4_t*>(fa.b1)[i];
    *reinterpret_cast<f32x4_t*>(tabs + 2 * FF_HID + 4 * i) =
        fa.cs1 ? reinterpret_cast<const f32x4_t*>(fa.cs1)[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  // ---- weight pieces (the layouts of the 4-wave kernel): instruction j of a piece moves 16 rows x 64 B to stage + j KB;
  //      T1: j = 4 * (k block) + strip, T2: j = 16-row block.  Wave w issues j = w, 8 + w and (w < 4) 16 + w.
  const int ksl = ((lane & 3) ^ ff_swz(lane >> 4)) * 8;
  const int vA = ((16 * wm + (lane >> 2)) * FF_C + ksl) * 2;         // T1: strip = wave & 3 for each of its instructions
  const int vB = ((lane >> 2) * FF_K2 + ksl) * 2;                     // T2: the 16-row block rides in the scalar offset
  const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(fa.w1, 2u * FF_HID * FF_C * 2u);
  const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(a.w, (uint32_t)FF_C * FF_K2 * 2u);
  const __amdgpu_buffer_rsrc_t rs_0 = make_rsrc(fa.w1, 0u);
  // the third instruction exists for waves 0-3 only: the others aim a zero-sized descriptor at a dummy KB (uniform vmcnt)
  const __amdgpu_buffer_rsrc_t rs_1c = make_rsrc(fa.w1, wn == 0 ? 2u * FF_HID * FF_C * 2u : 0u);
  const __amdgpu_buffer_rsrc_t rs_2c = make_rsrc(a.w, wn == 0 ? (uint32_t)FF_C * FF_K2 * 2u : 0u);
  const int j0 = wave, j1 = 8 + wave, j2 = 16 + wm;
  auto issue = [&](auto ISS_, int so, int stage) __attribute__((always_inline)) {
    constexpr int ISS = decltype(ISS_)::value;
    if constexpr (DBG & 1) return;
    char* st = smem + stage * FF_STAGE;
    char* st2 = wn == 0 ? st + j2 * 1024 : smem + FF8_DUMMY + wm * 1024;
    if constexpr (ISS == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (ff_lds_ptr_t)(st + j0 * 1024), 16, vA, so + (j0 >> 2) * 64, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (ff_lds_ptr_t)(st + j1 * 1024), 16, vA, so + (j1 >> 2) * 64, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1c, (ff_lds_ptr_t)st2, 16, vA, so + 4 * 64, 0, 0);
    } else if constexpr (ISS == 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (ff_lds_ptr_t)(st + j0 * 1024), 16, vB, so + j0 * (16 * FF_K2 * 2), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (ff_lds_ptr_t)(st + j1 * 1024), 16, vB, so + j1 * (16 * FF_K2 * 2), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2c, (ff_lds_ptr_t)st2, 16, vB, so + j2 * (16 * FF_K2 * 2), 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_0, (ff_lds_ptr_t)(st + j0 * 1024), 16, vA, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_0, (ff_lds_ptr_t)(st + j1 * 1024), 16, vA, 64, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_0, (ff_lds_ptr_t)st2, 16, vA, 128, 0, 0);
    }
  };
  auto so_t1 = [](int c, int h) __attribute__((always_inline)) { return (c * 64 * FF_C + h * 160) * 2; };
  auto so_t2 = [](int c) __attribute__((always_inline)) { return c * FF_HC * 2; };
  auto so_tail = [](int t) __attribute__((always_inline)) { return (FF_HID + 32 * t) * 2; };

  f32x4_t S[2][2][2];        // [parity of the chunk][column block of this wave's half][row block]
  f32x4_t out[10][2];
#pragma unroll
  for (int nj = 0; nj < 10; ++nj)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) out[nj][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // GEGLU of chunk `c`, this wave's 4 quads, 40 slices (stage = slice % 10 as in the 4-wave kernel); stage 8 stores the two
  // 16-bit values into the activation tile of the wave's row block
  const int soB = r16 * 64 + ((g ^ ff_swz(r16 >> 2)) << 4);
  char* const act_w = smem + FF8_ACT + (wm * 2) * 1024 + r16 * 64 + 4 * g;      // + mi KB + swizzled slot of the unit pair
  f32x4_t g_cs = {0.f, 0.f, 0.f, 0.f}, g_b = {0.f, 0.f, 0.f, 0.f}, g_v = {0.f, 0.f, 0.f, 0.f};
  float g_t2 = 0.f, g_t3 = 0.f, g_p2 = 0.f, g_p3 = 0.f, g_e2 = 0.f, g_e3 = 0.f;
  auto gelu_slice = [&](auto GS, const f32x4_t (&Sq)[2][2], int c) __attribute__((always_inline)) {
    constexpr int gs = decltype(GS)::value;
    if constexpr (gs < 40) {
      constexpr int qd = gs / 10, stg = gs % 10, ni = qd >> 1, mi = qd & 1;
      if constexpr (stg == 0) {
        {
          const int n = c * 64 + wn * 32 + ni * 16 + 4 * g;
          g_b = *reinterpret_cast<const f32x4_t*>(tabs + n);
          g_cs = *reinterpret_cast<const f32x4_t*>(tabs + 2 * FF_HID + n);
        }
        g_v = Sq[ni][mi] - g_cs * mean[mi];
      } else if constexpr (stg == 1) {
        g_v = g_v * rstd[mi] + g_b;
      } else if constexpr (stg == 2) {
        g_t2 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g_v[2]), 0.3275911f * 0.70710678118654752440f, 1.0f));
        g_t3 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(g_v[3]), 0.3275911f * 0.70710678118654752440f, 1.0f));
      } else if constexpr (stg == 3) {
        g_p2 = __builtin_fmaf(g_t2, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
        g_p3 = __builtin_fmaf(g_t3, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * 1.421413741f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * 1.421413741f);
      } else if constexpr (stg == 4) {
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * -0.284496736f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * -0.284496736f);
        g_p2 = __builtin_fmaf(g_t2, g_p2, 0.5f * 0.254829592f);
        g_p3 = __builtin_fmaf(g_t3, g_p3, 0.5f * 0.254829592f);
      } else if constexpr (stg == 5) {
        g_e2 = __builtin_amdgcn_exp2f(g_v[2] * g_v[2] * (-0.5f * 1.44269504088896340736f));
        g_e3 = __builtin_amdgcn_exp2f(g_v[3] * g_v[3] * (-0.5f * 1.44269504088896340736f));
      } else if constexpr (stg == 6) {
        g_p2 = g_p2 * g_t2 * g_e2;
        g_p3 = g_p3 * g_t3 * g_e3;
      } else if constexpr (stg == 7) {
        g_p2 = __builtin_fmaf(-__builtin_fabsf(g_v[2]), g_p2, __builtin_fmaxf(g_v[2], 0.f));
        g_p3 = __builtin_fmaf(-__builtin_fabsf(g_v[3]), g_p3, __builtin_fmaxf(g_v[3], 0.f));
      } else if constexpr (stg == 8) {
        // units 16 wn + 8 ni + 2 g + {0, 1} of the chunk: 16-byte slot 2 wn + ni of the row, swizzled by ff_swz(row >> 2)
        *reinterpret_cast<uint32_t*>(act_w + mi * 1024 + (((2 * wn + ni) ^ ff_swz(r16 >> 2)) << 4)) =
            E::pack2(g_v[0] * g_p2, g_v[1] * g_p3);
      }
    }
  };

  int rd = 0, wr = 0;
  auto step_begin = [&](auto ISS_, int so) __attribute__((always_inline)) -> const char* {
    // (lgkmcnt: this wave's activation stores of the previous piece are in LDS before anyone passes the barrier)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * (FF_PD - 1)) : "memory");
    if constexpr (!(DBG & 16)) asm volatile("s_barrier" ::: "memory");
    wr = rd == 0 ? FF_NS - 1 : rd - 1;
    issue(ISS_, so, wr);
    const char* st = smem + rd * FF_STAGE;
    rd = rd + 1 == FF_NS ? 0 : rd + 1;
    return st;
  };
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  // one piece = this wave's 10 weight fragments x its 2 row blocks = 20 MFMA slots.
  //   KIND 0: first GEMM, half H of chunk c (parity P): fragment k = (k block kq = k >> 1, column block ni = k & 1 of the wave's
  //           half), B = the input rows' k block 5 H + kq; GELU: the slots carry the GEGLU slices of chunk c - 1;
  //   KIND 1: second GEMM: fragment k = output column block 10 wn + k; ACT: B = the activation tile (else the operands b0 / b1:
  //           the K tail, TAIL >= 0 carrying the GEGLU slices of the last chunk)
  auto piece = [&](auto KIND_, auto H_, auto P_, auto GELU_, auto TAIL_, auto ACT_, const v8_t b0, const v8_t b1, int c, auto ISS_,
                   int so) __attribute__((always_inline)) {
    constexpr int KIND = decltype(KIND_)::value, H = decltype(H_)::value, P = decltype(P_)::value, TAIL = decltype(TAIL_)::value;
    constexpr bool GELU = decltype(GELU_)::value, ACT = decltype(ACT_)::value;
    const char* st = step_begin(ISS_, so);
    v8_t ba = b0, bb = b1;
    if constexpr (ACT) {
      ba = *reinterpret_cast<const v8_t*>(smem + FF8_ACT + (wm * 2 + 0) * 1024 + soB);
      bb = *reinterpret_cast<const v8_t*>(smem + FF8_ACT + (wm * 2 + 1) * 1024 + soB);
    }
    v8_t frag[10];
    auto fetch = [&](int k) __attribute__((always_inline)) {
      if constexpr (DBG & 4) frag[k] = xf[0][k % 10];
      else if constexpr (KIND == 0) frag[k] = *reinterpret_cast<const v8_t*>(st + (k >> 1) * 4096 + (2 * wn + (k & 1)) * 1024 + soB);
      else frag[k] = *reinterpret_cast<const v8_t*>(st + (10 * wn + k) * 1024 + soB);
    };
#pragma unroll
    for (int k = 0; k < FD; ++k) fetch(k);
    ff_static_for<20>([&](auto F_) __attribute__((always_inline)) {
      constexpr int f = decltype(F_)::value, k = f >> 1, mi = f & 1;
      if constexpr (mi == 0 && k + FD < 10) fetch(k + FD);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KIND == 0) {
        constexpr int kq = k >> 1, ni = k & 1;
        if constexpr (DBG & 2) S[P][ni][mi] += __builtin_bit_cast(f32x4_t, frag[k]);
        else S[P][ni][mi] = E::mfma16(frag[k], xf[mi][5 * H + kq], (H == 0 && kq == 0) ? zero4 : S[P][ni][mi]);
        if constexpr (GELU && !(DBG & 8)) gelu_slice(std::integral_constant<int, H * 20 + f>{}, S[1 - P], c - 1);
      } else {
        if constexpr (DBG & 2) out[k][mi] += __builtin_bit_cast(f32x4_t, frag[k]);
        else out[k][mi] = E::mfma16(frag[k], mi ? bb : ba, out[k][mi]);
        if constexpr (TAIL >= 0 && !(DBG & 8))
          gelu_slice(std::integral_constant<int, TAIL * 20 + f>{}, S[(FF_NCH - 1) & 1], FF_NCH - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  constexpr std::integral_constant<int, -1> NOTAIL{};
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::integral_constant<int, 2> I2{};
  constexpr std::integral_constant<bool, true> YES{};
  constexpr std::integral_constant<bool, false> NO{};
  const v8_t vnone = xf[0][0];
  auto body = [&](auto MODE_, int c) __attribute__((always_inline)) {
    constexpr int MODE = decltype(MODE_)::value;
    if constexpr (MODE < 2) {
      piece(I0, I0, I1, YES, NOTAIL, NO, vnone, vnone, c, I1, so_t2(c));
      piece(I0, I1, I1, YES, NOTAIL, NO, vnone, vnone, c, I0, so_t1(c + 2, 0));
      piece(I1, I0, I0, NO, NOTAIL, YES, vnone, vnone, c, I0, so_t1(c + 2, 1));
      piece(I0, I0, I0, YES, NOTAIL, NO, vnone, vnone, c + 1, I1, so_t2(c + 1));
      if constexpr (MODE == 0) {
        piece(I0, I1, I0, YES, NOTAIL, NO, vnone, vnone, c + 1, I0, so_t1(c + 3, 0));
        piece(I1, I0, I0, NO, NOTAIL, YES, vnone, vnone, c + 1, I0, so_t1(c + 3, 1));
      } else {
        piece(I0, I1, I0, YES, NOTAIL, NO, vnone, vnone, c + 1, I1, so_tail(0));
        piece(I1, I0, I0, NO, NOTAIL, YES, vnone, vnone, c + 1, I1, so_tail(1));
      }
    } else {
      piece(I0, I0, I1, YES, NOTAIL, NO, vnone, vnone, c, I1, so_tail(2));
      piece(I0, I1, I1, YES, NOTAIL, NO, vnone, vnone, c, I1, so_tail(3));
      piece(I1, I0, I0, NO, NOTAIL, YES, vnone, vnone, c, I1, so_tail(4));
    }
  };

  issue(I0, so_t1(0, 0), 0);
  issue(I0, so_t1(0, 1), 1);
  issue(I0, so_t1(1, 0), 2);
  issue(I0, so_t1(1, 1), 3);
  issue(I1, so_t2(0), 4);
  piece(I0, I0, I0, NO, NOTAIL, NO, vnone, vnone, 0, I0, so_t1(2, 0));
  piece(I0, I1, I0, NO, NOTAIL, NO, vnone, vnone, 0, I0, so_t1(2, 1));
#pragma unroll 1
  for (int c = 1; c + 4 < FF_NCH; c += 2) body(I0, c);
  body(I1, FF_NCH - 3);
  body(I2, FF_NCH - 1);
  ff_static_for<10>([&](auto T_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value;
    if constexpr (t + 5 < 10) piece(I1, I0, I0, NO, T_, NO, xf[0][t], xf[1][t], 0, I1, so_tail(t + 5));
    else if constexpr (t + 5 == 10) piece(I1, I0, I0, NO, T_, NO, xf[0][t], xf[1][t], 0, I1, so_t2(FF_NCH - 1));
    else piece(I1, I0, I0, NO, T_, NO, xf[0][t], xf[1][t], 0, I2, 0);
  });
  piece(I1, I0, I0, NO, NOTAIL, YES, vnone, vnone, 0, I2, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ================= epilogue: two 64-row passes through LDS; 512 threads = 40 column strips x 12 rows per sweep ===========
  constexpr int EC = FF_C / 8, ER = 512 / EC, EP = (FF_EPI_ROWS + ER - 1) / ER;
  const int c8 = tid % EC, r0 = tid / EC;
  const int n = c8 * 8;
  const bool gns = a.gn_acc[0] || a.gn_acc[1];
  float gcs[8], gcq[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { gcs[jj] = 0.f; gcq[jj] = 0.f; }
#pragma unroll 1
  for (int pass = 0; pass < FF_BM / FF_EPI_ROWS; ++pass) {
    asm volatile("s_barrier" ::: "memory");
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 10; ++nj)
          *reinterpret_cast<f32x4_t*>(smem + ((wm & 1) * 32 + mi * 16 + r16) * FF_EPI_LD + (wn * 160 + nj * 16 + 4 * g) * 4) = out[nj][mi];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int m0 = m_blk + pass * FF_EPI_ROWS;
    if (r0 < ER) {
      const int r1shift = (a.res1_wrap_rows > 0 && m0 >= a.res1_wrap_rows) ? a.res1_wrap_rows : 0;
      u32x4_t r1[EP], r2[EP];
#pragma unroll
      for (int j = 0; j < EP; ++j) {
        const int row = r0 + j * ER, m = m0 + row;
        const bool ok = row < FF_EPI_ROWS;
        r1[j] = (ok && a.res1) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res1 + (size_t)(m - r1shift) * a.ldres1 + n)
                               : u32x4_t{0u, 0u, 0u, 0u};
        r2[j] = (ok && a.res2) ? *reinterpret_cast<const u32x4_t*>((const uint16_t*)a.res2 + (size_t)m * a.ldres2 + n)
                               : u32x4_t{0u, 0u, 0u, 0u};
      }
      f32x4_t bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        bs0 = *reinterpret_cast<const f32x4_t*>(a.bias + n);
        bs1 = *reinterpret_cast<const f32x4_t*>(a.bias + n + 4);
      }
#pragma unroll
      for (int j = 0; j < EP; ++j) {
        const int row = r0 + j * ER, m = m0 + row;
        if (row < FF_EPI_ROWS) {
          f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + row * FF_EPI_LD + c8 * 32);
          f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + row * FF_EPI_LD + c8 * 32 + 16);
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
          if (gns) {
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
  if (gns) {
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + FF_TAB);
    constexpr int SL = 2 * GN_SLOTS * 2;
    for (int i = tid; i < 2 * SL; i += 512) slots[i] = 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (r0 < ER) {
      float* dstp = reinterpret_cast<float*>(smem) + ((size_t)r0 * FF_C + c8 * 8) * 2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        *reinterpret_cast<f32x4_t*>(dstp + 4 * jj) = f32x4_t{gcs[2 * jj], gcq[2 * jj], gcs[2 * jj + 1], gcq[2 * jj + 1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < FF_C) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int r = 0; r < ER; ++r) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(smem + ((size_t)r * FF_C + tid) * 8);
        sm += v[0];
        sq += v[1];
      }
      const int h = tid >= 160 ? 1 : 0;
      gn_column(a, slots + h * SL, h * 160, tid - h * 160, sm, sq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    gn_flush(a, slots, m_blk, 0, 160, tid);
    gn_flush(a, slots + SL, m_blk, 160, 160, tid);
  }
}

}  // namespace

extern "C" int pp_ff_fused_supported(int M, int c, int rows_per_batch) {
  return (c == FF_C && M > 0 && M % FF_BM == 0 && rows_per_batch > 0 && rows_per_batch % FF_BM == 0 && M % rows_per_batch == 0)
             ? 1 : 0;
}

extern "C" int pp_ff_fused(const PPGemmArgs* g2, const void* w1, const float* b1, const float* cs1, const float* ln_stats,
                           int ln_tiles, float ln_eps, int w2_kperm, void* stream) {
  if (!g2 || !w1 || !b1) return PP_ERR_BAD_ARG;
  const PPGemmArgs& a = *g2;
  if (!pp_dt_ok(a.dtype) || a.x_mode != PP_X_PLAIN || !a.x2 || !a.w || !a.out) return PP_ERR_BAD_ARG;
  if (a.N != FF_C || a.K != FF_K2 || a.c1 != FF_HID || a.c2 != FF_C) return PP_ERR_UNSUPPORTED;
  if (!pp_ff_fused_supported(a.M, a.N, a.rows_per_batch > 0 ? a.rows_per_batch : FF_BM)) return PP_ERR_UNSUPPORTED;
  if (a.ldx2 < FF_C || (a.ldx2 & 7) || a.ldo < FF_C || (a.ldo & 7) || (a.res1 && (a.ldres1 & 7)) || (a.res2 && (a.ldres2 & 7)))
    return PP_ERR_BAD_ARG;
  if (a.act != PP_ACT_NONE || a.out_f32 || a.out_vt || a.rowvec || a.row_stats_out || a.ln_stats || a.gn_next_out ||
      a.out_dup_rows || a.splitk > 1 || a.gn_in_acc)
    return PP_ERR_UNSUPPORTED;
  if (a.res1_wrap_rows < 0 || (a.res1_wrap_rows > 0 && (!a.res1 || a.M > 2 * a.res1_wrap_rows || a.res1_wrap_rows % 64)))
    return PP_ERR_BAD_ARG;
  if ((a.gn_acc[0] || a.gn_acc[1]) && a.rows_per_batch <= 0) return PP_ERR_BAD_ARG;
  for (int k = 0; k < 2; ++k)
    if (a.gn_acc[k] && (a.gn_cg[k] < 8 || a.gn_groups[k] <= 0 || a.gn_c0[k] < 0)) return PP_ERR_UNSUPPORTED;
  if (ln_stats && (!cs1 || ln_tiles <= 0)) return PP_ERR_BAD_ARG;
  FFArgs fa;
  fa.g = a;
  fa.w1 = (const uint16_t*)w1; fa.b1 = b1; fa.cs1 = ln_stats ? cs1 : nullptr;
  fa.ln_stats = ln_stats; fa.ln_tiles = ln_tiles; fa.ln_eps = ln_eps;
  auto go = [&](auto kern) -> int {
    if (pp_func_lds(reinterpret_cast<const void*>(kern), FF_LDS, "hipFuncSetAttribute(ff_fused)") != PP_OK) return PP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.M / FF_BM), dim3(256), FF_LDS, (hipStream_t)stream, fa);
    PP_CHECK_LAUNCH("ff_fused_kernel");
    return PP_OK;
  };
  if (!w2_kperm) {      // W2' in natural hidden order: the 8-wave kernel (activations exchanged through LDS)
    auto go8 = [&](auto kern, bool) -> int {
      if (pp_func_lds(reinterpret_cast<const void*>(kern), FF8_LDS, "hipFuncSetAttribute(ff_fused8)") != PP_OK) return PP_ERR_LAUNCH;
      hipLaunchKernelGGL(kern, dim3(a.M / FF_BM), dim3(512), FF8_LDS, (hipStream_t)stream, fa);
      PP_CHECK_LAUNCH("ff_fused8_kernel");
      return PP_OK;
    };
#ifdef PP_LAB
    if (a.dtype == PP_DT_BF16) switch (pp_lab_env("PP_FF_DBG", 0)) {
        case 1: return go8(ff_fused8_kernel<PP_DT_BF16, 1>, true);
        case 8: return go8(ff_fused8_kernel<PP_DT_BF16, 8>, true);
        case 9: return go8(ff_fused8_kernel<PP_DT_BF16, 9>, true);
        case 13: return go8(ff_fused8_kernel<PP_DT_BF16, 13>, true);
        case 16: return go8(ff_fused8_kernel<PP_DT_BF16, 16>, true);
        default: break;
      }
#endif
    if (a.dtype == PP_DT_F16) return go8(ff_fused8_kernel<PP_DT_F16>, false);
    return go8(ff_fused8_kernel<PP_DT_BF16>, false);
  }
#ifdef PP_LAB
  if (a.dtype == PP_DT_BF16) switch (pp_lab_env("PP_FF_DBG", 0)) {      // tools/ff_one.py: time the loop with parts removed
      case 8: return go(ff_fused_kernel<PP_DT_BF16, 8>);
      default: break;
    }
  if (a.dtype == PP_DT_BF16) switch (pp_lab_env("PP_FF_VAR", 0)) {      // FD / DSP variants
      default: break;
    }
#endif
  if (a.dtype == PP_DT_F16) return go(ff_fused_kernel<PP_DT_F16>);
  return go(ff_fused_kernel<PP_DT_BF16>);
}
