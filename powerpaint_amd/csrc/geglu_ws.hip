// Wave-specialised GEGLU GEMM for gfx950 (round 3):  out[m][n/2 + j] = h_j * gelu(g_j),  (h, g) = LN-folded x W^T + b
// -- FeedForward.net[0] = GEGLU(Linear(C, 8C)) of every BasicTransformerBlock (reference ctor site
// /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300; diffusers 0.27 `GEGLU`), 16 launches per UNet step.
//
// Why a second kernel.  In pp_gemm_kernel_v2<.., EPI = 2> a tile is  K/64 MFMA steps  ->  GELU epilogue (two exact-erf
// GELUs per accumulator quad: ~25 VALU instructions per output, K-independent), strictly one after the other, and with
// K = 320 .. 1280 the epilogue is as long as the main loop: the matrix pipe idles while the vector ALUs work and vice versa
// (GEGLU at 64x64: 69 us for 53.7 GFLOP = 0.78 PFLOP/s).  MI355X runs an MFMA-only wave and a VALU-only wave of one SIMD
// concurrently (MI355X_MICROARCH.md, "Wave scheduling"), so here the workgroup's eight waves are TWO GROUPS of four (one wave
// of each group per SIMD) that alternate roles at TILE granularity:
//
//     phase p :   group p & 1       : main loop of tile p          (LDS-DMA refills, fragment reads, 40 MFMAs per K step)
//                 group (p - 1) & 1 : GELU epilogue of tile p - 1  from its OWN accumulator registers, stores to HBM
//
// -- no accumulator hand-off through LDS, the epilogue costs nothing as long as it is shorter than a main loop.  A
// workgroup owns one 128-row M tile and a run of consecutive 160-column N tiles; the K steps of all its tiles form ONE
// continuous two-stage LDS-DMA stream (the group in its main loop issues step s + 1 -- the first step of the next tile
// belongs to the OTHER group, which waits for it at its first barrier, before it has issued any store).  `s_barrier` is
// workgroup-wide, so the epilogue is cut into pieces of one 16-column block each and executes the main loop's barriers
// between them.  Same MFMA operand order, same fp32 arithmetic as the EPI = 2 epilogue -> bit-identical results.
//
// Output stores: a lane holds one dword (two outputs) per 16-column block; four blocks are transposed across the four
// 16-lane rows with two v_permlane16_swap + two v_permlane32_swap so that a lane owns the 8 outputs of ONE block:
// one 16-byte store per lane, 64 contiguous bytes per row (the fifth block keeps 4-byte stores).
#include "pp_common.h"

namespace {

constexpr int WS_BM = 128, WS_BN = 160, WS_STAGE = (WS_BM + WS_BN) * 128, WS_LDS = 2 * WS_STAGE;

typedef __attribute__((address_space(3))) void* ws_lds_ptr_t;

// both swaps as inline asm: __builtin_amdgcn_permlane32_swap returned its first result twice in attn_pipe_kernel (ROCm 7.2);
// s_nop 1 = the wait states between a VALU write of an operand and the swap (cdna_hip_programming.md, T21)
PP_DEVINL void swap16(uint32_t& a, uint32_t& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
PP_DEVINL void swap32(uint32_t& a, uint32_t& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }

template <int EDT>
__global__ void __launch_bounds__(512, 2) pp_geglu_ws_kernel(const PPGemmArgs a, int splits, int cnt) {
  typedef typename E16<EDT>::v8 v8_t;
  constexpr int MI = 4, NI = 5;                       // wave tile 64 x 80 of the group's 128 x 160 tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;           // role group, wave inside the group
  const int wm = w4 >> 1, wn = w4 & 1;
  const int KT = a.K >> 6;

  // XCD-aware bijective remap (block b runs on XCD b % 8): the workgroups of one XCD take consecutive ids = the N runs
  // of the same M tiles, so an activation tile is fetched into that XCD's L2 once
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = lid / splits, run = lid - tile_m * splits;
  const int m_blk = tile_m * WS_BM;
  const int n0 = run * cnt;                           // first N tile of this workgroup's run
  const int S = cnt * KT;                             // K steps of the whole run

  // ---- loader (the group in its main loop issues): 36 wave-wide 16-byte DMA pieces per K step (16 of the X tile, 20 of
  // the W tile), 9 per wave; lane -> (row of the 8-row strip, k-slot it FETCHES so that its lane-linear LDS slot is swizzled)
  const int lrow = lane >> 3, kslot = (lane & 7) ^ lrow;
  // wave w4 of the issuing group moves X strips w4 + 4 j (j < 4) and W strips w4 + 4 j (j < 5); a strip = 8 rows
  int vx[4], vw[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) vx[j] = ((m_blk + 8 * (w4 + 4 * j) + lrow) * a.ldx1 + kslot * 8) * 2;
#pragma unroll
  for (int j = 0; j < 5; ++j) vw[j] = ((8 * (w4 + 4 * j) + lrow) * a.K + kslot * 8) * 2;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x1, (uint32_t)a.M * (uint32_t)a.ldx1 * 2u);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.w, (uint32_t)a.N * (uint32_t)a.K * 2u);
  auto issue = [&](int s) __attribute__((always_inline)) {     // K step s of the run -> stage s & 1
    const int t = s / KT, kt = s - t * KT;
    char* st = smem + (s & 1) * WS_STAGE + w4 * (8 * 128);
    const int sox = kt * 128;                                   // (k0 * 2 bytes)
    const int sow = ((n0 + t) * WS_BN * a.K + kt * 64) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (ws_lds_ptr_t)(st + j * (32 * 128)), 16, vx[j], sox, 0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (ws_lds_ptr_t)(st + WS_BM * 128 + j * (32 * 128)), 16, vw[j], sow, 0, 0);
  };

  // fragment read offsets (as pp_gemm_kernel_v2): row = base + (lane & 15), k-slot = ks*4 + (lane >> 4), swizzled by row & 7
  const int frow = lane & 15, fk = lane >> 4, fsw = frow & 7;
  const int xrow0 = wm * (MI * 16) + frow, wrow0 = wn * (NI * 16) + frow;

  f32x4_t acc[NI][MI];
  const int r16 = lane & 15, g = lane >> 4;
  const bool ln = a.ln_stats != nullptr;

  // ---- epilogue state of the tile this group finished last (operands fetched at the start of its epilogue phase)
  f32x2_t mr[MI];                                     // (mean, rstd) of this lane's four rows
  uint32_t od[MI][NI];                                // packed outputs (two per dword) until they are stored
  auto epi_prepare = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float mean = 0.f, rstd = 1.f;
      if (ln) {
        const int m = m_blk + wm * (MI * 16) + mi * 16 + r16;
        const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(a.ln_stats) + (size_t)m * a.ln_tiles;
        float sm = 0.f, sq = 0.f;
        for (int t = 0; t < a.ln_tiles; ++t) { const f32x2_t v = pm[t]; sm += v[0]; sq += v[1]; }
        const float inv = 1.0f / (float)a.ln_dim;
        mean = sm * inv;
        rstd = rsqrtf(fmaxf(sq * inv - mean * mean, 0.f) + a.ln_eps);
      }
      mr[mi] = f32x2_t{mean, rstd};
    }
  };
  // one 16-column block of the finished tile `tn` (N tile index): LN correction, bias, GEGLU -> od[*][ni]
  auto epi_block = [&](int tn, int ni) __attribute__((always_inline)) {
    const int n = tn * WS_BN + wn * (NI * 16) + ni * 16 + 4 * g;
    f32x4_t bs = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bs = *reinterpret_cast<const f32x4_t*>(a.bias + n);
    if (ln) cs = *reinterpret_cast<const f32x4_t*>(a.ln_colsum + n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f32x4_t v = acc[ni][mi];
      if (ln) v = (v - cs * mr[mi][0]) * mr[mi][1];
      v += bs;
      od[mi][ni] = E16<EDT>::pack2(v[0] * gelu_fast_f(v[2]), v[1] * gelu_fast_f(v[3]));
    }
  };
  auto epi_store = [&](int tn) __attribute__((always_inline)) {
    uint16_t* ob = (uint16_t*)a.out + (size_t)(m_blk + wm * (MI * 16) + r16) * a.ldo + tn * (WS_BN / 2) + wn * (NI * 8);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // blocks 0..3: transpose (block x lane-row) so that lane row g holds the four dwords of block g
      uint32_t d0 = od[mi][0], d1 = od[mi][1], d2 = od[mi][2], d3 = od[mi][3];
      swap16(d0, d1);                                 // d0 = [a.g0 b.g0 a.g2 b.g2], d1 = [a.g1 b.g1 a.g3 b.g3]
      swap16(d2, d3);                                 // d2 = [c.g0 d.g0 c.g2 d.g2], d3 = [c.g1 d.g1 c.g3 d.g3]
      swap32(d0, d2);                                 // d0 = [a.g0 b.g0 c.g0 d.g0], d2 = [a.g2 b.g2 c.g2 d.g2]
      swap32(d1, d3);                                 // d1 = [a.g1 b.g1 c.g1 d.g1], d3 = [a.g3 b.g3 c.g3 d.g3]
      uint16_t* row = ob + (size_t)(mi * 16) * a.ldo;
      *reinterpret_cast<u32x4_t*>(row + g * 8) = u32x4_t{d0, d1, d2, d3};
      *reinterpret_cast<uint32_t*>(row + 32 + 2 * g) = od[mi][4];
    }
  };

  auto wait_dma = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto barrier = [&]() __attribute__((always_inline)) { asm volatile("s_barrier" ::: "memory"); };

  // Phase p (KT barriers) = main loop of tile p (group p & 1) beside the epilogue of tile p - 1 (the other group); a
  // group's own sequence is therefore  [group 1: one idle phase]  { main loop of tile t ; epilogue of tile t }  t += 2.
  // The waves that ISSUED the DMAs of K step s wait for them before the barrier that publishes the stage: the group in
  // its main loop for kt >= 1; for kt == 0 the group that has just left its main loop (first barrier of its epilogue
  // phase) -- or group 0 after the prologue.
  if (grp == 0) issue(0);
  else
    for (int kt = 0; kt < KT; ++kt) barrier();
  for (int t = grp; t < cnt; t += 2) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < KT; ++kt) {
      const int s = t * KT + kt;
      if (kt >= 1 || t == 0) wait_dma();
      barrier();
      if (s + 1 < S) issue(s + 1);
      const char* xs = smem + (s & 1) * WS_STAGE;
      const char* ws = xs + WS_BM * 128;
      v8_t xf[2][MI], wf[2][NI];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = ((ks * 4 + fk) ^ fsw) << 4;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ks][ni] = *reinterpret_cast<const v8_t*>(ws + (wrow0 + ni * 16) * 128 + so);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf[ks][mi] = *reinterpret_cast<const v8_t*>(xs + (xrow0 + mi * 16) * 128 + so);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = E16<EDT>::mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- epilogue of tile t, beside the other group's main loop of tile t + 1 (if there is one): one 16-column block
    // per barrier interval (K / 64 >= 5 is a launch condition), the stores behind the last block
    const int tn = n0 + t;
    epi_prepare();
    if (t + 1 < cnt) {
      wait_dma();                                     // (this group issued K step 0 of tile t + 1)
      barrier(); epi_block(tn, 0);
      barrier(); epi_block(tn, 1);
      barrier(); epi_block(tn, 2);
      barrier(); epi_block(tn, 3);
      barrier(); epi_block(tn, 4);
      epi_store(tn);
      for (int kt = NI; kt < KT; ++kt) barrier();
    } else {                                          // last tile of the run: nothing is shared any more
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) epi_block(tn, ni);
      epi_store(tn);
    }
  }
}

}  // namespace

// Conditions (everything else takes pp_gemm_kernel_v2<.., EPI = 2>): plain single-source X, GEGLU, whole 128 x 160 tiles,
// K / 64 >= 5 barriers per tile for the five epilogue pieces, 16-byte aligned output rows, and a decomposition with at
// least two N tiles per workgroup (a run of one tile has nothing to overlap).
bool pp_geglu_ws_ok(const PPGemmArgs& a) {
  if (a.x_mode != PP_X_PLAIN || a.act != PP_ACT_GEGLU || a.c2 != 0 || a.x2) return false;
  if (a.M % WS_BM || a.N % WS_BN || a.K % 64 || a.K / 64 < 5 || a.ldo % 8 || a.ldx1 % 8) return false;
  if (a.out_f32 || a.out_vt || a.res1 || a.res2 || a.rowvec || a.scale != 1.0f || a.row_stats_out) return false;
  if (a.gn_acc[0] || a.gn_acc[1] || a.tile != PP_TILE_AUTO || a.splitk > 1) return false;
  if (a.ln_stats && (!a.ln_colsum || a.ln_tiles <= 0)) return false;
  const int tiles_n = a.N / WS_BN, tiles_m = a.M / WS_BM;
  if (tiles_n < 2) return false;
  if (tiles_m * tiles_n < 512) return false;        // (smaller launches: the tiled kernel's finer decomposition fills the chip better)
  int cnt = tiles_n;
  while (cnt > 2 && (cnt > 16 || tiles_m * (tiles_n / cnt) < 256 || tiles_n % cnt)) --cnt;
  return tiles_n % cnt == 0 && tiles_m * (tiles_n / cnt) >= 256;
}

int pp_geglu_ws_launch(const PPGemmArgs& a, hipStream_t st) {
  const int tiles_n = a.N / WS_BN, tiles_m = a.M / WS_BM;
  // runs of `cnt` consecutive N tiles per workgroup: the longest run that still gives >= 256 workgroups (one per CU),
  // capped at 16 (M = 32768: 256 x 16, M = 8192: 64 M tiles x 4 runs of 8, M = 2048: 16 x 16 runs of 4)
  int cnt = tiles_n;
  while (cnt > 2 && (cnt > 16 || tiles_m * (tiles_n / cnt) < 256 || tiles_n % cnt)) --cnt;
  if (tiles_n % cnt) cnt = 1;
  if (cnt < 2) return PP_ERR_UNSUPPORTED;
  const int splits = tiles_n / cnt;
  static bool attr_set[3] = {false, false, false};
  auto go = [&](auto kern, int slot) -> int {
    if (!attr_set[slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS) != hipSuccess) {
        pp_set_last_error("hipFuncSetAttribute(geglu ws)", hipGetLastError());
        return PP_ERR_LAUNCH;
      }
      attr_set[slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * splits), dim3(512), WS_LDS, st, a, splits, cnt);
    PP_CHECK_LAUNCH("pp_geglu_ws_kernel");
    return PP_OK;
  };
  if (a.dtype == PP_DT_F16) return go(pp_geglu_ws_kernel<PP_DT_F16>, PP_DT_F16);
  return go(pp_geglu_ws_kernel<PP_DT_BF16>, PP_DT_BF16);
}
