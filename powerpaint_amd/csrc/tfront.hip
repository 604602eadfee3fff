// Front end of Transformer2DModel at C = 320 (the 64x64 level of SD-1.5; 128x128 on config 5) as ONE launch:
//
//     n  = GroupNorm(x)                         Transformer2DModel.norm          (eps 1e-6, no activation)
//     hs = proj_in(n)                           1x1 conv = Linear over the channels
//     q | k | v = to_q / to_k / to_v (LayerNorm1(hs))     BasicTransformerBlock.norm1 + attn1 projections
//
// (reference ctor site /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300 -> diffusers 0.27
// Transformer2DModel.forward / BasicTransformerBlock.forward; the unfused plan is engine.py `_transformer`:
// pp_groupnorm_apply_acc 10 us + `conv1x1` 18.8 us + `linear` (LayerNorm folded, V written transposed) 40.5 us per block
// inside the step, 2 x 21 MB of normalised-activation and 21 MB of hs round trips: profiles/r04_step_timeline.txt).
//
// Everything is row-local, so the scheme of xattn_fused.hip applies: one workgroup = 128 rows of one batch item, eight
// waves of 16 rows x ALL columns; the wave's input rows are MFMA B fragments loaded straight from global memory (and
// normalised in registers: x * scale[c] + shift[c] with the group statistics from the producer's accumulators, rounded to
// 16 bits exactly as pp_groupnorm_apply_acc stores them); only weights stream through LDS -- twenty 40 KB slabs
// (320 rows x 64 k): proj_in (5), then Q, K, V (5 each) -- through a three-stage LDS-DMA ring with counted vmcnt waits.
// The accumulator layout of the first GEMM (lane = row, four consecutive columns per quad) is the B-operand layout of
// the second one up to a fixed permutation of the contraction index, applied to the QKV weights when they are packed
// (engine.py `_kperm`).  LayerNorm1 is folded as everywhere else: rstd * (hs . (gamma (.) W)^T - mean * colsum) + W beta,
// with the row moments taken from the 16-bit values of hs as stored (what the chain's producer epilogue emits).
// V leaves transposed ([batch][320][hw], what the attention kernels read) through an LDS tile after the last slab.
#include <type_traits>
#include <utility>

#include "pp_common.h"
#include "rowtile_io.h"

namespace {

constexpr int TF_C = 320, TF_BM = 128;
constexpr int TF_SLAB = 320 * 128, TF_NS = 3, TF_NSLAB = 20;
constexpr int TF_TAB = TF_NS * TF_SLAB;                   // fp32 tables: GN scale[320] | shift[320] | colsum[960] | bias[960]
constexpr int TF_T_SC = 0, TF_T_SH = 320, TF_T_CS = 640, TF_T_B2 = 1600, TF_T_ST = 2560;   // (+ 64 group stats)
constexpr int TF_STG = TF_TAB + (2560 + 64) * 4;          // the eight waves' private output staging tiles (rowtile_io.h)
constexpr int TF_LDS = TF_STG + 8 * RT_TILE;
constexpr int TF_QD = 8;

typedef __attribute__((address_space(3))) void* tf_lds_ptr_t;

struct TFArgs {
  const uint16_t* x; int ldx;
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; float gn_eps; int gn_groups;
  const uint16_t* w1; const float* b1;
  const uint16_t* w2p; const float* cs2; const float* b2; float ln_eps;
  uint16_t* hs; int ldhs;
  uint16_t* qk; int ldqk;
  uint16_t* vt; int ldvt;
  int M, rows_per_batch;
  float q_scale;
};

template <int... I, class F>
PP_DEVINL void tf_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void tf_static_for(F&& f) { tf_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// DSP: the five DMA pieces of slab t + 2 spread over the 40 MFMA slots of step t (one every eight) instead of a burst behind
// the barrier (gemm.hip's DMAI: 8 waves x 5 pieces queue on the CU's one address unit before anybody's first MFMA)
// DBG (lab timing probes, results garbage): 1 no hs / q / k stores, 2 no MFMAs, 4 no V^T epilogue
// SM (lab): how the hs / q / k rows leave -- 0 through the wave-private LDS tile (ships), 1 register-direct: neighbouring
// 16-lane rows re-paired with v_permlane16_swap so that a lane owns 8 consecutive columns (16-byte stores, 64-byte pieces)
template <int EDT, int QD = TF_QD, bool DSP = false, int DBG = 0, int SM = 0>
__global__ void __launch_bounds__(512, 2) tfront_kernel(const TFArgs a) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabs = reinterpret_cast<float*>(smem + TF_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m_blk = lid * TF_BM;
  const int b = m_blk / a.rows_per_batch;
  const int m = m_blk + wave * 16 + r16;                 // this lane's row: MFMA B column / accumulator column

  // ---- slab loader (the scheme of xattn_block_kernel): 40 strips of 8 rows x 128 B per slab, five per wave
  const int lrow = lane >> 3, kslot = (lane & 7) ^ lrow;
  int vw[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) vw[j] = ((8 * (wave + 8 * j) + lrow) * TF_C + kslot * 8) * 2;
  const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(a.w1, TF_C * TF_C * 2);
  const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(a.w2p, 3 * TF_C * TF_C * 2);
  auto issue = [&](auto T) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    char* st = smem + (t % TF_NS) * TF_SLAB + wave * 1024;
    if constexpr (t < 5) {                                // proj_in rows 0 .. 319, input channels 64 t ..
      constexpr int so = t * 64 * 2;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (tf_lds_ptr_t)(st + j * 8192), 16, vw[j], so, 0, 0);
    } else {                                              // pass p = Q / K / V: rows 320 p .., (permuted) k 64 kt ..
      constexpr int p = (t - 5) / 5, kt = (t - 5) % 5;
      constexpr int so = (p * 320 * TF_C + kt * 64) * 2;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (tf_lds_ptr_t)(st + j * 8192), 16, vw[j], so, 0, 0);
    }
  };
  auto issue_piece = [&](auto T, int j) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    char* st = smem + (t % TF_NS) * TF_SLAB + wave * 1024;
    if constexpr (t < 5) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (tf_lds_ptr_t)(st + j * 8192), 16, vw[j], t * 64 * 2, 0, 0);
    } else {
      constexpr int p = (t - 5) / 5, kt = (t - 5) % 5;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (tf_lds_ptr_t)(st + j * 8192), 16, vw[j], (p * 320 * TF_C + kt * 64) * 2, 0, 0);
    }
  };
  // ---- the wave's 16 raw input rows (k-group g: eight consecutive channels) first, then the first two slabs: both in
  //      flight during the table set-up (the x loads are the OLDER ones, so waiting for them leaves the slabs in flight)
  u32x4_t xr[10];
  {
    const uint16_t* xp = a.x + (size_t)m * a.ldx + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) xr[s] = *reinterpret_cast<const u32x4_t*>(xp + 32 * s);
  }
  __builtin_amdgcn_sched_barrier(0);
  issue(std::integral_constant<int, 0>{});
  issue(std::integral_constant<int, 1>{});
  // ---- tables: GroupNorm (scale, shift) of the batch item (the arithmetic of gn_fold_acc, norm.hip), LN colsum / bias
  if (tid < a.gn_groups) {
    const long long* ap = a.gn_acc + ((size_t)b * a.gn_groups + tid) * 2;
    const double s = (double)ap[0] * (1.0 / (double)PP_GN_SUM_SCALE);
    const double q = (double)ap[1] * (1.0 / (double)PP_GN_SQ_SCALE);
    const double n = (double)a.rows_per_batch * (double)(TF_C / a.gn_groups);
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    tabs[TF_T_ST + tid] = (float)mean;
    tabs[TF_T_ST + 32 + tid] = (float)(1.0 / sqrt(var + (double)a.gn_eps));
  }
  for (int i = tid; i < 960; i += 512) {
    tabs[TF_T_CS + i] = a.cs2[i];
    tabs[TF_T_B2 + i] = a.b2[i];
  }
  __syncthreads();
  if (tid < TF_C) {
    const int gg = tid / (TF_C / a.gn_groups);
    const float sc = tabs[TF_T_ST + 32 + gg] * a.gn_gamma[tid];
    tabs[TF_T_SC + tid] = sc;
    tabs[TF_T_SH + tid] = a.gn_beta[tid] - tabs[TF_T_ST + gg] * sc;
  }
  __syncthreads();
  // ---- normalise the fragments in registers: x * scale[c] + shift[c], rounded to 16 bits (= what the apply launch stored)
  v8_t xf[10];
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int c = 32 * s + 8 * g;
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SC + c), a1 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SC + c + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SH + c), b1 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SH + c + 4);
    const u32x4_t v = xr[s];
    u32x4_t o;
    o[0] = E::pack2(E::lo(v[0]) * a0[0] + b0[0], E::hi(v[0]) * a0[1] + b0[1]);
    o[1] = E::pack2(E::lo(v[1]) * a0[2] + b0[2], E::hi(v[1]) * a0[3] + b0[3]);
    o[2] = E::pack2(E::lo(v[2]) * a1[0] + b1[0], E::hi(v[2]) * a1[1] + b1[1]);
    o[3] = E::pack2(E::lo(v[3]) * a1[2] + b1[2], E::hi(v[3]) * a1[3] + b1[3]);
    xf[s] = __builtin_bit_cast(v8_t, o);
  }

  f32x4_t acc[20];
  uint32_t pf[10][4];                                     // hs as B fragments of the second GEMM
  float mean = 0.f, rstd = 1.f;
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Output rows leave through the wave's private LDS tile, 64 columns at a time, as whole 128-byte lines (rowtile_io.h: as
  // 8-byte pieces scattered over 16 rows the hs / q / k stores were 21 of the kernel's 49 us, lab PP_TF_DBG=1)
  char* const stg = smem + TF_STG + wave * RT_TILE;
  auto stage_put = [&](int j, uint32_t o0, uint32_t o1) __attribute__((always_inline)) { rt_put(stg, r16, g, j, o0, o1); };
  auto stage_flush = [&](uint16_t* dst, int ld, int col0) __attribute__((always_inline)) {     // dst: row 0 of the wave's tile
    if constexpr (!(DBG & 1)) rt_flush(stg, lane, dst, ld, col0);
  };
  // SM 1: quads of n-blocks nb (x) and nb + 1 (y) of this lane -> 16 bytes of ONE of them: even 16-lane rows end up with n-block
  // nb, columns 4 g .. 4 g + 7, odd rows with n-block nb + 1, columns 4 (g - 1) .. 4 (g - 1) + 7
  auto pair_store = [&](uint16_t* rowp, int nb, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) __attribute__((always_inline)) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x0), "+v"(y0));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x1), "+v"(y1));
    const int col = (g & 1) ? 16 * (nb + 1) + 4 * (g - 1) : 16 * nb + 4 * g;
    if constexpr (!(DBG & 1)) *reinterpret_cast<u32x4_t*>(rowp + col) = u32x4_t{x0, x1, y0, y1};
  };
  uint16_t* const hs0 = a.hs + (size_t)(m_blk + wave * 16) * a.ldhs;
  uint16_t* const qk0 = a.qk + (size_t)(m_blk + wave * 16) * a.ldqk;
  // after the proj_in slabs: + bias, 16-bit store of hs, its row moments (LayerNorm1), and the fragments of GEMM 2
  auto finish_proj_in = [&]() __attribute__((always_inline)) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int nb = 0; nb < 20; ++nb) {
      const int n = nb * 16 + 4 * g;
      const f32x4_t v = acc[nb] + *reinterpret_cast<const f32x4_t*>(a.b1 + n);
      const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
      if constexpr (SM == 0) {
        stage_put(nb & 3, o0, o1);
        if ((nb & 3) == 3) stage_flush(hs0, a.ldhs, (nb >> 2) * 64);
      } else {
        if (nb & 1) pair_store(a.hs + (size_t)m * a.ldhs, nb - 1, pf[(nb - 1) >> 1][((nb - 1) & 1) * 2], pf[(nb - 1) >> 1][((nb - 1) & 1) * 2 + 1], o0, o1);
      }
      const float r0 = E::lo(o0), r1 = E::hi(o0), r2 = E::lo(o1), r3 = E::hi(o1);
      sm += (r0 + r1) + (r2 + r3);
      sq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
      pf[nb >> 1][(nb & 1) * 2 + 0] = o0;
      pf[nb >> 1][(nb & 1) * 2 + 1] = o1;
      acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
    sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
    mean = sm * (1.0f / TF_C);
    rstd = rsqrtf(fmaxf(sq * (1.0f / TF_C) - mean * mean, 0.f) + a.ln_eps);
  };
  // after the five slabs of pass p: the folded-LayerNorm correction; Q / K rows out, V kept for the transposed store
  uint32_t vkeep[20][2];
  auto finish_pass = [&](auto P) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value;
#pragma unroll
    for (int nb = 0; nb < 20; ++nb) {
      const int n = p * TF_C + nb * 16 + 4 * g;
      const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_CS + n);
      const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_B2 + n);
      f32x4_t v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = rstd * (acc[nb][i] - mean * cs[i]) + bb[i];
      if constexpr (p == 0) v *= a.q_scale;      // (1.0, or head_dim^-0.5 * log2 e for PP_ATTN_PIPE_LOG2)
      const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
      if constexpr (p < 2 && SM == 0) {
        stage_put(nb & 3, o0, o1);
        if ((nb & 3) == 3) stage_flush(qk0, a.ldqk, p * TF_C + (nb >> 2) * 64);
      } else if constexpr (p < 2) {
        vkeep[nb][0] = o0; vkeep[nb][1] = o1;         // (scratch until the partner n-block is ready: V overwrites it in pass 2)
        if (nb & 1) pair_store(a.qk + (size_t)m * a.ldqk + p * TF_C, nb - 1, vkeep[nb - 1][0], vkeep[nb - 1][1], o0, o1);
      } else { vkeep[nb][0] = o0; vkeep[nb][1] = o1; }
      acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  };

  tf_static_for<TF_NSLAB>([&](auto T) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    if constexpr (t + 1 < TF_NSLAB) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if constexpr (!DSP && t + 2 < TF_NSLAB) issue(std::integral_constant<int, t + 2>{});   // (its stage was read at step t - 1)
    const char* st = smem + (t % TF_NS) * TF_SLAB;
    v8_t q[QD + 1];
    const int so0 = ((0 * 4 + g) ^ (r16 & 7)) << 4, so1 = ((1 * 4 + g) ^ (r16 & 7)) << 4;
    auto load_frag = [&](int i) __attribute__((always_inline)) -> v8_t {
      return *reinterpret_cast<const v8_t*>(st + ((i % 20) * 16 + r16) * 128 + (i < 20 ? so0 : so1));
    };
    v8_t bfr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (t < 5) bfr[ks] = xf[t * 2 + ks];
      else {
        constexpr int f = ((t - 5) % 5) * 2;
        bfr[ks] = __builtin_bit_cast(v8_t, u32x4_t{pf[f + ks][0], pf[f + ks][1], pf[f + ks][2], pf[f + ks][3]});
      }
    }
#pragma unroll
    for (int i = 0; i < QD; ++i) q[i] = load_frag(i);
#pragma unroll
    for (int i = 0; i < 40; ++i) {
      if (i + QD < 40) q[(i + QD) % (QD + 1)] = load_frag(i + QD);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG & 2) acc[i % 20] += __builtin_bit_cast(f32x4_t, q[i % (QD + 1)]);
      else acc[i % 20] = E::mfma16(q[i % (QD + 1)], bfr[i / 20], acc[i % 20]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DSP && t + 2 < TF_NSLAB) {
        if (i % 8 == 2) {
          issue_piece(std::integral_constant<int, (t + 2 < TF_NSLAB ? t + 2 : 0)>{}, i / 8);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (t == 4) finish_proj_in();
    if constexpr (t == 9) finish_pass(std::integral_constant<int, 0>{});
    if constexpr (t == 14) finish_pass(std::integral_constant<int, 1>{});
    if constexpr (t == 19) finish_pass(std::integral_constant<int, 2>{});
  });

  // ---- V^T: [channel n][row of the tile] 16-bit words through LDS (the slabs are done), then 256-byte row segments
  if constexpr (DBG & 4) { if (vkeep[3][1] == 0x12345u) a.vt[m] = 1; return; }
  __syncthreads();
  uint16_t* vtile = reinterpret_cast<uint16_t*>(smem);   // [320][128 + 8]: 87 KB of the 120 KB slab area
  constexpr int VLD = TF_BM + 8;
  {
    const int ml = wave * 16 + r16;
#pragma unroll
    for (int nb = 0; nb < 20; ++nb) {
      const int n = nb * 16 + 4 * g;
      vtile[(n + 0) * VLD + ml] = (uint16_t)(vkeep[nb][0] & 0xffffu);
      vtile[(n + 1) * VLD + ml] = (uint16_t)(vkeep[nb][0] >> 16);
      vtile[(n + 2) * VLD + ml] = (uint16_t)(vkeep[nb][1] & 0xffffu);
      vtile[(n + 3) * VLD + ml] = (uint16_t)(vkeep[nb][1] >> 16);
    }
  }
  __syncthreads();
  {
    const int p0 = m_blk - b * a.rows_per_batch;          // first pixel of the tile inside its batch item
    uint16_t* vb = a.vt + (size_t)b * TF_C * a.ldvt + p0;
    for (int i = tid; i < TF_C * (TF_BM / 8); i += 512) { // 16 pieces of 16 bytes per channel row
      const int n = i >> 4, pc = i & 15;
      *reinterpret_cast<u32x4_t*>(vb + (size_t)n * a.ldvt + pc * 8) = *reinterpret_cast<const u32x4_t*>(vtile + n * VLD + pc * 8);
    }
  }
}


#ifdef PP_LAB
// ---- the same launch with FOUR waves of 32 rows (one per SIMD, the whole register file of its SIMD): every weight
// fragment read from LDS feeds TWO matrix instructions (both 16-row blocks of the wave).  The 8-wave kernel above reads one
// fragment per MFMA: 8 waves x 40 ds_read_b128 x 4 cycles = 1280 LDS cycles per slab beside 1280 matrix-pipe cycles per
// SIMD -- both units exactly saturated on paper, ~4200 cycles per slab measured.  Here: 640 LDS cycles per slab, the same
// 80 MFMAs per SIMD issued by ONE in-order wave (no partner to hide its DMA issue: ten pieces per wave and slab).
template <int EDT, int QD = TF_QD>
__global__ void __launch_bounds__(256, 1) tfront4_kernel(const TFArgs a) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabs = reinterpret_cast<float*>(smem + TF_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m_blk = lid * TF_BM;
  const int b = m_blk / a.rows_per_batch;
  const int m0 = m_blk + wave * 32 + r16;                // this lane's rows: m0 and m0 + 16

  // ---- slab loader: 40 strips of 8 rows x 128 B per slab, ten per wave (strip wave + 4 j)
  const int lrow = lane >> 3, kslot = (lane & 7) ^ lrow;
  int vw[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) vw[j] = ((8 * (wave + 4 * j) + lrow) * TF_C + kslot * 8) * 2;
  const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(a.w1, TF_C * TF_C * 2);
  const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(a.w2p, 3 * TF_C * TF_C * 2);
  auto issue = [&](auto T) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    char* st = smem + (t % TF_NS) * TF_SLAB + wave * 1024;
    if constexpr (t < 5) {
      constexpr int so = t * 64 * 2;
#pragma unroll
      for (int j = 0; j < 10; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1, (tf_lds_ptr_t)(st + j * 4096), 16, vw[j], so, 0, 0);
    } else {
      constexpr int p = (t - 5) / 5, kt = (t - 5) % 5;
      constexpr int so = (p * 320 * TF_C + kt * 64) * 2;
#pragma unroll
      for (int j = 0; j < 10; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_2, (tf_lds_ptr_t)(st + j * 4096), 16, vw[j], so, 0, 0);
    }
  };
  u32x4_t xr[2][10];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const uint16_t* xp = a.x + (size_t)(m0 + 16 * mi) * a.ldx + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) xr[mi][s] = *reinterpret_cast<const u32x4_t*>(xp + 32 * s);
  }
  __builtin_amdgcn_sched_barrier(0);
  issue(std::integral_constant<int, 0>{});
  issue(std::integral_constant<int, 1>{});
  if (tid < a.gn_groups) {
    const long long* ap = a.gn_acc + ((size_t)b * a.gn_groups + tid) * 2;
    const double s = (double)ap[0] * (1.0 / (double)PP_GN_SUM_SCALE);
    const double q = (double)ap[1] * (1.0 / (double)PP_GN_SQ_SCALE);
    const double n = (double)a.rows_per_batch * (double)(TF_C / a.gn_groups);
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    tabs[TF_T_ST + tid] = (float)mean;
    tabs[TF_T_ST + 32 + tid] = (float)(1.0 / sqrt(var + (double)a.gn_eps));
  }
  for (int i = tid; i < 960; i += 256) {
    tabs[TF_T_CS + i] = a.cs2[i];
    tabs[TF_T_B2 + i] = a.b2[i];
  }
  __syncthreads();
  for (int c = tid; c < TF_C; c += 256) {
    const int gg = c / (TF_C / a.gn_groups);
    const float sc = tabs[TF_T_ST + 32 + gg] * a.gn_gamma[c];
    tabs[TF_T_SC + c] = sc;
    tabs[TF_T_SH + c] = a.gn_beta[c] - tabs[TF_T_ST + gg] * sc;
  }
  __syncthreads();
  v8_t xf[2][10];
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int c = 32 * s + 8 * g;
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SC + c), a1 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SC + c + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SH + c), b1 = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_SH + c + 4);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const u32x4_t v = xr[mi][s];
      u32x4_t o;
      o[0] = E::pack2(E::lo(v[0]) * a0[0] + b0[0], E::hi(v[0]) * a0[1] + b0[1]);
      o[1] = E::pack2(E::lo(v[1]) * a0[2] + b0[2], E::hi(v[1]) * a0[3] + b0[3]);
      o[2] = E::pack2(E::lo(v[2]) * a1[0] + b1[0], E::hi(v[2]) * a1[1] + b1[1]);
      o[3] = E::pack2(E::lo(v[3]) * a1[2] + b1[2], E::hi(v[3]) * a1[3] + b1[3]);
      xf[mi][s] = __builtin_bit_cast(v8_t, o);
    }
  }

  f32x4_t acc[20][2];
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) acc[nb][0] = acc[nb][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // after the proj_in slabs: + bias, 16-bit store of hs, its row moments; the fragments of GEMM 2 take xf's registers
  auto finish_proj_in = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      uint16_t* hrow = a.hs + (size_t)(m0 + 16 * mi) * a.ldhs;
      float sm = 0.f, sq = 0.f;
      uint32_t pf[10][4];
#pragma unroll
      for (int nb = 0; nb < 20; ++nb) {
        const int n = nb * 16 + 4 * g;
        const f32x4_t v = acc[nb][mi] + *reinterpret_cast<const f32x4_t*>(a.b1 + n);
        const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(hrow + n) = u32x2_t{o0, o1};
        const float r0 = E::lo(o0), r1 = E::hi(o0), r2 = E::lo(o1), r3 = E::hi(o1);
        sm += (r0 + r1) + (r2 + r3);
        sq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        pf[nb >> 1][(nb & 1) * 2 + 0] = o0;
        pf[nb >> 1][(nb & 1) * 2 + 1] = o1;
        acc[nb][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < 10; ++s) xf[mi][s] = __builtin_bit_cast(v8_t, u32x4_t{pf[s][0], pf[s][1], pf[s][2], pf[s][3]});
      sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
      sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
      mean[mi] = sm * (1.0f / TF_C);
      rstd[mi] = rsqrtf(fmaxf(sq * (1.0f / TF_C) - mean[mi] * mean[mi], 0.f) + a.ln_eps);
    }
  };
  uint32_t vkeep[2][20][2];
  auto finish_pass = [&](auto P) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      uint16_t* orow = a.qk + (size_t)(m0 + 16 * mi) * a.ldqk + p * TF_C;
#pragma unroll
      for (int nb = 0; nb < 20; ++nb) {
        const int n = p * TF_C + nb * 16 + 4 * g;
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_CS + n);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(tabs + TF_T_B2 + n);
        f32x4_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = rstd[mi] * (acc[nb][mi][i] - mean[mi] * cs[i]) + bb[i];
        if constexpr (p == 0) v *= a.q_scale;
        const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
        if constexpr (p < 2) *reinterpret_cast<u32x2_t*>(orow + nb * 16 + 4 * g) = u32x2_t{o0, o1};
        else { vkeep[mi][nb][0] = o0; vkeep[mi][nb][1] = o1; }
        acc[nb][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  tf_static_for<TF_NSLAB>([&](auto T) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    if constexpr (t + 1 < TF_NSLAB) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if constexpr (t + 2 < TF_NSLAB) issue(std::integral_constant<int, t + 2>{});
    const char* st = smem + (t % TF_NS) * TF_SLAB;
    v8_t q[QD + 1];
    const int so0 = ((0 * 4 + g) ^ (r16 & 7)) << 4, so1 = ((1 * 4 + g) ^ (r16 & 7)) << 4;
    auto load_frag = [&](int i) __attribute__((always_inline)) -> v8_t {
      return *reinterpret_cast<const v8_t*>(st + ((i % 20) * 16 + r16) * 128 + (i < 20 ? so0 : so1));
    };
    constexpr int f0 = (t < 5 ? t : (t - 5) % 5) * 2;     // k block pair of this slab (GEMM 1: of x, GEMM 2: of hs)
#pragma unroll
    for (int i = 0; i < QD; ++i) q[i] = load_frag(i);
#pragma unroll
    for (int i = 0; i < 40; ++i) {
      if (i + QD < 40) q[(i + QD) % (QD + 1)] = load_frag(i + QD);
      __builtin_amdgcn_sched_barrier(0);
      acc[i % 20][0] = E::mfma16(q[i % (QD + 1)], xf[0][f0 + i / 20], acc[i % 20][0]);
      acc[i % 20][1] = E::mfma16(q[i % (QD + 1)], xf[1][f0 + i / 20], acc[i % 20][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (t == 4) finish_proj_in();
    if constexpr (t == 9) finish_pass(std::integral_constant<int, 0>{});
    if constexpr (t == 14) finish_pass(std::integral_constant<int, 1>{});
    if constexpr (t == 19) finish_pass(std::integral_constant<int, 2>{});
  });

  __syncthreads();
  uint16_t* vtile = reinterpret_cast<uint16_t*>(smem);   // [320][128 + 8]
  constexpr int VLD = TF_BM + 8;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int ml = wave * 32 + mi * 16 + r16;
#pragma unroll
    for (int nb = 0; nb < 20; ++nb) {
      const int n = nb * 16 + 4 * g;
      vtile[(n + 0) * VLD + ml] = (uint16_t)(vkeep[mi][nb][0] & 0xffffu);
      vtile[(n + 1) * VLD + ml] = (uint16_t)(vkeep[mi][nb][0] >> 16);
      vtile[(n + 2) * VLD + ml] = (uint16_t)(vkeep[mi][nb][1] & 0xffffu);
      vtile[(n + 3) * VLD + ml] = (uint16_t)(vkeep[mi][nb][1] >> 16);
    }
  }
  __syncthreads();
  {
    const int p0 = m_blk - b * a.rows_per_batch;
    uint16_t* vb = a.vt + (size_t)b * TF_C * a.ldvt + p0;
    for (int i = tid; i < TF_C * (TF_BM / 8); i += 256) {
      const int n = i >> 4, pc = i & 15;
      *reinterpret_cast<u32x4_t*>(vb + (size_t)n * a.ldvt + pc * 8) = *reinterpret_cast<const u32x4_t*>(vtile + n * VLD + pc * 8);
    }
  }
}
#endif   // PP_LAB

}  // namespace

extern "C" int pp_tfront_supported(int M, int c, int rows_per_batch, int gn_groups) {
  return (c == TF_C && M > 0 && M % TF_BM == 0 && rows_per_batch > 0 && rows_per_batch % TF_BM == 0 && M % rows_per_batch == 0 &&
          gn_groups > 0 && gn_groups <= 32 && TF_C % gn_groups == 0) ? 1 : 0;
}

extern "C" int pp_tfront(const void* x, int ldx, const void* gn_acc, const float* gn_gamma, const float* gn_beta, float gn_eps,
                         int gn_groups, const void* w1, const float* b1, const void* w2p, const float* cs2, const float* b2,
                         float ln_eps, void* hs, int ldhs, void* qk, int ldqk, void* vt, int ldvt, int M, int c,
                         int rows_per_batch, float q_scale, int dtype, void* stream) {
  if (!x || !gn_acc || !gn_gamma || !gn_beta || !w1 || !b1 || !w2p || !cs2 || !b2 || !hs || !qk || !vt || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  if (!pp_tfront_supported(M, c, rows_per_batch, gn_groups)) return PP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(hs) | reinterpret_cast<uintptr_t>(qk)) & 15) return PP_ERR_BAD_ARG;   // 16-byte row stores
  if (!(q_scale > 0.f)) return PP_ERR_BAD_ARG;
  if (ldx < c || (ldx & 7) || ldhs < c || (ldhs & 7) || ldqk < 2 * c || (ldqk & 7) || ldvt < rows_per_batch || (ldvt & 7))
    return PP_ERR_BAD_ARG;
  TFArgs a;
  a.x = (const uint16_t*)x; a.ldx = ldx;
  a.gn_acc = (const long long*)gn_acc; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.gn_eps = gn_eps; a.gn_groups = gn_groups;
  a.w1 = (const uint16_t*)w1; a.b1 = b1;
  a.w2p = (const uint16_t*)w2p; a.cs2 = cs2; a.b2 = b2; a.ln_eps = ln_eps;
  a.hs = (uint16_t*)hs; a.ldhs = ldhs;
  a.qk = (uint16_t*)qk; a.ldqk = ldqk;
  a.vt = (uint16_t*)vt; a.ldvt = ldvt;
  a.M = M; a.rows_per_batch = rows_per_batch;
  a.q_scale = q_scale;
  auto go = [&](auto kern) -> int {
    if (pp_func_lds(reinterpret_cast<const void*>(kern), TF_LDS, "hipFuncSetAttribute(tfront)") != PP_OK) return PP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(M / TF_BM), dim3(512), TF_LDS, (hipStream_t)stream, a);
    PP_CHECK_LAUNCH("tfront_kernel");
    return PP_OK;
  };
#ifdef PP_LAB
  auto go4 = [&](auto kern) -> int {
    if (pp_func_lds(reinterpret_cast<const void*>(kern), TF_LDS, "hipFuncSetAttribute(tfront4)") != PP_OK) return PP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(M / TF_BM), dim3(256), TF_LDS, (hipStream_t)stream, a);
    PP_CHECK_LAUNCH("tfront4_kernel");
    return PP_OK;
  };
  // (lab) PP_TF_W4=1: four waves of 32 rows (every weight fragment feeds two MFMAs) instead of eight of 16: 53.5 against
  // 49.2 us per launch, a draw on the step (profiles/r06_rejected_experiments.txt)
  if (pp_lab_env("PP_TF_W4", 0)) {
    if (dtype == PP_DT_F16) return go4(tfront4_kernel<PP_DT_F16>);
    return go4(tfront4_kernel<PP_DT_BF16>);
  }
  // (lab) PP_TF_DSP=1: slab DMA pieces spread over the MFMA slots
  if (pp_lab_env("PP_TF_DSP", 0)) {
    if (dtype == PP_DT_F16) return go(tfront_kernel<PP_DT_F16, TF_QD, true>);
    return go(tfront_kernel<PP_DT_BF16, TF_QD, true>);
  }
  if (dtype == PP_DT_BF16) switch (pp_lab_env("PP_TF_DBG", 0)) {
      case 1: return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 1>);
      case 2: return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 2>);
      case 4: return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 4>);
      case 5: return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 5>);
      case 7: return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 7>);
      default: break;
    }
  if (dtype == PP_DT_BF16 && pp_lab_env("PP_TF_SM", 0) == 1) return go(tfront_kernel<PP_DT_BF16, TF_QD, false, 0, 1>);
  if (dtype == PP_DT_BF16) switch (pp_lab_env("PP_TF_QD", TF_QD)) {      // fragment reads in flight ahead of their MFMA
      case 4: return go(tfront_kernel<PP_DT_BF16, 4>);
      case 12: return go(tfront_kernel<PP_DT_BF16, 12>);
      case 16: return go(tfront_kernel<PP_DT_BF16, 16>);
      default: break;
    }
#endif
  if (dtype == PP_DT_F16) return go(tfront_kernel<PP_DT_F16>);
  return go(tfront_kernel<PP_DT_BF16>);
}
