// Fused cross-attention sub-block of BasicTransformerBlock at C = 320 (the 64x64 level of SD-1.5; 128x128 on config 5):
//
//     h_out = h + to_out( softmax( to_q(LN2(h)) K^T / sqrt(d) ) V )                 one launch instead of three
//
// (reference ctor site /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300 -> diffusers 0.27
// BasicTransformerBlock.attn2 + norm2; the unfused plan is engine.py `_transformer`: `linear` (to_q, LayerNorm folded)
// -> `attention` (77 keys) -> `linear` (to_out + residual + row moments), 20 + 22 + 20 us and 4 x 21 MB of q / o
// round trips per block at 64x64; this kernel: 55 us, profiles/r03_xattn_fused_ab.txt).
//
// The encoder hidden states are step-invariant, so K and V can be folded into the two projections ONCE per prompt
// (pp_xattn_fold, part of the setup plan):
//     logits_h = LN2(h) . G_h ,   G_h = scale * Wq_h^T K_h^T      [C][80 keys]  per (batch item, head)
//     out      = sum_h P_h . H_h ,  H_h = V_h Wo_h^T               [80 keys][C]
// which turns the sub-block into two chained GEMMs with a per-head softmax in between -- N = 8 heads x 80 keys = 640
// logit columns, K = C; then K = 640, N = C -- whose intermediate (the probabilities) never leaves the registers:
// the MFMA accumulator layout of the first GEMM (lane = row, four consecutive columns per register quad) IS the
// B-operand layout of the second one, up to a fixed permutation of the contraction index that is applied to H when it is
// packed.  Twice the FLOPs of the unfused chain (13.4 + 13.4 instead of 6.7 + 3.3 + 3.3 + 6.7 GFLOP at 64x64 x 8),
// none of its HBM round trips.
//
// One workgroup = 128 rows of one batch item, eight waves of 16 rows each; a wave owns ALL columns of its rows (softmax
// and the second contraction need no exchange).  The wave's 16 x 320 input rows are MFMA B fragments loaded straight
// from global memory; the only thing that streams through LDS is "weights": twenty 40 KB slabs (320 rows x 64 k) --
// G^T heads 0-3 (5 slabs), G^T heads 4-7 (5), H^T (10) -- through a three-stage LDS-DMA ring with counted vmcnt waits.
#include <type_traits>
#include <utility>

#include "pp_common.h"
#include "rowtile_io.h"

namespace {

constexpr int XA_C = 320, XA_HEADS = 8, XA_KP = 80, XA_S = XA_HEADS * XA_KP;   // 640 logit columns
constexpr int XA_BM = 128;
constexpr int XA_SLAB = 320 * 128, XA_NS = 3, XA_NSLAB = 20;
constexpr int XA_TAB = XA_NS * XA_SLAB;                   // (logit colsum | logit bias) of the batch item, fp32 [2][640]
constexpr int XA_STG = XA_TAB + 2 * XA_S * 4 + XA_C * 4;   // (+ the bias of the Linear in front, PRE); then the eight waves'
constexpr int XA_LDS = XA_STG + 8 * RT_TILE;               // private output / residual staging tiles (rowtile_io.h)
constexpr int XA_QD = 8;                                  // fragment reads kept in flight ahead of their MFMA

typedef __attribute__((address_space(3))) void* xa_lds_ptr_t;

struct XAArgs {
  const uint16_t* x; int ldx;
  const uint16_t* res; int ldres;
  const float* ln_stats; int ln_tiles; float ln_eps;
  const uint16_t* gt; const float* gcs; const float* gbias; const uint16_t* ht;
  const float* bias_o;
  uint16_t* out; int ldo;
  float* row_stats_out;
  int M, rows_per_batch;
  int src_wrap;          // > 0: x / res / ln_stats hold src_wrap rows, row m reads row m mod src_wrap (CFG twin, pp_hip.h)
  const uint16_t* pre_w; // != NULL (C = 320): a Linear in FRONT of the sub-block, h = x pre_w^T + pre_b + res (attn1.to_out)
  const float* pre_b;
};

template <int... I, class F>
PP_DEVINL void xa_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
PP_DEVINL void xa_static_for(F&& f) { xa_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- once per prompt: G^T [B][640][C] (16-bit), logit colsum / bias [B][640] (fp32), H^T [B][C][640] (16-bit, k-permuted)
// Round 5: three kernels.  The first form (one thread per output pair, a D-deep loop of one 4-byte weight load per two
// FMAs) pulled every weight row through the L2 once per key -- 330 us per launch on average, 600 at C = 1280, sixteen
// launches per pipeline call = 5 ms per call (0.1 ms per denoise step at 50 steps: what the two-GEMM form of the C = 1280
// sub-blocks gains per step).  Now a workgroup owns one (batch item, head): the head's K (or V) tile sits in LDS as fp32
// [D][80 keys], a thread keeps ALL 80 keys of its two channels (G^T) / its output row (H^T) in registers and reads every
// weight exactly once -- same FMA order per output (j ascending), so the stored bits are those of the first form.
constexpr int XF_MAXD = 160;
template <int EDT>
__global__ void __launch_bounds__(256) xattn_fold_g_kernel(const uint16_t* __restrict__ k, int ldk, int nctx,
                                                          const uint16_t* __restrict__ wq, float qscale,
                                                          uint32_t* __restrict__ gt, int C, int kperm) {
  using E = E16<EDT>;
  __shared__ __attribute__((aligned(16))) float ks[XF_MAXD * XA_KP];          // [j][key]
  const int D = C / XA_HEADS, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < XA_KP * D; i += 256) {
    const int key = i / D, j = i - key * D;
    ks[j * XA_KP + key] = key < nctx ? E::to_f(k[((size_t)b * nctx + key) * ldk + h * D + j]) : 0.f;
  }
  __syncthreads();
  const int c2 = blockIdx.x * 256 + tid;
  if (c2 >= C / 2) return;
  // kperm & 1: storage position 8 kg + j of every group of 32 channels holds channel 16 (j >> 2) + 4 kg + (j & 3) (the pair
  // (j, j + 1), j even, stays a pair of neighbours) -- the logits' B operand then comes out of the accumulators of the
  // Linear in front (pre_w), as in tfront.hip
  int ch = 2 * c2;
  if (kperm & 1) {
    const int s32 = ch >> 5, kg = (ch >> 3) & 3, j = ch & 7;
    ch = 32 * s32 + 16 * (j >> 2) + 4 * kg + (j & 3);
  }
  const uint16_t* wr = wq + (size_t)(h * D) * C + ch;
  float a0[XA_KP], a1[XA_KP];
#pragma unroll
  for (int q = 0; q < XA_KP; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
  // (D = 40 / 80 / 160: whole eights -- eight independent weight loads in flight per thread)
#pragma unroll 1
  for (int j0 = 0; j0 < D; j0 += 8) {
    uint32_t w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w8[u] = *reinterpret_cast<const uint32_t*>(wr + (size_t)(j0 + u) * C);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float wl = E::lo(w8[u]), wh = E::hi(w8[u]);
      const f32x4_t* kr = reinterpret_cast<const f32x4_t*>(ks + (j0 + u) * XA_KP);
#pragma unroll
      for (int q4 = 0; q4 < XA_KP / 4; ++q4) {
        const f32x4_t kv = kr[q4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0[4 * q4 + e] = fmaf(kv[e], wl, a0[4 * q4 + e]);
          a1[4 * q4 + e] = fmaf(kv[e], wh, a1[4 * q4 + e]);
        }
      }
    }
  }
  uint32_t* dst = gt + ((size_t)b * XA_S + h * XA_KP) * (C / 2) + c2;
#pragma unroll
  for (int q = 0; q < XA_KP; ++q) dst[(size_t)q * (C / 2)] = E::pack2(a0[q] * qscale, a1[q] * qscale);
}

template <int EDT>
__global__ void __launch_bounds__(256) xattn_fold_h_kernel(const uint16_t* __restrict__ vt, int ldvt, int nctx,
                                                          const uint16_t* __restrict__ wo, uint32_t* __restrict__ ht, int C,
                                                          int kperm) {
  using E = E16<EDT>;
  __shared__ __attribute__((aligned(16))) float vs[XF_MAXD * XA_KP];          // [d][key]
  const int D = C / XA_HEADS, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < XA_KP * D; i += 256) {
    const int d = i / XA_KP, key = i - d * XA_KP;
    vs[i] = key < nctx ? E::to_f(vt[((size_t)b * C + h * D + d) * ldvt + key]) : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= C) return;
  const uint16_t* wr = wo + (size_t)n * C + h * D;
  float acc[XA_KP];
#pragma unroll
  for (int q = 0; q < XA_KP; ++q) acc[q] = 0.f;
  // (the thread's weight row segment in 16-byte pieces: n * C + h * D is a multiple of 8 elements)
#pragma unroll 1
  for (int d0 = 0; d0 < D; d0 += 8) {
    const u32x4_t w8 = *reinterpret_cast<const u32x4_t*>(wr + d0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = (u & 1) ? E::hi(w8[u >> 1]) : E::lo(w8[u >> 1]);
      const f32x4_t* vr = reinterpret_cast<const f32x4_t*>(vs + (d0 + u) * XA_KP);
#pragma unroll
      for (int q4 = 0; q4 < XA_KP / 4; ++q4) {
        const f32x4_t v = vr[q4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = fmaf(w, v[e], acc[4 * q4 + e]);
      }
    }
  }
  // storage position kp of a 32-group holds contraction index 32 s + 16 (j >> 2) + 4 kg + (j & 3), kp = 32 s + 8 kg + j:
  // a lane's 16-byte A fragment (k-group kg) then lines up with accumulator quads of logit blocks 2s and 2s + 1
  // (kperm & 2: natural order -- H^T as the weight matrix of a plain GEMM over the stored probabilities, ABI v20).
  // Contraction indices (kk, kk + 1), kk even, are neighbours in either order: one 32-bit store per pair.
  uint32_t* dst = ht + ((size_t)b * C + n) * (XA_S / 2);
#pragma unroll
  for (int q = 0; q < XA_KP; q += 2) {
    const int kk = h * XA_KP + q;
    int kp = kk;
    if (!(kperm & 2)) {
      const int r = kk & 31, j = 4 * (r >> 4) + (r & 3), kg = (r >> 2) & 3;
      kp = (kk & ~31) + 8 * kg + j;
    }
    dst[kp >> 1] = E::pack2(acc[q], acc[q + 1]);
  }
}

// folded-LayerNorm terms of the logits; -inf masks the padded keys
template <int EDT>
__global__ void __launch_bounds__(256) xattn_fold_vec_kernel(const uint16_t* __restrict__ k, int ldk, int batch, int nctx,
                                                            const float* __restrict__ qcs, const float* __restrict__ qb,
                                                            float qscale, float* __restrict__ gcs, float* __restrict__ gb,
                                                            int C) {
  using E = E16<EDT>;
  const int D = C / XA_HEADS;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= batch * XA_S) return;
  const int n = r % XA_S, b = r / XA_S;
  const int h = n / XA_KP, key = n % XA_KP;
  float s1 = 0.f;
  if (key < nctx) {
    const uint16_t* kr = k + ((size_t)b * nctx + key) * ldk + h * D;
    if (qb) {
#pragma unroll 8
      for (int j = 0; j < D; ++j) s1 = fmaf(E::to_f(kr[j]), qb[h * D + j], s1);
    }
    // (the mean term of the folded LayerNorm -- the column sum of G^T -- is taken over the entries AS STORED, by
    //  xattn_colsum_kernel behind this launch: ADVICE round 3)
  }
  if (!qcs) gcs[r] = 0.f;
  gb[r] = key < nctx ? s1 * qscale : -INFINITY;
}

// gcs[b][n] = sum over c of G^T[b][n][c] as stored (rounded to 16 bits): the mean term of the folded LayerNorm must cancel
// the mean component of x . G exactly, and the block kernels multiply with the stored G^T.  One wave per row.
template <int EDT>
__global__ void __launch_bounds__(256) xattn_colsum_kernel(const uint32_t* __restrict__ gt, float* __restrict__ gcs, long long rows,
                                                          int C) {
  using E = E16<EDT>;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const uint32_t* p = gt + row * (C / 2);
  float s = 0.f;
  for (int i = lane; i < C / 2; i += 64) {
    const uint32_t v = p[i];
    s += E::lo(v) + E::hi(v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) gcs[row] = s;
}

// DBG (lab build only): 1 no slab DMA, 2 no MFMAs, 4 no fragment reads, 8 no softmax, 16 no epilogue, 32 no barriers
// PRE: five more slabs in front -- h = x pre_w^T + pre_b + res (BasicTransformerBlock.attn1.to_out + residual, the launch
// that used to produce this kernel's input) on the same accumulators; h (rounded to 16 bits, as that launch stored it) is
// at once the logits' B operand (G^T packed with its channel index permuted: pp_xattn_fold(kperm = 1)), the source of the
// LayerNorm row moments and the residual of the epilogue.  25 slabs instead of 20, one launch and the h round trip less.
template <int EDT, int DBG = 0, int QD = XA_QD, bool PRE = false>
__global__ void __launch_bounds__(512, 2) xattn_block_kernel(const XAArgs a) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabs = reinterpret_cast<float*>(smem + XA_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // XCD-aware bijective remap (block b runs on XCD b % 8): consecutive tiles -- the rows of one batch item, which share
  // G and H -- land on one XCD's L2
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m_blk = lid * XA_BM;
  const int b = m_blk / a.rows_per_batch;
  const int m = m_blk + wave * 16 + r16;                 // this lane's row: MFMA B column / accumulator column
  const int ms = (a.src_wrap > 0 && m >= a.src_wrap) ? m - a.src_wrap : m;      // ... and the row its inputs come from

  // ---- the wave's 16 input rows as B fragments (k-group g: eight consecutive channels), straight from global memory
  v8_t xf[10];
  {
    const uint16_t* xr = a.x + (size_t)ms * a.ldx + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) xf[s] = *reinterpret_cast<const v8_t*>(xr + 32 * s);
  }
  for (int i = tid; i < 2 * XA_S; i += 512)
    tabs[i] = i < XA_S ? a.gcs[(size_t)b * XA_S + i] : a.gbias[(size_t)b * XA_S + i - XA_S];
  if (PRE && tid < XA_C) tabs[2 * XA_S + tid] = a.pre_b ? a.pre_b[tid] : 0.f;
  float mean = 0.f, rstd = 1.f;
  if (!PRE && a.ln_stats) {
    const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(a.ln_stats) + (size_t)ms * a.ln_tiles;
    float sm = 0.f, sq = 0.f;
    for (int t = 0; t < a.ln_tiles; ++t) { const f32x2_t v = pm[t]; sm += v[0]; sq += v[1]; }
    mean = sm * (1.0f / XA_C);
    rstd = rsqrtf(fmaxf(sq * (1.0f / XA_C) - mean * mean, 0.f) + a.ln_eps);
  }

  // ---- slab loader: 40 strips of 8 rows x 128 B per slab, five per wave; lane -> (row of the strip, k-slot it FETCHES so
  // that its lane-linear LDS slot is swizzled by the row)
  const int lrow = lane >> 3, kslot = (lane & 7) ^ lrow;
  int vg[5], vh[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int row = 8 * (wave + 8 * j) + lrow;
    vg[j] = (row * XA_C + kslot * 8) * 2;
    vh[j] = (row * XA_S + kslot * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.gt + (size_t)b * XA_S * XA_C, XA_S * XA_C * 2);
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.ht + (size_t)b * XA_C * XA_S, XA_C * XA_S * 2);
  const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(PRE ? a.pre_w : a.gt, PRE ? XA_C * XA_C * 2 : 0);
  constexpr int NPRE = PRE ? 5 : 0, NSLAB = XA_NSLAB + NPRE;
  auto issue = [&](auto T) __attribute__((always_inline)) {
    constexpr int tt = decltype(T)::value, t = tt - NPRE;      // tt: slab of this launch, t: slab of the sub-block proper
    if constexpr (DBG & 1) return;
    char* st = smem + (tt % XA_NS) * XA_SLAB + wave * 1024;
    if constexpr (t < 0) {                                // pre_w rows 0 .. 319, input channels 64 tt ..
      constexpr int so = tt * 64 * 2;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (xa_lds_ptr_t)(st + j * 8192), 16, vg[j], so, 0, 0);
    } else if constexpr (t < 10) {                        // G^T rows (t / 5) * 320 .., channels (t % 5) * 64 ..
      constexpr int so = ((t / 5) * 320 * XA_C + (t % 5) * 64) * 2;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (xa_lds_ptr_t)(st + j * 8192), 16, vg[j], so, 0, 0);
    } else {                                              // H^T rows 0 .. 319, logit columns (t - 10) * 64 ..
      constexpr int so = (t - 10) * 64 * 2;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (xa_lds_ptr_t)(st + j * 8192), 16, vh[j], so, 0, 0);
    }
  };

  f32x4_t acc[20];
  uint32_t pf[20][4];                                     // probabilities as B fragments of the second GEMM
  uint32_t hq[10][4];                                     // PRE: h as B fragments of the logits (and the epilogue's residual)
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // PRE, after the five pre_w slabs: + bias + residual, rounded to 16 bits; row moments of the rounded values (LayerNorm)
  // Residual rows arrive row-major, two 16-byte pieces per lane and 64-column chunk (whole 128-byte lines), and are turned
  // into the accumulator layout through the wave's private LDS tile when they are used; output rows leave the same way
  // (rowtile_io.h: as 8-byte pieces scattered over 16 rows the epilogue was 19 of this kernel's 56 us).
  u32x4_t rv4[10];
  f32x4_t bo[20];
  char* const stg = smem + XA_STG + wave * RT_TILE;
  const int m_w0 = m_blk + wave * 16;                    // first row of the wave's tile
  auto load_rows = [&](const uint16_t* base, int ld, bool wrap) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int q = lane + 64 * jj;
        int mr = m_w0 + (q >> 3);
        if (wrap && a.src_wrap > 0 && mr >= a.src_wrap) mr -= a.src_wrap;
        rv4[c * 2 + jj] = base ? *reinterpret_cast<const u32x4_t*>(base + (size_t)mr * ld + c * 64 + (q & 7) * 8) : u32x4_t{0u, 0u, 0u, 0u};
      }
  };
  // (its residual and bias: fetched one slab early, under the MFMAs of pre_w's fourth slab)
  auto prefetch_pre = [&]() __attribute__((always_inline)) { load_rows(a.res, a.ldres, true); };
  auto finish_pre = [&]() __attribute__((always_inline)) {
    float sm = 0.f, sq = 0.f;
    u32x2_t pk[4];
#pragma unroll
    for (int nb = 0; nb < 20; ++nb) {
      const int n = nb * 16 + 4 * g;
      if ((nb & 3) == 0) rt_fill(stg, lane, rv4[(nb >> 2) * 2], rv4[(nb >> 2) * 2 + 1]);
      const u32x2_t r = rt_get(stg, r16, g, nb & 3);
      const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(tabs + 2 * XA_S + n);
      const f32x4_t v = acc[nb] + bb + f32x4_t{E::lo(r[0]), E::hi(r[0]), E::lo(r[1]), E::hi(r[1])};
      const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
      const float r0 = E::lo(o0), r1 = E::hi(o0), r2 = E::lo(o1), r3 = E::hi(o1);
      sm += (r0 + r1) + (r2 + r3);
      sq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
      hq[nb >> 1][(nb & 1) * 2 + 0] = o0;
      hq[nb >> 1][(nb & 1) * 2 + 1] = o1;
      // h is also the residual of the epilogue, fifteen slabs from here: parked in the wave's own rows of `out` (40 more
      // live registers would spill) and read back by the same wave under the last slab
      acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      pk[nb & 3] = u32x2_t{o0, o1};
      if ((nb & 3) == 3) {                                // (the chunk's four residual reads are done: the tile takes the outputs)
#pragma unroll
        for (int j = 0; j < 4; ++j) rt_put(stg, r16, g, j, pk[j][0], pk[j][1]);
        rt_flush(stg, lane, a.out + (size_t)m_w0 * a.ldo, a.ldo, (nb >> 2) * 64);
      }
    }
    if (a.ln_tiles > 0) {                                  // (LayerNorm folded into the logits: ln_tiles > 0 says so)
      sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
      sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
      mean = sm * (1.0f / XA_C);
      rstd = rsqrtf(fmaxf(sq * (1.0f / XA_C) - mean * mean, 0.f) + a.ln_eps);
    }
  };

  // softmax of the four heads whose logits the accumulators hold (half hf of the 640 columns): head hh = blocks 5 hh ..
  // 5 hh + 4, a row's 80 keys spread over 20 registers x the four lanes that share r16
  auto softmax_half = [&](auto HF) __attribute__((always_inline)) {
    constexpr int hf = decltype(HF)::value;
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      float s[20];
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int n0 = hf * 320 + (5 * hh + q) * 16 + 4 * g;
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(tabs + n0);
        const f32x4_t gb = *reinterpret_cast<const f32x4_t*>(tabs + XA_S + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s[q * 4 + i] = rstd * (acc[5 * hh + q][i] - mean * cs[i]) + gb[i];
          mx = fmaxf(mx, s[q * 4 + i]);
        }
        acc[5 * hh + q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i] - mx);
        sum += s[i];
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int gbk = hf * 20 + 5 * hh + q;              // logit block 0 .. 39 -> fragment gbk / 2, register pair gbk & 1
        pf[gbk >> 1][(gbk & 1) * 2 + 0] = E::pack2(s[q * 4 + 0] * inv, s[q * 4 + 1] * inv);
        pf[gbk >> 1][(gbk & 1) * 2 + 1] = E::pack2(s[q * 4 + 2] * inv, s[q * 4 + 3] * inv);
      }
    }
  };

  // residual of the epilogue: fetched under the last slab's MFMAs (after stores nothing could be hoisted)
  const uint16_t* const rr = PRE ? a.out : a.res;       // (PRE: h, parked in `out` by this wave; never wrapped)
  const int rr_ld = PRE ? a.ldo : a.ldres;
  __syncthreads();                                        // the tables are in LDS
  issue(std::integral_constant<int, 0>{});
  issue(std::integral_constant<int, 1>{});
  xa_static_for<NSLAB>([&](auto T) __attribute__((always_inline)) {
    constexpr int tt = decltype(T)::value, t = tt - NPRE;
    // this wave's five pieces of slab tt have landed (slab tt + 1's five may still be in flight), then everybody's
    if constexpr (tt + 1 < NSLAB) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(DBG & 32)) asm volatile("s_barrier" ::: "memory");
    if constexpr (tt + 2 < NSLAB) issue(std::integral_constant<int, tt + 2>{});   // (its stage was read at step tt - 1)
    if constexpr (PRE && tt == 3) prefetch_pre();
    if constexpr (tt + 1 == NSLAB) load_rows(rr, rr_ld, !PRE);
    const char* st = smem + (tt % XA_NS) * XA_SLAB;
    // the slab's 40 weight fragments (ks 0 / 1 x 20 blocks) as a hand-made software pipeline: eight ds_read_b128 stay in
    // flight ahead of the MFMA that consumes the oldest (left to itself the compiler keeps ~5 reads ahead of an MFMA that
    // depends on each of them, with nothing else to issue); nine rotating fragment registers
    v8_t q[QD + 1];
    const int so0 = ((0 * 4 + g) ^ (r16 & 7)) << 4, so1 = ((1 * 4 + g) ^ (r16 & 7)) << 4;
    auto load_frag = [&](int i) __attribute__((always_inline)) -> v8_t {
      return *reinterpret_cast<const v8_t*>(st + ((i % 20) * 16 + r16) * 128 + (i < 20 ? so0 : so1));
    };
    v8_t bfr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (t < 0) bfr[ks] = xf[tt * 2 + ks];
      else if constexpr (t < 10) {
        constexpr int f = (t % 5) * 2;
        if constexpr (PRE) bfr[ks] = __builtin_bit_cast(v8_t, u32x4_t{hq[f + ks][0], hq[f + ks][1], hq[f + ks][2], hq[f + ks][3]});
        else bfr[ks] = xf[f + ks];
      } else {
        constexpr int f = (t - 10) * 2;
        bfr[ks] = __builtin_bit_cast(v8_t, u32x4_t{pf[f + ks][0], pf[f + ks][1], pf[f + ks][2], pf[f + ks][3]});
      }
    }
    if constexpr (!(DBG & 4)) {
#pragma unroll
      for (int i = 0; i < QD; ++i) q[i] = load_frag(i);
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) {
      if constexpr (!(DBG & 4)) {
        if (i + QD < 40) q[(i + QD) % (QD + 1)] = load_frag(i + QD);
        __builtin_amdgcn_sched_barrier(0);
      }
      const v8_t af = (DBG & 4) ? bfr[i / 20] : q[i % (QD + 1)];
      if constexpr (DBG & 2) acc[i % 20] += __builtin_bit_cast(f32x4_t, af);
      else acc[i % 20] = E::mfma16(af, bfr[i / 20], acc[i % 20]);
      if constexpr (!(DBG & 4)) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PRE && t == -1) finish_pre();
    if constexpr (!(DBG & 8)) {
      if constexpr (t == 4) softmax_half(std::integral_constant<int, 0>{});
      if constexpr (t == 9) softmax_half(std::integral_constant<int, 1>{});
    }
  });


  // ---- epilogue: + bias + residual, 16-bit stores, row moments of the stored values per 160-column tile
  if constexpr (DBG & 16) {
    if (acc[0][0] == 12345.f) a.out[m] = 1;
    return;
  }
#pragma unroll
  for (int nb = 0; nb < 20; ++nb)                        // (L2-resident 1.25 KB; all twenty loads before the first store)
    bo[nb] = a.bias_o ? *reinterpret_cast<const f32x4_t*>(a.bias_o + nb * 16 + 4 * g) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  float sm[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
  u32x2_t pk[4];
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) {
    if ((nb & 3) == 0) rt_fill(stg, lane, rv4[(nb >> 2) * 2], rv4[(nb >> 2) * 2 + 1]);
    const u32x2_t r = rt_get(stg, r16, g, nb & 3);
    const f32x4_t v = acc[nb] + bo[nb] + f32x4_t{E::lo(r[0]), E::hi(r[0]), E::lo(r[1]), E::hi(r[1])};
    const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
    pk[nb & 3] = u32x2_t{o0, o1};
    if ((nb & 3) == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rt_put(stg, r16, g, j, pk[j][0], pk[j][1]);
      rt_flush(stg, lane, a.out + (size_t)m_w0 * a.ldo, a.ldo, (nb >> 2) * 64);
    }
    const float r0 = E::lo(o0), r1 = E::hi(o0), r2 = E::lo(o1), r3 = E::hi(o1);
    sm[nb / 10] += (r0 + r1) + (r2 + r3);
    sq[nb / 10] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
  }
  if (a.row_stats_out) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float s0 = sm[t], s1 = sq[t];
      s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
      s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
      if (g == 0) *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)m * 2 + t) * 2) = f32x2_t{s0, s1};
    }
  }
}

// ---- C = 640 / 1280 (the 32x32, 16x16 and 8x8 levels): the same two chained GEMMs, re-tiled for few rows -----------------
// At those levels a 128-row tile per workgroup leaves most of the chip idle (M = 8192 / 2048 / 512 rows), so a workgroup
// takes 64 rows (four waves of 16) and ONE 320-column group of the output: grid = (M / 64) x (C / 320).  Every column group
// of a row tile recomputes the logits and the softmax (K = C, twice 320 logit columns) and contracts its own 320 rows of
// H^T -- redundant MFMA work (x2 at C = 640, x4 at C = 1280) bought back by the two launches (to_q, attention) and the
// q / o round trips that disappear: the three-launch chain costs 46 us per block at these levels, of which ~7 us are MFMA
// work (profiles/r04_step_timeline.txt).  The input rows are not register resident (C / 32 fragments would not fit): the two
// B fragments of a slab are loaded from global memory one slab ahead, in front of the slab DMAs of the step so that one
// counted vmcnt covers both.  Slab = 320 weight rows x 64 k (40 KB), three-stage LDS-DMA ring, ten pieces per wave.
template <int C, int EDT>
__global__ void __launch_bounds__(256, 1) xattn_wide_kernel(const XAArgs a) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  constexpr int NS1 = C / 64, NSLAB = 2 * NS1 + 10, NSPLIT = C / 320, BM = 64, QD = XA_QD;
  static_assert(NS1 % 2 == 0, "the first GEMM's slab loop is unrolled by two");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabs = reinterpret_cast<float*>(smem + XA_TAB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = lid / NSPLIT, cg = lid - tile * NSPLIT;       // (the column groups of a row tile share G: neighbours)
  const int m_blk = tile * BM;
  const int b = m_blk / a.rows_per_batch;
  const int m = m_blk + wave * 16 + r16;

  for (int i = tid; i < 2 * XA_S; i += 256)
    tabs[i] = i < XA_S ? a.gcs[(size_t)b * XA_S + i] : a.gbias[(size_t)b * XA_S + i - XA_S];
  float mean = 0.f, rstd = 1.f;
  if (a.ln_stats) {
    const f32x2_t* pm = reinterpret_cast<const f32x2_t*>(a.ln_stats) + (size_t)m * a.ln_tiles;
    float sm = 0.f, sq = 0.f;
    for (int t = 0; t < a.ln_tiles; ++t) { const f32x2_t v = pm[t]; sm += v[0]; sq += v[1]; }
    mean = sm * (1.0f / C);
    rstd = rsqrtf(fmaxf(sq * (1.0f / C) - mean * mean, 0.f) + a.ln_eps);
  }

  const int lrow = lane >> 3, kslot = (lane & 7) ^ lrow;
  // piece j of a wave = strip wave + 4 j (8 rows): row 8 (wave + 4 j) + lrow -- the j part rides in the scalar offset
  const int vg0 = ((8 * wave + lrow) * C + kslot * 8) * 2, vh0 = ((8 * wave + lrow) * XA_S + kslot * 8) * 2;
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.gt + (size_t)b * XA_S * C, (uint32_t)XA_S * C * 2u);
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.ht + ((size_t)b * C + (size_t)cg * 320) * XA_S, 320u * XA_S * 2u);
  // slab t: t < NS1 the logit rows 0 .. 319 (heads 0-3) of G^T, channels 64 t ..; t < 2 NS1 rows 320 .. 639; then the ten
  // 64-wide logit slices of this column group's 320 rows of H^T
  // (a wave's ten pieces of a slab are NOT issued as one burst: ten back-to-back LDS-DMA instructions hold an in-order
  //  wave ~800 cycles in VMEM issue before its first MFMA of the step -- one piece rides behind every fourth MFMA)
  // SRC (compile time): 0 = G^T slab tn (runtime, < 2 NS1), 1 = H^T slab tn (index among the ten), -1 = nothing.  The
  // descriptor is never chosen at run time: a run-time select between two descriptors lands in VGPRs and scratch.
  auto issue_piece = [&](auto SRC, int tn, int stage, auto J) __attribute__((always_inline)) {
    constexpr int src = decltype(SRC)::value, j = decltype(J)::value;
    char* st = smem + stage * XA_SLAB + wave * 1024 + j * 4096;
    if constexpr (src == 0) {
      const int half = tn >= NS1 ? 1 : 0;
      const int so = (half * 320 * C + (tn - half * NS1) * 64) * 2 + j * (32 * C * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (xa_lds_ptr_t)st, 16, vg0, so, 0, 0);
    } else if constexpr (src == 1) {
      const int so = tn * 64 * 2 + j * (32 * XA_S * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (xa_lds_ptr_t)st, 16, vh0, so, 0, 0);
    }
  };
  const uint16_t* const xr = a.x + (size_t)m * a.ldx + g * 8;
  auto load_x = [&](int t, v8_t (&dst)[2]) __attribute__((always_inline)) {
    const int kt = t >= NS1 ? t - NS1 : t;
    dst[0] = *reinterpret_cast<const v8_t*>(xr + kt * 64);
    dst[1] = *reinterpret_cast<const v8_t*>(xr + kt * 64 + 32);
  };

  f32x4_t acc[20];
  uint32_t pf[20][4];
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto softmax_half = [&](auto HF) __attribute__((always_inline)) {
    constexpr int hf = decltype(HF)::value;
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      float s[20];
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int n0 = hf * 320 + (5 * hh + q) * 16 + 4 * g;
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(tabs + n0);
        const f32x4_t gb = *reinterpret_cast<const f32x4_t*>(tabs + XA_S + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s[q * 4 + i] = rstd * (acc[5 * hh + q][i] - mean * cs[i]) + gb[i];
          mx = fmaxf(mx, s[q * 4 + i]);
        }
        acc[5 * hh + q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i] - mx);
        sum += s[i];
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int gbk = hf * 20 + 5 * hh + q;
        pf[gbk >> 1][(gbk & 1) * 2 + 0] = E::pack2(s[q * 4 + 0] * inv, s[q * 4 + 1] * inv);
        pf[gbk >> 1][(gbk & 1) * 2 + 1] = E::pack2(s[q * 4 + 2] * inv, s[q * 4 + 3] * inv);
      }
    }
  };

  // one slab against the wave's two B fragments: 40 weight fragments, QD reads in flight ahead of the MFMA that consumes
  // the oldest (the hand-made pipeline of xattn_block_kernel)
  const int so0 = ((0 * 4 + g) ^ (r16 & 7)) << 4, so1 = ((1 * 4 + g) ^ (r16 & 7)) << 4;
  auto slab_mma = [&](const char* st, const v8_t b0, const v8_t b1, auto SRC, int tn, int stage_n) __attribute__((always_inline)) {
    v8_t q[QD + 1];
    auto load_frag = [&](int i) __attribute__((always_inline)) -> v8_t {
      return *reinterpret_cast<const v8_t*>(st + ((i % 20) * 16 + r16) * 128 + (i < 20 ? so0 : so1));
    };
#pragma unroll
    for (int i = 0; i < QD; ++i) q[i] = load_frag(i);
    xa_static_for<40>([&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      if constexpr (i + QD < 40) q[(i + QD) % (QD + 1)] = load_frag(i + QD);
      __builtin_amdgcn_sched_barrier(0);
      acc[i % 20] = E::mfma16(q[i % (QD + 1)], i < 20 ? b0 : b1, acc[i % 20]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i % 4 == 1 && decltype(SRC)::value >= 0) {      // DMA piece i / 4 of the slab two steps ahead
        issue_piece(SRC, tn, stage_n, std::integral_constant<int, i / 4>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };

  __syncthreads();                                        // the tables are in LDS
  v8_t xq[2][2];
  load_x(0, xq[0]);
  __builtin_amdgcn_sched_barrier(0);
  xa_static_for<10>([&](auto J) __attribute__((always_inline)) { issue_piece(std::integral_constant<int, 0>{}, 0, 0, J); });
  xa_static_for<10>([&](auto J) __attribute__((always_inline)) { issue_piece(std::integral_constant<int, 0>{}, 1, 1, J); });
  // ---- first GEMM: logits, one half (four heads) at a time.  Per step: x fragments of slab t + 1, then (spread over the
  //      step's MFMAs) the DMAs of slab t + 2; the wait at the top of a step leaves only the previous step's ten DMAs in
  //      flight.  `stage` = t % 3 is carried, not divided.
  int stage = 0;
  auto step1 = [&](int t, auto BUF, auto SRC, int tn) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
    asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    if (t + 1 < 2 * NS1) load_x(t + 1, xq[buf ^ 1]);
    __builtin_amdgcn_sched_barrier(0);                    // (x fragments in FRONT of the DMAs: see the vmcnt above)
    const int stage_n = stage == 0 ? 2 : stage - 1;       // (t + 2) % 3
    slab_mma(smem + stage * XA_SLAB, xq[buf][0], xq[buf][1], SRC, tn, stage_n);
    stage = stage == 2 ? 0 : stage + 1;
  };
  constexpr std::integral_constant<int, 0> SG{};
  constexpr std::integral_constant<int, 1> SH{};
  constexpr std::integral_constant<int, 0> B0{};
  constexpr std::integral_constant<int, 1> B1{};
#pragma unroll 1
  for (int t = 0; t < NS1; t += 2) {
    step1(t, B0, SG, t + 2);
    step1(t + 1, B1, SG, t + 3);
  }
  softmax_half(std::integral_constant<int, 0>{});
#pragma unroll 1
  for (int t = NS1; t < 2 * NS1 - 2; t += 2) {
    step1(t, B0, SG, t + 2);
    step1(t + 1, B1, SG, t + 3);
  }
  step1(2 * NS1 - 2, B0, SH, 0);                          // (the last two steps already fetch H^T slabs 0 and 1)
  step1(2 * NS1 - 1, B1, SH, 1);
  softmax_half(std::integral_constant<int, 1>{});

  // ---- second GEMM: this column group's 320 outputs, K = 640 probabilities out of the registers
  const int n_out = cg * 320;
  // (residual in / rows out as whole 128-byte lines through the wave's private LDS tile: rowtile_io.h)
  const int m_w0 = m_blk + wave * 16;
  char* const stg = smem + XA_STG + wave * RT_TILE;
  u32x4_t rv4[10];
  f32x4_t bo[20];
  xa_static_for<10>([&](auto U) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value;
    constexpr int t = 2 * NS1 + u;
    if constexpr (u + 1 < 10) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (u + 1 == 10) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int q = lane + 64 * jj;
          rv4[c * 2 + jj] = a.res ? *reinterpret_cast<const u32x4_t*>(a.res + (size_t)(m_w0 + (q >> 3)) * a.ldres + n_out + c * 64 + (q & 7) * 8)
                                  : u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    const v8_t b0 = __builtin_bit_cast(v8_t, u32x4_t{pf[2 * u][0], pf[2 * u][1], pf[2 * u][2], pf[2 * u][3]});
    const v8_t b1 = __builtin_bit_cast(v8_t, u32x4_t{pf[2 * u + 1][0], pf[2 * u + 1][1], pf[2 * u + 1][2], pf[2 * u + 1][3]});
    slab_mma(smem + (t % XA_NS) * XA_SLAB, b0, b1, std::integral_constant<int, (u + 2 < 10 ? 1 : -1)>{}, u + 2, (t + 2) % XA_NS);
  });

  // ---- epilogue: + bias + residual, 16-bit stores, row moments of the stored values per 160-column tile
#pragma unroll
  for (int nb = 0; nb < 20; ++nb)
    bo[nb] = a.bias_o ? *reinterpret_cast<const f32x4_t*>(a.bias_o + n_out + nb * 16 + 4 * g) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  float sm[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
  u32x2_t pk[4];
#pragma unroll
  for (int nb = 0; nb < 20; ++nb) {
    if ((nb & 3) == 0) rt_fill(stg, lane, rv4[(nb >> 2) * 2], rv4[(nb >> 2) * 2 + 1]);
    const u32x2_t r = rt_get(stg, r16, g, nb & 3);
    const f32x4_t v = acc[nb] + bo[nb] + f32x4_t{E::lo(r[0]), E::hi(r[0]), E::lo(r[1]), E::hi(r[1])};
    const uint32_t o0 = E::pack2(v[0], v[1]), o1 = E::pack2(v[2], v[3]);
    pk[nb & 3] = u32x2_t{o0, o1};
    if ((nb & 3) == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rt_put(stg, r16, g, j, pk[j][0], pk[j][1]);
      rt_flush(stg, lane, a.out + (size_t)m_w0 * a.ldo, a.ldo, n_out + (nb >> 2) * 64);
    }
    const float r0 = E::lo(o0), r1 = E::hi(o0), r2 = E::lo(o1), r3 = E::hi(o1);
    sm[nb / 10] += (r0 + r1) + (r2 + r3);
    sq[nb / 10] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
  }
  if (a.row_stats_out) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float s0 = sm[t], s1 = sq[t];
      s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
      s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
      if (g == 0)
        *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)m * (C / 160) + 2 * cg + t) * 2) = f32x2_t{s0, s1};
    }
  }
}

}  // namespace

extern "C" int pp_xattn_block_supported(int M, int c, int rows_per_batch, int nctx, int heads) {
  if (heads != XA_HEADS || nctx <= 0 || nctx > XA_KP || M <= 0 || rows_per_batch <= 0 || M % rows_per_batch) return 0;
  const int bm = c == XA_C ? XA_BM : (c == 640 || c == 1280) ? 64 : 0;     // C = 320: 128-row tiles; wider: 64-row tiles
  return (bm && M % bm == 0 && rows_per_batch % bm == 0) ? 1 : 0;
}

extern "C" int pp_xattn_fold(const void* k, int ldk, const void* vt, int ldvt, int batch, int nctx, int heads, int c,
                             const void* wq, const float* q_colsum, const float* q_bias, const void* wo, float scale,
                             void* gt, float* gcs, float* gbias, void* ht, int kperm, int dtype, void* stream) {
  if (!k || !vt || !wq || !wo || !gt || !gcs || !gbias || !ht || batch <= 0 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if (!pp_xattn_block_supported(XA_BM, c, XA_BM, nctx, heads)) return PP_ERR_UNSUPPORTED;
  if (ldk < c || ldvt < nctx || (c & 1)) return PP_ERR_BAD_ARG;
  if (kperm < 0 || kperm > 2) return PP_ERR_BAD_ARG;
  if ((kperm & 1) && c != XA_C) return PP_ERR_UNSUPPORTED;      // (only the C = 320 block kernel chains a Linear in front)
  if (c / XA_HEADS > XF_MAXD || (c / XA_HEADS) % 8 || nctx > XA_KP) return PP_ERR_UNSUPPORTED;
  const float qscale = scale * 1.44269504088896340736f;   // the softmax runs in the exp2 domain
  hipStream_t st = (hipStream_t)stream;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL((xattn_fold_g_kernel<EDT>), dim3((c / 2 + 255) / 256, XA_HEADS, batch), dim3(256), 0, st,
                                         (const uint16_t*)k, ldk, nctx, (const uint16_t*)wq, qscale, (uint32_t*)gt, c, kperm));
  PP_CHECK_LAUNCH("xattn_fold_g_kernel");
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL((xattn_fold_h_kernel<EDT>), dim3((c + 255) / 256, XA_HEADS, batch), dim3(256), 0, st,
                                         (const uint16_t*)vt, ldvt, nctx, (const uint16_t*)wo, (uint32_t*)ht, c, kperm));
  PP_CHECK_LAUNCH("xattn_fold_h_kernel");
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL((xattn_fold_vec_kernel<EDT>), dim3((batch * XA_S + 255) / 256), dim3(256), 0, st,
                                         (const uint16_t*)k, ldk, batch, nctx, q_colsum, q_bias, qscale, gcs, gbias, c));
  PP_CHECK_LAUNCH("xattn_fold_vec_kernel");
  if (q_colsum) {
    const long long rows = (long long)batch * XA_S;
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((xattn_colsum_kernel<EDT>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                           (hipStream_t)stream, (const uint32_t*)gt, gcs, rows, c));
    PP_CHECK_LAUNCH("xattn_colsum_kernel");
  }
  return PP_OK;
}

extern "C" int pp_xattn_block(const void* x, int ldx, const void* res, int ldres, const float* ln_stats, int ln_tiles,
                              float ln_eps, const void* gt, const float* gcs, const float* gbias, const void* ht,
                              const float* bias_o, void* out, int ldo, float* row_stats_out, int M, int c,
                              int rows_per_batch, int src_wrap_rows, const void* pre_w, const float* pre_b, int dtype,
                              void* stream) {
  if (!x || !gt || !gcs || !gbias || !ht || !out || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if (!pp_xattn_block_supported(M, c, rows_per_batch, XA_KP, XA_HEADS)) return PP_ERR_UNSUPPORTED;
  if (src_wrap_rows < 0 || (src_wrap_rows > 0 && (M > 2 * src_wrap_rows || src_wrap_rows % rows_per_batch))) return PP_ERR_BAD_ARG;
  if (src_wrap_rows > 0 && c != XA_C) return PP_ERR_UNSUPPORTED;
  if (pre_w && c != XA_C) return PP_ERR_UNSUPPORTED;
  if (pre_b && !pre_w) return PP_ERR_BAD_ARG;
  if (ldx < c || (ldx & 7) || ldo < c || (ldo & 7) || (res && (ldres < c || (ldres & 7)))) return PP_ERR_BAD_ARG;   // (16-byte row pieces)
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 15) return PP_ERR_BAD_ARG;
  if (ln_stats && ln_tiles <= 0) return PP_ERR_BAD_ARG;
  XAArgs a;
  a.x = (const uint16_t*)x; a.ldx = ldx;
  a.res = (const uint16_t*)res; a.ldres = ldres;
  a.ln_stats = ln_stats; a.ln_tiles = ln_tiles; a.ln_eps = ln_eps;
  a.gt = (const uint16_t*)gt; a.gcs = gcs; a.gbias = gbias; a.ht = (const uint16_t*)ht;
  a.bias_o = bias_o;
  a.out = (uint16_t*)out; a.ldo = ldo;
  a.row_stats_out = row_stats_out;
  a.M = M; a.rows_per_batch = rows_per_batch;
  a.src_wrap = src_wrap_rows;
  a.pre_w = (const uint16_t*)pre_w; a.pre_b = pre_b;
  if (c != XA_C) {
    auto gow = [&](auto kern, int, int grid) -> int {
      if (pp_func_lds(reinterpret_cast<const void*>(kern), XA_LDS, "hipFuncSetAttribute(xattn wide)") != PP_OK) return PP_ERR_LAUNCH;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), XA_LDS, (hipStream_t)stream, a);
      PP_CHECK_LAUNCH("xattn_wide_kernel");
      return PP_OK;
    };
    if (c == 640)
      return dtype == PP_DT_F16 ? gow(xattn_wide_kernel<640, PP_DT_F16>, 0, (M / 64) * 2)
                                : gow(xattn_wide_kernel<640, PP_DT_BF16>, 0, (M / 64) * 2);
    return dtype == PP_DT_F16 ? gow(xattn_wide_kernel<1280, PP_DT_F16>, 1, (M / 64) * 4)
                              : gow(xattn_wide_kernel<1280, PP_DT_BF16>, 1, (M / 64) * 4);
  }
  auto go = [&](auto kern, int) -> int {
    if (pp_func_lds(reinterpret_cast<const void*>(kern), XA_LDS, "hipFuncSetAttribute(xattn block)") != PP_OK) return PP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(M / XA_BM), dim3(512), XA_LDS, (hipStream_t)stream, a);
    PP_CHECK_LAUNCH("xattn_block_kernel");
    return PP_OK;
  };
#ifdef PP_LAB
  if (dtype == PP_DT_BF16) switch (pp_lab_env("PP_XA_DBG", 0)) {
      case 1: return go(xattn_block_kernel<PP_DT_BF16, 1>, 0);
      case 2: return go(xattn_block_kernel<PP_DT_BF16, 2>, 0);
      case 4: return go(xattn_block_kernel<PP_DT_BF16, 4>, 0);
      case 6: return go(xattn_block_kernel<PP_DT_BF16, 6>, 0);
      case 7: return go(xattn_block_kernel<PP_DT_BF16, 7>, 0);
      case 8: return go(xattn_block_kernel<PP_DT_BF16, 8>, 0);
      case 16: return go(xattn_block_kernel<PP_DT_BF16, 16>, 0);
      case 33: return go(xattn_block_kernel<PP_DT_BF16, 33>, 0);
      case 39: return go(xattn_block_kernel<PP_DT_BF16, 39>, 0);
      default: break;
    }
  if (dtype == PP_DT_BF16) switch (pp_lab_env("PP_XA_QD", XA_QD)) {
      case 4: return go(xattn_block_kernel<PP_DT_BF16, 0, 4>, 0);
      case 6: return go(xattn_block_kernel<PP_DT_BF16, 0, 6>, 0);
      case 10: return go(xattn_block_kernel<PP_DT_BF16, 0, 10>, 0);
      case 12: return go(xattn_block_kernel<PP_DT_BF16, 0, 12>, 0);
      default: break;
    }
#endif
  if (pre_w) {
    auto gop = [&](auto kern) -> int {
      if (pp_func_lds(reinterpret_cast<const void*>(kern), XA_LDS, "hipFuncSetAttribute(xattn block, pre)") != PP_OK) return PP_ERR_LAUNCH;
      hipLaunchKernelGGL(kern, dim3(M / XA_BM), dim3(512), XA_LDS, (hipStream_t)stream, a);
      PP_CHECK_LAUNCH("xattn_block_kernel(pre)");
      return PP_OK;
    };
    if (dtype == PP_DT_F16) return gop(xattn_block_kernel<PP_DT_F16, 0, XA_QD, true>);
    return gop(xattn_block_kernel<PP_DT_BF16, 0, XA_QD, true>);
  }
  if (dtype == PP_DT_F16) return go(xattn_block_kernel<PP_DT_F16>, PP_DT_F16);
  return go(xattn_block_kernel<PP_DT_BF16>, PP_DT_BF16);
}
