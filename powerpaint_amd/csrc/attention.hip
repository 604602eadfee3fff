// Fused attention forward (flash-style, online softmax) for SD-1.5 head dims {40, 80, 160} on gfx950.
//
//   O[b,i,h,:] = softmax_j(Q[b,i,h,:].K[b,j,h,:] * scale) V[b,j,h,:]        no mask, non-causal
//
// CDNA4 mapping (wave = 64 lanes, v_mfma_f32_32x32x16_bf16):
//   * one wave owns 32 queries; a 256-thread block = 4 waves = 128 queries of one (batch, head);
//   * S^T = K Q^T ("swapped" QK^T): MFMA A-operand = K tile rows (keys), B-operand = Q rows (queries).  The C/D layout
//     then gives every lane ONE query column and 16 of the 32 keys -> the softmax row statistics (max, sum, rescale
//     factor) are per-lane scalars; the only cross-lane traffic is one lane^32 exchange of the running max;
//   * O^T = V^T P^T: A-operand = V^T tile (rows = head-dim columns), B-operand = P^T.  The fp32 S^T accumulators of a
//     lane, packed to bf16 in register order, ARE a valid B-operand as long as V^T is read with the same key
//     permutation (the contraction index may be permuted freely) -- no permlane / LDS round trip for P;
//   * V arrives already TRANSPOSED from the producing GEMM's epilogue (vt[b][h*d+j][key]), so both K and V^T tiles
//     are staged with full-line 16-byte loads and no in-kernel transpose;
//   * K and V^T tile rows are padded to an odd number of 16-B slots and the keys of a V^T row are permuted inside
//     each 16-key block so that every MFMA operand fragment is ONE conflict-free ds_read_b128;
//   * next K/V tile is prefetched into registers while the current one is consumed (global latency hidden behind
//     the MFMAs), exp2 with the softmax scale folded into one FMA.
#include <cstdlib>
#include <type_traits>

#include "pp_common.h"

namespace {

constexpr int KB = 64;   // keys per tile
constexpr int QW = 32;   // queries per wave
constexpr int NW = 4;    // waves per block

template <int D>
struct Cfg {
  static constexpr int DP = (D + 15) / 16 * 16;   // QK^T contraction length (zero padded)
  static constexpr int DS = DP / 16;              // MFMA k-steps for QK^T
  static constexpr int DT = (D + 31) / 32;        // 32-wide output tiles of O^T
  static constexpr int KS = DP * 2 + 16;          // K tile row stride in bytes (odd number of 16-B slots)
  static constexpr int VS = KB * 2 + 16;          // V^T tile row stride in bytes (odd number of 16-B slots)
  static constexpr int KBYTES = KB * KS;
  static constexpr int VROWS = DT * 32;
  static constexpr int VBYTES = VROWS * VS;
  static constexpr int KPIECES = KB * (DP / 8);   // 16-B pieces per K tile
  static constexpr int VPIECES = D * (KB / 8);    // 16-B pieces per V^T tile (rows < D only)
  static constexpr int KPT = (KPIECES + 255) / 256;
  static constexpr int VPT = (VPIECES + 255) / 256;
};

constexpr float RESCALE_THR = 8.0f;   // defer the O rescale while the running max grows by < 2^8 (log2 domain)

template <int D, int EDT, bool QR>     // QR: the K / V-reuse loop over query blocks (its own instantiation: the loop-carried
__global__ void __launch_bounds__(256, (D <= 80 ? 2 : 1))   // state costs registers the d = 80 / 160 kernels do not have)
attn_fwd_kernel(const uint16_t* __restrict__ q, int ldq, const uint16_t* __restrict__ k, int ldk,
                const uint16_t* __restrict__ vt, int ldvt, uint16_t* __restrict__ o, int ldo, int heads, int nq, int nk,
                float scale_log2e, int qrep) {
  using C = Cfg<D>;
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  constexpr int BUF = C::KBYTES + C::VBYTES;
  // (Folding scale and -max into the spare contraction slot of the padded d = 40 QK^T was tried: it needs Q pre-scaled
  //  in bf16, a second rounding that large scores amplify -- rejected by test_attention_strided_qkv_and_spike.)
  constexpr bool MFMA_SUM = C::VROWS > D;       // row D of V^T := 1 -> the P.V MFMAs also produce the denominator
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (K tile | V^T tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  // qrep > 1 (cross-attention at 64x64: nk <= 2 tiles, thousands of workgroups): a workgroup serves qrep blocks of 128
  // queries in turn.  After the first block both LDS buffers hold the whole K / V^T of this (batch, head): the later
  // blocks recompute from LDS with no tile loads and no barriers -- the K / V round trip (the longest latency hop of
  // this short kernel) is paid once per qrep blocks and the launch has 1 / qrep of the workgroups.
  const int reps = QR ? qrep : 1;
  int q0 = blockIdx.x * (QW * NW * reps) + wave * QW;
  const int qi = lane & 31, half = lane >> 5;

  if constexpr (MFMA_SUM) {   // padding rows of the V^T tile (both buffers), written once: row D all ones, the rest zero
    constexpr int PADP = (C::VROWS - D) * (C::VS / 8);
    for (int i = tid; i < 2 * PADP; i += 256) {
      const int bufi = i / PADP, r = i - bufi * PADP;
      const uint32_t v = (r < C::VS / 8) ? E::pack2(1.0f, 1.0f) : 0u;
      *reinterpret_cast<u32x2_t*>(smem + bufi * BUF + C::KBYTES + D * C::VS + r * 8) = u32x2_t{v, v};
    }
  }

  // ---- Q fragments (B operand): lane = query qi, k-slot = 8*half + jj  ->  Q[q0+qi][16 s + 8 half + jj]
  u32x4_t qraw[C::DS];
  // (macros, not lambdas: a lambda that captures the loop-carried q0 / the tile lambda by reference makes LLVM keep the
  //  closure -- and everything it points to -- in scratch memory)
#define PP_LOAD_Q()                                                                                     \
  do {                                                                                                  \
    const int qrow_ = q0 + qi;                                                                          \
    const uint16_t* qp_ = q + ((size_t)b * nq + (qrow_ < nq ? qrow_ : 0)) * ldq + h * D;                \
    _Pragma("unroll") for (int s_ = 0; s_ < C::DS; ++s_) {                                              \
      const int dc_ = 16 * s_ + 8 * half;                                                               \
      u32x4_t v_ = {0u, 0u, 0u, 0u};                                                                    \
      if (qrow_ < nq && dc_ < D) v_ = *reinterpret_cast<const u32x4_t*>(qp_ + dc_);                     \
      qraw[s_] = v_;                                                                                    \
    }                                                                                                   \
  } while (0)
  PP_LOAD_Q();

  // ---- tile loads: per-piece voffsets are constants; the walk over key tiles is SCALAR arithmetic on the buffer
  // descriptors (base += tile, size shrinks), and rows past the end of K fall outside the descriptor -> zeros.
  const uint16_t* k_b = k + (size_t)b * nk * ldk;
  const uint16_t* vt_bh = vt + ((size_t)b * heads + h) * D * (size_t)ldvt;
  constexpr int KSL = C::DP / 8;          // 16-B pieces per K row in LDS
  constexpr int KSLV = D / 8;             // ... of which come from memory
  int kvo[C::KPT], vvo[C::VPT];
#pragma unroll
  for (int i = 0; i < C::KPT; ++i) {
    const int pc = tid + i * 256;
    const int row = pc / KSL, sl = pc - row * KSL;
    kvo[i] = (pc < C::KPIECES && sl < KSLV) ? (row * ldk + h * D + sl * 8) * 2 : (int)PP_OOB;
  }
#pragma unroll
  for (int i = 0; i < C::VPT; ++i) {
    const int pc = tid + i * 256;
    vvo[i] = (pc < C::VPIECES) ? ((pc >> 3) * ldvt + (pc & 7) * 8) * 2 : (int)PP_OOB;
  }
  u32x4_t kreg[C::KPT], vreg[C::VPT];
  auto load_tile = [&](int t0) {
    const int left = nk - t0;                                  // keys remaining (>= 1)
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(k_b + (size_t)t0 * ldk, (uint32_t)left * (uint32_t)ldk * 2u);
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vt_bh + t0, ((uint32_t)D * (uint32_t)ldvt - (uint32_t)t0) * 2u);
#pragma unroll
    for (int i = 0; i < C::KPT; ++i) {
      const int vo = kvo[i];
      kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int vo = (((tid + i * 256) & 7) * 8 < left) ? vvo[i] : (int)PP_OOB;
      vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, 0, 0);
    }
  };
  auto store_tile = [&](int bufi) {
    char* ks = smem + bufi * BUF;
    char* vs = ks + C::KBYTES;
#pragma unroll
    for (int i = 0; i < C::KPT; ++i) {
      const int pc = tid + i * 256;
      if (pc < C::KPIECES) {
        const int row = pc / KSL, sl = pc - row * KSL;
        *reinterpret_cast<u32x4_t*>(ks + row * C::KS + sl * 16) = kreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int pc = tid + i * 256;
      if (pc < C::VPIECES) {
        const int row = pc >> 3, sl = pc & 7;
        // keys are permuted inside every 16-key block to [0-3, 8-11 | 4-7, 12-15] so that the 8 keys one lane half
        // contracts over (it owns S^T rows 4h..4h+3 and 8+4h..8+4h+3) are ONE contiguous 16-byte fragment
        char* blk = vs + row * C::VS + (sl >> 1) * 32;
        const int pos = (sl & 1) * 8;
        *reinterpret_cast<u32x2_t*>(blk + pos) = u32x2_t{vreg[i][0], vreg[i][1]};
        *reinterpret_cast<u32x2_t*>(blk + 16 + pos) = u32x2_t{vreg[i][2], vreg[i][3]};
      }
    }
  };

  f32x16_t oacc[C::DT];
#pragma unroll
  for (int t = 0; t < C::DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -1.0e30f;   // running max in the log2 domain (raw score * scale * log2 e)
  float l_run = 0.f;

  // one 64-key tile: S^T = K Q^T, online softmax (per-lane row state), O^T += V^T P^T
  // tail_tag: 0 full tile | 1 last tile, keys >= nk masked | 2 last tile with <= 32 live keys: only the first 32-key
  // block is computed at all (the masked block would contribute exp2(-1e30 - m) = 0 exactly).  Cross-attention over
  // 77 text tokens is one full tile + one 13-key tail: a quarter of its MFMA / exp work was spent on padding.
  auto tile = [&](const char* ks, const char* vs, int t0, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value != 0;
    constexpr int JN = decltype(tail_tag)::value == 2 ? 1 : 2;
    f32x16_t sacc[2];
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::DS; ++s) {
        const v8_t kf = *reinterpret_cast<const v8_t*>(ks + (32 * j + qi) * C::KS + (16 * s + 8 * half) * 2);
        sacc[j] = E::mfma32(kf, __builtin_bit_cast(v8_t, qraw[s]), s == 0 ? zero : sacc[j]);
      }
    }
    float tmax = -1.0e30f;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (TAIL) {   // keys >= nk exist only in the last tile
          const int key = t0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= nk) sacc[j][r] = -1.0e30f;
        }
        tmax = fmaxf(tmax, sacc[j][r]);
      }
    {   // cross-half row maximum through the VALU swap (lanes i <-> i + 32): no ds_bpermute round trip, no lgkmcnt drain
        // (inline asm: see attn_pipe_kernel; s_nop 1 = the wait states between the VALU write and the swap)
      float r0 = tmax, r1 = tmax;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(r0), "+v"(r1));
      tmax = fmaxf(r0, r1);
    }
    float psum = 0.f;
    v8_t pf[2][2];
    float p[2][16];
    {
      tmax *= scale_log2e;
      if (!__all(tmax - m_run <= RESCALE_THR)) {
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        if constexpr (!MFMA_SUM) l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      }
      const f32x2_t c2 = {scale_log2e, scale_log2e}, nm2 = {-m_run, -m_run};
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t s2 = {sacc[j][r], sacc[j][r + 1]};
          const f32x2_t e2 = __builtin_elementwise_fma(s2, c2, nm2);      // v_pk_fma_f32: two scores per VALU op
          p[j][r] = __builtin_amdgcn_exp2f(e2[0]);
          p[j][r + 1] = __builtin_amdgcn_exp2f(e2[1]);
          if constexpr (!MFMA_SUM) psum += p[j][r] + p[j][r + 1];
        }
    }
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t w;
        w[0] = E::pack2(p[j][8 * u + 0], p[j][8 * u + 1]);
        w[1] = E::pack2(p[j][8 * u + 2], p[j][8 * u + 3]);
        w[2] = E::pack2(p[j][8 * u + 4], p[j][8 * u + 5]);
        w[3] = E::pack2(p[j][8 * u + 6], p[j][8 * u + 7]);
        pf[j][u] = __builtin_bit_cast(v8_t, w);
      }
    if constexpr (!MFMA_SUM) l_run += psum;
    // O^T += V^T P^T : key slot (half, jj) <-> key 16u + 4 half + jj (jj<4) | 16u + 8 + 4 half + jj-4
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const v8_t vf = *reinterpret_cast<const v8_t*>(vs + (dt * 32 + qi) * C::VS + (2 * j + u) * 32 + half * 16);
          oacc[dt] = E::mfma32(vf, pf[j][u], oacc[dt]);
        }
  };

  const int ntiles = (nk + KB - 1) / KB;
#define PP_RUN_TILE(T_)                                                                          \
  do {                                                                                           \
    const int t0_ = (T_) * KB;                                                                   \
    const char* ks_ = smem + ((T_) & 1) * BUF;                                                   \
    if (t0_ + KB / 2 >= nk) tile(ks_, ks_ + C::KBYTES, t0_, std::integral_constant<int, 2>{});   \
    else if (t0_ + KB > nk) tile(ks_, ks_ + C::KBYTES, t0_, std::integral_constant<int, 1>{});   \
    else tile(ks_, ks_ + C::KBYTES, t0_, std::integral_constant<int, 0>{});                      \
  } while (0)
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int rep = 0; rep < reps; ++rep) {
  if (rep == 0) {
    for (int t = 0; t < ntiles; ++t) {
      const bool more = t + 1 < ntiles;
      if (more) load_tile(t * KB + KB);                   // global -> registers, in flight during the MFMAs below
      PP_RUN_TILE(t);
      if (more) store_tile((t + 1) & 1);                  // the other buffer: last read two barriers ago
      __syncthreads();
    }
  } else {                                                // (host: qrep > 1 only with ntiles <= 2 -- both tiles are in LDS)
    q0 += QW * NW;
    PP_LOAD_Q();
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    m_run = -1.0e30f;
    l_run = 0.f;
    for (int t = 0; t < ntiles; ++t) PP_RUN_TILE(t);
  }

  // ---- epilogue: lane = query qi; rows (dcols) = dt*32 + (r&3) + 8 (r>>2) + 4 half
  float l_tot;
  if constexpr (MFMA_SUM) {
    // output row D of O^T: tile D/32, row D%32 = 8*(r>>2) + (r&3) + 4*half  ->  held by the half-0 lanes
    constexpr int LR = D % 32;
    static_assert((LR & 7) < 4, "row D must live in lane half 0");
    const float l0 = oacc[D / 32][4 * (LR >> 3) + (LR & 3)];
    l_tot = __shfl(l0, qi, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + qi;
  if (qrow < nq) {
    uint16_t* op = o + ((size_t)b * nq + qrow) * ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = dt * 32 + 8 * g + 4 * half;
        if (dc < D) {
          u32x2_t w;
          w[0] = E::pack2(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv);
          w[1] = E::pack2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
          *reinterpret_cast<u32x2_t*>(op + dc) = w;
        }
      }
  }
  }   // rep
#undef PP_LOAD_Q
#undef PP_RUN_TILE
}

// [rows = b*nk + t][cols] (row stride ld) -> vt[b][col][t] (row stride ldvt), pad columns t in [nk, ldvt) zeroed.
__global__ void __launch_bounds__(256) transpose_v_kernel(const uint16_t* __restrict__ v, int ld, int nk, int cols,
                                                         uint16_t* __restrict__ vt, int ldvt) {
  __shared__ uint16_t tile[64][66];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (t < nk && c < cols) ? v[((size_t)b * nk + t) * ld + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, t = t0 + tx;
    if (c < cols && t < ldvt) vt[((size_t)b * cols + c) * ldvt + t] = tile[tx][r];
  }
}

}  // namespace

template <int D, int EDT>
static int launch_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                       int batch, int heads, int nq, int nk, float sl2, hipStream_t st) {
  constexpr int LDS = 2 * (Cfg<D>::KBYTES + Cfg<D>::VBYTES);
  if (LDS > 64 * 1024 &&
      pp_func_lds(reinterpret_cast<const void*>(attn_fwd_kernel<D, EDT, false>), LDS, "hipFuncSetAttribute(attention)") != PP_OK)
    return PP_ERR_LAUNCH;
  static_assert(D > 40 || LDS <= 64 * 1024, "the d = 40 kernels run with the default dynamic LDS limit");
  // K / V reuse over query blocks (see the kernel): only where the keys fit the two LDS buffers and the launch would
  // otherwise be >= 4 rounds of workgroups (two 4-wave workgroups per CU).  PP_ATTN_QREP=1|2 forces it (A/B, tests).
  static const int qrep_env = pp_lab_env("PP_ATTN_QREP", 0);
  const long long wgs = (long long)((nq + QW * NW - 1) / (QW * NW)) * heads * batch;
  int qrep = (D == 40 && nk <= 2 * KB && wgs >= 2048) ? 2 : 1;
  if (D == 40 && (qrep_env == 1 || qrep_env == 2)) qrep = (nk <= 2 * KB) ? qrep_env : 1;
  const dim3 grid((nq + QW * NW * qrep - 1) / (QW * NW * qrep), heads, batch), block(256);
  if constexpr (D == 40) {
    if (qrep > 1) {
      hipLaunchKernelGGL((attn_fwd_kernel<D, EDT, true>), grid, block, LDS, st, (const uint16_t*)q, ldq, (const uint16_t*)k,
                         ldk, (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2, qrep);
      PP_CHECK_LAUNCH("attn_fwd_kernel");
      return PP_OK;
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<D, EDT, false>), grid, block, LDS, st, (const uint16_t*)q, ldq, (const uint16_t*)k,
                     ldk, (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2, 1);
  PP_CHECK_LAUNCH("attn_fwd_kernel");
  return PP_OK;
}

// attention_pipe.hip: software-pipelined kernel for the hot self-attention shapes (PP_ERR_UNSUPPORTED otherwise)
int pp_attention_pipe_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                             int batch, int heads, int nq, int nk, int d, float sl2, int dtype, int variant,
                             hipStream_t st);

extern "C" int pp_attention_fwd_variant(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o,
                                        int ldo, int batch, int heads, int nq, int nk, int d, float scale, int dtype,
                                        int variant, void* stream) {
  if (!q || !k || !vt || !o || batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || ldvt < nk) return PP_ERR_BAD_ARG;
  if (variant < PP_ATTN_AUTO || variant > PP_ATTN_PIPE_LOG2) return PP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  bool use_pipe = variant != PP_ATTN_PHASED;
  if (variant == PP_ATTN_AUTO && pp_lab_env("PP_ATTN_PIPE", 1) == 0) use_pipe = false;   // (lab: A/B against the phased kernel)
  if (use_pipe) {
    const int rc = pp_attention_pipe_launch(q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, d, sl2, dtype, variant, st);
    if (rc != PP_ERR_UNSUPPORTED || variant != PP_ATTN_AUTO) return rc;
  }
  switch (d) {
    case 40: PP_DT_SWITCH(dtype, return (launch_attn<40, EDT>(q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, sl2, st)));
    case 80: PP_DT_SWITCH(dtype, return (launch_attn<80, EDT>(q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, sl2, st)));
    case 160: PP_DT_SWITCH(dtype, return (launch_attn<160, EDT>(q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, sl2, st)));
    default: return PP_ERR_UNSUPPORTED;
  }
}

// shapes of the software-pipelined kernels = shapes PP_ATTN_PIPE_LOG2 covers (a producer asks before it pre-multiplies Q)
bool pp_attention_pipe_keys_ok(int nk);      // attention_pipe.hip: the launcher's own shape predicate
extern "C" int pp_attention_log2_ok(int nq, int nk, int d) { return (nq > 0 && d == 40 && pp_attention_pipe_keys_ok(nk)) ? 1 : 0; }

extern "C" int pp_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o,
                                int ldo, int batch, int heads, int nq, int nk, int d, float scale, int dtype,
                                void* stream) {
  return pp_attention_fwd_variant(q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, d, scale, dtype, PP_ATTN_AUTO,
                                  stream);
}

extern "C" int pp_transpose_v(const void* v, int ld, int batch, int nk, int cols, void* vt, int ldvt, void* stream) {
  if (!v || !vt || batch <= 0 || nk <= 0 || cols <= 0 || ldvt < nk) return PP_ERR_BAD_ARG;
  const dim3 grid((ldvt + 63) / 64, (cols + 63) / 64, batch);
  hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)v, ld, nk, cols,
                     (uint16_t*)vt, ldvt);
  PP_CHECK_LAUNCH("transpose_v_kernel");
  return PP_OK;
}
