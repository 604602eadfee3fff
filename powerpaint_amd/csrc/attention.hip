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
//   * K tile rows are padded to an odd number of 16-B slots, V^T rows to an odd number of 8-B slots: the
//     ds_read_b128 (K) and ds_read_b64 (V^T) fragment reads are bank-conflict free;
//   * next K/V tile is prefetched into registers while the current one is consumed (global latency hidden behind
//     the MFMAs), exp2 with the softmax scale folded into one FMA.
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
  static constexpr int VS = KB * 2 + 8;           // V^T tile row stride in bytes (odd number of 8-B slots)
  static constexpr int KBYTES = KB * KS;
  static constexpr int VROWS = DT * 32;
  static constexpr int VBYTES = VROWS * VS;
  static constexpr int KPIECES = KB * (DP / 8);   // 16-B pieces per K tile
  static constexpr int VPIECES = D * (KB / 8);    // 16-B pieces per V^T tile (rows < D only)
  static constexpr int KPT = (KPIECES + 255) / 256;
  static constexpr int VPT = (VPIECES + 255) / 256;
};

template <int D>
__global__ void __launch_bounds__(256, 1)
attn_fwd_kernel(const uint16_t* __restrict__ q, int ldq, const uint16_t* __restrict__ k, int ldk,
                const uint16_t* __restrict__ vt, int ldvt, uint16_t* __restrict__ o, int ldo, int heads, int nq, int nk,
                float scale_log2e) {
  using C = Cfg<D>;
  __shared__ __attribute__((aligned(16))) char smem[C::KBYTES + C::VBYTES];
  char* ks = smem;
  char* vs = smem + C::KBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (QW * NW) + wave * QW;
  const int qi = lane & 31, half = lane >> 5;

  // zero the V^T rows >= D once (they only feed output rows that are never stored, but must stay finite)
  for (int i = tid; i < (C::VROWS - D) * (C::VS / 8); i += 256)
    *reinterpret_cast<u32x2_t*>(vs + D * C::VS + i * 8) = u32x2_t{0u, 0u};

  // ---- Q fragments (B operand): lane = query qi, k-slot = 8*half + jj  ->  Q[q0+qi][16 s + 8 half + jj]
  bf16x8_t qf[C::DS];
  {
    const int qrow = q0 + qi;
    const uint16_t* qp = q + ((size_t)b * nq + (qrow < nq ? qrow : 0)) * ldq + h * D;
#pragma unroll
    for (int s = 0; s < C::DS; ++s) {
      const int dc = 16 * s + 8 * half;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (qrow < nq && dc < D) v = *reinterpret_cast<const u32x4_t*>(qp + dc);
      qf[s] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(k + (size_t)b * nk * ldk, (uint32_t)nk * (uint32_t)ldk * 2u);
  const uint16_t* vt_bh = vt + ((size_t)b * heads + h) * D * (size_t)ldvt;
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vt_bh, (uint32_t)D * (uint32_t)ldvt * 2u);

  u32x4_t kreg[C::KPT], vreg[C::VPT];
  auto load_tile = [&](int t0) {
#pragma unroll
    for (int i = 0; i < C::KPT; ++i) {
      const int pc = tid + i * 256;
      const int row = pc / (C::DP / 8), sl = pc - row * (C::DP / 8);
      const int key = t0 + row;
      const bool ok = pc < C::KPIECES && key < nk && sl * 8 < D;
      const uint32_t off = ok ? (uint32_t)(key * ldk + h * D + sl * 8) * 2u : PP_OOB;
      kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int pc = tid + i * 256;
      const int row = pc >> 3, sl = pc & 7;
      const int key = t0 + sl * 8;
      const bool ok = pc < C::VPIECES && key < nk;
      const uint32_t off = ok ? (uint32_t)(row * ldvt + key) * 2u : PP_OOB;
      vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, 0, 0);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < C::KPT; ++i) {
      const int pc = tid + i * 256;
      if (pc < C::KPIECES) {
        const int row = pc / (C::DP / 8), sl = pc - row * (C::DP / 8);
        *reinterpret_cast<u32x4_t*>(ks + row * C::KS + sl * 16) = kreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int pc = tid + i * 256;
      if (pc < C::VPIECES) {
        const int row = pc >> 3, sl = pc & 7;
        char* dst = vs + row * C::VS + sl * 16;   // 8-B aligned only (VS = 136)
        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{vreg[i][0], vreg[i][1]};
        *reinterpret_cast<u32x2_t*>(dst + 8) = u32x2_t{vreg[i][2], vreg[i][3]};
      }
    }
  };

  f32x16_t oacc[C::DT];
#pragma unroll
  for (int t = 0; t < C::DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int ntiles = (nk + KB - 1) / KB;
  load_tile(0);
  store_tile();
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int t0 = t * KB;
    if (t + 1 < ntiles) load_tile(t0 + KB);

    // ---- S^T = K Q^T : two 32-key sub-tiles
    f32x16_t sacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::DS; ++s) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + (32 * j + qi) * C::KS + (16 * s + 8 * half) * 2);
        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[j], 0, 0, 0);
      }
    }
    // ---- mask keys >= nk (last tile only), tile max
    float tmax = -1.0e30f;
    const bool tail = (t0 + KB > nk);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (tail) {
          const int key = t0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= nk) sacc[j][r] = -1.0e30f;
        }
        tmax = fmaxf(tmax, sacc[j][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f((m_run - m_new) * scale_log2e);
    const float mc = m_new * scale_log2e;
    m_run = m_new;
    float psum = 0.f;
    bf16x8_t pf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = exp2f(fmaf(sacc[j][r], scale_log2e, -mc));
        psum += p[r];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t w;
        w[0] = pack2bf(p[8 * u + 0], p[8 * u + 1]);
        w[1] = pack2bf(p[8 * u + 2], p[8 * u + 3]);
        w[2] = pack2bf(p[8 * u + 4], p[8 * u + 5]);
        w[3] = pack2bf(p[8 * u + 6], p[8 * u + 7]);
        pf[j][u] = __builtin_bit_cast(bf16x8_t, w);
      }
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

    // ---- O^T += V^T P^T : key slot (half, jj) <-> key 16u + 4 half + jj (jj<4) | 16u + 8 + 4 half + jj-4
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int koff = (32 * j + 16 * u + 4 * half) * 2;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const char* vp = vs + (dt * 32 + qi) * C::VS + koff;
          const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(vp);
          const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(vp + 16);
          const u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), pf[j][u], oacc[dt], 0, 0, 0);
        }
      }

    __syncthreads();                       // everyone done reading this tile
    if (t + 1 < ntiles) {
      store_tile();
      __syncthreads();                     // next tile visible
    }
  }

  // ---- epilogue: lane = query qi; rows (dcols) = dt*32 + (r&3) + 8 (r>>2) + 4 half
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
          w[0] = pack2bf(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv);
          w[1] = pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
          *reinterpret_cast<u32x2_t*>(op + dc) = w;
        }
      }
  }
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

extern "C" int pp_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o,
                                int ldo, int batch, int heads, int nq, int nk, int d, float scale, void* stream) {
  if (!q || !k || !vt || !o || batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0) return PP_ERR_BAD_ARG;
  if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || ldvt < nk) return PP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((nq + QW * NW - 1) / (QW * NW), heads, batch), block(256);
  const float sl2 = scale * 1.4426950408889634f;
  switch (d) {
    case 40:
      hipLaunchKernelGGL(attn_fwd_kernel<40>, grid, block, 0, st, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk,
                         (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2);
      break;
    case 80:
      hipLaunchKernelGGL(attn_fwd_kernel<80>, grid, block, 0, st, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk,
                         (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2);
      break;
    case 160:
      hipLaunchKernelGGL(attn_fwd_kernel<160>, grid, block, 0, st, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk,
                         (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2);
      break;
    default:
      return PP_ERR_UNSUPPORTED;
  }
  PP_CHECK_LAUNCH("attn_fwd_kernel");
  return PP_OK;
}

extern "C" int pp_transpose_v(const void* v, int ld, int batch, int nk, int cols, void* vt, int ldvt, void* stream) {
  if (!v || !vt || batch <= 0 || nk <= 0 || cols <= 0 || ldvt < nk) return PP_ERR_BAD_ARG;
  const dim3 grid((ldvt + 63) / 64, (cols + 63) / 64, batch);
  hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)v, ld, nk, cols,
                     (uint16_t*)vt, ldvt);
  PP_CHECK_LAUNCH("transpose_v_kernel");
  return PP_OK;
}
