// HBM-/latency-bound helpers of the denoising step: timestep embedding, skinny linears, direct 3x3 convs for
// MFMA-unfriendly channel counts, layout conversion at the NCHW module boundary, the fused CFG + scheduler step and the
// bit-exact mask preparation.  All are coalesced / wave-reduction kernels; none allocates or synchronises.
#include "pp_common.h"

namespace {

// ------------------------------------------------------------------------------------------ timestep embedding
__global__ void timestep_embedding_kernel(const float* __restrict__ t_dev, int rows, int dim, float* __restrict__ out) {
  const int half = dim >> 1;
  const float t = t_dev[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * half; i += gridDim.x * blockDim.x) {
    const int r = i / half, k = i - r * half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float a = t * f;
    out[(size_t)r * dim + k] = cosf(a);          // flip_sin_to_cos=True -> [cos | sin]
    out[(size_t)r * dim + half + k] = sinf(a);
  }
}

// ------------------------------------------------------------------------------------------ skinny linear (GEMV-like)
template <int R, int EDT>
__global__ void __launch_bounds__(256) linear_skinny_kernel(const float* __restrict__ x, int rows, int K,
                                                           const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                           int N, float* __restrict__ out, int ldo, int act_in, int act_out) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [R][K]
  for (int i = threadIdx.x; i < R * K; i += 256) {
    const int r = i / K;
    float v = (r < rows) ? x[(size_t)r * K + (i - r * K)] : 0.f;
    if (act_in == PP_ACT_SILU) v = silu_f(v);
    xs[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = K >> 3;
  for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int pc = lane; pc < S; pc += 64) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(w + (size_t)n * K + pc * 8);
      float wv[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[2 * j] = E16<EDT>::lo(u[j]); wv[2 * j + 1] = E16<EDT>::hi(u[j]); }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float* xp = xs + r * K + pc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r] = fmaf(wv[j], xp[j], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s = wave_sum(acc[r]);
      if (lane == 0 && r < rows) {
        float v = s + (bias ? bias[n] : 0.f);
        if (act_out == PP_ACT_SILU) v = silu_f(v);
        out[(size_t)r * ldo + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ direct 3x3 conv, cout % 8 == 0
// thread = (pixel, 8 consecutive output channels); weights [3][3][cin][cout] so the 8 weights of a tap are one 16-B load
template <int EDT>
__global__ void __launch_bounds__(256) conv3x3_direct_kernel(const uint16_t* __restrict__ x, int batch, int hin, int win,
                                                            int cin, const uint16_t* __restrict__ w,
                                                            const float* __restrict__ bias, int cout, int stride,
                                                            int hout, int wout, int silu_out,
                                                            const uint16_t* __restrict__ add, uint16_t* __restrict__ out) {
  const int S = cout >> 3;
  const long long total = (long long)batch * hout * wout * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int sl = (int)(i % S);
    const long long pix = i / S;
    const int ox = (int)(pix % wout);
    const int oy = (int)((pix / wout) % hout);
    const int b = (int)(pix / ((long long)wout * hout));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[sl * 8 + j] : 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride + ky - 1;
      if ((unsigned)iy >= (unsigned)hin) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride + kx - 1;
        if ((unsigned)ix >= (unsigned)win) continue;
        const uint16_t* xp = x + (((size_t)b * hin + iy) * win + ix) * cin;
        const uint16_t* wp = w + ((size_t)(ky * 3 + kx) * cin) * cout + sl * 8;
        for (int ci = 0; ci < cin; ++ci) {
          const float xv = E16<EDT>::to_f(xp[ci]);
          const u32x4_t u = *reinterpret_cast<const u32x4_t*>(wp + (size_t)ci * cout);
          acc[0] = fmaf(xv, E16<EDT>::lo(u[0]), acc[0]); acc[1] = fmaf(xv, E16<EDT>::hi(u[0]), acc[1]);
          acc[2] = fmaf(xv, E16<EDT>::lo(u[1]), acc[2]); acc[3] = fmaf(xv, E16<EDT>::hi(u[1]), acc[3]);
          acc[4] = fmaf(xv, E16<EDT>::lo(u[2]), acc[4]); acc[5] = fmaf(xv, E16<EDT>::hi(u[2]), acc[5]);
          acc[6] = fmaf(xv, E16<EDT>::lo(u[3]), acc[6]); acc[7] = fmaf(xv, E16<EDT>::hi(u[3]), acc[7]);
        }
      }
    }
    const size_t o = (size_t)pix * cout + sl * 8;
    if (add) {
      const u32x4_t r = *reinterpret_cast<const u32x4_t*>(add + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += E16<EDT>::lo(r[j]); acc[2 * j + 1] += E16<EDT>::hi(r[j]); }
    }
    if (silu_out) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = silu_f(acc[j]);
    }
    u32x4_t ov;
    ov[0] = E16<EDT>::pack2(acc[0], acc[1]); ov[1] = E16<EDT>::pack2(acc[2], acc[3]);
    ov[2] = E16<EDT>::pack2(acc[4], acc[5]); ov[3] = E16<EDT>::pack2(acc[6], acc[7]);
    *reinterpret_cast<u32x4_t*>(out + o) = ov;
  }
}

// ------------------------------------------------------------------------------------------ 3x3 conv with tiny cout (conv_out)
// one wave per output pixel; lanes split K = 9*cin (cin % 8 == 0); weights [cout][3][3][cin]; out fp32 NCHW
template <int CO, int EDT>
__global__ void __launch_bounds__(256) conv3x3_smallcout_kernel(const uint16_t* __restrict__ x, int batch, int h, int w_,
                                                               int cin, const uint16_t* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long npix = (long long)batch * h * w_;
  if (pix >= npix) return;
  const int ox = (int)(pix % w_), oy = (int)((pix / w_) % h), b = (int)(pix / ((long long)w_ * h));
  const int S = cin >> 3;   // 16-B pieces per tap
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  for (int pc = lane; pc < 9 * S; pc += 64) {
    const int tap = pc / S, sl = pc - tap * S;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy + ky - 1, ix = ox + kx - 1;
    if ((unsigned)iy >= (unsigned)h || (unsigned)ix >= (unsigned)w_) continue;
    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(x + (((size_t)b * h + iy) * w_ + ix) * cin + sl * 8);
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + ((size_t)c * 9 + tap) * cin + sl * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[c] = fmaf(E16<EDT>::lo(xv[j]), E16<EDT>::lo(wv[j]), acc[c]);
        acc[c] = fmaf(E16<EDT>::hi(xv[j]), E16<EDT>::hi(wv[j]), acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float s = wave_sum(acc[c]);
    if (lane == 0) out[(((size_t)b * CO + c) * h + oy) * w_ + ox] = s + (bias ? bias[c] : 0.f);
  }
}

// The same conv on the matrix cores (cin % 32 == 0, cin <= 384: the UNet's 320 -> 4 conv_out, the VAE decoder's 128 -> 3(4)):
//   D[cout 16 (4 live)][pixel 16] += W[cout][k 32] . X^T[k 32][pixel 16]      v_mfma_f32_16x16x32
// A workgroup owns 32 consecutive output pixels (two 16-pixel MFMA column groups that share every weight fragment); its
// four waves split every tap's channel range in 32-deep steps (s = wave, wave + 4, ...), each lane fetching the 16
// bytes of its (row, k-group) straight from global memory -- the 23 KB of weights stay in L1 / L2, the activations are
// re-read nine times from L2 -- and the four partial tiles are summed through LDS.  1024 workgroups at 64x64 x 8: one
// round on 256 CUs at four waves per SIMD (16-pixel workgroups needed 8 per CU, 7 fit: the tail round doubled the time).
template <int EDT>
__global__ void __launch_bounds__(256, 4) conv3x3_cout4_mfma_kernel(const uint16_t* __restrict__ x, int batch, int h,
                                                                   int w_, int cin, const uint16_t* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ out) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  constexpr int PG = 2;                                // 16-pixel groups per workgroup
  __shared__ f32x4_t part[4][PG][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kg = lane >> 4;           // B column = pixel / A row = cout; k-group of 8 channels
  const long long npix = (long long)batch * h * w_;
  bool pv[PG];
  int ox[PG], oy[PG];
  size_t img[PG];                                      // element offset of the pixel's image
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const long long pix = (long long)blockIdx.x * (16 * PG) + g * 16 + col;
    pv[g] = pix < npix;
    const long long pc = pv[g] ? pix : npix - 1;
    ox[g] = (int)(pc % w_);
    oy[g] = (int)((pc / w_) % h);
    img[g] = (size_t)(pc / ((long long)w_ * h)) * h * w_;
  }
  const int steps = cin >> 5;                          // <= 12 (host-checked): at most three 32-deep steps per wave
  f32x4_t acc[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Addresses are clamped to something readable and the value is masked instead (no divergent loads, no branches), and
  // the nine taps are software-pipelined by hand: tap t + 1's nine loads are in flight under tap t's six MFMAs.
  uint32_t wm[3];
  int sc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int st = wave + 4 * j;
    sc[j] = (st < steps ? st : 0) * 32;
    wm[j] = (col < 4 && st < steps) ? 0xffffffffu : 0u;
  }
  u32x4_t xv[2][3][PG], wv[2][3];
  uint32_t xm[2][PG];
  auto load_tap = [&](int tap, int buf) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const uint16_t* wp = w + ((size_t)(col & 3) * 9 + tap) * cin + kg * 8;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      const int iy = oy[g] + ky - 1, ix = ox[g] + kx - 1;
      const bool ok = pv[g] && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w_;
      xm[buf][g] = ok ? 0xffffffffu : 0u;
      const uint16_t* xp = x + (img[g] + (size_t)(ok ? iy : oy[g]) * w_ + (ok ? ix : ox[g])) * cin + kg * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j) xv[buf][j][g] = *reinterpret_cast<const u32x4_t*>(xp + sc[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) wv[buf][j] = *reinterpret_cast<const u32x4_t*>(wp + sc[j]);
  };
  load_tap(0, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int buf = tap & 1;
    if (tap + 1 < 9) load_tap(tap + 1, buf ^ 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const u32x4_t wj = wv[buf][j] & wm[j];
#pragma unroll
      for (int g = 0; g < PG; ++g)
        acc[g] = E::mfma16(__builtin_bit_cast(v8_t, wj), __builtin_bit_cast(v8_t, xv[buf][j][g] & xm[buf][g]), acc[g]);
    }
  }
  if (lane < 16) {                                     // lanes 0-15: couts 0-3 (registers) of pixel `lane` of each group
#pragma unroll
    for (int g = 0; g < PG; ++g) part[wave][g][lane] = acc[g];
  }
  __syncthreads();
  if (threadIdx.x < 64 * PG) {                         // thread = (group, cout, pixel): 64-byte runs per output plane
    const int g = threadIdx.x >> 6, r = (threadIdx.x >> 4) & 3, p = threadIdx.x & 15;
    const long long q = (long long)blockIdx.x * (16 * PG) + g * 16 + p;
    if (q < npix) {
      const int qx = (int)(q % w_), qy = (int)((q / w_) % h), qb = (int)(q / ((long long)w_ * h));
      const float sum = (part[0][g][p][r] + part[1][g][p][r]) + (part[2][g][p][r] + part[3][g][p][r]);
      out[(((size_t)qb * 4 + r) * h + qy) * w_ + qx] = sum + (bias ? bias[r] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------ layout
PP_DEVINL float load_as_f32(const void* p, int dtype, size_t i) {
  if (dtype == 0) return ((const float*)p)[i];
  if (dtype == 1) return bf2f(((const uint16_t*)p)[i]);
  return (float)(((const _Float16*)p)[i]);
}

template <int EDT>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const void* __restrict__ src, int dtype, int batch, int c,
                                                          int hw, int bmod, uint16_t* __restrict__ dst, int ldc, int c0) {
  const long long total = (long long)batch * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    const int sb = bmod > 0 ? b % bmod : b;
    for (int j = 0; j < c; ++j)
      dst[(size_t)i * ldc + c0 + j] = E16<EDT>::from_f(load_as_f32(src, dtype, ((size_t)sb * c + j) * hw + p));
  }
}

template <int EDT>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const uint16_t* __restrict__ src, int batch, int c, int hw,
                                                          void* __restrict__ dst, int dtype) {
  const long long total = (long long)batch * c * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const int ch = (int)((i / hw) % c);
    const int b = (int)(i / ((long long)hw * c));
    const uint16_t v = src[((size_t)b * hw + p) * c + ch];
    if (dtype == 0) ((float*)dst)[i] = E16<EDT>::to_f(v);
    else ((uint16_t*)dst)[i] = v;
  }
}

// ------------------------------------------------------------------------------------------ elementwise add (bf16)
template <int EDT>
__global__ void __launch_bounds__(256) add_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                      uint16_t* __restrict__ out, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4_t x = *reinterpret_cast<const u32x4_t*>(a + i * 8);
    const u32x4_t y = *reinterpret_cast<const u32x4_t*>(b + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = E16<EDT>::pack2(E16<EDT>::lo(x[j]) + E16<EDT>::lo(y[j]), E16<EDT>::hi(x[j]) + E16<EDT>::hi(y[j]));
    *reinterpret_cast<u32x4_t*>(out + i * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------ CFG + scheduler step
PP_DEVINL void cfg_sched_step_body(const float* __restrict__ eps2, int cfg, float g, float* __restrict__ x,
                                   float* __restrict__ m_prev, int n, int kind, const float* __restrict__ coef,
                                   const int32_t* step_dev) {
  if (kind == 2) {
    // PNDM / PLMS: table row (16 floats) = w0..w3 (linear multistep weights over eps, h1, h2, h3), a (sample coefficient),
    // b (model-output coefficient), then as floats: ring slots of h1, h2, h3, the slot this eps is pushed to (-1: not
    // stored), use_saved (transfer from the saved sample), save (keep the incoming sample).  state = [5][n]: 4 history
    // slots + the saved sample (the first transfer is redone with the average of the first two predictions).
    const float* c = coef + (size_t)step_dev[0] * 16;
    const float w0 = c[0], w1 = c[1], w2 = c[2], w3 = c[3], ca = c[4], cb = c[5];
    const float* h1 = m_prev + (size_t)(int)c[6] * n;
    const float* h2 = m_prev + (size_t)(int)c[7] * n;
    const float* h3 = m_prev + (size_t)(int)c[8] * n;
    const int push = (int)c[9];
    const bool use_saved = c[10] != 0.f, save = c[11] != 0.f;
    float* hp = push >= 0 ? m_prev + (size_t)push * n : nullptr;
    float* saved = m_prev + (size_t)4 * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
      float e;
      if (cfg) {
        const float eu = eps2[i], ec = eps2[n + i];
        e = eu + g * (ec - eu);
      } else {
        e = eps2[i];
      }
      const float mo = w0 * e + w1 * h1[i] + w2 * h2[i] + w3 * h3[i];
      const float xv = x[i];
      const float src = use_saved ? saved[i] : xv;
      if (save) saved[i] = xv;
      if (hp) hp[i] = e;
      x[i] = ca * src + cb * mo;
    }
    return;
  }
  if (kind == 3) {
    // UniPC (predict_x0, order <= 3): row (16 floats) = sigma_t, alpha_t, use_corr, corrector weights over
    // (last, m1, m2, m3, x0), predictor weights over (xc, x0, m1, m2); state = [4][n]: last corrected sample and the
    // three most recent x0 predictions.  All weights come from the host's float64 solve of the UniPC conditions.
    const float* c = coef + (size_t)step_dev[0] * 16;
    const float sg = c[0], al = c[1];
    const bool corr = c[2] != 0.f;
    const float k_last = c[3], k_m1 = c[4], k_m2 = c[5], k_m3 = c[6], k_x0 = c[7];
    const float p_xc = c[8], p_x0 = c[9], p_m1 = c[10], p_m2 = c[11];
    float* last = m_prev;
    float* m1 = m_prev + (size_t)n;
    float* m2 = m_prev + (size_t)2 * n;
    float* m3 = m_prev + (size_t)3 * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
      float e;
      if (cfg) {
        const float eu = eps2[i], ec = eps2[n + i];
        e = eu + g * (ec - eu);
      } else {
        e = eps2[i];
      }
      const float xv = x[i];
      const float x0 = (xv - sg * e) / al;
      const float a1 = m1[i], a2 = m2[i];
      float xc = xv;
      if (corr) xc = k_last * last[i] + k_m1 * a1 + k_m2 * a2 + k_m3 * m3[i] + k_x0 * x0;
      x[i] = p_xc * xc + p_x0 * x0 + p_m1 * a1 + p_m2 * a2;
      last[i] = xc;
      m3[i] = a2;
      m2[i] = a1;
      m1[i] = x0;
    }
    return;
  }
  const float* c = coef + (size_t)step_dev[0] * 8;
  const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float e;
    if (cfg) {
      const float eu = eps2[i], ec = eps2[n + i];
      e = eu + g * (ec - eu);                      // pipeline_PowerPaint.py:1019-1020
    } else {
      e = eps2[i];
    }
    const float xv = x[i];
    if (kind == 0) {
      // DDIM: c0 = sqrt(1-a_t), c1 = sqrt(a_t), c2 = sqrt(a_prev), c3 = sqrt(1-a_prev-std_dev_t^2)  (c4 = std_dev_t = 0
      // at eta = 0; the variance noise of eta > 0 is added by pp_ddim_variance_noise)
      const float x0 = (xv - c0 * e) / c1;
      x[i] = c2 * x0 + c3 * e;
    } else {
      // DPM-Solver++(2M): c0 = sigma_t(cur), c1 = alpha_t(cur), c2 = sigma_next/sigma_cur, c3 = alpha_next*(exp(-h)-1),
      //                   c4 = 0.5*c3 (0 on first-order steps), c5 = 1/r0
      const float x0 = (xv - c0 * e) / c1;
      const float d1 = c5 * (x0 - m_prev[i]);
      x[i] = c2 * xv - c3 * x0 - c4 * d1;
      m_prev[i] = x0;
    }
  }
}

// ticket != NULL (ABI v20): the launch is the last reader of the step counter in its step and moves it on itself -- the
// block that takes the last ticket does `step[0] += 1` (every other block has read the counter before it drew its ticket;
// the ticket returns to zero for the next step).  One launch less per denoise step than pp_step_advance behind it.
__global__ void __launch_bounds__(256) cfg_sched_step_kernel(const float* __restrict__ eps2, int cfg, float g,
                                                            float* __restrict__ x, float* __restrict__ m_prev, int n,
                                                            int kind, const float* __restrict__ coef, int32_t* step_dev,
                                                            unsigned* ticket) {
  cfg_sched_step_body(eps2, cfg, g, x, m_prev, n, kind, coef, step_dev);
  if (ticket) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();      // this block's reads of the counter are performed before its ticket becomes visible
      if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        *ticket = 0u;
        step_dev[0] += 1;
      }
    }
  }
}

// stochastic DDIM (eta > 0): x += std_dev_t * z, std_dev_t = column 4 of the step's table row, z drawn by the host
__global__ void __launch_bounds__(256) ddim_variance_noise_kernel(float* __restrict__ x, const float* __restrict__ z, int n,
                                                                 const float* __restrict__ coef,
                                                                 const int32_t* __restrict__ step_dev) {
  const float sd = coef[(size_t)step_dev[0] * 8 + 4];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = x[i] + sd * z[i];
}

// ppt-v1 with a 4-channel UNet (pipeline_PowerPaint.py:1025-1039): after the scheduler step the known region is put back,
//   latents = (1 - m) * (a * x0 + b * noise) + m * latents,   (a, b) = row `step` of the re-noise table
// (sqrt(abar), sqrt(1 - abar)) of the NEXT timestep, (1, 0) on the last step.  x0 [chw] and m [hw] are the first image's
// latents / mask, broadcast over the batch as `image_latents[:1]` / `mask[:1]` are.
__global__ void __launch_bounds__(256) latent_blend_kernel(float* __restrict__ x, const float* __restrict__ x0,
                                                          const float* __restrict__ m, const float* __restrict__ z,
                                                          const float* __restrict__ tab, const int32_t* __restrict__ step_dev,
                                                          int chw, int hw, int n) {
  const int r = step_dev[0];
  const float a = tab[2 * r], b = tab[2 * r + 1];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int e = i % chw;
    const float mk = m[e % hw];
    const float proper = b == 0.f ? x0[e] : a * x0[e] + b * z[i];
    x[i] = (1.f - mk) * proper + mk * x[i];
  }
}

__global__ void step_select_t_kernel(const float* __restrict__ ts, const int32_t* __restrict__ step, float* __restrict__ t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = ts[step[0]];
}
__global__ void step_advance_kernel(int32_t* __restrict__ step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

// ------------------------------------------------------------------------------------------ bit-exact mask prep
__global__ void __launch_bounds__(256) mask_prep_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, int batch, int c, int h, int w, int ho,
                                                       int wo) {
  const long long hw = (long long)h * w;
  long long total;
  if (mode == 0) total = (long long)batch * c * hw;
  else if (mode == 1) total = (long long)batch * c * hw;
  else if (mode == 2) total = (long long)batch * ho * wo;
  else total = (long long)batch * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    if (mode == 0) {
      out[i] = a[i] >= 0.5f ? 1.0f : 0.0f;                 // mask[mask<0.5]=0; mask[mask>=0.5]=1
    } else if (mode == 1) {
      const long long bi = i / (c * hw), p = i % hw;
      out[i] = b[bi * hw + p] < 0.5f ? a[i] : a[i] * 0.0f;  // image * (mask < 0.5)  (keeps -0.0 / NaN semantics)
    } else if (mode == 2) {
      const int x = (int)(i % wo), y = (int)((i / wo) % ho);
      const long long bi = i / ((long long)wo * ho);
      // torch nearest: src = floor(dst * (in/out)) with float scale
      int sy = (int)floorf((float)y * ((float)h / (float)ho));
      int sx = (int)floorf((float)x * ((float)w / (float)wo));
      if (sy > h - 1) sy = h - 1;
      if (sx > w - 1) sx = w - 1;
      out[i] = a[bi * hw + (long long)sy * w + sx];
    } else {
      const long long bi = i / hw, p = i % hw;
      float s = 0.f;
      for (int j = 0; j < c; ++j) s += a[(bi * c + j) * hw + p];   // sequential fp32 sum over channels like .sum(1)
      out[i] = s < 0.f ? 1.0f : 0.0f;
    }
  }
}

}  // namespace

static int grid_for_host(long long total) {
  long long nb = (total + 255) / 256;
  return (int)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
}

extern "C" int pp_timestep_embedding(const float* t_dev, int rows, int dim, float* out, void* stream) {
  if (!t_dev || !out || rows <= 0 || dim <= 0 || dim % 2) return PP_ERR_BAD_ARG;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((rows * dim / 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     t_dev, rows, dim, out);
  PP_CHECK_LAUNCH("timestep_embedding_kernel");
  return PP_OK;
}

extern "C" int pp_linear_skinny(const float* x, int rows, int K, const void* w, const float* bias, int N, float* out,
                                int ldo, int act_in, int act_out, int dtype, void* stream) {
  if (!x || !w || !out || rows <= 0 || rows > 16 || K <= 0 || K % 8 || N <= 0 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nb = (N + 3) / 4;
  if (nb > 2048) nb = 2048;
#define PP_SKINNY(R, E)                                                                                           \
  do {                                                                                                            \
    const size_t lds = (size_t)(R) * K * 4;                                                                       \
    if (lds > 160 * 1024) return PP_ERR_UNSUPPORTED;                                                              \
    if (lds > 64 * 1024 && pp_func_lds(reinterpret_cast<const void*>(linear_skinny_kernel<R, E>), (int)lds,       \
                                       "hipFuncSetAttribute(linear_skinny)") != PP_OK)                            \
      return PP_ERR_LAUNCH;                                                                                       \
    hipLaunchKernelGGL((linear_skinny_kernel<R, E>), dim3(nb), dim3(256), lds, st, x, rows, K, (const uint16_t*)w, \
                       bias, N, out, ldo, act_in, act_out);                                                       \
  } while (0)
#define PP_SKINNY_R(E)            \
  do {                            \
    if (rows == 1) PP_SKINNY(1, E);    \
    else if (rows <= 4) PP_SKINNY(4, E); \
    else if (rows <= 8) PP_SKINNY(8, E); \
    else PP_SKINNY(16, E);        \
  } while (0)
  if (dtype == PP_DT_F16) PP_SKINNY_R(PP_DT_F16);
  else PP_SKINNY_R(PP_DT_BF16);
#undef PP_SKINNY_R
#undef PP_SKINNY
  PP_CHECK_LAUNCH("linear_skinny_kernel");
  return PP_OK;
}

extern "C" int pp_conv3x3_direct(const void* x, int batch, int hin, int win, int cin, const void* w, const float* bias,
                                 int cout, int stride, int silu_out, const void* add, void* out, int dtype,
                                 void* stream) {
  if (!x || !w || !out || batch <= 0 || hin <= 0 || win <= 0 || cin <= 0 || cout <= 0 || cout % 8 || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return PP_ERR_BAD_ARG;
  const int hout = (hin + 2 - 3) / stride + 1, wout = (win + 2 - 3) / stride + 1;
  const long long total = (long long)batch * hout * wout * (cout / 8);
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(conv3x3_direct_kernel<EDT>, dim3(grid_for_host(total)), dim3(256), 0,
                                         (hipStream_t)stream, (const uint16_t*)x, batch, hin, win, cin, (const uint16_t*)w,
                                         bias, cout, stride, hout, wout, silu_out, (const uint16_t*)add, (uint16_t*)out));
  PP_CHECK_LAUNCH("conv3x3_direct_kernel");
  return PP_OK;
}

extern "C" int pp_conv3x3_smallcout(const void* x, int batch, int h, int w_, int cin, const void* w, const float* bias,
                                    int cout, float* out_nchw, int dtype, void* stream) {
  if (!x || !w || !out_nchw || batch <= 0 || h <= 0 || w_ <= 0 || cin <= 0 || cin % 8 || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  if (cout != 4) return PP_ERR_UNSUPPORTED;
  const long long npix = (long long)batch * h * w_;
  if (cin % 32 == 0 && cin <= 384) {
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((conv3x3_cout4_mfma_kernel<EDT>), dim3((unsigned)((npix + 31) / 32)), dim3(256), 0,
                                           (hipStream_t)stream, (const uint16_t*)x, batch, h, w_, cin, (const uint16_t*)w,
                                           bias, out_nchw));
    PP_CHECK_LAUNCH("conv3x3_cout4_mfma_kernel");
    return PP_OK;
  }
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL((conv3x3_smallcout_kernel<4, EDT>), dim3((unsigned)((npix + 3) / 4)), dim3(256), 0,
                                         (hipStream_t)stream, (const uint16_t*)x, batch, h, w_, cin, (const uint16_t*)w,
                                         bias, out_nchw));
  PP_CHECK_LAUNCH("conv3x3_smallcout_kernel");
  return PP_OK;
}

extern "C" int pp_nchw_to_nhwc(const void* src, int src_dtype, int batch, int c, int hw, int src_batch_mod, void* dst,
                               int ldc, int c0, int dtype, void* stream) {
  if (!src || !dst || batch <= 0 || c <= 0 || hw <= 0 || src_dtype < 0 || src_dtype > 2 || c0 < 0 || c0 + c > ldc ||
      !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<EDT>, dim3(grid_for_host((long long)batch * hw)), dim3(256), 0,
                                         (hipStream_t)stream, src, src_dtype, batch, c, hw, src_batch_mod, (uint16_t*)dst,
                                         ldc, c0));
  PP_CHECK_LAUNCH("nchw_to_nhwc_kernel");
  return PP_OK;
}

extern "C" int pp_nhwc_to_nchw(const void* src, int batch, int c, int hw, void* dst, int dst_dtype, int dtype,
                               void* stream) {
  // dst_dtype: fp32, or the source's own 16-bit format (a plain re-layout)
  if (!src || !dst || batch <= 0 || c <= 0 || hw <= 0 || !pp_dt_ok(dtype) || (dst_dtype != 0 && dst_dtype != dtype))
    return PP_ERR_BAD_ARG;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<EDT>, dim3(grid_for_host((long long)batch * c * hw)), dim3(256),
                                         0, (hipStream_t)stream, (const uint16_t*)src, batch, c, hw, dst, dst_dtype));
  PP_CHECK_LAUNCH("nhwc_to_nchw_kernel");
  return PP_OK;
}

extern "C" int pp_add_bf16(const void* a, const void* b, void* out, long long n, int dtype, void* stream) {
  if (!a || !b || !out || n <= 0 || n % 8 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(add_bf16_kernel<EDT>, dim3(grid_for_host(n / 8)), dim3(256), 0, (hipStream_t)stream,
                                         (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out, n / 8));
  PP_CHECK_LAUNCH("add_bf16_kernel");
  return PP_OK;
}

extern "C" int pp_cfg_sched_step(const float* eps2, int cfg, float guidance, float* latents, float* m_prev, int n,
                                 int kind, const float* coef_table, int32_t* step_dev, uint32_t* advance_ticket,
                                 void* stream) {
  if (!eps2 || !latents || !coef_table || !step_dev || n <= 0 || kind < 0 || kind > 3) return PP_ERR_BAD_ARG;
  if (kind >= 1 && !m_prev) return PP_ERR_BAD_ARG;
  hipLaunchKernelGGL(cfg_sched_step_kernel, dim3(grid_for_host(n)), dim3(256), 0, (hipStream_t)stream, eps2, cfg,
                     guidance, latents, m_prev, n, kind, coef_table, step_dev, (unsigned*)advance_ticket);
  PP_CHECK_LAUNCH("cfg_sched_step_kernel");
  return PP_OK;
}

extern "C" int pp_ddim_variance_noise(float* latents, const float* noise, int n, const float* coef_table,
                                      const int32_t* step_dev, void* stream) {
  if (!latents || !noise || !coef_table || !step_dev || n <= 0) return PP_ERR_BAD_ARG;
  hipLaunchKernelGGL(ddim_variance_noise_kernel, dim3(grid_for_host(n)), dim3(256), 0, (hipStream_t)stream, latents, noise,
                     n, coef_table, step_dev);
  PP_CHECK_LAUNCH("ddim_variance_noise_kernel");
  return PP_OK;
}

extern "C" int pp_latent_blend(float* latents, const float* image_latents, const float* mask, const float* noise,
                               const float* renoise_table, const int32_t* step_dev, int batch, int channels, int hw,
                               void* stream) {
  if (!latents || !image_latents || !mask || !noise || !renoise_table || !step_dev) return PP_ERR_BAD_ARG;
  if (batch <= 0 || channels <= 0 || hw <= 0 || (long long)batch * channels * hw > 0x7fffffffLL) return PP_ERR_BAD_ARG;
  const int n = batch * channels * hw;
  hipLaunchKernelGGL(latent_blend_kernel, dim3(grid_for_host(n)), dim3(256), 0, (hipStream_t)stream, latents, image_latents,
                     mask, noise, renoise_table, step_dev, channels * hw, hw, n);
  PP_CHECK_LAUNCH("latent_blend_kernel");
  return PP_OK;
}

extern "C" int pp_step_select_t(const float* timesteps, const int32_t* step_dev, float* t_out, void* stream) {
  if (!timesteps || !step_dev || !t_out) return PP_ERR_BAD_ARG;
  hipLaunchKernelGGL(step_select_t_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, timesteps, step_dev, t_out);
  PP_CHECK_LAUNCH("step_select_t_kernel");
  return PP_OK;
}

extern "C" int pp_step_advance(int32_t* step_dev, void* stream) {
  if (!step_dev) return PP_ERR_BAD_ARG;
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev);
  PP_CHECK_LAUNCH("step_advance_kernel");
  return PP_OK;
}

extern "C" int pp_mask_prep(int mode, const float* a, const float* b, float* out, int batch, int c, int h, int w,
                            int ho, int wo, void* stream) {
  if (!a || !out || mode < 0 || mode > 3 || batch <= 0 || c <= 0 || h <= 0 || w <= 0) return PP_ERR_BAD_ARG;
  if (mode == 1 && !b) return PP_ERR_BAD_ARG;
  if (mode == 2 && (ho <= 0 || wo <= 0)) return PP_ERR_BAD_ARG;
  long long total = mode == 2 ? (long long)batch * ho * wo : mode == 3 ? (long long)batch * h * w : (long long)batch * c * h * w;
  hipLaunchKernelGGL(mask_prep_kernel, dim3(grid_for_host(total)), dim3(256), 0, (hipStream_t)stream, mode, a, b, out,
                     batch, c, h, w, ho, wo);
  PP_CHECK_LAUNCH("mask_prep_kernel");
  return PP_OK;
}

namespace {
__global__ void __launch_bounds__(256) zero_u64_kernel(unsigned long long* dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = 0ull;
}
}  // namespace

extern "C" int pp_zero_u64(void* dst, long long n, void* stream) {
  if (!dst || n <= 0) return PP_ERR_BAD_ARG;
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(zero_u64_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)dst, n);
  PP_CHECK_LAUNCH("zero_u64_kernel");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ head of a denoising step
// The three bookkeeping launches in front of a network's step plan as one (each was 5-6 us of pure launch floor):
//   blocks [0, nb_t)            temb_out[:] = temb_table[step][:]            (the time-embedding rows of this timestep)
//   blocks [nb_t, nb_t + nb_x)  x_in[b][p][c0 + j] = latents[b mod wrap][j][p]   (fp32 NCHW -> 16-bit NHWC, CFG duplication)
//   the rest                    acc[:] = 0                                    (GroupNorm-statistics accumulators)
namespace {
template <int EDT>
__global__ void __launch_bounds__(256) step_head_kernel(const float* __restrict__ table, const int32_t* __restrict__ step_dev,
                                                       float* __restrict__ temb_out, int row_floats, int nb_t,
                                                       const float* __restrict__ lat, int batch, int c, int hw, int bmod,
                                                       uint16_t* __restrict__ dst, int ldc, int c0, int nb_x,
                                                       unsigned long long* __restrict__ zdst, long long nz) {
  const int bid = blockIdx.x;
  if (bid < nb_t) {
    const float* row = table + (size_t)step_dev[0] * row_floats;
    for (int i = bid * 256 + threadIdx.x; i < row_floats; i += nb_t * 256) temb_out[i] = row[i];
  } else if (bid < nb_t + nb_x) {
    const long long total = (long long)batch * hw;
    for (long long i = (long long)(bid - nb_t) * 256 + threadIdx.x; i < total; i += (long long)nb_x * 256) {
      const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
      const int sb = bmod > 0 ? b % bmod : b;
      for (int j = 0; j < c; ++j) dst[(size_t)i * ldc + c0 + j] = E16<EDT>::from_f(lat[((size_t)sb * c + j) * hw + p]);
    }
  } else {
    const int nb_z = gridDim.x - nb_t - nb_x;
    for (long long i = (long long)(bid - nb_t - nb_x) * 256 + threadIdx.x; i < nz; i += (long long)nb_z * 256) zdst[i] = 0ull;
  }
}
}  // namespace

extern "C" int pp_step_head(const float* temb_table, const int32_t* step_dev, float* temb_out, int row_floats,
                            const float* latents, int batch, int c, int hw, int src_batch_mod, void* x_in, int ldc, int c0,
                            int dtype, void* zero_dst, long long n_zero, void* stream) {
  if (!temb_table || !step_dev || !temb_out || row_floats <= 0 || !latents || !x_in || batch <= 0 || c <= 0 || hw <= 0 ||
      c0 < 0 || c0 + c > ldc || !pp_dt_ok(dtype) || !zero_dst || n_zero <= 0)
    return PP_ERR_BAD_ARG;
  int nb_t = (row_floats + 255) / 256, nb_x = (int)(((long long)batch * hw + 255) / 256), nb_z = (int)((n_zero + 255) / 256);
  if (nb_t > 64) nb_t = 64;
  if (nb_x > 256) nb_x = 256;
  if (nb_z > 256) nb_z = 256;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(step_head_kernel<EDT>, dim3(nb_t + nb_x + nb_z), dim3(256), 0, (hipStream_t)stream,
                                         temb_table, step_dev, temb_out, row_floats, nb_t, latents, batch, c, hw, src_batch_mod,
                                         (uint16_t*)x_in, ldc, c0, nb_x, (unsigned long long*)zero_dst, n_zero));
  PP_CHECK_LAUNCH("step_head_kernel");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ embedding splice
// One workgroup per output row; the row is a plain byte copy from one of two tables, so the kernel is dtype-agnostic
// (fp32 / fp16 / bf16 token embeddings) and bit-exact by construction.
namespace {
template <typename V>
__global__ __launch_bounds__(128) void embed_splice_kernel(const unsigned char* __restrict__ table,
                                                           const unsigned char* __restrict__ ext,
                                                           const int* __restrict__ src_row,
                                                           unsigned char* __restrict__ out, long long row_bytes) {
  const int r = blockIdx.x;
  const int s = src_row[r];
  const unsigned char* src = s >= 0 ? table + (long long)s * row_bytes : ext + (long long)(-s - 1) * row_bytes;
  const V* sv = (const V*)src;
  V* dv = (V*)(out + (long long)r * row_bytes);
  const int nv = (int)(row_bytes / (long long)sizeof(V));
  // (gridDim.y blocks share a row: the denoise loop copies ONE 113 KB time-embedding row per step -- a single 128-thread
  //  block took 18.7 us for it, profiles/r04_step_timeline.txt)
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nv; i += gridDim.y * blockDim.x) dv[i] = sv[i];
}
}  // namespace

extern "C" int pp_embed_splice(const void* table, const void* ext, const int32_t* src_row, void* out, int n_rows,
                               long long row_bytes, void* stream) {
  if (!table || !src_row || !out || n_rows <= 0 || row_bytes <= 0) return PP_ERR_BAD_ARG;
  const unsigned char* t = (const unsigned char*)table;
  const unsigned char* e = (const unsigned char*)(ext ? ext : table);
  unsigned char* o = (unsigned char*)out;
  const unsigned long long al = (unsigned long long)(uintptr_t)t | (unsigned long long)(uintptr_t)e |
                                (unsigned long long)(uintptr_t)o | (unsigned long long)row_bytes;
  hipStream_t st = (hipStream_t)stream;
  // blocks per row: one per 4 KB of the row, as many as keep the whole launch near 256 blocks
  auto per_row = [&](long long elem) -> unsigned {
    long long want = (row_bytes / elem + 255) / 256, cap = 256 / (n_rows < 256 ? n_rows : 256);
    if (want > cap) want = cap;
    return (unsigned)(want < 1 ? 1 : want);
  };
  if ((al & 15) == 0)
    hipLaunchKernelGGL(embed_splice_kernel<uint4>, dim3(n_rows, per_row(16)), dim3(128), 0, st, t, e, src_row, o, row_bytes);
  else if ((al & 3) == 0)
    hipLaunchKernelGGL(embed_splice_kernel<unsigned int>, dim3(n_rows, per_row(4)), dim3(128), 0, st, t, e, src_row, o,
                       row_bytes);
  else
    hipLaunchKernelGGL(embed_splice_kernel<unsigned char>, dim3(n_rows, per_row(1)), dim3(128), 0, st, t, e, src_row, o,
                       row_bytes);
  PP_CHECK_LAUNCH("embed_splice_kernel");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ row softmax
// p[r][:] = softmax(scale * s[r][:]) for fp32 logits (the single-head, head-dim-512 attention of the VAE mid blocks is
// run as two GEMMs around this kernel: its logits are kept in fp32, only the probabilities are rounded to bf16).
// One 256-thread block per row, two passes over the (L2-resident) row: running (max, sum), then normalise + store.
namespace {
template <int EDT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long long lds, int n,
                                                           float scale_log2e, uint16_t* __restrict__ p,
                                                           long long ldp) {
  __shared__ float red_m[4], red_s[4];
  const float* row = s + (long long)blockIdx.x * lds;
  uint16_t* out = p + (long long)blockIdx.x * ldp;
  const int tid = threadIdx.x;
  float m = -INFINITY, sum = 0.f;
  const int n4 = n >> 2;
  for (int i = tid; i < n4; i += 256) {
    const float4 v = reinterpret_cast<const float4*>(row)[i];
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) * scale_log2e;
    if (mx > m) { sum *= exp2f(m - mx); m = mx; }
    sum += exp2f(v.x * scale_log2e - m) + exp2f(v.y * scale_log2e - m) + exp2f(v.z * scale_log2e - m) +
           exp2f(v.w * scale_log2e - m);
  }
  for (int i = (n4 << 2) + tid; i < n; i += 256) {
    const float x = row[i] * scale_log2e;
    if (x > m) { sum *= exp2f(m - x); m = x; }
    sum += exp2f(x - m);
  }
  const float wm = wave_max(m);
  sum *= (m == -INFINITY) ? 0.f : exp2f(m - wm);
  const float ws = wave_sum(sum);
  if ((tid & 63) == 0) { red_m[tid >> 6] = wm; red_s[tid >> 6] = ws; }
  __syncthreads();
  const float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float S = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) S += (red_m[w] == -INFINITY) ? 0.f : red_s[w] * exp2f(red_m[w] - M);
  const float inv = 1.0f / S;
  for (int i = tid; i < n4; i += 256) {
    const float4 v = reinterpret_cast<const float4*>(row)[i];
    uint2 o;
    o.x = E16<EDT>::pack2(exp2f(v.x * scale_log2e - M) * inv, exp2f(v.y * scale_log2e - M) * inv);
    o.y = E16<EDT>::pack2(exp2f(v.z * scale_log2e - M) * inv, exp2f(v.w * scale_log2e - M) * inv);
    reinterpret_cast<uint2*>(out)[i] = o;
  }
  for (int i = (n4 << 2) + tid; i < n; i += 256) out[i] = E16<EDT>::from_f(exp2f(row[i] * scale_log2e - M) * inv);
}
}  // namespace

extern "C" int pp_softmax_rows(const float* s, long long lds, int rows, int n, float scale, void* p, long long ldp,
                               int dtype, void* stream) {
  if (!s || !p || rows <= 0 || n <= 0 || lds < n || ldp < n || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if ((lds & 3) || (ldp & 3) || ((uintptr_t)s & 15) || ((uintptr_t)p & 7)) return PP_ERR_BAD_ARG;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(softmax_rows_kernel<EDT>, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, lds, n,
                                         scale * 1.44269504088896340736f, (uint16_t*)p, ldp));
  PP_CHECK_LAUNCH("softmax_rows_kernel");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ small attention
// Attention for short sequences (CLIP text tower: 77 tokens, 12 heads of 64, causal): one block per (head, batch item),
// K / V of the head staged once in LDS, one wave per query row -- lanes over keys for the logits, lanes over the head
// dim for P V.  ~150 MFLOP per call in total: latency, not throughput, is what matters here.
namespace {
constexpr int AS_D = 64, AS_MAXK = 128, AS_KS = AS_D + 2;   // K rows padded to 66 halves: lane j -> bank (33 j) mod 32

template <int EDT>
__global__ __launch_bounds__(256) void attn_small_kernel(const uint16_t* __restrict__ q, int ldq,
                                                         const uint16_t* __restrict__ k, int ldk,
                                                         const uint16_t* __restrict__ v, int ldv,
                                                         uint16_t* __restrict__ o, int ldo, int nq, int nk, float scale,
                                                         int causal) {
  __shared__ uint16_t Ks[AS_MAXK * AS_KS];
  __shared__ uint16_t Vs[AS_MAXK * AS_D];
  __shared__ float Qs[4][AS_D];
  __shared__ float Ps[4][AS_MAXK];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < nk * (AS_D / 2); i += 256) {          // 2 halves per thread-iteration
    const int r = i / (AS_D / 2), c = (i - r * (AS_D / 2)) * 2;
    const uint32_t kv = *reinterpret_cast<const uint32_t*>(k + ((size_t)b * nk + r) * ldk + h * AS_D + c);
    const uint32_t vv = *reinterpret_cast<const uint32_t*>(v + ((size_t)b * nk + r) * ldv + h * AS_D + c);
    *reinterpret_cast<uint32_t*>(&Ks[r * AS_KS + c]) = kv;
    *reinterpret_cast<uint32_t*>(&Vs[r * AS_D + c]) = vv;
  }
  __syncthreads();
  const int nkb = (nk + 63) >> 6;
  for (int i = wave; i < nq; i += 4) {
    Qs[wave][lane] = E16<EDT>::to_f(q[((size_t)b * nq + i) * ldq + h * AS_D + lane]);
    __builtin_amdgcn_wave_barrier();
    float sv[AS_MAXK / 64];
    float m = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < AS_MAXK / 64; ++jb) {
      sv[jb] = -INFINITY;
      const int j = jb * 64 + lane;
      if (jb < nkb && j < nk && (!causal || j <= i)) {
        float acc = 0.f;
#pragma unroll 16
        for (int c = 0; c < AS_D; ++c) acc += Qs[wave][c] * E16<EDT>::to_f(Ks[j * AS_KS + c]);
        sv[jb] = acc * scale;
      }
      m = fmaxf(m, sv[jb]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int jb = 0; jb < AS_MAXK / 64; ++jb) {
      const float p = (sv[jb] == -INFINITY) ? 0.f : __expf(sv[jb] - m);
      l += p;
      if (jb < nkb) Ps[wave][jb * 64 + lane] = p;
    }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
    const int jend = causal ? (i + 1 < nk ? i + 1 : nk) : nk;
    for (int j = 0; j < jend; ++j) acc += Ps[wave][j] * E16<EDT>::to_f(Vs[j * AS_D + lane]);
    o[((size_t)b * nq + i) * ldo + h * AS_D + lane] = E16<EDT>::from_f(acc / l);
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace

extern "C" int pp_attention_small(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                                  int ldo, int batch, int heads, int nq, int nk, int d, float scale, int causal,
                                  int dtype, void* stream) {
  if (!q || !k || !v || !o || batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if (d != AS_D || nk > AS_MAXK) return PP_ERR_UNSUPPORTED;
  if ((ldk & 1) || (ldv & 1) || ((uintptr_t)k & 3) || ((uintptr_t)v & 3)) return PP_ERR_BAD_ARG;
  if (causal && nq != nk) return PP_ERR_BAD_ARG;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(attn_small_kernel<EDT>, dim3(heads, batch), dim3(256), 0, (hipStream_t)stream,
                                         (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv,
                                         (uint16_t*)o, ldo, nq, nk, scale, causal));
  PP_CHECK_LAUNCH("attn_small_kernel");
  return PP_OK;
}
