// GroupNorm(32)(+SiLU) and LayerNorm on NHWC / [rows][C] bf16 -- HBM-bound wavefront-reduction kernels.
//
// GroupNorm on NHWC: the C/32 channels of a group are contiguous per pixel but strided across pixels, so a
// one-block-per-group layout would read 20..160-byte fragments.  Instead every thread owns ONE fixed 16-byte channel
// slot (8 channels) and walks pixels: all global reads are full-line coalesced, per-channel partial sums stay in
// registers, and only the tiny per-block fold touches LDS.  Two launches:
//   stats : [batch][chunk] partial (sum, sumsq) per group -> workspace                (reads x once)
//   apply : every block first folds the chunk partials (fp64, fixed order) into per-channel scale / shift in LDS,
//           then y = act(x*scale + shift)                                             (reads x once, writes y once)
// Two launches per norm; the fold order is fixed, so results are bit-reproducible run to run (no float atomics).
#include "pp_common.h"

namespace {

constexpr int GN_MAX_T = 512;

// grid (nchunk, batch); block = S*P threads, S = C/8 slots, P pixel lanes
template <int EDT>
__global__ void __launch_bounds__(GN_MAX_T) gn_stats_kernel(const uint16_t* __restrict__ x1, int c1,
                                                           const uint16_t* __restrict__ x2, int c2, int hw, int groups,
                                                           int S, int P, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [P][C][2]
  const int C = c1 + c2;
  const int tid = threadIdx.x;
  const int slot = tid % S, pl = tid / S;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int per = (hw + nchunk - 1) / nchunk;
  const int p0 = chunk * per;
  int p1 = p0 + per;
  if (p1 > hw) p1 = hw;
  const int c = slot * 8;
  const uint16_t* src;
  int cs, cl;
  if (c < c1) { src = x1; cs = c1; cl = c; } else { src = x2; cs = c2; cl = c - c1; }
  src += (size_t)b * hw * cs + cl;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  auto acc8 = [&](const u32x4_t v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = E16<EDT>::lo(v[j]), hi = E16<EDT>::hi(v[j]);
      s[2 * j] += lo; q[2 * j] += lo * lo;
      s[2 * j + 1] += hi; q[2 * j + 1] += hi * hi;
    }
  };
  int p = p0 + pl;
  for (; p + 3 * P < p1; p += 4 * P) {      // four independent 16-B loads in flight per lane
    const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(src + (size_t)p * cs);
    const u32x4_t v1 = *reinterpret_cast<const u32x4_t*>(src + (size_t)(p + P) * cs);
    const u32x4_t v2 = *reinterpret_cast<const u32x4_t*>(src + (size_t)(p + 2 * P) * cs);
    const u32x4_t v3 = *reinterpret_cast<const u32x4_t*>(src + (size_t)(p + 3 * P) * cs);
    acc8(v0); acc8(v1); acc8(v2); acc8(v3);
  }
  for (; p < p1; p += P) acc8(*reinterpret_cast<const u32x4_t*>(src + (size_t)p * cs));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[(pl * C + c + j) * 2 + 0] = s[j];
    red[(pl * C + c + j) * 2 + 1] = q[j];
  }
  __syncthreads();
  if (tid < groups) {
    const int cg = C / groups;
    float ss = 0.f, qq = 0.f;
    for (int l = 0; l < P; ++l)
      for (int j = 0; j < cg; ++j) {
        ss += red[(l * C + tid * cg + j) * 2 + 0];
        qq += red[(l * C + tid * cg + j) * 2 + 1];
      }
    float* o = partial + (((size_t)b * nchunk + chunk) * groups + tid) * 2;
    o[0] = ss;
    o[1] = qq;
  }
}

// Fold of the per-chunk partials (fixed order => deterministic) into per-channel scale / shift, executed in the
// prologue of EVERY apply block (a few KB of L2-resident reads) instead of a separate launch.
//   ss_out (optional, block 0 of each batch item): [batch][2][C] for callers that want the affine form.
PP_DEVINL void gn_fold(const float* __restrict__ partial, int nchunk, int groups, int C, int hw, float eps,
                       const float* __restrict__ gamma, const float* __restrict__ beta, int b, float* mean_s,
                       float* rstd_s, float* sc_s, float* sh_s) {
  const int tid = threadIdx.x;
  // 8 lanes per group, each folds a strided subset of the chunks; combined in fixed lane order
  const int g = tid >> 3, sub = tid & 7;
  if (g < groups) {
    double s = 0.0, q = 0.0;
    for (int k = sub; k < nchunk; k += 8) {
      const float* o = partial + (((size_t)b * nchunk + k) * groups + g) * 2;
      s += (double)o[0];
      q += (double)o[1];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      s += __shfl_down(s, off, 8);
      q += __shfl_down(q, off, 8);
    }
    if (sub == 0) {
      const double n = (double)hw * (double)(C / groups);
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      mean_s[g] = (float)mean;
      rstd_s[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int cg = C / groups;
  for (int c = tid; c < C; c += blockDim.x) {
    const int gg = c / cg;
    const float sc = rstd_s[gg] * gamma[c];
    sc_s[c] = sc;
    sh_s[c] = beta[c] - mean_s[gg] * sc;
  }
  __syncthreads();
}

// Same tables from the fixed-point accumulators the producers' epilogues filled (PPGemmArgs.gn_acc): no partials to fold.
PP_DEVINL void gn_fold_acc(const long long* __restrict__ acc, int groups, int C, int hw, float eps,
                           const float* __restrict__ gamma, const float* __restrict__ beta, int b, float* mean_s,
                           float* rstd_s, float* sc_s, float* sh_s) {
  const int tid = threadIdx.x;
  // gamma / beta do not depend on the statistics: fetch them first (<= 8 channels per thread, C <= 2048) so that the
  // accumulator round trip and the fp64 arithmetic are the only serial part
  constexpr int PRE = 8;
  float gpre[PRE], bpre[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int c = tid + k * 256;
    gpre[k] = (c < C) ? gamma[c] : 0.f;
    bpre[k] = (c < C) ? beta[c] : 0.f;
  }
  if (tid < groups) {
    const double s = (double)acc[((size_t)b * groups + tid) * 2] * (1.0 / (double)PP_GN_SUM_SCALE);
    const double q = (double)acc[((size_t)b * groups + tid) * 2 + 1] * (1.0 / (double)PP_GN_SQ_SCALE);
    const double n = (double)hw * (double)(C / groups);
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cg = C / groups;
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int c = tid + k * 256;
    if (c < C) {
      const int gg = c / cg;
      const float sc = rstd_s[gg] * gpre[k];
      sc_s[c] = sc;
      sh_s[c] = bpre[k] - mean_s[gg] * sc;
    }
  }
  for (int c = tid + PRE * 256; c < C; c += 256) {
    const int gg = c / cg;
    const float sc = rstd_s[gg] * gamma[c];
    sc_s[c] = sc;
    sh_s[c] = beta[c] - mean_s[gg] * sc;
  }
  __syncthreads();
}

// grid (blocks, batch): flat over 16-B pieces of one batch item.  ACC: statistics come from the int64 accumulators
// (`partial` then points at them) instead of the chunk partials of gn_stats_kernel.
template <bool SILU, bool ACC, int EDT>
__global__ void __launch_bounds__(256) gn_apply_kernel(const uint16_t* __restrict__ x1, int c1,
                                                      const uint16_t* __restrict__ x2, int c2, int hw,
                                                      const float* __restrict__ partial, int nchunk, int groups,
                                                      float eps, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, uint16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float tab[];   // [2][C] scale | shift, then 2 x 64 group stats
  const int C = c1 + c2, S = C >> 3;
  const int b = blockIdx.y;
  float* sc = tab;
  float* sh = tab + C;
  auto fold = [&]() {
    if (ACC) gn_fold_acc(reinterpret_cast<const long long*>(partial), groups, C, hw, eps, gamma, beta, b, tab + 2 * C,
                         tab + 2 * C + 64, sc, sh);
    else gn_fold(partial, nchunk, groups, C, hw, eps, gamma, beta, b, tab + 2 * C, tab + 2 * C + 64, sc, sh);
  };
  auto one = [&](const u32x4_t v, const f32x4_t a0, const f32x4_t a1, const f32x4_t b0, const f32x4_t b1) -> u32x4_t {
    float r[8];
    r[0] = E16<EDT>::lo(v[0]) * a0[0] + b0[0]; r[1] = E16<EDT>::hi(v[0]) * a0[1] + b0[1];
    r[2] = E16<EDT>::lo(v[1]) * a0[2] + b0[2]; r[3] = E16<EDT>::hi(v[1]) * a0[3] + b0[3];
    r[4] = E16<EDT>::lo(v[2]) * a1[0] + b1[0]; r[5] = E16<EDT>::hi(v[2]) * a1[1] + b1[1];
    r[6] = E16<EDT>::lo(v[3]) * a1[2] + b1[2]; r[7] = E16<EDT>::hi(v[3]) * a1[3] + b1[3];
    if (SILU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = silu_fast_f(r[j]);
    }
    u32x4_t o;
    o[0] = E16<EDT>::pack2(r[0], r[1]); o[1] = E16<EDT>::pack2(r[2], r[3]);
    o[2] = E16<EDT>::pack2(r[4], r[5]); o[3] = E16<EDT>::pack2(r[6], r[7]);
    return o;
  };
  if (S <= 256) {
    // A thread keeps ONE 8-channel slot for the whole launch: its scale / shift live in registers, the per-piece index
    // arithmetic is one add, and four independent 16-byte loads are in flight per lane (the old flat loop had one, a
    // division per piece and four LDS reads: 1.9 TB/s on the 21 MB tensors of the 64x64 level).  256 / S pixel rows
    // per block pass; the threads past R * S idle (<= 37 % at C = 1280, none of the hot 64x64-level shapes).
    const int R = 256 / S;
    const int tid = threadIdx.x;
    const bool active = tid < R * S;
    const int r = active ? tid / S : 0, c = active ? (tid - r * S) * 8 : 0;
    const bool first = c < c1;
    const uint16_t* src = first ? x1 + (size_t)b * hw * c1 + c : x2 + (size_t)b * hw * c2 + (c - c1);
    const size_t ss = first ? c1 : c2;
    uint16_t* dst = y + (size_t)b * hw * C + c;
    const int step = gridDim.x * R;
    int p = blockIdx.x * R + r;
    // The first four rows' loads are issued BEFORE the statistics are folded: the fold is two dependent global round
    // trips (accumulators, then gamma / beta) plus fp64 arithmetic, ~4 us during which nothing else was in flight --
    // the launch cost ~13 us whatever the tensor size.
    u32x4_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = u32x4_t{0u, 0u, 0u, 0u};
      if (active && p + k * step < hw) v[k] = *reinterpret_cast<const u32x4_t*>(src + (size_t)(p + k * step) * ss);
    }
    fold();
    if (!active) return;
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(sc + c), a1 = *reinterpret_cast<const f32x4_t*>(sc + c + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sh + c), b1 = *reinterpret_cast<const f32x4_t*>(sh + c + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (p + k * step < hw) *reinterpret_cast<u32x4_t*>(dst + (size_t)(p + k * step) * C) = one(v[k], a0, a1, b0, b1);
    for (p += 4 * step; p < hw; p += step)
      *reinterpret_cast<u32x4_t*>(dst + (size_t)p * C) =
          one(*reinterpret_cast<const u32x4_t*>(src + (size_t)p * ss), a0, a1, b0, b1);
    return;
  }
  fold();
  const int total = hw * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int p = i / S;
    const int c = (i - p * S) * 8;
    const uint16_t* src = (c < c1) ? x1 + ((size_t)b * hw + p) * c1 + c : x2 + ((size_t)b * hw + p) * c2 + (c - c1);
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src);
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(sc + c), a1 = *reinterpret_cast<const f32x4_t*>(sc + c + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sh + c), b1 = *reinterpret_cast<const f32x4_t*>(sh + c + 4);
    *reinterpret_cast<u32x4_t*>(y + ((size_t)b * hw + p) * C + c) = one(v, a0, a1, b0, b1);
  }
}

// conv_norm_out + SiLU + conv_out (C -> 4 channels) in ONE launch -- the tail of UNet2DConditionModel.forward
// (/root/reference/powerpaint/models/unet_2d_condition.py:1351-1354).  The unfused pair writes the normalised 64x64x320
// activation (21 MB) and re-reads it nine times through L1 / L2 (gn_apply 12 us + conv3x3_cout4_mfma 40 us per step).
// Here a workgroup owns an 8 x 8 output patch: it normalises the 10 x 10 input patch on the fly (statistics from the
// producers' accumulators, as gn_apply_kernel<.., ACC>; rounded to the 16-bit format exactly where the unfused path
// stores it) and runs it ONCE through the matrix cores against all 36 (channel, tap) weight rows,
//     z[pixel][co * 9 + tap] = sum_c W[co][tap][c] * act[pixel][c]          (7 groups of 16 pixels x 3 row blocks),
// then gathers  out[co][y][x] = bias[co] + sum_tap z[(y + ky, x + kx)][co * 9 + tap]  from LDS.  Input bytes per patch:
// 100 / 64 of the activation instead of 9x.
constexpr int GC_ZLD = 37;                               // z row stride (floats): 36 + 1 against bank conflicts
template <int EDT>
__global__ void __launch_bounds__(256) gn_conv3x3_cout4_kernel(const uint16_t* __restrict__ x, int h, int w_, int C,
                                                              const long long* __restrict__ acc, int groups, float eps,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const uint16_t* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out) {
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  extern __shared__ __attribute__((aligned(16))) float tab[];   // scale[C] | shift[C] | 2 x 64 group stats | W36 | z
  const int WLD = 2 * C + 16;                            // weight row stride in bytes (+16: sixteen rows, sixteen bank slots)
  float* sc = tab;
  float* sh = tab + C;
  char* wl = reinterpret_cast<char*>(tab + 2 * C + 128);
  float* z = reinterpret_cast<float*>(wl + 48 * WLD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int tiles_x = (w_ + 7) >> 3;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int b = blockIdx.y;
  const int steps = C >> 5;                              // 32-channel MFMA steps (host: C % 32 == 0, C <= 384: LDS < 64 KB)

  // the 36 weight rows ([co][tap][C] is already row = co * 9 + tap), rows 36 .. 47 zero
  const int pcs = C >> 3;
  for (int i = tid; i < 48 * pcs; i += 256) {
    const int row = i / pcs, pc = i - row * pcs;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < 36) v = *reinterpret_cast<const u32x4_t*>(w + (size_t)row * C + pc * 8);
    *reinterpret_cast<u32x4_t*>(wl + row * WLD + pc * 16) = v;
  }
  gn_fold_acc(acc, groups, C, h * w_, eps, gamma, beta, b, tab + 2 * C, tab + 2 * C + 64, sc, sh);   // (syncs inside)
  __syncthreads();

  for (int pg = wave; pg < 7; pg += 4) {
    const int q = pg * 16 + r16;                         // pixel of the 10 x 10 patch (>= 100: padding of the last group)
    const int py = q / 10, px = q - py * 10;
    const int iy = ty * 8 + py - 1, ix = tx * 8 + px - 1;
    const bool ok = q < 100 && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w_;
    const uint16_t* xp = x + (((size_t)b * h + (ok ? iy : 0)) * w_ + (ok ? ix : 0)) * C + kg * 8;
    f32x4_t a3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    constexpr int PF = 6;                                // loads in flight per lane
    u32x4_t xv[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) xv[s] = (s < steps) ? *reinterpret_cast<const u32x4_t*>(xp + s * 32) : u32x4_t{0u, 0u, 0u, 0u};
    for (int s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = s0 + u;
        if (s >= steps) break;
        const u32x4_t v = xv[u];
        if (s + PF < steps) xv[u] = *reinterpret_cast<const u32x4_t*>(xp + (s + PF) * 32);
        const int c = s * 32 + kg * 8;
        const f32x4_t s0v = *reinterpret_cast<const f32x4_t*>(sc + c), s1v = *reinterpret_cast<const f32x4_t*>(sc + c + 4);
        const f32x4_t h0v = *reinterpret_cast<const f32x4_t*>(sh + c), h1v = *reinterpret_cast<const f32x4_t*>(sh + c + 4);
        float r[8];
        r[0] = E::lo(v[0]) * s0v[0] + h0v[0]; r[1] = E::hi(v[0]) * s0v[1] + h0v[1];
        r[2] = E::lo(v[1]) * s0v[2] + h0v[2]; r[3] = E::hi(v[1]) * s0v[3] + h0v[3];
        r[4] = E::lo(v[2]) * s1v[0] + h1v[0]; r[5] = E::hi(v[2]) * s1v[1] + h1v[1];
        r[6] = E::lo(v[3]) * s1v[2] + h1v[2]; r[7] = E::hi(v[3]) * s1v[3] + h1v[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = silu_fast_f(r[j]);
        u32x4_t o;
        o[0] = E::pack2(r[0], r[1]); o[1] = E::pack2(r[2], r[3]);
        o[2] = E::pack2(r[4], r[5]); o[3] = E::pack2(r[6], r[7]);
        if (!ok) o = u32x4_t{0u, 0u, 0u, 0u};            // zero padding applies to the NORMALISED activation
        const v8_t bfr = __builtin_bit_cast(v8_t, o);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
          const v8_t af = *reinterpret_cast<const v8_t*>(wl + (rb * 16 + r16) * WLD + s * 64 + kg * 16);
          a3[rb] = E::mfma16(af, bfr, a3[rb]);
        }
      }
    }
    // D[weight row 16 rb + 4 kg + i][pixel r16]
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb * 16 + kg * 4 + i;
        if (row < 36) z[q * GC_ZLD + row] = a3[rb][i];
      }
  }
  __syncthreads();
  {
    const int co = tid >> 6, p = tid & 63, oy = p >> 3, ox = p & 7;
    const int gy = ty * 8 + oy, gx = tx * 8 + ox;
    float sum = bias ? bias[co] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      sum += z[((oy + ky) * 10 + ox + kx) * GC_ZLD + co * 9 + tap];
    }
    if (gy < h && gx < w_) out[(((size_t)b * 4 + co) * h + gy) * w_ + gx] = sum;
  }
}

// LayerNorm: one wave per row, <= 4 16-B pieces per lane (C <= 2048), two-pass statistics in registers.
template <int NP, int EDT>
__global__ void __launch_bounds__(256) layernorm_kernel(const uint16_t* __restrict__ x, int rows, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, uint16_t* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int S = C >> 3;
  float v[NP][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = lane + i * 64;
    if (pc < S) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(x + (size_t)row * C + pc * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][2 * j] = E16<EDT>::lo(u[j]); v[i][2 * j + 1] = E16<EDT>::hi(u[j]); }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[i][j];
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = lane + i * 64;
    if (pc < S) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = lane + i * 64;
    if (pc < S) {
      const int c = pc * 8;
      const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gamma + c), g1 = *reinterpret_cast<const f32x4_t*>(gamma + c + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(beta + c), b1 = *reinterpret_cast<const f32x4_t*>(beta + c + 4);
      float r[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = (v[i][j] - mean) * rstd * g0[j] + b0[j];
        r[4 + j] = (v[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      u32x4_t o;
      o[0] = E16<EDT>::pack2(r[0], r[1]); o[1] = E16<EDT>::pack2(r[2], r[3]); o[2] = E16<EDT>::pack2(r[4], r[5]); o[3] = E16<EDT>::pack2(r[6], r[7]);
      *reinterpret_cast<u32x4_t*>(y + (size_t)row * C + c) = o;
    }
  }
}

}  // namespace

// workgroups per batch item of the apply kernel: with the fixed-slot mapping (C <= 2048) one pass of four pixel rows per
// thread -- every load of the launch is in flight at once; else >= 8 pieces per thread (amortises the fold prologue)
static int gn_apply_blocks(int hw, int C, long long total) {
  const int S = C / 8;
  int nb = S <= 256 ? (hw + (256 / S) * 4 - 1) / ((256 / S) * 4) : (int)((total + 256 * 8 - 1) / (256 * 8));
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return nb;
}

static int gn_nchunk(int hw) {
  int c = hw / 16;
  if (c < 1) c = 1;
  if (c > 128) c = 128;
  return c;
}

extern "C" size_t pp_groupnorm_workspace_bytes(int batch, int hw, int C) {
  (void)C;
  (void)hw;
  return (size_t)batch * 128 * 64 * 2 * sizeof(float);  // [batch][<=128 chunks][<=64 groups][2]
}

extern "C" int pp_groupnorm_stats(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups,
                                  float* workspace, int dtype, void* stream) {
  const int C = c1 + c2;
  if (!x1 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  if (!workspace) return PP_ERR_WORKSPACE;
  if (c1 <= 0 || c1 % 8 || c2 % 8 || (c2 > 0 && !x2) || groups <= 0 || groups > 32 || C % groups) return PP_ERR_BAD_ARG;
  const int S = C / 8;
  if (S > GN_MAX_T) return PP_ERR_UNSUPPORTED;
  int P = GN_MAX_T / S;
  if (P > 8) P = 8;
  const size_t lds = (size_t)P * C * 2 * sizeof(float);
  if (lds > 64 * 1024) return PP_ERR_UNSUPPORTED;
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL(gn_stats_kernel<EDT>, dim3(gn_nchunk(hw), batch), dim3(S * P), lds,
                                         (hipStream_t)stream, (const uint16_t*)x1, c1, (const uint16_t*)x2, c2, hw, groups,
                                         S, P, workspace));
  PP_CHECK_LAUNCH("gn_stats_kernel");
  return PP_OK;
}

extern "C" int pp_groupnorm_apply(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups,
                                  float eps, const float* gamma, const float* beta, const float* workspace, int silu,
                                  void* y, int dtype, void* stream) {
  if (!x1 || !workspace || !gamma || !beta || !y || c1 <= 0 || c1 % 8 || c2 % 8 || (c2 > 0 && !x2) || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  const int C = c1 + c2;
  if (groups <= 0 || groups > 32 || C % groups) return PP_ERR_BAD_ARG;
  const long long total = (long long)hw * (C / 8);
  int nb = gn_apply_blocks(hw, C, total);
  const size_t lds = (size_t)(2 * C + 128) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = gn_nchunk(hw);
  if (silu)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((gn_apply_kernel<true, false, EDT>), dim3(nb, batch), dim3(256), lds, st,
                                           (const uint16_t*)x1, c1, (const uint16_t*)x2, c2, hw, workspace, nchunk, groups, eps,
                                           gamma, beta, (uint16_t*)y));
  else
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((gn_apply_kernel<false, false, EDT>), dim3(nb, batch), dim3(256), lds, st,
                                           (const uint16_t*)x1, c1, (const uint16_t*)x2, c2, hw, workspace, nchunk, groups, eps,
                                           gamma, beta, (uint16_t*)y));
  PP_CHECK_LAUNCH("gn_apply_kernel");
  return PP_OK;
}

extern "C" int pp_groupnorm_apply_acc(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups,
                                      float eps, const float* gamma, const float* beta, const int64_t* acc, int silu,
                                      void* y, int dtype, void* stream) {
  if (!x1 || !acc || !gamma || !beta || !y || c1 <= 0 || c1 % 8 || c2 % 8 || (c2 > 0 && !x2) || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  const int C = c1 + c2;
  if (groups <= 0 || groups > 32 || C % groups) return PP_ERR_BAD_ARG;
  const long long total = (long long)hw * (C / 8);
  int nb = gn_apply_blocks(hw, C, total);
  const size_t lds = (size_t)(2 * C + 128) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const float* accf = reinterpret_cast<const float*>(acc);
  if (silu)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((gn_apply_kernel<true, true, EDT>), dim3(nb, batch), dim3(256), lds, st,
                                           (const uint16_t*)x1, c1, (const uint16_t*)x2, c2, hw, accf, 0, groups, eps,
                                           gamma, beta, (uint16_t*)y));
  else
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((gn_apply_kernel<false, true, EDT>), dim3(nb, batch), dim3(256), lds, st,
                                           (const uint16_t*)x1, c1, (const uint16_t*)x2, c2, hw, accf, 0, groups, eps,
                                           gamma, beta, (uint16_t*)y));
  PP_CHECK_LAUNCH("gn_apply_kernel(acc)");
  return PP_OK;
}

extern "C" int pp_gn_conv3x3_smallcout_supported(int cin, int cout, int groups) {
  return (cout == 4 && cin > 0 && cin % 32 == 0 && cin <= 384 && groups > 0 && groups <= 32 && cin % groups == 0) ? 1 : 0;
}

extern "C" int pp_gn_conv3x3_smallcout(const void* x, int batch, int h, int w_, int cin, int groups, float eps,
                                       const float* gamma, const float* beta, const int64_t* acc, const void* w,
                                       const float* bias, int cout, float* out_nchw, int dtype, void* stream) {
  if (!x || !gamma || !beta || !acc || !w || !out_nchw || batch <= 0 || h <= 0 || w_ <= 0 || !pp_dt_ok(dtype))
    return PP_ERR_BAD_ARG;
  if (!pp_gn_conv3x3_smallcout_supported(cin, cout, groups)) return PP_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(2 * cin + 128) * sizeof(float) + 48 * (size_t)(2 * cin + 16) + 112 * GC_ZLD * sizeof(float);
  const dim3 grid(((w_ + 7) / 8) * ((h + 7) / 8), batch);
  PP_DT_SWITCH(dtype, hipLaunchKernelGGL((gn_conv3x3_cout4_kernel<EDT>), grid, dim3(256), lds, (hipStream_t)stream,
                                         (const uint16_t*)x, h, w_, cin, reinterpret_cast<const long long*>(acc), groups,
                                         eps, gamma, beta, (const uint16_t*)w, bias, out_nchw));
  PP_CHECK_LAUNCH("gn_conv3x3_cout4_kernel");
  return PP_OK;
}

extern "C" int pp_layernorm(const void* x, int rows, int C, const float* gamma, const float* beta, float eps, void* y,
                            int dtype, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || C % 8 || !pp_dt_ok(dtype)) return PP_ERR_BAD_ARG;
  const int S = C / 8;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((rows + 3) / 4), block(256);
  if (S <= 64)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<1, EDT>), grid, block, 0, st, (const uint16_t*)x, rows, C, gamma,
                                           beta, eps, (uint16_t*)y));
  else if (S <= 128)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<2, EDT>), grid, block, 0, st, (const uint16_t*)x, rows, C, gamma,
                                           beta, eps, (uint16_t*)y));
  else if (S <= 192)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<3, EDT>), grid, block, 0, st, (const uint16_t*)x, rows, C, gamma,
                                           beta, eps, (uint16_t*)y));
  else if (S <= 256)
    PP_DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<4, EDT>), grid, block, 0, st, (const uint16_t*)x, rows, C, gamma,
                                           beta, eps, (uint16_t*)y));
  else
    return PP_ERR_UNSUPPORTED;
  PP_CHECK_LAUNCH("layernorm_kernel");
  return PP_OK;
}
