/*
 * pp_hip.h -- C ABI of libpp_hip.so: the MI355X (gfx950) kernels of the PowerPaint denoising hot path.
 *
 * The reference (open-mmlab/PowerPaint) has NO FFI / plugin / custom-op interface for this path: it reaches the
 * GPU only through torch/diffusers module calls.  Each entry point below therefore cites the
 * reference *call site / module* whose device work it replaces.  Conventions for every entry:
 *   - `extern "C"`, plain pointers + sizes, no torch types;
 *   - all pointers are DEVICE pointers unless named `host_*`;
 *   - activations are 16-bit (raw uint16 storage; bf16 or fp16, selected per call by the `dtype` argument / field --
 *     where a comment below says "bf16" read "the call's 16-bit format") in NHWC / [rows][channels] layout,
 *     parameters as documented;
 *   - the last argument is a `hipStream_t` (passed as void*; torch.cuda.current_stream().cuda_stream);
 *   - returns PP_OK (0) or a negative PP_ERR_*; never throws, never allocates, never synchronises, keeps no
 *     pointer after return, is re-entrant per stream; workspaces are caller-provided and sized by *_workspace_bytes.
 */
#ifndef PP_HIP_H_
#define PP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_ERR_BAD_ARG (-1)      /* shape / alignment / enum outside the supported set            */
#define PP_ERR_UNSUPPORTED (-2)  /* valid request the kernels do not implement (e.g. head_dim 24) */
#define PP_ERR_LAUNCH (-3)       /* hipLaunchKernel / hipFuncSetAttribute failed                  */
#define PP_ERR_WORKSPACE (-4)    /* workspace pointer null or too small                           */

#define PP_ABI_VERSION 22
/* 16-bit storage format of activations and matrix weights ("dtype" arguments; the same codes pp_nchw_to_nhwc uses for
 * its source): bf16 or fp16 -- the reference's default is fp16 (/root/reference/app.py:548,559).  MFMA accumulation,
 * norm statistics, softmax, biases and latents are fp32 with either. */
#define PP_DT_F32 0
#define PP_DT_BF16 1
#define PP_DT_F16 2
int pp_abi_version(void);
/* Digest (hex) of the sources, headers and extra flags this library was built from -- what a committed profile or bench
 * record names as "the build it was taken from" (stable across rebuilds and checkout paths). */
const char* pp_build_id(void);
/* hipGetLastError() text of the last PP_ERR_LAUNCH on this thread (host pointer, static storage). */
const char* pp_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM 3x3 convolution on bf16 MFMA tiles.
 *   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )
 * Replaces: every nn.Conv2d(3x3) / nn.Conv2d(1x1) / nn.Linear on the path --
 *   ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D.conv, Upsample2D.conv (ctor sites
 *   /root/reference/powerpaint/models/unet_2d_blocks.py:1274-1285,1319-1321,2542), Transformer2DModel.proj_in/out,
 *   Attention.to_q/k/v/to_out, FeedForward GEGLU proj + out (unet_2d_blocks.py:1289-1300), BrushNet zero-convs
 *   (/root/reference/powerpaint/models/BrushNet_CA.py:842-845,861,899-902) incl. `* conditioning_scale` (:930-934),
 *   and the residual adds `hidden_states + *_add_samples.pop(0)` (unet_2d_blocks.py:1388-1398,2629-2638).
 *
 * X operand modes:
 *   PP_X_PLAIN   : X[m][k] row-major; k <  c1 read from x1 (row stride ldx1), k >= c1 from x2 (row stride ldx2)
 *                  -> channel-concat of two tensors without materialising it (up-block skip concat,
 *                  unet_2d_blocks.py:2589,2732).  K = c1 + c2.
 *   PP_X_CONV3X3 : X is the im2col view of an NHWC tensor [batch][hin][win][c1(+c2)]; pad 1; k = (ky*3+kx)*C + c;
 *                  `stride` 1|2 (Downsample2D); `up`=1 fuses a nearest-2x upsample of the input (Upsample2D);
 *                  M = batch*hout*wout, K = 9*(c1+c2).  c1, c2 multiples of 64.
 * W operand: bf16 [N][K] row-major (conv weights pre-permuted to [Cout][ky][kx][Cin] by the host).
 * Epilogue:  v = (acc + bias[n] + rowvec[m / rows_per_batch][n]) * scale + res1[m][n] + res2[m][n]
 *            act = PP_ACT_GEGLU: columns come in interleaved groups of four (h0,h1,g0,g1); out[m][n/2+j] = h_j*gelu_erf(g_j)
 *            (weights/bias pre-interleaved by the host), out has N/2 columns.
 *            out_vt != NULL: columns >= vt_col0 are written TRANSPOSED per batch item to out_vt[b][n - vt_col0][row in batch]
 *            (V^T for attention), the others to `out`.
 */
#define PP_X_PLAIN 0
#define PP_X_CONV3X3 1
#define PP_ACT_NONE 0
#define PP_ACT_GEGLU 1
#define PP_ACT_SILU 2
/* (ABI v20) softmax over every group of 80 consecutive output columns, in the exp2 domain (the weights carry log2 e); a
 * -inf bias entry masks its column.  16-bit output [M][N], N % 80 == 0.  With the folded LayerNorm on the input side and
 * per-item weights (w_batch_stride / vec_batch_stride) this is the FIRST of the two GEMMs the cross-attention sub-block
 * becomes once K and V are folded into its projections (pp_xattn_fold): probabilities = softmax_h(LN2(h) G_h); the second,
 * plain GEMM multiplies them with H = V Wo^T.  No residuals / row vector / statistics / split-K on this launch. */
#define PP_ACT_SOFTMAX80 3

typedef struct PPGemmArgs {
  int32_t M, N, K;
  int32_t x_mode;
  const void* x1;
  const void* x2;
  int32_t c1, c2;
  int32_t ldx1, ldx2;
  int32_t batch, hin, win, hout, wout, stride, up;
  const void* w;
  const float* bias;
  const float* rowvec;
  int32_t ld_rowvec, rows_per_batch;
  const void* res1;
  int32_t ldres1;
  const void* res2;
  int32_t ldres2;
  float scale;
  int32_t act;
  void* out;
  int32_t ldo;
  int32_t out_f32;
  void* out_vt;
  int32_t vt_col0;
  int32_t vt_ld;      /* row stride of out_vt (>= rows_per_batch) */
  int32_t splitk;     /* 0 = auto */
  int32_t tile;       /* 0 = auto; else PP_TILE_* */
  float* workspace;   /* split-K partials, pp_gemm_workspace_bytes() */
  int32_t dbg;        /* 0.  (Phase-ablation switches of the measurement scripts under tools/: honoured by -DPP_LAB builds
                       * only, libpp_hip.so ignores the field.) */
  int32_t dtype;      /* PP_DT_BF16 | PP_DT_F16: format of x*, w, res*, out (unless out_f32), out_vt */
  /* (ABI v19) The two halves of a classifier-free-guidance batch are bit-identical from `torch.cat([latents] * 2)`
   * (/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:990-996) down to the first cross-attention
   * (/root/reference/powerpaint/models/unet_2d_condition.py:1183-1236): the launches in front of it run on ONE half.
   *   out_dup_rows   > 0: every output row m is ALSO stored at row m + out_dup_rows of `out` (the tensor has 2 M rows; its
   *                  consumers behind the prefix -- the up-block skip concat -- see the full batch).  Single-pass 16-bit
   *                  epilogue only (no split-K, GEGLU, V^T, fp32 output, row moments): PP_ERR_UNSUPPORTED otherwise.
   *   res1_wrap_rows > 0: res1 has only res1_wrap_rows rows; output row m adds res1 row (m mod res1_wrap_rows)
   *                  (requires M <= 2 * res1_wrap_rows): the first full-batch launch reads the half-batch residual. */
  int32_t out_dup_rows;
  int32_t res1_wrap_rows;
  /* LayerNorm folded into the GEMM (BasicTransformerBlock.norm1/2/3 -> the Linear that follows):
   *   LN(x) W^T = rstd * (x (gamma.W)^T - mean * colsum) + beta W^T, so the host packs W' = gamma (.) W, passes
   *   ln_colsum[n] = sum_k W'[n][k] and adds beta W^T to the bias; mean / rstd come from per-row moments.
   * Producer side: row_stats_out != NULL makes the epilogue write, for every output row m and N-tile t (160 columns),
   *   (sum, sum of squares) of the stored bf16 values to row_stats_out[(m * tiles_n + t) * 2].
   * Consumer side: ln_stats != NULL (layout above, ln_tiles tiles per row, ln_dim = row length) switches the epilogue to
   *   v = rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n] ... (then residuals / activation as usual). */
  float* row_stats_out;
  const float* ln_stats;
  const float* ln_colsum;
  int32_t ln_tiles;
  int32_t ln_dim;
  float ln_eps;
  /* (ABI v20; the slot of ABI v12-v14's experiment of the same name) > 0: PP_X_PLAIN only -- batch item b = m / rows_per_batch
   * multiplies with the [N][K] matrix at w + b * w_batch_stride elements (rows_per_batch % 64 == 0: 64-row tiles inside
   * one item).  The encoder hidden states are step-invariant, so pp_xattn_fold leaves one G^T / H^T per prompt. */
  int32_t w_batch_stride;
  /* GroupNorm statistics of the OUTPUT, accumulated by the epilogue (the stats launch of the consuming GroupNorm
   * disappears).  Up to two consumers per tensor (a UNet skip tensor feeds the next layer's norm and, concatenated,
   * an up-block norm):  gn_acc[k] -> int64 [batch][gn_groups[k]][2] = (sum, sum of squares) of the stored bf16 values
   * in fixed point (PP_GN_SUM_SCALE / PP_GN_SQ_SCALE), added with 64-bit integer atomics => order-independent,
   * bit-reproducible.  Column n of this GEMM is channel gn_c0[k] + n of the consumer's (possibly concatenated)
   * input, whose groups are gn_cg[k] channels wide.  Requires rows_per_batch % 64 == 0 (set rows_per_batch), a v2
   * tile, no GEGLU / V^T / fp32 output; pp_gemm_gn_stats_ok() tells.  The caller zeroes gn_acc before the launch. */
  int64_t* gn_acc[2];
  int32_t gn_cg[2];
  int32_t gn_c0[2];
  int32_t gn_groups[2];
  /* PP_X_CONV3X3 only: the contraction continues after the nine taps with c3 + c4 channels read at the OUTPUT pixel
   * (a 1x1 convolution of concat(x3, x4), NHWC bf16 at the output resolution) -- ResnetBlock2D.conv_shortcut(input)
   * summed into conv2:  K = 9 (c1 + c2) + c3 + c4, weight columns in that order.  Requires stride 1, no upsample,
   * c3 % 64 == 0, c4 % 64 == 0, a v2 tile. */
  const void* x3;
  const void* x4;
  int32_t c3;
  int32_t c4;
  /* PP_X_CONV3X3, stride 1, no upsample (ABI v14): `GroupNorm(gn_in_groups) -> SiLU` of the conv INPUT concat(x1, x2)
   * applied by the loader, i.e. ResnetBlock2D's `norm1 -> nonlinearity -> conv1` / `norm2 -> nonlinearity -> dropout
   * -> conv2` (diffusers 0.27 ResnetBlock2D.forward; ctor sites /root/reference/powerpaint/models/unet_2d_blocks.py:
   * 1274-1285) as ONE launch: the normalised activation is never written to memory.  x1 / x2 then hold the RAW
   * (un-normalised) tensors; gn_in_acc -> int64 [batch][gn_in_groups][2] = the (sum, sum of squares) accumulators of
   * concat(x1, x2) in the fixed-point format of gn_acc above, COMPLETE before this launch (written by the producers'
   * epilogues); gn_in_gb -> fp32 [c1 + c2][2] = (gamma, beta) interleaved per channel.  The optional 1x1 tail (x3, x4)
   * is NOT normalised.  Zero padding applies to the normalised tensor, as in the reference.  Supported shapes:
   * pp_conv_gn_supported(); the arithmetic of the normalisation is that of pp_groupnorm_apply_acc -- values rounded to
   * the 16-bit format before the convolution where the two-launch path stores them -- up to 1-ulp differences of the SiLU
   * (the loader evaluates it with v_exp / v_rcp: a few per cent of the normalised values differ from the apply launch's by
   * one unit of the 16-bit format; only the gn_next_* combine path below is bit-identical to the apply launch).  gn_in_*
   * on a PP_X_PLAIN launch is PP_ERR_UNSUPPORTED, gn_in_acc without gn_in_gb (or the reverse) PP_ERR_BAD_ARG. */
  const int64_t* gn_in_acc;
  const float* gn_in_gb;
  int32_t gn_in_groups;
  int32_t gn_in_silu;   /* must be 1 (every GroupNorm in front of a 3x3 conv of the path is followed by SiLU) */
  float gn_in_eps;
  /* (ABI v19) with out_dup_rows: the statistics subscriptions k whose bit is set in gn_dup_mask belong to a consumer of the
   * FULL (duplicated) tensor -- the epilogue adds every (batch item b, group) contribution to batch item b + gn_dup_batch of
   * gn_acc[k] as well.  A subscription without its bit is a consumer of the half batch (the next layer of the prefix). */
  int32_t gn_dup_batch;
  /* (ABI v17) The GroupNorm (+ SiLU) that CONSUMES this launch's output, applied by the split-K combine itself: where one
   * workgroup of the combine owns a whole (batch item, 160-column tile) -- rows_per_batch <= 256, i.e. the 16 x 16 and 8 x 8
   * levels -- and the consumer's groups lie whole inside the tile, the tile's fixed-point (sum, sum of squares) ARE the
   * groups' statistics, so the combine writes `out` (the raw tensor: residual / skip consumers) AND
   * gn_next_out = [SiLU](GroupNorm(out)) with the arithmetic of pp_groupnorm_apply_acc (bit-identical), and the separate
   * apply launch (6.5 .. 12 us each at those levels) disappears.  gn_next_sub = which of gn_acc[0 / 1] carries the
   * consumer's grouping (its accumulator is still updated).  Honoured only where pp_gemm_gn_next_ok() = 1; ignored
   * (never silently: the caller asks first) otherwise.  Reference: the norm1 / norm2 / Transformer2DModel.norm modules
   * behind a ResnetBlock2D conv, /root/reference/powerpaint/models/unet_2d_blocks.py:1274-1300. */
  void* gn_next_out;
  const float* gn_next_gamma;
  const float* gn_next_beta;
  float gn_next_eps;
  int32_t gn_next_silu;
  int32_t gn_next_sub;
  int32_t gn_dup_mask;   /* see gn_dup_batch */
  /* (ABI v20) with w_batch_stride and act = PP_ACT_SOFTMAX80: bias and ln_colsum advance by this many floats per batch item */
  int32_t vec_batch_stride;
  /* (ABI v21; v22: every split combines its own share) The split-K combine INSIDE the producing kernel.  tile_ctr != NULL
   * permits it: pp_gemm_combine_ctr_bytes() bytes of device memory (two 64-bit words per tile), ZERO before the first launch
   * that uses them, private to this launch within a step.  The words are monotonic arrival counts: a launch adds its split
   * count to bits 40.. of each and leaves bits 0..39 (the round's bookkeeping) zero; they may be re-zeroed between launches
   * (a step's accumulator-pool zeroing does) but never while one runs.  Where pp_gemm_bf16 finds the launch eligible -- lean
   * epilogue, an 8-wave ping-pong or halo-tile kernel, 2 / 4 / 8 splits of a tile of >= 16 x splits rows, tiles % 8 == 0 so
   * that the splits of a tile share an XCD (csrc/gemm_combine.h) -- the splits of a tile wait for each other at the tile's
   * counter (XCD-local atomics; BOUNDED: a split that gives up marks its share abandoned and the last arriver combines it)
   * and each sums ITS 1 / S of the tile's rows over the fp32 slabs, in slab order, and runs the combine's epilogue on them
   * (incl. gn_acc; gn_next_* behind a second arrival, the shares' integer sums exchanged through per-tile scratch that
   * pp_gemm_workspace_bytes() already counts behind the slabs): same bits as the separate combine launch, which then does
   * not happen.  Not eligible => the separate combine as before; tile_ctr is a permission, never a request that can fail.
   * combine_fault (optional): a device counter that stays non-zero if an abandoned share was never taken back (splits of a
   * tile on different XCDs count in different L2s and never reach S) or a second-arrival wait hit its bound: the caller
   * checks it at its synchronisation points and refuses the results (pp_* never synchronises).  Replaces the combine
   * behind the split-K convs / Linears of the 32x32 and 16x16 levels
   * (/root/reference/powerpaint/models/unet_2d_blocks.py:1457-1500, 850-899, 2696-2770). */
  uint64_t* tile_ctr;
  uint32_t* combine_fault;
} PPGemmArgs;
#define PP_GN_SUM_SCALE 16777216.0f /* 2^24 */
#define PP_GN_SQ_SCALE 1048576.0f   /* 2^20 */

#define PP_TILE_AUTO 0
#define PP_TILE_128x160 1
#define PP_TILE_64x160 2
#define PP_TILE_256x160 3

int pp_gemm_bf16(const PPGemmArgs* args, void* stream);   /* (historic name: bf16 or fp16 per args->dtype) */
/* bytes of PPGemmArgs.workspace this request needs: the fp32 slabs of a split-K launch + (ABI v22) behind them 6 KB per tile
 * of statistics scratch where the launch could combine in-kernel (tile_ctr); 0 = single pass */
size_t pp_gemm_workspace_bytes(const PPGemmArgs* args);
/* 1 if this launch (as pp_gemm_bf16 would configure it) can accumulate GroupNorm statistics (gn_acc), else 0 */
int pp_gemm_gn_stats_ok(const PPGemmArgs* args);
/* 1 if pp_gemm_bf16 runs this PP_X_CONV3X3 request with the GroupNorm + SiLU of its input fused into the loader
 * (gn_in_acc / gn_in_gb set), else 0 -- the caller then keeps pp_groupnorm_apply_acc + a plain conv.
 * 2 (round 6) for a PLAIN request (no gn_in_*) that pp_gemm_bf16 routes to the same halo-tile loop WITHOUT the
 * normalisation instead of the tap-major implicit GEMM: stride 1 (Upsample2D's nearest-2x conv included), tile = PP_TILE_AUTO, output images at least 16 wide
 * (every input pixel crosses the global -> LDS path once per tile instead of once per tap: 52 against 58 us at 64x64,
 * K = 2880; profiles/r06_conv_raw.txt).  Same request, same result contract; a routing fact for tests and planners. */
int pp_conv_gn_supported(const PPGemmArgs* args);
/* (ABI v16) 1 if the fused launch is supported AND, by the per-shape measurements on MI355X (profiles/
 * r04_conv_gn_variants.txt), at least as fast as pp_groupnorm_apply_acc + the plain conv it replaces; the engine asks this
 * one when it lays out a ResnetBlock2D.  The normalisation costs ~600 wave cycles per 8-pixel x 64-channel strip and is
 * repeated per 160-column tile and per halo row.  Round 6 (ABI v22): since plain convs run on the same loop WITHOUT the
 * normalisation (pp_conv_gn_supported() == 2) and the apply kernels' SiLU uses the hardware reciprocal, apply + plain conv
 * wins at every level (step -2.0 % same-box, profiles/r06_conv_raw.txt): this returns 0 for every shape of the SD-1.5 plans;
 * the fused launch remains an operator for a caller that sets gn_in_* . */
int pp_conv_gn_preferred(const PPGemmArgs* args);
/* (ABI v17) 1 if this launch, as pp_gemm_bf16 would configure it, ends in the split-K combine that can apply the consumer
 * GroupNorm of subscription `sub` (PPGemmArgs.gn_next_*), else 0. */
int pp_gemm_gn_next_ok(const PPGemmArgs* args, int sub);
/* (ABI v21 / v22) Bytes of PPGemmArgs.tile_ctr the library ADVISES for this launch (16 per tile): 0 = the launch does not run
 * split-K, can never combine in-kernel (tile form, tiles % 8 != 0, epilogue not lean, placement check failed), or is one where
 * the separate combine launch measured faster on MI355X.  Measured inside the headline step's hipGraph (profiles/
 * r06_fused_combine.txt): launches of 2 and 4 splits whose workgroups are all resident at once (tiles x splits <= CUs) are a
 * draw to 8 us better per launch and a draw on the step with 18 kernels fewer -- advised; 8 splits (shares of 16 / 32 rows at
 * the 8x8 level and the 32 -> 16 downsample: the tail is a chain of ~1 us memory round trips against one 11 us kernel) lose
 * 2 .. 3 us per launch, 0.4 % on the step -- not advised.  (The first form, ONE workgroup combining the whole tile, lost
 * everywhere: +3 %.)  A caller that sets tile_ctr anyway (16 bytes per 128-row x 160-column tile always suffice) gets the
 * in-kernel combine wherever it is correct, bit for bit the separate one.  Asked at plan-build time, never inside a stream
 * capture: the first call per process may run pp_xcd_placement_ok(). */
size_t pp_gemm_combine_ctr_bytes(const PPGemmArgs* args);
/* (ABI v21) 1 if pp_gemm_bf16 will combine this launch in-kernel (tile_ctr set and every condition met, incl. the
 * gn_next_* apply where requested), else 0: how many kernels the launch is. */
int pp_gemm_combine_fused(const PPGemmArgs* args);
/* (ABI v21) Launches a probe grid on the current device and SYNCHRONISES it (the one entry point that does): 1 if the
 * workgroups of dim3(tiles, splits) grids are placed on the XCDs round-robin by their linear id (what the in-kernel
 * combine's co-location rests on), else 0.  Cached per device. */
int pp_xcd_placement_ok(void);

/* Small-M ("skinny") linear in fp32 accumulate: out[b][n] = act_in(x[b][:]) . W[n][:] + bias[n], b < rows <= 16.
 * Replaces TimestepEmbedding.linear_1/linear_2 and the 22 ResnetBlock2D.time_emb_proj (batched into one call by
 * concatenating their weights along N) -- [diffusers-0.27.0]; reference call site unet_2d_condition.py:1155-1156.
 * x fp32 [rows][K]; W bf16 [N][K]; bias fp32; out fp32 [rows][ldo]. act_in: 0 none, 2 silu. act_out likewise. */
int pp_linear_skinny(const float* x, int rows, int K, const void* w, const float* bias, int N, float* out, int ldo,
                     int act_in, int act_out, int dtype, void* stream);

/* Sinusoidal timestep embedding (Timesteps(dim, flip_sin_to_cos=True, freq_shift=0)), unet_2d_condition.py:914-938.
 * out fp32 [rows][dim] = [cos(t f_k), sin(t f_k)], t read from device pointer (one value broadcast to all rows). */
int pp_timestep_embedding(const float* t_dev, int rows, int dim, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) on NHWC bf16, two launches: statistics, then apply.
 * Replaces ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out + conv_act
 * (unet_2d_condition.py:1351-1353) -- nn.GroupNorm(32, C, eps).
 *   stats : x = concat(x1[c1], x2[c2]) per pixel; per-chunk partial (sum, sum of squares) per group -> workspace
 *           (pp_groupnorm_workspace_bytes()).
 *   apply : folds the partials (fp64, fixed order), then y[b][p][c] = act((x - mean) * rstd * gamma + beta) as bf16,
 *           y row stride = c1 + c2.  `workspace` is the buffer the stats call wrote (same batch / hw / channels).
 */
size_t pp_groupnorm_workspace_bytes(int batch, int hw, int C);
int pp_groupnorm_stats(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups,
                       float* workspace, int dtype, void* stream);
int pp_groupnorm_apply(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups, float eps,
                       const float* gamma, const float* beta, const float* workspace, int silu, void* y,
                       int dtype, void* stream);

/* GroupNorm apply from accumulated statistics: acc = int64 [batch][groups][2] filled by the producers' epilogues
 * (PPGemmArgs.gn_acc); otherwise identical to pp_groupnorm_apply. */
int pp_groupnorm_apply_acc(const void* x1, int c1, const void* x2, int c2, int batch, int hw, int groups, float eps,
                           const float* gamma, const float* beta, const int64_t* acc, int silu, void* y, int dtype,
                           void* stream);
/* dst[0..n) = 0 (64-bit words): one launch zeroes the statistics accumulators of a whole forward pass */
int pp_zero_u64(void* dst, long long n, void* stream);

/* LayerNorm over the last dim, bf16 [rows][C] -> bf16 [rows][C]; BasicTransformerBlock.norm1/2/3 (eps 1e-5). */
int pp_layernorm(const void* x, int rows, int C, const float* gamma, const float* beta, float eps, void* y,
                 int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused attention forward  O = softmax(Q K^T * scale) V, non-causal, no mask (AttnProcessor2_0 /
 * F.scaled_dot_product_attention) -- [diffusers-0.27.0] Attention used by Transformer2DModel.
 *   q  : bf16, element (b, i, h, j) at q[(b*nq + i)*ldq + h*d + j]
 *   k  : bf16, element (b, t, h, j) at k[(b*nk + t)*ldk + h*d + j]
 *   vt : bf16 V TRANSPOSED, element (b, t, h, j) at vt[((b*heads + h)*d + j)*ldvt + t]   (ldvt >= nk, mult of 8)
 *   o  : bf16, same indexing as q with ldo.
 * head_dim d in {40, 80, 160}; nk arbitrary (keys >= nk masked).
 */
int pp_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                     int batch, int heads, int nq, int nk, int d, float scale, int dtype, void* stream);
/* The same op with the kernel named explicitly (since ABI v8; op-level parity tests address every shipping kernel,
 * the pipelines use PP_ATTN_AUTO = what pp_attention_fwd picks):
 *   PP_ATTN_PHASED   attn_fwd_kernel   (three-phase, every head dim / key count)
 *   PP_ATTN_PIPE_Q32 attn_pipe_kernel  (software-pipelined, 32 queries per wave, 64-key tiles; d = 40, nk % 64 == 0, nk >= 256)
 *   PP_ATTN_PIPE_Q64 attn_pipe_kernel  (64 queries per wave on 32-key tiles; same shapes; AUTO takes it when
 *                                       batch * heads * ceil(nq / 256) >= 512, i.e. the 64x64 and 128x128 latents)
 *   PP_ATTN_PIPE_LOG2 (ABI v20) the pipelined kernel AUTO would take, for a q that already holds Q * scale * log2(e) (the
 *                                       producing GEMM's epilogue multiplies: pp_tfront q_scale): the running softmax
 *                                       reference enters as the initial accumulator of the QK^T MFMAs, a score leaves the
 *                                       matrix pipe as the exp2 argument (no VALU multiply-add per score); `scale` is NOT
 *                                       applied.  Shapes: pp_attention_log2_ok(); never chosen by AUTO.
 * A named kernel that does not cover the shape returns PP_ERR_UNSUPPORTED. */
#define PP_ATTN_AUTO 0
#define PP_ATTN_PHASED 1
#define PP_ATTN_PIPE_Q32 2
#define PP_ATTN_PIPE_Q64 3
#define PP_ATTN_PIPE_LOG2 4
int pp_attention_log2_ok(int nq, int nk, int d);
int pp_attention_fwd_variant(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                             int batch, int heads, int nq, int nk, int d, float scale, int dtype, int variant,
                             void* stream);
/* [rows = b*nk + t][cols] bf16 (row stride ld) -> vt[b][col][t] (row stride ldvt).  Used for the cross-attention V
 * (computed once per call: encoder_hidden_states are step-invariant) and as the unfused fallback for self-attention. */
int pp_transpose_v(const void* v, int ld, int batch, int nk, int cols, void* vt, int ldvt, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Direct 3x3 convolutions (pad 1) for MFMA-unfriendly channel counts:
 *   pp_conv3x3_direct    : cout % 8 == 0, any cin, stride 1|2 -- conv_in (Cin 4/9 -> 320; BrushNet conv_in_condition,
 *                          /root/reference/powerpaint/models/BrushNet_CA.py:223-228,822-823) and the
 *                          ControlNetConditioningEmbedding convs (3->16->...->320, SiLU between).
 *                          x: bf16 NHWC [batch][hin][win][cin]; w: bf16 [3][3][cin][cout] (cout contiguous);
 *                          bias fp32; add: optional bf16 NHWC tensor added before the optional SiLU; out: bf16 NHWC.
 *   pp_conv3x3_smallcout : cout == 4, cin % 8 == 0, stride 1 -- conv_out (unet_2d_condition.py:1354).
 *                          x: bf16 NHWC; w: bf16 [cout][3][3][cin]; out: fp32 NCHW [batch][cout][h][w] (the UNet
 *                          boundary layout, consumed directly by pp_cfg_sched_step).
 */
int pp_conv3x3_direct(const void* x, int batch, int hin, int win, int cin, const void* w, const float* bias,
                      int cout, int stride, int silu_out, const void* add, void* out, int dtype, void* stream);
int pp_conv3x3_smallcout(const void* x, int batch, int h, int w_, int cin, const void* w, const float* bias, int cout,
                         float* out_nchw, int dtype, void* stream);
/* conv_norm_out + SiLU + conv_out in one launch (unet_2d_condition.py:1351-1354): GroupNorm statistics from the
 * producers' accumulators (`acc`, as pp_groupnorm_apply_acc), the normalised activation rounded to the 16-bit format
 * where the two-launch path stores it, 3x3 / pad 1 conv to `cout` = 4 channels, fp32 NCHW out.  x NHWC [batch][h][w][cin];
 * w [4][9][cin] (pp_conv3x3_smallcout's layout).  pp_gn_conv3x3_smallcout_supported(): cout == 4, cin % 32 == 0, cin <= 384. */
int pp_gn_conv3x3_smallcout_supported(int cin, int cout, int groups);
int pp_gn_conv3x3_smallcout(const void* x, int batch, int h, int w, int cin, int groups, float eps, const float* gamma,
                            const float* beta, const int64_t* acc, const void* wgt, const float* bias, int cout,
                            float* out_nchw, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Layout / assembly kernels at the module boundary (NCHW torch tensors <-> internal NHWC bf16).
 * pp_nchw_to_nhwc : src fp32|bf16|f16 NCHW [batch][c][hw] -> dst bf16 [batch][hw][ldc] at channel offset c0.
 *                   src_batch_mod > 0 reads batch item (b % src_batch_mod)  -> `torch.cat([latents]*2)`
 *                   (pipeline_PowerPaint.py:990) without a copy.
 * pp_nhwc_to_nchw : 16-bit [batch][hw][c] -> fp32 | the same 16-bit format, NCHW.
 * dtype codes: PP_DT_F32 0, PP_DT_BF16 1, PP_DT_F16 2; the trailing `dtype` is the format of the NHWC side.
 */
int pp_nchw_to_nhwc(const void* src, int src_dtype, int batch, int c, int hw, int src_batch_mod, void* dst, int ldc,
                    int c0, int dtype, void* stream);
int pp_nhwc_to_nchw(const void* src, int batch, int c, int hw, void* dst, int dst_dtype, int dtype, void* stream);
/* out = a + b on bf16 tensors of n elements (n % 8 == 0).  The residual adds that cannot ride a GEMM epilogue:
 * ControlNet `down_block_res_sample + down_block_additional_residual` on the already-consumed skip tensors
 * (/root/reference/powerpaint/models/unet_2d_condition.py:1263-1272,1296-1297). */
int pp_add_bf16(const void* a, const void* b, void* out, long long n, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused classifier-free guidance + scheduler step on fp32 NCHW latents.
 * Replaces pipeline_PowerPaint.py:1018-1023 / pipeline_PowerPaint_Brushnet_CA.py:1444-1449:
 *     eps = eps_u + g * (eps_c - eps_u);  latents = scheduler.step(eps, t, latents)
 * eps2 : fp32 [2*n] (uncond half first, pipeline_PowerPaint.py:516) when cfg != 0, else [n].
 * The step is the linear form  x_next = cx * x + ce * eps_or_x0 + cm * m_prev  with per-step coefficients read from a
 * DEVICE table coef[step][8] = {cx, c0, cm, a_inv, s_over_a, _, _, _} and a device step counter:
 *     x0   = a_inv * x - s_over_a * eps          (DPM-Solver++ data prediction; DDIM: folded into cx/c0)
 *     kind 0 (DDIM, eta=0): x_next = cx * x + c0 * eps
 *     kind 1 (DPM-Solver++ 2M): x_next = cx * x + c0 * x0 + cm * m_prev ; m_prev <- x0
 *     kind 2 (PNDM / PLMS, PNDMScheduler(skip_prk_steps=True) -- the scheduler the SD-1.5 checkpoint config names):
 *             table rows are 16 floats {w0..w3, a, b, slot(h1), slot(h2), slot(h3), push_slot | -1, use_saved, save};
 *             m = w0 eps + w1 h1 + w2 h2 + w3 h3;  x_next = a * (use_saved ? saved : x) + b * m;  `m_prev` is the state
 *             [5][n] fp32 (4 history slots + the saved sample), zero before the first step.  N steps = N + 1 rows.
 *     kind 3 (UniPC, predict_x0 / bh1|bh2 / order <= 3 -- what app.py:197 installs on the ppt-v2 pipeline):
 *             rows are 16 floats {sigma_t, alpha_t, use_corr, k_last, k_m1, k_m2, k_m3, k_x0, p_xc, p_x0, p_m1, p_m2};
 *             x0 = (x - sigma_t eps) / alpha_t;  xc = use_corr ? k . (last, m1, m2, m3, x0) : x;
 *             x_next = p . (xc, x0, m1, m2);  state [4][n] fp32 <- (xc, x0, m1, m2), zero before the first step.
 * `step_dev` (int32 on device) selects the row.  advance_ticket == NULL: it is NOT incremented here (pp_step_advance
 * does).  advance_ticket != NULL (ABI v20; a zero-initialised device word that belongs to this call site): this launch is
 * the step's last reader of the counter and moves it on itself -- the block that finishes last does step_dev[0] += 1
 * and returns the ticket to zero; no pp_step_advance launch behind it.
 */
int pp_cfg_sched_step(const float* eps2, int cfg, float guidance, float* latents, float* m_prev, int n, int kind,
                      const float* coef_table, int32_t* step_dev, uint32_t* advance_ticket, void* stream);
/* Stochastic DDIM, `eta > 0` (the pipelines' `eta` argument, pipeline_PowerPaint.py:736-745 / 1023: it reaches
 * `DDIMScheduler.step` only): after pp_cfg_sched_step of the same step,  latents += std_dev_t * noise  with
 * std_dev_t = coef[step][4] of the kind-0 table (eta * sqrt((1-a_prev)/(1-a_t) * (1-a_t/a_prev)); column 3 then holds
 * sqrt(1-a_prev-std_dev_t^2)) and `noise` fp32 [n] drawn by the host from the caller's generator, once per step. */
int pp_ddim_variance_noise(float* latents, const float* noise, int n, const float* coef_table, const int32_t* step_dev,
                           void* stream);

/* (ABI v18) Front end of Transformer2DModel at C = 320 in one launch (csrc/tfront.hip):
 *     hs = proj_in(GroupNorm(x)),   q | k | v = to_q / to_k / to_v(LayerNorm1(hs))
 * i.e. `hidden_states = self.norm(hidden_states); hidden_states = self.proj_in(hidden_states)` of diffusers 0.27
 * Transformer2DModel.forward plus `norm_hidden_states = self.norm1(hidden_states)` and the attn1 projections of
 * BasicTransformerBlock.forward (ctor site /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300).  x: raw rows
 * [M][c]; gn_acc: the (sum, sum of squares) accumulators of x in the format of PPGemmArgs.gn_acc, complete before the launch;
 * w1 / b1: proj_in [c][c] ([out][in]) and bias; w2p: the QKV weight [3c][c] with LayerNorm1's gamma folded in AND its input
 * index permuted inside every group of 32 (position 8 kg + j holds index 16 (j >> 2) + 4 kg + (j & 3): the MFMA
 * accumulator layout of the first GEMM is then the B-operand layout of the second); cs2 / b2: its column sums and W beta.
 * Outputs: hs [M][c] (the residual of attn1.to_out), qk [M][>= 2c] = Q | K, vt [batch][c][ldvt] = V transposed -- what
 * pp_attention_fwd reads.  Arithmetic = pp_groupnorm_apply_acc -> pp_gemm_bf16(row_stats_out) -> pp_gemm_bf16(ln_stats)
 * up to the fp32 summation order.  pp_tfront_supported() = 1 for c = 320, 128-row tiles inside one batch item.
 * q_scale (ABI v20): the Q third is multiplied by it in fp32 before its one rounding to 16 bits -- 1.0 for
 * pp_attention_fwd, head_dim^-0.5 * log2(e) for PP_ATTN_PIPE_LOG2.  hs and qk: 16-byte aligned, ldhs % 8 == ldqk % 8 == 0
 * (round 6: the rows leave as whole 128-byte lines). */
int pp_tfront_supported(int M, int c, int rows_per_batch, int gn_groups);
int pp_tfront(const void* x, int ldx, const void* gn_acc, const float* gn_gamma, const float* gn_beta, float gn_eps, int gn_groups,
              const void* w1, const float* b1, const void* w2p, const float* cs2, const float* b2, float ln_eps, void* hs,
              int ldhs, void* qk, int ldqk, void* vt, int ldvt, int M, int c, int rows_per_batch, float q_scale, int dtype,
              void* stream);
/* Fused cross-attention sub-block of BasicTransformerBlock (norm2 -> attn2 -> residual) for C = 320, 8 heads, <= 80
 * context tokens -- the three launches `to_q` (pp_gemm_bf16, LayerNorm folded) -> pp_attention_fwd -> `to_out`
 * (pp_gemm_bf16 + residual + row moments) of the 64x64 level as ONE (ctor site /root/reference/powerpaint/models/
 * unet_2d_blocks.py:1289-1300; diffusers 0.27 `BasicTransformerBlock.forward`: `attn2(norm2(h), encoder_hidden_states) + h`).
 * The encoder hidden states do not change during a call, so pp_xattn_fold contracts K into the query projection and V
 * into the output projection once per prompt:
 *     gt   [batch][heads*80][c]   G^T[b][h*80+key][:] = scale*log2(e) * sum_d K[b][key][h*d'+d] * wq[h*d'+d][:]   (16-bit)
 *     gcs, gbias [batch][heads*80] fp32: the same contraction of q_colsum / q_bias (the folded-LayerNorm terms of
 *          to_q; NULL = 0); gbias = -inf on the padded keys (key >= nctx), which masks them
 *     ht   [batch][c][heads*80]   H^T[b][n][h*80+key] = sum_d wo[n][h*d'+d] * V^T[b][h*d'+d][key], the contraction
 *          index permuted inside every group of 32 (position 8*kg + j holds index 16*(j>>2) + 4*kg + (j&3)) so that the
 *          accumulator registers of the logits are the B fragments of the second GEMM without any data movement
 * with k [batch*nctx][ldk] and vt [batch][c][ldvt] as pp_attention_fwd takes them, wq / wo [c][c] row-major
 * ([out][in]).  pp_xattn_block then computes, per 128-row tile (rows_per_batch % 128 == 0),
 *     p   = softmax_per_head( rstd*(x gt^T - mean*gcs) + gbias )          (exp2 domain; mean / rstd from ln_stats as in
 *                                                                           PPGemmArgs, NULL = no LayerNorm folded)
 *     out = p ht^T + bias_o + res ,   row_stats_out[m][c/160][2] = (sum, sum of squares) of the stored values.
 * (ABI v19) pre_w != NULL (C = 320): the Linear in FRONT of the sub-block rides in the same launch --
 *     h = x pre_w^T + pre_b + res     (BasicTransformerBlock: `attn1(norm1(h0)) + h0`, i.e. attn1.to_out applied to the
 *     self-attention output x, residual res = h0; pre_w [c][c] 16-bit ([out][in]), pre_b fp32 or NULL),
 * then the sub-block on h (rounded to 16 bits where the separate launch stored it): h is the logits' input, the source of
 * the LayerNorm row moments (computed in the kernel; ln_stats is ignored, ln_tiles > 0 = "a LayerNorm is folded into gt")
 * and the residual of the output.  The logits' B operand then comes out of the first GEMM's accumulators, so gt must be
 * packed with its channel index permuted inside every group of 32: pp_xattn_fold(kperm = 1) (position 8 kg + j holds
 * channel 16 (j >> 2) + 4 kg + (j & 3), the permutation pp_tfront uses).
 * (ABI v19) src_wrap_rows > 0: x, res and ln_stats hold only src_wrap_rows rows and output row m reads row
 * (m mod src_wrap_rows), M <= 2 * src_wrap_rows -- the two halves of a CFG batch are identical up to this sub-block (the
 * first place the prompt enters, unet_2d_condition.py:1183-1236), so everything in front of it ran on one half; the folded
 * operands (gt, gcs, gbias, ht) are per batch item of the FULL batch.  C = 320 only (PP_ERR_UNSUPPORTED otherwise).
 * pp_xattn_block_supported() = 1 when the shape is one this kernel takes (the caller keeps the three-launch chain
 * otherwise); both entry points return PP_ERR_UNSUPPORTED for other shapes.
 * (ABI v20) pp_xattn_fold(kperm = 2) stores ht with its contraction index in NATURAL order (column h * 80 + key): the
 * operands of the two-GEMM form of the same sub-block, where the row-local kernels lose (C = 1280: M <= 2048 rows) --
 *     P   = pp_gemm_bf16(x, w = gt, w_batch_stride = 640 c, bias = gbias, ln_colsum = gcs, vec_batch_stride = 640,
 *                        ln_stats, act = PP_ACT_SOFTMAX80)                          [M][640] probabilities
 *     out = pp_gemm_bf16(P, w = ht, w_batch_stride = 640 c, bias = bias_o, res1 = h, row_stats_out)
 * with half the multiplications of the chain at C = 1280 (2 x 640 c per row instead of 2 c^2 + 4 * 77 c). */
int pp_xattn_block_supported(int M, int c, int rows_per_batch, int nctx, int heads);
int pp_xattn_fold(const void* k, int ldk, const void* vt, int ldvt, int batch, int nctx, int heads, int c, const void* wq,
                  const float* q_colsum, const float* q_bias, const void* wo, float scale, void* gt, float* gcs,
                  float* gbias, void* ht, int kperm, int dtype, void* stream);
int pp_xattn_block(const void* x, int ldx, const void* res, int ldres, const float* ln_stats, int ln_tiles, float ln_eps,
                   const void* gt, const float* gcs, const float* gbias, const void* ht, const float* bias_o, void* out,
                   int ldo, float* row_stats_out, int M, int c, int rows_per_batch, int src_wrap_rows, const void* pre_w,
                   const float* pre_b, int dtype, void* stream);

/* (ABI v19) FeedForward + proj_out of a C = 320 transformer in ONE launch (csrc/ff_fused.hip):
 *     out = epilogue2( [ h (.) gelu_erf(g) | hs ] W2'^T ),    h | g = rstd (hs W1^T - mean colsum1) + bias1   (interleaved)
 * i.e. `ff(norm3(hidden_states)) + hidden_states` of diffusers 0.27 BasicTransformerBlock.forward (FeedForward with GEGLU)
 * followed by Transformer2DModel.proj_out + the block residual (ctor site /root/reference/powerpaint/models/
 * unet_2d_blocks.py:1289-1300): the two launches pp_gemm_bf16(act = PP_ACT_GEGLU, folded LayerNorm) -> pp_gemm_bf16 over the
 * K-concatenation [g | hs] -- without the [M][1280] GEGLU tensor ever being written: the hidden dimension is streamed in
 * chunks of 64 units against persistent fp32 accumulators of the second GEMM.
 *   g2 : the SECOND GEMM exactly as pp_gemm_bf16 would take it (x_mode PLAIN, N = 320, c1 = 1280, c2 = 320, K = 1600):
 *        x2 / ldx2 = hs (raw rows: LayerNorm3 is folded), w = [W_po W_ff2 | W_po] ([320][1600], 16-bit) with the HIDDEN index
 *        (the first 1280 columns) permuted inside every group of 32 -- storage position 8 kg + 2 q + e holds unit
 *        8 q + 2 kg + e -- so that the GEGLU values in the first GEMM's accumulator registers are the second GEMM's B
 *        fragments; bias, scale, res1 (+ res1_wrap_rows), res2, out / ldo, dtype and the gn_acc subscriptions are honoured as
 *        in pp_gemm_bf16; x1 is NOT read (the tensor it would name does not exist).  rowvec, GEGLU / V^T / fp32 outputs, row
 *        moments, gn_next_*, out_dup_rows: PP_ERR_UNSUPPORTED.
 *   w1 : [2560][320] 16-bit, rows interleaved (h0, h1, g0, g1) as PP_ACT_GEGLU takes them, LayerNorm gamma folded in;
 *   b1 / cs1 : [2560] fp32 bias (incl. W beta) and column sums of w1 (cs1 may be NULL when ln_stats is NULL);
 *   ln_stats : row moments of hs in the layout of PPGemmArgs.ln_stats (ln_tiles partials per row) or NULL (no LayerNorm folded).
 * Arithmetic = the two-launch chain up to the fp32 summation order of the second GEMM (the GEGLU values are rounded to the
 * 16-bit format exactly where the chain stores them).  pp_ff_fused_supported(): c == 320, M and rows_per_batch multiples of 128. */
int pp_ff_fused_supported(int M, int c, int rows_per_batch);
int pp_ff_fused(const PPGemmArgs* g2, const void* w1, const float* b1, const float* cs1, const float* ln_stats, int ln_tiles,
                float ln_eps, int w2_kperm, void* stream);

/* ppt-v1 with a 4-channel (non-inpainting) UNet -- the `num_channels_unet == 4` branch of the loop body,
 * pipeline_PowerPaint.py:1025-1039: after pp_cfg_sched_step of the same step
 *     latents[b] = (1 - mask) * (a * image_latents + b * noise[b]) + mask * latents[b]
 * with (a, b) = renoise_table[step] = (sqrt(abar), sqrt(1 - abar)) of the NEXT timestep (`scheduler.add_noise(
 * init_latents_proper, noise, timesteps[i + 1])`), (1, 0) on the last step (the clean image latents are put back).
 * fp32; latents / noise [batch][channels][hw], image_latents [channels][hw] and mask [hw] = the FIRST image's
 * (`image_latents[:1]`, `mask[:1]` broadcast, as the reference does); renoise_table [steps][2]. */
int pp_latent_blend(float* latents, const float* image_latents, const float* mask, const float* noise,
                    const float* renoise_table, const int32_t* step_dev, int batch, int channels, int hw, void* stream);

/* (ABI v19) The head of a captured denoising step as ONE launch -- what pp_embed_splice (one row), pp_nchw_to_nhwc and
 * pp_zero_u64 did in three (each a launch floor of 5-6 us): temb_out[0 .. row_floats) = temb_table[step][:] (the per-schedule
 * table of the time-embedding chain's output, unet_2d_condition.py:1155-1156: it depends on the timestep alone);
 * x_in[b][p][c0 + j] = latents[b mod src_batch_mod][j][p] for j < c (fp32 NCHW -> 16-bit NHWC: `torch.cat([latents] * 2)`,
 * pipeline_PowerPaint.py:990, without the copy); zero_dst[0 .. n_zero) = 0 (64-bit words: the GroupNorm accumulators). */
int pp_step_head(const float* temb_table, const int32_t* step_dev, float* temb_out, int row_floats, const float* latents,
                 int batch, int c, int hw, int src_batch_mod, void* x_in, int ldc, int c0, int dtype, void* zero_dst,
                 long long n_zero, void* stream);
/* t_out[0] = timesteps[step]; used at the top of a captured step.  advance: ++step. */
int pp_step_select_t(const float* timesteps, const int32_t* step_dev, float* t_out, void* stream);
int pp_step_advance(int32_t* step_dev, void* stream);

/* Bit-exact mask prep (pipeline_PowerPaint.py:143-147,677-679; pipeline_PowerPaint_Brushnet_CA.py:1312,1342-1344):
 *   mode 0: out = (mask >= 0.5) ? 1 : 0                                  fp32 [n]
 *   mode 1: out = image * (mask < 0.5), mask broadcast over `c` channels  fp32 [batch][c][hw]
 *   mode 2: nearest downsample of a [batch][1][h][w] mask to [batch][1][ho][wo]  (F.interpolate default)
 *   mode 3: out = (sum_c rgb[b][c][p] < 0) ? 1 : 0                        (BrushNet original_mask)
 */
int pp_mask_prep(int mode, const float* a, const float* b, float* out, int batch, int c, int h, int w, int ho, int wo,
                 void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Task-prompt embedding splice -- the gather half of EmbeddingLayerWithFixes.forward
 * (/root/reference/powerpaint/utils/utils.py:378-483: ids >= num_embeddings -> row 0 of the base table, then the
 * learned [n_vec][dim] block of each placeholder is spliced over every occurrence of its id run).
 * The host walks the ids exactly as the reference does and hands over one source row per output row:
 *   src_row[i] >= 0 : out[i] = table[src_row[i]]            (base nn.Embedding weight)
 *   src_row[i] <  0 : out[i] = ext[-src_row[i] - 1]         (all external embeddings, concatenated along rows)
 * Rows are copied as `row_bytes` raw bytes (any dtype; bit-exact).  `ext` may be NULL when no src_row is negative.
 */
int pp_embed_splice(const void* table, const void* ext, const int32_t* src_row, void* out, int n_rows,
                    long long row_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Row softmax  p[r][c] = softmax_c(scale * s[r][c]),  fp32 logits -> bf16 probabilities.
 * The VAE mid-block attention (diffusers AutoencoderKL: one head of dim 512 over H*W tokens -- the call sites are
 * /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:657-669,1051) is two pp_gemm_bf16 launches around this
 * kernel; pp_attention_fwd covers the UNet's head dims only.  lds / ldp: row strides in elements (multiples of 4).
 */
int pp_softmax_rows(const float* s, long long lds, int rows, int n, float scale, void* p, long long ldp, int dtype,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Attention for short sequences: the CLIP text tower the pipelines call through `self.text_encoder(ids)[0]`
 * (/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:378-423; transformers CLIPTextModel: 77 tokens, 12 heads
 * of 64, causal mask).  q / k / v / o: bf16, element (b, t, h, j) at x[(b*n + t)*ld + h*d + j] (v NOT transposed).
 * d == 64, nk <= 128; causal: key j visible to query i iff j <= i (needs nq == nk).
 */
int pp_attention_small(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                       int batch, int heads, int nq, int nk, int d, float scale, int causal, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H_ */
