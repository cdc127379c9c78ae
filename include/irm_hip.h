/* libirm_hip.so - C ABI of the MI355X (gfx950) image-restoration hot path.
 *
 * The reference (leducthanhig/image-restoration-models) has no FFI of its own:
 * its hot path is `model(input_tensor)` inside run_model_inference
 * (src/utils.py:403-437), executed by PyTorch ATen.  These entry points are the
 * operations a maintainer binds (ctypes stubs in INTEGRATION.md) to replace that
 * forward and the per-tile host loop around it.  Each declaration cites the
 * reference code it stands in for.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; tensors are float32,
 *    planar NCHW: element (b, c, n) of a tensor with batch stride `bs` (in
 *    elements) lives at base[b*bs + c*N + n], N = H*W.  Batch strides let a
 *    kernel read / write a channel slice of a larger (concat) buffer.
 *  - no allocation, no host synchronisation inside: all work is enqueued on
 *    `stream` (a hipStream_t), workspaces are passed in.  The only state the
 *    library keeps is a per-(kernel, device) flag "the >64 KiB LDS limit has
 *    been raised" and read-only device constants (a zero page for border
 *    loads, a dump page for masked stores).  No environment variable is read
 *    in the product build (experiment switches exist only under -DIRM_PROBES).
 *  - return value: 0 = enqueued, IRM_EINVAL (-1) = rejected arguments (nothing
 *    launched), IRM_ELAUNCH (-2) = the HIP launch failed.
 *  - packed weights ("wp"): MFMA B-operand order produced on the host,
 *      wp[mtile][kstep][lane] = W[16*mtile + (lane&15)][4*kstep + (lane>>4)],
 *    zero padded, ksteps = 4*ceil(K/16) for the 1x1 GEMM and, for the 3x3
 *    conv, wp[tap][mtile][kstep][lane] with ksteps = 2*ceil(Ci/8).
 */
#ifndef IRM_HIP_H
#define IRM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* irm_stream_t; /* == hipStream_t */

#define IRM_OK 0
#define IRM_EINVAL (-1)
#define IRM_ELAUNCH (-2)

#define IRM_ACT_NONE 0
#define IRM_ACT_RELU 1
#define IRM_ACT_GELU 2
#define IRM_ACT_SILU 3

#define IRM_LN_NONE 0
#define IRM_LN_WITHBIAS 1
#define IRM_LN_BIASFREE 2

/* ABI version of this header (bumped on any signature change). */
int irm_version(void);

/* Per-pixel LayerNorm statistics over channels: stats[b][0][n] = mean,
 * stats[b][1][n] = 1/sqrt(biased_var + eps).
 * Replaces the statistics part of BiasFree_/WithBias_LayerNorm
 * (src/restormer/restormer.py:25-70); the normalisation itself is applied as
 * the prologue of irm_gemm1x1_f32. */
int irm_ln_stats_f32(const float* x, long x_bs, float* stats, int B, int C, int N, float eps,
                     irm_stream_t stream);

/* 1x1 convolution as a GEMM on the exact-f32 MFMA:
 *   y[b][co][n] = act(sum_k W[co][k] * LN(x[b][k][n]) + bias[co]) (+ res[b][co][n])
 * Replaces the bias-free 1x1 nn.Conv2d projections of Restormer
 * (restormer.py:82 project_in, :86/:107 project_out, :105 qkv, :223/:228
 * reduce_chan_level{3,2}, :240 skip_conv), the LayerNorm that precedes them
 * (ln_mode, stats from irm_ln_stats_f32, lnw/lnb = LayerNorm weight/bias) and
 * the residual add of TransformerBlock.forward (restormer.py:146-150).
 * With w_bs != 0 each batch element has its own packed matrix (the folded
 * attention matrix from irm_mdta_finalize_f32).
 * ct: output-channel tiles per pass (3,4,6,8,9); ygroups: grid split of the
 * passes (>=1).  stats_out (optional, NULL to skip): [B][2][N] LayerNorm
 * statistics (mean, 1/sqrt(var+eps)) of y over its M channels for the NEXT
 * LayerNorm, produced in the epilogue; needs ceil(M/16) <= ct (one pass).
 * res_scale (optional, [M]): y = ... + res * res_scale[co] (MaIR's skip_scale,
 * mairunet_arch.py:375-377).
 * Operands whose rows are 16-byte aligned (N % 4 == 0, aligned bases/strides)
 * take the LDS-DMA ring kernel, anything else an exact scalar-path kernel. */
int irm_gemm1x1_f32(const float* wp, long w_bs, const float* x, long x_bs, float* y, long y_bs,
                    const float* res, long r_bs, const float* bias, const float* stats, const float* lnw,
                    const float* lnb, int ln_mode, int act, int B, int M, int K, int N, int ct, int ygroups,
                    float* stats_out, float eps, const float* res_scale, irm_stream_t stream);

/* The same 1x1 conv with the GEMM proper on the fp16 matrix cores as an fp32 emulation (shared weights; a
 * residual only without the LayerNorm prologue, as in irm_gemm1x1_f32): x (after the LayerNorm prologue) is
 * split in registers into fp16 hi + lo, the weights arrive split by the host, three MFMAs (lo*hi, hi*lo, hi*hi)
 * accumulate in fp32 - 2^-21 relative per product.
 * wp_split: same size as wp; per (mtile, 16-channel stage) 64 lanes x 4 fp16 hi, then 64 lanes x 4 fp16 lo,
 *   lane (g = lane>>4, m = lane&15) holds W[16 mtile + m][16 stage + 4 j + g], j = 0..3  (zero padded).
 * Range: a LayerNorm output cannot leave fp16; inputs without the prologue are scaled by 2^-4 for the split
 * (|x| < 1e6), the accumulators are rescaled before bias / residual. */
int irm_gemm1x1_f16x3_f32(const float* wp_split, long w_bs, const float* x, long x_bs, float* y, long y_bs, const float* res,
                          long r_bs, const float* bias, const float* stats, const float* lnw, const float* lnb,
                          int ln_mode, int act, int B, int M, int K, int N, int ct, int ygroups, float* stats_out,
                          float eps, const float* res_scale, irm_stream_t stream);

/* LayerNorm + 1x1 conv with PRE-SPLIT fp16 operands (gemm_ps.hip) - the C >= 192 levels of Restormer: LayerNorm
 * (restormer.py:25-70) + Attention.qkv / FeedForward.project_in (restormer.py:82,105), K in {192, 384}.
 *
 * irm_ln_split_f16: xs = fp16 hi + lo of LN(x) * scale in MFMA fragment order (statistics computed in the kernel,
 *   two passes in registers; biased variance, eps inside the root):
 *     xs [B * N / 16][K / 32][hi | lo][64 lanes][8 halves], lane = 16 g + i: pixel 16 tile + i, channel 32 ks + 8 g + e
 *   (2 + 2 bytes per element: the size of the fp32 tensor).  scale: a power of two that keeps LN(x) * scale inside
 *   fp16 (host: ln_split_scale - from the static bound sqrt(K - 1) |w| + |b| of a WithBias LayerNorm; values are
 *   clamped at +-65000).  ln_mode 1 = WithBias, 2 = BiasFree.  N % 16 == 0, K % 32 == 0, K <= 384.
 * irm_gemm_presplit_f16x3_f32: y[b][m][n] = out_scale * sum_k W'[m][k] xs[b][k][n] + bias[m] with three
 *   v_mfma_f32_16x16x32_f16 per product (lo*hi, hi*lo, hi*hi), fp32 accumulate; out_scale = 1 / (s_w scale).
 *   wps [ceil(M/16)][K / 32][hi | lo][64 lanes][8 halves]: W s_w split by the host in the same fragment order (lane =
 *   16 g + m: output channel 16 tile + m), s_w a power of two with max|W| s_w in [2^13, 2^14).
 *   ct output tiles per pass; mgroups: workgroups sharing the output tiles of a pixel block (no empty group);
 *   wg_shape = 10 x waves per workgroup + pixel tiles per wave: K 192: 42 or 32 (ct 6 / 8), 43 (ct 4); K 384: 81
 *   (ct 6 / 8); 0 = default.  N % 16 == 0 (pixel tiles are numbered through the batch); act must be 0. */
/* irm_ln_split_f16 + irm_gemm_presplit_f16x3_f32 in one launch for K = 192 (same arithmetic): every wave normalises
 * and splits its own pixel tiles while loading them; mgroups as above (1 where the launch is a single round). */
int irm_ln_gemm_presplit_f16x3_f32(const void* wps, const float* x, long x_bs, const float* lnw, const float* lnb, int ln_mode,
                                   float x_scale, float eps, float* y, long y_bs, const float* bias, float out_scale, int B,
                                   int M, int K, int N, int mgroups, irm_stream_t stream);
/* The same launch with y tile-major channel-last in chunks of 64 channels [H/8 * W/32 tiles][M / 64][256 pixels][64] (N = H W;
 * H % 8 == 0, W % 32 == 0, M % 64 == 0): the MFMA operands are swapped so that a lane holds 4 channels of its pixel (16-byte
 * stores), and the 64 channels a workgroup produces at a time form one contiguous run over its consecutive pixels. */
int irm_ln_gemm_presplit_cl_f16x3_f32(const void* wps, const float* x, long x_bs, const float* lnw, const float* lnb, int ln_mode,
                                   float x_scale, float eps, float* y, long y_bs, const float* bias, float out_scale, int B,
                                   int M, int K, int H, int W, int mgroups, irm_stream_t stream);
int irm_ln_split_f16(const float* x, long x_bs, const float* lnw, const float* lnb, int ln_mode, float scale, float eps,
                     void* xs, int B, int K, int N, irm_stream_t stream);
int irm_gemm_presplit_f16x3_f32(const void* wps, const void* xs, float* y, long y_bs, const float* bias, float out_scale,
                                int act, int B, int M, int K, int N, int ct, int mgroups, int wg_shape,
                                irm_stream_t stream);

/* GDFN tail of the C >= 192 levels on pre-split operands (gemm_ps.hip):
 * irm_dwconv3x3_gate_split_f16: gs = fp16 hi + lo of gelu_erf(dw(x[c])) * dw(x[c + hid]) * scale in MFMA fragment order
 *   gs [B * H W / 16][ceil(hid/32)][hi | lo][64 lanes][8 halves] (lane = 16 g + i: pixel 16 tile + i, gate channel
 *   32 ks + 8 g + e; channels beyond hid zero; values clamped at +-65000) - replaces FeedForward.dwconv + chunk +
 *   F.gelu(x1) * x2 (restormer.py:84, 89-91) where the consumer is irm_gemm_presplit_res_f16x3_f32.  W % 16 == 0;
 *   ch = gate channels per workgroup (8 / 16 / 32, 0 = by row length).
 * irm_gemm_presplit_res_f16x3_f32: y = res + bias + out_scale * (W' gs), K = 32 KS streamed (project_out + the block's
 *   residual, restormer.py:86, 92, 148); wps [ceil(M/16)][KS][hi | lo][64 lanes][8 halves] (W s_w, columns zero padded to
 *   32 KS, split by the host), out_scale = 1 / (s_w scale).  M <= 384, KS % 4 == 0, N % 16 == 0; y may alias res.
 *   wg_shape: 0 = default; M <= 192: 42 / 82, M <= 384: 41 / 81 (10 x waves + pixel tiles per wave). */
int irm_dwconv3x3_gate_split_f16(const float* x, long x_bs, const float* w, const float* bias, void* gs, float scale,
                                 int B, int hid, int H, int W, int ch, irm_stream_t stream);
int irm_gemm_presplit_res_f16x3_f32(const void* wps, const void* xs, float* y, long y_bs, const float* res, long r_bs,
                                    const float* bias, float out_scale, int B, int M, int KS, int N, int wg_shape,
                                    irm_stream_t stream);

/* Depth-wise 3x3 convolution, zero pad 1: y[b][c] = act(dw3x3(x[b][c]; w[c]) + bias[c]).
 * Replaces Attention.qkv_dwconv (restormer.py:106) and MaIR's conv2d+SiLU.
 * w: [C][9] (device), bias: [C] or NULL. */
int irm_dwconv3x3_f32(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs, int B,
                      int C, int H, int W, int act, irm_stream_t stream);

/* GDFN depth-wise conv + gate: x has 2*hid channels,
 *   y[b][c] = gelu_erf(dw(x[b][c])) * dw(x[b][c + hid]),  c < hid.
 * Replaces FeedForward.dwconv + chunk + F.gelu(x1)*x2 (restormer.py:84,89-91). */
int irm_dwconv3x3_gate_f32(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs,
                           int B, int hid, int H, int W, irm_stream_t stream);

/* Depth-wise 3x3 fused into the 1x1 conv that consumes it (LDS-DMA ring, 8x32 pixel tiles):
 *   g[k] = gate ? gelu_erf(dw(x[k])) * dw(x[k + K]) : dw(x[k]),   k < K
 *   y    = W g + bias (+ res),                                     M <= 96, W % 4 == 0, 16-byte aligned
 * wp: irm_gemm1x1_f32's packed weight (w_bs != 0: one matrix per sample).
 * dwp: [4*ceil(K/4)][DWS] floats per input channel k (rows beyond K zero), every value stored twice
 * (v, v) for the packed-fp32 stencil:
 *   gate:    DWS = 40: taps of channel k (9 pairs), bias[k], taps of channel k+K (9 pairs), bias[k+K]
 *   no gate: DWS = 20: taps (9 pairs), bias
 * GELU uses erf from Abramowitz-Stegun 7.1.26 (absolute error <= 1.5e-7).
 * stats_out: optional [B][2][H*W] LayerNorm statistics of y (as irm_gemm1x1_f32).
 * gate = 1 replaces FeedForward.dwconv + chunk + gelu(x1)*x2 + project_out (+ the block's residual)
 * (restormer.py:84-93,148); gate = 0 replaces qkv_dwconv on v + attn @ v + project_out with the folded
 * matrix of irm_mdta_finalize_f32 (restormer.py:118-131,147). */
int irm_dwgemm_f32(const float* wp, long w_bs, const float* dwp, const float* x, long x_bs, float* y, long y_bs,
                   const float* res, long r_bs, const float* bias, int gate, int B, int M, int K, int H, int W,
                   float* stats_out, float eps, irm_stream_t stream);

/* irm_dwgemm_f32 with the 1x1 part as an fp32 emulation on the fp16 matrix cores.
 * wp_split: the output conv's weight in irm_gemm1x1_f16x3_f32's hi/lo fp16 order.  The stencil outputs are
 * scaled by 2^-4 before the split (range: |g| < 1e6), the fp32 accumulators carry that scale to the end. */
int irm_dwgemm_f16x3_f32(const float* wp_split, long w_bs, const float* dwp, const float* x, long x_bs, float* y, long y_bs,
                         const float* res, long r_bs, const float* bias, int gate, int B, int M, int K, int H, int W,
                         float* stats_out, float eps, irm_stream_t stream);

/* Whole-branch kernels for C <= 96 (fused_block.hip): the input tile (8 x 32 pixels + 1-pixel halo, all C channels)
 * is read once, normalised in registers (LayerNorm statistics computed in the kernel) and kept there as fp16 hi/lo
 * MFMA operands; the wide intermediate (2*hid or 3*C channels) exists only in LDS, 32 channels at a time.  All 1x1
 * convs are the fp32 emulation of irm_gemm1x1_f16x3_f32 (three fp16 MFMAs on hi/lo splits, fp32 accumulate).
 * x != y (tiles read their neighbours' halo); W % 4 == 0; channel / pixel axes dense, batch strides free.
 *
 * irm_gdfn_fused_f16x3_f32:  y = x + project_out(gelu_erf(dw(h)[:hid]) * dw(h)[hid:]) + bias2,  h = project_in(LN(x)) + b
 *   replaces norm2 + FeedForward + residual (restormer.py:25-70, 76-93, 148); ln_mode 1 = WithBias, 2 = BiasFree.
 * Operands packed by the host (Python: _hip.pack_gdfn_fused), KS = ceil(C/32), S = ceil(hid/16), CT = ceil(C/16):
 *   rec [S+1][KS*1024 + 512] floats, record i =
 *     [2 tiles][KS][hi|lo][64 lanes][8 halves]: W1' * s1 split into fp16 hi + lo, W1' = project_in.weight * diag(ln.weight);
 *        tile 0 = gate channels 16 i + m, tile 1 = channels hid + 16 i + m; lane = 16 g + m, half j -> input
 *        channel 32 ks + 8 g + j (zero beyond C / hid; record S: zeros)
 *     [10][32] floats: the 9 depth-wise taps + bias of the channels of stage i - 1 (tile 0 | tile 1) (record 0: zeros);
 *        irm_gdfn_fused_f16x3_f32 only: the taps AND the bias of tile 1 (the channels hid + ..., the gate's multiplier)
 *        multiplied by 2^-4 - the kernel computes gelu(dw(h1)) * (dw(h2) / 16), saturates at +-65000 and splits that
 *        into fp16 hi + lo; inv_s2 carries the factor 16 back
 *     [32] floats: bias of h for stage i = project_in.bias + project_in.weight @ ln.bias;  pad to 512
 *   w2 [ceil(S/2)][CT][hi|lo][64 lanes][8 halves]: project_out.weight * s2 split; lane = 16 g + co, half j of
 *        super-stage T -> gate channel 32 T + 16 (j >> 2) + 4 g + (j & 3)
 *   inv_s1 = 1 / (16 s1), inv_s2 = 16 / s2 with power-of-two s1, s2 chosen so that max |W| * s lies in [2^13, 2^14)
 *   (the fp16 lo parts stay normal numbers for weights of any magnitude).
 *
 * irm_qkv_dw_fused_f16x3_f32:  y[:, :M] = dw3x3(W LN(x) + b)  (M = 3C: norm1 + Attention.qkv + qkv_dwconv,
 *   restormer.py:105-106, 116).  rec as above with S = ceil(M/32) and tile t of record i = output channels
 *   32 i + 16 t + m (Python: _hip.pack_qkv_fused). */
int irm_gdfn_fused_f16x3_f32(const float* rec, const float* w2, const float* bias2, const float* x, long x_bs, float* y,
                             long y_bs, int ln_mode, float eps, float inv_s1, float inv_s2, int B, int C, int hid, int H,
                             int W, irm_stream_t stream);
int irm_qkv_dw_fused_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode, float eps,
                               float inv_s1, int B, int C, int M, int H, int W, irm_stream_t stream);
/* irm_qkv_dw_fused_tm_f16x3_f32 (round 3): M = 3C with q, k (output channels [0, 2C)) stored TILE-MAJOR inside the q, k part
 * of y - [H/8 * W/32 tiles][2C][256 pixels of the 8 x 32 tile, row-major] per image, the same 2C N floats - and v planar
 * at channels [2C, 3C) as before.  q, k are read by the Gram pass only (irm_mdta_gram_tm_f16x3_f32): a stage of that pass
 * then reads 2c row segments 1 KiB apart inside one contiguous block instead of 2c segments a plane apart.
 * H % 8 == 0, W % 32 == 0, C % 16 == 0.
 * Tile-major channel-LAST activations inside a stage (x_tm / v_tm / the `lay` bits of irm_attn_gdfn_fused_f16x3_f32): element
 * (channel, y, x) of an image's Cb-channel tensor at (((y >> 3) (W / 32) + (x >> 5)) 256 + (y & 7) 32 + (x & 31)) Cb + channel:
 * a pixel's channels are contiguous (a lane fetches the 8 channels of a k-step as two 16-byte loads, every fetched line is
 * used whole - the planar 34-pixel halo rows start one pixel before a line boundary) and an 8 x 32 tile is ONE contiguous
 * Cb KiB block.  x_tm: x is read in that layout (Cb = C); v_tm: v is written in it inside channels [2C, 3C) of y (Cb = C).
 * The first kernel of a stage reads planar, the last writes planar. */
int irm_qkv_dw_fused_tm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode, float eps,
                                  float inv_s1, int B, int C, int H, int W, int x_tm, int v_tm, irm_stream_t stream);
/* irm_qkv_dw_cm_f16x3_f32 (round 3, fused_qkv_cm.hip): the same contract as irm_qkv_dw_fused_tm_f16x3_f32 (same `rec`, q, k
 * tile-major, v channel-last when v_tm else planar, x channel-last when x_tm else planar; bit-identical results) with a
 * channel-major hidden image: the 1x1 conv with the pixels as the MFMA row index, the stencil with the channel on the lane
 * (~40 % fewer LDS reads per stage), q, k stored 16 bytes at a time. */
int irm_qkv_dw_cm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode, float eps,
                            float inv_s1, int B, int C, int H, int W, int x_tm, int v_tm, irm_stream_t stream);
/* irm_qkv_gram_cm_f16x3_f32 (round 3): irm_qkv_dw_cm_f16x3_f32 for C = 48 with ONE head (the full-resolution encoder level)
 * with the Gram pass inside: q and k are never written - the stencil outputs are scaled by gscale[channel] (the powers of two
 * of irm_mdta_gram_f16x3_f32, [2C]), split into fp16 hi/lo and multiplied on the matrix cores, per chunk of 4 consecutive
 * tiles of an image one partial record [48 x 48 Gram | 48 |q|^2 | 48 |k|^2] goes to part [B][H W / 1024][2400] (the format of
 * irm_mdta_gram_*: irm_mdta_finalize_* reduces it with nchunk = H W / 1024); v as in irm_qkv_dw_cm_f16x3_f32.  The chunks are
 * fixed sets of tiles of one image, so the result does not depend on the batch.  (H / 8) (W / 32) % 4 == 0.
 * Replaces qkv_dwconv(qkv(norm1(x))) + the Gram / norm part of Attention.forward (restormer.py:105-106, 116-123). */
int irm_qkv_gram_cm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, const float* gscale, float* part,
                              int ln_mode, float eps, float inv_s1, int B, int C, int H, int W, int x_tm, int v_tm,
                              irm_stream_t stream);
/* irm_attn_gdfn_fused_f16x3_f32 (round 3): the last step of the attention branch inside the GDFN kernel's prologue,
 *   x' = x + bias_o + Mfold[b] v          (restormer.py:131, 147: project_out(attn @ v) + x, Mfold by irm_mdta_finalize_frag_f16x3_f32)
 *   y  = x' + project_out(gelu_erf(dw(h)[:hid]) * dw(h)[hid:]) + bias2,  h = project_in(LN(x')) + b      (:76-93, 148)
 * so that x' is neither written nor read back (replaces irm_gemm1x1_f16x3_f32(Mfold, v, res = x) + irm_gdfn_fused_f16x3_f32).
 * v: [B][C][H][W] (batch stride v_bs), split behind a fixed 2^-4 scale like every un-normalised GEMM input;
 * mfold_frag: [B][2 KS][KS][hi|lo][64 lanes][8 halves], KS = ceil(C/32); bias_o: [C] or NULL; C % 16 == 0.
 * rec as for irm_gdfn_fused_f16x3_f32 except for the order of project_in's input channels: half j of lane 16 g + m of
 * k-step ks <-> input channel 16 (2 ks + (j >> 2)) + 4 g + (j & 3) (the order in which the MFMA of the first step
 * leaves x' in the registers; Python: _hip.pack_gdfn_fused(..., kperm=True)).
 * lay: bit 0 - x tile-major, bit 1 - v tile-major, bit 2 - y written tile-major (above; needs H % 8 == 0, W % 32 == 0). */
int irm_attn_gdfn_fused_f16x3_f32(const float* rec, const float* w2, const float* bias2, const float* x, long x_bs,
                                  const float* v, long v_bs, const float* mfold_frag, const float* bias_o, float* y,
                                  long y_bs, int ln_mode, float eps, float inv_s1, float inv_s2, int B, int C, int hid,
                                  int H, int W, int lay, irm_stream_t stream);

/* GDFN tail of the C = 192 level in one kernel (round 3, fused_tail.hip):
 *   x += project_out(gelu_erf(dw(h)[:hid]) * dw(h)[hid:]) + bias2      (restormer.py:84-93, 148), in place on x [B][C][H][W]
 * h_cl: h = project_in(LN(x)) + b as irm_ln_gemm_presplit_cl_f16x3_f32 writes it - tile-major channel-last
 *   in chunks of 64 channels, [H/8 * W/32 tiles][2 hid_pad / 64][256 pixels][64], the two halves of h (gate | multiplier)
 *   at channels [0, hid) and [hid_pad, hid_pad + hid), the rest zero (project_in packed with padded halves:
 *   _hip.pack_pin_padded);
 * rec [S = ceil(hid/16)][512] floats: [10][32] = the 9 depth-wise taps + bias of the stage's channels (16 gate | 16 multiplier;
 *   the multiplier's taps and bias x 2^-4), pad;  w2, inv_s2 as for irm_gdfn_fused_f16x3_f32 with CT = 12
 *   (Python: _hip.pack_gdfn_tail).  C == 192, H % 8 == 0, W % 32 == 0, hid_pad % 64 == 0.
 * Replaces irm_dwconv3x3_gate_f32 + irm_gemm1x1_f16x3_f32(res = x): the gated tensor never reaches HBM. */
int irm_gdfn_tail_f16x3_f32(const float* h_cl, long h_bs, const float* rec, const float* w2, const float* bias2, float* x,
                            long x_bs, float inv_s2, int B, int C, int hid, int hid_pad, int H, int W, irm_stream_t stream);

/* MDTA pass 1: per-chunk partial Gram matrices and squared norms.
 * qkv: [B][3C][N] (after qkv_dwconv; q rows [0,C), k rows [C,2C)).
 * part: workspace [B][heads][ceil(N/chunk)][c*c + 2c] floats, c = C/heads,
 * c % 16 == 0, chunk % 64 == 0.
 * Replaces F.normalize + q @ k^T of Attention.forward (restormer.py:122-125);
 * the normalisation is applied to the Gram matrix afterwards. */
int irm_mdta_gram_f32(const float* qkv, long bs, float* part, int B, int C, int heads, int N, int chunk,
                      irm_stream_t stream);

/* The same pass as an fp32 emulation on the fp16 matrix cores (three v_mfma_f32_16x16x32_f16 per product, fp32
 * accumulation; the squared norms stay in fp32 on the vector pipe): the sweep over q, k becomes a memory stream.
 * scale: [2C] power-of-two factors (q channels, then k channels) applied to the operands before the fp16 hi/lo split
 * and removed again from the records (exact).  The caller guarantees |q_c| scale_c < 65504 for EVERY input - i.e.
 * scale comes from a static bound of the producing layers (WithBias LayerNorm -> qkv conv -> depth-wise conv; host
 * side: gram_scales); without such a bound (BiasFree LayerNorm) irm_mdta_gram_f32 is the entry point to use.
 * c in {48, 96}, N % 64 == 0, chunk % 64 == 0, 16-byte aligned rows; same partial records as irm_mdta_gram_f32. */
int irm_mdta_gram_f16x3_f32(const float* qkv, long bs, const float* scale, float* part, int B, int C, int heads, int N,
                            int chunk, irm_stream_t stream);
/* The same pass over q, k in the tile-major order of irm_qkv_dw_fused_tm_f16x3_f32 (N % 256 == 0; v is not read). */
int irm_mdta_gram_tm_f16x3_f32(const float* qkv, long bs, const float* scale, float* part, int B, int C, int heads, int N,
                            int chunk, irm_stream_t stream);
/* ... and the f32-input ring pass (no operand scales: any LayerNorm flavour) over tile-major q, k: c = 48 / 96, N % 256 == 0. */
int irm_mdta_gram_tm_f32(const float* qkv, long bs, float* part, int B, int C, int heads, int N, int chunk,
                         irm_stream_t stream);

/* MDTA pass 2: reduce the partials (gsum: workspace [B][heads][c*c+2c]),
 * softmax((G_ij / (max(|q_i|,1e-12) max(|k_j|,1e-12))) * temperature[head]) and
 * fold with project_out: mfold[b] = packed(W_out * blockdiag(A_heads)), a C x C
 * matrix in irm_gemm1x1_f32's packed layout ([B][ceil(C/16)][4*ceil(C/16)][64];
 * must be zero-initialised once when C % 16 != 0).  attn (optional, may be
 * NULL) receives A as [B][heads][c][c].  wout: [C][C] row-major.
 * Replaces restormer.py:125-131 (softmax, attn @ v, project_out) together with
 * irm_gemm1x1_f32(mfold, v, res = block input). */
int irm_mdta_finalize_f32(const float* part, float* gsum, const float* temperature, const float* wout,
                          float* mfold, float* attn, int B, int C, int heads, int nchunk, irm_stream_t stream);
/* Same, with the folded matrix written in irm_gemm1x1_f16x3_f32's fp16 hi/lo order (same number of floats per
 * sample), for irm_dwgemm_f16x3_f32 / irm_gemm1x1_f16x3_f32 with w_bs = that size. */
int irm_mdta_finalize_f16x3_f32(const float* part, float* gsum, const float* temperature, const float* wout,
                                float* mfold_split, float* attn, int B, int C, int heads, int nchunk,
                                irm_stream_t stream);
/* Same, with the folded matrix as fp16 hi/lo 16x16x32 MFMA fragments for irm_attn_gdfn_fused_f16x3_f32:
 * mfold_frag [B][2 KS][KS][hi|lo][64 lanes][8 halves], KS = ceil(C/32), lane 16 g + m half e of fragment (mtile, ks) =
 * Mfold[16 mtile + m][32 ks + 8 g + e]; zero-initialise once (the slots beyond C are never written).  C % 16 == 0. */
int irm_mdta_finalize_frag_f16x3_f32(const float* part, float* gsum, const float* temperature, const float* wout,
                                     float* mfold_frag, float* attn, int B, int C, int heads, int nchunk,
                                     irm_stream_t stream);

/* Dense 3x3 convolution, stride 1, zero pad 1, implicit GEMM on the f32 MFMA:
 *   v = conv(x)[co] + bias[co]; if relu1: v = max(v,0);
 *   res_mode 1: v += res; res_mode 2: v = res - v; 3: v = clamp(tanh(v) + res, -1, 1)
 *   (DeblurGANv2 output, fpn_mobilenet.py:68-70); if relu2: v = max(v,0);
 * store_mode 0: y[b][co][h][w]; 1: PixelUnshuffle(2) -> [4Co][H/2][W/2];
 * 2: PixelShuffle(2) -> [Co/4][2H][2W].
 * Replaces OverlapPatchEmbed / Downsample / Upsample / output (+ inp_img) of
 * Restormer (restormer.py:156-189, 243, 281), B.conv 'CR'/'C' stacks of DnCNN
 * incl. x - n (network_dncnn.py:40-71) and REDNet's conv / ConvTranspose2d +
 * ReLU + skip (rednet.py:64-136).  ct in {1,2,3,4,6}. */
int irm_conv3x3_f32(const float* wp, const float* x, long x_bs, float* y, long y_bs, const float* res, long r_bs,
                    const float* bias, int B, int Ci, int Co, int H, int W, int relu1, int res_mode, int relu2,
                    int store_mode, int ct, int ygroups, irm_stream_t stream);

/* Dense 3x3 conv with <= 4 channels on one side (Co <= 4 or Ci <= 4), exact fp32 on the vector pipe (conv3x3_thin.hip):
 * image <-> feature convs are memory streams.  w: the plain weight [Co][Ci][3][3] (device).  Epilogue as irm_conv3x3_f32
 * (store_mode 0 only).  W % 4 == 0, 16-byte aligned tensors.  Replaces OverlapPatchEmbed and `output` (+ inp_img) of
 * Restormer (restormer.py:156-164, 281), the first / last conv of DnCNN (network_dncnn.py:40-71) and REDNet
 * (rednet.py:64-136), DeblurGANv2's `final` conv (fpn_mobilenet.py:68-70). */
int irm_conv3x3_thin_f32(const float* w, const float* x, long x_bs, float* y, long y_bs, const float* res, long r_bs,
                         const float* bias, int B, int Ci, int Co, int H, int W, int relu1, int res_mode, int relu2,
                         irm_stream_t stream);

/* irm_conv3x3_f32 as an fp32 emulation on the fp16 matrix cores (three v_mfma_f32_16x16x32_f16 per product: lo*hi,
 * hi*lo, hi*hi, fp32 accumulate; the staged halo tile is split ONCE per 32-channel stage into a channel-minor fp16 hi/lo
 * image in LDS and reused by 9 taps x all output tiles).  Same epilogue and store modes; needs W % 4 == 0 and 16-byte
 * aligned rows.  wp_split [ceil(Co/16)][ceil(Ci/32)][9 taps][hi|lo][64 lanes][8 halves]: lane = 16 g + m, half j ->
 * W[16 mtile + m][32 stage + 8 g + j][tap] * s, split into fp16 hi + lo; s a power of two with max|W| s in [2^13, 2^14);
 * inv_scale = 16 / s (activations are scaled by 2^-4 before their split: |x| < 1e6).  ct in {1, 2, 3, 4}. */
int irm_conv3x3_f16x3_f32(const float* wp_split, float inv_scale, const float* x, long x_bs, float* y, long y_bs,
                          const float* res, long r_bs, const float* bias, int B, int Ci, int Co, int H, int W, int relu1,
                          int res_mode, int relu2, int store_mode, int ct, int ygroups, irm_stream_t stream);

/* Tile extraction for the tiled-patch loop (src/utils.py:379-417):
 * img [H][W][C] uint8 (is_u16=0) or uint16 -> tiles [T][C][ph][pw] float32 =
 * img/255 (or /65535), + optional float64 noise field [th][tw][C] then clip
 * to [0,1] (add_gaussian_noise, utils.py:29-36; the field is generated on the
 * host with the reference's seed); with (mean, inv_std) != (0, 1) the value is
 * instead (raw - mean) * inv_std on the raw integer (DeblurGANv2 normalize =
 * albumentations Normalize, aug.py:31-39: mean 127.5, inv_std 1/127.5); padded
 * from (th,tw) to (ph,pw) by reflection (utils.pad, utils.py:174-181) or, with
 * pad_zero, by zeros (deblurganv2.pad, __init__.py:16-24).  origins: [T][2]
 * int32 (y0,x0). */
int irm_tile_extract(const void* img, int is_u16, const int* origins, const double* noise, float* tiles, int H,
                     int W, int C, int th, int tw, int ph, int pw, int T, float mean, float inv_std, int pad_zero,
                     irm_stream_t stream);

/* Gaussian-window blend + normalise + requantise (src/utils.py:427-450) with the
 * reference's float32 operation order; pred [T][Cp][ph][pw] (first Co channels,
 * [:th][:tw] used), window [ps][ps], out [H][W][Co] uint8/uint16.  If target
 * and sse are given, the integer sum of squared errors vs target is added to
 * *sse (device, 64-bit) for PSNR (utils.py:146). */
int irm_window_blend(const float* pred, const int* origins, const float* window, void* out, int is_u16,
                     const void* target, unsigned long long* sse, int H, int W, int Co, int Cp, int th, int tw,
                     int ph, int pw, int ps, int T, float post_scale, float post_shift, irm_stream_t stream);

/* --- DeblurGANv2 FPN-MobileNet (train-mode norms = per-(sample, channel) statistics) ---
 * stats[b][c] = {mean, 1/sqrt(biased var + eps)} over the H*W plane: BatchNorm2d in train mode on one
 * tile (mobilenet_v2.py:5-57 with deblurganv2/__init__.py:38) and InstanceNorm2d (fpn_mobilenet.py:96-104). */
int irm_chan_stats_f32(const float* x, long x_bs, float* stats, int B, int C, int N, float eps, irm_stream_t stream);
/* Same result contract with a caller-provided workspace (ws_floats >= 3 * B * C * ceil(1024 / (B*C)) floats is
 * always enough): large planes are split over several workgroups and merged in a fixed order (Chan's
 * parallel variance).  Falls back to the single-workgroup kernel when the planes are small or unaligned. */
int irm_chan_stats_ws_f32(const float* x, long x_bs, float* stats, float* ws, long ws_floats, int B, int C, int N,
                          float eps, irm_stream_t stream);
/* y = act((x - mean) * rstd * w[c] + b[c]) (+ res); w, b may be NULL (affine=False); act 0 none, 1 ReLU,
 * 4 ReLU6; in place allowed.  The residual is the InvertedResidual skip (mobilenet_v2.py:52-56). */
int irm_chan_norm_act_f32(const float* x, long x_bs, const float* stats, const float* w, const float* b,
                          const float* res, long r_bs, float* y, long y_bs, int B, int C, int N, int act,
                          irm_stream_t stream);
/* Stem conv: dense 3x3, stride 2, pad 1, no bias; w [Co][Ci][3][3]; out ceil(H/2) x ceil(W/2). */
int irm_conv3x3_s2_f32(const float* x, long x_bs, const float* w, float* y, long y_bs, int B, int Ci, int Co, int H,
                       int W, irm_stream_t stream);
/* Depth-wise 3x3, stride 2, pad 1, no bias; w [C][9]. */
int irm_dwconv3x3_s2_f32(const float* x, long x_bs, const float* w, float* y, long y_bs, int B, int C, int H, int W,
                         irm_stream_t stream);
/* out = (add ? add : 0) + nearest_upsample(src, scale)  (F.interpolate(mode="nearest"), fpn_mobilenet.py:57-66,
 * 141-145); src [B][C][Hs][Ws], out/add [B][C][Hs*scale][Ws*scale]. */
int irm_upsample_add_f32(const float* src, long s_bs, const float* add, long a_bs, float* out, long o_bs, int B, int C,
                         int Hs, int Ws, int scale, irm_stream_t stream);

/* [B][R][C] -> [B][C][R] (planar NCHW <-> channel-last tokens around the selective scan). */
int irm_transpose_f32(const float* in, long in_bs, float* out, long out_bs, int B, int R, int C,
                      irm_stream_t stream);

/* Selective-scan SSM of MaIR's LoSh2D.forward_core (mairunet_arch.py:226-261), replacing the third-party
 * mamba_ssm `selective_scan_fn(u, delta, A, B, C, D, delta_bias, delta_softplus=True)` (call site :252-258)
 * together with the scan-order gather / inverse gather (shift_scanf_util.py:206-244) and dt_proj (:243):
 *   for direction k, time t, pixel p = ids[k][t], channel d:
 *     dt = softplus(dtb[k][d] + sum_r dtw[k][d][r] * pT[p][k*J + r])                      J = R + 2N
 *     h[n] = exp(dt * A[k*D+d][n]) * h[n] + dt * pT[p][k*J + R + n] * xT[p][d]
 *     yT[k][p][d] = sum_n h[n] * pT[p][k*J + R + N + n] + Dskip[k*D+d] * xT[p][d]
 * xT [B][L][D], pT [B][L][4J] channel-last; ids [4][L] int32 (device); A = -exp(A_logs).
 * Chunked over L (chunk steps per wave): workspaces state [2][B][4][DB][nchunk][N][64], sdt and ysum
 * [B][4][DB][nchunk][64] with DB = ceil(D/64), nchunk = ceil(L/chunk); ysum receives per-chunk sums of y
 * (for the ShuffleAttn mean).  (N, R) in {(4,3), (8,6), (16,12), (32,24)}. */
int irm_selective_scan_f32(const float* xT, const float* pT, const int* ids, const float* dtw, const float* dtb,
                           const float* A, const float* Dskip, float* yT, float* state, float* sdt, float* ysum,
                           int B, int L, int D, int N, int R, int chunk, irm_stream_t stream);

/* After the scan: ShuffleAttn gate g = sigmoid(W * mean_HW(y) + b) per (direction, channel)
 * (mairunet_arch.py:21-60, :273; gw [4D][4], gb [4D]), direction sum (:274-275), out_norm LayerNorm over D
 * (:277) and * silu(z) (:278); z and out planar [B][D][L].  gate: workspace [B][4][D]; ysum (from the scan)
 * is reduced in place into its chunk-0 slots. */
int irm_losh_combine_f32(float* ysum, const float* gw, const float* gb, float* gate, const float* yT,
                         const float* nw, const float* nb, const float* z, long z_bs, float* out, long out_bs,
                         int B, int L, int D, int nchunk, float eps, irm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IRM_HIP_H */
