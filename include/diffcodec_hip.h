/*
 * diffcodec_hip.h — C-ABI of the MI355X (gfx950) decode hot path of DiffCodec.
 *
 * One shared object (libdiffcodec_hip.so) of `extern "C"` launchers.  Every entry point:
 *   - takes plain device pointers + sizes (no torch / C++ types), the HIP stream as `void*`;
 *   - allocates nothing (workspaces are passed in); the only process-wide state is one lock-free bit per
 *     (kernel, device) recording that the kernel's dynamic-LDS limit was raised on that device;
 *   - enqueues on the given stream and returns immediately: 0 = ok, -1 = invalid argument,
 *     -2 = launch failure.  Safe to capture into a hipGraph.
 *
 * Tensor conventions: activations of the diffusion models are NHWC bf16 ("pixel rows x channels");
 * the control-pyramid stage (extractor + splat) is NCHW fp32 like the reference (softsplat.py:279
 * forces fp32).  Weights: conv [Cout][KH*KW][Cin] bf16 (repacked from the checkpoint's OIHW), linear
 * [Cout][Cin] bf16 (checkpoint layout).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo).
 * The reference's only native interface on this path is the CuPy launch of `softsplat_out`
 * (controlnet/softsplat.py:284-345); everything else is reached through torch / diffusers module
 * calls, cited by call site.
 */
#ifndef DIFFCODEC_HIP_H
#define DIFFCODEC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ forward splatting (fp32, NCHW) */
/* Replaces cuda_launch(cuda_kernel('softsplat_out', ...)) — controlnet/softsplat.py:284-345 — together
 * with the 'soft' wrapper math of softsplat.py:246-247,253-270 and the optional `warped*(1-mask)` of
 * control_utils.py:69-70.  Deterministic gather form (sources binned by landing cell, every target sums its sources in
 * ascending raster order): bit-identical from run to run.  ws: scratch of dc_splat_ws_bytes(N, H, W) bytes (any contents).
 * metric [N,1,H,W]; mask [N,1,H,W] or NULL. */
long long dc_splat_ws_bytes(int N, int H, int W);
int dc_splat_soft_f32(const float* in, const float* flow, const float* metric, const float* mask,
                      float* out, void* ws, int N, int C, int H, int W, void* stream);
/* 'sum' mode: out = splat(in, flow) (softsplat.py:235,251); bit-exact with the sequential restatement oracle/splat_oracle.c. */
int dc_splat_sum_f32(const float* in, const float* flow, float* out, void* ws, int N, int C, int H, int W, void* stream);
/* compute_mask(a, b) — controlnet/control_utils.py:11-17: occ = (|| b + softsplat(a, b, ones, 'soft') ||_2 > 0.3).
 * ws: dc_splat_ws_bytes(N, H, W) bytes. */
int dc_occlusion_mask_f32(const float* flow_a, const float* flow_b, float* mask_out, void* ws,
                          int N, int H, int W, void* stream);
/* resize_and_normalize_flow_batched — controlnet/control_utils.py:74-97 (bilinear, align_corners=False,
 * then u/((w-1)/2), v/((h-1)/2)).  src [N,2,H,W] with batch stride `src_batch_stride` floats (lets the
 * caller pass flow[:, :2] / flow[:, 2:] views of the [N,4,H,W] control, extractors.py:268-269). */
int dc_flow_resize_normalize_f32(const float* src, long long src_batch_stride, float* dst,
                                 int N, int H, int W, int h, int w, void* stream);
/* Bi_Dir_ResidueExtractor's flow scaling — controlnet/extractors.py:181-183: bilinear (align_corners=False) then u/div_x,
 * v/div_y (div = 512/res: pixel rescale instead of the grid-style normalisation above). */
int dc_flow_resize_divide_f32(const float* src, long long src_batch_stride, float* dst, int N, int H, int W, int h, int w,
                              float div_x, float div_y, void* stream);
/* Confidence fusion + double-hole fill — controlnet/extractors.py:297-310. All [N,*,H,W] fp32.
 * occ_f = occ_b = NULL: fusion only (Bi_Dir_ResidueExtractor, extractors.py:199-203, has no hole fill). */
int dc_fuse_warped_f32(const float* warped_first, const float* warped_last, const float* conf_f,
                       const float* conf_b, const float* occ_f, const float* occ_b, float* fused,
                       int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------ fp32 NCHW direct conv (extractor) */
/* nn.Conv2d(k=3, padding=1, stride s in {1,2,4}) [+ SiLU] of the control extractors — controlnet/extractors.py:215-262,
 * control_utils.py:43-47.  x [N,Cin,H,W] (batch stride given, for channel-sliced views), w OIHW fp32. */
int dc_conv3x3_nchw_f32(const float* x, long long x_batch_stride, const float* w, const float* bias, float* y,
                        int N, int Cin, int H, int W, int Cout, int stride, int silu, void* stream);

/* ------------------------------------------------------------------ layout / dtype */
int dc_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int N, int C, int H, int W, void* stream);
int dc_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, void* stream);
int dc_nhwc_f32_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, void* stream);
int dc_f32_to_bf16(const float* src, void* dst, long long n, void* stream);

/* ------------------------------------------------------------------ MFMA implicit GEMM (bf16 in, fp32 acc) */
/* One kernel family serves: F.conv2d 3x3 / 1x1 inside diffusers ResnetBlock2D / Downsample2D / Upsample2D /
 * Transformer2DModel.proj_in/out (call sites flownet.py:83-124, pipeline.py:358-367,391), every nn.Linear
 * (to_q/k/v/out, ff, time_emb_proj), the FDN gamma/beta convs (control_utils.py:31-32) and the ControlNet
 * zero-convs (flownet.py:120-128).  Fusions: GroupNorm-affine(+SiLU) on load, nearest-2x upsample on load,
 * channel-concat of two inputs on load (UNet skip connections), bias + per-sample channel add (time
 * embedding) + out_scale (conditioning_scale) + residual add, or GEGLU, on store. */
typedef struct dc_conv_desc {
    const void* x1;         /* NHWC bf16 [N,H,W,C1] */
    const void* x2;         /* NHWC bf16 [N,H,W,C2] or NULL (channel concat: cat[x1,x2]) */
    const void* w;          /* bf16 [Cout][ksize*ksize][C1+C2] */
    const float* bias;      /* [Cout] or NULL */
    const float* gn_ab;     /* [gn_batch][C1+C2][2] fp32 (scale, shift) applied on load, or NULL.  1x1 launches: with gn_silu == 0 and a
                             * shape the K = 320 row-panel GEMM takes (>= 65,536 rows, whole 256-row panels inside one sample) the affine is
                             * applied to the kernel's register-resident activations and `stats_out` / `gn_part_out` stay available (same
                             * bits as dc_gn_apply_nhwc_bf16 followed by the plain launch); other 1x1 launches take the gather GEMM,
                             * which has no statistics epilogue (DC_ERR_INVALID if one is requested). */
    const float* row_add;   /* [N][Cout] fp32 added per sample (time-embedding projection) or NULL */
    const void* residual;   /* NHWC bf16 [M][Cout] added after scaling, or NULL */
    void* out;              /* bf16 [M][Cout] (f32 if out_f32; [M][Cout/2] for GEGLU) */
    float* splitk_ws;       /* fp32 [splitk][M][Cout] when splitk > 1: one slab per split, summed by the finish pass */
    int N, H, W;            /* input dims (before the fused upsample) */
    int C1, C2, Cout;
    int ksize;              /* 1 or 3 */
    int stride;             /* 1 or 2 */
    int pad;                /* 1: symmetric pad 1; 0: F.pad(x,(0,1,0,1)) + pad 0 (VAE encoder downsample) */
    int upsample;           /* 1: F.interpolate(scale_factor=2, nearest) fused on load */
    int Ho, Wo;             /* output dims */
    int gn_silu;            /* SiLU after the GN affine */
    int epilogue;           /* 0: linear; 1: GEGLU (weights/bias rows pre-interleaved 16 hid | 16 gate) */
    int out_f32;
    float out_scale;
    int splitk;             /* >= 1 */
    int gn_batch;           /* rows of gn_ab (sample n uses row n % gn_batch) */
    int act;                /* 0 none, 1 SiLU, 2 quick-GELU x*sigmoid(1.702x) — applied after bias/row_add, before out_scale */
    long long row_add_stride; /* floats between consecutive samples of row_add (0 = Cout) */
    /* nn.LayerNorm folded into the following nn.Linear (BasicTransformerBlock.norm1/2/3 -> attn1.to_q/k/v, attn2.to_q,
     * ff.net.0.proj): with W' = W diag(gamma) and b' = b + W beta prepared by the caller,
     *   Linear(LN(x)) = rstd * (x W'^T - mean * colsum(W')) + b',
     * so the GEMM runs on the raw rows and the normalisation is two per-row scalars in the epilogue; the standalone
     * LayerNorm pass (read + write of the whole activation) disappears.  1x1 / linear descriptors only. */
    const float* ln_stats;  /* consumer: [M][2] fp32 (mean, rstd) of every input row (dc_ln_finalize), or NULL */
    const float* ln_colsum; /* consumer: [Cout] fp32 sum over k of the (bf16) weight row, required with ln_stats */
    float* stats_out;       /* producer: [M][dc_gemm_row_stats_parts(Cout)][2] partial (sum, sum sq) of every OUTPUT row, or NULL */
    /* GroupNorm statistics of the OUTPUT from the epilogue that produces it (ResnetBlock2D.norm1/norm2, Transformer2DModel.norm
     * and FDN read them next): [dc_conv_gn_part_chunks(desc)][N][Cout][2] fp32 per-(pixel tile, sample, channel) partial
     * (sum, sum of squares) — the operand dc_gn_finalize takes — instead of a separate read pass over the tensor.  NULL = off.
     * Only the launches for which dc_conv_gn_part_chunks returns > 0 accept it. */
    float* gn_part_out;
    /* The LayerNorm finalize folded into the consumer (round 3).  ln_parts > 0: `ln_stats` holds the RAW partials
     * [M][ln_parts][2] (sum, sum of squares) exactly as a producer's `stats_out` (or dc_row_stats_bf16, parts = 1) wrote them, and
     * the launch forms (mean, rstd) over the C1 + C2 input channels itself, with dc_ln_finalize's arithmetic and `ln_eps`: kernels
     * whose waves own whole rows do it in their prologue; for the other kernels the launcher runs the dc_ln_finalize pass into
     * `ln_scratch` ([M][2] floats, caller-owned, required) first — either way the same bits as finalizing beforehand.
     * ln_parts == 0: `ln_stats` is the finalized [M][2] (mean, rstd). */
    int ln_parts;
    float ln_eps;
    float* ln_scratch;
} dc_conv_desc;
int dc_conv_igemm_bf16(const dc_conv_desc* desc, void* stream);
/* Workspace bytes needed for splitk>1 (0 otherwise). */
long long dc_conv_igemm_ws_bytes(const dc_conv_desc* desc);
/* Partials per output row that a 1x1 / linear launch with `stats_out` writes (one per wave column slice of the tile grid). */
int dc_gemm_row_stats_parts(int Cout);
/* Pixel-tile partials per sample that a launch of `desc` writes to gn_part_out; 0 if that launch cannot emit them (split-K,
 * GEGLU, fp32 output of a 1x1, kernels without the statistics epilogue, pixel tiles that straddle samples). */
int dc_conv_gn_part_chunks(const dc_conv_desc* desc);
/* Row statistics of a bf16 matrix x [M][C] for `ln_stats` when the producing launch cannot emit them (one partial per row):
 * stats [M][1][2] = (sum, sum of squares). */
int dc_row_stats_bf16(const void* x, float* stats, long long M, int C, void* stream);
/* Partials [M][parts][2] (from `stats_out` or dc_row_stats_bf16) -> mean_rstd [M][2] = (mean, 1/sqrt(var + eps)) over C channels:
 * the `ln_stats` operand.  One small launch per LayerNorm instead of the read + write pass over the activation. */
int dc_ln_finalize(const float* partials, float* mean_rstd, long long M, int parts, int C, float eps, void* stream);

/* Small-channel direct convs (NHWC bf16): Cin <= 8 (conv_in 4->320, VAE conv_in) and Cout <= 8
 * (conv_out 320->4, VAE conv_out 128->3 / 512->8, quant convs).  Same fusions on load as the igemm.
 * w_small_cin: bf16 [k*k][Cin][Cout]; w_small_cout: bf16 [Cout][k*k][Cin]. */
int dc_conv_small_cin_bf16(const void* x, const void* w, const float* bias, void* out, int N, int H, int W,
                           int Cin, int Cout, int ksize, int stride, int pad, int Ho, int Wo, void* stream);
int dc_conv_small_cout_bf16(const void* x, const void* w, const float* bias, const float* gn_ab, int gn_silu,
                            int gn_batch, void* out, int out_f32, int N, int H, int W, int Cin, int Cout, int ksize,
                            void* stream);

/* ------------------------------------------------------------------ normalisation */
/* GroupNorm (diffusers ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out; FDN.param_free_norm,
 * control_utils.py:24,29): per-(chunk,sample,channel) partial sums -> per-(sample,channel) scale/shift.
 * partials: fp32 [dc_gn_stats_chunks(HW,C)][N][C][2], fully written (no zeroing needed, no atomics). */
int dc_gn_stats_chunks(long long HW, int C);
int dc_gn_stats_nhwc_bf16(const void* x, float* partials, int N, long long HW, int C, void* stream);
/* Sums the chunk slabs, combines the channel sums of cat[x1,x2] into `groups` groups; writes ab [N][C1+C2][2].
 * gamma/beta may be NULL (affine=False, FDN). */
int dc_gn_finalize(const float* sums1, int C1, int chunks1, const float* sums2, int C2, int chunks2, const float* gamma,
                   const float* beta, float* ab, int N, int groups, long long HW, float eps, void* stream);
/* Same result as dc_gn_stats_nhwc_bf16 (+ a second source) followed by dc_gn_finalize, in one launch: one workgroup
 * per (sample, group) reads its channels of cat[x1, x2] directly.  For small maps (<= 16x16), where the two dependent
 * launches dominate.  Group width and C1, C2 must be even. */
int dc_gn_direct_nhwc_bf16(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                           float* ab, int N, long long HW, int groups, float eps, void* stream);
/* y = (x*a+b) [SiLU]; x = cat[x1,x2] NHWC bf16. */
int dc_gn_apply_nhwc_bf16(const void* x1, int C1, const void* x2, int C2, const float* ab, void* y,
                          int N, long long HW, int silu, void* stream);
/* FDN modulation — control_utils.py:33: y = (x*a+b)*(1+gamma)+beta; gamma/beta NHWC bf16 [Bp,HW,C] (sample n uses n % Bp). */
int dc_fdn_modulate_nhwc_bf16(const void* x, const float* ab, const void* gamma, const void* beta, void* y,
                              int N, int Bp, long long HW, int C, void* stream);
/* nn.LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3). x [M][C] bf16. */
int dc_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, long long M, int C, float eps,
                      void* stream);

/* ------------------------------------------------------------------ attention */
/* softmax(Q K^T * scale) V, flash-style, heads interleaved on the channel dim (diffusers Attention /
 * F.scaled_dot_product_attention inside BasicTransformerBlock.attn1/attn2).  q [B][Nq][q_stride] etc.;
 * head h occupies columns [h*D, (h+1)*D).  D in {8,16,32,40,64,80,128,160}. */
int dc_attention_bf16(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk, int D,
                      long long q_stride, long long k_stride, long long v_stride, long long o_stride, float scale,
                      void* stream);
/* Causal softmax(Q K^T * scale) V over a short context (T <= 128, D <= 128): CLIPTextModel self-attention behind
 * `encode_prompt` (pipeline.py:223-236).  Same operand layout as dc_attention_bf16; key j is visible to query i iff j <= i. */
int dc_attention_causal_small_bf16(const void* q, const void* k, const void* v, void* out, int B, int heads, int T, int D,
                                   long long q_stride, long long k_stride, long long v_stride, long long o_stride,
                                   float scale, void* stream);
/* CLIPTextEmbeddings: out[b][t][:] = tok_emb[ids[b][t]][:] + pos_emb[t][:]  (ids int64 [B][T]; tables and out bf16). */
int dc_embed_tokens_bf16(const long long* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int T, int C,
                         int vocab, void* stream);
/* Row softmax fp32 -> bf16 (VAE single-head attention, d=512, done as GEMM/softmax/GEMM). */
int dc_softmax_rows_f32_to_bf16(const float* s, void* p, long long rows, int cols, float scale, void* stream);

/* ------------------------------------------------------------------ elementwise */
/* time_proj — flownet.py:74: sinusoidal embedding of the timestep t_dev[step_dev ? *step_dev : 0] (device-resident so
 * that a captured hipGraph of one denoising step can be replayed for every step). */
int dc_timestep_embedding_f32(const float* t_dev, const int* step_dev, float* out, int n, int dim, void* stream);
/* FreeU — pipe.enable_freeu(s1,s2,b1,b2) (validation.py:106) -> diffusers apply_freeu [recalled]: skip features of up-blocks
 * 0/1 get their 2x2 lowest-frequency block scaled by s (fourier_filter, threshold 1); the backbone's first C/2 channels get *b. */
int dc_freeu_lowfreq_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, float s, void* stream);
int dc_freeu_backbone_nhwc_bf16(const void* x, void* y, long long pixels, int C, float b, void* stream);
/* y = a*x0 + b*x1 + c*x2 + d*x3 on fp32 tensors (x1..x3 may be NULL): `scheduler.step` of the multistep schedulers the reference
 * instantiates (UniPCMultistepScheduler, validation.py:37) and the CFG combine of the generic loop (pipeline.py:370-375). */
int dc_lincomb4_f32(const float* x0, const float* x1, const float* x2, const float* x3, float a, float b, float c, float d,
                    float* y, long long n, void* stream);
/* dst[c][r] = src[r][c], bf16, batch of `batch` matrices (VAE attention: V -> V^T). */
int dc_transpose_bf16(const void* src, void* dst, int batch, int R, int C, void* stream);
/* DiagonalGaussianDistribution.sample() * scale — train_controlnet.py:1081, pipeline.ipynb cell 7:
 * moments NHWC fp32 [N,h,w,2*C] (mean | logvar), noise NCHW fp32 -> latents NCHW fp32. */
int dc_vae_sample_latents(const float* moments, const float* noise, float* latents, float scale, int N, int C, int H, int W, void* stream);
int dc_silu_f32(const float* x, float* y, long long n, void* stream);
int dc_add_bf16(const void* a, const void* b, void* y, long long n, void* stream);
int dc_add_f32(const float* a, const float* b, float* y, long long n, void* stream);   /* P + W pyramids, flow_resnet.py:90 */
/* CFG combine + DDIM step — pipeline.py:370-375.  eps fp32 NHWC [cfg?2B:B][h][w][4]; latents fp32 NCHW in/out;
 * coef_dev [steps][4] = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)}; step_dev int32 counter (incremented).
 * Also writes the next model input NHWC bf16 [cfg?2B:B][h][w][C]. */
int dc_cfg_ddim_step(const float* eps, float* latents, void* model_in, const float* coef_dev, int* step_dev,
                     float guidance, int cfg, int B, int C, int H, int W, void* stream);
/* CFG combine + one UniPCMultistepScheduler.step (the scheduler validation.py:37 instantiates; bh2, predict_x0, order <= 2) in
 * one pass, for the captured denoising step of pipeline.py:308-385: same operand conventions as dc_cfg_ddim_step; m0 / m1 / last
 * fp32 [B,C,H,W] scheduler state (the two most recent x0-predictions, the previous predictor's start sample), updated in place;
 * coef_dev [steps][12] = {1/alpha_i, -sigma_i/alpha_i, flags (1 corrector | 2 corrector uses m1 | 4 predictor uses m1),
 * corrector coefficients of (last, m0, m_t, m1), predictor coefficients of (x, m_t, m1), 0, 0}.  Arithmetic = the
 * dc_lincomb4_f32 sequence of the scheduler's generic `step`, bit for bit. */
int dc_cfg_unipc_step(const float* eps, float* latents, float* m0, float* m1, float* last, void* model_in,
                      const float* coef_dev, int* step_dev, float guidance, int cfg, int B, int C, int H, int W, void* stream);
/* latents NCHW f32 [B,C,H,W] * mul -> NHWC bf16 [rep*B,H,W,C] (pipeline.py:313-320, :391 scaling) */
int dc_latents_to_model_input(const float* latents, void* model_in, float mul, int rep, int B, int C, int H, int W, void* stream);
/* image_processor.postprocess — pipeline.py:397-398: (x/2+0.5).clamp(0,1); x NHWC f32 [N,H,W,3] -> NCHW f32 and/or NHWC u8 */
int dc_postprocess_image(const float* x, float* out_nchw_f32, uint8_t* out_nhwc_u8, int N, int C, int H, int W,
                         int x_pixel_stride /* input elements per pixel, 0 = C (4 when conv_out ran with a padded 4th channel) */, void* stream);

/* ------------------------------------------------------------------ input side (controlnet/utils.py) */
/* resize_flow_to (utils.py:21-28) on the .flo payload layout: src [H][W][2] fp32 (pixel units) -> dst [2][th][tw] fp32,
 * bilinear with align_corners=True, then u *= tw/W and v *= th/H. */
int dc_flow_hw2_resize_scale_f32(const float* src_hw2, int H, int W, float* dst_2hw, int th, int tw, void* stream);
/* load_pair_to_sixch (utils.py:30-39) after the PIL resize: two RGB uint8 [H][W][3] images -> [6][H][W] fp32 in [0,1]
 * (TF.to_tensor's x/255, image 0 in planes 0-2, image 1 in planes 3-5). */
int dc_pack_sixch_u8_f32(const uint8_t* img0_hw3, const uint8_t* img1_hw3, float* dst_6hw, int H, int W, void* stream);

/* ------------------------------------------------------------------ tiled decode (patch_exp.ipynb / patch_utils.py) */
/* Blend of full-size decoded tiles into the frame with half-cosine ramps on inner tile edges (this package's
 * tiled-decode policy, tiling.merge_ramp; the reference's merge_costiles window is kept host-side for parity only).
 * tiles [T][C][th][tw] fp32 in [0,1]; coords int32 [T][4] = (y1, y2, x1, x2) with y2-y1 == th, x2-x1 == tw;
 * ramp [feather] fp32; out uint8 [H][W][C] = clip(rint(sum(scale*tile*w) / sum(w))); every pixel must be covered. */
int dc_blend_tiles_ramp_u8(const float* tiles_nchw, const int* coords_dev, int T, int C, int th, int tw,
                           const float* ramp_dev, int feather, uint8_t* out_hwc, int H, int W, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
