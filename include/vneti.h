/*
 * vneti.h — C ABI of libvneti_hip.so: the MI355X (gfx950) kernels behind the ViewNeTI
 * textual-inversion train step.
 *
 * The reference (jmhb0/view_neti) has no native/FFI boundary of its own: its hot path runs
 * through Python protocols into third-party diffusers/transformers modules
 * (training/coach.py:165-198,211-218; models/xti_attention_processor.py:9-57;
 * models/neti_clip_text_encoder.py:57-225; models/net_clip_text_embedding.py:34-137;
 * models/neti_mapper.py:165-197).  This header is the boundary we introduce *beneath* those
 * protocols; each entry point names the reference call it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only: raw device pointers, sizes, strides (in ELEMENTS unless noted);
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every function returns 0 on success or a negative VNETI_E* code; the message is
 *     retrievable with vneti_last_error().  Nothing throws or aborts across the ABI.
 *   - the library borrows pointers for the duration of the call only; it never allocates,
 *     frees or synchronises, so every call is hipGraph-capturable.
 *   - activations are channels-last: images are [B][H][W][C] f16, token matrices are
 *     [rows][C]; "ld" is the row (pixel) stride in elements so tensors can be views into
 *     wider buffers (this is how skip-connection concats are made free).
 */
#ifndef VNETI_H
#define VNETI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNETI_ABI_VERSION 1

int vneti_version(void);
/* The 16-bit storage / MFMA operand format this build of the library computes in: 0 = IEEE half (libvneti_hip.so, the
 * reference's `optim.mixed_precision: fp16`, training/coach.py:792-794), 1 = bfloat16 (libvneti_hip_bf16.so, its
 * `mixed_precision: bf16` branch, training/coach.py:796-802).  Every `const void*` activation / weight operand of the
 * entry points below (named ..._f16 for the default build) holds that format; f32 operands are unaffected. */
int vneti_precision(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int vneti_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM 3x3 convolution (MFMA).  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)
 * Replaces: torch.nn.functional.conv2d / linear issued by diffusers' ResnetBlock2D,
 * Downsample2D, Upsample2D, Transformer2DModel, CrossAttention.to_{q,k,v,out}
 * (models/xti_attention_processor.py:30-42,53) and transformers' CLIPEncoder linears
 * (models/neti_clip_text_encoder.py:111-118), forward and input-gradient.
 * ------------------------------------------------------------------------------------------ */
typedef struct vneti_gemm_desc {
  const void* A;      /* f16: plain [M][lda] matrix, or NHWC image when conv_mode != 0 */
  const void* B;      /* f16 weights, [N][ldb] with K contiguous ("NT" form) */
  void* C;            /* f16 (or f32 when out_f32) [M][ldc] */
  long long lda, ldb, ldc;
  int M, N, K;        /* K % 64 == 0 */
  int batch;          /* grid.y batches (0/1 = none) with element strides below */
  long long strideA, strideB, strideC;
  const float* bias;  /* f32 [N] or NULL, added before activation and rounding */
  const void* rowadd; /* f16 [M / rows_per_group][ld_rowadd] or NULL: per-row-group broadcast add
                         (ResnetBlock2D's "+ time_emb_proj(...)[:, :, None, None]") */
  long long ld_rowadd;
  int rows_per_group;
  const void* resid;  /* same dtype/shape as C (row stride ldr) or NULL: fused residual add */
  long long ldr;
  float alpha;
  int act;            /* 0 none, 1 SiLU, 2 quick-GELU, 3 exact GELU */
  int out_f32;
  /* implicit convolution: A is an NHWC image [Bn][Hi][Wi][ldx], K = 9*Ci ordered (tap, ci),
     M = Bn*Ho*Wo.  conv_mode 1: forward gather  in(oy*stride+dy-pad_t, ox*stride+dx-pad_l),
     optionally through a fused nearest-2x upsample (ups: Hi,Wi are the LOW-res dims);
     conv_mode 2: transposed gather for dgrad  in((oy+pad_t-dy)/stride, (ox+pad_l-dx)/stride). */
  int conv_mode;
  int Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l, ups;
  long long ldx;
  int tile_hint;      /* 0 = heuristic; 1: 128x128, 2: 128x64, 3: 64x64, 4: 256x128 (8 waves), 5: 256x256 (16 waves, f16 out),
                         6: 256x128 with a 3-stage LDS ring and cross-barrier fragment prefetch,
                         7: the same ring with 16 waves (64x32 wave tiles), 8: 256x128 / 16 waves, 9: 128x128 / 8 waves,
                         10 / 11 / 12: the 3-stage ring on 128x128 (8 waves) / 128x64 / 64x64 (under-filled grids: 2 stages in flight);
                         13 / 14 / 15: the same three tiles with a 4-stage ring (three stages in flight: short-K, latency-bound launches);
                         16 / 17: 256x256 / 256x128 as 8 waves in the 8-phase ping-pong structure (csrc/gemm8.hip; f16 out,
                         convs without fused upsampling; other launches fall back to 5 / 7);
                         18: the halo-patch form of 17 for stride-1 pad-1 3x3 convolutions (conv_mode 1, or 2 = the transposed
                         gather of the input gradient) with chunk-major K (conv_korder 1) on a 16-pixel grid: a block owns 16 x 16 output pixels and keeps the 18 x 18 input
                         patch of a 64-channel chunk in LDS for all nine taps (bit-identical to 17; split-K in whole 64-channel
                         chunks; other launches fall back to 17);
                         +100 selects the register-staged (non LDS-DMA) reference variant */
  /* split-K: f32 partials go to `workspace` (>= split_k*batch*M*N*4 bytes) and a second kernel
     reduces them and applies the epilogue.  split_k 0 = heuristic (only if a workspace is given),
     1 = off. */
  void* workspace;
  long long workspace_bytes;
  int split_k;
  /* f16-output epilogue extensions (non-batched), applied after bias/act/rowadd/resid:
     gate: value *= act'(gate_src[m][n]) (gate_act 1..3) -- the activation backward of a saved
     pre-activation tensor fused into the producing dgrad GEMM (transformers CLIPMLP backward);
     C2: a second f16 output C2[m][n] = act2(value written to C) (fc1 writes both the
     pre-activation kept for backward and the activated tensor). */
  const void* gate_src;
  long long ld_gate;
  int gate_act;
  void* C2;
  long long ldc2;
  int act2;
  /* GroupNorm statistics of the f16 output fused into the epilogue (split_k must be 1): the rows are images
     of gn_hw pixels, the N columns gn_groups groups of gn_cpg channels; every tile adds its (sum, sum of
     squares) per (image, group) to gn_sums[image][tile_m % gn_slots][group][4] — 64-bit fixed-point words
     (sum.hi, sum.lo, sumsq.hi, sumsq.lo; csrc/common.h vn_fx_encode / vn_fx_decode) added with INTEGER atomics, so the
     totals are bit-identical from run to run whatever order the tiles finish in.  The caller
     zeroes gn_sums; vneti_groupnorm_fwd_sums consumes it (the statistics pass of ResnetBlock2D.norm2 etc.
     without re-reading the tensor). */
  void* gn_sums;
  int gn_hw, gn_cpg, gn_groups, gn_slots;
  /* GEGLU (diffusers FeedForward: h, g = proj(x).chunk(2); h * gelu(g)) fused into the epilogue (f16 output, no
     split-K, N % 8 == 0).  The projection's output columns are INTERLEAVED in groups of four, [h0..h3 g0..g3 h4..h7
     g4..g7 ...] (the caller permutes the weight rows once at pack time), so one 16-byte chunk holds matching halves:
       geglu = 1 (forward, the ff.net.0.proj GEMM): C receives the pre-activation in that layout (kept for the
                 backward) and C2[m][n/2 + e] = C[m][n+e] * gelu(C[m][n+4+e]), e < 4 — the [M][N/2] input of ff.net.2;
       geglu = 2 (backward, the ff.net.2 dgrad GEMM whose result d(h*gelu(g)) is [M][N]): gate_src is the saved
                 pre-activation [M][2N]; C is [M][2N] in the same interleaved layout and receives
                 d_h = d * gelu(g), d_g = d * h * gelu'(g); the [M][N] GEMM result itself is not stored. */
  int geglu;
  /* K order of an implicit convolution (B must be packed to match): 0 = (tap, channel); 1 = (64-channel chunk, tap,
     channel in chunk): the nine taps of a chunk are consecutive k-steps, so the im2col re-reads of an input pixel
     hit L1/L2 instead of coming back after a whole channel sweep. */
  int conv_korder;
} vneti_gemm_desc;

int vneti_gemm_f16(const vneti_gemm_desc* d, void* stream);
/* the tile configuration (1..4, see tile_hint) the heuristic picks for a problem size */
int vneti_gemm_select_tile(int M, int N, int batch);
/* the split-K factor split_k = 0 resolves to for a problem, tile configuration and workspace size (1 = no split) */
int vneti_gemm_select_split(int M, int N, int K, int batch, int tile_hint, long long workspace_bytes);

/* 3x3 im2col for convolutions with tiny Cin (conv_in of UNet / VAE, dgrad of conv_out):
 * out[m][tap*C + c] (f16, row length 64, zero padded) from an arbitrarily strided image.
 * Replaces the first conv2d of diffusers' UNet2DConditionModel / AutoencoderKL encoder. */
int vneti_im2col3x3_small(const void* x, int x_is_f32, long long sb, long long sc, long long sy,
                          long long sx, void* out, int Bn, int C, int Hi, int Wi, int Ho, int Wo,
                          int stride, int pad_t, int pad_l, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of an image with C <= 3 channels, straight from the strided f32 / f16 pixels to the
 * NHWC f16 output rows (AutoencoderKL encoder.conv_in, diffusers vae.py; reached from training/coach.py:165-169): no
 * im2col matrix.  `w_packed` = [Co][32] f16 with k = tap * C + c and the rows of every 128-channel block permuted as
 * view_neti_amd.packing.conv_in_direct does; Co % 128 == 0.  `gn_sums` (optional) accumulates the GroupNorm (sum, sum of
 * squares) of the stored values, layout and meaning as vneti_gemm_desc.gn_sums with gn_hw = H * W. */
int vneti_conv3x3_in(const void* x, int x_is_f32, long long sb, long long sc, long long sy, long long sx,
                     const void* w_packed, const float* bias, void* out, long long ldo, int Bn, int C, int H, int W,
                     int Co, void* gn_sums, int gn_groups, int gn_slots, void* stream);

/* batched 2-D transpose of f16 matrices: out[b][c][r] = in[b][r][c]; columns of `out` in
 * [rows, ld_out) are zero filled (attention kernels read K^T / V^T / Q^T / dO^T tiles). */
int vneti_transpose_f16(const void* in, long long ld_in, long long stride_in, void* out,
                        long long ld_out, long long stride_out, int rows, int cols, int batch,
                        void* stream);
/* up to VNETI_TRANSPOSE_MAX independent transposes in ONE launch (the Q^T / K^T / dO^T operand copies of
 * an attention backward): fields as the arguments of vneti_transpose_f16. */
#define VNETI_TRANSPOSE_MAX 4
typedef struct vneti_transpose_desc {
  const void* in;
  long long ld_in, stride_in;
  void* out;
  long long ld_out, stride_out;
  int rows, cols, batch, _pad;
} vneti_transpose_desc;
int vneti_transpose_f16_multi(const vneti_transpose_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) on NHWC f16, statistics in f32.
 * Replaces: diffusers ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm,
 * conv_norm_out (driven from training/coach.py:197), forward and input-gradient.
 * ws: f32 workspace of vneti_groupnorm_ws_floats(...) floats.
 * ------------------------------------------------------------------------------------------ */
long long vneti_groupnorm_ws_floats(int Bn, int HW, int C, int G);
int vneti_groupnorm_fwd(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                        const float* beta, float* mean, float* rstd, float* ws, int Bn, int HW,
                        int C, int G, float eps, int silu, void* stream);
/* GroupNorm(+SiLU) forward whose statistics pass already ran inside the GEMM that produced x
   (vneti_gemm_desc.gn_sums, `slots` slots per sample): one launch that finishes mean / rstd from the sums,
   normalises, and publishes mean / rstd for the backward. */
int vneti_groupnorm_fwd_sums(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                             const float* beta, const void* sums, int slots, float* mean, float* rstd,
                             int Bn, int HW, int C, int G, float eps, int silu, void* stream);
/* The same forward / backward in TWO launches for tensors beyond the one-block-per-group kernel: the statistics pass adds
   its slab sums to `sums` ([Bn][slots][G][4] 64-bit words, zeroed by the caller before the launch, layout of
   vneti_gemm_desc.gn_sums) and the apply kernel finishes them itself (no finalize launch).  Small tensors run the
   one-launch kernel of vneti_groupnorm_fwd / _bwd and leave `sums` alone (`ws` of the backward is only used there). */
int vneti_groupnorm_fwd_2l(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                           const float* beta, void* sums, int slots, float* mean, float* rstd, int Bn, int HW,
                           int C, int G, float eps, int silu, void* stream);
int vneti_groupnorm_bwd_2l(const void* dy, long long lddy, const void* x, long long ldx, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, void* dx, long long lddx,
                           const void* dx_accum, long long ldacc, void* sums, int slots, float* ws, int Bn,
                           int HW, int C, int G, int silu, void* stream);
/* dx = d(loss)/d(x) given dy = d(loss)/d(y); y = silu?(GN(x)).  If dx_accum != NULL it is
 * added (f16, row stride lddx) — used where two gradient paths meet. */
int vneti_groupnorm_bwd(const void* dy, long long lddy, const void* x, long long ldx,
                        const float* gamma, const float* beta, const float* mean,
                        const float* rstd, void* dx, long long lddx, const void* dx_accum,
                        long long ldacc, float* ws, int Bn, int HW, int C, int G, int silu,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim of [rows][C]; x is f16 or f32, y is f16.
 * Replaces: BasicTransformerBlock.norm1/2/3 and CLIPEncoderLayer.layer_norm1/2,
 * final_layer_norm (models/neti_clip_text_encoder.py:183-185).
 * ------------------------------------------------------------------------------------------ */
int vneti_layernorm_fwd(const void* x, int x_is_f32, long long ldx, void* y, long long ldy,
                        const float* gamma, const float* beta, float* mean, float* rstd, int rows,
                        int C, float eps, void* stream);
/* dx (+= dx_accum) ; dx/dx_accum dtype selected by dx_is_f32.  dx_f16_copy (optional) receives the same
 * values rounded to f16: the operand of the next dgrad GEMM when dx itself is the f32 residual stream. */
int vneti_layernorm_bwd(const void* dy, int dy_is_f32, long long lddy, const void* x,
                        int x_is_f32, long long ldx, const float* gamma, const float* mean,
                        const float* rstd, void* dx, int dx_is_f32, long long lddx,
                        const void* dx_accum, long long ldacc, void* dx_f16_copy, long long ldcopy,
                        int rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused (flash-style) multi-head attention, head_dim in {40, 64, 80, 160}.
 *   O[b,q,h,:] = softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,:]
 * Q/K/V/O (and dO, dQ, dK, dV) rows are token rows of [B*N][ld] matrices with head h at column h*D;
 * no transposed operand copies are needed: operands an MFMA wants key-/query-major are read
 * transposed out of the row-major LDS tiles (ds_read_b64_tr_b16).
 * K and V come from separate tensors: this is the XTI contract
 * (models/xti_attention_processor.py:38-42: key from CONTEXT_TENSOR_l, value from
 * CONTEXT_TENSOR_BYPASS_l).  Replaces attn.get_attention_scores + torch.bmm (:48-49)
 * without materialising the score matrix.  lse: f32 [B][H][Nq] (natural log).
 * ------------------------------------------------------------------------------------------ */
int vneti_attn_fwd(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                   long long ldv, void* O, long long ldo, float* lse, int Bn, int H, int Nq,
                   int Nk, int D, float scale, int causal, void* stream);
/* delta[b][h][q] = sum_d dO*O  (f32) */
int vneti_attn_bwd_delta(const void* dO, long long lddo, const void* O, long long ldo,
                         float* delta, int Bn, int H, int Nq, int D, void* stream);
/* dQ.  With O == NULL, delta is an input (from vneti_attn_bwd_delta).  With O given, the kernel computes
 * delta itself from the dO rows it already holds and WRITES it to `delta` for the dK/dV kernel that
 * follows — one launch fewer per attention. */
int vneti_attn_bwd_dq(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                      long long ldv, const void* dO, long long lddo, const float* lse, float* delta,
                      const void* O, long long ldo, void* dQ, long long lddq, int Bn, int H, int Nq,
                      int Nk, int D, float scale, int causal, void* stream);
/* dK, dV.  ws (optional): f32 scratch for the query-split partials of short-key (cross) attention. */
int vneti_attn_bwd_dkv(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                       long long ldv, const void* dO, long long lddo, const float* lse,
                       const float* delta, void* dK, long long lddk, void* dV, long long lddv,
                       int Bn, int H, int Nq, int Nk, int D, float scale, int causal, float* ws,
                       long long ws_floats, void* stream);
/* ws (optional f32 scratch): when the key side is short (cross-attention, Nk = 77) the query
 * range is split across workgroups and the f32 partials are reduced by a second kernel;
 * needs 2*qsplit*Bn*Nk*H*D floats (qsplit <= 32), NULL disables the split. */
/* dQ, dK and dV of a short self-attention in ONE launch: N <= 96 and D = 64 (the 77-token CLIP text encoder that
 * training/coach.py:289-305 runs once per UNet cross-attention layer, models/neti_clip_text_encoder.py:90-118), one
 * workgroup per (sequence, head) with Q, K, V, dO in LDS; bit-identical to vneti_attn_bwd_dq (O given) followed by
 * vneti_attn_bwd_dkv.  Returns VNETI unsupported (-2) for other sizes: the caller then uses the two kernels above. */
int vneti_attn_bwd_small(const void* Q, long long ldq, const void* K, long long ldk, const void* V, long long ldv,
                         const void* dO, long long lddo, const void* O, long long ldo, const float* lse, void* dQ,
                         long long lddq, void* dK, long long lddk, void* dV, long long lddv, int Bn, int H, int N, int D,
                         float scale, int causal, void* stream);
/* row softmax in place on f16 [rows][ld] (unfused attention of the VAE mid-block, d=512) */
int vneti_softmax_rows_f16(void* x, long long ld, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / small kernels
 * ------------------------------------------------------------------------------------------ */
/* out = a + b on f16 [rows][cols] views (gradient accumulation where two paths meet) */
int vneti_add_f16(const void* a, long long lda, const void* b, long long ldb, void* out,
                  long long ldo, int rows, int cols, void* stream);
/* diffusers GEGLU (FeedForward.net.0): p = [h | g], out = h * gelu(g); and its backward */
int vneti_geglu_fwd(const void* p, long long ldp, void* out, long long ldo, int rows, int C4,
                    void* stream);
int vneti_geglu_bwd(const void* dy, long long lddy, const void* p, long long ldp, void* dp,
                    long long lddp, int rows, int C4, void* stream);
/* y = act(x) / dx = dy*act'(x), contiguous f16, act: 1 SiLU, 2 quick-GELU, 3 GELU (CLIP MLP) */
int vneti_act_fwd_f16(const void* x, void* y, long long n, int act, void* stream);
int vneti_act_bwd_f16(const void* dy, const void* x, void* dx, long long n, int act, void* stream);
/* diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): t int64 [B] -> f16 [B][dim] */
int vneti_timestep_embedding(const void* t, void* out, int Bn, int dim, void* stream);
/* backward of nearest-2x upsample: out[b][y][x] = sum of the 2x2 block of in (NHWC f16) */
int vneti_sum2x2_f16(const void* in, long long ldi, void* out, long long ldo, int Bn, int H, int W,
                     int C, void* stream);
/* device RNG (counter hash + Box-Muller) so the step is graph-capturable:
 * state = uint32[2] {seed, step counter}.  Replaces torch.randn_like / torch.randint of
 * training/coach.py:172-178 (stream differs from torch's Philox: documented deviation). */
int vneti_rng_fill_normal(void* out_f32, long long n, const void* state, unsigned stream_id,
                          void* stream);
int vneti_rng_fill_randint(void* out_i64, int n, int high, const void* state, unsigned stream_id,
                           void* stream);
int vneti_rng_advance(void* state, void* stream);
/* nested dropout of the mapper's hidden vector (models/neti_mapper.py:401-414): per mapper call
 * (= per UNet layer) one Bernoulli(prob) draw; when it fires each sample gets a truncation index
 * ~ U{0..hidden-1}.  mask: f32 [nl*Bn][hidden] of 0/1, the `hidden_mask` of vneti_mapper_fwd/_bwd. */
int vneti_nested_dropout_mask(float* mask, int nl, int Bn, int hidden, float prob, const void* state,
                              unsigned stream_id, void* stream);
/* latent_dist.sample() * scaling_factor, DDPMScheduler.add_noise and the loss target
 * (training/coach.py:167-183,201-205) in one kernel.  moments: NHWC f16 [B][HW][ldm] (mean|logvar);
 * eps, noise, outputs: NCHW f32 [B][Lc][HW]. */
int vneti_sample_add_noise(const void* moments, long long ldm, const float* eps,
                           const float* noise, const void* timesteps_i64,
                           const float* alphas_cumprod, float scaling, int v_prediction,
                           float* latents, float* noisy, float* target, int Bn, int Lc, int HW,
                           void* stream);
/* The same three steps as separate entry points (identical arithmetic), for callers that keep the reference's
 * module-call order: `vae.encode(x).latent_dist.sample()` (training/coach.py:165-169; scaling = 1 leaves the
 * multiplication by vae.config.scaling_factor to the caller) ... */
int vneti_latent_sample(const void* moments, long long ldm, const float* eps, float scaling, float* latents, int Bn,
                        int Lc, int HW, void* stream);
/* ... and `noise_scheduler.add_noise(latents, noise, timesteps)` / `.get_velocity(...)` (training/coach.py:182-183,
 * 201-205): noisy and target are NCHW f32 like the inputs; either may be null; target = noise (epsilon) or the
 * velocity (v_prediction). */
int vneti_add_noise(const float* latents, const float* noise, const void* timesteps_i64, const float* alphas_cumprod,
                    int v_prediction, float* noisy, float* target, int Bn, int Lc, int HW, void* stream);
/* Inference (sd_pipeline_call.py:72-103): classifier-free guidance on the CFG-batched UNet output (rows
 * [0,B*HW) unconditional, [B*HW,2B*HW) conditional) and one sampler step in data-prediction form
 *   e = u + g (c - u);  x0 = (x - sigma_t e)/alpha_t  (epsilon)  |  alpha_t x - sigma_t e  (v_prediction)
 *   x <- cx x + c0 x0 + c1 m_prev;  m_prev <- x0;  x_in[2B] <- x (both CFG halves of the next UNet input)
 * which covers DPM-Solver++(2M) (DPMSolverMultistepScheduler, training/validate.py:568) and DDIM (eta = 0);
 * the per-step scalars come from the host (view_neti_amd/engine/infer.py::step_coefficients).
 * x, m_prev: f32 NCHW [B][Lc][HW]; x_in: f32 NCHW [2B][Lc][HW]; pred: f16 NHWC. */
int vneti_cfg_sampler_step(const void* pred, long long ldp, float* x, float* m_prev, float* x_in, int Bn,
                           int Lc, int HW, float guidance, float alpha_t, float sigma_t, float cx,
                           float c0, float c1, int v_prediction, void* stream);
/* hipGraph-replayable form of the sampler step: the scalars {alpha_t, sigma_t, cx, c0, c1} are row step[0]
 * of a device table (T x 5 floats) and the UNet/text timestep inputs are filled from a device table of the
 * sampler's timesteps, so one captured step is replayed for every timestep; vneti_counter_advance moves on. */
int vneti_cfg_sampler_step_table(const void* pred, long long ldp, float* x, float* m_prev, float* x_in,
                                 int Bn, int Lc, int HW, float guidance, const float* coef_table,
                                 const int* step, int v_prediction, void* stream);
int vneti_table_fill_i64(void* dst_i64, int n, const void* table_i64, const int* step, void* stream);
int vneti_counter_advance(int* counter, void* stream);
/* AutoencoderKL.post_quant_conv on the latents scaled by 1/scaling_factor (pipeline.decode_latents):
 * NCHW f32, out[b][o][p] = bias[o] + sum_c W[o][c] x[b][c][p] * in_scale; channel counts <= 8 */
int vneti_conv1x1_nchw_f32(const float* x, const float* W, const float* bias, float* out, int Bn, int Ci,
                           int Co, int HW, float in_scale, void* stream);
/* pipeline.decode_latents tail (sd_pipeline_call.py:115): (img/2 + 0.5).clamp(0,1), NHWC f16 -> f32 [n_pix][ch] */
int vneti_image_postprocess(const void* img, long long ldi, float* out, long long n_pix, int channels,
                            void* stream);
/* F.mse_loss(pred.float(), target.float()) partial sum (+= into loss_sum[0]) and the scaled
 * gradient seed dpred = 2 (pred - target) / N * loss_scale[0]  (training/coach.py:211-214) */
int vneti_mse_loss_grad(const void* pred, long long ldp, const float* target, void* dpred,
                        long long lddp, float* loss_sum, const float* loss_scale, int Bn, int Lc,
                        int HW, void* stream);
/* torch.optim.AdamW over one flat f32 bucket with torch.cuda.amp.GradScaler semantics
 * (training/coach.py:214-218,750-756).  hyper = {lr, beta1, beta2, eps, weight_decay, grad_div};
 * scaler = {loss_scale, growth_tracker, found_inf}; step = int32 optimizer step count.
 * phases (bit mask) lets one optimizer step span several buckets: VNETI_OPT_CHECK = unscale-time
 * inf/nan check of this bucket (sets found_inf), VNETI_OPT_APPLY = the update (skipped when
 * found_inf), VNETI_OPT_FINISH = GradScaler.update() + step advance.  One bucket: pass all three.
 * growth_interval <= 0 = static loss scale (bf16: accelerate builds no GradScaler, training/coach.py:796-802):
 * a non-finite step is skipped but the scale is neither halved nor grown.
 * Several: CHECK every bucket, then APPLY every bucket, FINISH on the last call. */
#define VNETI_OPT_CHECK 1
#define VNETI_OPT_APPLY 2
#define VNETI_OPT_FINISH 4
#define VNETI_OPT_ALL 7
int vneti_adamw_flat(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                     float* scaler, int* step, int growth_interval, int phases, void* stream);
/* The same optimizer over a bucket of n_seg equal-length segments, one per mapper (learnable_mode 3:
 * view mapper + one object mapper per scene, training/coach.py:505-598,750-756), keeping torch's
 * per-parameter semantics: a segment enters the update set the first time it is `active` (has a
 * gradient) and is stepped on every later iteration — with a zero gradient when inactive, because
 * zero_grad() keeps zero tensors — using its own seg_step[] for the bias correction.
 * active: device int32[n_active] segment ids that received gradients this step. */
int vneti_adamw_segments(float* p, const float* g, float* m, float* v, long long seg_len, int n_seg,
                         int* seg_step, const int* active, int n_active, const float* hyper,
                         float* scaler, int* step, int growth_interval, int phases, void* stream);

/* ------------------------------------------------------------------------------------------
 * NeTI text path
 * ------------------------------------------------------------------------------------------ */
/* NeTIMapper (arch_view_net=15) for R = n_layers*batch rows at once
 * (models/neti_mapper.py:165-197): data[R][nfeat] are the [-1,1]-scaled conditioning inputs
 * (t, l[, 12 camera params]); w_enc[enc_dim/2][nfeat] the Fourier frequencies
 * (models/positional_encoding.py:146-195); params the flat f32 bucket in state_dict order
 * (net.0.{weight,bias}, net.1.*, net.3.*, net.4.*, output_layer.0.*).  hidden_mask (optional,
 * [R][hidden] of 0/1) expresses nested dropout / truncation (:401-414).  norm_scale <= 0 disables
 * the output normalisation (:434-436).  word/bypass: f32 [R][D].  slot (optional device int32[1])
 * selects one mapper of a bucket holding several (`mapper_object_lookup`,
 * models/net_clip_text_embedding.py:65-76): params (and, in the backward, grads) are offset by
 * slot[0]*slot_stride floats on the device, so the choice does not break hipGraph replay. */
long long vneti_mapper_num_params(int enc_dim, int hidden, int D, int has_bypass);
long long vneti_mapper_save_floats(int R, int enc_dim, int hidden);
long long vneti_mapper_rowgrad_floats(int R, int hidden, int D, int has_bypass);
int vneti_mapper_fwd(const float* params, const int* slot, long long slot_stride, const float* data,
                     int nfeat, const float* w_enc,
                     const float* hidden_mask, float norm_scale, float* word, float* bypass,
                     float* save, int R, int enc_dim, int hidden, int D, int has_bypass, const float* enc_in,
                     void* stream);
/* enc_in (optional, f32 [R][enc_dim]): the first layer's input is given instead of being the Fourier encoding of
 * `data` (then data / w_enc may be NULL) — the legacy mapper of arch_view_net <= 14, whose first-layer input is the
 * output of its trainable input_layer (vneti_mapper_legacy_input_fwd below). */
/* parameter gradients (the only wgrad of the whole train step).  d(word) for mapper row r is
 * read from dword_src + dword_rows[r]*ld_src (the placeholder rows of the embedding gradient). */
int vneti_mapper_bwd(const float* params, const int* slot, long long slot_stride,
                     const float* hidden_mask, float norm_scale,
                     const float* word, const float* dword_src, const int* dword_rows,
                     long long ld_src, const float* dbypass, const float* save, float* rowgrads,
                     float* grads, int accumulate, int R, int enc_dim, int hidden, int D,
                     int has_bypass, float* denc, void* stream);
/* denc (optional, f32 [R][enc_dim]) receives d(loss)/d(first-layer input): what the legacy path back-propagates
 * into its input_layer.
 *
 * Legacy object mapper (the reference's dataclass default arch_view_net = 0; models/positional_encoding.py:10-51,
 * models/neti_mapper.py:155-163,200-206): e[(l,b)] = W_in v + b_in with v = cat[sin(w x), cos(w x)] / |.| of the RAW
 * x = (timestep_b, l), w_pe f32 [pe_dim/2][2] (NeTIPositionalEncoding.w), W_in f32 [enc_dim][pe_dim] initialised from
 * the 10 x 16 anchor encodings.  params_in / grads_in: [W_in | b_in], vneti_mapper_legacy_input_params() floats; `slot`
 * as above.  The backward recomputes v and writes (or accumulates) dW_in = sum_r denc[r] (x) v_r, db_in = sum_r denc[r];
 * nl*Bn <= 128 rows. */
long long vneti_mapper_legacy_input_params(int enc_dim, int pe_dim);
int vneti_mapper_legacy_input_fwd(const float* params_in, const int* slot, long long slot_stride,
                                  const void* timesteps_i64, const float* w_pe, float* enc_out, int nl, int Bn,
                                  int enc_dim, int pe_dim, void* stream);
int vneti_mapper_legacy_input_bwd(const void* timesteps_i64, const float* w_pe, const float* denc, float* grads_in,
                                  const int* slot, long long slot_stride, int accumulate, int nl, int Bn, int enc_dim,
                                  int pe_dim, void* stream);
/* NeTICLIPTextEmbeddings.forward (models/net_clip_text_embedding.py:34-137) for all layers:
 * X[(l,b,pos)][D] f32 = (pos == pos_obj[b] ? word_obj[(l,b)] : pos == pos_view[b] ? word_view : E[ids[b][pos]]) + P[pos] */
int vneti_text_embed(const float* tok_emb, const float* pos_emb, const void* ids_i64,
                     const int* pos_obj, const float* word_obj, const int* pos_view,
                     const float* word_view, float* X, int nl, int Bn, int L, int D, void* stream);
/* textual bypass + final_layer_norm on both variants (models/neti_clip_text_encoder.py:121-185):
 * ctx_k = LN(last), ctx_v = LN(last with the placeholder rows rewritten): constrained (:140-144)
 * x + alpha b/|b| |x|; unconstrained (:145-149) b/|b| * detach(mean_j |x_j|), object first, then view.
 * norm_terms: f32 [2][nl*Bn] scratch written by the forward and read by the backward (only
 * needed when an unconstrained flag is set). */
int vneti_text_final_fwd(const float* last, const float* gamma, const float* beta, float eps,
                         const int* pos_obj, const float* bypass_obj, float alpha_obj,
                         int unconstrained_obj, const int* pos_view, const float* bypass_view,
                         float alpha_view, int unconstrained_view, float* norm_terms, void* ctx_k,
                         void* ctx_v, int nl, int Bn, int L, int D, void* stream);
int vneti_text_final_bwd(const float* last, const float* gamma, float eps, const int* pos_obj,
                         const float* bypass_obj, float alpha_obj, int unconstrained_obj,
                         float* dbypass_obj, const int* pos_view, const float* bypass_view,
                         float alpha_view, int unconstrained_view, float* dbypass_view,
                         const float* norm_terms, const void* dctx_k, const void* dctx_v, float* dX,
                         int nl, int Bn, int L, int D, void* stream);
/* conditioning inputs of the mapper for every (layer, sample) row:
 * data[(l,b)] = [t_b/1000*2-1, l/nl*2-1, view_params[b][0..nv)]  (models/neti_mapper.py:545-562) */
int vneti_mapper_inputs(const void* timesteps_i64, const float* view_params, int nv, float* data,
                        int nl, int Bn, void* stream);
int vneti_cast_f32_f16(const float* x, void* y, long long n, void* stream);


/* ---- device-side input pipeline: the per-sample image work of training/dataset.py:238-316,700-740
   on uint8 HWC RGB images in HBM.  Each entry restates the Pillow / torchvision routine named beside it
   (csrc/image.hip); the random parameters are drawn on the host (compat/augment.py::draw_plan). ------------------- */
/* Resample.c precompute_coeffs + normalize_coeffs_8bpc for one axis; filter 0 = BICUBIC (dataset.py `_resize`),
   1 = BILINEAR (RandomResizedCrop).  bounds: 2*out_size ints, kk: out_size*ksize ints. */
int vneti_img_resample_ksize(int in_size, int out_size, int filter);
int vneti_img_resample_coeffs(int in_size, int out_size, int filter, int* bounds, int* kk, void* stream);
/* one pass of ImagingResampleHorizontal_8bpc (horizontal != 0) / Vertical_8bpc; in_w = row width of `in` */
int vneti_img_resample_pass(const void* in, int in_w, void* out, int out_h, int out_w, const int* bounds,
                            const int* kk, int ksize, int horizontal, void* stream);
/* Image.crop (+ Image.transpose(FLIP_LEFT_RIGHT), dataset.py:727-728) */
int vneti_img_crop(const void* in, int in_w, int top, int left, void* out, int h, int w, int flip, void* stream);
/* ImageEnhance in place: mode 0 Brightness, 1 Contrast (scratch8 = 8 bytes for the luma sum), 2 Color,
   3 RandomGrayscale (alpha unused) */
int vneti_img_enhance(void* img, int h, int w, int mode, float alpha, void* scratch8, void* stream);
/* torchvision adjust_hue on PIL images in place: H += shift (uint8 wrap) in Pillow's HSV */
int vneti_img_hue(void* img, int h, int w, int shift, void* stream);
/* GaussianBlur(5) with the host-computed kernel k5 (f32), tmp = h*w*3 floats */
int vneti_img_blur5(const void* in, void* out, float* tmp, int h, int w, const float* k5_host, void* stream);
/* Image.rotate(NEAREST, expand=False, fillcolor): Geometry.c affine_fixed with the six 16.16 coefficients */
int vneti_img_affine_nearest(const void* in, void* out, int h, int w, const int* a6_host, int fill, void* stream);
/* (uint8 / 127.5 - 1) -> f32 CHW plane set (dataset.py:738-739): writes straight into the VAE's input buffer */
int vneti_img_to_f32_chw(const void* img, float* out, int h, int w, void* stream);

/* ---- data-parallel exchange: ONE all-reduce(sum) of the flat f32 mapper-gradient bucket per
   optimisation step, on the caller's stream, straight into RCCL over xGMI (csrc/comm.hip).  Replaces the DDP all-reduce
   accelerate performs behind `accelerator.backward(loss)` for the wrapped text encoder (training/coach.py:97-99, 211-218);
   the mean is folded into vneti_adamw_flat's grad_div.  RCCL is resolved with dlopen on first use (a copy the process
   already holds is reused); there is no fallback — without it these entry points fail with VNETI_EUNSUP.
     vneti_comm_unique_id   rank 0: 128 opaque bytes to hand to every rank over any side channel (ncclGetUniqueId)
     vneti_comm_init        every rank, collectively: -> *comm (ncclCommInitRank on the current device)
     vneti_allreduce_flat   in place, sum over ranks, n floats; stream-ordered, no host sync
     vneti_comm_destroy     releases the communicator (NULL is fine) */
int vneti_comm_unique_id(void* id128);
int vneti_comm_init(const void* id128, int rank, int world, void** comm);
int vneti_allreduce_flat(void* comm, float* buf, long long n, void* stream);
int vneti_comm_destroy(void* comm);

/* ---- CU-partitioned streams (measurement aid): a stream whose kernels may only occupy the
   compute units of `mask` (hipExtStreamCreateWithCUMask; bit i = logical CU i, dealt round-robin over the XCDs by the driver).
   Built to test whether the NEXT batch's VAE encode (training/coach.py:165-169: frozen VAE, `.detach()` — no dependence on
   trainable state) hides beside the current step on a CU subset; measured: it does not (profiles/r06_cu_mask_probe.txt), so
   nothing on the step's path uses these.  Nothing in the reference corresponds: it encodes inline.
     vneti_stream_create_cu_mask   -> *stream (a hipStream_t as void*); nwords 32-bit words of mask
     vneti_stream_get_cu_mask      reads the mask the runtime holds for `stream` back
     vneti_stream_destroy          releases it (NULL is fine) */
int vneti_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream);
int vneti_stream_get_cu_mask(void* stream, unsigned* mask, int nwords);
int vneti_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VNETI_H */
