/* libgill_amd — C ABI of the MI355X-native GILL image-generation hot path.
 *
 * The reference (kohjingyu/gill) has no FFI seam: its "operator API" is the Python class surface
 * gill.models.{load_gill, GILL, GILLModel} + gill.layers.TextFcLayer, and every FLOP runs inside
 * transformers/diffusers/torch.  This header is the boundary the Python mirror (gill_amd/) binds
 * with ctypes; each entry point names the reference call it replaces.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers borrowed for the duration of the call (torch tensors'
 *    data_ptr()); the caller allocates every output.  Exceptions are marked "host".
 *  - Work is enqueued on the caller-supplied stream (a hipStream_t passed as void*; NULL = the
 *    default stream).  Nothing synchronises the device unless stated.
 *  - Every function returns 0 on success; on failure a negative code, with a thread-local message
 *    retrievable through gill_last_error().  No C++ exception crosses this boundary.
 *  - bf16 tensors are raw uint16 bit patterns.  Shapes are row-major, last index fastest.
 *  - Handles are not re-entrant (the reference serialises requests: demo/app_gradio.py:217).
 */
#ifndef GILL_AMD_H
#define GILL_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GILL_DTYPE_BF16 0
#define GILL_DTYPE_F32 1
#define GILL_DTYPE_F16 2

/* One named weight tensor (state-dict entry).  `data` is a device pointer; it is copied /
 * re-laid-out into handle-owned HBM during *_create and not referenced afterwards. */
typedef struct {
  const char* name;
  const void* data;
  int32_t dtype; /* GILL_DTYPE_* */
  int32_t ndim;
  int64_t shape[4];
} gill_tensor;

const char* gill_last_error(void);
int gill_version(void);

/* ------------------------------------------------------------------------------------------
 * Stage 1 — frozen OPT decoder.  Replaces self.lm(inputs_embeds=..., output_hidden_states=True)
 * of transformers OPTForCausalLM at gill/models.py:363-365 (batched) and :465 (generate loop).
 * State-dict names are those of OPTForCausalLM ("model.decoder.layers.N.self_attn.q_proj.weight").
 * ------------------------------------------------------------------------------------------ */
typedef struct gill_opt gill_opt;
typedef struct {
  int32_t vocab_size;   /* rows of embed_tokens after resize_token_embeddings (models.py:73) */
  int32_t hidden_size;  /* == word_embed_proj_dim (no project_in/out: every OPT but 350m) */
  int32_t num_layers;
  int32_t num_heads;
  int32_t ffn_dim;
  int32_t max_positions; /* 2048; learned positions carry the +2 offset */
  int32_t max_batch;     /* workspace sizing */
  int32_t max_seq;
} gill_opt_config;

int gill_opt_create(gill_opt** out, const gill_opt_config* cfg, const gill_tensor* weights, int n_weights);
void gill_opt_destroy(gill_opt* h);

/* input_embeddings(ids): models.py:180 / :620.  ids (B*T) int64 -> out (B*T, D) bf16. */
int gill_opt_embed(gill_opt* h, const int64_t* ids, int n, void* out_bf16, void* stream);

/* Full causal forward over right-padded sequences, no attention mask (models.py:363-365).
 *   inputs_embeds (B,T,D) bf16  ->  hidden_out (B,T,D) fp32 = hidden_states[-1] (post final LN).
 * hidden_out may be NULL when only gathered rows are wanted (see gill_opt_img_hidden). */
int gill_opt_forward(gill_opt* h, const void* inputs_embeds_bf16, int B, int T, float* hidden_out, void* stream);

/* KV-cached continuation of the same forward for the decode loop of GILLModel.generate (models.py:464-530; the reference
 * re-runs the whole sequence every step with use_cache unset).  inputs_embeds (B,T_new,D) bf16 are the tokens
 * past_len .. past_len+T_new-1 of each sequence; their keys/values are appended to the handle's cache and they attend to
 * cache[0, past_len) plus, causally, to each other.  past_len = 0 starts a new sequence (prefill).  The caller keeps B
 * fixed and past_len equal to the number of tokens already fed.  hidden_out (B,T_new,D) fp32 = hidden_states[-1] rows. */
int gill_opt_forward_cached(gill_opt* h, const void* inputs_embeds_bf16, int B, int T_new, int past_len, float* hidden_out,
                            void* stream);

/* The fast path of GILLModel.forward(mode='generation') (models.py:180-183, 384-385):
 *   ids (B,T) int64 right-padded, last_idx (B) int32 HOST array = caption_len-1;
 *   raw_out (B,8,D) bf16 = hidden_states[-1][i, last-7:last+1];  emb_out (B,8,D) bf16 = input_embs slice. */
int gill_opt_img_hidden(gill_opt* h, const int64_t* ids, const int32_t* last_idx_host, int B, int T, int num_tokens,
                        void* raw_out_bf16, void* emb_out_bf16, void* stream);

/* logits[:, -1, :] of the tied lm_head for the generate loop (models.py:470):
 *   hidden (B,T,D) fp32 from gill_opt_forward -> logits_out (B, vocab) fp32.  B <= 8. */
int gill_opt_last_logits(gill_opt* h, const float* hidden, int B, int T, float* logits_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image prompts — the frozen CLIP vision tower (transformers CLIPVisionModel built at gill/models.py:78-96 and called by
 * GILLModel.get_visual_embs, gill/models.py:129-145: `self.visual_model(pixel_values).pooler_output`).
 * State-dict names are CLIPVisionModel's ("vision_model.embeddings.patch_embedding.weight",
 * "vision_model.encoder.layers.0.self_attn.q_proj.weight", "vision_model.pre_layrnorm.weight" (sic), ...).
 * ------------------------------------------------------------------------------------------ */
typedef struct gill_clip_config {
  int32_t image_size;         /* 224 */
  int32_t patch_size;         /* 14 (ViT-L/14) */
  int32_t hidden_size;        /* 1024 */
  int32_t num_layers;         /* 24 */
  int32_t num_heads;          /* 16 */
  int32_t intermediate_size;  /* 4096 */
  int32_t max_batch;
} gill_clip_config;
typedef struct gill_clip gill_clip;

int gill_clip_create(gill_clip** out, const gill_clip_config* cfg, const gill_tensor* weights, int n_weights);
void gill_clip_destroy(gill_clip* h);

/* pixel_values (B,3,S,S) fp32 (already resized / normalised)  ->  pooler_output (B, hidden) fp32. */
int gill_clip_forward(gill_clip* h, const float* pixel_values, int B, float* pooled_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 2 — GILLMapper = gill.layers.TextFcLayer(mode='gill_mapper').forward (layers.py:28-53).
 * State-dict names are TextFcLayer's ("fc.weight", "tfm.encoder.layers.0.self_attn.in_proj_weight",
 * "query_embs", "model.weight", ...).
 * ------------------------------------------------------------------------------------------ */
typedef struct gill_mapper gill_mapper;
typedef struct {
  int32_t in_dim;         /* 4096 (opt-6.7b) / 768 (opt-125m) */
  int32_t out_dim;        /* 768 */
  int32_t hidden_dim;     /* 512 */
  int32_t num_heads;      /* 4 */
  int32_t ffn_dim;        /* 2048 */
  int32_t num_enc_layers; /* 4 */
  int32_t num_dec_layers; /* 4 */
  int32_t num_input_tokens;  /* 8 */
  int32_t num_output_tokens; /* 77 */
  int32_t max_batch;
} gill_mapper_config;

int gill_mapper_create(gill_mapper** out, const gill_mapper_config* cfg, const gill_tensor* weights, int n_weights);
void gill_mapper_destroy(gill_mapper* h);
/* x (B,8,in_dim) bf16, input_embs (Be,8,in_dim) bf16 with Be in {1,B} (broadcast like torch)
 *   -> out (B,77,out_dim) fp32 */
int gill_mapper_forward(gill_mapper* h, const void* x_bf16, const void* input_embs_bf16, int B, int Be, float* out,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 3 — Stable Diffusion UNet + PNDM/PLMS classifier-free-guidance loop.  Replaces
 * self.sd_pipe(prompt_embeds=..., guidance_scale=..., num_inference_steps=...) at
 * gill/models.py:730-731, whose driver is restated in-tree at gill/custom_sd.py:567-651.
 * State-dict names are diffusers UNet2DConditionModel's ("down_blocks.0.resnets.0.conv1.weight").
 * ------------------------------------------------------------------------------------------ */
typedef struct gill_unet gill_unet;
typedef struct {
  int32_t in_channels;          /* 4 */
  int32_t out_channels;         /* 4 */
  int32_t block_out_channels[4];/* 320,640,1280,1280 */
  int32_t layers_per_block;     /* 2 */
  int32_t cross_attention_dim;  /* 768 */
  int32_t num_heads;            /* 8 (SD-1.5's "attention_head_dim" is the head COUNT) */
  int32_t norm_num_groups;      /* 32 */
  int32_t sample_size;          /* 64 latent pixels per side */
  int32_t ctx_len;              /* 77 */
  int32_t max_batch;            /* largest UNet batch (2 x prompts with CFG) */
  int32_t heads_per_level[4];   /* all 0: num_heads at every level (SD-1.x); SD-2.x: 5,10,20,20 (head dim 64 everywhere) */
  int32_t v_prediction;         /* 0: the UNet predicts epsilon (SD-1.x, SD-2.1-base); 1: v (SD-2.1-768) */
  int32_t fp8_convs;            /* BASELINE configs[4] (no reference counterpart: the reference runs SD in fp16).  1: the ResnetBlock2D 3x3 convolutions
                                 * AND the GEGLU projections of the level 1-3 transformer blocks with fp8 (e4m3) activations and weights on the fp8
                                 * matrix instruction (csrc/conv_fp8.hip, csrc/linear_fp8.hip); 2: the convolutions only (the round-5 mode); 0: off
                                 * (bf16, the parity configuration) */
} gill_unet_config;

int gill_unet_create(gill_unet** out, const gill_unet_config* cfg, const gill_tensor* weights, int n_weights);
void gill_unet_destroy(gill_unet* h);

/* One UNet forward (custom_sd.py:633-638): sample (Bx,4,L,L) fp32 NCHW, timesteps (Bx) fp32 HOST,
 * ctx (Bx,77,768) bf16 -> eps_out (Bx,4,L,L) fp32 NCHW. */
int gill_unet_forward(gill_unet* h, const float* sample, const float* timesteps_host, const void* ctx_bf16, int Bx,
                      float* eps_out, void* stream);

/* The whole denoise loop (custom_sd.py:607-651 with PNDMScheduler(skip_prk_steps, steps_offset=1,
 * scaled_linear 0.00085..0.012, 1000 train steps)):
 *   cond (B,77,768) bf16, uncond (n_uncond,77,768) bf16 with n_uncond == 1 (one negative embedding repeated over the batch,
 *   custom_sd.py:365-369) or == B (per-sample negative_prompt_embeds), latents0 (B,4,L,L) fp32 (already * init_noise_sigma=1)
 *   -> latents_out (B,4,L,L) fp32.  num_steps "inference steps" = num_steps+1 UNet calls of batch 2B.
 *   guidance <= 1 disables CFG (batch B, uncond ignored) like do_classifier_free_guidance.
 *   The loop is enqueued on a stream the handle owns, ordered after / before the caller's `stream` by events. */
int gill_sd_denoise(gill_unet* h, const void* cond_bf16, const void* uncond_bf16, int n_uncond, const float* latents0, int B,
                    int num_steps, float guidance, float* latents_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 3b — VAE decode of the final latents.  Replaces StableDiffusionPipeline.decode_latents
 * (gill/custom_sd.py:385-392: latents / 0.18215 -> vae.decode -> (x/2+0.5).clamp(0,1)) and the uint8 conversion
 * of numpy_to_pil (custom_sd.py:660-661).  State-dict names are diffusers AutoencoderKL's
 * ("post_quant_conv.weight", "decoder.up_blocks.2.resnets.0.conv1.weight", ...).
 * ------------------------------------------------------------------------------------------ */
typedef struct gill_vae gill_vae;
typedef struct {
  int32_t latent_channels;       /* 4 */
  int32_t out_channels;          /* 3 */
  int32_t block_out_channels[4]; /* 128,256,512,512 */
  int32_t layers_per_block;      /* 2 */
  int32_t norm_num_groups;       /* 32 */
  int32_t latent_size;           /* 64 -> 512x512 pixels */
  float scaling_factor;          /* 0.18215 (hard-coded at custom_sd.py:387) */
  int32_t max_batch;
} gill_vae_config;

int gill_vae_create(gill_vae** out, const gill_vae_config* cfg, const gill_tensor* weights, int n_weights);
void gill_vae_destroy(gill_vae* h);
/* latents (B,4,L,L) fp32 -> image_f32 (B,3,8L,8L) fp32 in [-1,1] (vae.decode(...).sample; may be NULL) and/or
 * image_u8 (B,8L,8L,3) uint8 = round(255 * clamp(x/2+0.5, 0, 1)) (may be NULL). */
int gill_vae_decode(gill_vae* h, const float* latents, int B, float* image_f32, uint8_t* image_u8, void* stream);

/* PNDM schedule known-answers for tests (host arrays): timesteps_out must hold num_steps+1 ints;
 * returns the number written.  alphas_cumprod_out (optional) must hold 1000 doubles. */
int gill_pndm_schedule(int num_steps, int32_t* timesteps_out, double* alphas_cumprod_out);
/* Exclusive-device contract of the UNet engine (round 6).  Some of its 3x3 convolutions finish the GroupNorm that consumes them inside their own
 * launch: their workgroups wait for each other on arrival counters, which is deadlock-free only while the handle's stream has the device's CUs to
 * itself (one process per GPU, as DESIGN.md section 5 deploys it; kernels of other streams that do not themselves wait only delay it).  Two such
 * launches running side by side (two processes sharing one GPU, two handles on two streams) can starve each other: the waits are bounded, a
 * workgroup that gives up NaN-poisons its outputs and counts itself.  gill_unet_forward / gill_sd_denoise fail with -5 when the count is non-zero
 * on entry (and clear it); this function reads and clears it on demand (synchronous small copy; < 0 on a HIP error).  GILL_GEMM_COOP=0 in the
 * environment turns the in-kernel finishes off (every GroupNorm a launch of its own, the round-5 dataflow: -0.6 % on the loop). */
int gill_coop_timeouts(void);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (the kernels the three stages are built from), exported so the
 * parity tests can pin each against the CPU oracle.  All bf16 unless noted.
 * ------------------------------------------------------------------------------------------ */
/* C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias[N] + resid[M,N]); act: 0 none 1 relu 2 gelu(erf) 3 silu.
 * out_f32: C is fp32 instead of bf16.  splitk: 0 = auto, n > 1 = forced n-way split; < 0 = the general row-major tiles even for the
 * weight-streaming shapes (N * K >= 4 Mi) that otherwise run on a 64 x 64-blocked copy of W — the STREAM64 tile up to 256 rows, the general tiles reading
 * the blocked layout above — (-1: auto split, -n: n ways).
 * Split-K launches share one grow-only workspace owned by the library: one caller at a time (as everywhere on this path). */
int gill_op_gemm(const void* A, const void* W, const float* bias, const void* resid_bf16, void* C, int M, int N, int K,
                 float alpha, int act, int out_f32, int splitk, void* stream);
/* GEGLU projection: W (2*inner, K) in diffusers order [value rows | gate rows], bias (2*inner) ->
 * C (M, inner) = (A.Wv^T + bv) * gelu(A.Wg^T + bg) */
int gill_op_geglu(const void* A, const void* W, const float* bias, void* C, int M, int inner, int K, void* stream);
/* 3x3 pad-1 convolution over NHWC: x1 (B,IH,IW,C1) [++ x2 (B,IH,IW,C2) channel-concat], w OIHW fp32 (Cout,C1+C2,3,3),
 * stride 1|2, ups=1 -> nearest 2x upsample first.  rowvec (B,Cout) fp32 optional per-sample bias; resid NHWC optional. */
int gill_op_conv3x3(const void* x1, int C1, const void* x2, int C2, const float* w_oihw, const float* bias,
                    const float* rowvec, const void* resid, void* y, int B, int IH, int IW, int Cout, int stride, int ups,
                    int splitk, void* stream);
/* 3x3 convolution (stride 1, pad 1) + the GroupNorm (+ SiLU) that consumes it, without a GroupNorm launch where the geometry allows (diffusers
 * ResnetBlock2D: conv1 -> norm2 -> silu, conv2 -> the next block's norm; reference call site gill/custom_sd.py:633-638):
 * y_raw (optional, may be NULL) = conv(x) + bias + rowvec[b] + resid, y_norm = [silu](GroupNorm(y_raw)).
 * x (B,H,W,Cin) bf16 NHWC, w (Cout,Cin,3,3) fp32, rowvec (B,Cout) fp32 optional, gamma / beta (Cout) fp32.
 * splitk >= 2: H * W in {64, 256}, Cout % 80 == 0, Cout / groups a multiple of 4 that divides 80; the normalisation runs in the split-K reducer
 *   launch (coop = 0) or inside the convolution's own launch by its co-resident workgroups (coop = 1, where the grid fits the device's CUs and
 *   Cout % 160 == 0; otherwise as coop = 0).
 * splitk == 1: coop = 1: in the convolution's epilogue (rows per sample a multiple of the 128- / 256-row tile, B * H * W % 256 == 0, Cout % 160 == 0,
 *   grid <= CUs; an error elsewhere); coop = 0: convolution with fused statistics + a GroupNorm-apply launch (the reference dataflow).
 * ss_out (optional; splitk == 1 && coop only): the scale | shift table [B][2][Cout] fp32 (y = x * scale + shift); y_norm may then be NULL.
 * Synchronises. */
int gill_op_conv3x3_gn(const void* x, const float* w_oihw, const float* bias, const float* rowvec, const void* resid, const float* gamma,
                       const float* beta, int groups, float eps, int silu, void* y_raw, void* y_norm, float* ss_out, int B, int H, int W,
                       int Cin, int Cout, int splitk, int coop, void* stream);
/* conv3x3 (stride 1, pad 1) of x1 ++ x2 plus a fused 1x1 convolution of xs1 ++ xs2 (ResnetBlock2D.conv2 + conv_shortcut as one
 * implicit GEMM): y (B,IH,IW,Cout) bf16 NHWC; w_oihw (Cout, C1+C2, 3, 3) fp32, w_sc (Cout, CS1+CS2) fp32.  Synchronises. */
int gill_op_conv3x3_shortcut(const void* x1, int C1, const void* x2, int C2, const float* w_oihw, const float* bias,
                             const void* xs1, int CS1, const void* xs2, int CS2, const float* w_sc, void* y, int B, int IH, int IW,
                             int Cout, int splitk, void* stream);
/* softmax(scale * q k^T [+causal]) v over token-major q (B,nq,H*d), k/v (B,nkv,H*d) -> o (B,nq,H*d) */
int gill_op_attention(const void* q, const void* k, const void* v, void* o, int B, int H, int nq, int nkv, int d,
                      float scale, int causal, void* stream);
int gill_op_layernorm(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, int rows, int C,
                      float eps, void* stream);
int gill_op_groupnorm(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, const float* gamma,
                      const float* beta, float eps, int silu, void* y, void* stream);

/* The feed-forward sub-block of a level-0 (C = 320) transformer block + proj_out + outer residual as one kernel (csrc/ffn.hip):
 * out = proj_out(ff2(geglu(ff1(LN(t)))) + t) + resid on natural (diffusers-layout) operands; gn_stats (optional): GroupNorm partial sums
 * of the output, [(b * rows_per_batch / 64 + slab) * 64 + bin][2], bins of 5 channels.  Replaces, inside gill_unet_forward, the
 * BasicTransformerBlock.ff call + Transformer2DModel.proj_out of diffusers (reference call site: gill/custom_sd.py:633-638).
 * o2 / Wo / bo2 (optional, all or none): t := t + attn2.to_out(o2) first, inside the kernel (o2 (M, 320) = the cross-attention output,
 * Wo (320, 320), bo2 (320)) — the form the UNet engine runs. */
int gill_op_ffn_fused(const void* t_bf16, const float* ln_g, const float* ln_b, const void* W1_bf16, const float* b1,
                      const void* W2_bf16, const float* b2, const void* Wp_bf16, const float* bp, const void* resid_bf16,
                      void* out_bf16, float* gn_stats, int M, int rows_per_batch, const void* o2_bf16, const void* Wo_bf16,
                      const float* bo2, void* stream);

/* BASELINE configs[4] (not a reference function): the GEGLU projection of a BasicTransformerBlock (diffusers ff.net.0; reference call site
 * gill/custom_sd.py:633-638) on CDNA4's fp8 matrix instruction — out (M, inner) bf16 = h * gelu(g), [h | g] = LayerNorm(t; ln_g, ln_b) W^T + b with
 * t (M, C) bf16, W (2 inner, C) bf16 in diffusers order [value rows | gate rows], b (2 inner) fp32; activations and weights quantised to e4m3 exactly as
 * the engine's fp8 mode does (per-row weight scales, fp8(16 * LNhat(t))).  inner % 16 == 0, C % 128 == 0.  Synchronises. */
int gill_op_geglu_fp8(const void* t, const float* ln_g, const float* ln_b, const void* W, const float* b, void* out, int M, int inner, int C,
                      void* stream);
/* The two projections around norm1 / norm2 of a level-0 (C = 320, 8 heads of 40) transformer block as one kernel (csrc/lnproj.hip), on
 * natural (diffusers-layout) operands.  mode 0: t = proj_in(x); [q | k | v] = [to_q; to_k; to_v](LN(t)) (W2 = the three weights stacked,
 * (960, 320)).  mode 1: t += to_out(x) + b1 (in place, x = the attention output); q = to_q(LN(t)) (W2 (320, 320)).  Outputs in the
 * attention kernels' layouts: q, k [B][8][hw_pad][48] bf16 (q pre-scaled by log2(e) / sqrt(40)), vt [B][8][64][hw_pad] (row 48 = 1);
 * hw_pad = HW rounded up to 32; B * HW a multiple of 128.  Replaces, inside gill_unet_forward, Transformer2DModel.proj_in + norm1 +
 * attn1.to_q/k/v, resp. attn1.to_out + residual + norm2 + attn2.to_q of diffusers (reference call site: gill/custom_sd.py:633-638).
 * Synchronises. */
int gill_op_lnproj(int mode, const void* x_bf16, void* t_bf16, const void* W1_bf16, const float* b1, const float* ln_g, const float* ln_b,
                   const void* W2_bf16, void* q_bf16, void* k_bf16, void* vt_bf16, int B, int HW, void* stream);

/* attn2 (cross-attention) of a UNet transformer block with heads of 80..160 features, as the two GEMMs on per-sample weights the engine
 * runs at UNet levels 1-3 (csrc/unet.hip "XALG"): K and V are linear in the prompt context, so qs (g o Wq_h)^T Wk_h and Wo_h Wv_h are
 * folded at load, multiplied by ctx once per prompt, and each UNet call computes P = softmax(LN(t) Mq_b^T) and out = t + P Wo_b^T + bo.
 * Operands in diffusers layout: t (B * HW, C) bf16, norm2 gain / bias (C) fp32, to_q / to_out.0 weights (C, C) bf16, to_k / to_v (C, E)
 * bf16, to_out.0 bias (C) fp32, ctx (B, ctx_len <= 80, E) bf16 -> out (B * HW, C) bf16; P_bf16 (optional, B * HW x 80 H): the softmax
 * weights, key j of head h at column 80 h + j.  Replaces norm2 + CrossAttention(attn2) + residual of diffusers' BasicTransformerBlock inside
 * gill_unet_forward (reference call site: gill/custom_sd.py:633-638).  Synchronises. */
int gill_op_cross_attention_folded(const void* t_bf16, const float* ln_g, const float* ln_b, const void* Wq_bf16, const void* Wk_bf16,
                                   const void* Wv_bf16, const void* Wo_bf16, const float* bo, const void* ctx_bf16, void* out_bf16,
                                   void* P_bf16, int B, int HW, int C, int H, int ctx_len, int E, void* stream);

/* fp8 (OCP e4m3) 3x3 convolution on CDNA4's v_mfma_scale_f32_16x16x128_f8f6f4 — BASELINE.json configs[4]; no reference
 * counterpart (the reference runs SD in fp16, gill/models.py:550-551).  x (B,H,W,Cin) bf16 NHWC, w (Cout,Cin,3,3) fp32,
 * optional bias (Cout) fp32 and residual (B,H,W,Cout) bf16 -> y (B,H,W,Cout) bf16.  Operands are quantised inside
 * (activations x 8 per tensor, weights per output channel).  splitk 0 = heuristic. */
int gill_op_conv3x3_fp8(const void* x_bf16, const float* w_oihw, const float* bias, const void* resid_bf16, void* y_bf16,
                        int B, int H, int W, int Cin, int Cout, int splitk, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GILL_AMD_H */
