// Stage 3 — Stable Diffusion UNet + PNDM/PLMS classifier-free-guidance loop.
//
// Replaces self.sd_pipe(prompt_embeds=..., guidance_scale=..., num_inference_steps=...)
// (gill/models.py:730-731); the loop is the one restated in-tree at gill/custom_sd.py:607-651:
//   latent_model_input = cat([latents]*2)            custom_sd.py:630
//   noise_pred = unet(latent_model_input, t, encoder_hidden_states=prompt_embeds).sample      :633-638
//   noise_pred = uncond + g * (text - uncond)        :641-643
//   latents = scheduler.step(noise_pred, t, latents) :646          (PNDMScheduler, skip_prk_steps)
// The UNet itself (diffusers UNet2DConditionModel, SD-1.5 config) is not in the reference tree; its
// structure here follows the public architecture (state-dict key names are diffusers').
//
// MI355X design: activations are NHWC bf16 so that
//   * every 3x3 conv is an implicit GEMM whose K steps are contiguous 128-B channel runs (gemm.hip),
//   * (B, HW, C) attention tokens ARE the NHWC tensor: no transposes anywhere,
//   * skip-connection concats are never materialised (two-source GroupNorm / conv / shortcut GEMM),
//   * nearest-2x upsampling is folded into the following conv's gather.
// Step-invariant work is hoisted out of the 51-call loop: the cross-attention K/V of all 16 layers
// (they depend only on the prompt embedding) and the whole time-embedding MLP + the 22 resnet
// time projections for every timestep (one batched GEMM chain -> a [steps][sum Cout] table that the
// conv epilogues add as a per-channel row vector).
#include "engine_util.h"
#include <math.h>
#include <stdlib.h>
#include <map>
#include <set>
#include <hip/hip_fp16.h>

namespace {

struct ConvW {
  bf16_t* w = nullptr; float* b = nullptr; int cin = 0, cout = 0; int chunked = 0; /* K order: GemmArgs::k_chunked */ int ups4 = 0; /* w = the 4-tap parity-class form (GemmArgs::ups == 2) */
  // gill_unet_config.fp8_convs: e4m3 weights [cout][kpad] in conv_fp8.hip's K order + per-output-channel de-quantisation scale
  unsigned char* w8 = nullptr; float* cs = nullptr; int kpad = 0;
};
struct LinW { bf16_t* w = nullptr; float* b = nullptr; int out = 0, in = 0; };
struct NormW { float* g = nullptr; float* b = nullptr; int c = 0; };

struct ResnetW {
  NormW n1, n2;
  ConvW c1, c2;
  bool has_sc = false;
  LinW sc;
  // conv2 with the 1x1 conv_shortcut fused as extra K channels: weights [cout][9*cout + cin], bias b2 + b_sc
  bf16_t* c2f_w = nullptr; float* c2f_b = nullptr;
  int cin = 0, cout = 0, temb_off = 0;
};

struct XfW {  // Transformer2DModel with one BasicTransformerBlock
  int C = 0, heads = 0, d = 0, dp = 0, dpv = 0, layer_id = 0;
  NormW gn;
  LinW proj_in, proj_out;
  NormW ln1, ln2, ln3;
  bf16_t* wqkv1 = nullptr;    // [3*H*dp][C]
  LinW out1;                  // [C][H*dp]
  bf16_t* wq2 = nullptr;      // [H*dp][C]
  bf16_t* wkv2 = nullptr;     // [2*H*dp][ctx_dim]
  LinW out2;
  bf16_t* wff1 = nullptr; float* bff1 = nullptr;   // GEGLU-permuted [8C][C]
  unsigned char* wff1_8 = nullptr; float* cs_ff1 = nullptr;   // gill_unet_config.fp8_convs: the same (LayerNorm-folded) rows in e4m3 + per-row de-quantisation scale (linear_fp8.hip)
  bf16_t* w1c = nullptr; float* b1c = nullptr; bf16_t* w2p = nullptr;   // the same weights as the fused feed-forward kernel reads them (ffn.hip; C = 320 only)
  bf16_t* wpp = nullptr;        // proj_out's weight in the permuted k order: set when the fused feed-forward kernel also runs attn2.to_out (ffn.hip PRE)
  bool w1c_kperm = false;       // the layout w1c was written in: the PRE kernel's permuted k order (ffn_relayout_launch with Wpp) — the kernel form run in xf() must match it
  bf16_t *wqkv1p = nullptr, *wq2p = nullptr;   // wqkv1 / wq2 (LayerNorm-folded) with the K order of the fused projection pairs (lnproj.hip; C = 320 only)
  // norm1/2/3 are folded into wqkv1 / wq2 / wff1 at load (GemmArgs::ln_stats): column sums of g*W and beta.W^T (+ bias)
  float *s_qkv1 = nullptr, *c_qkv1 = nullptr, *s_q2 = nullptr, *c_q2 = nullptr, *s_ff1 = nullptr;
  bf16_t* xg = nullptr; float* xgb = nullptr;   // XALG (see xalg_fold_kernel): [2 H C][ctx_dim] = G | G2, and gb [H][ctx_dim]; null = attention-kernel form
  LinW ff2;                   // [C][4C]
  bf16_t* wfo = nullptr; float* bfo = nullptr;   // ff2 and proj_out as one map: [C][4C + C] = [Wp W2 | Wp], bias Wp b2 + bp
};

// stats: optional slot [Bx][H*W/64][C/sbin][2] that the PRODUCING GEMM epilogue fills with this tensor's per-slab GroupNorm
// partial sums (GemmArgs::gn_stats: written once each, added in slab order by the consumer)
// sbin: channels per statistics bin (C/64: finer than a group, so the sums also serve the wider groups of a skip concat)
// nslab: partials per (sample, bin) the producer actually wrote (set when the producing GEMM is launched)
struct Tensor { bf16_t* p = nullptr; int H = 0, W = 0, C = 0; float* stats = nullptr; int sbin = 0; int nslab = 1; };
// row sums feeding a folded LayerNorm: [planes][rows][2], planes fixed by the producing GEMM's tiling (gemm_row_planes)
struct RowStats { float* p = nullptr; int planes = 1; };

struct Arena {
  unsigned char* base = nullptr;
  size_t cap = 0, off = 0, high = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > high) high = off;
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

}  // namespace

struct GraphKey {     // (the guidance scale is a device-side scalar, not part of the captured step)
  int Bx, cfg;
  bool operator<(const GraphKey& o) const {
    if (Bx != o.Bx) return Bx < o.Bx;
    return cfg < o.cfg;
  }
};

struct gill_unet {
  gill_unet_config cfg;
  DevPool pool;
  // weights
  bf16_t* conv_in_w = nullptr; float* conv_in_b = nullptr;
  bf16_t* conv_out_w = nullptr; float* conv_out_b = nullptr;
  NormW norm_out;
  LinW te1, te2;
  bf16_t* temb_proj_w = nullptr; float* temb_proj_b = nullptr; int temb_total = 0;   // [sum Cout][1280]
  std::vector<ResnetW> down_res[4], up_res[4];
  std::vector<XfW> down_xf[4], up_xf[4];
  ConvW down_ds[3], up_us[3];
  ResnetW mid_res[2];
  XfW mid_xf;
  int n_xf = 0;
  int temb_dim = 0;
  // workspace
  Arena arena;
  unsigned char* arena_mem = nullptr;
  float* gn_stats = nullptr;      // per-forward pool of GroupNorm partial-sum slots (bump-allocated; the dry run sizes it)
  size_t gn_floats = 0, gn_next = 0;
  float* ln_stats = nullptr;      // per-forward pool of the row-sum planes feeding the folded LayerNorms
  size_t ln_floats = 0, ln_next = 0;
  float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
  // COOP arrival counters (GemmArgs::coop_ctr): one slot range per GEMM launch of a forward, bump-allocated in launch order (the dry run sizes the
  // pool), all zeroed by the forward's first kernel (im2col_nchw_launch)
  unsigned* coop_ctr = nullptr; size_t coop_n = 0, coop_next = 0;
  std::vector<bf16_t*> kcache, vcache;   // per transformer layer: [Bx][H][ctx_pad][dp] / [Bx][H][dpv][ctx_pad]
  int ctx_pad = 0;
  // XALG layers (XfW::xg): per-sample operands of the two cross-attention GEMMs — scores [Bx][80 H][C] + its folded-LayerNorm
  // column sums and constants [Bx][80 H], values [Bx][C][80 H]
  std::vector<bf16_t*> xq_w, xo_w; std::vector<float*> xq_cs, xq_b;
  // time embedding scratch
  float* t_dev = nullptr;       // [rows]
  bf16_t* t_sin = nullptr;      // [rows][320]
  bf16_t* t_h1 = nullptr;       // [rows][1280]
  bf16_t* t_h2 = nullptr;       // [rows][1280]
  float* temb_table = nullptr;  // [rows][temb_total]
  int temb_rows_cap = 0;
  // loop state
  float* lat = nullptr;         // [B][4*L*L]
  float* lat2 = nullptr;        // [2B][4*L*L]
  float* eps = nullptr;         // [2B][4*L*L]
  float* cur_sample = nullptr;
  float* ets = nullptr;         // [4][B][4*L*L]
  bf16_t* ctx_full = nullptr;   // [2B][77][768]
  float* temb_cur = nullptr;    // [temb_total]: time-embedding row of the step being replayed
  PlmsRow* plms_rows = nullptr; // [temb_rows_cap]: per-step PLMS coefficients of the running loop
  int* step_ctr = nullptr;      // [2]: next / current step of the running loop (SdLoopArgs::ctr)
  float* guidance_dev = nullptr; // [1]: guidance scale of the running loop (SdLoopArgs::guidance)
  // hipGraph of one UNet forward per UNet batch size (captured after the first eager forward of that size)
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::set<GraphKey> warmed;
  bool use_graph = true;
  // The denoise loop runs on the handle's private stream, fenced to the caller's stream by events (capture needs a
  // non-NULL stream anyway; see DESIGN.md "hipGraph replay and the NULL stream").
  StreamFence fence;
  ~gill_unet() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
  }
};

__global__ void vec_add_f32_kernel(const float* a, const float* b, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// dst[r][h*dp + dd] = src[r][h*d + dd] (dd < d), zero elsewhere.  dst pre-zeroed.
__global__ __launch_bounds__(256) void pad_head_cols_kernel(const void* src, int dtype, int rows, int H, int d, int dp,
                                                            bf16_t* dst) {
  const int64_t total = (int64_t)rows * H * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(i % d);
    const int h = (int)((i / d) % H);
    const int64_t r = i / ((int64_t)d * H);
    float v;
    if (dtype == 0) v = bf2f(((const bf16_t*)src)[i]);
    else if (dtype == 1) v = ((const float*)src)[i];
    else v = (float)(((const __half*)src)[i]);
    dst[r * (int64_t)H * dp + h * dp + dd] = f2bf(v);
  }
}
// dst[(h*dp + dd)][:] = src[(h*d + dd)][:]  (row padding of q/k/v projection weights).  dst pre-zeroed.
__global__ __launch_bounds__(256) void pad_head_rows_kernel(const void* src, int dtype, int H, int d, int dp, int cols,
                                                            bf16_t* dst) {
  const int64_t total = (int64_t)H * d * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const int64_t r = i / cols;
    const int h = (int)(r / d), dd = (int)(r % d);
    float v;
    if (dtype == 0) v = bf2f(((const bf16_t*)src)[i]);
    else if (dtype == 1) v = ((const float*)src)[i];
    else v = (float)(((const __half*)src)[i]);
    dst[((int64_t)h * dp + dd) * cols + c] = f2bf(v);
  }
}

// Feed-forward output and proj_out are two linear maps with only a residual add in between:
//   out = proj_out(ff2(h) + t) + x_in = h (Wp W2)^T + t Wp^T + (Wp b2 + bp) + x_in
// so they run as ONE two-source GEMM over K = [h (4C) | t (C)].  This builds its weight rows [Wp W2 | Wp] ([C][5C], products in
// fp32 from the checkpoint's own dtype, one rounding to bf16) and its bias Wp b2 + bp.  Load time only: plain loops.
__device__ __forceinline__ float ld_any(const void* p, int dtype, int64_t i) {
  if (dtype == 0) return bf2f(((const bf16_t*)p)[i]);
  if (dtype == 1) return ((const float*)p)[i];
  return (float)(((const __half*)p)[i]);
}
__global__ __launch_bounds__(256) void ffo_fuse_kernel(const void* wp, int dt_p, const void* w2, int dt_2, const void* b2, int dt_b2,
                                                       const void* bp, int dt_bp, int C, bf16_t* w_out, float* b_out) {
  const int n = blockIdx.y;                       // output row
  const int k = blockIdx.x * 256 + threadIdx.x;   // column of [4C | C | 1 (bias)]
  const int K4 = 4 * C;
  if (k < K4) {
    float a = 0.f;
    for (int j = 0; j < C; ++j) a = fmaf(ld_any(wp, dt_p, (int64_t)n * C + j), ld_any(w2, dt_2, (int64_t)j * K4 + k), a);
    w_out[(size_t)n * 5 * C + k] = f2bf(a);
  } else if (k < 5 * C) {
    w_out[(size_t)n * 5 * C + k] = f2bf(ld_any(wp, dt_p, (int64_t)n * C + (k - K4)));
  } else if (k == 5 * C) {
    float a = ld_any(bp, dt_bp, n);
    for (int j = 0; j < C; ++j) a = fmaf(ld_any(wp, dt_p, (int64_t)n * C + j), ld_any(b2, dt_b2, j), a);
    b_out[n] = a;
  }
}

// CROSS-ATTENTION AS TWO GEMMs ("XALG": UNet levels 1-3, head dim >= 80).  The keys and values of attn2 are linear maps of the 77 prompt
// tokens, fixed for the whole denoising loop, so per sample b and head h
//   scores[m][j] = qs LN(t)[m] . Wq_h^T K_bh[j]            = LN(t)[m] . (ctx_b[j] G_h)^T,     G_h  = qs (g o Wq_h)^T Wk_h      [C][768]
//   out[m]       = sum_h softmax(scores)[m][h][:] V_bh Wo_h^T = sum_h P[m][h][:] (ctx_b G2_h)^T, G2_h = Wo_h Wv_h             [C][768]
// G / G2 depend on the weights only (built here at load, fp32 products of the bf16 weights, one rounding); once per prompt the
// context turns them into per-sample weight matrices (unet_ctx_cache) and every UNet call then runs attn2 as
//   P = softmax80(LN(t) Mq_b^T)  (GEMM, N = 80 H: GemmArgs::OUT_SOFTMAX80)   and   t += P Wo_b^T + bias  (GEMM, K = 80 H)
// instead of to_q + the attention kernel + to_out: at d = 160 (levels 2-3) both GEMMs are half the size of the projections they
// replace, and the attention launch is gone.  GILL_UNET_XALG = 0 keeps the three-kernel form.
// rows [0, H C): G[h][c][:]; rows [H C, 2 H C): G2[h][co][:].  8 rows per workgroup (one head), threads over the 768 context features.
__global__ __launch_bounds__(256) void xalg_fold_kernel(const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wkv, const bf16_t* __restrict__ wo,
                                                        int H, int C, int dp, int E, float qs, bf16_t* __restrict__ G) {
  __shared__ float a[8][160];
  const int r0 = blockIdx.x * 8;                  // first of 8 rows (C % 8 == 0: one head, one half)
  const int half = r0 >= H * C;
  const int rr = r0 - half * H * C;
  const int h = rr / C, c0 = rr - h * C;
  const int hdp = H * dp;
  for (int i = threadIdx.x; i < 8 * dp; i += 256) {
    const int r = i / dp, n = i - r * dp;
    a[r][n] = half ? bf2f(wo[(size_t)(c0 + r) * hdp + h * dp + n]) : qs * bf2f(wq[(size_t)(h * dp + n) * C + c0 + r]);
  }
  __syncthreads();
  const bf16_t* wb = wkv + (size_t)(half * hdp + h * dp) * E;      // Wk_h | Wv_h: [dp][E]
  for (int e = threadIdx.x; e < E; e += 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int n = 0; n < dp; ++n) {
      const float b = bf2f(wb[(size_t)n * E + e]);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = fmaf(a[r][n], b, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) G[(size_t)(r0 + r) * E + e] = f2bf(acc[r]);
  }
}
// gb[h][e] = qs sum_n c_q[h dp + n] Wk[h dp + n][e]   (c_q = beta . Wq^T: the constant part of the folded norm2 -> to_q)
__global__ __launch_bounds__(256) void xalg_fold_bias_kernel(const float* __restrict__ cq, const bf16_t* __restrict__ wk, int dp, int E, float qs,
                                                             float* __restrict__ gb) {
  const int h = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float acc = 0.f;
  for (int n = 0; n < dp; ++n) acc = fmaf(cq[h * dp + n], bf2f(wk[(size_t)(h * dp + n) * E + e]), acc);
  gb[(size_t)h * E + e] = qs * acc;
}
// Once per prompt: T [Bx ctx_len][2 H C] = ctx [G | G2]^T (one GEMM) is dealt into the per-sample operands of the two GEMMs.
// Scores operand: Mq[b][80 h + j][:] = T[b ctx_len + j][h C ..], its row sums (folded LayerNorm) and the constant term ctx_b[j] . gb[h];
// key slots j >= ctx_len: zero rows with constant -1e30 (softmax weight 0).  One workgroup per (b, h, j).
__global__ __launch_bounds__(256) void xalg_scores_operand_kernel(const bf16_t* __restrict__ T, const bf16_t* __restrict__ ctx, const float* __restrict__ gb,
                                                                  int H, int C, int E, int ctx_len, bf16_t* __restrict__ Mq, float* __restrict__ cs,
                                                                  float* __restrict__ cb) {
  __shared__ float red[2][4];
  const int j = blockIdx.x % 80, h = (blockIdx.x / 80) % H, b = blockIdx.x / (80 * H);
  bf16_t* dst = Mq + (size_t)blockIdx.x * C;
  float sum = 0.f, dot = 0.f;
  if (j < ctx_len) {
    const bf16_t* src = T + (size_t)(b * ctx_len + j) * (2 * H * C) + (size_t)h * C;
    for (int c = threadIdx.x * 8; c < C; c += 2048) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + c);
      *reinterpret_cast<uint4*>(dst + c) = v;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) sum += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xffff0000u);
    }
    const bf16_t* cr = ctx + (size_t)(b * ctx_len + j) * E;
    for (int e = threadIdx.x; e < E; e += 256) dot = fmaf(bf2f(cr[e]), gb[(size_t)h * E + e], dot);
  } else {
    for (int c = threadIdx.x * 8; c < C; c += 2048) *reinterpret_cast<uint4*>(dst + c) = make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); dot += __shfl_xor(dot, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum; red[1][threadIdx.x >> 6] = dot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cs[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    cb[blockIdx.x] = j < ctx_len ? red[1][0] + red[1][1] + red[1][2] + red[1][3] : -1e30f;
  }
}
// Values operand: Wo_b[co][80 h + j] = T[b ctx_len + j][H C + h C + co] (0 for j >= ctx_len): an 80 x 64 transpose per workgroup (b, h, co / 64).
__global__ __launch_bounds__(256) void xalg_values_operand_kernel(const bf16_t* __restrict__ T, int H, int C, int ctx_len, bf16_t* __restrict__ Wo) {
  __shared__ bf16_t tile[80][66];
  const int cb = blockIdx.x % (C / 64), h = (blockIdx.x / (C / 64)) % H, b = blockIdx.x / ((C / 64) * H);
  for (int i = threadIdx.x; i < 80 * 64; i += 256) {
    const int j = i >> 6, c = i & 63;
    tile[j][c] = j < ctx_len ? T[(size_t)(b * ctx_len + j) * (2 * H * C) + (size_t)H * C + (size_t)h * C + cb * 64 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 80; i += 256) {
    const int c = i / 80, j = i - c * 80;
    Wo[((size_t)b * C + cb * 64 + c) * (80 * H) + h * 80 + j] = tile[j][c];
  }
}

// The feed-forward sub-blocks at C = 320 (level 0) run as one kernel (ffn.hip) instead of GEGLU + the two-source ffo GEMM: loop
// 528.9 -> 522.8 ms.  GILL_UNET_FFN_FUSED = 0 restores the two GEMMs.
// GILL_UNET_LNPROJ=0: proj_in / QKV and attn1.to_out / attn2.to_q of the level-0 blocks as the separate GEMMs
static bool lnproj_on() {
  static const bool on = [] { const char* e = getenv("GILL_UNET_LNPROJ"); return !(e && e[0] == '0'); }();
  return on;
}
// GILL_UNET_XALG=0: attn2 of levels 1-3 as to_q + attention kernel + to_out instead of the two per-sample GEMMs (xalg_fold_kernel)
static bool xalg_on() {
  static const bool on = [] { const char* e = getenv("GILL_UNET_XALG"); return !(e && e[0] == '0'); }();
  return on;
}
static bool ffn_fused_on() {
  static const bool on = [] { const char* e = getenv("GILL_UNET_FFN_FUSED"); return !(e && e[0] == '0'); }();
  return on;
}

namespace {

struct Loader {
  const WeightTable& wt;
  DevPool& pool;
  hipStream_t s;
  int ctx_len = 0;
  int norm(const std::string& p, int c, NormW* n) {
    n->c = c;
    GILL_TRY(load_f32(wt, pool, p + ".weight", c, &n->g, s));
    return load_f32(wt, pool, p + ".bias", c, &n->b, s);
  }
  // hw: pixels per sample of the conv's INPUT (decides the K order, see GemmArgs::k_chunked)
  // ups4: the conv follows a nearest-2x upsample — store the four pre-summed 2x2-tap kernels instead (gemm.hip "UPS4";
  // GILL_CONV_UPS4 = 0 keeps the 9-tap gather over the upsampled grid)
  int conv3(const std::string& p, int cin, int cout, int hw, ConvW* c, bool f8 = false, bool ups4 = false) {
    c->cin = cin; c->cout = cout; c->chunked = conv_k_chunked(hw, cin, cout) ? 1 : 0;
    const gill_tensor* t;
    GILL_TRY(wt.get(p + ".weight", (int64_t)cout * cin * 9, &t));
    static const int ups4_on = [] { const char* v = getenv("GILL_CONV_UPS4"); return v ? atoi(v) : 1; }();
    if (ups4 && ups4_on && !f8) {
      c->ups4 = 1; c->chunked = 0;
      GILL_TRY(pool.alloc(&c->w, (size_t)16 * cout * cin, false));
      GILL_TRY(conv_weight_relayout_ups4_launch(t->data, t->dtype, cout, cin, c->w, s));
      return load_f32(wt, pool, p + ".bias", cout, &c->b, s);
    }
    if (f8) {
      c->kpad = conv_fp8_kpad(cin);
      GILL_TRY(pool.alloc(&c->w8, (size_t)cout * c->kpad, false));
      GILL_TRY(pool.alloc(&c->cs, (size_t)cout, false));
      GILL_TRY(conv_weight_quant_fp8_launch(t->data, t->dtype, cout, cin, F8_ACT_SCALE, c->w8, c->cs, s));
      return load_f32(wt, pool, p + ".bias", cout, &c->b, s);
    }
    GILL_TRY(pool.alloc(&c->w, (size_t)cout * cin * 9, false));
    if (c->chunked) GILL_TRY(conv_weight_relayout_chunked_launch(t->data, t->dtype, cout, cin, c->w, s));
    else GILL_TRY(conv_weight_relayout_launch(t->data, t->dtype, cout, cin, c->w, s));
    return load_f32(wt, pool, p + ".bias", cout, &c->b, s);
  }
  int lin(const std::string& p, int out, int in, LinW* l, bool bias = true) {
    l->out = out; l->in = in;
    GILL_TRY(load_bf16(wt, pool, p + ".weight", (int64_t)out * in, &l->w, s));
    if (bias) return load_f32(wt, pool, p + ".bias", out, &l->b, s);
    return 0;
  }
  // projection weight [H*d][cols] -> [H*dp][cols] into a caller-provided slot
  int head_rows(const std::string& name, int H, int d, int dp, int cols, bf16_t* dst) {
    const gill_tensor* t;
    GILL_TRY(wt.get(name, (int64_t)H * d * cols, &t));
    hipLaunchKernelGGL(pad_head_rows_kernel, dim3(1024), dim3(256), 0, s, t->data, t->dtype, H, d, dp, cols, dst);
    GILL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  int resnet(const std::string& p, int cin, int cout, int hw, int temb_dim, int* temb_off, bf16_t* temb_w, float* temb_b,
             ResnetW* r, bool f8) {
    r->cin = cin; r->cout = cout;
    f8 = f8 && cin % 64 == 0 && cout % 64 == 0;
    GILL_TRY(norm(p + ".norm1", cin, &r->n1));
    GILL_TRY(conv3(p + ".conv1", cin, cout, hw, &r->c1, f8));
    GILL_TRY(norm(p + ".norm2", cout, &r->n2));
    GILL_TRY(conv3(p + ".conv2", cout, cout, hw, &r->c2, f8));
    r->has_sc = (cin != cout);
    if (r->has_sc) GILL_TRY(lin(p + ".conv_shortcut", cout, cin, &r->sc));
    if (r->has_sc && !f8) {   // (fp8 mode: the 1x1 shortcut stays a bf16 GEMM of its own whose output is conv2's residual)
      // fused weight rows: [conv2 taps (9*cout) | shortcut (cin)]
      const int kf = 9 * cout + cin;
      std::vector<int32_t> ident(cout);
      for (int i = 0; i < cout; ++i) ident[i] = i;
      int32_t* idx;
      GILL_TRY(pool.alloc(&idx, (size_t)cout, false));
      GILL_CHECK_HIP(hipMemcpy(idx, ident.data(), sizeof(int32_t) * cout, hipMemcpyHostToDevice));
      GILL_TRY(pool.alloc(&r->c2f_w, (size_t)cout * kf, false));
      GILL_TRY(scatter_rows_bf16_launch(r->c2.w, cout, 9 * cout, idx, r->c2f_w, kf, s));
      GILL_TRY(scatter_rows_bf16_launch(r->sc.w, cout, cin, idx, r->c2f_w + 9 * cout, kf, s));
      GILL_TRY(pool.alloc(&r->c2f_b, (size_t)cout, false));
      hipLaunchKernelGGL(vec_add_f32_kernel, dim3(cdiv(cout, 256)), dim3(256), 0, s, r->c2.b, r->sc.b, cout, r->c2f_b);
      GILL_CHECK_HIP(hipGetLastError());
    }
    // time_emb_proj rows go into the shared [sum Cout][temb_dim] matrix
    r->temb_off = *temb_off;
    const gill_tensor* t;
    GILL_TRY(wt.get(p + ".time_emb_proj.weight", (int64_t)cout * temb_dim, &t));
    GILL_TRY(convert_to_bf16_launch(t->data, t->dtype, (int64_t)cout * temb_dim, temb_w + (size_t)r->temb_off * temb_dim, s));
    GILL_TRY(wt.get(p + ".time_emb_proj.bias", cout, &t));
    GILL_TRY(convert_to_f32_launch(t->data, t->dtype, cout, temb_b + r->temb_off, s));
    *temb_off += cout;
    return 0;
  }
  // hw: tokens per sample at this layer (the softmax GEMM of the two-GEMM cross-attention runs on 64-row tiles of ONE sample)
  int xf(const std::string& p, int C, int H, int ctx_dim, int layer_id, int hw, XfW* x, bool f8 = false) {
    x->C = C; x->heads = H; x->d = C / H; x->dp = attn_padded_dim(x->d); x->dpv = round_up(x->dp, 32); x->layer_id = layer_id;
    GILL_REQUIRE(x->dp > 0, "unsupported attention head dim");
    const int hdp = H * x->dp;
    GILL_REQUIRE(hdp % 64 == 0, "padded attention width must be a multiple of 64");
    GILL_TRY(norm(p + ".norm", C, &x->gn));
    GILL_TRY(lin(p + ".proj_in", C, C, &x->proj_in));
    GILL_TRY(lin(p + ".proj_out", C, C, &x->proj_out));
    const std::string b = p + ".transformer_blocks.0";
    GILL_TRY(norm(b + ".norm1", C, &x->ln1));
    GILL_TRY(norm(b + ".norm2", C, &x->ln2));
    GILL_TRY(norm(b + ".norm3", C, &x->ln3));
    GILL_TRY(pool.alloc(&x->wqkv1, (size_t)3 * hdp * C, true));
    GILL_TRY(head_rows(b + ".attn1.to_q.weight", H, x->d, x->dp, C, x->wqkv1));
    GILL_TRY(head_rows(b + ".attn1.to_k.weight", H, x->d, x->dp, C, x->wqkv1 + (size_t)hdp * C));
    GILL_TRY(head_rows(b + ".attn1.to_v.weight", H, x->d, x->dp, C, x->wqkv1 + (size_t)2 * hdp * C));
    GILL_TRY(pool.alloc(&x->wq2, (size_t)hdp * C, true));
    GILL_TRY(head_rows(b + ".attn2.to_q.weight", H, x->d, x->dp, C, x->wq2));
    GILL_TRY(pool.alloc(&x->wkv2, (size_t)2 * hdp * ctx_dim, true));
    GILL_TRY(head_rows(b + ".attn2.to_k.weight", H, x->d, x->dp, ctx_dim, x->wkv2));
    GILL_TRY(head_rows(b + ".attn2.to_v.weight", H, x->d, x->dp, ctx_dim, x->wkv2 + (size_t)hdp * ctx_dim));
    for (int a = 1; a <= 2; ++a) {
      LinW* o = (a == 1) ? &x->out1 : &x->out2;
      const std::string on = b + ".attn" + std::to_string(a) + ".to_out.0";
      o->out = C; o->in = hdp;
      const gill_tensor* t;
      GILL_TRY(wt.get(on + ".weight", (int64_t)C * C, &t));
      GILL_TRY(pool.alloc(&o->w, (size_t)C * hdp, true));
      hipLaunchKernelGGL(pad_head_cols_kernel, dim3(1024), dim3(256), 0, s, t->data, t->dtype, C, H, x->d, x->dp, o->w);
      GILL_CHECK_HIP(hipGetLastError());
      GILL_TRY(load_f32(wt, pool, on + ".bias", C, &o->b, s));
    }
    // GEGLU projection: permute rows (value/gate 16-row interleave)
    {
      const int inner = 4 * C;
      bf16_t* tmpw; float* tmpb; int32_t* idx;
      GILL_TRY(load_bf16(wt, pool, b + ".ff.net.0.proj.weight", (int64_t)2 * inner * C, &tmpw, s));
      GILL_TRY(load_f32(wt, pool, b + ".ff.net.0.proj.bias", 2 * inner, &tmpb, s));
      std::vector<int32_t> map = geglu_row_permutation(inner);
      GILL_TRY(pool.alloc(&idx, map.size(), false));
      GILL_CHECK_HIP(hipMemcpy(idx, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice));
      GILL_TRY(pool.alloc(&x->wff1, (size_t)2 * inner * C, false));
      GILL_TRY(pool.alloc(&x->bff1, (size_t)2 * inner, false));
      GILL_TRY(scatter_rows_bf16_launch(tmpw, 2 * inner, C, idx, x->wff1, C, s));
      GILL_TRY(permute_f32_launch(tmpb, idx, 2 * inner, x->bff1, s));
    }
    GILL_TRY(lin(b + ".ff.net.2", C, 4 * C, &x->ff2));
    {
      const gill_tensor *tp, *t2, *tb2, *tbp;
      GILL_TRY(wt.get(p + ".proj_out.weight", (int64_t)C * C, &tp));
      GILL_TRY(wt.get(b + ".ff.net.2.weight", (int64_t)C * 4 * C, &t2));
      GILL_TRY(wt.get(b + ".ff.net.2.bias", C, &tb2));
      GILL_TRY(wt.get(p + ".proj_out.bias", C, &tbp));
      GILL_TRY(pool.alloc(&x->wfo, (size_t)C * 5 * C, false));
      GILL_TRY(pool.alloc(&x->bfo, (size_t)C, false));
      hipLaunchKernelGGL(ffo_fuse_kernel, dim3(cdiv(5 * C + 1, 256), C), dim3(256), 0, s, tp->data, tp->dtype, t2->data, t2->dtype,
                         tb2->data, tb2->dtype, tbp->data, tbp->dtype, C, x->wfo, x->bfo);
      GILL_CHECK_HIP(hipGetLastError());
    }
    // fold the three LayerNorms into the projections that consume them
    GILL_TRY(pool.alloc(&x->s_qkv1, (size_t)3 * hdp)); GILL_TRY(pool.alloc(&x->c_qkv1, (size_t)3 * hdp));
    GILL_TRY(pool.alloc(&x->s_q2, (size_t)hdp)); GILL_TRY(pool.alloc(&x->c_q2, (size_t)hdp));
    GILL_TRY(pool.alloc(&x->s_ff1, (size_t)8 * C));
    GILL_TRY(ln_fold_rows_launch(x->wqkv1, 3 * hdp, C, x->ln1.g, x->ln1.b, x->s_qkv1, x->c_qkv1, s));
    GILL_TRY(ln_fold_rows_launch(x->wq2, hdp, C, x->ln2.g, x->ln2.b, x->s_q2, x->c_q2, s));
    GILL_TRY(ln_fold_rows_launch(x->wff1, 8 * C, C, x->ln3.g, x->ln3.b, x->s_ff1, x->bff1, s));
    // fp8 mode (BASELINE configs[4]): the GEGLU projection of the blocks that run it as a GEMM (levels 1-3; level 0 has the fused feed-forward
    // kernel) on the fp8 matrix instruction — the folded rows quantised per output row
    if (f8 && C % 128 == 0 && !(ffn_fused_on() && ffn_fused_supported(C, 128))) {
      GILL_TRY(pool.alloc(&x->wff1_8, (size_t)8 * C * C, false));
      GILL_TRY(pool.alloc(&x->cs_ff1, (size_t)8 * C, false));
      GILL_TRY(linear_weight_quant_fp8_launch(x->wff1, 8 * C, C, F8_LIN_ACT_SCALE, x->wff1_8, x->cs_ff1, s));
    }
    // cross-attention as two GEMMs: where the 80 key slots per head are no wider than the head itself (d >= 80: SD-1.5 levels 1-3)
    // ... and a sample is whole 64-row tiles (per-sample weights: a tile must not straddle samples — sample_size 32 / 96 have 16- / 144-token
    // mid blocks); otherwise the layer keeps its K / V caches and the attention-kernel form
    if (xalg_on() && H % 2 == 0 && ctx_len <= 80 && 80 * H <= hdp && x->dp <= 160 && C % 64 == 0 && ctx_dim % 64 == 0 && hw % 64 == 0) {
      GILL_TRY(pool.alloc(&x->xg, (size_t)2 * H * C * ctx_dim, false));
      GILL_TRY(pool.alloc(&x->xgb, (size_t)H * ctx_dim));
      const float qs = 1.4426950408889634f / sqrtf((float)x->d);
      hipLaunchKernelGGL(xalg_fold_kernel, dim3(2 * H * C / 8), dim3(256), 0, s, x->wq2, x->wkv2, x->out2.w, H, C, x->dp, ctx_dim, qs, x->xg);
      GILL_CHECK_HIP(hipGetLastError());
      hipLaunchKernelGGL(xalg_fold_bias_kernel, dim3(cdiv(ctx_dim, 256), H), dim3(256), 0, s, x->c_q2, x->wkv2, x->dp, ctx_dim, qs, x->xgb);
      GILL_CHECK_HIP(hipGetLastError());
    }
    if (ffn_fused_on() && ffn_fused_supported(C, 128)) {
      GILL_TRY(pool.alloc(&x->w1c, (size_t)8 * C * C, false));
      GILL_TRY(pool.alloc(&x->b1c, (size_t)8 * C, false));
      GILL_TRY(pool.alloc(&x->w2p, (size_t)4 * C * C, false));
      if (hdp == 384) GILL_TRY(pool.alloc(&x->wpp, (size_t)C * C, false));
      GILL_TRY(ffn_relayout_launch(x->wff1, x->bff1, x->wfo, x->w1c, x->b1c, x->w2p, x->wpp, s));
      x->w1c_kperm = x->wpp != nullptr;
    }
    if (lnproj_on() && lnproj_supported(C, 128, H, x->dp)) {
      GILL_TRY(pool.alloc(&x->wqkv1p, (size_t)3 * hdp * C, false));
      GILL_TRY(pool.alloc(&x->wq2p, (size_t)hdp * C, false));
      GILL_TRY(lnproj_kperm_launch(x->wqkv1, 3 * hdp, x->wqkv1p, s));
      GILL_TRY(lnproj_kperm_launch(x->wq2, hdp, x->wq2p, s));
    }
    return 0;
  }
};

}  // namespace

static int unet_plan_and_alloc(gill_unet* m);

extern "C" int gill_unet_create(gill_unet** out, const gill_unet_config* cfg, const gill_tensor* weights, int n_weights) {
  GILL_REQUIRE(out && cfg && weights, "null argument");
  GILL_REQUIRE(cfg->layers_per_block == 2, "only layers_per_block == 2 (SD-1.x/2.x) is supported");
  GILL_REQUIRE(cfg->max_batch >= 1, "max_batch must be >= 1");
  for (int i = 0; i < 4; ++i)
    GILL_REQUIRE(cfg->block_out_channels[i] % 64 == 0, "block_out_channels must be multiples of 64");
  GILL_REQUIRE(cfg->cross_attention_dim % 64 == 0, "cross_attention_dim must be a multiple of 64");
  GILL_REQUIRE(cfg->sample_size % 8 == 0, "sample_size must be a multiple of 8");
  gill_unet* m = new gill_unet();
  m->cfg = *cfg;
  int rc = 0;
  auto fail = [&](int r) { delete m; return r; };
  WeightTable wt(weights, n_weights);
  hipStream_t s = nullptr;
  Loader L{wt, m->pool, s, cfg->ctx_len};
  const int* ch = cfg->block_out_channels;
  const int ctxd = cfg->cross_attention_dim;
  // heads per resolution level: SD-1.x uses num_heads everywhere, SD-2.x a fixed head dim of 64 (5, 10, 20, 20 heads)
  int Hl[4];
  for (int i = 0; i < 4; ++i) {
    Hl[i] = cfg->heads_per_level[i] > 0 ? cfg->heads_per_level[i] : cfg->num_heads;
    GILL_REQUIRE(Hl[i] > 0 && ch[i] % Hl[i] == 0, "heads must divide the channel count of their level");
  }
  const int temb_dim = ch[0] * 4;
  m->temb_dim = temb_dim;

  // total width of the per-resnet time projections
  int temb_total = 0;
  for (int i = 0; i < 4; ++i) temb_total += 2 * ch[i];
  temb_total += 2 * ch[3];
  for (int i = 0; i < 4; ++i) temb_total += 3 * ch[3 - i];
  m->temb_total = temb_total;
  if ((rc = m->pool.alloc(&m->temb_proj_w, (size_t)temb_total * temb_dim, false))) return fail(rc);
  if ((rc = m->pool.alloc(&m->temb_proj_b, (size_t)temb_total, false))) return fail(rc);
  int temb_off = 0;

  // conv_in / conv_out (direct kernels, [Cout][9][Cin] layout as well)
  {
    const gill_tensor* t;
    GILL_REQUIRE(cfg->in_channels * 9 <= 64, "conv_in: in_channels * 9 must fit one 64-wide K step");
    if ((rc = wt.get("conv_in.weight", (int64_t)ch[0] * cfg->in_channels * 9, &t))) return fail(rc);
    {
      // conv_in runs as im2col (K = 36 zero-padded to 64) + the MFMA GEMM: weights [Cout][tap*Cin + c] padded to [Cout][64]
      bf16_t* tmp; int32_t* idx;
      const int kk = cfg->in_channels * 9;
      if ((rc = m->pool.alloc(&tmp, (size_t)ch[0] * kk, false))) return fail(rc);
      if ((rc = conv_weight_relayout_launch(t->data, t->dtype, ch[0], cfg->in_channels, tmp, s))) return fail(rc);
      if ((rc = m->pool.alloc(&m->conv_in_w, (size_t)ch[0] * 64, true))) return fail(rc);
      std::vector<int32_t> rows(ch[0]);
      for (int i = 0; i < ch[0]; ++i) rows[i] = i;
      if ((rc = m->pool.alloc(&idx, (size_t)ch[0], false))) return fail(rc);
      if (hipMemcpy(idx, rows.data(), sizeof(int32_t) * ch[0], hipMemcpyHostToDevice) != hipSuccess) return fail(-1);
      if ((rc = scatter_rows_bf16_launch(tmp, ch[0], kk, idx, m->conv_in_w, 64, s))) return fail(rc);
    }
    if ((rc = load_f32(wt, m->pool, "conv_in.bias", ch[0], &m->conv_in_b, s))) return fail(rc);
    if ((rc = wt.get("conv_out.weight", (int64_t)cfg->out_channels * ch[0] * 9, &t))) return fail(rc);
    if ((rc = m->pool.alloc(&m->conv_out_w, (size_t)cfg->out_channels * ch[0] * 9, false))) return fail(rc);
    if ((rc = conv_weight_relayout_launch(t->data, t->dtype, cfg->out_channels, ch[0], m->conv_out_w, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, "conv_out.bias", cfg->out_channels, &m->conv_out_b, s))) return fail(rc);
  }
  if ((rc = L.norm("conv_norm_out", ch[0], &m->norm_out))) return fail(rc);
  if ((rc = L.lin("time_embedding.linear_1", temb_dim, ch[0], &m->te1))) return fail(rc);
  if ((rc = L.lin("time_embedding.linear_2", temb_dim, temb_dim, &m->te2))) return fail(rc);

  int layer_id = 0;
  const bool f8 = cfg->fp8_convs != 0;   // ResnetBlock2D convolutions on the fp8 matrix instruction (conv_fp8.hip)
  const bool f8lin = cfg->fp8_convs == 1; // ... and the GEGLU projections of levels 1-3 (linear_fp8.hip); fp8_convs = 2: the convolutions only (round-5 mode, for A/B)
  auto hw_of = [&](int level) { const int side = cfg->sample_size >> level; return side * side; };   // pixels per sample
  // down blocks: CrossAttnDownBlock2D x3, DownBlock2D
  for (int i = 0; i < 4; ++i) {
    const int cin = (i == 0) ? ch[0] : ch[i - 1];
    const std::string p = "down_blocks." + std::to_string(i);
    m->down_res[i].resize(2);
    for (int j = 0; j < 2; ++j)
      if ((rc = L.resnet(p + ".resnets." + std::to_string(j), j == 0 ? cin : ch[i], ch[i], hw_of(i), temb_dim, &temb_off,
                         m->temb_proj_w, m->temb_proj_b, &m->down_res[i][j], f8))) return fail(rc);
    if (i < 3) {
      m->down_xf[i].resize(2);
      for (int j = 0; j < 2; ++j)
        if ((rc = L.xf(p + ".attentions." + std::to_string(j), ch[i], Hl[i], ctxd, layer_id++, hw_of(i), &m->down_xf[i][j], f8lin))) return fail(rc);
      if ((rc = L.conv3(p + ".downsamplers.0.conv", ch[i], ch[i], hw_of(i), &m->down_ds[i]))) return fail(rc);
    }
  }
  // mid
  if ((rc = L.resnet("mid_block.resnets.0", ch[3], ch[3], hw_of(3), temb_dim, &temb_off, m->temb_proj_w, m->temb_proj_b, &m->mid_res[0], f8)))
    return fail(rc);
  if ((rc = L.xf("mid_block.attentions.0", ch[3], Hl[3], ctxd, layer_id++, hw_of(3), &m->mid_xf, f8lin))) return fail(rc);
  if ((rc = L.resnet("mid_block.resnets.1", ch[3], ch[3], hw_of(3), temb_dim, &temb_off, m->temb_proj_w, m->temb_proj_b, &m->mid_res[1], f8)))
    return fail(rc);
  // up blocks: UpBlock2D, CrossAttnUpBlock2D x3
  const int rev[4] = {ch[3], ch[2], ch[1], ch[0]};
  for (int i = 0; i < 4; ++i) {
    const int outc = rev[i];
    const int prev = (i == 0) ? rev[0] : rev[i - 1];
    const int inc = rev[i + 1 < 4 ? i + 1 : 3];
    const std::string p = "up_blocks." + std::to_string(i);
    m->up_res[i].resize(3);
    for (int j = 0; j < 3; ++j) {
      const int skip = (j == 2) ? inc : outc;
      const int rin = (j == 0) ? prev : outc;
      if ((rc = L.resnet(p + ".resnets." + std::to_string(j), rin + skip, outc, hw_of(3 - i), temb_dim, &temb_off, m->temb_proj_w,
                         m->temb_proj_b, &m->up_res[i][j], f8))) return fail(rc);
    }
    if (i > 0) {
      m->up_xf[i].resize(3);
      for (int j = 0; j < 3; ++j)
        if ((rc = L.xf(p + ".attentions." + std::to_string(j), outc, Hl[3 - i], ctxd, layer_id++, hw_of(3 - i), &m->up_xf[i][j], f8lin))) return fail(rc);
    }
    if (i < 3)
      if ((rc = L.conv3(p + ".upsamplers.0.conv", outc, outc, hw_of(3 - i), &m->up_us[i], false, true))) return fail(rc);
  }
  m->n_xf = layer_id;
  if (temb_off != temb_total) { gill_set_error("internal: temb table width mismatch"); return fail(-4); }

  if ((rc = unet_plan_and_alloc(m))) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { gill_set_error("unet create: device sync failed"); return fail(-1); }
  *out = m;
  return 0;
}

extern "C" void gill_unet_destroy(gill_unet* h) { delete h; }

// ------------------------------------------------------------------------------------------------------------------
namespace {

struct UNetRun {
  gill_unet* m;
  hipStream_t s;
  int Bx;
  const float* temb_rows;   // row for sample 0
  int temb_bstride;         // 0: every sample uses the same row
  bool dry;
  // classifier-free-guidance pair: samples b and b + Bx/2 carry the same latents and timestep and differ only in the prompt,
  // so everything before the first cross-attention runs once on the first half (gill_sd_denoise sets this)
  bool cfg_pair = false;
  // GILL_DEBUG_SYNC=1 (tools): synchronise after every launch of the forward and say which one it was, to attribute a device fault
  int dbg_sync(const char* what, int a = 0, int b = 0, int c = 0) {
    static const bool on = getenv("GILL_DEBUG_SYNC") != nullptr;
    if (!on || dry) return 0;
    GILL_CHECK_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[unet] ok: %s %d %d %d\n", what, a, b, c);
    return 0;
  }
  float* stats_slot(size_t floats) {   // next slot of the per-forward GroupNorm partial-sum pool (the dry run sizes it)
    float* p = dry ? (float*)(uintptr_t)16 : m->gn_stats + m->gn_next;
    m->gn_next += (floats + 3) & ~(size_t)3;
    return p;
  }
  RowStats ln_slot(int rows, int C) {   // row-sum planes of a [rows][C] residual stream (sized for any tiling of C columns)
    RowStats r;
    r.p = dry ? (float*)(uintptr_t)16 : m->ln_stats + m->ln_next;
    m->ln_next += (size_t)rows * 2 * GEMM_MAX_ROW_PLANES(C);
    return r;
  }
  Tensor talloc(int H, int W, int C, bool want_stats = false) {
    Tensor t; t.H = H; t.W = W; t.C = C;
    t.p = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)Bx * H * W * C);
    if (want_stats && (H * W) % GN_SLAB_ROWS == 0 && C % m->cfg.norm_num_groups == 0) {
      const int sbin = (C % 64 == 0 && C / 64 >= 2) ? C / 64 : C / m->cfg.norm_num_groups;
      if (gemm_fused_gn_ok(C, sbin)) {
        t.sbin = sbin;
        t.stats = stats_slot((size_t)Bx * (H * W / GN_SLAB_ROWS_MIN) * (C / sbin) * 2);
      }
    }
    return t;
  }
  Tensor talloc8(int H, int W, int C) {   // fp8 activation tensor (one byte per element)
    Tensor t; t.H = H; t.W = W; t.C = C;
    t.p = (bf16_t*)m->arena.alloc((size_t)Bx * H * W * C);
    return t;
  }
  void fuse_stats(GemmArgs& g, const Tensor& y) {
    if (!y.stats) return;
    g.gn_stats = y.stats; g.gn_groups = y.C / y.sbin; g.gn_cg = y.sbin;
    g.rows_per_batch = y.H * y.W;
  }
  int pick_sk(GemmArgs& g, bool generic = false) {
    g.splitk = gemm_pick_splitk(g.M, g.N, g.K, g.act, !g.conv, generic);
    while (g.splitk > 1 && (size_t)g.splitk * g.M * g.N > m->splitk_ws_floats) --g.splitk;
    g.ws = m->splitk_ws;
    return 0;
  }
  // The GroupNorm (+ SiLU) that consumes a GEMM's output, offered to the producer: a split-K launch of a supported geometry runs it in
  // its reducer (gemm.hip "REDUCE + GROUPNORM") and sets `done`; otherwise the consumer runs its GroupNorm-apply as before.
  // raw_needed = false: nobody else reads the raw output (conv1 -> norm2 inside a ResnetBlock2D): it is not even written.
  // ss: (optional) the consumer wants the per-(sample, channel) scale | shift table [Bx][2][C] instead of a normalised copy (the level-0 transformer
  // blocks: lnproj.hip applies it to the rows it loads) — only the in-kernel finish (COOP) writes it; y is then unused.
  struct FusedNorm { const NormW* n; float eps; int silu; Tensor y; bool raw_needed; bool done; float* ss = nullptr; };
  int gemm(GemmArgs& g, RowStats* rs = nullptr, Tensor* ys = nullptr, FusedNorm* fn = nullptr) {
    // COOP counters of this launch: the same slot range in the dry run and in every real run (whether or not the launch ends up using them)
    unsigned* ctr = dry ? nullptr : m->coop_ctr + m->coop_next;
    m->coop_next += (size_t)gemm_coop_counters(g);
    if (dry) return 0;
    GILL_REQUIRE(m->coop_next <= m->coop_n, "internal: COOP counter pool exhausted");
    pick_sk(g);
    // (split-K partials come from 128-row tiles: not where a sample's rows are fewer — the 8 x 8 maps of the mid block)
    if (g.out_mode == OUT_SOFTMAX80 || (g.wb_rows && g.wb_rows % 128 != 0)) g.splitk = 1;
    g.coop_ctr = ctr;
    g.coop_splitk = gemm_coop_mode() >= 2;      // (the split-K finish in-kernel: a measured no-go, opt-in — GemmArgs::coop_splitk)
    if (fn) {
      g.rows_per_batch = fn->y.H * fn->y.W;
      g.fn_Y = fn->ss ? nullptr : fn->y.p; g.fn_ss = fn->ss;
      g.fn_gamma = fn->n->g; g.fn_beta = fn->n->b; g.fn_eps = fn->eps; g.fn_silu = fn->silu;
      g.fn_cg = g.N / m->cfg.norm_num_groups;
      // the finish inside the producing launch (gemm.hip "COOP"), else — split-K only — the reducer launch that also normalises
      if (gemm_coop_ok(g) || (g.splitk > 1 && !fn->ss && gemm_fused_norm_ok(g))) {
        fn->done = true;
        if (!fn->raw_needed) {
          g.C = nullptr;
          if (g.splitk > 1) { g.gn_stats = nullptr; if (ys) ys->stats = nullptr; }      // (splitk == 1: the partials ARE the hand-off between the workgroups)
        }
      } else {
        g.fn_Y = nullptr; g.fn_ss = nullptr;
      }
    }
    if (rs) { g.row_stats = rs->p; rs->planes = gemm_row_planes(g); }
    if (ys && ys->stats) ys->nslab = ys->H * ys->W / gemm_gn_slab_rows(g);
    // (round 4: touching every GEMM's weights right before it is worth 3.4 % of a forward's kernel time, and a side-stream prefetcher costs 4.5-15 %:
    // profiles/r04_weight_touch.md, r04_weight_prefetch.md; the probe switch is gone)
    GILL_TRY(gemm_launch(g, s));
    return dbg_sync(g.conv ? "conv" : (g.act == ACT_GEGLU ? "geglu" : (g.out_mode == OUT_QKV ? "qkv" : (g.out_mode == OUT_SOFTMAX80 ? "scores+softmax" : "gemm"))), g.M, g.N, g.K);
  }
  // y8_scale > 0: y holds fp8(y8_scale * value) instead of bf16 (same shape; the A operand of conv8())
  // ss_out: write the per-(sample, channel) scale / shift table instead of normalising (single-source inputs whose producer filed the
  // statistics; *folded says whether that was possible — if not, y is normalised as usual)
  int gnorm(const Tensor& x1, const Tensor* x2, const NormW& n, float eps, int silu, const Tensor& y, float y8_scale = 0.f,
            float* ss_out = nullptr, bool* folded = nullptr) {
    // single-source input whose producer already accumulated the sums: no statistics pass
    const int C = x1.C + (x2 ? x2->C : 0);
    const bool ready = x1.stats != nullptr && (x2 == nullptr || x2->stats != nullptr) &&
                       groupnorm_bins_align(C / m->cfg.norm_num_groups, x1.C, x1.sbin, x2 ? x2->sbin : 0);
    const int HW = x1.H * x1.W;
    float* stats = ready ? nullptr : stats_slot(groupnorm_stats_floats(Bx, HW, m->cfg.norm_num_groups));
    // out-of-place totals for producers that wrote more than 64 partials per bin (SD-2.1-768: 96x96 maps); a skip tensor's
    // partials are read again by the up block's concatenated norm1 and must stay as their producer wrote them
    float* tot = ready ? stats_slot(groupnorm_totals_floats(Bx, x1.C / x1.sbin, x2 ? x2->C / x2->sbin : 0)) : nullptr;
    if (dry) return 0;
    GILL_REQUIRE(m->gn_next <= m->gn_floats, "internal: GroupNorm stats pool exhausted");
    GILL_TRY(dbg_sync("before groupnorm", x1.C, x2 ? x2->C : 0, HW));
    if (ready) {
      const bool fold = ss_out != nullptr && x2 == nullptr;
      if (folded) *folded = fold;
      return groupnorm_apply_launch(x1.p, x1.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, Bx, HW, m->cfg.norm_num_groups,
                                    n.g, n.b, eps, silu, y.p, x1.stats, x1.sbin, x1.C, x1.nslab,
                                    x2 ? x2->stats : nullptr, x2 ? x2->sbin : 0, x2 ? x2->nslab : 0, s, y8_scale, tot, fold ? ss_out : nullptr);
    }
    return groupnorm_launch(x1.p, x1.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, Bx, HW, m->cfg.norm_num_groups, n.g,
                            n.b, eps, silu, y.p, stats, s, y8_scale);
  }
  // 3x3 conv (pad 1, stride 1) of an fp8 activation tensor x8 (gnorm(..., F8_ACT_SCALE)) with fp8 weights:
  // y = conv + bias + rowvec + resid (resid may alias y.p), GroupNorm partials of y fused like conv()
  int conv8(const Tensor& x8, const ConvW& w, const float* rowvec, int rv_bstride, const bf16_t* resid, Tensor& y) {
    if (dry) return 0;
    ConvF8Args a;
    a.B = Bx; a.H = x8.H; a.W = x8.W; a.Cin = w.cin; a.M = Bx * x8.H * x8.W; a.N = w.cout; a.Kpad = w.kpad;
    a.A8 = reinterpret_cast<const unsigned char*>(x8.p); a.W8 = w.w8; a.colscale = w.cs; a.C = y.p;
    a.rows_per_batch = x8.H * x8.W;
    GemmArgs g;      // split-K geometry + the reducer's epilogue
    g.M = a.M; g.N = a.N; g.K = 9 * w.cin;
    pick_sk(g, true);     // (the fp8 kernel has neither 64-row tiles nor a ping-pong variant: generic split rule)
    a.splitk = g.splitk;
    if (a.splitk > 1) {
      a.ws = g.ws;
      GILL_TRY(conv3x3_fp8_launch(a, s));
      g.bias = w.b; g.rowvec = rowvec; g.rows_per_batch = a.rows_per_batch; g.rowvec_bstride = rv_bstride;
      g.resid = resid; g.ldr = w.cout; g.C = y.p; g.ldc = w.cout;
      fuse_stats(g, y);
      if (y.stats) y.nslab = y.H * y.W / gemm_gn_slab_rows(g);
      return gemm_splitk_reduce_launch(g, s);
    }
    a.bias = w.b; a.rowvec = rowvec; a.rowvec_bstride = rv_bstride; a.resid = resid;
    if (y.stats) { a.gn_stats = y.stats; a.gn_groups = y.C / y.sbin; a.gn_cg = y.sbin; y.nslab = y.H * y.W / GN_SLAB_ROWS; }
    return conv3x3_fp8_launch(a, s);
  }
  // 3x3 conv (pad 1) over x1 (++ x2): stride 1|2, optional fused nearest-2x upsample
  int conv(const Tensor& x1, const Tensor* x2, const ConvW& w, int stride, int ups, const float* rowvec, int rv_bstride,
           const bf16_t* resid, Tensor& y, FusedNorm* fn = nullptr) {
    GemmArgs g;
    g.conv = 1; g.IH = x1.H; g.IW = x1.W; g.OH = y.H; g.OW = y.W; g.Cin = w.cin; g.stride = stride; g.ups = ups;
    g.M = Bx * y.H * y.W; g.N = w.cout; g.K = 9 * w.cin;
    g.A = x1.p; g.A2 = x2 ? x2->p : nullptr; g.K1 = x1.C;
    g.W = w.w; g.bias = w.b; g.k_chunked = w.chunked;
    if (ups && w.ups4) { g.ups = 2; g.K = 4 * w.cin; }
    g.rowvec = rowvec; g.rows_per_batch = y.H * y.W; g.rowvec_bstride = rv_bstride;
    g.resid = resid; g.ldr = w.cout;
    g.C = y.p; g.ldc = w.cout;
    fuse_stats(g, y);
    return gemm(g, nullptr, &y, fn);
  }
  int linear(const bf16_t* A, int lda, const bf16_t* A2, int lda2, int K1, int M, const bf16_t* W, const float* b, int N,
             int K, const bf16_t* resid, int act, bf16_t* out, int ldc, Tensor* ystats = nullptr,
             RowStats* row_stats = nullptr, FusedNorm* fn = nullptr) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.K1 = K1; g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.W = W; g.bias = b;
    g.resid = resid; g.ldr = N; g.act = act; g.C = out; g.ldc = ldc;
    if (ystats) fuse_stats(g, *ystats);
    return gemm(g, row_stats, ystats, fn);
  }

  // out_stats: the output feeds a single-source GroupNorm next (accumulate its sums in conv2's epilogue)
  // pre: x1 already normalised by its producer's reducer (norm1 + SiLU of THIS block: single-source input only);
  // next: the GroupNorm that consumes this block's output, offered to conv2's reducer
  int resnet(const Tensor& x1, const Tensor* x2, const ResnetW& w, Tensor* out, bool out_stats, const Tensor* pre = nullptr,
             FusedNorm* next = nullptr) {
    const int H = x1.H, Wd = x1.W;
    *out = talloc(H, Wd, w.cout, out_stats);
    const size_t mk = m->arena.mark();
    if (w.c1.w8) {
      // fp8 mode: both GroupNorm-apply passes write e4m3 (half the bytes of the bf16 tensors they replace) and both convs run on
      // the fp8 matrix instruction; the 1x1 shortcut (bf16 GEMM over the raw input) lands in `out` first and conv2 adds onto it
      Tensor n1 = talloc8(H, Wd, w.cin);
      GILL_TRY(gnorm(x1, x2, w.n1, 1e-5f, 1, n1, F8_ACT_SCALE));
      Tensor h = talloc(H, Wd, w.cout, true);   // -> norm2
      GILL_TRY(conv8(n1, w.c1, temb_rows ? temb_rows + w.temb_off : nullptr, temb_bstride, nullptr, h));
      Tensor n2 = talloc8(H, Wd, w.cout);
      GILL_TRY(gnorm(h, nullptr, w.n2, 1e-5f, 1, n2, F8_ACT_SCALE));
      const bf16_t* resid = x1.p;
      if (w.has_sc) {
        GILL_TRY(linear(x1.p, x1.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, x1.C, Bx * H * Wd, w.sc.w, w.sc.b, w.cout, w.cin, nullptr, 0,
                        out->p, w.cout));
        resid = out->p;
      }
      GILL_TRY(conv8(n2, w.c2, nullptr, 0, resid, *out));
      m->arena.release(mk);
      return 0;
    }
    Tensor n1 = talloc(H, Wd, w.cin);
    if (pre && !x2) n1 = *pre;
    else GILL_TRY(gnorm(x1, x2, w.n1, 1e-5f, 1, n1));
    Tensor h = talloc(H, Wd, w.cout, true);   // -> norm2
    Tensor n2 = talloc(H, Wd, w.cout);
    FusedNorm f2{&w.n2, 1e-5f, 1, n2, false, false};      // norm2 + SiLU in conv1's split-K reducer where there is one
    GILL_TRY(conv(n1, nullptr, w.c1, 1, 0, temb_rows ? temb_rows + w.temb_off : nullptr, temb_bstride, nullptr, h, &f2));
    if (!f2.done) GILL_TRY(gnorm(h, nullptr, w.n2, 1e-5f, 1, n2));
    if (w.has_sc) {
      // out = conv2(n2) + conv_shortcut(x1 ++ x2): ONE implicit GEMM whose K runs over the 9 taps of n2 and then over
      // the raw input channels (no separate 1x1 GEMM, no shortcut tensor written and re-read as a residual)
      GemmArgs g;
      g.conv = 1; g.IH = H; g.IW = Wd; g.OH = H; g.OW = Wd; g.Cin = w.cout; g.stride = 1; g.ups = 0;
      g.M = Bx * H * Wd; g.N = w.cout; g.K = 9 * w.cout + w.cin;
      g.A = n2.p; g.K1 = w.cout;
      g.X1 = x1.p; g.X2 = x2 ? x2->p : nullptr; g.KX = w.cin; g.KX1 = x1.C;
      g.W = w.c2f_w; g.bias = w.c2f_b; g.k_chunked = w.c2.chunked;
      g.rows_per_batch = H * Wd;
      g.C = out->p; g.ldc = w.cout;
      fuse_stats(g, *out);
      GILL_TRY(gemm(g, nullptr, out, next));
    } else {
      GILL_TRY(conv(n2, nullptr, w.c2, 1, 0, nullptr, 0, x1.p, *out, next));
    }
    m->arena.release(mk);
    return 0;
  }

  int attend(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int nq, int nkv, int nq_pad, int nkv_pad,
             const XfW& w) {
    if (dry) return 0;
    AttnArgs a;
    a.Q = q; a.K = k; a.Vt = vt; a.O = o;
    a.B = Bx; a.H = w.heads; a.nq = nq; a.nkv = nkv; a.nq_pad = nq_pad; a.nkv_pad = nkv_pad;
    a.dp = w.dp; a.dpv = w.dpv; a.ldo = w.heads * w.dp; a.d = w.d;
    a.scale = 1.0f / sqrtf((float)w.d);
    return attention_launch(a, s);
  }

  // does xf() take its GroupNorm as a table (the fused projection kernel applies it on load)?  Mirrors the conditions inside xf().
  bool xf_wants_table(const XfW& w, int HW, bool shared) const {
    return w.wqkv1p != nullptr && !shared && lnproj_supported(w.C, Bx * HW, w.heads, w.dp) && HW % 128 == 0;
  }

  // shared: `x` (and Bx on entry) cover only the first half of a CFG pair; the block runs at that half batch up to the end of
  // the self-attention, then the residual stream is duplicated and the rest runs on the full pair (Bx restored on return)
  // pre: x already normalised (this block's GroupNorm, no SiLU) by its producer's reducer; next: see resnet()
  // pre_ss: this block's GroupNorm as a scale | shift table already written by its producer's in-kernel finish (FusedNorm::ss; only where
  // xf_wants_table() said so)
  int xf(const Tensor& x, const XfW& w, Tensor* out, bool out_stats, bool shared = false, const Tensor* pre = nullptr,
         FusedNorm* next = nullptr, const float* pre_ss = nullptr) {
    const int Bpre = Bx, Bfull = shared ? 2 * Bx : Bx;
    Bx = Bfull;                    // every buffer is sized for the full batch
    const int H = x.H, Wd = x.W, C = w.C, HW = H * Wd, M = Bfull * HW, M1 = Bpre * HW;
    const int nh = w.heads, hdp = nh * w.dp;
    const bool ffn_fused = w.w1c != nullptr && ffn_fused_supported(C, M) && HW % 128 == 0;
    // proj_in + norm1 + QKV, and attn1.to_out + residual + norm2 + attn2.to_q, as one kernel each (lnproj.hip).  Not on the shared-prefix
    // block: its M1 = M / 2 rows are 128 tiles for 256 CUs — one tile per CU takes as long as at full size, the two GEMMs take half.
    const bool lnproj = w.wqkv1p != nullptr && !shared && lnproj_supported(C, M, nh, w.dp) && HW % 32 == 0;
    *out = talloc(H, Wd, C, out_stats);
    const size_t mk = m->arena.mark();
    Tensor n = talloc(H, Wd, C);
    Tensor xd = x;                 // the outer residual at full batch
    if (shared) xd = talloc(H, Wd, C);
    Bx = Bpre;
    // the fused projection kernel applies this GroupNorm itself to the rows it loads (lnproj.hip: gn_ss) where the producer filed the
    // statistics: the stand-alone pass (13 us, 21 MB read + 21 MB written per level-0 block) becomes a 640-float table per sample
    float* gn_ss = nullptr;
    bool gn_folded = false;
    if (lnproj && HW % 128 == 0) gn_ss = (float*)m->arena.alloc(sizeof(float) * (size_t)Bx * 2 * C);
    if (pre_ss && gn_ss) { gn_ss = const_cast<float*>(pre_ss); gn_folded = true; }
    else if (pre) n = *pre;
    else GILL_TRY(gnorm(x, nullptr, w.gn, 1e-6f, 0, n, 0.f, gn_ss, &gn_folded));
    Bx = Bfull;
    Tensor t = talloc(H, Wd, C);   // transformer residual stream
    // norm1/2/3 never materialise: the GEMM that writes the residual stream also accumulates each row's sum and sum of
    // squares, and the projection that follows applies mean / rstd in its epilogue on weights pre-multiplied by the LN gain
    RowStats st1 = ln_slot(M, C), st2 = ln_slot(M, C), st3 = ln_slot(M, C);
    if (!lnproj) GILL_TRY(linear(n.p, C, nullptr, 0, C, M1, w.proj_in.w, w.proj_in.b, C, C, nullptr, ACT_NONE, t.p, C, nullptr, &st1));
    const int hw_pad = round_up(HW, 32);   // kv tiles are 32 wide; pad rows hold finite stale data and are masked
    bf16_t* q = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)Bx * nh * hw_pad * w.dp);
    bf16_t* k = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)Bx * nh * hw_pad * w.dp);
    bf16_t* vt = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)Bx * nh * w.dpv * hw_pad);
    bf16_t* o = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * hdp);
    // --- self attention
    LnProjArgs lp;
    lp.M = M; lp.T = t.p; lp.heads = nh; lp.dp = w.dp; lp.dpv = w.dpv; lp.ntok = HW; lp.ntok_pad = hw_pad;
    lp.Cq = q; lp.Ck = k; lp.Cvt = vt; lp.qscale = 1.4426950408889634f / sqrtf((float)w.d);
    if (lnproj) {
      lp.mode = 0; lp.X = gn_folded ? x.p : n.p; lp.gn_ss = gn_folded ? gn_ss : nullptr; lp.W1 = w.proj_in.w; lp.b1 = w.proj_in.b; lp.W2p = w.wqkv1p; lp.c2 = w.c_qkv1;
      if (!dry) GILL_TRY(lnproj_launch(lp, s));
    } else {
      GemmArgs g;
      g.M = M1; g.N = 3 * hdp; g.K = C; g.K1 = C; g.A = t.p; g.lda = C; g.W = w.wqkv1;
      g.ln_stats = st1.p; g.ln_planes = st1.planes; g.ln_colsum = w.s_qkv1; g.bias = w.c_qkv1;
      g.out_mode = OUT_QKV; g.Cq = q; g.Ck = k; g.Cvt = vt; g.heads = nh; g.dp = w.dp; g.dpv = w.dpv;
      g.ntok = HW; g.ntok_pad_q = hw_pad; g.ntok_pad_kv = hw_pad; g.seg_base = 0;
      g.qscale = 1.4426950408889634f / sqrtf((float)w.d);
      GILL_TRY(gemm(g));
    }
    Bx = Bpre;
    GILL_TRY(attend(q, k, vt, o, HW, HW, hw_pad, hw_pad, w));
    Bx = Bfull;
    const bf16_t* tres = t.p;         // the residual stream after the two attention sub-blocks
    bool ffn_pre = false;             // ... or, PRE: before attn2.to_out, which the feed-forward kernel then runs itself
    bool xalg_done = false;
    {
    if (lnproj) {
      GILL_REQUIRE(!w.xg, "internal: the fused projection pairs and the two-GEMM cross-attention are alternatives");
      lp.mode = 1; lp.X = o; lp.gn_ss = nullptr; lp.W1 = w.out1.w; lp.b1 = w.out1.b; lp.W2p = w.wq2p; lp.c2 = w.c_q2;
      if (!dry) GILL_TRY(lnproj_launch(lp, s));
    } else {
    GILL_TRY(linear(o, hdp, nullptr, 0, hdp, M1, w.out1.w, w.out1.b, C, hdp, t.p, ACT_NONE, t.p, C, nullptr, &st2));
    if (shared && !dry) {
      // second half of the pair := first half (residual stream, its LayerNorm row sums, the block input)
      // (out1 ran on M1 rows: its row-sum planes are [planes][M1][2]; the consumer below wraps rows >= M1 onto them)
      GILL_TRY(dup_pair_launch(t.p, x.p, xd.p, sizeof(bf16_t) * (size_t)M1 * C, s));
    }
    // --- cross attention (K/V cached per prompt)
    if (w.xg) {
      // ... as P = softmax80(LN(t) Mq_b^T), t += P Wo_b^T + bias on per-sample weights (xalg_fold_kernel)
      const int n80 = 80 * nh, id = w.layer_id;
      GemmArgs g;
      g.M = M; g.N = n80; g.K = C; g.K1 = C; g.A = t.p; g.lda = C; g.W = m->xq_w[id];
      g.wb_rows = HW; g.wb_stride = (int64_t)n80 * C; g.vb_stride = n80;
      g.ln_stats = st2.p; g.ln_planes = st2.planes; g.ln_rows = M1; g.ln_colsum = m->xq_cs[id]; g.bias = m->xq_b[id];
      g.out_mode = OUT_SOFTMAX80; g.C = o; g.ldc = n80;
      GILL_TRY(gemm(g));
      GemmArgs g2;
      g2.M = M; g2.N = C; g2.K = n80; g2.K1 = n80; g2.A = o; g2.lda = n80; g2.W = m->xo_w[id]; g2.bias = w.out2.b;
      g2.wb_rows = HW; g2.wb_stride = (int64_t)n80 * C;
      g2.resid = t.p; g2.ldr = C; g2.C = t.p; g2.ldc = C;
      GILL_TRY(gemm(g2, &st3));
      xalg_done = true;
    } else {
      GemmArgs g;
      g.M = M; g.N = hdp; g.K = C; g.K1 = C; g.A = t.p; g.lda = C; g.W = w.wq2;
      g.ln_stats = st2.p; g.ln_planes = st2.planes; g.ln_rows = M1; g.ln_colsum = w.s_q2; g.bias = w.c_q2;
      g.out_mode = OUT_QKV; g.Cq = q; g.Ck = k; g.Cvt = vt; g.heads = nh; g.dp = w.dp; g.dpv = w.dpv;
      g.ntok = HW; g.ntok_pad_q = hw_pad; g.ntok_pad_kv = hw_pad; g.seg_base = 0;
      g.qscale = 1.4426950408889634f / sqrtf((float)w.d);
      GILL_TRY(gemm(g));
    }
    }
    if (!xalg_done) {
    GILL_REQUIRE(m->kcache[w.layer_id] != nullptr || dry, "internal: cross-attention K/V cache missing");
    GILL_TRY(attend(q, m->kcache[w.layer_id], m->vcache[w.layer_id], o, HW, m->cfg.ctx_len, hw_pad, m->ctx_pad, w));
    // (with the fused feed-forward kernel's PRE form, attn2.to_out + residual run inside it: ffn.hip)
    ffn_pre = ffn_fused && w.wpp != nullptr;
    GILL_REQUIRE(!ffn_fused || ffn_pre == w.w1c_kperm, "internal: fused feed-forward kernel form does not match the layout its weights were written in");
    if (!ffn_pre) GILL_TRY(linear(o, hdp, nullptr, 0, hdp, M, w.out2.w, w.out2.b, C, hdp, t.p, ACT_NONE, t.p, C, nullptr, &st3));
    }
    }
    if (ffn_fused) {
      // --- GEGLU feed-forward, its residual, proj_out and the outer residual as ONE kernel (ffn.hip)
      if (!dry) {
        FfnArgs fa;
        fa.M = M; fa.T = tres; fa.ln_stats = st3.p; fa.ln_planes = st3.planes;
        fa.W1c = w.w1c; fa.b1c = w.b1c; fa.W2p = w.w2p; fa.Wfo = w.wfo; fa.bo = w.bfo; fa.resid = xd.p; fa.out = out->p;
        if (ffn_pre) { fa.X = o; fa.Wo = w.out2.w; fa.bo2 = w.out2.b; fa.Wpp = w.wpp; }
        if (out->stats && out->sbin == 5) { fa.gn_stats = out->stats; fa.rows_per_batch = HW; out->nslab = HW / GN_SLAB_ROWS; }
        else out->stats = nullptr;        // (no partials from this producer: the consumer runs its own statistics pass)
        GILL_TRY(ffn_fused_launch(fa, s));
      }
      m->arena.release(mk);
      return 0;
    }
    // --- GEGLU feed-forward
    bf16_t* ffh = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * 4 * C);
    unsigned char* t8 = w.wff1_8 ? (unsigned char*)m->arena.alloc((size_t)M * C) : nullptr;
    if (w.wff1_8) {
      // fp8 mode: norm3 applied explicitly into an e4m3 copy of t (one pass: 3 bytes per element), then the GEGLU GEMM on the fp8 matrix instruction
      if (!dry) {
        GILL_TRY(ln_quant_fp8_launch(tres, M, C, st3.p, st3.planes, 0, 1e-5f, F8_LIN_ACT_SCALE, t8, s));
        LinF8Args a;
        a.M = M; a.N = 8 * C; a.K = C; a.A8 = t8; a.W8 = w.wff1_8; a.colscale = w.cs_ff1; a.bias = w.bff1; a.C = ffh;
        GILL_TRY(geglu_fp8_launch(a, s));
        GILL_TRY(dbg_sync("geglu fp8", M, 8 * C, C));
      }
    } else {
      GemmArgs g;
      g.M = M; g.N = 8 * C; g.K = C; g.K1 = C; g.A = tres; g.lda = C; g.W = w.wff1; g.bias = w.bff1;
      g.ln_stats = st3.p; g.ln_planes = st3.planes; g.ln_colsum = w.s_ff1;
      g.act = ACT_GEGLU; g.C = ffh; g.ldc = 4 * C;
      GILL_TRY(gemm(g));
    }
    // --- feed-forward output, its residual, proj_out and the outer residual: one GEMM over K = [h | t] (see ffo_fuse_kernel; the
    // reference graph's two GEMMs measured 604.3 -> 588.9 ms against it, profiles/HISTORY.md)
    GILL_TRY(linear(ffh, 4 * C, tres, C, 4 * C, M, w.wfo, w.bfo, C, 5 * C, xd.p, ACT_NONE, out->p, C, out, nullptr, next));
    m->arena.release(mk);
    return 0;
  }

  // sample: (Bx, in_ch, L, L) fp32 NCHW -> eps (Bx, out_ch, L, L) fp32 NCHW
  int forward(const float* sample, float* eps_out) {
    const gill_unet_config& c = m->cfg;
    const int* ch = c.block_out_channels;
    const int L = c.sample_size;
    m->arena.off = 0;
    m->gn_next = 0;
    m->ln_next = 0;
    m->coop_next = 0;
    // (the statistics pools need no zeroing: every partial sum is written exactly once by its producer; the COOP arrival counters are zeroed
    // by the forward's first kernel, below)
    std::vector<Tensor> skips;
    Tensor x = talloc(L, L, ch[0], true);
    {
      // conv_in: im2col (K = 9*Cin padded to 64) + MFMA GEMM
      const size_t mk = m->arena.mark();
      bf16_t* col = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)Bx * L * L * 64);
      if (!dry) GILL_TRY(im2col_nchw_launch(sample, Bx, c.in_channels, L, L, 64, col, s, m->coop_ctr, (int)m->coop_n));
      GILL_TRY(linear(col, 64, nullptr, 0, 64, Bx * L * L, m->conv_in_w, m->conv_in_b, ch[0], 64, nullptr, ACT_NONE, x.p, ch[0], &x));
      m->arena.release(mk);
    }
    skips.push_back(x);
    // Where a block's last GEMM is a split-K launch (levels 2-3 at the 8-sample batch) and the next consumer's first op is a
    // single-source GroupNorm, that norm is offered to the producer's reducer (FusedNorm): `pn` carries the normalised copy forward.
    // The copy is allocated whether or not the offer is taken (same arena layout in the dry run and in every real run).
    // Round 6: the same offer goes to non-split producers too — the 3x3 convolutions of levels 0-1 finish the norm in their own epilogue
    // (gemm.hip "COOP": the workgroups of a sample wait for each other's partials) — as a normalised copy, or (table) as the scale | shift table
    // the level-0 projection kernel applies on load.
    FusedNorm pn{nullptr, 0.f, 0, Tensor(), true, false};
    auto offer = [&](const NormW& n, float eps, int silu, int H, int W, int C, bool table = false) -> FusedNorm* {
      Tensor y; y.H = H; y.W = W; y.C = C;
      if (!table) y = talloc(H, W, C);
      pn = FusedNorm{&n, eps, silu, y, true, false};
      if (table) pn.ss = (float*)m->arena.alloc(sizeof(float) * (size_t)Bx * 2 * C);
      return &pn;
    };
    auto taken = [&]() -> const Tensor* { return (pn.done && !pn.ss) ? &pn.y : nullptr; };
    auto taken_ss = [&]() -> const float* { return (pn.done && pn.ss) ? pn.ss : nullptr; };
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 2; ++j) {
        Tensor y;
        // CFG pair: the first resnet and the first transformer block up to its self-attention see identical inputs in
        // both halves of the batch -> run them on the first half only (xf() widens back to the full batch)
        const bool share = cfg_pair && i == 0 && j == 0 && Bx % 2 == 0 && temb_bstride == 0;
        if (share) Bx /= 2;
        const Tensor* pre = taken();
        Tensor pre_t; if (pre) { pre_t = *pre; pre = &pre_t; }
        pn.done = false;
        // this resnet's consumer: the transformer block's GroupNorm (i < 3), else the next resnet's / the mid block's norm1
        FusedNorm* nx = nullptr;
        if (i < 3) nx = offer(m->down_xf[i][j].gn, 1e-6f, 0, x.H, x.W, ch[i], xf_wants_table(m->down_xf[i][j], x.H * x.W, share));
        else nx = offer(j == 0 ? m->down_res[3][1].n1 : m->mid_res[0].n1, 1e-5f, 1, x.H, x.W, ch[i]);
        GILL_TRY(resnet(x, nullptr, m->down_res[i][j], &y, true, pre, nx));
        x = y;
        // next consumer: resnet norm1 (j == 0) / the downsample conv or the mid block's norm1 (j == 1)
        if (i < 3) {
          const Tensor* pre2 = taken();
          Tensor pre2_t; if (pre2) { pre2_t = *pre2; pre2 = &pre2_t; }
          const float* pre2_ss = taken_ss();
          pn.done = false;
          FusedNorm* nx2 = (i >= 2 && j == 0) ? offer(m->down_res[i][1].n1, 1e-5f, 1, x.H, x.W, ch[i]) : nullptr;
          Tensor z; GILL_TRY(xf(x, m->down_xf[i][j], &z, true, share, pre2, nx2, pre2_ss)); x = z;   // next GroupNorm and / or a skip
        }
        skips.push_back(x);
      }
      if (i < 3) {
        Tensor y = talloc(x.H / 2, x.W / 2, ch[i], true);
        pn.done = false;
        FusedNorm* nx = offer(m->down_res[i + 1][0].n1, 1e-5f, 1, y.H, y.W, ch[i]);
        GILL_TRY(conv(x, nullptr, m->down_ds[i], 2, 0, nullptr, 0, nullptr, y, nx));
        x = y;
        skips.push_back(x);
      }
    }
    {
      const Tensor* pre = taken();
      Tensor pre_t; if (pre) { pre_t = *pre; pre = &pre_t; }
      pn.done = false;
      FusedNorm* nx = offer(m->mid_xf.gn, 1e-6f, 0, x.H, x.W, ch[3]);
      Tensor y; GILL_TRY(resnet(x, nullptr, m->mid_res[0], &y, true, pre, nx)); x = y;
      const Tensor* pre2 = taken();
      Tensor pre2_t; if (pre2) { pre2_t = *pre2; pre2 = &pre2_t; }
      pn.done = false;
      FusedNorm* nx2 = offer(m->mid_res[1].n1, 1e-5f, 1, x.H, x.W, ch[3]);
      Tensor z; GILL_TRY(xf(x, m->mid_xf, &z, true, false, pre2, nx2)); x = z;
      const Tensor* pre3 = taken();
      Tensor pre3_t; if (pre3) { pre3_t = *pre3; pre3 = &pre3_t; }
      pn.done = false;
      Tensor u; GILL_TRY(resnet(x, nullptr, m->mid_res[1], &u, true, pre3, nullptr)); x = u;   // -> two-source norm1 of up block 0
    }
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 3; ++j) {
        Tensor skip = skips.back(); skips.pop_back();
        Tensor y;
        pn.done = false;
        // (norm1 of an up-block resnet is two-source: never offered; its conv2 feeds the transformer block's GroupNorm)
        FusedNorm* nx = (i >= 1) ? offer(m->up_xf[i][j].gn, 1e-6f, 0, x.H, x.W, ch[3 - i], xf_wants_table(m->up_xf[i][j], x.H * x.W, false)) : nullptr;
        GILL_TRY(resnet(x, &skip, m->up_res[i][j], &y, true, nullptr, nx));
        x = y;
        if (i > 0) {
          const Tensor* pre = taken();
          Tensor pre_t; if (pre) { pre_t = *pre; pre = &pre_t; }
          const float* pre_ss = taken_ss();
          pn.done = false;
          Tensor z; GILL_TRY(xf(x, m->up_xf[i][j], &z, true, false, pre, nullptr, pre_ss)); x = z;
        }
      }
      if (i < 3) {
        // (4-tap form: the epilogue's GroupNorm slabs are 64 SOURCE rows of one parity class — tiny grids leave the sums to the consumer)
        Tensor y = talloc(x.H * 2, x.W * 2, x.C, !m->up_us[i].ups4 || (x.H * x.W) % GN_SLAB_ROWS == 0);
        GILL_TRY(conv(x, nullptr, m->up_us[i], 1, 1, nullptr, 0, nullptr, y));
        x = y;
      }
    }
    Tensor n = talloc(L, L, ch[0]);
    GILL_TRY(gnorm(x, nullptr, m->norm_out, 1e-5f, 1, n));
    GILL_TRY(dbg_sync("last groupnorm"));
    if (!dry) GILL_TRY(conv_out_launch(n.p, m->conv_out_w, m->conv_out_b, Bx, ch[0], L, L, c.out_channels, eps_out, s));
    GILL_TRY(dbg_sync("conv_out"));
    // (a real run must never need more than the dry run of unet_plan_and_alloc() counted)
    GILL_REQUIRE(dry || m->arena.high <= m->arena.cap, "internal: activation arena overflow (the dry run and the real run allocate differently)");
    return 0;
  }
};

}  // namespace

static int unet_plan_and_alloc(gill_unet* m) {
  const gill_unet_config& c = m->cfg;
  const int Bx = c.max_batch;
  const int L = c.sample_size;
  const size_t n_lat = (size_t)c.in_channels * L * L;
  m->ctx_pad = round_up(c.ctx_len, 64);     // whole 64-key tiles: the LDS-DMA attention kernel streams them unclamped (attention.hip)
  // dry run to size the activation arena
  m->arena.dry = true; m->arena.off = 0; m->arena.high = 0;
  m->kcache.assign(m->n_xf, nullptr); m->vcache.assign(m->n_xf, nullptr);
  m->xq_w.assign(m->n_xf, nullptr); m->xo_w.assign(m->n_xf, nullptr); m->xq_cs.assign(m->n_xf, nullptr); m->xq_b.assign(m->n_xf, nullptr);
  UNetRun r{m, nullptr, Bx, nullptr, 0, true};
  GILL_TRY(r.forward(nullptr, nullptr));
  if (Bx % 2 == 0) {               // the CFG shared-prefix path allocates differently: size for the larger of the two
    const size_t gn1 = m->gn_next; const size_t ln1 = m->ln_next; const size_t co1 = m->coop_next;
    r.cfg_pair = true;
    GILL_TRY(r.forward(nullptr, nullptr));   // (arena.high is a running maximum)
    if (gn1 > m->gn_next) m->gn_next = gn1;
    if (ln1 > m->ln_next) m->ln_next = ln1;
    if (co1 > m->coop_next) m->coop_next = co1;
  }
  // (unet_ctx_cache stages the XALG layers' T = ctx [G | G2]^T here: at most ctx_len x 2 x 8 x 1280 bf16 per sample)
  const size_t t_stage = sizeof(bf16_t) * (size_t)Bx * c.ctx_len * 2 * 8 * c.block_out_channels[3];
  const size_t need = (m->arena.high > t_stage ? m->arena.high : t_stage) + (1 << 20);
  GILL_TRY(m->pool.alloc(&m->arena_mem, need, true));
  m->arena.base = m->arena_mem; m->arena.cap = need; m->arena.dry = false; m->arena.off = 0;
  m->gn_floats = m->gn_next + 64;           // counted by the dry run
  GILL_TRY(m->pool.alloc(&m->gn_stats, m->gn_floats));
  m->ln_floats = m->ln_next + 64;           // counted by the dry run (max batch)
  GILL_TRY(m->pool.alloc(&m->ln_stats, m->ln_floats));
  m->coop_n = m->coop_next + 64;            // COOP arrival counters: counted by the dry run
  GILL_TRY(m->pool.alloc(&m->coop_ctr, m->coop_n));
  m->splitk_ws_floats = (size_t)48 << 20;   // 192 MiB of fp32 partials
  GILL_TRY(m->pool.alloc(&m->splitk_ws, m->splitk_ws_floats, false));
  // cross-attention K/V caches
  auto alloc_cache = [&](const XfW& w) -> int {
    if (w.xg) {
      const size_t n80 = (size_t)80 * w.heads;
      GILL_TRY(m->pool.alloc(&m->xq_w[w.layer_id], (size_t)Bx * n80 * w.C, false));
      GILL_TRY(m->pool.alloc(&m->xo_w[w.layer_id], (size_t)Bx * n80 * w.C, false));
      GILL_TRY(m->pool.alloc(&m->xq_cs[w.layer_id], (size_t)Bx * n80));
      GILL_TRY(m->pool.alloc(&m->xq_b[w.layer_id], (size_t)Bx * n80));
      // (unet_ctx_cache stages T = ctx [G | G2]^T in the activation arena, which is idle between forwards)
      GILL_REQUIRE(m->arena.cap >= sizeof(bf16_t) * (size_t)Bx * c.ctx_len * 2 * w.heads * w.C, "internal: arena smaller than the cross-attention staging tensor");
      return 0;
    }
    GILL_TRY(m->pool.alloc(&m->kcache[w.layer_id], (size_t)Bx * w.heads * m->ctx_pad * w.dp, true));
    GILL_TRY(m->pool.alloc(&m->vcache[w.layer_id], (size_t)Bx * w.heads * w.dpv * m->ctx_pad, true));
    return 0;
  };
  for (int i = 0; i < 3; ++i) for (const XfW& w : m->down_xf[i]) GILL_TRY(alloc_cache(w));
  GILL_TRY(alloc_cache(m->mid_xf));
  for (int i = 1; i < 4; ++i) for (const XfW& w : m->up_xf[i]) GILL_TRY(alloc_cache(w));
  // time embedding scratch: up to 1024 rows (timesteps of a schedule, or per-sample timesteps)
  m->temb_rows_cap = 1024 > Bx ? 1024 : Bx;
  GILL_TRY(m->pool.alloc(&m->t_dev, (size_t)m->temb_rows_cap));
  GILL_TRY(m->pool.alloc(&m->t_sin, (size_t)m->temb_rows_cap * c.block_out_channels[0]));
  GILL_TRY(m->pool.alloc(&m->t_h1, (size_t)m->temb_rows_cap * m->temb_dim));
  GILL_TRY(m->pool.alloc(&m->t_h2, (size_t)m->temb_rows_cap * m->temb_dim));
  GILL_TRY(m->pool.alloc(&m->temb_table, (size_t)m->temb_rows_cap * m->temb_total));
  // loop state
  GILL_TRY(m->pool.alloc(&m->lat, (size_t)Bx * n_lat));
  GILL_TRY(m->pool.alloc(&m->lat2, (size_t)Bx * n_lat));
  GILL_TRY(m->pool.alloc(&m->eps, (size_t)Bx * n_lat));
  GILL_TRY(m->pool.alloc(&m->cur_sample, (size_t)Bx * n_lat));
  GILL_TRY(m->pool.alloc(&m->ets, (size_t)4 * Bx * n_lat));
  GILL_TRY(m->pool.alloc(&m->ctx_full, (size_t)Bx * c.ctx_len * c.cross_attention_dim));
  GILL_TRY(m->pool.alloc(&m->temb_cur, (size_t)m->temb_total));
  GILL_TRY(m->pool.alloc(&m->plms_rows, (size_t)m->temb_rows_cap));
  GILL_TRY(m->pool.alloc(&m->step_ctr, (size_t)2));
  GILL_TRY(m->pool.alloc(&m->guidance_dev, (size_t)4));
  { const char* e = getenv("GILL_NO_GRAPH"); m->use_graph = !(e && e[0] == '1'); }
  if (getenv("GILL_DEBUG_SYNC")) m->use_graph = false;   // dbg_sync() synchronises the stream after every launch: illegal inside a capture
  return 0;
}

// time-embedding MLP + all resnet time projections for `rows` timesteps -> m->temb_table [rows][temb_total]
static int unet_time_table(gill_unet* m, const float* t_host, int rows, hipStream_t s) {
  GILL_REQUIRE(rows >= 1 && rows <= m->temb_rows_cap, "too many timesteps for the time-embedding scratch");
  const int c0 = m->cfg.block_out_channels[0], td = m->temb_dim;
  GILL_CHECK_HIP(hipMemcpyAsync(m->t_dev, t_host, sizeof(float) * rows, hipMemcpyHostToDevice, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));   // t_host may be a transient host buffer
  GILL_TRY(timestep_embed_launch(m->t_dev, rows, c0, m->t_sin, s));
  GemmArgs g;
  g.M = rows; g.N = td; g.K = c0; g.K1 = c0; g.A = m->t_sin; g.lda = c0; g.W = m->te1.w; g.bias = m->te1.b;
  g.act = ACT_SILU; g.C = m->t_h1; g.ldc = td;
  GILL_TRY(gemm_launch(g, s));
  GemmArgs g2;
  g2.M = rows; g2.N = td; g2.K = td; g2.K1 = td; g2.A = m->t_h1; g2.lda = td; g2.W = m->te2.w; g2.bias = m->te2.b;
  g2.act = ACT_SILU;   // every resnet applies SiLU to emb before its time_emb_proj
  g2.C = m->t_h2; g2.ldc = td;
  GILL_TRY(gemm_launch(g2, s));
  GemmArgs g3;
  g3.M = rows; g3.N = m->temb_total; g3.K = td; g3.K1 = td; g3.A = m->t_h2; g3.lda = td; g3.W = m->temb_proj_w;
  g3.bias = m->temb_proj_b; g3.out_mode = OUT_F32; g3.C = m->temb_table; g3.ldc = m->temb_total;
  return gemm_launch(g3, s);
}

// cross-attention K/V of every transformer layer for ctx (Bx,77,ctx_dim)
static int unet_ctx_cache(gill_unet* m, const bf16_t* ctx, int Bx, hipStream_t s) {
  const gill_unet_config& c = m->cfg;
  auto one = [&](const XfW& w) -> int {
    if (w.xg) {
      const int H = w.heads, C = w.C, E = c.cross_attention_dim, id = w.layer_id;
      bf16_t* T = (bf16_t*)m->arena.base;
      GemmArgs g;
      g.M = Bx * c.ctx_len; g.N = 2 * H * C; g.K = E; g.K1 = E; g.A = ctx; g.lda = E; g.W = w.xg; g.C = T; g.ldc = g.N;
      GILL_TRY(gemm_launch(g, s));
      hipLaunchKernelGGL(xalg_scores_operand_kernel, dim3(Bx * H * 80), dim3(256), 0, s, T, ctx, w.xgb, H, C, E, c.ctx_len, m->xq_w[id], m->xq_cs[id],
                         m->xq_b[id]);
      GILL_CHECK_HIP(hipGetLastError());
      hipLaunchKernelGGL(xalg_values_operand_kernel, dim3(Bx * H * (C / 64)), dim3(256), 0, s, T, H, C, c.ctx_len, m->xo_w[id]);
      GILL_CHECK_HIP(hipGetLastError());
      return 0;
    }
    GemmArgs g;
    g.M = Bx * c.ctx_len; g.N = 2 * w.heads * w.dp; g.K = c.cross_attention_dim; g.K1 = g.K;
    g.A = ctx; g.lda = c.cross_attention_dim; g.W = w.wkv2;
    g.out_mode = OUT_QKV; g.Ck = m->kcache[w.layer_id]; g.Cvt = m->vcache[w.layer_id];
    g.heads = w.heads; g.dp = w.dp; g.dpv = w.dpv; g.ntok = c.ctx_len; g.ntok_pad_q = m->ctx_pad;
    g.ntok_pad_kv = m->ctx_pad; g.seg_base = 1;
    return gemm_launch(g, s);
  };
  for (int i = 0; i < 3; ++i) for (const XfW& w : m->down_xf[i]) GILL_TRY(one(w));
  GILL_TRY(one(m->mid_xf));
  for (int i = 1; i < 4; ++i) for (const XfW& w : m->up_xf[i]) GILL_TRY(one(w));
  return 0;
}

// The in-kernel GroupNorm finishes (gemm.hip "COOP") wait for each other inside a launch: legal only while this handle's stream has the device's CUs
// to itself.  A bounded wait that ran out NaN-poisons its outputs and counts itself on the device; every entry point looks at that count first
// (its own time-table upload synchronises with the stream anyway) and refuses to go on silently.
static int unet_coop_check() {
  unsigned n = 0;
  GILL_TRY(gemm_coop_giveups(&n));
  if (n) {
    gill_set_error("an earlier call's in-kernel GroupNorm finish timed out " + std::to_string(n) + " time(s) and NaN-poisoned its outputs: the GPU is shared with "
                   "another stream or process that also runs waiting workgroups.  Give the handle the device to itself, or set GILL_GEMM_COOP=0");
    return -5;
  }
  return 0;
}
extern "C" int gill_coop_timeouts(void) {
  unsigned n = 0;
  if (gemm_coop_giveups(&n) != 0) return -1;
  return (int)n;
}

extern "C" int gill_unet_forward(gill_unet* m, const float* sample, const float* timesteps_host, const void* ctx_bf16,
                                 int Bx, float* eps_out, void* stream) {
  GILL_REQUIRE(m && sample && timesteps_host && ctx_bf16 && eps_out, "null argument");
  GILL_TRY(unet_coop_check());
  GILL_REQUIRE(Bx >= 1 && Bx <= m->cfg.max_batch, "batch exceeds the UNet handle's max_batch");
  hipStream_t s = (hipStream_t)stream;
  GILL_TRY(unet_time_table(m, timesteps_host, Bx, s));
  GILL_TRY(unet_ctx_cache(m, (const bf16_t*)ctx_bf16, Bx, s));
  UNetRun r{m, s, Bx, m->temb_table, m->temb_total, false};
  return r.forward(sample, eps_out);
}

// ------------------------------------------------------------------------------------------------------------------
// PNDM (PLMS, skip_prk_steps=True, steps_offset=1, scaled_linear betas 0.00085..0.012 over 1000 train steps).
static void pndm_alphas_cumprod(std::vector<float>& ac) {
  const int T = 1000;
  ac.resize(T);
  const float a = sqrtf(0.00085f), b = sqrtf(0.012f);
  const float step = (b - a) / (float)(T - 1);
  float prod = 1.f;
  for (int i = 0; i < T; ++i) {
    // torch.linspace (fp32): symmetric evaluation around the midpoint
    const float v = (i < T / 2) ? a + step * (float)i : b - step * (float)(T - 1 - i);
    const float beta = v * v;
    prod *= (1.f - beta);
    ac[i] = prod;
  }
}
static void pndm_timesteps(int num_steps, std::vector<int>& ts, int* ratio_out) {
  const int ratio = 1000 / num_steps;
  std::vector<int> base(num_steps);
  for (int i = 0; i < num_steps; ++i) base[i] = i * ratio + 1;   // steps_offset = 1
  // plms_timesteps = concat(base[:-1], base[-2:-1], base[-1:])[::-1]
  std::vector<int> seq(base.begin(), base.end() - 1);
  if (num_steps >= 2) seq.push_back(base[num_steps - 2]);
  seq.push_back(base[num_steps - 1]);
  ts.assign(seq.rbegin(), seq.rend());
  *ratio_out = ratio;
}

extern "C" int gill_pndm_schedule(int num_steps, int32_t* timesteps_out, double* alphas_cumprod_out) {
  GILL_REQUIRE(num_steps >= 2 && num_steps <= 1000, "num_steps out of range");
  std::vector<int> ts; int ratio;
  pndm_timesteps(num_steps, ts, &ratio);
  if (timesteps_out) for (size_t i = 0; i < ts.size(); ++i) timesteps_out[i] = ts[i];
  if (alphas_cumprod_out) {
    std::vector<float> ac; pndm_alphas_cumprod(ac);
    for (int i = 0; i < 1000; ++i) alphas_cumprod_out[i] = (double)ac[i];
  }
  return (int)ts.size();
}

static int sd_denoise_on(gill_unet* m, const void* cond_bf16, const void* uncond_bf16, int n_uncond, const float* latents0, int B,
                         int num_steps, float guidance, float* latents_out, hipStream_t s);

extern "C" int gill_sd_denoise(gill_unet* m, const void* cond_bf16, const void* uncond_bf16, int n_uncond, const float* latents0,
                               int B, int num_steps, float guidance, float* latents_out, void* stream) {
  GILL_REQUIRE(m && cond_bf16 && latents0 && latents_out, "null argument");
  GILL_REQUIRE(guidance <= 1.0f || uncond_bf16 == nullptr || n_uncond == 1 || n_uncond == B,
               "negative embeddings: batch must be 1 or B");
  GILL_TRY(unet_coop_check());
  hipStream_t caller = (hipStream_t)stream;
  GILL_TRY(m->fence.enter(caller));
  const int rc = sd_denoise_on(m, cond_bf16, uncond_bf16, n_uncond, latents0, B, num_steps, guidance, latents_out, m->fence.stream);
  GILL_TRY(m->fence.leave(caller));
  return rc;
}

static int sd_denoise_on(gill_unet* m, const void* cond_bf16, const void* uncond_bf16, int n_uncond, const float* latents0, int B,
                         int num_steps, float guidance, float* latents_out, hipStream_t s) {
  GILL_REQUIRE(num_steps >= 2 && num_steps <= 1000, "num_steps out of range");
  const bool cfg = guidance > 1.0f;     // do_classifier_free_guidance (custom_sd.py:588)
  const int Bx = cfg ? 2 * B : B;
  GILL_REQUIRE(B >= 1 && Bx <= m->cfg.max_batch, "batch exceeds the UNet handle's max_batch");
  GILL_REQUIRE(!cfg || uncond_bf16 != nullptr, "uncond embedding required when guidance > 1");
  const gill_unet_config& c = m->cfg;
  const int L = c.sample_size;
  const int64_t n_lat = (int64_t)c.in_channels * L * L;
  const size_t ctx_elems = (size_t)c.ctx_len * c.cross_attention_dim;

  std::vector<int> ts; int ratio;
  pndm_timesteps(num_steps, ts, &ratio);
  std::vector<float> ac; pndm_alphas_cumprod(ac);
  const int ncalls = (int)ts.size();
  GILL_REQUIRE(ncalls <= m->temb_rows_cap, "too many steps for the time-embedding scratch");

  // the PLMS schedule of every call (host arithmetic in double, like the scheduler's numpy/torch-CPU tables)
  std::vector<PlmsRow> rows(ncalls);
  {
    int counter = 0, n_ets = 0, last = -1;
    for (int i = 0; i < ncalls; ++i) {
      int t = ts[i];
      int prev_t = t - ratio;
      PlmsRow& a = rows[i];
      a.slot_new = -1; a.s1 = a.s2 = a.s3 = 0;
      if (counter != 1) {
        a.slot_new = (last + 1) & 3;
        a.s1 = last & 3; a.s2 = (last + 3) & 3; a.s3 = (last + 2) & 3;
        last = a.slot_new;
        if (n_ets < 4) ++n_ets;
      } else {
        prev_t = t; t = t + ratio;
        a.s1 = last & 3;
      }
      if (n_ets == 1 && counter == 0) a.mode = 0;
      else if (n_ets == 1 && counter == 1) a.mode = 1;
      else if (n_ets == 2) a.mode = 2;
      else if (n_ets == 3) a.mode = 3;
      else a.mode = 4;
      // _get_prev_sample
      const double at = ac[t];
      const double ap = prev_t >= 0 ? (double)ac[prev_t] : (double)ac[0];   // set_alpha_to_one = False
      const double bt = 1.0 - at, bp = 1.0 - ap;
      const double sample_coeff = sqrt(ap / at);
      const double denom = at * sqrt(bp) + sqrt(at * bt * ap);
      double sc = sample_coeff, ec = (ap - at) / denom;
      if (c.v_prediction) {   // the model output is v: eps' = sqrt(a_t) v + sqrt(1 - a_t) sample, folded into the two coefficients
        sc -= ec * sqrt(bt);
        ec *= sqrt(at);
      }
      a.sample_coeff = (float)sc;
      a.eps_coeff = (float)ec;
      ++counter;
    }
  }
  GILL_CHECK_HIP(hipMemcpyAsync(m->plms_rows, rows.data(), sizeof(PlmsRow) * ncalls, hipMemcpyHostToDevice, s));
  GILL_CHECK_HIP(hipMemsetAsync(m->step_ctr, 0, sizeof(int) * 2, s));
  GILL_CHECK_HIP(hipMemcpyAsync(m->guidance_dev, &guidance, sizeof(float), hipMemcpyHostToDevice, s));   // (synchronised below)
  // hoisted: time-embedding table for every call (its stream sync also covers the host `rows` buffer), prompt K/V caches
  std::vector<float> tf(ncalls);
  for (int i = 0; i < ncalls; ++i) tf[i] = (float)ts[i];
  GILL_TRY(unet_time_table(m, tf.data(), ncalls, s));
  // prompt_embeds = cat([negative_prompt_embeds.repeat(B), prompt_embeds])  (custom_sd.py:365-371)
  if (cfg) {
    for (int b = 0; b < B; ++b)
      GILL_CHECK_HIP(hipMemcpyAsync(m->ctx_full + (size_t)b * ctx_elems,
                                    (const bf16_t*)uncond_bf16 + (n_uncond == B ? (size_t)b * ctx_elems : 0),
                                    sizeof(bf16_t) * ctx_elems, hipMemcpyDeviceToDevice, s));
    GILL_CHECK_HIP(hipMemcpyAsync(m->ctx_full + (size_t)B * ctx_elems, cond_bf16, sizeof(bf16_t) * ctx_elems * B,
                                  hipMemcpyDeviceToDevice, s));
  } else {
    GILL_CHECK_HIP(hipMemcpyAsync(m->ctx_full, cond_bf16, sizeof(bf16_t) * ctx_elems * B, hipMemcpyDeviceToDevice, s));
  }
  GILL_TRY(unet_ctx_cache(m, m->ctx_full, Bx, s));
  GILL_CHECK_HIP(hipMemcpyAsync(m->lat, latents0, sizeof(float) * n_lat * B, hipMemcpyDeviceToDevice, s));

  // One loop step = stage kernel (latents -> UNet input, time-embedding row of the device-side step counter) + UNet forward
  // (~390 launches at ~10+ us of host time each: at small batch the GPU outruns the host) + CFG/PLMS kernel (reads its
  // coefficients from the device-side table, bumps the counter).  The step is captured ONCE per batch size into a hipGraph
  // and replayed ncalls times back to back: nothing but graph launches sits between two steps.
  SdLoopArgs la;
  la.rows = m->plms_rows; la.ctr = m->step_ctr; la.temb_table = m->temb_table; la.temb_total = m->temb_total;
  la.temb_cur = m->temb_cur; la.eps = m->eps; la.lat = m->lat; la.lat2 = m->lat2; la.cur_sample = m->cur_sample; la.ets = m->ets;
  la.B = B; la.n = n_lat; la.guidance = m->guidance_dev; la.cfg = cfg ? 1 : 0;
  // the graph bakes in B and the CFG flag besides the buffer addresses
  const GraphKey gkey{Bx, cfg ? 1 : 0};
  auto one_step = [&](hipStream_t st) -> int {
    GILL_TRY(sd_stage_launch(la, st));
    UNetRun r{m, st, Bx, m->temb_cur, 0, false};
    r.cfg_pair = cfg;
    GILL_TRY(r.forward(m->lat2, m->eps));
    return plms_step_launch(la, st);
  };
  for (int i = 0; i < ncalls; ++i) {
    auto git = m->graphs.find(gkey);
    if (git == m->graphs.end() && m->use_graph && m->warmed.count(gkey)) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      // s is the handle's private stream (never the legacy NULL stream, which cannot be captured)
      GILL_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      const int rc_status = one_step(s);
      const hipError_t ec = hipStreamEndCapture(s, &graph);
      if (rc_status != 0) { if (graph) (void)hipGraphDestroy(graph); return rc_status; }
      GILL_CHECK_HIP(ec);
      GILL_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      git = m->graphs.emplace(gkey, exec).first;
    }
    if (git != m->graphs.end()) {
      GILL_CHECK_HIP(hipGraphLaunch(git->second, s));
    } else {
      // first step of this batch size runs eagerly: it also performs every one-time kernel attribute set-up,
      // which must not happen inside a stream capture
      GILL_TRY(one_step(s));
      m->warmed.insert(gkey);
    }
  }
  GILL_CHECK_HIP(hipMemcpyAsync(latents_out, m->lat, sizeof(float) * n_lat * B, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Operator-level entry for the fused feed-forward block (ffn.hip) on NATURAL operands (diffusers parameter layouts): folds norm3 into the
// GEGLU projection, builds [Wp.W2 | Wp] and the kernel's weight layouts exactly as the engine's loader does, forms the LayerNorm row sums
// of t, launches the kernel.  out = proj_out(ff2(geglu(ff1(LN(t)))) + t) + resid.  For tests/test_ops_gpu.py and tools; synchronises.
__global__ __launch_bounds__(256) void ffn_op_rowsums_kernel(const bf16_t* __restrict__ t, int M, int C, float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float a = 0.f, q = 0.f;
  for (int k = lane; k < C; k += 64) { const float v = bf2f(t[(size_t)row * C + k]); a += v; q += v * v; }
  a = wave_sum(a); q = wave_sum(q);
  if (lane == 0) { stats[(size_t)row * 2] = a; stats[(size_t)row * 2 + 1] = q; }
}
// o2 / Wo / bo2 (optional, all or none): the PRE form — t := t + to_out(o2) first, inside the kernel (o2 [M][320] = the cross-attention
// output, heads x 40; Wo [320][320], bo2 [320]: BasicTransformerBlock.attn2.to_out[0]).
extern "C" int gill_op_ffn_fused(const void* t, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                                 const float* b2, const void* Wp, const float* bp, const void* resid, void* out, float* gn_stats,
                                 int M, int rows_per_batch, const void* o2, const void* Wo, const float* bo2, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int C = 320, inner = 4 * C, heads = 8, d = 40, dp = attn_padded_dim(d), hdp = heads * dp;
  GILL_REQUIRE(t && ln_g && ln_b && W1 && b1 && W2 && b2 && Wp && bp && resid && out, "null argument");
  GILL_REQUIRE((o2 != nullptr) == (Wo != nullptr) && (o2 != nullptr) == (bo2 != nullptr), "o2 / Wo / bo2: all or none");
  GILL_REQUIRE(ffn_fused_supported(C, M), "ffn_fused: M must be a multiple of 128");
  DevBuf idx, wff1, bff1, sff1, wfo, bfo, w1c, b1c, w2p, st, wpp, o2p, wop;
  if (o2) {
    GILL_TRY(wpp.alloc(sizeof(bf16_t) * (size_t)C * C));
    GILL_TRY(o2p.alloc_zero(sizeof(bf16_t) * (size_t)M * hdp, s));
    GILL_TRY(wop.alloc_zero(sizeof(bf16_t) * (size_t)C * hdp, s));
    hipLaunchKernelGGL(pad_head_cols_kernel, dim3(1024), dim3(256), 0, s, o2, 0, M, heads, d, dp, (bf16_t*)o2p.p);
    hipLaunchKernelGGL(pad_head_cols_kernel, dim3(1024), dim3(256), 0, s, Wo, 0, C, heads, d, dp, (bf16_t*)wop.p);
    GILL_CHECK_HIP(hipGetLastError());
  }
  std::vector<int32_t> map = geglu_row_permutation(inner);
  GILL_TRY(idx.alloc(sizeof(int32_t) * map.size()));
  GILL_CHECK_HIP(hipMemcpyAsync(idx.p, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice, s));
  GILL_TRY(wff1.alloc(sizeof(bf16_t) * (size_t)2 * inner * C)); GILL_TRY(bff1.alloc(sizeof(float) * 2 * inner));
  GILL_TRY(sff1.alloc(sizeof(float) * 2 * inner));
  GILL_TRY(scatter_rows_bf16_launch((const bf16_t*)W1, 2 * inner, C, (const int32_t*)idx.p, (bf16_t*)wff1.p, C, s));
  GILL_TRY(permute_f32_launch(b1, (const int32_t*)idx.p, 2 * inner, (float*)bff1.p, s));
  GILL_TRY(ln_fold_rows_launch((bf16_t*)wff1.p, 2 * inner, C, ln_g, ln_b, (float*)sff1.p, (float*)bff1.p, s));
  GILL_TRY(wfo.alloc(sizeof(bf16_t) * (size_t)C * 5 * C)); GILL_TRY(bfo.alloc(sizeof(float) * C));
  hipLaunchKernelGGL(ffo_fuse_kernel, dim3(cdiv(5 * C + 1, 256), C), dim3(256), 0, s, Wp, 0, W2, 0, (const void*)b2, 1, (const void*)bp, 1, C,
                     (bf16_t*)wfo.p, (float*)bfo.p);
  GILL_CHECK_HIP(hipGetLastError());
  GILL_TRY(w1c.alloc(sizeof(bf16_t) * (size_t)8 * C * C)); GILL_TRY(b1c.alloc(sizeof(float) * 8 * C));
  GILL_TRY(w2p.alloc(sizeof(bf16_t) * (size_t)4 * C * C));
  GILL_TRY(ffn_relayout_launch((const bf16_t*)wff1.p, (const float*)bff1.p, (const bf16_t*)wfo.p, (bf16_t*)w1c.p, (float*)b1c.p,
                               (bf16_t*)w2p.p, o2 ? (bf16_t*)wpp.p : nullptr, s));
  GILL_TRY(st.alloc(sizeof(float) * (size_t)M * 2));
  hipLaunchKernelGGL(ffn_op_rowsums_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, (const bf16_t*)t, M, C, (float*)st.p);
  GILL_CHECK_HIP(hipGetLastError());
  FfnArgs fa;
  fa.M = M; fa.T = (const bf16_t*)t; fa.ln_stats = (const float*)st.p; fa.ln_planes = 1;
  fa.W1c = (const bf16_t*)w1c.p; fa.b1c = (const float*)b1c.p; fa.W2p = (const bf16_t*)w2p.p;
  fa.Wfo = (const bf16_t*)wfo.p; fa.bo = (const float*)bfo.p; fa.resid = (const bf16_t*)resid; fa.out = (bf16_t*)out;
  fa.gn_stats = gn_stats; fa.rows_per_batch = rows_per_batch;
  if (o2) { fa.X = (const bf16_t*)o2p.p; fa.Wo = (const bf16_t*)wop.p; fa.bo2 = bo2; fa.Wpp = (const bf16_t*)wpp.p; }
  const int rep = [] { const char* e = getenv("GILL_OP_REPEAT"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  for (int r = 0; r < rep; ++r) GILL_TRY(ffn_fused_launch(fa, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Operator-level entry for the fp8 GEGLU projection (linear_fp8.hip) on NATURAL operands (diffusers parameter layouts): permutes the rows into the
// engine's value / gate interleave, folds norm3 into them, quantises rows and LayerNorm-ed activations to e4m3 exactly as the engine's fp8 mode does,
// launches the kernel.  out [M][inner] = h * gelu(g), [h | g] = LN(t) W^T + b (W [2 inner][C], diffusers order [value rows | gate rows]).  C % 128 == 0.
// For tests/test_fp8_gpu.py and tools; synchronises.
extern "C" int gill_op_geglu_fp8(const void* t, const float* ln_g, const float* ln_b, const void* W, const float* b, void* out, int M, int inner, int C,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GILL_REQUIRE(t && ln_g && ln_b && W && b && out && M > 0 && inner % 16 == 0 && C % 128 == 0, "geglu_fp8: null argument / inner % 16 / C % 128");
  DevBuf idx, wperm, bperm, cs, w8, sc, st, t8;
  std::vector<int32_t> map = geglu_row_permutation(inner);
  GILL_TRY(idx.alloc(sizeof(int32_t) * map.size()));
  GILL_CHECK_HIP(hipMemcpyAsync(idx.p, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice, s));
  GILL_TRY(wperm.alloc(sizeof(bf16_t) * (size_t)2 * inner * C)); GILL_TRY(bperm.alloc(sizeof(float) * 2 * inner)); GILL_TRY(cs.alloc(sizeof(float) * 2 * inner));
  GILL_TRY(scatter_rows_bf16_launch((const bf16_t*)W, 2 * inner, C, (const int32_t*)idx.p, (bf16_t*)wperm.p, C, s));
  GILL_TRY(permute_f32_launch(b, (const int32_t*)idx.p, 2 * inner, (float*)bperm.p, s));
  GILL_TRY(ln_fold_rows_launch((bf16_t*)wperm.p, 2 * inner, C, ln_g, ln_b, (float*)cs.p, (float*)bperm.p, s));
  GILL_TRY(w8.alloc((size_t)2 * inner * C)); GILL_TRY(sc.alloc(sizeof(float) * 2 * inner));
  GILL_TRY(linear_weight_quant_fp8_launch((const bf16_t*)wperm.p, 2 * inner, C, F8_LIN_ACT_SCALE, (unsigned char*)w8.p, (float*)sc.p, s));
  GILL_TRY(st.alloc(sizeof(float) * (size_t)M * 2));
  hipLaunchKernelGGL(ffn_op_rowsums_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, (const bf16_t*)t, M, C, (float*)st.p);
  GILL_CHECK_HIP(hipGetLastError());
  GILL_TRY(t8.alloc((size_t)M * C));
  LinF8Args a;
  a.M = M; a.N = 2 * inner; a.K = C; a.A8 = (const unsigned char*)t8.p; a.W8 = (const unsigned char*)w8.p; a.colscale = (const float*)sc.p;
  a.bias = (const float*)bperm.p; a.C = (bf16_t*)out;
  const int rep = [] { const char* e = getenv("GILL_OP_REPEAT"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  for (int r = 0; r < rep; ++r) {
    GILL_TRY(ln_quant_fp8_launch((const bf16_t*)t, M, C, (const float*)st.p, 1, 0, 1e-5f, F8_LIN_ACT_SCALE, (unsigned char*)t8.p, s));
    GILL_TRY(geglu_fp8_launch(a, s));
  }
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Operator-level entry for the fused projection pairs around norm1 / norm2 of a level-0 block (lnproj.hip) on NATURAL operands
// (unpadded heads, plain LayerNorm parameters): pads / folds / permutes them exactly as the engine's loader does, launches the kernel.
//   mode 0: t = W1 . x + b1;  [q | k | v] = W2 . LN(t)            x [M][320], W2 [3 * 320][320] = to_q | to_k | to_v rows
//   mode 1: t = W1 . x + b1 + t;  q = W2 . LN(t)                  x [M][320] = the attention output (heads x 40), W2 [320][320]
// q, k: [B][8][hw_pad][48] (q scaled by log2(e) / sqrt(40)); vt: [B][8][64][hw_pad] with row 48 = 1.  For tests and tools; synchronises.
extern "C" int gill_op_lnproj(int mode, const void* x, void* t, const void* W1, const float* b1, const float* ln_g, const float* ln_b,
                              const void* W2, void* q, void* k, void* vt, int B, int HW, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int C = 320, heads = 8, d = 40, dp = attn_padded_dim(d), dpv = round_up(dp, 32), hdp = heads * dp, M = B * HW;
  const int nseg = mode == 0 ? 3 : 1;
  GILL_REQUIRE(mode == 0 || mode == 1, "mode must be 0 or 1");
  GILL_REQUIRE(x && t && W1 && b1 && ln_g && ln_b && W2 && q && (mode == 1 || (k && vt)), "null argument");
  GILL_REQUIRE(lnproj_supported(C, M, heads, dp), "lnproj: B * HW must be a multiple of 128");
  DevBuf xp, w1p, w2, w2p, cs, cb;
  GILL_TRY(w2.alloc_zero(sizeof(bf16_t) * (size_t)nseg * hdp * C, s));
  GILL_TRY(w2p.alloc(sizeof(bf16_t) * (size_t)nseg * hdp * C));
  GILL_TRY(cs.alloc_zero(sizeof(float) * (size_t)nseg * hdp, s));
  GILL_TRY(cb.alloc_zero(sizeof(float) * (size_t)nseg * hdp, s));
  for (int sg = 0; sg < nseg; ++sg)
    hipLaunchKernelGGL(pad_head_rows_kernel, dim3(1024), dim3(256), 0, s, (const void*)((const bf16_t*)W2 + (size_t)sg * C * C), 0, heads, d, dp, C,
                       (bf16_t*)w2.p + (size_t)sg * hdp * C);
  GILL_CHECK_HIP(hipGetLastError());
  GILL_TRY(ln_fold_rows_launch((bf16_t*)w2.p, nseg * hdp, C, ln_g, ln_b, (float*)cs.p, (float*)cb.p, s));
  GILL_TRY(lnproj_kperm_launch((const bf16_t*)w2.p, nseg * hdp, (bf16_t*)w2p.p, s));
  LnProjArgs a;
  a.mode = mode; a.M = M; a.T = (bf16_t*)t; a.b1 = b1; a.W2p = (const bf16_t*)w2p.p; a.c2 = (const float*)cb.p;
  a.Cq = (bf16_t*)q; a.Ck = (bf16_t*)k; a.Cvt = (bf16_t*)vt; a.heads = heads; a.dp = dp; a.dpv = dpv; a.ntok = HW; a.ntok_pad = round_up(HW, 32);
  a.qscale = 1.4426950408889634f / sqrtf((float)d);
  if (mode == 0) {
    a.X = (const bf16_t*)x; a.W1 = (const bf16_t*)W1;
  } else {
    GILL_TRY(xp.alloc_zero(sizeof(bf16_t) * (size_t)M * hdp, s));
    GILL_TRY(w1p.alloc_zero(sizeof(bf16_t) * (size_t)C * hdp, s));
    hipLaunchKernelGGL(pad_head_cols_kernel, dim3(1024), dim3(256), 0, s, x, 0, M, heads, d, dp, (bf16_t*)xp.p);
    hipLaunchKernelGGL(pad_head_cols_kernel, dim3(1024), dim3(256), 0, s, W1, 0, C, heads, d, dp, (bf16_t*)w1p.p);
    GILL_CHECK_HIP(hipGetLastError());
    a.X = (const bf16_t*)xp.p; a.W1 = (const bf16_t*)w1p.p;
  }
  const int rep = [] { const char* e = getenv("GILL_OP_REPEAT"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  for (int r = 0; r < rep; ++r) GILL_TRY(lnproj_launch(a, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// Op-level entry of the two-GEMM cross-attention (xalg_fold_kernel) on a torch-layout attn2 (to_q / to_out [C][C], to_k / to_v [C][E], heads
// of d = C / H >= 80 features, norm2's gain and bias): folds the weights as the loader does, builds the per-sample operands from `ctx`
// [B][ctx_len][E] as unet_ctx_cache does, then out = t + softmax(LN(t) Wq^T K^T / sqrt(d)) V Wo^T + bo on t [B * HW][C].  `P` (optional)
// receives the softmax weights [B * HW][80 H] (key slot j of head h at column 80 h + j).  For tests and tools; synchronises.
extern "C" int gill_op_cross_attention_folded(const void* t, const float* ln_g, const float* ln_b, const void* Wq, const void* Wk, const void* Wv,
                                              const void* Wo, const float* bo, const void* ctx, void* out, void* P, int B, int HW, int C, int H,
                                              int ctx_len, int E, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GILL_REQUIRE(t && ln_g && ln_b && Wq && Wk && Wv && Wo && bo && ctx && out, "null argument");
  GILL_REQUIRE(H > 0 && H % 2 == 0 && C % H == 0, "cross_attention_folded: an even number of heads dividing C");
  const int d = C / H, M = B * HW, n80 = 80 * H;
  GILL_REQUIRE(d % 16 == 0 && d >= 80 && d <= 160 && C % 64 == 0 && E % 64 == 0 && ctx_len >= 1 && ctx_len <= 80 && HW % 64 == 0,
               "cross_attention_folded: head dim 80..160 (multiple of 16), C and E multiples of 64, at most 80 context tokens, HW a multiple of 64");
  DevBuf wq, wkv, cs, cq, xg, xgb, T, mq, mcs, mb, wo, st, p;
  GILL_TRY(wq.alloc(sizeof(bf16_t) * (size_t)C * C));
  GILL_TRY(wkv.alloc(sizeof(bf16_t) * (size_t)2 * C * E));
  GILL_CHECK_HIP(hipMemcpyAsync(wq.p, Wq, sizeof(bf16_t) * (size_t)C * C, hipMemcpyDeviceToDevice, s));
  GILL_CHECK_HIP(hipMemcpyAsync(wkv.p, Wk, sizeof(bf16_t) * (size_t)C * E, hipMemcpyDeviceToDevice, s));
  GILL_CHECK_HIP(hipMemcpyAsync((bf16_t*)wkv.p + (size_t)C * E, Wv, sizeof(bf16_t) * (size_t)C * E, hipMemcpyDeviceToDevice, s));
  GILL_TRY(cs.alloc(sizeof(float) * C)); GILL_TRY(cq.alloc_zero(sizeof(float) * C, s));      // (ln_fold_rows ADDS beta . W^T to the bias it is given)
  GILL_TRY(ln_fold_rows_launch((bf16_t*)wq.p, C, C, ln_g, ln_b, (float*)cs.p, (float*)cq.p, s));
  GILL_TRY(xg.alloc(sizeof(bf16_t) * (size_t)2 * H * C * E)); GILL_TRY(xgb.alloc(sizeof(float) * (size_t)H * E));
  const float qs = 1.4426950408889634f / sqrtf((float)d);
  hipLaunchKernelGGL(xalg_fold_kernel, dim3(2 * H * C / 8), dim3(256), 0, s, (const bf16_t*)wq.p, (const bf16_t*)wkv.p, (const bf16_t*)Wo, H, C, d, E, qs,
                     (bf16_t*)xg.p);
  hipLaunchKernelGGL(xalg_fold_bias_kernel, dim3(cdiv(E, 256), H), dim3(256), 0, s, (const float*)cq.p, (const bf16_t*)wkv.p, d, E, qs, (float*)xgb.p);
  GILL_CHECK_HIP(hipGetLastError());
  GILL_TRY(T.alloc(sizeof(bf16_t) * (size_t)B * ctx_len * 2 * H * C));
  GILL_TRY(mq.alloc(sizeof(bf16_t) * (size_t)B * n80 * C)); GILL_TRY(wo.alloc(sizeof(bf16_t) * (size_t)B * n80 * C));
  GILL_TRY(mcs.alloc(sizeof(float) * (size_t)B * n80)); GILL_TRY(mb.alloc(sizeof(float) * (size_t)B * n80));
  {
    GemmArgs g;
    g.M = B * ctx_len; g.N = 2 * H * C; g.K = E; g.K1 = E; g.A = (const bf16_t*)ctx; g.lda = E; g.W = (const bf16_t*)xg.p; g.C = T.p; g.ldc = g.N;
    GILL_TRY(gemm_launch(g, s));
  }
  hipLaunchKernelGGL(xalg_scores_operand_kernel, dim3(B * H * 80), dim3(256), 0, s, (const bf16_t*)T.p, (const bf16_t*)ctx, (const float*)xgb.p, H, C, E,
                     ctx_len, (bf16_t*)mq.p, (float*)mcs.p, (float*)mb.p);
  hipLaunchKernelGGL(xalg_values_operand_kernel, dim3(B * H * (C / 64)), dim3(256), 0, s, (const bf16_t*)T.p, H, C, ctx_len, (bf16_t*)wo.p);
  GILL_CHECK_HIP(hipGetLastError());
  GILL_TRY(st.alloc(sizeof(float) * (size_t)M * 2));
  hipLaunchKernelGGL(ffn_op_rowsums_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, (const bf16_t*)t, M, C, (float*)st.p);
  GILL_CHECK_HIP(hipGetLastError());
  void* pp = P;
  if (!pp) { GILL_TRY(p.alloc(sizeof(bf16_t) * (size_t)M * n80)); pp = p.p; }
  GemmArgs g;
  g.M = M; g.N = n80; g.K = C; g.K1 = C; g.A = (const bf16_t*)t; g.lda = C; g.W = (const bf16_t*)mq.p;
  g.wb_rows = HW; g.wb_stride = (int64_t)n80 * C; g.vb_stride = n80;
  g.ln_stats = (const float*)st.p; g.ln_planes = 1; g.ln_colsum = (const float*)mcs.p; g.bias = (const float*)mb.p;
  g.out_mode = OUT_SOFTMAX80; g.C = pp; g.ldc = n80;
  GemmArgs g2;
  g2.M = M; g2.N = C; g2.K = n80; g2.K1 = n80; g2.A = (const bf16_t*)pp; g2.lda = n80; g2.W = (const bf16_t*)wo.p; g2.bias = bo;
  g2.wb_rows = HW; g2.wb_stride = (int64_t)n80 * C;
  g2.resid = t; g2.ldr = C; g2.C = out; g2.ldc = C;
  const int rep = [] { const char* e = getenv("GILL_OP_REPEAT"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  for (int r = 0; r < rep; ++r) { GILL_TRY(gemm_launch(g, s)); GILL_TRY(gemm_launch(g2, s)); }
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}
