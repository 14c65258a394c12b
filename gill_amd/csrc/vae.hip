// Stage 3b — Stable Diffusion VAE decoder: final latents -> 512x512 RGB.
//
// Replaces StableDiffusionPipeline.decode_latents (gill/custom_sd.py:385-392):
//     latents = 1 / 0.18215 * latents
//     image = self.vae.decode(latents).sample
//     image = (image / 2 + 0.5).clamp(0, 1)
// The decoder itself is diffusers' AutoencoderKL (SD-1.5 vae/config.json: block_out_channels 128/256/512/512,
// layers_per_block 2, GroupNorm(32, eps 1e-6), SiLU, one single-head 512-d attention in the mid block) — not in the
// reference tree; structure restated from the published model, state-dict names are diffusers'
// ("decoder.up_blocks.2.resnets.0.conv_shortcut.weight", "post_quant_conv.weight", ...).
//
// Built from the same kernels as the UNet: NHWC bf16 activations, implicit-GEMM 3x3 convs with the 1x1 shortcut and
// the nearest-2x upsample folded in, GroupNorm statistics accumulated by the producing conv's epilogue.  The single
// 512-wide attention head does not fit the flash kernel's register budget, so it runs as two MFMA GEMMs per image
// (S = Q K^T, O = P V) around a row-softmax kernel: 68 GFLOP per image, once per image — not a hot spot.
#include "engine_util.h"
#include <math.h>

namespace {

struct ConvW { bf16_t* w = nullptr; float* b = nullptr; int cin = 0, cout = 0; int ups4 = 0; };
struct NormW { float* g = nullptr; float* b = nullptr; };
struct VResW {
  NormW n1, n2;
  ConvW c1, c2;
  bool has_sc = false;
  bf16_t* c2f_w = nullptr; float* c2f_b = nullptr;   // conv2 with the 1x1 shortcut fused (see unet.hip)
  int cin = 0, cout = 0;
};
struct VTensor { bf16_t* p = nullptr; int H = 0, W = 0, C = 0; float* stats = nullptr; int nslab = 1; };

struct VArena {
  unsigned char* base = nullptr;
  size_t off = 0, high = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > high) high = off;
    return p;
  }
};

}  // namespace

struct gill_vae {
  gill_vae_config cfg;
  DevPool pool;
  float* pq_w = nullptr; float* pq_b = nullptr;        // post_quant_conv (4x4 + 4), fp32
  bf16_t* conv_in_w = nullptr; float* conv_in_b = nullptr;   // [C][64] (im2col K = 36 padded to 64)
  VResW mid_res[2];
  NormW attn_gn;
  bf16_t* attn_wqkv = nullptr; float* attn_bqkv = nullptr;   // [3C][C]
  bf16_t* attn_wo = nullptr; float* attn_bo = nullptr;
  std::vector<VResW> up_res[4];
  ConvW up_us[3];
  NormW norm_out;
  bf16_t* conv_out_w = nullptr; float* conv_out_b = nullptr;
  // workspace
  VArena arena;
  unsigned char* arena_mem = nullptr;
  float* gn_stats = nullptr; size_t gn_floats = 0, gn_next = 0;   // per-decode pool of GroupNorm partial-sum slots
  float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
  float* lat_prep = nullptr;   // [B][4][L][L] after scaling + post_quant_conv
  float* img_f32 = nullptr;    // [B][3][8L][8L]
};

__global__ void vae_vec_add_kernel(const float* a, const float* b, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// z' = post_quant_conv(z / scaling_factor): per-pixel CxC matrix on NCHW fp32 (C = 4)
__global__ __launch_bounds__(256) void vae_latent_prep_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                              const float* __restrict__ b, float inv_scale, int C, int HW,
                                                              int64_t total, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int64_t bb = i / ((int64_t)HW * C);
    float acc = b[c];
    for (int k = 0; k < C; ++k) acc += w[c * C + k] * (z[(bb * C + k) * HW + p] * inv_scale);
    out[i] = acc;
  }
}

// in-place softmax over rows of a bf16 matrix [rows][n] (n % 512 == 0 not required; n % 8 == 0): one wave per row
__global__ __launch_bounds__(256) void vae_row_softmax_kernel(bf16_t* __restrict__ s, int rows, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  bf16_t* r = s + (size_t)row * n;
  float mx = -INFINITY;
  for (int c = lane * 8; c < n; c += 512) {
    const uint4 u = *reinterpret_cast<const uint4*>(r + c);
    const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(bf2f((bf16_t)(uu[i] & 0xffff)), bf2f((bf16_t)(uu[i] >> 16))));
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane * 8; c < n; c += 512) {
    const uint4 u = *reinterpret_cast<const uint4*>(r + c);
    const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      sum += __expf(bf2f((bf16_t)(uu[i] & 0xffff)) - mx) + __expf(bf2f((bf16_t)(uu[i] >> 16)) - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane * 8; c < n; c += 512) {
    const uint4 u = *reinterpret_cast<const uint4*>(r + c);
    const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
    uint4 o;
    uint32_t oo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      oo[i] = pack_bf2(__expf(bf2f((bf16_t)(uu[i] & 0xffff)) - mx) * inv, __expf(bf2f((bf16_t)(uu[i] >> 16)) - mx) * inv);
    o.x = oo[0]; o.y = oo[1]; o.z = oo[2]; o.w = oo[3];
    *reinterpret_cast<uint4*>(r + c) = o;
  }
}

// image = (x / 2 + 0.5).clamp(0, 1): NCHW fp32 -> NHWC uint8 (round(x * 255), what numpy_to_pil produces)
__global__ __launch_bounds__(256) void vae_to_uint8_kernel(const float* __restrict__ x, int C, int HW, int64_t total,
                                                           uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int p = (int)((i / C) % HW);
    const int64_t bb = i / ((int64_t)C * HW);
    float v = x[(bb * C + c) * HW + p] * 0.5f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = (uint8_t)__float2int_rn(v * 255.f);
  }
}

namespace {

struct VLoader {
  const WeightTable& wt;
  DevPool& pool;
  hipStream_t s;
  int norm(const std::string& p, int c, NormW* n) {
    GILL_TRY(load_f32(wt, pool, p + ".weight", c, &n->g, s));
    return load_f32(wt, pool, p + ".bias", c, &n->b, s);
  }
  // ups4: the conv follows a nearest-2x upsample — store the four pre-summed 2x2-tap kernels (gemm.hip "UPS4"; GILL_CONV_UPS4 = 0: off)
  int conv3(const std::string& p, int cin, int cout, ConvW* c, bool ups4 = false) {
    c->cin = cin; c->cout = cout;
    const gill_tensor* t;
    GILL_TRY(wt.get(p + ".weight", (int64_t)cout * cin * 9, &t));
    static const int ups4_on = [] { const char* v = getenv("GILL_CONV_UPS4"); return v ? atoi(v) : 1; }();
    if (ups4 && ups4_on) {
      c->ups4 = 1;
      GILL_TRY(pool.alloc(&c->w, (size_t)16 * cout * cin, false));
      GILL_TRY(conv_weight_relayout_ups4_launch(t->data, t->dtype, cout, cin, c->w, s));
      return load_f32(wt, pool, p + ".bias", cout, &c->b, s);
    }
    GILL_TRY(pool.alloc(&c->w, (size_t)cout * cin * 9, false));
    GILL_TRY(conv_weight_relayout_launch(t->data, t->dtype, cout, cin, c->w, s));   // tap-major K order (k_chunked = 0)
    return load_f32(wt, pool, p + ".bias", cout, &c->b, s);
  }
  int resnet(const std::string& p, int cin, int cout, VResW* r) {
    r->cin = cin; r->cout = cout;
    GILL_TRY(norm(p + ".norm1", cin, &r->n1));
    GILL_TRY(conv3(p + ".conv1", cin, cout, &r->c1));
    GILL_TRY(norm(p + ".norm2", cout, &r->n2));
    GILL_TRY(conv3(p + ".conv2", cout, cout, &r->c2));
    r->has_sc = (cin != cout);
    if (r->has_sc) {
      bf16_t* scw; float* scb;
      GILL_TRY(load_bf16(wt, pool, p + ".conv_shortcut.weight", (int64_t)cout * cin, &scw, s));
      GILL_TRY(load_f32(wt, pool, p + ".conv_shortcut.bias", cout, &scb, s));
      const int kf = 9 * cout + cin;
      std::vector<int32_t> ident(cout);
      for (int i = 0; i < cout; ++i) ident[i] = i;
      int32_t* idx;
      GILL_TRY(pool.alloc(&idx, (size_t)cout, false));
      GILL_CHECK_HIP(hipMemcpy(idx, ident.data(), sizeof(int32_t) * cout, hipMemcpyHostToDevice));
      GILL_TRY(pool.alloc(&r->c2f_w, (size_t)cout * kf, false));
      GILL_TRY(scatter_rows_bf16_launch(r->c2.w, cout, 9 * cout, idx, r->c2f_w, kf, s));
      GILL_TRY(scatter_rows_bf16_launch(scw, cout, cin, idx, r->c2f_w + 9 * cout, kf, s));
      GILL_TRY(pool.alloc(&r->c2f_b, (size_t)cout, false));
      hipLaunchKernelGGL(vae_vec_add_kernel, dim3(cdiv(cout, 256)), dim3(256), 0, s, r->c2.b, scb, cout, r->c2f_b);
      GILL_CHECK_HIP(hipGetLastError());
    }
    return 0;
  }
  // attention projection weight under either naming (diffusers >= 0.16 "to_q" / legacy "query")
  int attn_lin(const std::string& base, const char* modern, const char* legacy, int C, bf16_t* wdst, float* bdst) {
    std::string n = base + "." + modern;
    if (!wt.find(n + ".weight")) n = base + "." + legacy;
    const gill_tensor* t;
    GILL_TRY(wt.get(n + ".weight", (int64_t)C * C, &t));
    GILL_TRY(convert_to_bf16_launch(t->data, t->dtype, (int64_t)C * C, wdst, s));
    GILL_TRY(wt.get(n + ".bias", C, &t));
    return convert_to_f32_launch(t->data, t->dtype, C, bdst, s);
  }
};

struct VRun {
  gill_vae* m;
  hipStream_t s;
  int B;
  bool dry;

  float* stats_slot(size_t floats) {
    float* p = dry ? (float*)(uintptr_t)16 : m->gn_stats + m->gn_next;
    m->gn_next += (floats + 3) & ~(size_t)3;
    return p;
  }
  VTensor talloc(int H, int W, int C, bool want_stats) {
    VTensor t; t.H = H; t.W = W; t.C = C;
    t.p = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)B * H * W * C);
    const int G = m->cfg.norm_num_groups;
    if (want_stats && (H * W) % GN_SLAB_ROWS == 0 && C % G == 0 && gemm_fused_gn_ok(C, C / G))
      t.stats = stats_slot((size_t)B * (H * W / GN_SLAB_ROWS_MIN) * G * 2);
    return t;
  }
  void fuse_stats(GemmArgs& g, const VTensor& y) {
    if (!y.stats) return;
    g.gn_stats = y.stats; g.gn_groups = m->cfg.norm_num_groups; g.gn_cg = y.C / m->cfg.norm_num_groups;
    g.rows_per_batch = y.H * y.W;
  }
  int gemm(GemmArgs& g, VTensor* ys = nullptr) {
    if (dry) return 0;
    g.splitk = gemm_pick_splitk(g.M, g.N, g.K, g.act);
    while (g.splitk > 1 && (size_t)g.splitk * g.M * g.N > m->splitk_ws_floats) --g.splitk;
    g.ws = m->splitk_ws;
    if (ys && ys->stats) ys->nslab = ys->H * ys->W / gemm_gn_slab_rows(g);
    return gemm_launch(g, s);
  }
  int gnorm(const VTensor& x, const NormW& n, int silu, const VTensor& y) {
    const bool ready = x.stats != nullptr;
    const int HW = x.H * x.W, G = m->cfg.norm_num_groups;
    float* stats = ready ? x.stats : stats_slot(groupnorm_stats_floats(B, HW, G));
    float* tot = ready ? stats_slot(groupnorm_totals_floats(B, G, 0)) : nullptr;   // out-of-place totals (> 64 partials per group)
    if (dry) return 0;
    if (ready)
      return groupnorm_apply_launch(x.p, x.C, nullptr, 0, B, HW, G, n.g, n.b, 1e-6f, silu, y.p, stats, x.C / G, x.C,
                                    x.nslab, nullptr, 0, 0, s, 0.f, tot);
    return groupnorm_launch(x.p, x.C, nullptr, 0, B, HW, G, n.g, n.b, 1e-6f, silu, y.p, stats, s);
  }
  int conv(const VTensor& x, const ConvW& w, int ups, const bf16_t* resid, VTensor& y) {
    GemmArgs g;
    g.conv = 1; g.IH = x.H; g.IW = x.W; g.OH = y.H; g.OW = y.W; g.Cin = w.cin; g.stride = 1; g.ups = ups;
    g.M = B * y.H * y.W; g.N = w.cout; g.K = 9 * w.cin;
    g.A = x.p; g.K1 = x.C; g.W = w.w; g.bias = w.b;
    if (ups && w.ups4) { g.ups = 2; g.K = 4 * w.cin; }
    g.rows_per_batch = y.H * y.W;
    g.resid = resid; g.ldr = w.cout;
    g.C = y.p; g.ldc = w.cout;
    fuse_stats(g, y);
    return gemm(g, &y);
  }
  int resnet(const VTensor& x, const VResW& w, VTensor* out, bool out_stats) {
    const int H = x.H, Wd = x.W;
    *out = talloc(H, Wd, w.cout, out_stats);
    const size_t mk = m->arena.off;
    VTensor n1 = talloc(H, Wd, w.cin, false);
    GILL_TRY(gnorm(x, w.n1, 1, n1));
    VTensor h = talloc(H, Wd, w.cout, true);
    GILL_TRY(conv(n1, w.c1, 0, nullptr, h));
    VTensor n2 = talloc(H, Wd, w.cout, false);
    GILL_TRY(gnorm(h, w.n2, 1, n2));
    if (w.has_sc) {
      GemmArgs g;
      g.conv = 1; g.IH = H; g.IW = Wd; g.OH = H; g.OW = Wd; g.Cin = w.cout; g.stride = 1;
      g.M = B * H * Wd; g.N = w.cout; g.K = 9 * w.cout + w.cin;
      g.A = n2.p; g.K1 = w.cout;
      g.X1 = x.p; g.KX = w.cin; g.KX1 = w.cin;
      g.W = w.c2f_w; g.bias = w.c2f_b;
      g.rows_per_batch = H * Wd;
      g.C = out->p; g.ldc = w.cout;
      fuse_stats(g, *out);
      GILL_TRY(gemm(g, out));
    } else {
      GILL_TRY(conv(n2, w.c2, 0, x.p, *out));
    }
    m->arena.off = mk;
    return 0;
  }
  // single-head attention over the HW tokens of a C-channel map (+ residual)
  int attention(const VTensor& x, VTensor* out, bool out_stats) {
    const int C = x.C, HW = x.H * x.W, M = B * HW;
    *out = talloc(x.H, x.W, C, out_stats);
    const size_t mk = m->arena.off;
    VTensor n = talloc(x.H, x.W, C, false);
    GILL_TRY(gnorm(x, m->attn_gn, 0, n));
    bf16_t* q = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * C);
    bf16_t* k = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * C);
    bf16_t* vt = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * C);
    bf16_t* sc = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)HW * HW);   // scores of ONE image at a time
    bf16_t* o = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)M * C);
    {
      // heads = 1, dp = C: Q and K come out plain row-major [B*HW][C], V transposed [B][C][HW]
      GemmArgs g;
      g.M = M; g.N = 3 * C; g.K = C; g.K1 = C; g.A = n.p; g.lda = C; g.W = m->attn_wqkv; g.bias = m->attn_bqkv;
      g.out_mode = OUT_QKV; g.Cq = q; g.Ck = k; g.Cvt = vt; g.heads = 1; g.dp = C; g.dpv = C; g.ntok = HW;
      g.ntok_pad_q = HW; g.ntok_pad_kv = HW; g.seg_base = 0;
      g.qscale = 1.0f / sqrtf((float)C);
      GILL_TRY(gemm(g));
    }
    for (int b = 0; b < B; ++b) {
      GemmArgs g1;   // S = Q K^T
      g1.M = HW; g1.N = HW; g1.K = C; g1.K1 = C; g1.A = q + (size_t)b * HW * C; g1.lda = C; g1.W = k + (size_t)b * HW * C;
      g1.C = sc; g1.ldc = HW;
      GILL_TRY(gemm(g1));
      if (!dry) {
        hipLaunchKernelGGL(vae_row_softmax_kernel, dim3(cdiv(HW, 4)), dim3(256), 0, s, sc, HW, HW);
        GILL_CHECK_HIP(hipGetLastError());
      }
      GemmArgs g2;   // O = P V
      g2.M = HW; g2.N = C; g2.K = HW; g2.K1 = HW; g2.A = sc; g2.lda = HW; g2.W = vt + (size_t)b * C * HW;
      g2.C = o + (size_t)b * HW * C; g2.ldc = C;
      GILL_TRY(gemm(g2));
    }
    GemmArgs g3;   // to_out + residual
    g3.M = M; g3.N = C; g3.K = C; g3.K1 = C; g3.A = o; g3.lda = C; g3.W = m->attn_wo; g3.bias = m->attn_bo;
    g3.resid = x.p; g3.ldr = C; g3.C = out->p; g3.ldc = C;
    fuse_stats(g3, *out);
    GILL_TRY(gemm(g3, out));
    m->arena.off = mk;
    return 0;
  }

  int decode(const float* latents, float* img_out_f32) {
    const gill_vae_config& c = m->cfg;
    const int* ch = c.block_out_channels;
    const int L = c.latent_size;
    const int ctop = ch[3];
    m->arena.off = 0;
    m->gn_next = 0;
    if (!dry) {
      const int64_t total = (int64_t)B * c.latent_channels * L * L;
      int blocks = (int)((total + 255) / 256);
      hipLaunchKernelGGL(vae_latent_prep_kernel, dim3(blocks), dim3(256), 0, s, latents, m->pq_w, m->pq_b,
                         1.0f / c.scaling_factor, c.latent_channels, L * L, total, m->lat_prep);
      GILL_CHECK_HIP(hipGetLastError());
    }
    VTensor x = talloc(L, L, ctop, true);
    {
      bf16_t* col = (bf16_t*)m->arena.alloc(sizeof(bf16_t) * (size_t)B * L * L * 64);
      if (!dry) GILL_TRY(im2col_nchw_launch(m->lat_prep, B, c.latent_channels, L, L, 64, col, s));
      GemmArgs g;
      g.M = B * L * L; g.N = ctop; g.K = 64; g.K1 = 64; g.A = col; g.lda = 64; g.W = m->conv_in_w; g.bias = m->conv_in_b;
      g.C = x.p; g.ldc = ctop;
      fuse_stats(g, x);
      GILL_TRY(gemm(g, &x));
    }
    { VTensor y; GILL_TRY(resnet(x, m->mid_res[0], &y, true)); x = y; }
    { VTensor y; GILL_TRY(attention(x, &y, true)); x = y; }
    { VTensor y; GILL_TRY(resnet(x, m->mid_res[1], &y, true)); x = y; }
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 3; ++j) {
        VTensor y;
        // the output feeds the next resnet's norm1 / conv_norm_out, except before an upsampling conv
        GILL_TRY(resnet(x, m->up_res[i][j], &y, !(j == 2 && i < 3)));
        x = y;
      }
      if (i < 3) {
        // (4-tap form: the epilogue's GroupNorm slabs are 64 SOURCE rows of one parity class — tiny grids leave the sums to the consumer)
        VTensor y = talloc(x.H * 2, x.W * 2, x.C, !m->up_us[i].ups4 || (x.H * x.W) % GN_SLAB_ROWS == 0);
        GILL_TRY(conv(x, m->up_us[i], 1, nullptr, y));
        x = y;
      }
    }
    VTensor n = talloc(x.H, x.W, x.C, false);
    GILL_TRY(gnorm(x, m->norm_out, 1, n));
    if (!dry) GILL_TRY(conv_out_launch(n.p, m->conv_out_w, m->conv_out_b, B, x.C, x.H, x.W, c.out_channels, img_out_f32, s));
    return 0;
  }
};

}  // namespace

extern "C" int gill_vae_create(gill_vae** out, const gill_vae_config* cfg, const gill_tensor* weights, int n_weights) {
  GILL_REQUIRE(out && cfg && weights, "null argument");
  GILL_REQUIRE(cfg->layers_per_block == 2 && cfg->max_batch >= 1, "only layers_per_block == 2 is supported");
  GILL_REQUIRE(cfg->latent_channels * 9 <= 64 && cfg->out_channels <= 8, "latent/out channel counts too large");
  for (int i = 0; i < 4; ++i) GILL_REQUIRE(cfg->block_out_channels[i] % 64 == 0, "block_out_channels must be multiples of 64");
  GILL_REQUIRE((cfg->latent_size * cfg->latent_size) % 64 == 0, "latent_size^2 must be a multiple of 64");
  gill_vae* m = new gill_vae();
  m->cfg = *cfg;
  int rc = 0;
  auto fail = [&](int r) { delete m; return r; };
  WeightTable wt(weights, n_weights);
  hipStream_t s = nullptr;
  VLoader L{wt, m->pool, s};
  const int* ch = cfg->block_out_channels;
  const int ctop = ch[3], lc = cfg->latent_channels;
  if ((rc = load_f32(wt, m->pool, "post_quant_conv.weight", (int64_t)lc * lc, &m->pq_w, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, "post_quant_conv.bias", lc, &m->pq_b, s))) return fail(rc);
  {
    const gill_tensor* t;
    bf16_t* tmp; int32_t* idx;
    const int kk = lc * 9;
    if ((rc = wt.get("decoder.conv_in.weight", (int64_t)ctop * kk, &t))) return fail(rc);
    if ((rc = m->pool.alloc(&tmp, (size_t)ctop * kk, false))) return fail(rc);
    if ((rc = conv_weight_relayout_launch(t->data, t->dtype, ctop, lc, tmp, s))) return fail(rc);
    if ((rc = m->pool.alloc(&m->conv_in_w, (size_t)ctop * 64, true))) return fail(rc);
    std::vector<int32_t> rows(ctop);
    for (int i = 0; i < ctop; ++i) rows[i] = i;
    if ((rc = m->pool.alloc(&idx, (size_t)ctop, false))) return fail(rc);
    if (hipMemcpy(idx, rows.data(), sizeof(int32_t) * ctop, hipMemcpyHostToDevice) != hipSuccess) return fail(-1);
    if ((rc = scatter_rows_bf16_launch(tmp, ctop, kk, idx, m->conv_in_w, 64, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, "decoder.conv_in.bias", ctop, &m->conv_in_b, s))) return fail(rc);
  }
  if ((rc = L.resnet("decoder.mid_block.resnets.0", ctop, ctop, &m->mid_res[0]))) return fail(rc);
  if ((rc = L.resnet("decoder.mid_block.resnets.1", ctop, ctop, &m->mid_res[1]))) return fail(rc);
  {
    const std::string a = "decoder.mid_block.attentions.0";
    if ((rc = L.norm(a + ".group_norm", ctop, &m->attn_gn))) return fail(rc);
    if ((rc = m->pool.alloc(&m->attn_wqkv, (size_t)3 * ctop * ctop, false))) return fail(rc);
    if ((rc = m->pool.alloc(&m->attn_bqkv, (size_t)3 * ctop, false))) return fail(rc);
    if ((rc = m->pool.alloc(&m->attn_wo, (size_t)ctop * ctop, false))) return fail(rc);
    if ((rc = m->pool.alloc(&m->attn_bo, (size_t)ctop, false))) return fail(rc);
    if ((rc = L.attn_lin(a, "to_q", "query", ctop, m->attn_wqkv, m->attn_bqkv))) return fail(rc);
    if ((rc = L.attn_lin(a, "to_k", "key", ctop, m->attn_wqkv + (size_t)ctop * ctop, m->attn_bqkv + ctop))) return fail(rc);
    if ((rc = L.attn_lin(a, "to_v", "value", ctop, m->attn_wqkv + (size_t)2 * ctop * ctop, m->attn_bqkv + 2 * ctop))) return fail(rc);
    if ((rc = L.attn_lin(a, "to_out.0", "proj_attn", ctop, m->attn_wo, m->attn_bo))) return fail(rc);
  }
  // up blocks walk the channel list backwards: [512, 512, 256, 128]
  int prev = ctop;
  for (int i = 0; i < 4; ++i) {
    const int outc = ch[3 - i];
    m->up_res[i].resize(3);
    for (int j = 0; j < 3; ++j) {
      const std::string p = "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
      if ((rc = L.resnet(p, j == 0 ? prev : outc, outc, &m->up_res[i][j]))) return fail(rc);
    }
    if (i < 3)
      if ((rc = L.conv3("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", outc, outc, &m->up_us[i], true))) return fail(rc);
    prev = outc;
  }
  if ((rc = L.norm("decoder.conv_norm_out", ch[0], &m->norm_out))) return fail(rc);
  {
    const gill_tensor* t;
    if ((rc = wt.get("decoder.conv_out.weight", (int64_t)cfg->out_channels * ch[0] * 9, &t))) return fail(rc);
    if ((rc = m->pool.alloc(&m->conv_out_w, (size_t)cfg->out_channels * ch[0] * 9, false))) return fail(rc);
    if ((rc = conv_weight_relayout_launch(t->data, t->dtype, cfg->out_channels, ch[0], m->conv_out_w, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, "decoder.conv_out.bias", cfg->out_channels, &m->conv_out_b, s))) return fail(rc);
  }
  // workspace: dry run sizes the arena and counts the GroupNorm slots
  const int B = cfg->max_batch, Lz = cfg->latent_size;
  m->arena.dry = true; m->arena.off = 0; m->arena.high = 0;
  VRun r{m, nullptr, B, true};
  if ((rc = r.decode(nullptr, nullptr))) return fail(rc);
  if ((rc = m->pool.alloc(&m->arena_mem, m->arena.high + (1 << 20), true))) return fail(rc);
  m->arena.base = m->arena_mem; m->arena.dry = false;
  m->gn_floats = m->gn_next + 64;
  if ((rc = m->pool.alloc(&m->gn_stats, m->gn_floats))) return fail(rc);
  m->splitk_ws_floats = (size_t)16 << 20;
  if ((rc = m->pool.alloc(&m->splitk_ws, m->splitk_ws_floats, false))) return fail(rc);
  if ((rc = m->pool.alloc(&m->lat_prep, (size_t)B * cfg->latent_channels * Lz * Lz))) return fail(rc);
  if ((rc = m->pool.alloc(&m->img_f32, (size_t)B * cfg->out_channels * 64 * Lz * Lz))) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { gill_set_error("vae create: device sync failed"); return fail(-1); }
  *out = m;
  return 0;
}

extern "C" void gill_vae_destroy(gill_vae* h) { delete h; }

extern "C" int gill_vae_decode(gill_vae* m, const float* latents, int B, float* image_f32, uint8_t* image_u8, void* stream) {
  GILL_REQUIRE(m && latents && (image_f32 || image_u8), "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "batch exceeds the VAE handle's max_batch");
  hipStream_t s = (hipStream_t)stream;
  float* img = image_f32 ? image_f32 : m->img_f32;
  VRun r{m, s, B, false};
  GILL_TRY(r.decode(latents, img));
  if (image_u8) {
    const int side = 8 * m->cfg.latent_size;
    const int64_t total = (int64_t)B * side * side * m->cfg.out_channels;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(vae_to_uint8_kernel, dim3(blocks), dim3(256), 0, s, img, m->cfg.out_channels, side * side, total, image_u8);
    GILL_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
