// CLIP vision tower (transformers CLIPVisionModel as the reference calls it: gill/models.py:78-96 builds it,
// :129-152 get_visual_embs reads `outputs.pooler_output`) on gfx950, through the same GEMM / attention / LayerNorm kernels
// as the OPT engine.
//   pixel_values (B,3,S,S) fp32 -> patch_embedding (PxP conv, stride P, no bias: im2col + MFMA GEMM) -> [class | patches]
//   + position_embedding -> pre_layrnorm -> L x { LN1, QKV (bias), softmax(QK^T/sqrt(d)) V, out_proj, +res, LN2, fc1,
//   quick_gelu, fc2, +res } -> post_layernorm(row 0) = pooler_output (B, D) fp32.
// fp32 residual stream, bf16 GEMM operands, fp32 accumulation (as opt.hip).
#include "ops.h"
#include "engine_util.h"
#include <string>
#include <vector>

namespace {
struct ClipLayer {
  bf16_t* wqkv = nullptr; float* bqkv = nullptr;   // [3D][D] rows: q | k | v
  bf16_t* wo = nullptr; float* bo = nullptr;
  bf16_t* w1 = nullptr; float* b1 = nullptr;
  bf16_t* w2 = nullptr; float* b2 = nullptr;
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
};
}  // namespace

struct gill_clip {
  gill_clip_config cfg;
  DevPool pool;
  int ntok = 0, npatch = 0, kpatch = 0, kpad = 0, dp = 0, dpv = 0;
  bf16_t* wpatch = nullptr;   // [D][kpad]: patch_embedding.weight flattened (c, ky, kx), zero-padded to a multiple of 64
  float* cls = nullptr;       // [D]
  float* pos = nullptr;       // [ntok][D]
  float *preg = nullptr, *preb = nullptr, *postg = nullptr, *postb = nullptr;
  std::vector<ClipLayer> layers;
  // workspace
  bf16_t* col = nullptr;      // [B*npatch][kpad]
  float* h = nullptr;         // [B*ntok][D]
  bf16_t* nbuf = nullptr;     // [B*ntok][D]
  bf16_t* ff = nullptr;       // [B*ntok][F]
  bf16_t *q = nullptr, *k = nullptr, *vt = nullptr, *o = nullptr;
  float* gath = nullptr;      // [B][D]
  int32_t* idx_dev = nullptr; // [B]
  float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
};

// col[(b*npatch + py*G + px)][(c*P + ky)*P + kx] = bf16(pixel[b][c][py*P + ky][px*P + kx]); columns >= 3*P*P are zero
__global__ __launch_bounds__(256) void clip_im2col_kernel(const float* __restrict__ px, int B, int S, int P, int G, int kpatch,
                                                          int kpad, bf16_t* __restrict__ col) {
  const int64_t total = (int64_t)B * G * G * kpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kcol = (int)(i % kpad);
    const int64_t row = i / kpad;
    float v = 0.f;
    if (kcol < kpatch) {
      const int kx = kcol % P, ky = (kcol / P) % P, c = kcol / (P * P);
      const int pxi = (int)(row % G), pyi = (int)((row / G) % G), b = (int)(row / ((int64_t)G * G));
      v = px[(((int64_t)b * 3 + c) * S + pyi * P + ky) * S + pxi * P + kx];
    }
    col[i] = f2bf(v);
  }
}

// h[b][0][:] = class_embedding + pos[0]; h[b][1+p][:] += pos[1+p]   (rows 1.. were written by the patch GEMM)
__global__ __launch_bounds__(256) void clip_embed_finish_kernel(float* __restrict__ h, const float* __restrict__ cls,
                                                                const float* __restrict__ pos, int ntok, int D) {
  const int row = blockIdx.x;          // b * ntok + t
  const int t = row % ntok;
  float* o = h + (size_t)row * D;
  const float* pe = pos + (size_t)t * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) o[c] = (t == 0 ? cls[c] : o[c]) + pe[c];
}

extern "C" int gill_clip_create(gill_clip** out, const gill_clip_config* cfg, const gill_tensor* weights, int n_weights) {
  GILL_REQUIRE(out && cfg && weights, "null argument");
  const int D = cfg->hidden_size, F = cfg->intermediate_size, H = cfg->num_heads, P = cfg->patch_size, S = cfg->image_size;
  GILL_REQUIRE(D % 64 == 0 && F % 64 == 0 && H > 0 && D % H == 0, "CLIP dims must be multiples of 64");
  GILL_REQUIRE(P > 0 && S % P == 0 && cfg->max_batch >= 1 && cfg->num_layers >= 1, "bad CLIP geometry");
  const int hd = D / H;
  GILL_REQUIRE(attn_padded_dim(hd) == hd, "CLIP head dim must be one of 48/64/80/128/160");
  gill_clip* m = new gill_clip();
  m->cfg = *cfg;
  m->dp = hd; m->dpv = round_up(hd, 32);
  const int G = S / P;
  m->npatch = G * G; m->ntok = m->npatch + 1; m->kpatch = 3 * P * P; m->kpad = round_up(m->kpatch, 64);
  WeightTable wt(weights, n_weights);
  hipStream_t s = nullptr;
  int rc = 0;
  auto fail = [&](int r) { delete m; return r; };
  const std::string vm = "vision_model.";
  {
    // patch_embedding.weight [D][3][P][P] -> bf16 [D][kpad] (zero padded columns)
    const gill_tensor* t;
    if ((rc = wt.get(vm + "embeddings.patch_embedding.weight", (int64_t)D * m->kpatch, &t))) return fail(rc);
    bf16_t* tmp;
    if ((rc = m->pool.alloc(&tmp, (size_t)D * m->kpatch, false))) return fail(rc);
    if ((rc = convert_to_bf16_launch(t->data, t->dtype, (int64_t)D * m->kpatch, tmp, s))) return fail(rc);
    if ((rc = m->pool.alloc(&m->wpatch, (size_t)D * m->kpad, true))) return fail(rc);
    if (hipMemcpy2D(m->wpatch, sizeof(bf16_t) * m->kpad, tmp, sizeof(bf16_t) * m->kpatch, sizeof(bf16_t) * m->kpatch, D,
                    hipMemcpyDeviceToDevice) != hipSuccess) { gill_set_error("clip create: weight re-layout failed"); return fail(-1); }
  }
  if ((rc = load_f32(wt, m->pool, vm + "embeddings.class_embedding", D, &m->cls, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, vm + "embeddings.position_embedding.weight", (int64_t)m->ntok * D, &m->pos, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, vm + "pre_layrnorm.weight", D, &m->preg, s))) return fail(rc);   // (sic: the HF name)
  if ((rc = load_f32(wt, m->pool, vm + "pre_layrnorm.bias", D, &m->preb, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, vm + "post_layernorm.weight", D, &m->postg, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, vm + "post_layernorm.bias", D, &m->postb, s))) return fail(rc);
  m->layers.resize(cfg->num_layers);
  for (int i = 0; i < cfg->num_layers; ++i) {
    ClipLayer& L = m->layers[i];
    const std::string p = vm + "encoder.layers." + std::to_string(i) + ".";
    if ((rc = m->pool.alloc(&L.wqkv, (size_t)3 * D * D, false))) return fail(rc);
    if ((rc = m->pool.alloc(&L.bqkv, (size_t)3 * D, false))) return fail(rc);
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    for (int j = 0; j < 3; ++j) {
      const gill_tensor* t;
      if ((rc = wt.get(p + "self_attn." + names[j] + ".weight", (int64_t)D * D, &t))) return fail(rc);
      if ((rc = convert_to_bf16_launch(t->data, t->dtype, (int64_t)D * D, L.wqkv + (size_t)j * D * D, s))) return fail(rc);
      if ((rc = wt.get(p + "self_attn." + names[j] + ".bias", D, &t))) return fail(rc);
      if ((rc = convert_to_f32_launch(t->data, t->dtype, D, L.bqkv + (size_t)j * D, s))) return fail(rc);
    }
    if ((rc = load_bf16(wt, m->pool, p + "self_attn.out_proj.weight", (int64_t)D * D, &L.wo, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "self_attn.out_proj.bias", D, &L.bo, s))) return fail(rc);
    if ((rc = load_bf16(wt, m->pool, p + "mlp.fc1.weight", (int64_t)F * D, &L.w1, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "mlp.fc1.bias", F, &L.b1, s))) return fail(rc);
    if ((rc = load_bf16(wt, m->pool, p + "mlp.fc2.weight", (int64_t)D * F, &L.w2, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "mlp.fc2.bias", D, &L.b2, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "layer_norm1.weight", D, &L.ln1g, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "layer_norm1.bias", D, &L.ln1b, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "layer_norm2.weight", D, &L.ln2g, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "layer_norm2.bias", D, &L.ln2b, s))) return fail(rc);
  }
  const size_t R = (size_t)cfg->max_batch * m->ntok;
  const size_t Tpad = round_up(m->ntok, 32);
  if ((rc = m->pool.alloc(&m->col, (size_t)cfg->max_batch * m->npatch * m->kpad))) return fail(rc);
  if ((rc = m->pool.alloc(&m->h, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->nbuf, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->ff, R * F))) return fail(rc);
  if ((rc = m->pool.alloc(&m->q, (size_t)cfg->max_batch * H * Tpad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->k, (size_t)cfg->max_batch * H * Tpad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->vt, (size_t)cfg->max_batch * H * m->dpv * Tpad))) return fail(rc);
  if ((rc = m->pool.alloc(&m->o, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->gath, (size_t)cfg->max_batch * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->idx_dev, (size_t)cfg->max_batch))) return fail(rc);
  {
    std::vector<int32_t> idx(cfg->max_batch);
    for (int b = 0; b < cfg->max_batch; ++b) idx[b] = b * m->ntok;     // the class-token row of every image
    if (hipMemcpy(m->idx_dev, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice) != hipSuccess) {
      gill_set_error("clip create: index upload failed"); return fail(-1);
    }
  }
  m->splitk_ws_floats = (size_t)16 * R * (size_t)(F > 3 * D ? F : 3 * D);
  if ((rc = m->pool.alloc(&m->splitk_ws, m->splitk_ws_floats, false))) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { gill_set_error("clip create: device sync failed"); return fail(-1); }
  *out = m;
  return 0;
}

extern "C" void gill_clip_destroy(gill_clip* h) { delete h; }

namespace {
struct ClipRun {
  gill_clip* m;
  hipStream_t s;
  int linear(const bf16_t* A, int M, const bf16_t* W, const float* b, int N, int K, const float* resid, int act, void* out,
             bool out_f32, int ldc) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.K1 = K; g.A = A; g.lda = K; g.W = W; g.bias = b;
    g.resid = resid; g.ldr = N; g.resid_f32 = 1;
    g.act = act; g.out_mode = out_f32 ? OUT_F32 : OUT_BF16; g.C = out; g.ldc = ldc;
    g.splitk = gemm_pick_splitk(M, N, K, act);
    if ((size_t)g.splitk * M * N > m->splitk_ws_floats) g.splitk = 1;
    g.ws = m->splitk_ws;
    return gemm_launch(g, s);
  }
};
}  // namespace

extern "C" int gill_clip_forward(gill_clip* m, const float* pixel_values, int B, float* pooled_out, void* stream) {
  GILL_REQUIRE(m && pixel_values && pooled_out, "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "batch exceeds the CLIP handle's max_batch");
  hipStream_t s = (hipStream_t)stream;
  const gill_clip_config& c = m->cfg;
  const int D = c.hidden_size, F = c.intermediate_size, T = m->ntok, R = B * T;
  const int Tpad = round_up(T, 32);
  ClipRun r{m, s};
  {
    const int64_t total = (int64_t)B * m->npatch * m->kpad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(clip_im2col_kernel, dim3(blocks), dim3(256), 0, s, pixel_values, B, c.image_size, c.patch_size,
                       c.image_size / c.patch_size, m->kpatch, m->kpad, m->col);
    GILL_CHECK_HIP(hipGetLastError());
    for (int b = 0; b < B; ++b)      // rows b*T+1 .. of the fp32 stream (row b*T is the class token)
      GILL_TRY(r.linear(m->col + (size_t)b * m->npatch * m->kpad, m->npatch, m->wpatch, nullptr, D, m->kpad, nullptr, ACT_NONE,
                        m->h + ((size_t)b * T + 1) * D, true, D));
    hipLaunchKernelGGL(clip_embed_finish_kernel, dim3(R), dim3(256), 0, s, m->h, m->cls, m->pos, T, D);
    GILL_CHECK_HIP(hipGetLastError());
  }
  GILL_TRY(layernorm_f32out_launch(m->h, 1, m->preg, m->preb, m->h, R, D, 1e-5f, s));   // in place, row by row
  for (const ClipLayer& L : m->layers) {
    GILL_TRY(layernorm_launch(m->h, 1, L.ln1g, L.ln1b, m->nbuf, R, D, 1e-5f, s));
    {
      GemmArgs g;
      g.M = R; g.N = 3 * D; g.K = D; g.K1 = D; g.A = m->nbuf; g.lda = D; g.W = L.wqkv; g.bias = L.bqkv;
      g.out_mode = OUT_QKV; g.Cq = m->q; g.Ck = m->k; g.Cvt = m->vt;
      g.heads = c.num_heads; g.dp = m->dp; g.dpv = m->dpv; g.ntok = T; g.ntok_pad_q = Tpad; g.ntok_pad_kv = Tpad;
      g.seg_base = 0;
      g.qscale = 1.4426950408889634f / sqrtf((float)m->dp);
      g.splitk = gemm_pick_splitk(R, 3 * D, D, 0);
      if ((size_t)g.splitk * R * 3 * D > m->splitk_ws_floats) g.splitk = 1;
      g.ws = m->splitk_ws;
      GILL_TRY(gemm_launch(g, s));
    }
    {
      AttnArgs a;
      a.Q = m->q; a.K = m->k; a.Vt = m->vt; a.O = m->o;
      a.B = B; a.H = c.num_heads; a.nq = T; a.nkv = T; a.nq_pad = Tpad; a.nkv_pad = Tpad;
      a.dp = m->dp; a.dpv = m->dpv; a.ldo = D; a.scale = 1.0f / sqrtf((float)m->dp); a.causal = 0;
      GILL_TRY(attention_launch(a, s));
    }
    GILL_TRY(r.linear(m->o, R, L.wo, L.bo, D, D, m->h, ACT_NONE, m->h, true, D));
    GILL_TRY(layernorm_launch(m->h, 1, L.ln2g, L.ln2b, m->nbuf, R, D, 1e-5f, s));
    GILL_TRY(r.linear(m->nbuf, R, L.w1, L.b1, F, D, nullptr, ACT_QUICK_GELU, m->ff, false, F));
    GILL_TRY(r.linear(m->ff, R, L.w2, L.b2, D, F, m->h, ACT_NONE, m->h, true, D));
  }
  // pooler_output = post_layernorm(last_hidden_state[:, 0])
  GILL_TRY(gather_rows_launch(m->h, 1, m->idx_dev, B, D, m->gath, 1, s));
  GILL_TRY(layernorm_f32out_launch(m->gath, 1, m->postg, m->postb, pooled_out, B, D, 1e-5f, s));
  return 0;
}
