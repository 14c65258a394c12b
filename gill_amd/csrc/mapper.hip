// Stage 2 — GILLMapper: gill.layers.TextFcLayer(mode='gill_mapper').forward  (reference
// gill/layers.py:28-53; module built at layers.py:17-24).
//
//   x = x + input_embs                                    layers.py:31-32
//   x = fc(x)                         Linear(in_dim, 512)  layers.py:42
//   x = tfm(src=x, tgt=query_embs)    nn.Transformer(d=512, nhead=4, enc 4 / dec 4, ff 2048, ReLU,
//                                     norm_first, eps 1e-5, no masks)          layers.py:43
//   out = model(x)                    Linear(512, out_dim) layers.py:44
//
// The residual stream is kept in fp32 (rows are few: 8 and 77 per sample); every GEMM runs bf16 x bf16
// on MFMA with fp32 accumulation; attention is the shared flash kernel (4 heads x 128).
#include "engine_util.h"

namespace {

struct LinearW { bf16_t* w = nullptr; float* b = nullptr; int out = 0, in = 0; };
struct NormW { float* g = nullptr; float* b = nullptr; };
struct MhaW { LinearW in_proj; LinearW out_proj; };
struct EncLayer { MhaW sa; LinearW l1, l2; NormW n1, n2; };
struct DecLayer { MhaW sa, ca; LinearW l1, l2; NormW n1, n2, n3; };

}  // namespace

struct gill_mapper {
  gill_mapper_config cfg;
  DevPool pool;
  LinearW fc, model;
  float* query = nullptr;  // (n_out, hidden) fp32
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  NormW enc_norm, dec_norm;
  // workspace (sized for max_batch)
  bf16_t* x0 = nullptr;     // (B*8, in_dim)
  float* h_enc = nullptr;   // (B*8, hidden)
  float* h_dec = nullptr;   // (B*77, hidden)
  bf16_t* nbuf = nullptr;   // (B*77, hidden) normalised activations
  bf16_t* mem = nullptr;    // (B*8, hidden) encoder output
  bf16_t* ff = nullptr;     // (B*77, ffn)
  bf16_t* q = nullptr; bf16_t* k = nullptr; bf16_t* vt = nullptr; bf16_t* o = nullptr;
  float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
  int dp = 0, dpv = 0;
};

static int load_linear(const WeightTable& wt, DevPool& pool, const std::string& wname, const std::string& bname, int out,
                       int in, LinearW* l, hipStream_t s) {
  l->out = out; l->in = in;
  GILL_TRY(load_bf16(wt, pool, wname, (int64_t)out * in, &l->w, s));
  GILL_TRY(load_f32(wt, pool, bname, out, &l->b, s));
  return 0;
}
static int load_norm(const WeightTable& wt, DevPool& pool, const std::string& prefix, int dim, NormW* n, hipStream_t s) {
  GILL_TRY(load_f32(wt, pool, prefix + ".weight", dim, &n->g, s));
  GILL_TRY(load_f32(wt, pool, prefix + ".bias", dim, &n->b, s));
  return 0;
}
static int load_mha(const WeightTable& wt, DevPool& pool, const std::string& prefix, int dim, MhaW* m, hipStream_t s) {
  GILL_TRY(load_linear(wt, pool, prefix + ".in_proj_weight", prefix + ".in_proj_bias", 3 * dim, dim, &m->in_proj, s));
  GILL_TRY(load_linear(wt, pool, prefix + ".out_proj.weight", prefix + ".out_proj.bias", dim, dim, &m->out_proj, s));
  return 0;
}

extern "C" int gill_mapper_create(gill_mapper** out, const gill_mapper_config* cfg, const gill_tensor* weights,
                                  int n_weights) {
  GILL_REQUIRE(out && cfg && weights, "null argument");
  GILL_REQUIRE(cfg->hidden_dim % 64 == 0 && cfg->in_dim % 64 == 0 && cfg->ffn_dim % 64 == 0,
               "mapper dims must be multiples of 64");
  GILL_REQUIRE(cfg->hidden_dim % cfg->num_heads == 0, "hidden_dim must divide by num_heads");
  const int hd = cfg->hidden_dim / cfg->num_heads;
  GILL_REQUIRE(attn_padded_dim(hd) == hd, "mapper head dim must be one of 48/64/80/128/160");
  GILL_REQUIRE(cfg->max_batch > 0, "max_batch must be positive");
  gill_mapper* m = new gill_mapper();
  m->cfg = *cfg;
  m->dp = hd; m->dpv = round_up(hd, 32);
  WeightTable wt(weights, n_weights);
  hipStream_t s = nullptr;
  const int Hd = cfg->hidden_dim, F = cfg->ffn_dim;
  int rc = 0;
  auto fail = [&](int r) { delete m; return r; };
  if ((rc = load_linear(wt, m->pool, "fc.weight", "fc.bias", Hd, cfg->in_dim, &m->fc, s))) return fail(rc);
  if ((rc = load_linear(wt, m->pool, "model.weight", "model.bias", cfg->out_dim, Hd, &m->model, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, "query_embs", (int64_t)cfg->num_output_tokens * Hd, &m->query, s))) return fail(rc);
  m->enc.resize(cfg->num_enc_layers);
  for (int i = 0; i < cfg->num_enc_layers; ++i) {
    const std::string p = "tfm.encoder.layers." + std::to_string(i);
    EncLayer& L = m->enc[i];
    if ((rc = load_mha(wt, m->pool, p + ".self_attn", Hd, &L.sa, s))) return fail(rc);
    if ((rc = load_linear(wt, m->pool, p + ".linear1.weight", p + ".linear1.bias", F, Hd, &L.l1, s))) return fail(rc);
    if ((rc = load_linear(wt, m->pool, p + ".linear2.weight", p + ".linear2.bias", Hd, F, &L.l2, s))) return fail(rc);
    if ((rc = load_norm(wt, m->pool, p + ".norm1", Hd, &L.n1, s))) return fail(rc);
    if ((rc = load_norm(wt, m->pool, p + ".norm2", Hd, &L.n2, s))) return fail(rc);
  }
  if ((rc = load_norm(wt, m->pool, "tfm.encoder.norm", Hd, &m->enc_norm, s))) return fail(rc);
  m->dec.resize(cfg->num_dec_layers);
  for (int i = 0; i < cfg->num_dec_layers; ++i) {
    const std::string p = "tfm.decoder.layers." + std::to_string(i);
    DecLayer& L = m->dec[i];
    if ((rc = load_mha(wt, m->pool, p + ".self_attn", Hd, &L.sa, s))) return fail(rc);
    if ((rc = load_mha(wt, m->pool, p + ".multihead_attn", Hd, &L.ca, s))) return fail(rc);
    if ((rc = load_linear(wt, m->pool, p + ".linear1.weight", p + ".linear1.bias", F, Hd, &L.l1, s))) return fail(rc);
    if ((rc = load_linear(wt, m->pool, p + ".linear2.weight", p + ".linear2.bias", Hd, F, &L.l2, s))) return fail(rc);
    if ((rc = load_norm(wt, m->pool, p + ".norm1", Hd, &L.n1, s))) return fail(rc);
    if ((rc = load_norm(wt, m->pool, p + ".norm2", Hd, &L.n2, s))) return fail(rc);
    if ((rc = load_norm(wt, m->pool, p + ".norm3", Hd, &L.n3, s))) return fail(rc);
  }
  if ((rc = load_norm(wt, m->pool, "tfm.decoder.norm", Hd, &m->dec_norm, s))) return fail(rc);

  const size_t B = cfg->max_batch;
  const size_t Ti = cfg->num_input_tokens, To = cfg->num_output_tokens;
  const size_t To_pad = round_up((int)To, 32), Ti_pad = round_up((int)Ti, 32);
  const size_t tok_pad = To_pad > Ti_pad ? To_pad : Ti_pad;
  const size_t Tm = To > Ti ? To : Ti;   // nbuf / ff / o serve the encoder (B*Ti rows) and the decoder (B*To rows)
  const int H = cfg->num_heads;
  if ((rc = m->pool.alloc(&m->x0, B * Ti * cfg->in_dim))) return fail(rc);
  if ((rc = m->pool.alloc(&m->h_enc, B * Ti * Hd))) return fail(rc);
  if ((rc = m->pool.alloc(&m->h_dec, B * To * Hd))) return fail(rc);
  if ((rc = m->pool.alloc(&m->nbuf, B * Tm * Hd))) return fail(rc);
  if ((rc = m->pool.alloc(&m->mem, B * Ti * Hd))) return fail(rc);
  if ((rc = m->pool.alloc(&m->ff, B * Tm * F))) return fail(rc);
  if ((rc = m->pool.alloc(&m->q, B * H * tok_pad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->k, B * H * tok_pad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->vt, B * H * m->dpv * tok_pad))) return fail(rc);
  if ((rc = m->pool.alloc(&m->o, B * Tm * Hd))) return fail(rc);
  m->splitk_ws_floats = (size_t)16 * B * Tm * (size_t)(F > 3 * Hd ? F : 3 * Hd);
  if ((rc = m->pool.alloc(&m->splitk_ws, m->splitk_ws_floats, false))) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { gill_set_error("mapper create: device sync failed"); return fail(-1); }
  *out = m;
  return 0;
}

extern "C" void gill_mapper_destroy(gill_mapper* h) { delete h; }

namespace {

struct MapperRun {
  gill_mapper* m;
  hipStream_t s;

  // y = act(A . W^T + b [+ resid_f32]) ; out fp32 or bf16
  int linear(const bf16_t* A, int M, const bf16_t* W, const float* b, int N, int K, const float* resid, int act, void* out,
             bool out_f32) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.K1 = K; g.A = A; g.lda = K; g.W = W; g.bias = b;
    g.resid = resid; g.ldr = N; g.resid_f32 = 1;
    g.act = act; g.out_mode = out_f32 ? OUT_F32 : OUT_BF16; g.C = out; g.ldc = N;
    g.splitk = gemm_pick_splitk(M, N, K, act);
    if ((size_t)g.splitk * M * N > m->splitk_ws_floats) g.splitk = 1;
    g.ws = m->splitk_ws;
    return gemm_launch(g, s);
  }
  // scatter a projection into the head-major attention operands; seg_base: 0 = q(,k,v) ; 1 = k,v only
  int qkv(const bf16_t* A, int B, int ntok, const bf16_t* W, const float* b, int nseg, int seg_base, int npad_q,
          int npad_kv) {
    const int Hd = m->cfg.hidden_dim;
    GemmArgs g;
    g.M = B * ntok; g.N = nseg * Hd; g.K = Hd; g.K1 = Hd; g.A = A; g.lda = Hd; g.W = W; g.bias = b;
    g.out_mode = OUT_QKV; g.Cq = m->q; g.Ck = m->k; g.Cvt = m->vt;
    g.heads = m->cfg.num_heads; g.dp = m->dp; g.dpv = m->dpv; g.ntok = ntok;
    g.ntok_pad_q = npad_q; g.ntok_pad_kv = npad_kv; g.seg_base = seg_base;
    g.qscale = 1.4426950408889634f / sqrtf((float)m->dp);
    return gemm_launch(g, s);
  }
  int attend(int B, int nq, int nkv) {
    AttnArgs a;
    a.Q = m->q; a.K = m->k; a.Vt = m->vt; a.O = m->o;
    a.B = B; a.H = m->cfg.num_heads; a.nq = nq; a.nkv = nkv;
    a.nq_pad = round_up(nq, 32); a.nkv_pad = round_up(nkv, 32);
    a.dp = m->dp; a.dpv = m->dpv; a.ldo = m->cfg.hidden_dim;
    a.scale = 1.0f / sqrtf((float)m->dp);
    return attention_launch(a, s);
  }
  // pre-LN self-attention sub-block: h += out_proj(attn(LN(h)))
  int self_attn_block(float* h, int B, int ntok, const NormW& n, const MhaW& w) {
    const int Hd = m->cfg.hidden_dim;
    GILL_TRY(layernorm_launch(h, 1, n.g, n.b, m->nbuf, B * ntok, Hd, 1e-5f, s));
    const int npad = round_up(ntok, 32);
    GILL_TRY(qkv(m->nbuf, B, ntok, w.in_proj.w, w.in_proj.b, 3, 0, npad, npad));
    GILL_TRY(attend(B, ntok, ntok));
    return linear(m->o, B * ntok, w.out_proj.w, w.out_proj.b, Hd, Hd, h, ACT_NONE, h, true);
  }
  int ff_block(float* h, int rows, const NormW& n, const LinearW& l1, const LinearW& l2) {
    const int Hd = m->cfg.hidden_dim, F = m->cfg.ffn_dim;
    GILL_TRY(layernorm_launch(h, 1, n.g, n.b, m->nbuf, rows, Hd, 1e-5f, s));
    GILL_TRY(linear(m->nbuf, rows, l1.w, l1.b, F, Hd, nullptr, ACT_RELU, m->ff, false));
    return linear(m->ff, rows, l2.w, l2.b, Hd, F, h, ACT_NONE, h, true);
  }
};

}  // namespace

extern "C" int gill_mapper_forward(gill_mapper* m, const void* x_bf16, const void* input_embs_bf16, int B, int Be,
                                   float* out, void* stream) {
  GILL_REQUIRE(m && x_bf16 && out, "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "batch exceeds max_batch of the mapper handle");
  GILL_REQUIRE(input_embs_bf16 == nullptr || Be == 1 || Be == B, "input_embs batch must be 1 or B");
  hipStream_t s = (hipStream_t)stream;
  MapperRun r{m, s};
  const gill_mapper_config& c = m->cfg;
  const int Ti = c.num_input_tokens, To = c.num_output_tokens, Hd = c.hidden_dim;
  // x = x + input_embs (layers.py:31-32); bf16 operand for the first MFMA GEMM
  const bf16_t* x0 = (const bf16_t*)x_bf16;
  if (input_embs_bf16) {
    const int64_t n = (int64_t)B * Ti * c.in_dim;
    const int64_t period = (Be == 1) ? (int64_t)Ti * c.in_dim : n;
    GILL_TRY(add_cast_launch(x_bf16, 0, input_embs_bf16, 0, n, period, m->x0, s));
    x0 = m->x0;
  }
  // fc (layers.py:42) -> fp32 encoder stream
  GILL_TRY(r.linear(x0, B * Ti, m->fc.w, m->fc.b, Hd, c.in_dim, nullptr, ACT_NONE, m->h_enc, true));
  for (const EncLayer& L : m->enc) {
    GILL_TRY(r.self_attn_block(m->h_enc, B, Ti, L.n1, L.sa));
    GILL_TRY(r.ff_block(m->h_enc, B * Ti, L.n2, L.l1, L.l2));
  }
  GILL_TRY(layernorm_launch(m->h_enc, 1, m->enc_norm.g, m->enc_norm.b, m->mem, B * Ti, Hd, 1e-5f, s));
  // tgt = query_embs.repeat(B,1,1) (layers.py:43)
  for (int b = 0; b < B; ++b)
    GILL_CHECK_HIP(hipMemcpyAsync(m->h_dec + (size_t)b * To * Hd, m->query, sizeof(float) * To * Hd,
                                  hipMemcpyDeviceToDevice, s));
  const int To_pad = round_up(To, 32), Ti_pad = round_up(Ti, 32);
  for (const DecLayer& L : m->dec) {
    GILL_TRY(r.self_attn_block(m->h_dec, B, To, L.n1, L.sa));
    // cross attention: q from the target stream, k/v from the encoder memory
    GILL_TRY(layernorm_launch(m->h_dec, 1, L.n2.g, L.n2.b, m->nbuf, B * To, Hd, 1e-5f, s));
    GILL_TRY(r.qkv(m->nbuf, B, To, L.ca.in_proj.w, L.ca.in_proj.b, 1, 0, To_pad, Ti_pad));
    GILL_TRY(r.qkv(m->mem, B, Ti, L.ca.in_proj.w + (size_t)Hd * Hd, L.ca.in_proj.b + Hd, 2, 1, To_pad, Ti_pad));
    GILL_TRY(r.attend(B, To, Ti));
    GILL_TRY(r.linear(m->o, B * To, L.ca.out_proj.w, L.ca.out_proj.b, Hd, Hd, m->h_dec, ACT_NONE, m->h_dec, true));
    GILL_TRY(r.ff_block(m->h_dec, B * To, L.n3, L.l1, L.l2));
  }
  GILL_TRY(layernorm_launch(m->h_dec, 1, m->dec_norm.g, m->dec_norm.b, m->nbuf, B * To, Hd, 1e-5f, s));
  // model (layers.py:44)
  return r.linear(m->nbuf, B * To, m->model.w, m->model.b, c.out_dim, Hd, nullptr, ACT_NONE, out, true);
}
