// Stage 1 — frozen OPT decoder (transformers OPTForCausalLM as called at gill/models.py:363-365 and :465).
//
// Pre-LN decoder layer (do_layer_norm_before=True: every OPT size but 350m):
//   h += out_proj(causal_attn(q,k,v = *_proj(LN(h))))         q scaled by head_dim^-0.5
//   h += fc2(relu(fc1(LN(h))))
// learned positions with the +2 offset, final LayerNorm; hidden_states[-1] is post-final-LN.
// The residual stream is fp32; GEMMs are bf16 MFMA with fp32 accumulation, split-K so that the
// skinny (M = B*T) weight-streaming GEMMs cover all 256 CUs; attention is the shared flash kernel.
#include "engine_util.h"

namespace {
struct OptLayer {
  bf16_t* wqkv = nullptr; float* bqkv = nullptr;   // [3D][D] rows: q | k | v
  bf16_t* wo = nullptr; float* bo = nullptr;
  bf16_t* w1 = nullptr; float* b1 = nullptr;
  bf16_t* w2 = nullptr; float* b2 = nullptr;
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
  int blk_qkv = 0, blk_o = 0, blk_1 = 0, blk_2 = 0;   // matrix stored 64 x 64-blocked (gemm_stream64_weights: the STREAM64 layout)
};
}  // namespace

struct gill_opt {
  gill_opt_config cfg;
  DevPool pool;
  bf16_t* embed = nullptr;      // [vocab][D]
  bf16_t* lm_head = nullptr;    // tied to embed unless lm_head.weight was supplied
  bf16_t* pos = nullptr;        // [max_positions+2][D]
  float *lnfg = nullptr, *lnfb = nullptr;
  std::vector<OptLayer> layers;
  int dp = 0, dpv = 0;
  // workspace
  float* h = nullptr;       // [B*T][D]
  bf16_t* nbuf = nullptr;   // [B*T][D]
  bf16_t* ff = nullptr;     // [B*T][ffn]
  bf16_t *q = nullptr, *k = nullptr, *vt = nullptr, *o = nullptr;
  bf16_t* emb_tmp = nullptr;  // [B*T][D]
  float* gath = nullptr;      // [B*8][D]
  bf16_t* last_bf = nullptr;  // [8][D]
  int32_t* idx_dev = nullptr; // [B*8 + B*8]
  float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
  // KV cache of gill_opt_forward_cached (allocated at its first call): per layer K [B][H][Tcap][dp], Vt [B][H][dpv][Tcap]
  std::vector<bf16_t*> kcache, vcache;
  int cache_tcap = 0;
};

// out[row][:] = bf16->f32(emb[row][:]) + pos[t + off][:]
__global__ __launch_bounds__(256) void opt_add_pos_kernel(const bf16_t* __restrict__ emb, const bf16_t* __restrict__ pos,
                                                          int pos_offset, int T, int D, float* __restrict__ out) {
  const int row = blockIdx.x;
  const int t = row % T;
  const bf16_t* e = emb + (size_t)row * D;
  const bf16_t* pe = pos + (size_t)(t + pos_offset) * D;
  float* o = out + (size_t)row * D;
  for (int c = threadIdx.x * 2; c < D; c += blockDim.x * 2) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(e + c);
    const uint32_t v = *reinterpret_cast<const uint32_t*>(pe + c);
    *reinterpret_cast<float2*>(o + c) = make_float2(bf2f((bf16_t)(u & 0xffff)) + bf2f((bf16_t)(v & 0xffff)),
                                                    bf2f((bf16_t)(u >> 16)) + bf2f((bf16_t)(v >> 16)));
  }
}

// REDUCE + RESIDUAL + LAYERNORM.  The narrow GEMMs of a layer (out_proj, fc2: N = D) run split-K; their reducer is also the natural place
// for the LayerNorm that follows them (the next sub-block's pre-LN): one workgroup per row adds the row's fp32 partials in split order, the
// bias and the fp32 residual stream (the order of gemm.hip's reducer epilogue), stores the new stream row, and — the row being complete in its
// registers — normalises it (two-pass variance like layernorm_kernel) into the bf16 operand of the next GEMM.  Two launches per layer gone
// (the LayerNorm passes) and the stream row is not re-read.  Fixed summation order: bit-repeatable.
template <int VPT>      // float4 vectors per thread: D <= 1024 * VPT
__global__ __launch_bounds__(256) void opt_reduce_ln_kernel(const float* __restrict__ ws, int sk, int M, int D, const float* __restrict__ bias,
                                                            float* __restrict__ h, const float* __restrict__ g, const float* __restrict__ b,
                                                            bf16_t* __restrict__ nb, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, nv = D >> 2;
  const size_t slice = (size_t)M * D;
  // every load of the row goes out before the first addition (slices four at a time, clamped addresses: no branch between a load and its use,
  // so the compiler does not fence them one by one): the kernel is one memory round trip deep, not sk + 2
  float4 v[VPT], bb[VPT], rr[VPT];
  float4 q[VPT][4];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int idx = (tid + i * 256 < nv) ? tid + i * 256 : 0;
    const int c = idx * 4;
    const float* p = ws + (size_t)row * D + c;
#pragma unroll
    for (int z = 0; z < 4; ++z) q[i][z] = *reinterpret_cast<const float4*>(p + (size_t)(z < sk ? z : 0) * slice);
    bb[i] = *reinterpret_cast<const float4*>(bias + c);
    rr[i] = *reinterpret_cast<const float4*>(h + (size_t)row * D + c);
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (tid + i * 256) * 4;
    const bool ok = tid + i * 256 < nv;
    float4 a = q[i][0];
#pragma unroll
    for (int z = 1; z < 4; ++z)
      if (z < sk) { a.x += q[i][z].x; a.y += q[i][z].y; a.z += q[i][z].z; a.w += q[i][z].w; }
    for (int z = 4; z < sk; ++z) {      // (more than four slices: not the OPT shapes)
      const float4 e = *reinterpret_cast<const float4*>(ws + (size_t)row * D + (ok ? c : 0) + (size_t)z * slice);
      a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
    }
    a.x = a.x + bb[i].x + rr[i].x; a.y = a.y + bb[i].y + rr[i].y; a.z = a.z + bb[i].z + rr[i].z; a.w = a.w + bb[i].w + rr[i].w;
    if (ok) *reinterpret_cast<float4*>(h + (size_t)row * D + c) = a;
    v[i] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto block_sum = [&](float x) -> float {      // fixed order: lanes (xor tree), then the four waves in index order
    x = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = block_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    if (tid + i * 256 < nv) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = rsqrtf(block_sum(ss) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (tid + i * 256) * 4;
    if (tid + i * 256 < nv) {
      const float4 gg = *reinterpret_cast<const float4*>(g + c);
      const float4 be = *reinterpret_cast<const float4*>(b + c);
      uint2 o;
      o.x = pack_bf2((v[i].x - mean) * rstd * gg.x + be.x, (v[i].y - mean) * rstd * gg.y + be.y);
      o.y = pack_bf2((v[i].z - mean) * rstd * gg.z + be.z, (v[i].w - mean) * rstd * gg.w + be.w);
      *reinterpret_cast<uint2*>(nb + (size_t)row * D + c) = o;
    }
  }
}
static int opt_reduce_ln_launch(const float* ws, int sk, int M, int D, const float* bias, float* h, const float* g, const float* b, bf16_t* nb,
                                float eps, hipStream_t s) {
  GILL_REQUIRE(D % 4 == 0 && D <= 8192 && sk >= 1, "reduce + LayerNorm: D must be a multiple of 4, at most 8192");
  if (D <= 1024) hipLaunchKernelGGL((opt_reduce_ln_kernel<1>), dim3(M), dim3(256), 0, s, ws, sk, M, D, bias, h, g, b, nb, eps);
  else if (D <= 4096) hipLaunchKernelGGL((opt_reduce_ln_kernel<4>), dim3(M), dim3(256), 0, s, ws, sk, M, D, bias, h, g, b, nb, eps);
  else hipLaunchKernelGGL((opt_reduce_ln_kernel<8>), dim3(M), dim3(256), 0, s, ws, sk, M, D, bias, h, g, b, nb, eps);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gill_opt_create(gill_opt** out, const gill_opt_config* cfg, const gill_tensor* weights, int n_weights) {
  GILL_REQUIRE(out && cfg && weights, "null argument");
  const int D = cfg->hidden_size, F = cfg->ffn_dim, H = cfg->num_heads;
  GILL_REQUIRE(D % 64 == 0 && F % 64 == 0 && H > 0 && D % H == 0, "OPT dims must be multiples of 64");
  const int hd = D / H;
  GILL_REQUIRE(attn_padded_dim(hd) == hd, "OPT head dim must be one of 48/64/80/128/160");
  GILL_REQUIRE(cfg->max_batch > 0 && cfg->max_seq > 0 && cfg->max_seq <= cfg->max_positions, "bad workspace sizing");
  gill_opt* m = new gill_opt();
  m->cfg = *cfg;
  m->dp = hd; m->dpv = round_up(hd, 32);
  WeightTable wt(weights, n_weights);
  hipStream_t s = nullptr;
  int rc = 0;
  auto fail = [&](int r) { delete m; return r; };
  const std::string dec = "model.decoder.";
  if ((rc = load_bf16(wt, m->pool, dec + "embed_tokens.weight", (int64_t)cfg->vocab_size * D, &m->embed, s))) return fail(rc);
  if ((rc = load_bf16(wt, m->pool, dec + "embed_positions.weight", (int64_t)(cfg->max_positions + 2) * D, &m->pos, s)))
    return fail(rc);
  if (wt.find("lm_head.weight")) {
    if ((rc = load_bf16(wt, m->pool, "lm_head.weight", (int64_t)cfg->vocab_size * D, &m->lm_head, s))) return fail(rc);
  } else {
    m->lm_head = m->embed;
  }
  if ((rc = load_f32(wt, m->pool, dec + "final_layer_norm.weight", D, &m->lnfg, s))) return fail(rc);
  if ((rc = load_f32(wt, m->pool, dec + "final_layer_norm.bias", D, &m->lnfb, s))) return fail(rc);
  m->layers.resize(cfg->num_layers);
  for (int i = 0; i < cfg->num_layers; ++i) {
    OptLayer& L = m->layers[i];
    const std::string p = dec + "layers." + std::to_string(i) + ".";
    if ((rc = m->pool.alloc(&L.wqkv, (size_t)3 * D * D, false))) return fail(rc);
    if ((rc = m->pool.alloc(&L.bqkv, (size_t)3 * D, false))) return fail(rc);
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    L.blk_qkv = gemm_stream64_weights(3 * D, D); L.blk_o = gemm_stream64_weights(D, D);
    L.blk_1 = gemm_stream64_weights(F, D); L.blk_2 = gemm_stream64_weights(D, F);
    // (blocked or row-major, a [D][D] third of the stacked q | k | v matrix is a contiguous range of whole 64-row blocks)
    auto load_w = [&](const std::string& name, int N, int K, int blk, bf16_t* dst) -> int {
      const gill_tensor* t;
      int r = wt.get(name, (int64_t)N * K, &t);
      if (r) return r;
      return blk ? convert_to_bf16_blk64_launch(t->data, t->dtype, N, K, dst, s) : convert_to_bf16_launch(t->data, t->dtype, (int64_t)N * K, dst, s);
    };
    for (int j = 0; j < 3; ++j) {
      const gill_tensor* t;
      if ((rc = load_w(p + "self_attn." + names[j] + ".weight", D, D, L.blk_qkv, L.wqkv + (size_t)j * D * D))) return fail(rc);
      if ((rc = wt.get(p + "self_attn." + names[j] + ".bias", D, &t))) return fail(rc);
      if ((rc = convert_to_f32_launch(t->data, t->dtype, D, L.bqkv + (size_t)j * D, s))) return fail(rc);
    }
    if ((rc = m->pool.alloc(&L.wo, (size_t)D * D, false))) return fail(rc);
    if ((rc = m->pool.alloc(&L.w1, (size_t)F * D, false))) return fail(rc);
    if ((rc = m->pool.alloc(&L.w2, (size_t)D * F, false))) return fail(rc);
    if ((rc = load_w(p + "self_attn.out_proj.weight", D, D, L.blk_o, L.wo))) return fail(rc);
    if ((rc = load_w(p + "fc1.weight", F, D, L.blk_1, L.w1))) return fail(rc);
    if ((rc = load_w(p + "fc2.weight", D, F, L.blk_2, L.w2))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "self_attn.out_proj.bias", D, &L.bo, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "fc1.bias", F, &L.b1, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "fc2.bias", D, &L.b2, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "self_attn_layer_norm.weight", D, &L.ln1g, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "self_attn_layer_norm.bias", D, &L.ln1b, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "final_layer_norm.weight", D, &L.ln2g, s))) return fail(rc);
    if ((rc = load_f32(wt, m->pool, p + "final_layer_norm.bias", D, &L.ln2b, s))) return fail(rc);
  }
  const size_t R = (size_t)cfg->max_batch * cfg->max_seq;
  const size_t Tpad = round_up(cfg->max_seq, 32);
  if ((rc = m->pool.alloc(&m->h, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->nbuf, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->ff, R * F))) return fail(rc);
  if ((rc = m->pool.alloc(&m->q, (size_t)cfg->max_batch * H * Tpad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->k, (size_t)cfg->max_batch * H * Tpad * m->dp))) return fail(rc);
  if ((rc = m->pool.alloc(&m->vt, (size_t)cfg->max_batch * H * m->dpv * Tpad))) return fail(rc);
  if ((rc = m->pool.alloc(&m->o, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->emb_tmp, R * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->gath, (size_t)cfg->max_batch * 64 * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->last_bf, (size_t)8 * D))) return fail(rc);
  if ((rc = m->pool.alloc(&m->idx_dev, (size_t)cfg->max_batch * 64 * 2))) return fail(rc);
  m->splitk_ws_floats = (size_t)16 * R * (size_t)(F > 3 * D ? F : 3 * D);
  if ((rc = m->pool.alloc(&m->splitk_ws, m->splitk_ws_floats, false))) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { gill_set_error("opt create: device sync failed"); return fail(-1); }
  *out = m;
  return 0;
}

extern "C" void gill_opt_destroy(gill_opt* h) { delete h; }

extern "C" int gill_opt_embed(gill_opt* m, const int64_t* ids, int n, void* out_bf16, void* stream) {
  GILL_REQUIRE(m && ids && out_bf16 && n > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  // embedding rows are bf16 already: the lookup is a pure row copy (no allocation, no sync: both pointers are the caller's
  // device memory and the call is stream-ordered like every other entry point)
  GILL_TRY(embed_rows_bf16_launch(ids, m->embed, m->cfg.vocab_size, n, m->cfg.hidden_size, (bf16_t*)out_bf16, s));
  return 0;
}

namespace {
struct OptRun {
  gill_opt* m;
  hipStream_t s;
  // ln_g / ln_b / ln_out: the LayerNorm that consumes the result (in-place residual GEMMs into the fp32 stream only: out == resid, N == D).
  // Split-K launches hand it to the reducer (opt_reduce_ln_kernel); unsplit ones run the stand-alone pass.
  int linear(const bf16_t* A, int M, const bf16_t* W, int blk, const float* b, int N, int K, const float* resid, int act, void* out,
             bool out_f32, const float* ln_g = nullptr, const float* ln_b = nullptr, bf16_t* ln_out = nullptr) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.K1 = K; g.A = A; g.lda = K; g.W = W; g.bias = b;
    g.resid = resid; g.ldr = N; g.resid_f32 = 1;
    g.act = act; g.out_mode = out_f32 ? OUT_F32 : OUT_BF16; g.C = out; g.ldc = N;
    g.w_blk64 = blk;
    g.splitk = blk ? gemm_pick_splitk_blk64(M, N, K) : gemm_pick_splitk(M, N, K, act);
    if ((size_t)g.splitk * M * N > m->splitk_ws_floats) g.splitk = 1;
    g.ws = m->splitk_ws;
    if (ln_out && g.splitk > 1 && out_f32 && resid == (const float*)out && act == ACT_NONE && b && N % 4 == 0 && N <= 8192) {
      g.partials_only = 1;
      GILL_TRY(gemm_launch(g, s));
      return opt_reduce_ln_launch(m->splitk_ws, g.splitk, M, N, b, (float*)out, ln_g, ln_b, ln_out, 1e-5f, s);
    }
    GILL_TRY(gemm_launch(g, s));
    if (ln_out) GILL_TRY(layernorm_launch(out, 1, ln_g, ln_b, ln_out, M, N, 1e-5f, s));
    return 0;
  }
  // layers over the fp32 stream m->h (B*T rows).  past < 0: plain causal forward over T tokens.  past >= 0: the T rows are
  // the tokens past .. past+T-1 of each sequence; their K/V are appended to the handle's cache and they attend to it.
  int run_layers(int B, int T, int past = -1) {
    const gill_opt_config& c = m->cfg;
    const int D = c.hidden_size, F = c.ffn_dim, R = B * T;
    const int Tpad = round_up(T, 32);
    const bool cached = past >= 0;
    const int kvpad = cached ? m->cache_tcap : Tpad;
    int li = 0;
    const int nl = (int)m->layers.size();
    for (const OptLayer& L : m->layers) {
      bf16_t* kbuf = cached ? m->kcache[li] : m->k;
      bf16_t* vbuf = cached ? m->vcache[li] : m->vt;
      const OptLayer* next = li + 1 < nl ? &m->layers[li + 1] : nullptr;
      ++li;
      // (layers past the first: the previous layer's fc2 launch has normalised the stream into nbuf already)
      if (li == 1) GILL_TRY(layernorm_launch(m->h, 1, L.ln1g, L.ln1b, m->nbuf, R, D, 1e-5f, s));
      {
        GemmArgs g;
        g.M = R; g.N = 3 * D; g.K = D; g.K1 = D; g.A = m->nbuf; g.lda = D; g.W = L.wqkv; g.bias = L.bqkv;
        g.out_mode = OUT_QKV; g.Cq = m->q; g.Ck = kbuf; g.Cvt = vbuf;
        g.heads = c.num_heads; g.dp = m->dp; g.dpv = m->dpv; g.ntok = T; g.ntok_pad_q = Tpad; g.ntok_pad_kv = kvpad;
        g.seg_base = 0; g.kv_tok_offset = cached ? past : 0;
        g.qscale = 1.4426950408889634f / sqrtf((float)m->dp);   // HF scales q by head_dim^-0.5
        g.w_blk64 = L.blk_qkv;
        g.splitk = L.blk_qkv ? gemm_pick_splitk_blk64(R, 3 * D, D) : gemm_pick_splitk(R, 3 * D, D, 0);
        if ((size_t)g.splitk * R * 3 * D > m->splitk_ws_floats) g.splitk = 1;
        g.ws = m->splitk_ws;
        GILL_TRY(gemm_launch(g, s));
      }
      {
        AttnArgs a;
        a.Q = m->q; a.K = kbuf; a.Vt = vbuf; a.O = m->o;
        a.B = B; a.H = c.num_heads; a.nq = T; a.nkv = cached ? past + T : T; a.nq_pad = Tpad; a.nkv_pad = kvpad;
        a.dp = m->dp; a.dpv = m->dpv; a.ldo = D; a.scale = 1.0f / sqrtf((float)m->dp); a.causal = 1;
        GILL_TRY(attention_launch(a, s));
      }
      GILL_TRY(linear(m->o, R, L.wo, L.blk_o, L.bo, D, D, m->h, ACT_NONE, m->h, true, L.ln2g, L.ln2b, m->nbuf));
      GILL_TRY(linear(m->nbuf, R, L.w1, L.blk_1, L.b1, F, D, nullptr, ACT_RELU, m->ff, false));
      if (next) GILL_TRY(linear(m->ff, R, L.w2, L.blk_2, L.b2, D, F, m->h, ACT_NONE, m->h, true, next->ln1g, next->ln1b, m->nbuf));
      else GILL_TRY(linear(m->ff, R, L.w2, L.blk_2, L.b2, D, F, m->h, ACT_NONE, m->h, true));
    }
    return 0;
  }
};
}  // namespace

extern "C" int gill_opt_forward(gill_opt* m, const void* inputs_embeds_bf16, int B, int T, float* hidden_out,
                                void* stream) {
  GILL_REQUIRE(m && inputs_embeds_bf16, "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch && T >= 1 && T <= m->cfg.max_seq, "B/T exceed the handle's workspace");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->cfg.hidden_size;
  hipLaunchKernelGGL(opt_add_pos_kernel, dim3(B * T), dim3(256), 0, s, (const bf16_t*)inputs_embeds_bf16, m->pos, 2, T, D,
                     m->h);
  GILL_CHECK_HIP(hipGetLastError());
  OptRun r{m, s};
  GILL_TRY(r.run_layers(B, T));
  if (hidden_out) GILL_TRY(layernorm_f32out_launch(m->h, 1, m->lnfg, m->lnfb, hidden_out, B * T, D, 1e-5f, s));
  return 0;
}

extern "C" int gill_opt_forward_cached(gill_opt* m, const void* inputs_embeds_bf16, int B, int T_new, int past_len,
                                       float* hidden_out, void* stream) {
  GILL_REQUIRE(m && inputs_embeds_bf16 && hidden_out, "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch && T_new >= 1 && past_len >= 0 && past_len + T_new <= m->cfg.max_seq,
               "B / past_len + T_new exceed the handle's workspace");
  GILL_REQUIRE(past_len + T_new + 2 <= m->cfg.max_positions + 2, "sequence longer than the position table");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->cfg.hidden_size;
  if (m->kcache.empty()) {
    // zero-filled: rows beyond the valid length are masked in attention, but must stay finite (0 * NaN would poison PV)
    m->cache_tcap = round_up(m->cfg.max_seq, 32);
    const size_t kn = (size_t)m->cfg.max_batch * m->cfg.num_heads * m->cache_tcap * m->dp;
    const size_t vn = (size_t)m->cfg.max_batch * m->cfg.num_heads * m->dpv * m->cache_tcap;
    m->kcache.assign(m->layers.size(), nullptr); m->vcache.assign(m->layers.size(), nullptr);
    for (size_t l = 0; l < m->layers.size(); ++l) {
      GILL_TRY(m->pool.alloc(&m->kcache[l], kn, true));
      GILL_TRY(m->pool.alloc(&m->vcache[l], vn, true));
    }
  }
  hipLaunchKernelGGL(opt_add_pos_kernel, dim3(B * T_new), dim3(256), 0, s, (const bf16_t*)inputs_embeds_bf16, m->pos,
                     2 + past_len, T_new, D, m->h);
  GILL_CHECK_HIP(hipGetLastError());
  OptRun r{m, s};
  GILL_TRY(r.run_layers(B, T_new, past_len));
  GILL_TRY(layernorm_f32out_launch(m->h, 1, m->lnfg, m->lnfb, hidden_out, B * T_new, D, 1e-5f, s));
  return 0;
}

extern "C" int gill_opt_img_hidden(gill_opt* m, const int64_t* ids, const int32_t* last_idx_host, int B, int T,
                                   int num_tokens, void* raw_out_bf16, void* emb_out_bf16, void* stream) {
  GILL_REQUIRE(m && ids && last_idx_host && raw_out_bf16, "null argument");
  GILL_REQUIRE(B >= 1 && B <= m->cfg.max_batch && T >= 1 && T <= m->cfg.max_seq, "B/T exceed the handle's workspace");
  GILL_REQUIRE(num_tokens >= 1 && num_tokens <= 64, "num_tokens out of range");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->cfg.hidden_size;
  std::vector<int32_t> idx((size_t)B * num_tokens);
  for (int b = 0; b < B; ++b) {
    GILL_REQUIRE(last_idx_host[b] - num_tokens + 1 >= 0 && last_idx_host[b] < T, "last_idx out of range");
    for (int j = 0; j < num_tokens; ++j) idx[(size_t)b * num_tokens + j] = b * T + last_idx_host[b] - num_tokens + 1 + j;
  }
  GILL_CHECK_HIP(hipMemcpyAsync(m->idx_dev, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));  // idx is a stack-lifetime host buffer
  // input_embs = input_embeddings(labels)  (models.py:180), bf16 rows
  GILL_TRY(embed_tokens_launch(ids, m->embed, m->cfg.vocab_size, nullptr, 0, B, T, D, m->h, s));
  GILL_TRY(cast_f32_to_bf16_launch(m->h, m->emb_tmp, (int64_t)B * T * D, s));
  if (emb_out_bf16) GILL_TRY(gather_rows_launch(m->emb_tmp, 0, m->idx_dev, B * num_tokens, D, emb_out_bf16, 0, s));
  // + learned positions -> fp32 stream
  GILL_TRY(embed_tokens_launch(ids, m->embed, m->cfg.vocab_size, m->pos, 2, B, T, D, m->h, s));
  OptRun r{m, s};
  GILL_TRY(r.run_layers(B, T));
  // final LN only on the rows that are read (models.py:384)
  GILL_TRY(gather_rows_launch(m->h, 1, m->idx_dev, B * num_tokens, D, m->gath, 1, s));
  GILL_TRY(layernorm_launch(m->gath, 1, m->lnfg, m->lnfb, (bf16_t*)raw_out_bf16, B * num_tokens, D, 1e-5f, s));
  return 0;
}

extern "C" int gill_opt_last_logits(gill_opt* m, const float* hidden, int B, int T, float* logits_out, void* stream) {
  GILL_REQUIRE(m && hidden && logits_out, "null argument");
  GILL_REQUIRE(B >= 1 && B <= 8, "last_logits supports B <= 8");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->cfg.hidden_size;
  for (int b = 0; b < B; ++b)
    GILL_TRY(cast_f32_to_bf16_launch(hidden + ((size_t)b * T + (T - 1)) * D, m->last_bf + (size_t)b * D, D, s));
  return skinny_gemm_launch(m->last_bf, m->lm_head, B, m->cfg.vocab_size, D, logits_out, s);
}
