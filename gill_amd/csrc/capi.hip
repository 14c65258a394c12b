// extern "C" surface of libgill_amd (see include/gill_amd.h): error plumbing and the operator-level
// entry points.  The three stage engines export their own entry points from opt.hip / mapper.hip /
// unet.hip.
#include "../../include/gill_amd.h"
#include "ops.h"
#include "engine_util.h"
#include <stdlib.h>
#include <vector>

static thread_local std::string g_last_error;
void gill_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* gill_last_error(void) { return g_last_error.c_str(); }
extern "C" int gill_version(void) { return 100; }

// GILL_OP_REPEAT=n makes the operator entry points launch their kernel n times per call (tools/bench_ops.py: amortises
// the wrapper's allocation / re-layout so the kernel itself can be timed); default 1.
static int op_repeat() {
  const char* v = getenv("GILL_OP_REPEAT");     // read at every call: bench.py times a call at two repeat counts and takes the difference
  const int r = v ? (atoi(v) < 1 ? 1 : atoi(v)) : 1;
  return r;
}

// One grow-only split-K workspace for the operator entry points (they synchronise before they return and the reference's threading model is one
// request at a time — SURVEY 8b — so one buffer serves them all): a per-call hipMalloc / hipFree of up to 100 MB sat inside every timed call
// (profiles/r05_opt_stream64.md: it poisoned a probe).  Freed at process exit with the context.
static float* op_splitk_ws(size_t floats) {
  static float* ws = nullptr;
  static size_t cap = 0;
  if (floats > cap) {
    if (ws) (void)hipFree(ws);
    ws = nullptr; cap = 0;
    const size_t want = floats + (floats >> 2);
    if (hipMalloc((void**)&ws, sizeof(float) * want) != hipSuccess) { ws = nullptr; return nullptr; }
    cap = want;
  }
  return ws;
}

// splitk: 0 = the launcher's heuristic, n > 1 = forced.  splitk < 0: FORCE THE GENERAL ROW-MAJOR TILES — never the 64 x 64-blocked
// blocked copy that weight-streaming shapes (N * K >= 4 Mi; the STREAM64 tile up to 256 rows, the general tiles on the blocked layout above) otherwise take — with the heuristic split (-1) or -splitk ways (< -1), so that
// the operator tests can pin both paths on the same shape (the UNet and CLIP engines run such shapes on the general tiles).
extern "C" int gill_op_gemm(const void* A, const void* W, const float* bias, const void* resid_bf16, void* C, int M, int N,
                            int K, float alpha, int act, int out_f32, int splitk, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.K1 = K;
  g.A = (const bf16_t*)A; g.lda = K;
  g.W = (const bf16_t*)W;
  g.alpha = alpha; g.bias = bias;
  g.resid = resid_bf16; g.ldr = N; g.resid_f32 = 0;
  g.act = act;
  g.out_mode = out_f32 ? OUT_F32 : OUT_BF16;
  g.C = C; g.ldc = N;
  const bool row_major = splitk < 0;
  if (row_major) splitk = (splitk == -1) ? 0 : -splitk;
  // weight-streaming shapes run the way the OPT engine runs them: on a 64 x 64-blocked copy of W (gemm_launch picks the tile by M)
  DevBuf wblk;
  if (!row_major && act != ACT_GEGLU && gemm_stream64_weights(N, K)) {
    GILL_TRY(wblk.alloc(sizeof(bf16_t) * (size_t)N * K));
    GILL_TRY(convert_to_bf16_blk64_launch(W, 0, N, K, (bf16_t*)wblk.p, s));
    g.W = (const bf16_t*)wblk.p; g.w_blk64 = 1;
  }
  g.splitk = splitk > 0 ? splitk : (g.w_blk64 ? gemm_pick_splitk_blk64(M, N, K) : gemm_pick_splitk(M, N, K, act));
  if (g.splitk > 1) {
    g.ws = op_splitk_ws((size_t)g.splitk * M * N);
    GILL_REQUIRE(g.ws != nullptr, "split-K workspace allocation failed");
  }
  for (int r = 0; r < op_repeat(); ++r) GILL_TRY(gemm_launch(g, s));
  if (g.w_blk64) GILL_CHECK_HIP(hipStreamSynchronize(s));  // the blocked copy is freed on return
  return 0;
}

extern "C" int gill_op_geglu(const void* A, const void* W, const float* bias, void* C, int M, int inner, int K,
                             void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GILL_REQUIRE(inner % 64 == 0, "geglu: inner dim must be a multiple of 64");
  DevBuf wperm, bperm, idx;
  GILL_TRY(wperm.alloc(sizeof(bf16_t) * (size_t)2 * inner * K));
  GILL_TRY(bperm.alloc(sizeof(float) * (size_t)2 * inner));
  std::vector<int32_t> map = geglu_row_permutation(inner);
  GILL_TRY(idx.alloc(sizeof(int32_t) * map.size()));
  GILL_CHECK_HIP(hipMemcpyAsync(idx.p, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice, s));
  GILL_TRY(scatter_rows_bf16_launch((const bf16_t*)W, 2 * inner, K, (const int32_t*)idx.p, (bf16_t*)wperm.p, K, s));
  if (bias) GILL_TRY(permute_f32_launch(bias, (const int32_t*)idx.p, 2 * inner, (float*)bperm.p, s));
  GemmArgs g;
  g.M = M; g.N = 2 * inner; g.K = K; g.K1 = K;
  g.A = (const bf16_t*)A; g.lda = K;
  g.W = (const bf16_t*)wperm.p;
  g.bias = bias ? (const float*)bperm.p : nullptr;
  g.act = ACT_GEGLU; g.out_mode = OUT_BF16;
  g.C = C; g.ldc = inner;
  for (int r = 0; r < op_repeat(); ++r) GILL_TRY(gemm_launch(g, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gill_op_conv3x3(const void* x1, int C1, const void* x2, int C2, const float* w_oihw, const float* bias,
                               const float* rowvec, const void* resid, void* y, int B, int IH, int IW, int Cout,
                               int stride, int ups, int splitk, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C1 + C2;
  DevBuf wr, ws;
  GILL_TRY(wr.alloc(sizeof(bf16_t) * (size_t)Cout * 16 * Cin));     // 9 taps, or 4 classes x 4 taps
  // K order as the engines choose it
  int chunked = conv_k_chunked(IH * IW, Cin, Cout) ? 1 : 0;
  // ups = 1: the engines' form — four pre-summed 2x2-tap kernels on the source grid (GILL_CONV_UPS4 = 0, or a residual / row
  // vector, which that form does not take: the 9-tap gather over the upsampled grid)
  const char* u4 = getenv("GILL_CONV_UPS4");
  const bool ups4 = ups && stride == 1 && !rowvec && !resid && !(u4 && atoi(u4) == 0);
  if (ups4) { chunked = 0; GILL_TRY(conv_weight_relayout_ups4_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s)); }
  else if (chunked) GILL_TRY(conv_weight_relayout_chunked_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s));
  else GILL_TRY(conv_weight_relayout_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s));
  GemmArgs g;
  g.conv = 1;
  g.IH = IH; g.IW = IW; g.Cin = Cin; g.stride = stride; g.ups = ups4 ? 2 : ups;
  if (ups) { g.OH = 2 * IH; g.OW = 2 * IW; }
  else { g.OH = (IH + 2 - 3) / stride + 1; g.OW = (IW + 2 - 3) / stride + 1; }
  g.M = B * g.OH * g.OW; g.N = Cout; g.K = (ups4 ? 4 : 9) * Cin;
  g.A = (const bf16_t*)x1; g.A2 = (const bf16_t*)x2; g.K1 = C1;
  g.W = (const bf16_t*)wr.p; g.k_chunked = chunked;
  g.bias = bias;
  g.rowvec = rowvec; g.rows_per_batch = g.OH * g.OW; g.rowvec_bstride = Cout;
  g.resid = resid; g.ldr = Cout;
  g.C = y; g.ldc = Cout;
  g.splitk = splitk > 0 ? splitk : gemm_pick_splitk(g.M, g.N, g.K, 0);
  if (g.splitk > 1) {
    GILL_TRY(ws.alloc(sizeof(float) * (size_t)g.splitk * g.M * g.N));
    g.ws = (float*)ws.p;
  }
  for (int r = 0; r < op_repeat(); ++r) GILL_TRY(gemm_launch(g, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// 3x3 convolution + the GroupNorm (+ SiLU) that consumes its output, without a GroupNorm launch of its own where the geometry allows:
// y_raw (optional) = conv(x) + bias + rowvec[b] + resid, y_norm = [silu](GroupNorm(y_raw; groups, gamma, beta, eps)).
//   splitk >= 2: the split-K reduction also normalises (H * W in {64, 256}, Cout / groups a multiple of 4 dividing 80) — coop = 0: in the reducer launch
//                (gemm.hip "REDUCE + GROUPNORM"); coop = 1: inside the convolution's own launch (gemm.hip "COOP", EPI 7) when gemm_coop_ok() takes it
//                (B * H * W / 128 x Cout / 160 x splitk workgroups <= the device's CUs, Cout % 160 == 0), else as coop = 0;
//   splitk == 1: coop = 1: in the convolution's epilogue (COOP, EPI 6: H * W a multiple of the 128- / 256-row tile, <= 4096 pixels per sample,
//                grid <= CUs) — an error where the geometry does not allow it; coop = 0: the reference dataflow, conv with fused statistics +
//                groupnorm_apply_launch.
// ss_out (optional, splitk == 1 && coop == 1 only): the per-(sample, channel) scale | shift table [B][2][Cout] instead of / next to y_norm.
// For the operator tests.
extern "C" int gill_op_conv3x3_gn(const void* x, const float* w_oihw, const float* bias, const float* rowvec, const void* resid, const float* gamma,
                                  const float* beta, int groups, float eps, int silu, void* y_raw, void* y_norm, float* ss_out, int B, int H, int W,
                                  int Cin, int Cout, int splitk, int coop, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GILL_REQUIRE(x && w_oihw && gamma && beta && (y_norm || ss_out) && groups > 0 && Cout % groups == 0 && splitk >= 1, "bad argument");
  GILL_REQUIRE(ss_out == nullptr || (splitk == 1 && coop), "conv3x3_gn: the scale | shift table comes from the in-kernel finish only");
  DevBuf wr, ws, ctr, st, raw_tmp;
  GILL_TRY(wr.alloc(sizeof(bf16_t) * (size_t)Cout * 9 * Cin));
  GILL_TRY(conv_weight_relayout_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s));
  GemmArgs g;
  g.conv = 1; g.IH = H; g.IW = W; g.OH = H; g.OW = W; g.Cin = Cin; g.stride = 1;
  g.M = B * H * W; g.N = Cout; g.K = 9 * Cin;
  g.A = (const bf16_t*)x; g.K1 = Cin; g.W = (const bf16_t*)wr.p; g.bias = bias;
  g.rowvec = rowvec; g.rowvec_bstride = Cout;
  g.rows_per_batch = H * W; g.resid = resid; g.ldr = Cout; g.C = y_raw; g.ldc = Cout;
  g.splitk = splitk;
  if (splitk > 1) {
    GILL_TRY(ws.alloc(sizeof(float) * (size_t)splitk * g.M * g.N));
    g.ws = (float*)ws.p;
  }
  g.fn_Y = (bf16_t*)y_norm; g.fn_ss = ss_out; g.fn_gamma = gamma; g.fn_beta = beta; g.fn_eps = eps; g.fn_silu = silu; g.fn_cg = Cout / groups;
  const int ncnt = gemm_coop_counters(g);
  GILL_TRY(ctr.alloc(sizeof(unsigned) * (size_t)ncnt));
  // statistics bins as the engine's talloc() picks them (C / 64 channels, else one group), one partial per 16 rows at most
  const int sbin = (Cout % 64 == 0 && Cout / 64 >= 2) ? Cout / 64 : Cout / groups;
  if (splitk == 1) {
    GILL_REQUIRE(gemm_fused_gn_ok(Cout, sbin) && (H * W) % GN_SLAB_ROWS == 0, "conv3x3_gn: no fused statistics for this width / map size");
    GILL_TRY(st.alloc(sizeof(float) * (size_t)B * (H * W / GN_SLAB_ROWS_MIN) * (Cout / sbin) * 2));
    g.gn_stats = (float*)st.p; g.gn_groups = Cout / sbin; g.gn_cg = sbin;
  }
  if (coop) { g.coop_ctr = (unsigned*)ctr.p; g.coop_splitk = 1; }
  if (splitk == 1 && !coop) {
    // reference dataflow: the convolution files its statistics, a GroupNorm-apply launch normalises the rounded tensor
    g.fn_Y = nullptr; g.fn_ss = nullptr;
    if (!g.C) { GILL_TRY(raw_tmp.alloc(sizeof(bf16_t) * (size_t)g.M * g.N)); g.C = raw_tmp.p; }
    for (int r = 0; r < op_repeat(); ++r) {
      GILL_TRY(gemm_launch(g, s));
      GILL_TRY(groupnorm_apply_launch((const bf16_t*)g.C, Cout, nullptr, 0, B, H * W, groups, gamma, beta, eps, silu, (bf16_t*)y_norm, g.gn_stats, sbin, Cout,
                                      H * W / gemm_gn_slab_rows(g), nullptr, 0, 0, s, 0.f, nullptr, nullptr));
    }
    GILL_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
  }
  if (splitk == 1) GILL_REQUIRE(gemm_coop_ok(g), "conv3x3_gn: splitk == 1 with coop needs the in-kernel finish's geometry (gemm_coop_ok)");
  else GILL_REQUIRE(gemm_coop_ok(g) || gemm_fused_norm_ok(g), "conv3x3_gn: unsupported geometry (H * W in {64, 256}, Cout % 80 == 0, group width | 80)");
  for (int r = 0; r < op_repeat(); ++r) {
    GILL_CHECK_HIP(hipMemsetAsync(ctr.p, 0, sizeof(unsigned) * (size_t)ncnt, s));
    GILL_TRY(gemm_launch(g, s));
  }
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ResnetBlock2D's conv2 with the 1x1 conv_shortcut of the raw block input fused as extra K channels (the engine's c2f weights):
// y = conv3x3(x1 ++ x2) + bias + conv1x1(xs1 ++ xs2), one implicit GEMM with K = 9 (C1 + C2) + CS1 + CS2.  For the operator tests.
extern "C" int gill_op_conv3x3_shortcut(const void* x1, int C1, const void* x2, int C2, const float* w_oihw, const float* bias,
                                        const void* xs1, int CS1, const void* xs2, int CS2, const float* w_sc, void* y, int B, int IH,
                                        int IW, int Cout, int splitk, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C1 + C2, CS = CS1 + CS2, kf = 9 * Cin + CS;
  GILL_REQUIRE(x1 && w_oihw && xs1 && w_sc && y && CS % 64 == 0 && CS1 % 64 == 0, "bad argument");
  DevBuf wr, wsb, wf, idx, ws;
  GILL_TRY(wr.alloc(sizeof(bf16_t) * (size_t)Cout * 9 * Cin));
  GILL_TRY(wsb.alloc(sizeof(bf16_t) * (size_t)Cout * CS));
  GILL_TRY(wf.alloc(sizeof(bf16_t) * (size_t)Cout * kf));
  GILL_TRY(idx.alloc(sizeof(int32_t) * (size_t)Cout));
  const int chunked = conv_k_chunked(IH * IW, Cin, Cout) ? 1 : 0;
  if (chunked) GILL_TRY(conv_weight_relayout_chunked_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s));
  else GILL_TRY(conv_weight_relayout_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, (bf16_t*)wr.p, s));
  GILL_TRY(convert_to_bf16_launch(w_sc, GILL_DTYPE_F32, (int64_t)Cout * CS, (bf16_t*)wsb.p, s));
  std::vector<int32_t> ident(Cout);
  for (int i = 0; i < Cout; ++i) ident[i] = i;
  GILL_CHECK_HIP(hipMemcpyAsync(idx.p, ident.data(), sizeof(int32_t) * Cout, hipMemcpyHostToDevice, s));
  GILL_TRY(scatter_rows_bf16_launch((const bf16_t*)wr.p, Cout, 9 * Cin, (const int32_t*)idx.p, (bf16_t*)wf.p, kf, s));
  GILL_TRY(scatter_rows_bf16_launch((const bf16_t*)wsb.p, Cout, CS, (const int32_t*)idx.p, (bf16_t*)wf.p + 9 * Cin, kf, s));
  GemmArgs g;
  g.conv = 1; g.IH = IH; g.IW = IW; g.OH = IH; g.OW = IW; g.Cin = Cin; g.stride = 1; g.ups = 0;
  g.M = B * IH * IW; g.N = Cout; g.K = kf;
  g.A = (const bf16_t*)x1; g.A2 = (const bf16_t*)x2; g.K1 = C1;
  g.X1 = (const bf16_t*)xs1; g.X2 = (const bf16_t*)xs2; g.KX = CS; g.KX1 = CS1;
  g.W = (const bf16_t*)wf.p; g.k_chunked = chunked; g.bias = bias;
  g.rows_per_batch = IH * IW;
  g.C = y; g.ldc = Cout;
  g.splitk = splitk > 0 ? splitk : gemm_pick_splitk(g.M, g.N, g.K, 0);
  if (g.splitk > 1) {
    GILL_TRY(ws.alloc(sizeof(float) * (size_t)g.splitk * g.M * g.N));
    g.ws = (float*)ws.p;
  }
  for (int r = 0; r < op_repeat(); ++r) GILL_TRY(gemm_launch(g, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gill_op_attention(const void* q, const void* k, const void* v, void* o, int B, int H, int nq, int nkv, int d,
                                 float scale, int causal, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int dp = attn_padded_dim(d);
  GILL_REQUIRE(dp > 0, "attention: head dim > 160 unsupported");
  const int dpv = round_up(dp, 32);
  const int nq_pad = round_up(nq, 32), nkv_pad = round_up(nkv, 32);
  DevBuf Q, K, Vt, O;
  GILL_TRY(Q.alloc_zero(sizeof(bf16_t) * (size_t)B * H * nq_pad * dp, s));
  GILL_TRY(K.alloc_zero(sizeof(bf16_t) * (size_t)B * H * nkv_pad * dp, s));
  GILL_TRY(Vt.alloc_zero(sizeof(bf16_t) * (size_t)B * H * dpv * nkv_pad, s));
  GILL_TRY(O.alloc_zero(sizeof(bf16_t) * (size_t)B * nq * H * dp, s));
  GILL_TRY(pack_heads_launch((const bf16_t*)q, B, nq, H, d, nq_pad, dp, dpv, 0, (bf16_t*)Q.p, s, scale * 1.4426950408889634f));
  GILL_TRY(pack_heads_launch((const bf16_t*)k, B, nkv, H, d, nkv_pad, dp, dpv, 0, (bf16_t*)K.p, s));
  GILL_TRY(pack_heads_launch((const bf16_t*)v, B, nkv, H, d, nkv_pad, dp, dpv, 1, (bf16_t*)Vt.p, s));
  AttnArgs a;
  a.Q = (const bf16_t*)Q.p; a.K = (const bf16_t*)K.p; a.Vt = (const bf16_t*)Vt.p; a.O = (bf16_t*)O.p;
  a.B = B; a.H = H; a.nq = nq; a.nkv = nkv; a.nq_pad = nq_pad; a.nkv_pad = nkv_pad; a.dp = dp; a.dpv = dpv;
  a.ldo = H * dp; a.scale = scale; a.causal = causal; a.d = d;
  for (int r = 0; r < op_repeat(); ++r) GILL_TRY(attention_launch(a, s));
  GILL_TRY(unpad_heads_launch((const bf16_t*)O.p, (int64_t)B * nq, H, d, dp, (bf16_t*)o, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gill_op_layernorm(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, int rows,
                                 int C, float eps, void* stream) {
  return layernorm_launch(x, x_f32, gamma, beta, (bf16_t*)y_bf16, rows, C, eps, (hipStream_t)stream);
}

extern "C" int gill_op_groupnorm(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups,
                                 const float* gamma, const float* beta, float eps, int silu, void* y, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  DevBuf stats;
  GILL_TRY(stats.alloc(sizeof(float) * groupnorm_stats_floats(B, HW, groups)));
  GILL_TRY(groupnorm_launch((const bf16_t*)x1, C1, (const bf16_t*)x2, C2, B, HW, groups, gamma, beta, eps, silu,
                            (bf16_t*)y, (float*)stats.p, s));
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// fp8 convolution (conv_fp8.hip) on bf16 NHWC input: quantises x (x F8_ACT_SCALE) and the fp32 OIHW weights (per output
// channel), runs the fp8 MFMA kernel (+ the split-K reducer), returns bf16 NHWC.  For tests/test_fp8_gpu.py and tools.
extern "C" int gill_op_conv3x3_fp8(const void* x_bf16, const float* w_oihw, const float* bias, const void* resid_bf16, void* y_bf16,
                                   int B, int H, int W, int Cin, int Cout, int splitk, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GILL_REQUIRE(x_bf16 && w_oihw && y_bf16, "null argument");
  const int M = B * H * W, Kpad = conv_fp8_kpad(Cin);
  DevBuf x8, w8, sc, ws;
  GILL_TRY(x8.alloc((size_t)M * Cin));
  GILL_TRY(w8.alloc((size_t)Cout * Kpad));
  GILL_TRY(sc.alloc(sizeof(float) * (size_t)Cout));
  GILL_TRY(quant_bf16_fp8_launch((const bf16_t*)x_bf16, F8_ACT_SCALE, (int64_t)M * Cin, (unsigned char*)x8.p, s));
  GILL_TRY(conv_weight_quant_fp8_launch(w_oihw, GILL_DTYPE_F32, Cout, Cin, F8_ACT_SCALE, (unsigned char*)w8.p, (float*)sc.p, s));
  ConvF8Args a;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.M = M; a.N = Cout; a.Kpad = Kpad;
  a.A8 = (const unsigned char*)x8.p; a.W8 = (const unsigned char*)w8.p; a.colscale = (const float*)sc.p;
  a.C = (bf16_t*)y_bf16;
  a.splitk = splitk > 0 ? splitk : gemm_pick_splitk(M, Cout, 9 * Cin, 0);
  if (a.splitk > 1) {
    GILL_TRY(ws.alloc(sizeof(float) * (size_t)a.splitk * M * Cout));
    a.ws = (float*)ws.p;
    for (int r = 0; r < op_repeat(); ++r) {
      GILL_TRY(conv3x3_fp8_launch(a, s));
      GemmArgs g;     // finish the partials: + bias + residual -> bf16
      g.M = M; g.N = Cout; g.K = 9 * Cin; g.splitk = a.splitk; g.ws = a.ws; g.bias = bias;
      g.resid = resid_bf16; g.ldr = Cout; g.C = y_bf16; g.ldc = Cout;
      GILL_TRY(gemm_splitk_reduce_launch(g, s));
    }
  } else {
    a.bias = bias; a.resid = (const bf16_t*)resid_bf16;
    for (int r = 0; r < op_repeat(); ++r) GILL_TRY(conv3x3_fp8_launch(a, s));
  }
  GILL_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}
