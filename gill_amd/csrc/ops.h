// Internal C++ operator layer of libgill_amd: every hot op of the generate_images path as a
// stream-ordered launcher.  The engines (opt.hip / mapper.hip / unet.hip) are sequences of
// these launches; capi.hip re-exports a subset 1:1 for the operator-level parity tests.
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_GEGLU = 4, ACT_QUICK_GELU = 5 /* x*sigmoid(1.702x): CLIP */ };
enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_QKV = 2, OUT_SOFTMAX80 = 3 };

// C[M,N] = epilogue( alpha * A[M,K] . W[N,K]^T )       (bf16 operands, fp32 MFMA accumulate)
//  A is either a plain row-major matrix (optionally the K-concatenation of two matrices) or the
//  implicit im2col view of an NHWC tensor for a 3x3 / pad 1 convolution (stride 1|2, optional
//  fused nearest-2x upsample, optional channel-concatenation of two NHWC tensors).
struct GemmArgs {
  int M = 0, N = 0, K = 0;
  const bf16_t* A = nullptr;  int lda = 0;
  const bf16_t* A2 = nullptr; int lda2 = 0;
  int K1 = 0;                 // plain: columns taken from A (K1 == K when single source); conv: channels of A
  int conv = 0;               // 0 plain GEMM, 1 conv3x3 pad 1 over NHWC
  int IH = 0, IW = 0, OH = 0, OW = 0, Cin = 0, stride = 1, ups = 0;
  // conv only: extra plain K segment appended after the 9 taps (a fused 1x1 conv of the tensor X1 ++ X2 at the same
  // pixel, i.e. ResnetBlock2D's conv_shortcut): KX channels in total, the first KX1 from X1
  const bf16_t* X1 = nullptr; const bf16_t* X2 = nullptr; int KX = 0, KX1 = 0;
  const bf16_t* W = nullptr;  // [N][K], K contiguous.  conv: k = tap*Cin + c (conv_weight_relayout_launch), or with k_chunked
                              // k = (c / 64) * 576 + tap * 64 + c % 64 — 64-channel chunks, the 9 taps inside a chunk
                              // (conv_weight_relayout_chunked_launch) — then the KX shortcut channels
  int k_chunked = 0;          // conv: the chunk-major K order.  The 9 shifted reads of a chunk's input patch are then 9
                              // consecutive K steps and hit L2; worth it (conv_k_chunked()) once a sample's input no longer
                              // fits the 4 MiB L2 next to the weights: tap-major re-streams it from the fabric 9 times
                              // (398 MB fetched for a 42 MB input at 64x64x640), but pays one address set-up per K step
  float alpha = 1.f;
  const float* bias = nullptr;
  const float* rowvec = nullptr; int rows_per_batch = 1; int rowvec_bstride = 0;  // + rowvec[(m/rows_per_batch)*bstride + n]
  const void* resid = nullptr; int ldr = 0; int resid_f32 = 0;
  int act = ACT_NONE;
  int out_mode = OUT_BF16;
  void* C = nullptr; int ldc = 0;
  // OUT_QKV: scatter into head-major attention operands.
  //   segment = seg_base + n / (heads*dp): 0 -> Q[b][h][t][dp], 1 -> K[b][h][t][dp], 2 -> Vt[b][h][dd][t]
  bf16_t* Cq = nullptr; bf16_t* Ck = nullptr; bf16_t* Cvt = nullptr;
  int heads = 0, dp = 0, dpv = 0, ntok = 0, ntok_pad_q = 0, ntok_pad_kv = 0, seg_base = 0;
  float qscale = 1.f;   // multiplies the Q segment: softmax scale * log2(e) (attention.hip works in the log2 domain)
  int kv_tok_offset = 0;  // K / Vt rows land at token t + kv_tok_offset (appending to a KV cache); Q rows stay at t
  // split-K (0/1 = off).  ws must hold splitk*M*N floats.
  int splitk = 1; float* ws = nullptr;
  int w_blk64 = 0;         // W is stored as [N / 64][K / 64][64][64] (convert_to_bf16_blk64_launch): the STREAM64 kernel up to GEMM_STREAM64_MAX_ROWS rows, the general tiles above
  int partials_only = 0;   // split-K: write the fp32 partials and stop — the caller runs its own reducer (opt.hip: reduce + residual + LayerNorm)
  // Both kinds of fused statistics are FIXED-ORDER: a producer writes each partial sum exactly once (no atomics, nothing to
  // zero), the consumer adds the partials in index order, so two runs of the same launch sequence are bit-identical.
  // fused GroupNorm statistics of the output: gn_stats[(b * nslab + slab) * gn_groups + g][2] = {sum, sum of squares} of the
  // gn_cg channels of bin g over the output rows of slab `slab` of sample b = m / rows_per_batch; rows per slab =
  // gemm_gn_slab_rows() (64, or 16 when a split-K reducer with small blocks produces them), nslab = rows_per_batch / that
  // (row-major epilogue only; the tile width must be a multiple of gn_cg: gemm_fused_gn_ok())
  float* gn_stats = nullptr; int gn_groups = 0; int gn_cg = 0;
  // LayerNorm folded into the NEXT GEMM: LN(x) W^T + b = rstd (x (g*W)^T - mean * colsum(g*W)) + (beta W^T + b).
  //  producer: row_stats[(plane * M + m)][2] = {sum, sum of squares} of the bf16-rounded outputs of row m in the columns of
  //            plane `plane` (row-major bf16 epilogue); gemm_row_planes() tells how many planes this launch writes;
  //  consumer: ln_stats = the producer's row_stats of A over ln_planes planes, ln_colsum[n] = sum_k W'[n,k] (W' = g*W is what
  //            `W` holds, and `bias` holds beta W^T + b); mean/rstd over K with eps ln_eps.  GEGLU and QKV epilogues
  //            (+ split-K reducer).
  float* row_stats = nullptr;
  //            ln_rows (0 = M): rows per plane; rows m >= ln_rows read the sums of row m - ln_rows (a batch whose second half
  //            repeats the first: the shared classifier-free-guidance prefix).
  const float* ln_stats = nullptr; int ln_planes = 1; int ln_rows = 0; const float* ln_colsum = nullptr; float ln_eps = 1e-5f;
  // Per-sample operands (the cross-attention of UNet levels 1-3 as two GEMMs, unet.hip "XALG"): rows [b * wb_rows, (b + 1) * wb_rows)
  // multiply W + b * wb_stride (elements); OUT_SOFTMAX80 also reads bias / ln_colsum at + b * vb_stride.  Tiles must not straddle
  // samples (wb_rows % tile rows == 0).  0 = one weight matrix for all rows.
  int wb_rows = 0; int64_t wb_stride = 0; int vb_stride = 0;
  // OUT_SOFTMAX80 (folded-LayerNorm consumer, like OUT_QKV): C[m][80 h .. 80 h + 79] = softmax over the 80 columns of group h of
  // exp2-domain scores rstd (A W^T - mean colsum) + bias; N % 160 == 0; no split-K.  Columns that must not take part carry bias -1e30.
  // split-K only: the GroupNorm (+ SiLU) that consumes this output, run by the reducer itself (gemm.hip "REDUCE + GROUPNORM"):
  // fn_Y [M][N] bf16 = [silu]((C - mean) * rstd * fn_gamma + fn_beta) with mean / rstd over (rows_per_batch rows, fn_cg channels).
  // C may then be null (the raw tensor has no other reader).  gemm_fused_norm_ok() says which geometries the reducer takes.
  bf16_t* fn_Y = nullptr; const float* fn_gamma = nullptr; const float* fn_beta = nullptr; float fn_eps = 1e-5f; int fn_silu = 0; int fn_cg = 0;
  // COOP (gemm.hip "COOP"): the finish that used to be a second launch, done by the producing kernel's own workgroups, which wait for each other on
  // an arrival counter — legal because the grid is at most one workgroup per CU (gemm_coop_ok() checks it against the device), i.e. co-resident.
  //   splitk > 1: every workgroup publishes its fp32 partial tile (write-through), arrives on the counter of its (sample group, N tile), and once all
  //     slices have arrived finishes one (sample, 40-column) unit exactly as the stand-alone reducer would (slices summed in slice order, bias / row
  //     vector / residual, GroupNorm partials, and the consuming GroupNorm when fn_Y is set): no reducer launch;
  //   splitk == 1 (3x3 convolutions on the ping-pong tiles): the epilogue publishes its per-slab GroupNorm partials (gn_stats), waits for the other
  //     M tiles of its (sample, N tile), totals them in the consumer's order and writes fn_Y = [silu](GroupNorm(C)) straight from the accumulators
  //     (C itself only when non-null), and / or the per-(sample, channel) scale | shift table fn_ss [B][2][N] (what lnproj.hip applies on load).
  // coop_ctr: one zeroed 32-bit counter per (sample group, N tile) of THIS launch (gemm_coop_counters() says how many).
  unsigned* coop_ctr = nullptr; float* fn_ss = nullptr;
  // coop_splitk: also allow the split-K finish (EPI 7).  MEASURED NO-GO (profiles/r06_coop_finish.md: +1 ... +6 us per launch against conv + reducer,
  // the loop +1.7 %): the engine sets it only under GILL_GEMM_COOP=2, the operator entry sets it on request (the parity tests, tools/coop_bench.py,
  // tools/ubench/persist_resnet.hip build on it).
  int coop_splitk = 0;
};
// can this launch finish in-kernel (GemmArgs::coop_ctr)?  Geometry of the ping-pong tiles, whole samples per counter group, grid <= CUs of the device.
bool gemm_coop_ok(const GemmArgs& a);
int gemm_coop_counters(const GemmArgs& a);
int gemm_coop_giveups(unsigned* count);      // read + clear the device-side give-up counter of the bounded waits (synchronous)
int gemm_coop_mode();      // GILL_GEMM_COOP: 0 = every finish a launch of its own, 1 (default) = GroupNorm finish in the conv epilogue, 2 = + the split-K finish
bool gemm_fused_norm_ok(const GemmArgs& a);
int gemm_launch(const GemmArgs& a, hipStream_t s);
#define GN_SLAB_ROWS 64        // rows per fused GroupNorm-statistics partial of the in-kernel epilogue
#define GN_SLAB_ROWS_MIN 16    // ... of the smallest producer (size statistics buffers for rows / GN_SLAB_ROWS_MIN partials)
int gemm_gn_slab_rows(const GemmArgs& a);
#define GEMM_MAX_ROW_PLANES(N) (((N) + 63) / 64 > 2 * (((N) + 127) / 128) ? ((N) + 63) / 64 : 2 * (((N) + 127) / 128))
// planes of row_stats this launch writes (depends on the tile width and on split-K, both fixed by the arguments)
int gemm_row_planes(const GemmArgs& a);
// can a GEMM with N output columns produce fused GroupNorm statistics for bins of cg channels?
bool gemm_fused_gn_ok(int N, int cg);
// heuristic split-K factor for under-filled grids.  plain: not a conv (64-row tiles available); generic: 128-row 4-wave tiles only (the fp8 conv
// kernel): neither the 64-row nor the ping-pong rules
int gemm_pick_splitk(int M, int N, int K, int act, bool plain = false, bool generic = false);
// STREAM64 (gemm.hip): which weight matrices are stored 64 x 64-blocked, and the split-K factor of a GEMM on one.  Up to GEMM_STREAM64_MAX_ROWS rows
// such a GEMM runs on the 128 x 64 streaming tile, above on the general tiles, which read the blocked layout as well
#ifndef GEMM_STREAM64_MAX_ROWS
#define GEMM_STREAM64_MAX_ROWS 256
#endif
bool gemm_stream64_weights(int N, int K);
int gemm_pick_splitk_blk64(int M, int N, int K);

// split-K reducer of gemm_launch on its own (partials a.ws [splitk][M][N] fp32 written by another kernel: conv_fp8.hip)
int gemm_splitk_reduce_launch(const GemmArgs& a, hipStream_t s);

// 3x3 / pad 1 / stride 1 convolution with fp8 (OCP e4m3) activations and weights on v_mfma_scale_f32_16x16x128_f8f6f4
// (conv_fp8.hip): C[M, N] bf16 = colscale[n] * (A8 (*) W8) + bias + rowvec + resid, M = B*H*W, N = Cout
#define F8_ACT_SCALE 8.0f     // activations are stored as fp8(8 * x): SiLU(GroupNorm) lives in [-0.28, ~8]
struct ConvF8Args {
  int B = 0, H = 0, W = 0, Cin = 0, M = 0, N = 0, Kpad = 0;
  const unsigned char* A8 = nullptr;       // [B][H][W][Cin] fp8
  const unsigned char* W8 = nullptr;       // [N][Kpad] fp8, K order of conv_weight_quant_fp8_launch
  const float* colscale = nullptr;         // [N]: weight scale / activation scale
  const float* bias = nullptr;
  const float* rowvec = nullptr; int rows_per_batch = 1; int rowvec_bstride = 0;
  const bf16_t* resid = nullptr;           // [M][N] bf16
  bf16_t* C = nullptr;                     // [M][N] bf16
  float* gn_stats = nullptr; int gn_groups = 0; int gn_cg = 0;   // fused GroupNorm partials (GemmArgs::gn_stats layout; splitk == 1)
  int splitk = 1; float* ws = nullptr;     // fp32 partials [splitk][M][N]; finish with gemm_splitk_reduce_launch
};
int conv3x3_fp8_launch(const ConvF8Args& a, hipStream_t s);

// GEGLU projection with fp8 (e4m3) activations and weights (linear_fp8.hip): C[M][N / 2] bf16 = (A8 . Wv8^T * sv + bv) * gelu(A8 . Wg8^T * sg + bg),
// W8 [N][K] in the engine's value / gate-interleaved row order (16 value rows | 16 gate rows per 16 outputs), K % 128 == 0
#define F8_LIN_ACT_SCALE 16.0f     // LayerNorm-ed activations are stored as fp8(16 * x): |x| up to 28 sigma in range, the bulk in e4m3's normal range
struct LinF8Args {
  int M = 0, N = 0, K = 0;
  const unsigned char* A8 = nullptr;       // [M][K] fp8: ln_quant_fp8_launch
  const unsigned char* W8 = nullptr;       // [N][K] fp8: linear_weight_quant_fp8_launch
  const float* colscale = nullptr;         // [N]: weight scale / activation scale
  const float* bias = nullptr;             // [N]
  bf16_t* C = nullptr;                     // [M][N / 2] bf16
};
int geglu_fp8_launch(const LinF8Args& a, hipStream_t s);
int linear_weight_quant_fp8_launch(const bf16_t* w, int N, int K, float act_scale, unsigned char* w8, float* colscale, hipStream_t s);
int ln_quant_fp8_launch(const bf16_t* t, int M, int C, const float* ln_stats, int planes, int ln_rows, float eps, float act_scale, unsigned char* y,
                        hipStream_t s);
int conv_fp8_kpad(int Cin);
int quant_bf16_fp8_launch(const bf16_t* x, float scale, int64_t n, unsigned char* y, hipStream_t s);
int conv_weight_quant_fp8_launch(const void* w_oihw, int dtype, int Cout, int Cin, float act_scale, unsigned char* w8, float* colscale,
                                 hipStream_t s);

// Flash-style attention over head-major operands (see GemmArgs OUT_QKV):
//   Q [B][H][nq_pad][dp], K [B][H][nkv_pad][dp], Vt [B][H][dpv][nkv_pad]  ->  O [B*nq][H*dp] (token-major)
struct AttnArgs {
  const bf16_t* Q = nullptr; const bf16_t* K = nullptr; const bf16_t* Vt = nullptr; bf16_t* O = nullptr;
  int B = 0, H = 0, nq = 0, nkv = 0, nq_pad = 0, nkv_pad = 0, dp = 0, dpv = 0;
  int ldo = 0;          // row stride of O in elements (>= H*dp)
  float scale = 1.f;    // informative only: Q must arrive pre-multiplied by scale * log2(e) (GemmArgs::qscale)
  int d = 0;            // the real head dim when known (0: not stated).  dp = 48 with d = 40: the kernel uses padding dim 40 (see QF)
  int causal = 0;
  int kv_bstride_zero = 0;  // 1: K/Vt have a single batch entry shared by every b (learned queries etc.)
  int xcd_map = 1;          // 1: all query tiles of a (sample, head) on one XCD (set by the launcher)
};
int attention_launch(const AttnArgs& a, hipStream_t s);


// The feed-forward sub-block + proj_out + outer residual of a level-0 transformer block (C = 320) as one kernel (ffn.hip):
//   out = [Wp.W2 | Wp] . [ value * gelu(gate) | t ] + bo + resid,  [value | gate] = W1' . LN(t) + b1'  (LN from the row-sum planes).
// W1c / b1c / W2p: ffn_relayout_launch() of the engine's wff1 / bff1 (LayerNorm-folded GEGLU rows) and wfo ([Wp.W2 | Wp], [C][5C]).
struct FfnArgs {
  int M = 0;
  const bf16_t* T = nullptr;                                           // [M][320] residual stream (raw)
  const float* ln_stats = nullptr; int ln_planes = 0, ln_rows = 0; float ln_eps = 1e-5f;
  const bf16_t* W1c = nullptr; const float* b1c = nullptr;             // [20][128][320], [20][128]
  const bf16_t* W2p = nullptr;                                         // [320][1280], hidden order permuted within every 16
  const bf16_t* Wfo = nullptr; const float* bo = nullptr;              // [320][1600] (its last 320 columns = Wp), [320]
  const bf16_t* resid = nullptr; bf16_t* out = nullptr;                // [M][320]
  float* gn_stats = nullptr; int rows_per_batch = 0;                   // optional: GroupNorm partials of the output, bins of 5 channels,
                                                                       // [(b * rows_per_batch / 64 + slab) * 64 + bin][2] (GemmArgs::gn_stats layout)  // X != nullptr: the block's attn2.to_out + residual run in front, inside the kernel: t = Wo . X + bo2 + T (T = the stream BEFORE
  // attn2.to_out; t is never stored, its LayerNorm statistics are formed in registers: ln_stats unused).  W1c then in the permuted k
  // order and Wpp = Wp with it (ffn_relayout_launch with Wpp != nullptr).
  const bf16_t* X = nullptr;                                           // [M][384] cross-attention output (8 heads x padded 48)
  const bf16_t* Wo = nullptr; const float* bo2 = nullptr;              // [320][384], [320]
  const bf16_t* Wpp = nullptr;                                         // [320][320]
};
bool ffn_fused_supported(int C, int M);
int ffn_relayout_launch(const bf16_t* wff1, const float* bff1, const bf16_t* wfo, bf16_t* W1c, float* b1c, bf16_t* W2p, bf16_t* Wpp,
                        hipStream_t s);
int ffn_fused_launch(const FfnArgs& a, hipStream_t s);

// Two GEMMs of a level-0 transformer block around a LayerNorm as one kernel (lnproj.hip):
//   mode 0: T = W1 . X + b1 (proj_in, K1 = 320);  [q | k | v] = W2p . LN(T) + c2 scattered head-major (three segments of 384 columns)
//   mode 1: T = W1 . X + b1 + T (attn1.to_out, K1 = 384, in place);  q = W2p . LN(T) + c2 (one segment)
// W2p: lnproj_kperm_launch() of the engine's LayerNorm-folded wqkv1 / wq2; c2 their folded bias (ln_fold_rows_launch).
struct LnProjArgs {
  int mode = 0;
  int M = 0;
  const bf16_t* X = nullptr;       // [M][K1]
  bf16_t* T = nullptr;             // [M][320] residual stream (written; mode 1: read as the residual first)
  const bf16_t* W1 = nullptr;      // [320][K1]
  const float* b1 = nullptr;       // [320]
  const bf16_t* W2p = nullptr;     // [nseg * 384][320], K order permuted
  const float* c2 = nullptr;       // [nseg * 384]
  float ln_eps = 1e-5f;
  bf16_t *Cq = nullptr, *Ck = nullptr, *Cvt = nullptr;       // GemmArgs' OUT_QKV layout
  int heads = 0, dp = 0, dpv = 0, ntok = 0, ntok_pad = 0, seg_base = 0;
  float qscale = 1.f;
  // mode 0: X is the RAW block input and gn_ss [B][2][320] its GroupNorm scale / shift (groupnorm_apply_launch(..., ss_out)): the kernel
  // normalises the rows it has just loaded, y = bf16(x * scale + shift) as the stand-alone pass rounds it (ntok % 128 == 0)
  const float* gn_ss = nullptr;
};
bool lnproj_supported(int C, int M, int heads, int dp);
int lnproj_kperm_launch(const bf16_t* src, int N, bf16_t* dst, hipStream_t s);
int lnproj_launch(const LnProjArgs& a, hipStream_t s);

// LayerNorm over the last dim (eps inside sqrt), fp32 or bf16 rows in, bf16 rows out.
int layernorm_launch(const void* x, int x_f32, const float* gamma, const float* beta, bf16_t* y,
                     int rows, int C, float eps, hipStream_t s);
// same but fp32 output (used for the final norms whose result is returned to the caller)
int layernorm_f32out_launch(const void* x, int x_f32, const float* gamma, const float* beta, float* y,
                            int rows, int C, float eps, hipStream_t s);

// GroupNorm (+ optional SiLU) over NHWC bf16, input = channel-concat of up to two tensors.
//   stats: fp32 scratch of groupnorm_stats_floats(B, HW, groups) floats (per-slab partial sums, written here).
#define GN_STATS_ROWS 32       // pixels per partial of the stand-alone statistics pass
static inline size_t groupnorm_stats_floats(int B, int HW, int groups) {
  return (size_t)B * ((HW + GN_STATS_ROWS - 1) / GN_STATS_ROWS) * groups * 2 + (size_t)B * groups * 2;   // partials + totals scratch
}
// floats of the totals scratch groupnorm_apply_launch needs when a statistics block holds more than 64 partials per bin
static inline size_t groupnorm_totals_floats(int B, int nbins1, int nbins2) { return (size_t)B * (nbins1 + nbins2) * 2; }
int groupnorm_launch(const bf16_t* x1, int C1, const bf16_t* x2, int C2, int B, int HW, int groups,
                     const float* gamma, const float* beta, float eps, int silu, bf16_t* y,
                     float* stats, hipStream_t s, float out8_scale = 0.f);
// normalise from per-slab partial sums written in bins by producer epilogues (GemmArgs::gn_stats layout: nslab1 / nslab2
// partials per (sample, bin)); groupnorm_bins_align() tells whether a (channels per group, split point, bin sizes)
// combination is usable
int groupnorm_apply_launch(const bf16_t* x1, int C1, const bf16_t* x2, int C2, int B, int HW, int groups, const float* gamma,
                           const float* beta, float eps, int silu, bf16_t* y, const float* stats1, int bin1, int sc1, int nslab1,
                           const float* stats2, int bin2, int nslab2, hipStream_t s, float out8_scale = 0.f,
                           float* tot_scratch = nullptr, float* ss_out = nullptr);
// ss_out != nullptr: nothing is normalised — the per-(sample, channel) scale and shift are written as [B][2][C] floats for a consumer
// that applies y = bf16(x * scale + shift) itself (lnproj.hip mode 0: the transformer block's GroupNorm on its way into proj_in)
// out8_scale > 0: y is an fp8 (e4m3) tensor [B][HW][C] holding fp8(out8_scale * value) — the A operand of conv3x3_fp8_launch
bool groupnorm_bins_align(int cg, int sc1, int bin1, int bin2);

// ---- small elementwise / gather kernels ----
int embed_rows_bf16_launch(const int64_t* ids, const bf16_t* table, int vocab, int n, int D, bf16_t* out, hipStream_t s);
int embed_tokens_launch(const int64_t* ids, const bf16_t* table, int vocab, const bf16_t* pos_table, int pos_offset,
                        int B, int T, int D, float* out_f32, hipStream_t s);
int gather_rows_launch(const void* src, int src_f32, const int32_t* row_idx, int nrows, int D, void* dst, int dst_f32,
                       hipStream_t s);
int add_cast_launch(const void* a, int a_f32, const void* b, int b_f32, int64_t n, int64_t b_period, bf16_t* out,
                    hipStream_t s);  // out = bf16(a + b[i % b_period])
int cast_f32_to_bf16_launch(const float* x, bf16_t* y, int64_t n, hipStream_t s);
int cast_bf16_to_f32_launch(const bf16_t* x, float* y, int64_t n, hipStream_t s);
int timestep_embed_launch(const float* t, int n, int dim, bf16_t* out, hipStream_t s);  // [n][dim] = [cos | sin]
int silu_bf16_launch(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t s);
// conv_in: NCHW fp32 (B,Cin,H,W) -> NHWC bf16 (B,H,W,Cout), 3x3 pad 1, direct (Cin tiny)
int conv_in_launch(const float* x, const bf16_t* w /*[Cout][9][Cin]*/, const float* bias, int B, int Cin, int H, int W,
                   int Cout, bf16_t* y, hipStream_t s);
// im2col of a tiny-Cin NCHW fp32 tensor for conv_in: out [B*H*W][kpad] bf16, k = tap*Cin + c (zero beyond 9*Cin)
int im2col_nchw_launch(const float* x, int B, int Cin, int H, int W, int kpad, bf16_t* out, hipStream_t s, unsigned* zero = nullptr, int nzero = 0);
// conv_out: NHWC bf16 (B,H,W,Cin) -> NCHW fp32 (B,Cout,H,W), 3x3 pad 1, direct (Cout tiny)
int conv_out_launch(const bf16_t* x, const bf16_t* w /*[Cout][9][Cin]*/, const float* bias, int B, int Cin, int H, int W,
                    int Cout, float* y, hipStream_t s);
// skinny GEMV-ish: out[M][N] (f32) = x[M][K] (bf16) . W[N][K]^T ; M <= 8, any N (lm_head)
int skinny_gemm_launch(const bf16_t* x, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s);
// classifier-free guidance + PLMS update on fp32 NCHW latents (see unet.hip for the coefficient layout)
// One row per UNet call of a denoise loop, computed on the host once per call of gill_sd_denoise and read by the
// device: the loop's only per-step inputs.  The step index itself lives on the device (SdLoopArgs::ctr), so replaying one
// captured UNet step N times needs no host-side per-step argument and no copy between replays.
struct PlmsRow {
  int mode;              // 0: first step, 1: repeated step, 2..4: multistep orders
  int slot_new;          // ring slot to write the new guided eps into (-1: don't store)
  int s1, s2, s3;        // ring slots of ets[-2], ets[-3], ets[-4] where needed (ets[-1] = new / slot of last)
  float sample_coeff, eps_coeff;   // x_prev = sample_coeff * sample - eps_coeff * eps'
};
struct SdLoopArgs {
  const PlmsRow* rows;   // [ncalls] on device
  int* ctr;              // ctr[0]: next step (written by the PLMS kernel only), ctr[1]: current step (written by the stage kernel only)
  const float* temb_table; int temb_total;   // [ncalls][temb_total] -> temb_cur
  float* temb_cur;
  const float* eps;      // [2B][n] : uncond rows then cond rows
  float* lat;            // [B][n] in/out
  float* lat2;           // [2B][n] (or [B][n]): the UNet input = cat([latents] * 2)
  float* cur_sample;     // [B][n] saved sample (PLMS warm-up)
  float* ets;            // [4][B][n] ring of past eps'
  int B; int64_t n;      // n = C*H*W per sample
  const float* guidance; // device scalar (not a kernel argument: one captured step graph serves every guidance value)
  int cfg;               // 1: eps holds [uncond | cond] halves and guidance is applied; 0: eps is [B][n]
};
// plain device-side fill / copy kernels for use INSIDE a captured forward: hipMemsetAsync / hipMemcpyAsync become memset /
// memcpy graph nodes, whose replay was not reliable (profiles/r02_soak_bisect.md).  16-byte aligned pointers and sizes.
int zero_bytes_launch(void* dst, size_t bytes, hipStream_t s);
int copy_bytes_launch(void* dst, const void* src, size_t bytes, hipStream_t s);
int dup_pair_launch(void* a, const void* b, void* b2, size_t bytes, hipStream_t s);   // a[bytes..] := a[0..bytes); b2[0..bytes) = b2[bytes..] := b[0..bytes)
// first kernel of a step: latents -> UNet input (both CFG halves), time-embedding row of the current step -> temb_cur
int sd_stage_launch(const SdLoopArgs& a, hipStream_t s);
// last kernel of a step: CFG combine + PLMS update of the latents (custom_sd.py:641-646), then step counter + 1
int plms_step_launch(const SdLoopArgs& a, hipStream_t s);

// weight re-layout helpers (run once at engine creation)
int conv_weight_relayout_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out /*[Cout][9][Cin]*/, hipStream_t s);   // conv_in / conv_out
// measured (tools/one_op.py, 8 samples): 64x64x640 -> 320: 150.9 -> 145.4 us, 64x64x960: 207.6 -> 193.7 us; but 64x64x320: 75.9 ->
// 77.8 us and the split-K level-1 convs lose 14-18 % (32x32x1280 -> 640: 166.8 -> 193.2 us)
// (not for convolutions that run on the ping-pong 256 x 160 tile — gemm_conv_pingpong(): there a new (tap, chunk) segment every K
// step lengthens the memory phase, measured 155 us chunk-major against 126 us tap-major for 64 x 64 x 640 -> 320)
bool gemm_conv_pingpong(int rows_multiple_of, int Cout);     // true: gemm_launch runs this conv's 3x3 GEMM on the ping-pong kernel
bool conv_k_chunked(int HW, int Cin, int Cout);   // (GILL_CONV_KORDER = 0 | 1 forces tap-major / chunk-major: tests and tools)
int conv_weight_relayout_ups4_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out /*[4][Cout][4][Cin]*/, hipStream_t s);   // GemmArgs::ups == 2
int conv_weight_relayout_chunked_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out /*[Cout][Cin/64][9][64]*/, hipStream_t s);   // GemmArgs::conv
int convert_to_bf16_launch(const void* src, int dtype, int64_t n, bf16_t* dst, hipStream_t s);
int convert_to_bf16_blk64_launch(const void* src, int dtype, int N, int K, bf16_t* dst, hipStream_t s);   // -> [N / 64][K / 64][64][64] (STREAM64)
int convert_to_f32_launch(const void* src, int dtype, int64_t n, float* dst, hipStream_t s);
// copy rows of a [rows][cols] matrix into a (possibly wider/padded/permuted) destination:
//   dst[dst_row_of(r)][0..cols) = src[r][0..cols)   with dst_row index list on device
int scatter_rows_bf16_launch(const bf16_t* src, int rows, int cols, const int32_t* dst_rows, bf16_t* dst, int dst_ld,
                             hipStream_t s);

int permute_f32_launch(const float* src, const int32_t* idx, int n, float* dst, hipStream_t s);
// fold LayerNorm(g, beta) into the Linear that follows it: W <- bf16(W*g) in place, colsum[n] = sum_k W', bias[n] += beta.W[n]
int ln_fold_rows_launch(bf16_t* W, int N, int K, const float* g, const float* beta, float* colsum, float* bias, hipStream_t s);
int pack_heads_launch(const bf16_t* src, int B, int n, int H, int d, int n_pad, int dp, int dpv, int mode, bf16_t* dst,
                      hipStream_t s, float mul = 1.f);
int unpad_heads_launch(const bf16_t* src, int64_t rows, int H, int d, int dp, bf16_t* dst, hipStream_t s);

// padded head dim used by the attention kernel for a true head dim d (0 = unsupported)
static inline int attn_padded_dim(int d) {
  if (d <= 48) return 48;
  if (d <= 64) return 64;
  if (d <= 80) return 80;
  if (d <= 128) return 128;
  if (d <= 160) return 160;
  return 0;
}

const bf16_t* gill_zero_page();   // >= 256 B of device zeros
