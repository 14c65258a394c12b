// The cross-attention sub-block of a UNet BasicTransformerBlock as ONE kernel (gfx950):
//
//   t'  = t + o1 . Wo1^T + bo1                  attn1.to_out[0] + residual            (was: one GEMM launch)
//   q   = LN2(t') . Wq^T                        norm2 folded into attn2.to_q          (was: one GEMM launch, head-major scatter)
//   o2  = softmax(q K^T / sqrt(d)) V            77 cached keys of the prompt          (was: one attention launch)
//   t'' = t' + o2 . Wo2^T + bo2                 attn2.to_out[0] + residual            (was: one GEMM launch)
//
// Reference semantics: diffusers BasicTransformerBlock as restated in oracle/unet_ref.py:_transformer (attn1 output
// projection, norm2, attn2, residuals); call site gill/custom_sd.py:633-638.
//
// Every one of these maps is row-local (a token only meets its own sample's 77 prompt keys), so a workgroup owns ROWS
// consecutive tokens of one sample and walks the whole chain: t' and q never leave the CU (t' stays in registers as packed bf16
// for the last residual and in LDS as the A operand of the q projection; q and o2 share that LDS tile), three launches, their
// first-fetch latencies and four M x C round trips through HBM per block are gone.  Weights (3 x C x HDP bf16, 0.7 MB at level 0)
// stream from L2 through one LDS ring that runs on across the three GEMMs: the first stages of the next GEMM are in flight
// while the previous one runs its epilogue (and, for the last GEMM, during the attention phase).
//
// Workgroup = 8 waves.  GEMM phases: waves 2 (M) x 4 (N), wave tile (MI * 16) x (BN / 4) on v_mfma_f32_16x16x32_bf16 with the
// weight fragment as the A operand (a lane ends up with 4 consecutive columns of one row per 16-wide sub-tile; weight rows are
// dealt to LDS tile rows so that two sub-tiles give 8 consecutive columns: 16-byte LDS / global stores).  K stages are 32 wide
// (a stage = all BN weight rows x 64 B, + the ROWS x 64 B of o1 in the first GEMM), filled by global_load_lds_dwordx4 with the
// bank swizzle on the source address, 3-4 stages deep with counted vmcnt.  Attention phase: one wave per head, everything
// transposed as in attention.hip (S^T = K q^T, O^T = V^T P^T on v_mfma_f32_32x32x16_bf16; P never leaves registers); K / V^T
// fragments come straight from L2 (9-15 KB per (sample, head), shared by every workgroup of the sample).
#include "ops.h"

namespace {

template <int C_, int HDP_, int DP_, int H_, int MI_, int NPASS_, int STAGES_>
struct XCfg {
  static constexpr int C = C_, HDP = HDP_, DP = DP_, H = H_, MI = MI_, NPASS = NPASS_, STAGES = STAGES_;
  static constexpr int ROWS = 2 * MI * 16;
  static constexpr int BN1 = C / NPASS, BN2 = HDP / NPASS;        // output columns per pass: GEMMs 1/3 (N = C), GEMM 2 (N = HDP)
  static constexpr int NT1 = BN1 / 64, NT2 = BN2 / 64;            // 16-wide sub-tiles per wave (wave = BN / 4 columns)
  static constexpr int KS1 = HDP / 32, KS2 = C / 32;              // 32-wide K stages: GEMMs 1/3 (K = HDP), GEMM 2 (K = C)
  static constexpr int PA = ROWS / 16, PW1 = BN1 / 16, PW2 = BN2 / 16;   // 1-KiB DMA pieces per stage
  static constexpr int BNMAX = BN1 > BN2 ? BN1 : BN2;
  static constexpr int STAGE_BYTES = (ROWS + BNMAX) * 64;
  static constexpr int TSTR = (HDP > C ? HDP : C) + 8;            // T tile row stride in elements (odd number of 16-B slots)
  static constexpr int T_BYTES = ROWS * TSTR * 2;
  static constexpr int STAT_BYTES = ROWS * 4 * 8;
  static constexpr int SMEM = T_BYTES + STAT_BYTES + STAGES * STAGE_BYTES;
  static constexpr int NS1 = NPASS * KS1, NS2 = NPASS * KS2, NS = 2 * NS1 + NS2;
  static constexpr int NG1 = (NT1 + 1) / 2, NG2 = (NT2 + 1) / 2;
  static constexpr int DPV = (DP + 31) / 32 * 32;
  static constexpr int KSA = DP / 16;                              // QK^T k-steps
  static constexpr int NDT = DPV / 32;                             // 32-row d tiles of O^T
  static constexpr int NKT = 3;                                    // 32-key tiles: ctx_pad = 96
  static_assert(BN1 % 64 == 0 && BN2 % 64 == 0, "wave tiles are whole 16-wide sub-tiles");
  static_assert(HDP % 32 == 0 && C % 32 == 0 && DP % 16 == 0, "K stages / QK steps");
  static_assert(((TSTR * 2 / 16) & 1) == 1, "T rows must start at odd multiples of 16 B (conflict-free fragment reads)");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (waiting for more than needed is always correct)
  }
}

// LDS tile row R of a pass (BN weight rows, the wave with column index R / (BN/4) owns it) -> column of the pass whose weight row it
// holds: sub-tiles (2g, 2g+1) of a lane hold 8 consecutive columns (see gemm.hip "WIDE EPILOGUE STORES")
template <int BN>
__device__ __forceinline__ int xa_tile_col(int R) {
  constexpr int WC = BN / 4, NT = WC / 16;
  const int quarter = R / WC, q = R - quarter * WC;
  const int j = q >> 4, x = q & 15;
  const int c = ((j | 1) < NT) ? (j >> 1) * 32 + (x >> 2) * 8 + (j & 1) * 4 + (x & 3) : q;
  return quarter * WC + c;
}

}  // namespace

template <typename Cfg>
__global__ __launch_bounds__(512, 1) void xattn_block_kernel(const XattnArgs p) {
  constexpr int C = Cfg::C, HDP = Cfg::HDP, DP = Cfg::DP, H = Cfg::H, MI = Cfg::MI, NPASS = Cfg::NPASS, STAGES = Cfg::STAGES;
  constexpr int ROWS = Cfg::ROWS, TSTR = Cfg::TSTR;
  extern __shared__ __attribute__((aligned(16))) unsigned char xa_smem[];
  bf16_t* T = reinterpret_cast<bf16_t*>(xa_smem);                                   // [ROWS][TSTR]: t', then q, then o2
  float2* stats = reinterpret_cast<float2*>(xa_smem + Cfg::T_BYTES);                // [ROWS][4] row sums of t' per wave column
  unsigned char* ring = xa_smem + Cfg::T_BYTES + Cfg::STAT_BYTES;                   // [STAGES][(ROWS + BNMAX) x 64 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int frow = lane & 15, fkc = lane >> 4;
  const int m0 = blockIdx.x * ROWS;
  const int bsample = m0 / p.HW;                    // a workgroup's rows belong to one sample (HW % ROWS == 0)

  // ---------------------------------------------------------------- the stage stream of the three GEMMs
  // stage g (0 .. NS-1): phase 0 = GEMM 1 (A = o1 rows through the ring, W = Wo1), 1 = GEMM 2 (W = Wq), 2 = GEMM 3 (W = Wo2);
  // inside a phase NPASS passes over the output columns, inside a pass the K stages.
  const int drow = lane >> 2, dchunk = lane & 3;    // DMA lane geometry: a piece = 16 rows x 64 B
  auto issue = [&](int g) -> int {
    int ph, pass, k;
    if (g < Cfg::NS1) { ph = 0; pass = g / Cfg::KS1; k = g - pass * Cfg::KS1; }
    else if (g < Cfg::NS1 + Cfg::NS2) { ph = 1; const int r = g - Cfg::NS1; pass = r / Cfg::KS2; k = r - pass * Cfg::KS2; }
    else { ph = 2; const int r = g - Cfg::NS1 - Cfg::NS2; pass = r / Cfg::KS1; k = r - pass * Cfg::KS1; }
    unsigned char* slot = ring + (g % STAGES) * Cfg::STAGE_BYTES;
    const int npa = (ph == 0) ? Cfg::PA : 0;
    const int npw = (ph == 1) ? Cfg::PW2 : Cfg::PW1;
    const bf16_t* Wm = (ph == 0) ? p.Wo1 : (ph == 1 ? p.Wq : p.Wo2);
    const int Kw = (ph == 1) ? C : HDP;
    int cnt = 0;
    for (int q = w; q < npa + npw; q += 8) {
      const bf16_t* src;
      unsigned char* dst;
      if (q < npa) {
        const int row = q * 16 + drow;
        int m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        if (m >= p.src_rows) m -= p.src_rows;     // shared CFG prefix: o1 / t exist for the first half of the batch only
        src = p.o1 + (size_t)m * HDP + k * 32 + ((dchunk ^ ((row >> 2) & 3)) * 8);
        dst = slot + q * 1024;
      } else {
        const int R = (q - npa) * 16 + drow;
        const int n = (ph == 1) ? pass * Cfg::BN2 + xa_tile_col<Cfg::BN2>(R) : pass * Cfg::BN1 + xa_tile_col<Cfg::BN1>(R);
        src = Wm + (size_t)n * Kw + k * 32 + ((dchunk ^ ((R >> 2) & 3)) * 8);
        dst = slot + ROWS * 64 + (q - npa) * 1024;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      ++cnt;
    }
    return cnt;
  };

  // in-flight bookkeeping: pc[j] = pieces this wave issued for stage (cur + 1 + j), j < STAGES - 2 ... kept as named scalars
  int gcur = 0;                 // stage being consumed
  int pend1 = 0, pend2 = 0;   // pieces of stages gcur+1, gcur+2 (issued, possibly in flight)
  {
    int c0 = issue(0);
    (void)c0;
    if (1 < Cfg::NS && STAGES >= 3) pend1 = issue(1);
    if (2 < Cfg::NS && STAGES >= 4) pend2 = issue(2);
  }
  // one K stage of a GEMM pass: wait for the stage, barrier, read fragments, refill the slot freed by the previous stage, multiply
  auto consume_begin = [&]() -> const unsigned char* {
    const int allowed = (STAGES >= 4) ? pend1 + pend2 : pend1;     // stages gcur+1 .. gcur+STAGES-2 may stay in flight
    wait_vm_n(allowed);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes (epilogue tiles, statistics) are done before the barrier
    __builtin_amdgcn_s_barrier();
    return ring + (gcur % STAGES) * Cfg::STAGE_BYTES;
  };
  auto consume_refill = [&]() {
    const int gn = gcur + STAGES - 1;
    const int c = (gn < Cfg::NS) ? issue(gn) : 0;
    if (STAGES >= 4) { pend1 = pend2; pend2 = c; } else { pend1 = c; }
    ++gcur;
  };

  // ---------------------------------------------------------------- GEMM 1: t' = t + o1 Wo1^T + bo1
  uint4 tpk[NPASS][MI][Cfg::NG1];       // t' as packed bf16 (the residual of GEMM 3), per (pass, row sub-tile, column group)
  float rs[MI], rq[MI];                 // row sums of the bf16-rounded t' over this wave's columns (LayerNorm statistics)
#pragma unroll
  for (int i = 0; i < MI; ++i) { rs[i] = 0.f; rq[i] = 0.f; }
  const int arow0 = wm * (MI * 16) + frow;           // this lane's first row inside the tile (sub-tile i adds 16 i)
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    constexpr int NT = Cfg::NT1, BN = Cfg::BN1;
    f32x4 acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Cfg::KS1; ++k) {
      const unsigned char* slot = consume_begin();
      bf16x8 af[MI], bfr[NT];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = arow0 + i * 16;
        af[i] = *reinterpret_cast<const bf16x8*>(slot + row * 64 + ((fkc ^ ((row >> 2) & 3)) * 16));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int R = wn * (BN / 4) + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(slot + ROWS * 64 + R * 64 + ((fkc ^ ((R >> 2) & 3)) * 16));
      }
      consume_refill();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    // epilogue: + bias + residual, round to bf16, keep (registers + LDS tile), LayerNorm row sums
#pragma unroll
    for (int g = 0; g < Cfg::NG1; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int j1 = pair ? 2 * g + 1 : 2 * g;
      const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;
      const int n = pass * BN + wn * (BN / 4) + cl;
      const float4 b0 = *reinterpret_cast<const float4*>(p.bo1 + n);
      const float4 b1 = pair ? *reinterpret_cast<const float4*>(p.bo1 + n + 4) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = arow0 + i * 16;
        int m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        if (m >= p.src_rows) m -= p.src_rows;
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
        float v[8] = {a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w, a1[0] + b1.x, a1[1] + b1.y, a1[2] + b1.z, a1[3] + b1.w};
        uint4 r = make_uint4(0, 0, 0, 0);
        if (pair) r = *reinterpret_cast<const uint4*>(p.t + (size_t)m * C + n);
        else { const uint2 r2 = *reinterpret_cast<const uint2*>(p.t + (size_t)m * C + n); r.x = r2.x; r.y = r2.y; }
        v[0] += bf2f((bf16_t)(r.x & 0xffff)); v[1] += bf2f((bf16_t)(r.x >> 16));
        v[2] += bf2f((bf16_t)(r.y & 0xffff)); v[3] += bf2f((bf16_t)(r.y >> 16));
        v[4] += bf2f((bf16_t)(r.z & 0xffff)); v[5] += bf2f((bf16_t)(r.z >> 16));
        v[6] += bf2f((bf16_t)(r.w & 0xffff)); v[7] += bf2f((bf16_t)(r.w >> 16));
        uint4 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
        tpk[pass][i][g] = o;
        bf16_t* dst = T + row * TSTR + n;
        if (pair) *reinterpret_cast<uint4*>(dst) = o;
        else *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y);
        const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
        const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
        rs[i] += (r0 + r1) + (r2 + r3);
        rq[i] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        if (pair) {
          const float r4 = __uint_as_float(o.z << 16), r5 = __uint_as_float(o.z & 0xffff0000u);
          const float r6 = __uint_as_float(o.w << 16), r7 = __uint_as_float(o.w & 0xffff0000u);
          rs[i] += (r4 + r5) + (r6 + r7);
          rq[i] += (r4 * r4 + r5 * r5) + (r6 * r6 + r7 * r7);
        }
      }
    }
  }
  // LayerNorm statistics of t': a row's columns of this wave sit in the 4 lanes {frow, +16, +32, +48}; the 4 wave columns meet in LDS
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    rs[i] += __shfl_xor(rs[i], 16, 64); rq[i] += __shfl_xor(rq[i], 16, 64);
    rs[i] += __shfl_xor(rs[i], 32, 64); rq[i] += __shfl_xor(rq[i], 32, 64);
    if (fkc == 0) stats[(arow0 + i * 16) * 4 + wn] = make_float2(rs[i], rq[i]);
  }
  if (p.debug_stop == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  // (the barrier of GEMM 2's first stage orders these LDS writes — t' tile and statistics — before every wave's reads)

  // ---------------------------------------------------------------- GEMM 2: q = LN(t') Wq^T (folded LayerNorm), x qscale
  uint4 qpk[NPASS][MI][Cfg::NG2];
  float ln_rr[MI], ln_rm[MI];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    constexpr int NT = Cfg::NT2, BN = Cfg::BN2;
    f32x4 acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Cfg::KS2; ++k) {
      const unsigned char* slot = consume_begin();
      bf16x8 af[MI], bfr[NT];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(T + (arow0 + i * 16) * TSTR + k * 32 + fkc * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int R = wn * (BN / 4) + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(slot + ROWS * 64 + R * 64 + ((fkc ^ ((R >> 2) & 3)) * 16));
      }
      consume_refill();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (pass == 0) {
      // row factors of the folded LayerNorm: the 4 wave-column partials of a row, added in wave-column order (fixed order)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const float2* st = stats + (arow0 + i * 16) * 4;
        const float2 s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3];
        const float sum = ((s0.x + s1.x) + s2.x) + s3.x, sq = ((s0.y + s1.y) + s2.y) + s3.y;
        const float mean = sum * (1.f / (float)C);
        const float var = fmaxf(sq * (1.f / (float)C) - mean * mean, 0.f);
        ln_rr[i] = rsqrtf(var + p.ln_eps);
        ln_rm[i] = mean * ln_rr[i];
      }
    }
#pragma unroll
    for (int g = 0; g < Cfg::NG2; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int j1 = pair ? 2 * g + 1 : 2 * g;
      const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;
      const int n = pass * BN + wn * (BN / 4) + cl;
      const float4 b0 = *reinterpret_cast<const float4*>(p.q_bias + n), c0 = *reinterpret_cast<const float4*>(p.q_colsum + n);
      const float4 b1 = pair ? *reinterpret_cast<const float4*>(p.q_bias + n + 4) : make_float4(0, 0, 0, 0);
      const float4 c1 = pair ? *reinterpret_cast<const float4*>(p.q_colsum + n + 4) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
        const float rr = ln_rr[i], rm = ln_rm[i], qs = p.qscale;
        float v[8] = {(a0[0] * rr - rm * c0.x + b0.x) * qs, (a0[1] * rr - rm * c0.y + b0.y) * qs, (a0[2] * rr - rm * c0.z + b0.z) * qs,
                      (a0[3] * rr - rm * c0.w + b0.w) * qs, (a1[0] * rr - rm * c1.x + b1.x) * qs, (a1[1] * rr - rm * c1.y + b1.y) * qs,
                      (a1[2] * rr - rm * c1.z + b1.z) * qs, (a1[3] * rr - rm * c1.w + b1.w) * qs};
        uint4 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
        qpk[pass][i][g] = o;
      }
    }
  }
  // every wave is done reading t' from the tile: q takes its place
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
    for (int g = 0; g < Cfg::NG2; ++g) {
      constexpr int NT = Cfg::NT2, BN = Cfg::BN2;
      const bool pair = (2 * g + 1 < NT);
      const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;
      const int n = pass * BN + wn * (BN / 4) + cl;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        bf16_t* dst = T + (arow0 + i * 16) * TSTR + n;
        if (pair) *reinterpret_cast<uint4*>(dst) = qpk[pass][i][g];
        else *reinterpret_cast<uint2*>(dst) = make_uint2(qpk[pass][i][g].x, qpk[pass][i][g].y);
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (p.debug_stop == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }

  // ---------------------------------------------------------------- attention over the prompt's keys: one wave per head
  {
    const int lq = lane & 31, hi = lane >> 5;
    for (int h = w; h < H; h += 8) {
      const bf16_t* Kb = p.Kc + (size_t)(bsample * H + h) * p.ctx_pad * DP;
      const bf16_t* Vb = p.Vt + (size_t)(bsample * H + h) * Cfg::DPV * p.ctx_pad;
      bf16x8 kf[Cfg::NKT][Cfg::KSA];
#pragma unroll
      for (int kt = 0; kt < Cfg::NKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < Cfg::KSA; ++ks)
          kf[kt][ks] = *reinterpret_cast<const bf16x8*>(Kb + (size_t)(kt * 32 + lq) * DP + ks * 16 + hi * 8);
#pragma unroll
      for (int qt = 0; qt < ROWS / 32; ++qt) {
        const bf16_t* qrow = T + (qt * 32 + lq) * TSTR + h * DP + hi * 8;
        bf16x8 qf[Cfg::KSA];
#pragma unroll
        for (int ks = 0; ks < Cfg::KSA; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
        f32x16 s[Cfg::NKT];
#pragma unroll
        for (int kt = 0; kt < Cfg::NKT; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < Cfg::KSA; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[ks], s[kt], 0, 0, 0);
        }
        // scores are in the log2 domain (q carries scale * log2 e); keys >= ctx_len are padding
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < Cfg::NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kv >= p.ctx_len) s[kt][r] = -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < Cfg::NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); l += s[kt][r]; }
        l += __shfl_xor(l, 32, 64);
        f32x16 oacc[Cfg::NDT];
#pragma unroll
        for (int t = 0; t < Cfg::NDT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
#pragma unroll
        for (int s16 = 0; s16 < 2 * Cfg::NKT; ++s16) {       // 16-key steps of the PV product
          union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf2(s[s16 >> 1][(s16 & 1) * 8 + 2 * j], s[s16 >> 1][(s16 & 1) * 8 + 2 * j + 1]);
#pragma unroll
          for (int t = 0; t < Cfg::NDT; ++t) {
            const bf16_t* vrow = Vb + (size_t)(t * 32 + lq) * p.ctx_pad + 16 * s16 + 4 * hi;
            union { bf16x8 v; uint2 u[2]; } vf;
            vf.u[0] = *reinterpret_cast<const uint2*>(vrow);
            vf.u[1] = *reinterpret_cast<const uint2*>(vrow + 8);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pk.v, oacc[t], 0, 0, 0);
          }
        }
        const float inv = 1.f / l;
        bf16_t* orow = T + (qt * 32 + lq) * TSTR + h * DP;
#pragma unroll
        for (int t = 0; t < Cfg::NDT; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = t * 32 + 8 * g + 4 * hi;
            if (d < DP) {
              uint2 o;
              o.x = pack_bf2(oacc[t][g * 4 + 0] * inv, oacc[t][g * 4 + 1] * inv);
              o.y = pack_bf2(oacc[t][g * 4 + 2] * inv, oacc[t][g * 4 + 3] * inv);
              *reinterpret_cast<uint2*>(orow + d) = o;
            }
          }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (p.debug_stop == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  // (GEMM 3's first stage barrier orders the o2 tile writes before the fragment reads)

  // ---------------------------------------------------------------- GEMM 3: t'' = t' + o2 Wo2^T + bo2 (+ LayerNorm row sums of t'')
  float ws[MI], wq[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { ws[i] = 0.f; wq[i] = 0.f; }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    constexpr int NT = Cfg::NT1, BN = Cfg::BN1;
    f32x4 acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Cfg::KS1; ++k) {
      const unsigned char* slot = consume_begin();
      bf16x8 af[MI], bfr[NT];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(T + (arow0 + i * 16) * TSTR + k * 32 + fkc * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int R = wn * (BN / 4) + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(slot + ROWS * 64 + R * 64 + ((fkc ^ ((R >> 2) & 3)) * 16));
      }
      consume_refill();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < Cfg::NG1; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int j1 = pair ? 2 * g + 1 : 2 * g;
      const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;
      const int n = pass * BN + wn * (BN / 4) + cl;
      const float4 b0 = *reinterpret_cast<const float4*>(p.bo2 + n);
      const float4 b1 = pair ? *reinterpret_cast<const float4*>(p.bo2 + n + 4) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + arow0 + i * 16;
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
        const uint4 r = tpk[pass][i][g];
        float v[8] = {a0[0] + b0.x + bf2f((bf16_t)(r.x & 0xffff)), a0[1] + b0.y + bf2f((bf16_t)(r.x >> 16)),
                      a0[2] + b0.z + bf2f((bf16_t)(r.y & 0xffff)), a0[3] + b0.w + bf2f((bf16_t)(r.y >> 16)),
                      a1[0] + b1.x + bf2f((bf16_t)(r.z & 0xffff)), a1[1] + b1.y + bf2f((bf16_t)(r.z >> 16)),
                      a1[2] + b1.z + bf2f((bf16_t)(r.w & 0xffff)), a1[3] + b1.w + bf2f((bf16_t)(r.w >> 16))};
        uint4 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
        if (m < p.M) {
          bf16_t* dst = p.out + (size_t)m * C + n;
          if (pair) *reinterpret_cast<uint4*>(dst) = o;
          else *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y);
        }
        const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
        const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
        ws[i] += (r0 + r1) + (r2 + r3);
        wq[i] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        if (pair) {
          const float r4 = __uint_as_float(o.z << 16), r5 = __uint_as_float(o.z & 0xffff0000u);
          const float r6 = __uint_as_float(o.w << 16), r7 = __uint_as_float(o.w & 0xffff0000u);
          ws[i] += (r4 + r5) + (r6 + r7);
          wq[i] += (r4 * r4 + r5 * r5) + (r6 * r6 + r7 * r7);
        }
      }
    }
  }
  if (p.row_stats) {     // planes of the folded LayerNorm that follows (norm3 -> GEGLU): one per wave column, [4][M][2]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      ws[i] += __shfl_xor(ws[i], 16, 64); wq[i] += __shfl_xor(wq[i], 16, 64);
      ws[i] += __shfl_xor(ws[i], 32, 64); wq[i] += __shfl_xor(wq[i], 32, 64);
      const int m = m0 + arow0 + i * 16;
      if (fkc == 0 && m < p.M) *reinterpret_cast<float2*>(p.row_stats + ((size_t)wn * p.M + m) * 2) = make_float2(ws[i], wq[i]);
    }
  }
}

template <typename Cfg>
static int xattn_launch_cfg(const XattnArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_block_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    attr_set = true;
  }
  hipLaunchKernelGGL((xattn_block_kernel<Cfg>), dim3(cdiv(a.M, Cfg::ROWS)), dim3(512), Cfg::SMEM, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// geometries with a fused kernel: SD-1.x levels 0 / 1 (C 320 / 640, 8 heads of 40 / 80 padded to 48 / 80) and SD-2.x levels 0 / 1
// (5 / 10 heads of 64).  Level 2 (C = 1280) stays on the four launches: a row tile would have to stream 3 x 3.3 MB of weights.
bool xattn_block_supported(int C, int heads, int dp, int HW, int ctx_pad) {
  if (ctx_pad != 96) return false;
  if (C == 320 && heads == 8 && dp == 48) return HW % 64 == 0;
  if (C == 640 && heads == 8 && dp == 80) return HW % 32 == 0;
  if (C == 320 && heads == 5 && dp == 64) return HW % 64 == 0;
  if (C == 640 && heads == 10 && dp == 64) return HW % 32 == 0;
  return false;
}

int xattn_block_launch(const XattnArgs& a, hipStream_t s) {
  GILL_REQUIRE(a.o1 && a.t && a.out && a.Wo1 && a.Wq && a.Wo2 && a.Kc && a.Vt && a.bo1 && a.bo2 && a.q_bias && a.q_colsum, "null argument");
  GILL_REQUIRE(xattn_block_supported(a.C, a.heads, a.dp, a.HW, a.ctx_pad), "xattn_block: unsupported geometry");
  GILL_REQUIRE(a.M > 0 && a.M % a.HW == 0 && a.src_rows > 0 && a.src_rows <= a.M && a.M % a.src_rows == 0, "xattn_block: bad row counts");
  GILL_REQUIRE(a.ctx_len > 0 && a.ctx_len <= a.ctx_pad, "xattn_block: bad context length");
  if (a.C == 320 && a.heads == 8) return xattn_launch_cfg<XCfg<320, 384, 48, 8, 2, 1, 3>>(a, s);
  if (a.C == 640 && a.heads == 8) return xattn_launch_cfg<XCfg<640, 640, 80, 8, 1, 2, 4>>(a, s);
  if (a.C == 320 && a.heads == 5) return xattn_launch_cfg<XCfg<320, 320, 64, 5, 2, 1, 3>>(a, s);
  return xattn_launch_cfg<XCfg<640, 640, 64, 10, 1, 2, 4>>(a, s);
}
