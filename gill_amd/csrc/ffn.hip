// The feed-forward sub-block of a BasicTransformerBlock (diffusers: norm3 -> GEGLU -> ff.net.2 -> + residual, followed in the UNet by
// proj_out + the outer residual) as ONE kernel per 128-row tile, for C = 320 (hidden 1280: SD-1.5 / SD-2.1 level 0):
//   out[m] = [Wp.W2 | Wp] . [ value(m) * gelu(gate(m)) | t(m) ] + b + x_in(m),     [value | gate](m) = W1' . LN(t(m)) + b1'
// (W1' / b1' carry norm3's gain / shift: ln_fold_rows_launch; [Wp.W2 | Wp], b: ffo_fuse_kernel in unet.hip.)
// It replaces the GEGLU GEMM (97 us at M = 32768) and the two-source ffo GEMM (50 us) and the 84 MB tensor between them.
//
// Structure (prototype and measurements: tools/ubench/ffn_fused.hip, profiles/r03_ffn_fused_prototype.md).  Inside the 256-register
// budget of two waves per SIMD this tile does not fit, so:
//   * FOUR waves per workgroup, ONE per SIMD (amdgpu_waves_per_eu(1,1): the whole 512-register file), one workgroup per CU.  A wave owns
//     32 rows of the tile through the whole block; nothing is exchanged between waves.
//       t fragments (32 rows x 320) are loaded once, straight from global memory, as the B operand of 32x32x16 MFMAs (80 registers);
//       first  out = Wp . t  on the RAW rows (10 stages), then the fragments are LayerNorm-ed in place (mean / rstd from the row-sum
//       planes their producer wrote: GemmArgs::row_stats) and serve the 20 hidden chunks of 64:
//       S1: acc[2 value + 2 gate tiles] = W1' chunk . LN(t)  (5 stages) -> bias, GEGLU in registers -> the products ARE the B operand of
//       S2 (the accumulator registers 8hh .. 8hh+7 of a lane are the k slots of a 32x32x16 B fragment once the A operand's hidden order
//       is permuted [0-3][8-11][4-7][12-15] within every 16: ffn_relayout_launch does that to [Wp.W2], as attention.hip does for V)
//       S2: out += [Wp.W2] chunk . p  (2 stages of 160 output rows).
//   * only WEIGHTS go through LDS: a ring of 7 slots x 20 KiB filled by LDS-DMA six stages ahead; a stage = a [128 | 160 rows][64 k] slab
//     (1-KiB pieces of 8 rows, XOR-swizzled 16-B chunks); one s_barrier per stage; every counted s_waitcnt vmcnt is an immediate, because
//     the stage sequence is a compile-time function of the position in the chunk.  The wait + barrier for stage g+1 sit in front of the
//     last k16 step of stage g, whose MFMAs cover the first fragment reads of stage g+1.  A k16 step is one scheduling region:
//     sched_group_barrier puts the next step's LDS reads behind its first MFMA and the LDS-DMA piece + address arithmetic in the gaps.
// The output's GroupNorm partial sums (bins of 5 channels, 64-row slabs: what fuse_stats() files for C = 320) come out of the epilogue.
#include "ops.h"
#include <type_traits>

namespace {

constexpr int FC = 320;            // channels
constexpr int FH = 1280;           // hidden
constexpr int FTM = 128;           // rows per workgroup
constexpr int FSLOT = 20 * 1024;   // ring slot
constexpr int FNSLOT = 7;
constexpr int FNCH = FH / 64;
constexpr int FBIAS_OFF = FNSLOT * FSLOT;                 // b1 table [20][128] fp32 behind the ring
constexpr int FBO2_OFF = FBIAS_OFF + FNCH * 128 * 4;  // PRE: attn2.to_out's bias [320] fp32
constexpr int FW_BYTES = 2560;     // PRE: one 32 x 80 B staging patch per wave (residual rows on their way to the accumulator layout), in ring
                                   // slot 6 — free until stage 0 starts stage 6's LDS-DMA; a barrier separates the two uses
constexpr int FSMEM = FBIAS_OFF + FNCH * 128 * 4;
constexpr int FSMEM_PRE = FBO2_OFF + FC * 4;
static_assert(FSMEM_PRE <= 160 * 1024, "LDS budget");
constexpr int FKO = 384;           // PRE: K of attn2.to_out (8 heads x padded dim 48)
constexpr int FNPRE = 2 * (FKO / 64);      // its stages: 2 row halves x 6 k slabs

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int N> __device__ __forceinline__ void ffn_wait_vm() {
  // gfx9 s_waitcnt: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt left open
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

struct Slab { const bf16_t* src; int ld; };   // rows x 64 k starting at src, row stride ld (elements)

// PRE: the block's attn2.to_out + residual run in front, inside the kernel — t = Wo . o + bo + t_prev never leaves the registers (see the
// prologue); the ring slots of every later stage shift by FNPRE mod 7.
template <bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(const FfnArgs p) {
  kernarg_warm<sizeof(FfnArgs)>();
  constexpr int SB = PRE ? FNPRE % 7 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * FTM + w * 32 + l31;            // this lane's row (M % 128 == 0: always valid)

  // ---- LDS-DMA staging: piece pc of a stage covers slab rows 8pc .. 8pc+7 (1 KiB); lane: row 8pc + (lane >> 3), physical 16-B chunk
  // lane & 7 holds logical chunk (lane & 7) ^ (row & 7).  Wave w issues pieces w, w+4, w+8, ... (4 per wave for 128 rows, 5 for 160).
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  // stage sequence: g = 0..9 final stages F f (row half f / 5, k slab f % 5 of Wp); then g = 10 + 7c + q: chunk c, q = 0..4 S1 (k slab q of
  // W1c's 128 chunk rows), q = 5, 6 S2 (row half q - 5 of W2p's 64 hidden columns of the chunk).  Ring slot = g mod 7.
  auto slab_f = [&](int f) -> Slab {
    if constexpr (PRE) return Slab{p.Wpp + (size_t)(f / 5) * 160 * FC + 64 * (f % 5), FC};
    else return Slab{p.Wfo + (size_t)(f / 5) * 160 * (5 * FC) + 4 * FC + 64 * (f % 5), 5 * FC};
  };
  auto slab_g = [&](int g) -> Slab { return Slab{p.Wo + (size_t)(g / 6) * 160 * FKO + 64 * (g % 6), FKO}; };     // PRE only
  auto slab_c = [&](int c, int q) -> Slab {
    if (q < 5) return Slab{p.W1c + (size_t)c * 128 * FC + 64 * q, FC};
    return Slab{p.W2p + (size_t)(q - 5) * 160 * FH + 64 * c, FH};
  };
  auto issue_piece = [&](const Slab& st, int slot, int i) {      // piece i (0..) of this wave's share; no branch: see the prototype's note
    const int pc = w + 4 * i;
    // uniform base (scalar arithmetic, SGPR pair) + one 32-bit per-lane offset: the saddr form of the load — no 64-bit per-piece address
    // registers to keep (the compiler hoists them: 60+ VGPRs) or to compute in the MFMA gaps
    // (readfirstlane pins the base in SGPRs: left alone, the compiler re-associates it into hoisted 64-bit per-lane addresses)
    const uint64_t b64 = (uint64_t)(uintptr_t)(st.src + (size_t)(8 * pc) * st.ld);
    const bf16_t* base = (const bf16_t*)(uintptr_t)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32)) << 32) |
                                                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b64));
    const bf16_t* src = (const bf16_t*)((const char*)base + (unsigned)((srow * st.ld + schunk * 8) * 2));     // (a 32-bit BYTE offset)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + slot * FSLOT + pc * 1024), 16, 0, 0);
  };
  // A fragment of n-tile `tile` (32 rows) at k16 step s of the staged slab: row = tile * 32 + l31, logical chunk 2 s + hi
  auto wfrag = [&](const unsigned char* slot, int tile, int s) -> bf16x8 {
    const int row = tile * 32 + l31;
    return *reinterpret_cast<const bf16x8*>(slot + row * 128 + (((2 * s + hi) ^ (row & 7)) * 16));
  };

  // the first six stages' LDS-DMA goes out FIRST, the operand loads behind it: one memory latency in front of the stream instead of two
#pragma unroll
  for (int f = 0; f < 6; ++f) {
    const Slab st = PRE ? slab_g(f) : slab_f(f);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_piece(st, f, i);
  }
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 tf[20];               // t fragments (B operand): k16 step s -> t[m][16 s + 8 hi .. +8]  (PRE: in the permuted k order of its producer)
  float ln_mean, ln_rstd;      // LayerNorm factors of the row
  // PRE: the operands of t = Wo . o + bo + t_prev — o as B fragments (k16 step s -> o[m][16 s + 8 hi .. +8]), the residual rows as 16 B
  // per lane over 64-B runs (row lane / 4 (+ 16), part lane % 4 of each 32-column tile; through the wave's LDS patch into the
  // accumulator layout below)
  bf16x8 xf[PRE ? FKO / 16 : 1];
  u32x4 res[PRE ? 20 : 1];
  if constexpr (PRE) {
    const bf16_t* tsrc = p.T + (size_t)(blockIdx.x * FTM + w * 32 + (lane >> 2)) * FC + (lane & 3) * 8;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      res[2 * i] = *reinterpret_cast<const u32x4*>(tsrc + 32 * i);
      res[2 * i + 1] = *reinterpret_cast<const u32x4*>(tsrc + 32 * i + 16 * FC);
    }
#pragma unroll
    for (int s = 0; s < FKO / 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(p.X + (size_t)m * FKO + 16 * s + 8 * hi);
  } else {
#pragma unroll
    for (int s = 0; s < 20; ++s) tf[s] = *reinterpret_cast<const bf16x8*>(p.T + (size_t)m * FC + 16 * s + 8 * hi);
    // (as ln_row_factors() in gemm.hip: planes added in plane order)
    const int R = p.ln_rows ? p.ln_rows : p.M;
    const int mr = m >= R ? m - R : m;
    const float* base = p.ln_stats + (size_t)mr * 2;
    const size_t pstride = (size_t)R * 2;
    float2 st = make_float2(0.f, 0.f);
    for (int pl = 0; pl < p.ln_planes; ++pl) {
      const float2 v = *reinterpret_cast<const float2*>(base + (size_t)pl * pstride);
      st.x += v.x; st.y += v.y;
    }
    ln_mean = st.x * (1.f / FC);
    const float var = fmaxf(st.y * (1.f / FC) - ln_mean * ln_mean, 0.f);
    ln_rstd = rsqrtf(var + p.ln_eps);
  }
  // b1' table -> LDS
  {
    float* bt = reinterpret_cast<float*>(smem + FBIAS_OFF);
    static_assert(FNCH * 128 % 256 == 0, "table fill: whole rounds of the workgroup");
    float tb[FNCH * 128 / 256];
#pragma unroll
    for (int j = 0; j < FNCH * 128 / 256; ++j) tb[j] = p.b1c[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < FNCH * 128 / 256; ++j) bt[tid + 256 * j] = tb[j];
    if constexpr (PRE) {
      float* b2t = reinterpret_cast<float*>(smem + FBO2_OFF);
      const float v0 = p.bo2[tid], v1 = p.bo2[min(tid + 256, FC - 1)];
      b2t[tid] = v0; b2t[min(tid + 256, FC - 1)] = v1;
    }
  }

  // one explicit full wait: it also covers the t fragments and the row sums (without it the compiler cannot prove inside the loop that
  // those registers have landed and fences the first MFMA of EVERY stage with vmcnt(0))
  ffn_wait_vm<0>();

  bf16x8 wf[2][5];
  // NTL / NTN: 32-row tiles of this / the next stage (4 | 5; NTN = 0: last stage); WAITN: pieces that may stay outstanding when the next
  // stage must have landed (those of stages g+2..g+5 plus the three of stage g+6 issued by then); NXP: pieces per wave of stage g+6 (0: none)
  auto run_stage = [&](auto ntile_tag, auto ntile_next_tag, auto wait_tag, auto nxp_tag, int slot_idx, int next_slot_idx, const Slab& nx,
                       int nx_slot, auto&& mma, auto phase_end_tag) {
    // PHASE_END: register-hungry code follows — the next stage's first fragments are read after it, not kept live across it
    constexpr bool PHASE_END = decltype(phase_end_tag)::value;
    constexpr int NTL = decltype(ntile_tag)::value;
    constexpr int NTN = decltype(ntile_next_tag)::value;
    constexpr int WAITN = decltype(wait_tag)::value;
    constexpr int NXP = decltype(nxp_tag)::value;
    const unsigned char* slot = smem + slot_idx * FSLOT;
    const unsigned char* nslot = smem + next_slot_idx * FSLOT;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s == 3 && NTN > 0) {
        ffn_wait_vm<WAITN>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s + 1 < 4) {
#pragma unroll
        for (int q = 0; q < NTL; ++q) wf[(s + 1) & 1][q] = wfrag(slot, q, s + 1);
      } else if (NTN > 0 && !PHASE_END) {
#pragma unroll
        for (int q = 0; q < NTN; ++q) wf[0][q] = wfrag(nslot, q, 0);
      }
      if (NXP > 0) { issue_piece(nx, nx_slot, s); if (s == 3 && NXP == 5) issue_piece(nx, nx_slot, 4); }
      mma(s, wf[s & 1]);
#pragma unroll
      for (int q = 0; q < NTL; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        // 1 MFMA
        if (q == 0 && (s + 1 < 4 || (NTN > 0 && !PHASE_END))) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // the next step's reads behind the first MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                                        // up to 3 VALU / SALU
        if (q == 2 && NXP > 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 // the LDS-DMA piece
        if (q == 3 && NXP == 5 && s == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // (the fifth piece)
      }
    }
  };
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

  // prologue of the stream: stage 0 has landed; barrier (also: the bias table is in LDS); its first fragments
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the bias table's ds_writes
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int q = 0; q < 5; ++q) wf[0][q] = wfrag(smem, q, 0);

  if constexpr (PRE) {
    // ---- attn2.to_out + residual: acc1[10 tiles of 32 columns] = bo2 + t_prev + Wo . o  (12 stages of 160 rows of Wo), rounded to bf16 as
    // the residual stream is everywhere else; LayerNorm statistics of the rounded row (160 values in this lane + the other half wave's);
    // the rounded values, packed, ARE the t fragments under the permuted k order (accumulator registers 8hh .. 8hh+7 = the k slots of a
    // 32x32x16 B fragment: lnproj.hip) — Wp and W1' are read in that order (ffn_relayout_launch, kperm).  t itself is never stored.
    f32x16 acc1[10];
    {
      unsigned char* wb = smem + 6 * FSLOT + w * FW_BYTES;
      const float* b2t = reinterpret_cast<const float*>(smem + FBO2_OFF);
      const unsigned wb_wr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wb + (lane >> 2) * 80 + (lane & 3) * 16;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        // (inline asm: a compiler-visible LDS store is made to wait for every LDS-DMA in flight; LDS runs a wave's accesses in order)
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1280" ::"v"(wb_wr), "v"(res[2 * i]), "v"(res[2 * i + 1]) : "memory");
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b2t + 32 * i + 8 * q4 + 4 * hi);
          const u32x2 r = *reinterpret_cast<const u32x2*>(wb + l31 * 80 + q4 * 16 + hi * 8);
          acc1[i][4 * q4] = bv.x + __uint_as_float(r.x << 16);
          acc1[i][4 * q4 + 1] = bv.y + __uint_as_float(r.x & 0xffff0000u);
          acc1[i][4 * q4 + 2] = bv.z + __uint_as_float(r.y << 16);
          acc1[i][4 * q4 + 3] = bv.w + __uint_as_float(r.y & 0xffff0000u);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // every wave is done with its patch in slot 6: stage 0 may start filling it
    __builtin_amdgcn_sched_barrier(0);
#define G_STAGE(G)                                                                                                               \
    {                                                                                                                            \
      auto mma = [&](int s, const bf16x8* wfp) {                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 5; ++q)                                                                           \
          acc1[5 * ((G) / 6) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], xf[4 * ((G) % 6) + s], acc1[5 * ((G) / 6) + q], 0, 0, 0); \
      };                                                                                                                         \
      const Slab nx = (G) + 6 < FNPRE ? slab_g((G) + 6) : slab_f((G) + 6 - FNPRE);                                               \
      run_stage(I5{}, I5{}, std::integral_constant<int, 23>{}, I5{}, (G) % 7, ((G) + 1) % 7, nx, ((G) + 6) % 7, mma,             \
                std::integral_constant<bool, (G) + 1 == FNPRE>{});                                                               \
    }
    G_STAGE(0) G_STAGE(1) G_STAGE(2) G_STAGE(3) G_STAGE(4) G_STAGE(5) G_STAGE(6) G_STAGE(7) G_STAGE(8) G_STAGE(9) G_STAGE(10) G_STAGE(11)
#undef G_STAGE
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const unsigned o = pack_bf2(acc1[i][8 * hh + 2 * jj], acc1[i][8 * hh + 2 * jj + 1]);
          const float v0 = __uint_as_float(o << 16), v1 = __uint_as_float(o & 0xffff0000u);
          sum += v0 + v1;
          sq += v0 * v0 + v1 * v1;
          pk.u[jj] = o;
        }
        tf[2 * i + hh] = pk.v;
      }
    sum += __shfl_xor(sum, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    ln_mean = sum * (1.f / FC);
    ln_rstd = rsqrtf(fmaxf(sq * (1.f / FC) - ln_mean * ln_mean, 0.f) + p.ln_eps);
#pragma unroll
    for (int q = 0; q < 5; ++q) wf[0][q] = wfrag(smem + (SB % 7) * FSLOT, q, 0);
  }

  __builtin_amdgcn_sched_barrier(0);
  f32x16 out[10];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[i][r] = 0.f;

  // ---- final segment first: out = Wp . t on the raw rows.  WAITN for f = 0..9: 23 23 23 23 23 22 21 20 19 19
#define F_STAGE(F, N)                                                                                                            \
  {                                                                                                                              \
    auto mma = [&](int s, const bf16x8* wfp) {                                                                                   \
      _Pragma("unroll") for (int q = 0; q < 5; ++q)                                                                             \
        out[5 * ((F) / 5) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], tf[4 * ((F) % 5) + s], out[5 * ((F) / 5) + q], 0, 0, 0); \
    };                                                                                                                           \
    const Slab nx = (F) + 6 < 10 ? slab_f(((F) + 6) % 10) : slab_c(0, ((F) + 6 - 10) % 7);                                       \
    run_stage(I5{}, std::integral_constant<int, (F) + 1 < 10 ? 5 : 4>{}, std::integral_constant<int, N>{},                       \
              std::integral_constant<int, ((F) + 6 < 10 || (F) + 6 - 10 >= 5) ? 5 : 4>{}, (SB + (F)) % 7, (SB + (F) + 1) % 7, nx,            \
              (SB + (F) + 6) % 7, mma, std::false_type{});                                                                                               \
  }
  F_STAGE(0, 23) F_STAGE(1, 23) F_STAGE(2, 23) F_STAGE(3, 23) F_STAGE(4, 23)
  F_STAGE(5, 22) F_STAGE(6, 21) F_STAGE(7, 20) F_STAGE(8, 19) F_STAGE(9, 19)
#undef F_STAGE

  // ---- LayerNorm the resident fragments in place: t -> (t - mean) * rstd, rounded to bf16 (gain / shift live in W1' / b1')
#pragma unroll
  for (int s = 0; s < 20; ++s) {
    union { bf16x8 v; unsigned u[4]; } x;
    x.v = tf[s];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // (PRE: opaque to the optimiser — it otherwise recognises the unpacked values of the statistics pass above, keeps those 160 floats
      // alive across the ten F stages instead of unpacking again, and spills them to scratch)
      if constexpr (PRE) asm volatile("" : "+v"(x.u[j]));
      const float a = (__uint_as_float(x.u[j] << 16) - ln_mean) * ln_rstd;
      const float b = (__uint_as_float(x.u[j] & 0xffff0000u) - ln_mean) * ln_rstd;
      x.u[j] = pack_bf2(a, b);
    }
    tf[s] = x.v;
  }
  const float* btab = reinterpret_cast<const float*>(smem + FBIAS_OFF);

  // one chunk; LAST (compile-time): nothing is left to prefetch behind position 0
  auto run_chunk = [&](int c, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // slot of position q: (SB + 3 + q) mod 7.  WAITN by position: 20 21 21 21 20 19 19; last chunk: 20 18 14 10 5 0 -
#define S1_STAGE(J, N, NL)                                                                                                      \
    {                                                                                                                            \
      auto mma = [&](int s, const bf16x8* wfp) {                                                                                 \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                        \
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[nt], tf[4 * (J) + s], acc[nt], 0, 0, 0);                         \
      };                                                                                                                         \
      const Slab nx = (J) == 0 ? slab_c(c, 6) : slab_c(LAST ? c : c + 1, (J) - 1);                                               \
      run_stage(I4{}, std::integral_constant<int, (J) == 4 ? 5 : 4>{}, std::integral_constant<int, LAST ? NL : N>{},             \
                std::integral_constant<int, (J) == 0 ? 5 : (LAST ? 0 : 4)>{}, (SB + 3 + (J)) % 7, (SB + 4 + (J)) % 7, nx,        \
                (SB + 2 + (J)) % 7, mma, std::false_type{});                                                                     \
    }
    S1_STAGE(0, 20, 20) S1_STAGE(1, 21, 18) S1_STAGE(2, 21, 14) S1_STAGE(3, 21, 10) S1_STAGE(4, 20, 5)
#undef S1_STAGE
    // ---- bias, GEGLU in registers: tiles 0, 1 = value of hidden tiles 0, 1; tiles 2, 3 = their gates.  Lane (m, hi) holds of a tile the
    // rows n = (r & 3) + 8 (r >> 2) + 4 hi: four runs of 4 consecutive
    bf16x8 pf[4];          // B fragments of the 4 k16 steps (tt, hh) of S2
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {      // register run r = 8 hh + 4 g2 .. +3  <->  rows 16 hh + 8 g2 + 4 hi .. +3
          // (a NATIVE vector type: a float4 is a struct, its LDS load carries no alias info, and the waitcnt pass then makes it wait for
          // every LDS-DMA in flight — s_waitcnt vmcnt(0), the ring drained once per chunk)
          const f32x4 bv = *reinterpret_cast<const f32x4*>(btab + c * 128 + tt * 32 + 16 * hh + 8 * g2 + 4 * hi);
          const f32x4 bg = *reinterpret_cast<const f32x4*>(btab + c * 128 + 64 + tt * 32 + 16 * hh + 8 * g2 + 4 * hi);
          const int r = 8 * hh + 4 * g2;
          const float o0 = (acc[tt][r] + bv.x) * gelu_logistic(acc[2 + tt][r] + bg.x);
          const float o1 = (acc[tt][r + 1] + bv.y) * gelu_logistic(acc[2 + tt][r + 1] + bg.y);
          const float o2 = (acc[tt][r + 2] + bv.z) * gelu_logistic(acc[2 + tt][r + 2] + bg.z);
          const float o3 = (acc[tt][r + 3] + bv.w) * gelu_logistic(acc[2 + tt][r + 3] + bg.w);
          pk.u[2 * g2] = pack_bf2(o0, o1);
          pk.u[2 * g2 + 1] = pack_bf2(o2, o3);
        }
        pf[2 * tt + hh] = pk.v;
      }
    // ---- S2: out += [Wp.W2] chunk . p
#define S2_STAGE(U, N, NL)                                                                                                      \
    {                                                                                                                            \
      auto mma = [&](int s, const bf16x8* wfp) {                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 5; ++q)                                                                           \
          out[5 * (U) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], pf[s], out[5 * (U) + q], 0, 0, 0);                  \
      };                                                                                                                         \
      const Slab nx = slab_c(LAST ? c : c + 1, 4 + (U));                                                                         \
      run_stage(I5{}, std::integral_constant<int, (U) == 0 ? 5 : (LAST ? 0 : 4)>{}, std::integral_constant<int, LAST ? NL : N>{},\
                std::integral_constant<int, LAST ? 0 : ((U) == 1 ? 5 : 4)>{}, (SB + 8 + (U)) % 7, (SB + 9 + (U)) % 7, nx,        \
                (SB + 7 + (U)) % 7, mma, std::false_type{});                                                                     \
    }
    S2_STAGE(0, 19, 0) S2_STAGE(1, 19, 0)
#undef S2_STAGE
  };
  for (int c = 0; c + 1 < FNCH; ++c) run_chunk(c, std::false_type{});
  run_chunk(FNCH - 1, std::true_type{});

  // ---- epilogue: + bias + outer residual, bf16.  Lane (m, hi) holds columns n = tile * 32 + 8 q4 + 4 hi .. +3 of its row: moved in that
  // layout, the residual / the output are 8-B pieces of 32 different rows per instruction, and with one wave per SIMD (every workgroup of
  // the launch in the same phase) the memory pipeline's time for them is fully exposed.  The ring is dead: both go through an LDS tile
  // (rows OSTR B apart), as 16 B per lane over the wave's 32 x 640 B = 20 KiB contiguous rows.  The tile then also feeds the GroupNorm
  // partial sums of the output: per (64-row slab, column) over the rows, then per bin of 5 columns — every sum in a fixed order, written
  // once: the layout fuse_stats() consumers read (GemmArgs::gn_stats).
  constexpr int OSTR = 656;                      // 16-B aligned rows; 164 words: the 32 row lanes of an 8-byte access spread over the banks
  __syncthreads();                               // every wave is done with the ring
  unsigned char* wtile = smem + (size_t)(w * 32) * OSTR;          // this wave's 32 rows
  unsigned char* otile = wtile + (size_t)l31 * OSTR;
  {
    const bf16_t* rsrc = p.resid + (size_t)(blockIdx.x * FTM + w * 32) * FC;
    u32x4 rr[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) rr[j] = *reinterpret_cast<const u32x4*>(rsrc + (size_t)(lane + 64 * j) * 8);
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int c = lane + 64 * j;               // 16-B chunk c of the 32 x 40: row c / 40, part c % 40
      *reinterpret_cast<u32x4*>(wtile + (c / 40) * OSTR + (c % 40) * 16) = rr[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int n = i * 32 + 8 * q4 + 4 * hi;
      const f32x4 bo = *reinterpret_cast<const f32x4*>(p.bo + n);
      const u32x2 rv = *reinterpret_cast<const u32x2*>(otile + n * 2);
      const float v0 = out[i][4 * q4] + bo.x + __uint_as_float(rv.x << 16);
      const float v1 = out[i][4 * q4 + 1] + bo.y + __uint_as_float(rv.x & 0xffff0000u);
      const float v2 = out[i][4 * q4 + 2] + bo.z + __uint_as_float(rv.y << 16);
      const float v3 = out[i][4 * q4 + 3] + bo.w + __uint_as_float(rv.y & 0xffff0000u);
      *reinterpret_cast<u32x2*>(otile + n * 2) = u32x2{pack_bf2(v0, v1), pack_bf2(v2, v3)};
    }
  }
  {
    bf16_t* odst = p.out + (size_t)(blockIdx.x * FTM + w * 32) * FC;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int c = lane + 64 * j;
      *reinterpret_cast<u32x4*>(odst + (size_t)c * 8) = *reinterpret_cast<const u32x4*>(wtile + (c / 40) * OSTR + (c % 40) * 16);
    }
  }
  if (p.gn_stats) {
    __syncthreads();
    float2* colsum = reinterpret_cast<float2*>(smem + 128 * OSTR);        // [2 slabs][320 columns] {sum, sum of squares}
    for (int idx = tid; idx < 2 * FC; idx += 256) {
      const int slab = idx / FC, col = idx - slab * FC;
      const unsigned char* src = smem + (size_t)(slab * 64) * OSTR + col * 2;
      float a = 0.f, q = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) {
        const float v = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(src + (size_t)r * OSTR) << 16);
        a += v; q += v * v;
      }
      colsum[idx] = make_float2(a, q);
    }
    __syncthreads();
    {
      // 2 slabs x 64 bins x 2 moments = 256 sums, one per thread; bin = 5 columns (C / 64: the bins fuse_stats() files for C = 320)
      const int which = tid & 1, bin = (tid >> 1) & 63, slab = tid >> 7;
      const float* cs = reinterpret_cast<const float*>(colsum + slab * FC + bin * 5) + which;
      const float a = (((cs[0] + cs[2]) + cs[4]) + cs[6]) + cs[8];
      const int m0 = blockIdx.x * FTM + slab * 64;
      const int b = m0 / p.rows_per_batch;
      const int sl = (m0 - b * p.rows_per_batch) / 64;
      const int nslab = p.rows_per_batch / 64;
      p.gn_stats[(((size_t)b * nslab + sl) * 64 + bin) * 2 + which] = a;
    }
  }
}

// W1c[c][0..63 | 64..127][k] := the value | gate rows of hidden 64c .. 64c+63 of wff1 (16-row interleaved GEGLU layout, LayerNorm-folded);
// b1c likewise; W2p[n][pos] := wfo[n][perm(pos)], position 16 g + 8 hi + j <- hidden 16 g + 8 (j >> 2) + 4 hi + (j & 3)
// kperm (the PRE kernel's t fragments come out of accumulators): the k order of W1c and of Wpp — Wp = wfo's last 320 columns, as its own
// [320][320] matrix — is permuted within every 16 the same way
__global__ __launch_bounds__(256) void ffn_relayout_kernel(const bf16_t* __restrict__ wff1, const float* __restrict__ bff1,
                                                           const bf16_t* __restrict__ wfo, bf16_t* __restrict__ W1c,
                                                           float* __restrict__ b1c, bf16_t* __restrict__ W2p, bf16_t* __restrict__ Wpp) {
  const int64_t n1 = (int64_t)2 * FH * FC, n2 = (int64_t)FC * FH, n3 = Wpp ? (int64_t)FC * FC : 0;
  auto perm16 = [](int pos) { const int g16 = pos / 16, r = pos % 16, hi = r / 8, j = r % 8; return 16 * g16 + 8 * (j >> 2) + 4 * hi + (j & 3); };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + n3; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int k = (int)(i % FC);
      const int row = (int)(i / FC);                 // destination row: c * 128 + (gate ? 64 : 0) + j
      const int c = row / 128, rr = row % 128, gate = rr / 64, j = rr % 64;
      const int h = c * 64 + j;
      const int src = (h / 16) * 32 + gate * 16 + (h % 16);
      W1c[i] = wff1[(size_t)src * FC + (Wpp ? perm16(k) : k)];
      if (k == 0) b1c[row] = bff1[src];
    } else if (i < n1 + n2) {
      const int64_t t = i - n1;
      const int pos = (int)(t % FH), n = (int)(t / FH);
      W2p[t] = wfo[(size_t)n * (5 * FC) + perm16(pos)];
    } else {
      const int64_t t = i - n1 - n2;
      const int pos = (int)(t % FC), n = (int)(t / FC);
      Wpp[t] = wfo[(size_t)n * (5 * FC) + 4 * FC + perm16(pos)];
    }
  }
}

}  // namespace

bool ffn_fused_supported(int C, int M) { return C == FC && M % FTM == 0 && M > 0; }

int ffn_relayout_launch(const bf16_t* wff1, const float* bff1, const bf16_t* wfo, bf16_t* W1c, float* b1c, bf16_t* W2p, bf16_t* Wpp,
                        hipStream_t s) {
  hipLaunchKernelGGL(ffn_relayout_kernel, dim3(1024), dim3(256), 0, s, wff1, bff1, wfo, W1c, b1c, W2p, Wpp);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

int ffn_fused_launch(const FfnArgs& a, hipStream_t s) {
  GILL_REQUIRE(ffn_fused_supported(FC, a.M), "fused feed-forward block: M must be a multiple of 128");
  GILL_REQUIRE(a.T && a.W1c && a.b1c && a.W2p && a.bo && a.resid && a.out, "fused feed-forward block: null operand");
  GILL_REQUIRE(!a.gn_stats || (a.rows_per_batch > 0 && a.rows_per_batch % 128 == 0), "fused feed-forward block: GroupNorm partials need whole 64-row slabs per sample");
  const bool pre = a.X != nullptr;
  if (pre) {
    GILL_REQUIRE(a.Wo && a.bo2 && a.Wpp, "fused feed-forward block with attn2.to_out in front: null operand");
  } else {
    GILL_REQUIRE(a.Wfo && a.ln_stats && a.ln_planes >= 1, "fused feed-forward block: LayerNorm row-sum planes missing");
  }
  static bool attr_set[2] = {false, false};
  const void* fn = pre ? (const void*)ffn_fused_kernel<true> : (const void*)ffn_fused_kernel<false>;
  const int smem = pre ? FSMEM_PRE : FSMEM;
  if (!attr_set[pre]) {
    GILL_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[pre] = true;
  }
  if (pre) hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3(a.M / FTM), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3(a.M / FTM), dim3(256), smem, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
