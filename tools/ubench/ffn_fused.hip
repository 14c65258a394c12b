// Prototype of the fused feed-forward block of a level-0 BasicTransformerBlock (C = 320, hidden 1280) — VERDICT r02 item 2(ii):
//   out[m][:] = [Wp.W2 | Wp] . [ h(m) | t(m) ],   h(m) = value(m) * gelu(gate(m)),   [value | gate](m) = W1 . t(m)
// as ONE kernel per 128-row tile, go / no-go against GEGLU + ffo back to back (97 + 50 us in the engine at M = 32768).
//
// Structure under test (what the production kernels cannot do inside 256 registers per wave):
//   * FOUR waves per workgroup, ONE per SIMD (amdgpu_waves_per_eu(1,1): the full 512-register file per wave), one workgroup per CU;
//     a wave owns 32 rows of the tile through the whole block, so nothing is exchanged between waves:
//       t fragments (32 rows x 320: 80 registers) loaded once, straight from global memory, as the B operand of 32x32x16 MFMAs;
//       per 64-wide hidden chunk: S1 acc[4 tiles: 2 value, 2 gate] (64 registers) -> GEGLU in registers -> the products ARE the B
//       operand of S2 (the accumulator registers r = 8hh .. 8hh+7 of a lane are the k slots of a 32x32x16 B fragment when the A
//       operand's K order is permuted [0-3][8-11][4-7][12-15] within every 16 — done once on the weights, as attention.hip does for V);
//       S2 accumulates into the 32 x 320 output (10 tiles, 160 registers).
//   * only WEIGHTS go through LDS: a ring of 7 slots x 20 KiB filled by LDS-DMA six stages ahead (100-120 KB in flight), one stage =
//     a [128 | 160 rows][64 k] slab; one s_barrier per stage; counted s_waitcnt vmcnt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ffn_fused.hip -o tools/ubench/ffn_fused && tools/ubench/ffn_fused
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define C_IN 320
#define HID 1280
#define TM 128
#define SLOT_BYTES (20 * 1024)
#define NSLOT 7
#define DEPTH 6

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ __forceinline__ float gelu_f(float x) {
#ifdef GELU_EXACT_FORM      // erf by Abramowitz-Stegun 7.1.26 (the engine's form): 14 VALU + exp + rcp
  const float z = x * 0.70710678118654752f, a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
  const float r = copysignf(fmaf(-p * t, e, 1.0f), z);
  return 0.5f * x * (1.0f + r);
#else                       // x * Phi(x), Phi as a logistic of an odd quintic (|error| <= 2.7e-5): 8 VALU + exp + rcp
  const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
  const float x2 = xc * xc;
  float p = fmaf(0.0010187963489443064f, x2, -0.10680364072322845f);
  p = fmaf(p, x2, -2.301090717315674f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xc));
#endif
}

template <int N> __device__ __forceinline__ void wait_vm_imm() {
  // gfx9 s_waitcnt: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt 7, lgkmcnt 15 left open
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
#define WV(k) case k: wait_vm_imm<k>(); break;
    WV(0) WV(1) WV(2) WV(3) WV(4) WV(5) WV(6) WV(7) WV(8) WV(9) WV(10) WV(11) WV(12) WV(13) WV(14) WV(15) WV(16) WV(17) WV(18) WV(19)
    WV(20) WV(21) WV(22) WV(23) WV(24) WV(25) WV(26) WV(27) WV(28) WV(29) WV(30)
#undef WV
    default: wait_vm_imm<0>(); break;
  }
}

// W1c: [HID/64 chunks][128 rows (64 value | 64 gate)][C_IN]; W2p: [320][HID] with the hidden index permuted within every 16;
// Wp: [320][C_IN].  STAGE LIST of a workgroup: chunk c -> 5 S1 stages (k slab j) + 2 S2 stages (row half u); then 10 final stages
// (row half u, k slab j) of Wp.
struct Stage { const bf16_t* src; int ld; int rows; };   // slab = rows x 64 k starting at src, row stride ld

__device__ long long g_stamp[8];
template <bool DO_GELU, int ABL>      // ABL: 0 full, 1 no MFMA work (DMA + waits + barriers only), 2 no LDS-DMA (MFMAs on stale LDS)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffn_fused_kernel(const bf16_t* __restrict__ T, const bf16_t* __restrict__ W1c, const bf16_t* __restrict__ W2p,
                      const bf16_t* __restrict__ Wp, bf16_t* __restrict__ O, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * TM + w * 32;
  const int m = m0 + l31;

  // ---- t fragments (B operand): k16 step s -> t[m][16 s + 8 hi .. +8]
  bf16x8 tf[20];
#pragma unroll
  for (int s = 0; s < 20; ++s) tf[s] = *reinterpret_cast<const bf16x8*>(T + (size_t)m * C_IN + 16 * s + 8 * hi);

  // ---- LDS-DMA staging: piece p of a stage covers slab rows 8p .. 8p+7 (1 KiB); lane: row 8p + (lane >> 3), physical 16-B chunk
  // lane & 7 holds logical chunk (lane & 7) ^ (row & 7).  Wave w issues pieces w, w+4, w+8, ...
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;     // (row & 7) == srow: piece bases are multiples of 8
  // Stage sequence of a workgroup: chunk c = 0..19 -> positions q = 0..4 (S1, k slab q of the chunk's W1 rows) and q = 5, 6 (S2, row half
  // q - 5 of W2p's 64 hidden columns of the chunk); then final stages f = 0..9 (row half f / 5, k slab f % 5 of Wp).  Ring slot =
  // stage index mod 7 = q (the final stages continue the count: f mod 7).  The stage DEPTH = 6 ahead of (c, q) is (c, 6) for q = 0 and
  // (c + 1, q - 1) otherwise — or the final stage q - 1 when c is the last chunk; everything is a compile-time function of q.
  constexpr int NCH = HID / 64;
  auto slab = [&](int c, int q) -> Stage {         // q < 7: chunk stage; q >= 7: final stage q - 7
    Stage st;
    if (q < 5) { st.src = W1c + (size_t)c * 128 * C_IN + 64 * q; st.ld = C_IN; st.rows = 128; }
    else if (q < 7) { st.src = W2p + (size_t)(q - 5) * 160 * HID + 64 * c; st.ld = HID; st.rows = 160; }
    else { const int f = q - 7; st.src = Wp + (size_t)(f / 5) * 160 * C_IN + 64 * (f % 5); st.ld = C_IN; st.rows = 160; }
    return st;
  };
  auto issue_piece = [&](const Stage& st, int slot, int i) {      // piece i (0..) of this wave's share
    const int p = w + 4 * i;         // pieces 0..3 of a wave exist in every stage (128 | 160 rows = 16 | 20 pieces); the caller asks for
    if (ABL == 2) return;            // piece 4 only where the slab has 160 rows — no branch here: a control-flow join in front of the
    const bf16_t* src = st.src + (size_t)(8 * p + srow) * st.ld + schunk * 8;     // MFMAs makes the compiler drain lgkmcnt
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + slot * SLOT_BYTES + p * 1024), 16, 0, 0);
  };
  // A fragment of n-tile `tile` (32 rows) at k16 step s of the staged slab: row = tile * 32 + l31, logical chunk 2 s + hi
  auto wfrag = [&](const unsigned char* slot, int tile, int s) -> bf16x8 {
    const int row = tile * 32 + l31;
    return *reinterpret_cast<const bf16x8*>(slot + row * 128 + (((2 * s + hi) ^ (row & 7)) * 16));
  };

  f32x16 out[10];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[i][r] = 0.f;

#pragma unroll
  for (int q = 0; q < DEPTH; ++q) {
    const Stage st = slab(0, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(st, q, i);
    if (q >= 5) issue_piece(st, q, 4);
  }
  // (one explicit full wait here — it also covers the t fragments: without it the compiler cannot prove inside the loop that those
  // registers have landed and fences the first MFMA of EVERY stage with vmcnt(0), i.e. waits for the whole prefetch each time)
  wait_vm_imm<0>();
  // One stage: counted wait for the own pieces (WAITN = pieces of the five stages behind it: an immediate), barrier, first fragments,
  // then per k16 step: prefetch the next step's fragments, issue ONE piece of the stage six ahead (the LDS-DMA issue rides between
  // the MFMA groups), counted lgkmcnt, 4 | 5 MFMAs.
  // NXP: pieces per wave of the stage six ahead (4 | 5; 0 = nothing left to issue) — compile-time, so that a step is straight-line code:
  // the compiler then places exact lgkmcnt waits per fragment, and sched_group_barrier can interleave the step's LDS reads, its LDS-DMA
  // piece and their address arithmetic INTO the gaps between its MFMAs (one wave per SIMD: whatever is not issued between two MFMAs
  // runs with the matrix pipe idle).
  // The fragment prefetch runs ACROSS the stage boundary: the counted wait + barrier for stage g+1 sit in front of step 3 of stage g, whose
  // MFMAs then cover the first reads of stage g+1 (NTN = its tiles; WAITN = pieces behind stage g+1 at that point: those of stages
  // g+2..g+5 plus the three of stage g+6 already issued).  The fragments of step 0 are in wf[0] when a stage starts.
  bf16x8 wf[2][5];
  auto run_stage = [&](auto ntile_tag, auto ntile_next_tag, auto wait_tag, auto nxp_tag, int slot_idx, int next_slot_idx,
                       const Stage& nx, int nx_slot, auto&& mma) {
    constexpr int NTL = decltype(ntile_tag)::value;        // 4 (S1: 128 rows) | 5 (S2 / final: 160 rows)
    constexpr int NTN = decltype(ntile_next_tag)::value;   // tiles of the next stage (0: this is the last stage)
    constexpr int WAITN = decltype(wait_tag)::value;
    constexpr int NXP = decltype(nxp_tag)::value;
    const unsigned char* slot = smem + slot_idx * SLOT_BYTES;
    const unsigned char* nslot = smem + next_slot_idx * SLOT_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_barrier(0);      // a step is one scheduling region: its groups pair MFMAs of step s with reads of step s + 1
      if (s == 3 && NTN > 0) {
        if (ABL != 2) wait_vm_imm<WAITN>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int NRD = 5;
      if (s + 1 < 4) {
#pragma unroll
        for (int q = 0; q < NTL; ++q) wf[(s + 1) & 1][q] = wfrag(slot, q, s + 1);
      } else if (NTN > 0) {
#pragma unroll
        for (int q = 0; q < NTN; ++q) wf[0][q] = wfrag(nslot, q, 0);
      }
      if (NXP > 0) { issue_piece(nx, nx_slot, s); if (s == 3 && NXP == 5) issue_piece(nx, nx_slot, 4); }
      if (ABL != 1) mma(s, wf[s & 1]);
#pragma unroll
      for (int q = 0; q < NTL; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
        if (q == 0 && (s + 1 < 4 || NTN > 0)) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);   // the next step's reads behind the first MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                      // up to 3 VALU / SALU
        if (q == 2 && NXP > 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 // VMEM read (the LDS-DMA piece)
        if (q == 3 && NXP == 5 && s == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // (the fifth piece)
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

  // prologue of the stream: stage 0 has landed (the full wait above); barrier; its first fragments
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) wf[0][q] = wfrag(smem, q, 0);

  // one chunk; LAST (compile-time): the stages six ahead are the final segment's
  auto run_chunk = [&](int c, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const bool stamp = (c == 10 && blockIdx.x == 7 && tid == 0);
    if (stamp) g_stamp[0] = clock64();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // ---- S1: [value | gate] chunk = W1 chunk . t.  WAITN by position 0..6: 20 21 21 21 20 19 19 (in the last chunk the stages behind are
    // 5-piece final stages: the same immediates then wait for a few pieces too many)
#define S1_STAGE(J, N)                                                                                                          \
    {                                                                                                                            \
      auto mma = [&](int s, const bf16x8* wfp) {                                                                                 \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                        \
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[nt], tf[4 * (J) + s], acc[nt], 0, 0, 0);                         \
      };                                                                                                                         \
      const Stage nx = (J) == 0 ? slab(c, 6) : (LAST ? slab(0, 7 + (J) - 1) : slab(c + 1, (J) - 1));                             \
      run_stage(I4{}, std::integral_constant<int, (J) == 4 ? 5 : 4>{}, std::integral_constant<int, N>{},                         \
                std::integral_constant<int, ((J) == 0 || LAST) ? 5 : 4>{}, (J), (J) + 1, nx, (J) == 0 ? 6 : (J) - 1, mma);       \
    }
    S1_STAGE(0, 20) S1_STAGE(1, 21) S1_STAGE(2, 21) S1_STAGE(3, 21) S1_STAGE(4, 20)
#undef S1_STAGE
    if (stamp) g_stamp[1] = clock64();
    // ---- GEGLU in registers: tiles 0, 1 = value of hidden tiles 0, 1; tiles 2, 3 = their gates
    bf16x8 pf[4];          // B fragments of the 4 k16 steps (tt, hh) of S2
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int r = 8 * hh + 2 * jj;
          const float a = DO_GELU ? acc[tt][r] * gelu_f(acc[2 + tt][r]) : acc[tt][r] + acc[2 + tt][r];
          const float b = DO_GELU ? acc[tt][r + 1] * gelu_f(acc[2 + tt][r + 1]) : acc[tt][r + 1] + acc[2 + tt][r + 1];
          pk.u[jj] = pack2(a, b);
        }
        pf[2 * tt + hh] = pk.v;
      }
    if (stamp) g_stamp[2] = clock64();
    // ---- S2: out += W2p chunk . p
#define S2_STAGE(U, N)                                                                                                          \
    {                                                                                                                            \
      auto mma = [&](int s, const bf16x8* wfp) {                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 5; ++q)                                                                           \
          out[5 * (U) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], pf[s], out[5 * (U) + q], 0, 0, 0);                  \
      };                                                                                                                         \
      const Stage nx = LAST ? slab(0, 7 + 4 + (U)) : slab(c + 1, 4 + (U));                                                       \
      run_stage(I5{}, std::integral_constant<int, ((U) == 0 || LAST) ? 5 : 4>{}, std::integral_constant<int, N>{},               \
                std::integral_constant<int, (LAST || (U) == 1) ? 5 : 4>{}, 5 + (U), (U) == 0 ? 6 : 0, nx, 4 + (U), mma);         \
    }
    S2_STAGE(0, 19) S2_STAGE(1, 19)
#undef S2_STAGE
    if (stamp) g_stamp[3] = clock64();
  };
  for (int c = 0; c + 1 < NCH; ++c) run_chunk(c, std::false_type{});
  run_chunk(NCH - 1, std::true_type{});
  // ---- final segment: out += Wp . t   (stage 140 + f sits in slot f mod 7; WAITN for f = 0..8: 23 23 23 23 20 15 10 5 0)
#define F_STAGE(F, N)                                                                                                            \
  {                                                                                                                              \
    auto mma = [&](int s, const bf16x8* wfp) {                                                                                   \
      _Pragma("unroll") for (int q = 0; q < 5; ++q)                                                                             \
        out[5 * ((F) / 5) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], tf[4 * ((F) % 5) + s], out[5 * ((F) / 5) + q], 0, 0, 0); \
    };                                                                                                                           \
    const Stage nx = slab(0, 7 + ((F) + 6 < 10 ? (F) + 6 : 0));                                                                  \
    run_stage(I5{}, std::integral_constant<int, (F) + 1 < 10 ? 5 : 0>{}, std::integral_constant<int, N>{},                       \
              std::integral_constant<int, ((F) + 6 < 10) ? 5 : 0>{}, (F) % 7, ((F) + 1) % 7, nx, ((F) + 6) % 7, mma);            \
  }
  F_STAGE(0, 23) F_STAGE(1, 23) F_STAGE(2, 23) F_STAGE(3, 23) F_STAGE(4, 20)
  F_STAGE(5, 15) F_STAGE(6, 10) F_STAGE(7, 5) F_STAGE(8, 0) F_STAGE(9, 0)
#undef F_STAGE
  // ---- store: lane (m, hi) holds n = tile * 32 + (r & 3) + 8 (r >> 2) + 4 hi
  if (m < M) {
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 o; o.x = pack2(out[i][4 * q4], out[i][4 * q4 + 1]); o.y = pack2(out[i][4 * q4 + 2], out[i][4 * q4 + 3]);
        *reinterpret_cast<uint2*>(O + (size_t)m * 320 + i * 32 + 8 * q4 + 4 * hi) = o;
      }
  }
}

// naive reference for the first `rows` rows (one thread per output)
__global__ void ffn_ref_kernel(const bf16_t* T, const bf16_t* W1c, const bf16_t* W2, const bf16_t* Wp, float* R, int rows, int do_gelu) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * 320) return;
  const int mrow = idx / 320, n = idx % 320;
  float acc = 0.f;
  for (int h = 0; h < HID; ++h) {
    const int c = h / 64, hh = h % 64;
    const bf16_t* wv = W1c + ((size_t)c * 128 + hh) * C_IN;
    const bf16_t* wg = W1c + ((size_t)c * 128 + 64 + hh) * C_IN;
    float v = 0.f, gt = 0.f;
    for (int k = 0; k < C_IN; ++k) { const float t = bf2f(T[(size_t)mrow * C_IN + k]); v += bf2f(wv[k]) * t; gt += bf2f(wg[k]) * t; }
    const float p = do_gelu ? v * (0.5f * gt * (1.f + erff(gt * 0.70710678f))) : v + gt;
    acc += bf2f(W2[(size_t)n * HID + h]) * bf2f(f2bf(p));
  }
  for (int k = 0; k < C_IN; ++k) acc += bf2f(Wp[(size_t)n * C_IN + k]) * bf2f(T[(size_t)mrow * C_IN + k]);
  R[idx] = acc;
}

static bf16_t h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float h_bf2f(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int M = 32768;
  std::vector<bf16_t> hT((size_t)M * C_IN), hW1((size_t)2 * HID * C_IN), hW2((size_t)320 * HID), hW2p((size_t)320 * HID), hWp((size_t)320 * C_IN);
  unsigned s = 777;
  auto rnd = [&](float scale) { s = s * 1664525u + 1013904223u; return ((int)(s >> 8 & 0xffff) - 32768) / 32768.0f * scale; };
  for (auto& v : hT) v = h_f2bf(rnd(1.0f));
  for (auto& v : hW1) v = h_f2bf(rnd(0.1f));
  for (auto& v : hW2) v = h_f2bf(rnd(0.05f));
  for (auto& v : hWp) v = h_f2bf(rnd(0.05f));
  // K order of W2 as the S2 A fragments read it: position 16 g + 8 hi + j holds hidden 16 g + 8 (j >> 2) + 4 hi + (j & 3)
  for (int n = 0; n < 320; ++n)
    for (int pos = 0; pos < HID; ++pos) {
      const int g16 = pos / 16, r = pos % 16, hi = r / 8, j = r % 8;
      hW2p[(size_t)n * HID + pos] = hW2[(size_t)n * HID + 16 * g16 + 8 * (j >> 2) + 4 * hi + (j & 3)];
    }
  bf16_t *T, *W1, *W2, *W2p, *Wp, *O; float* R;
  hipMalloc(&T, hT.size() * 2); hipMalloc(&W1, hW1.size() * 2); hipMalloc(&W2, hW2.size() * 2); hipMalloc(&W2p, hW2p.size() * 2);
  hipMalloc(&Wp, hWp.size() * 2); hipMalloc(&O, (size_t)M * 320 * 2); hipMalloc(&R, (size_t)256 * 320 * 4);
  hipMemcpy(T, hT.data(), hT.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W1, hW1.data(), hW1.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W2, hW2.data(), hW2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W2p, hW2p.data(), hW2p.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(Wp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
  const int lds = NSLOT * SLOT_BYTES;
  hipFuncSetAttribute((const void*)ffn_fused_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)ffn_fused_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)ffn_fused_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)ffn_fused_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const double flop = 2.0 * M * (2.0 * HID * C_IN + 320.0 * HID + 320.0 * C_IN);
  for (int variant = 0; variant < 4; ++variant) {
    auto launch = [&]() {
      if (variant == 0) hipLaunchKernelGGL((ffn_fused_kernel<true, 0>), dim3(M / TM), dim3(256), lds, 0, T, W1, W2p, Wp, O, M);
      else if (variant == 1) hipLaunchKernelGGL((ffn_fused_kernel<false, 0>), dim3(M / TM), dim3(256), lds, 0, T, W1, W2p, Wp, O, M);
      else if (variant == 2) hipLaunchKernelGGL((ffn_fused_kernel<true, 1>), dim3(M / TM), dim3(256), lds, 0, T, W1, W2p, Wp, O, M);
      else hipLaunchKernelGGL((ffn_fused_kernel<true, 2>), dim3(M / TM), dim3(256), lds, 0, T, W1, W2p, Wp, O, M);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // correctness: rows 0..255 against the naive kernel
    hipLaunchKernelGGL(ffn_ref_kernel, dim3((256 * 320 + 255) / 256), dim3(256), 0, 0, T, W1, W2, Wp, R, 256, variant == 0 ? 1 : 0);
    std::vector<float> hr((size_t)256 * 320); std::vector<bf16_t> ho((size_t)256 * 320);
    hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ho.data(), O, ho.size() * 2, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < hr.size(); ++i) { const double d = h_bf2f(ho[i]) - hr[i]; num += d * d; den += (double)hr[i] * hr[i]; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int Rn = 20;
    for (int i = 0; i < Rn; ++i) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / Rn;
    { long long st[8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st)); printf("   chunk 10 of block 7, wave 0 (s_memtime ticks): S1 5 stages %lld, GEGLU %lld, S2 2 stages %lld\n", st[1] - st[0], st[2] - st[1], st[3] - st[2]); }
    printf("fused FFN (C 320, hidden 1280, M %d, %s): %7.1f us per launch, %6.0f TFLOP/s; rel-L2 vs naive (256 rows) %.3e\n", M,
           variant == 0 ? "GEGLU" : variant == 1 ? "no GELU (value + gate)" : variant == 2 ? "ABLATION: no MFMAs" : "ABLATION: no LDS-DMA", us, flop / us / 1e6, sqrt(num / den));
  }
  return 0;
}
