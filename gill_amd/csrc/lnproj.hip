// Two-GEMM chains of a level-0 (C = 320) transformer block around a LayerNorm, one kernel per 128-row tile each:
//   MODE 0  "proj_in -> norm1 -> attn1.to_q/k/v":   t = Wpi . n + bpi;                 [q | k | v] = Wqkv' . LN(t) + c'   (head-major scatter)
//   MODE 1  "attn1.to_out -> + residual -> norm2 -> attn2.to_q":   t' = Wo . o + bo + t;   q = Wq' . LN(t') + c'
// (W' / c' carry the LayerNorm gain / shift: ln_fold_rows_launch; their K order is permuted within every 16: lnproj_kperm_launch.)
// They replace, per block, the proj_in GEMM + the QKV GEMM (13-18 + 58-62 us at M = 32768), resp. the to_out GEMM + the to_q GEMM
// (23 + 26 us), the row-sum planes between them and one launch each.
//
// Same structure as ffn.hip (see there and profiles/r03_ffn_fused_prototype.md): four waves per workgroup, ONE per SIMD with the whole
// 512-register file; a wave owns 32 rows through the chain.  The first GEMM's B operand (the rows of n / o) is loaded once, straight from
// global memory; its 32 x 320 result lives in the accumulators, gets bias (+ residual), is rounded and stored (the residual stream), and —
// every lane holding ONE row — is LayerNorm-ed right there (row sums: 160 values in the lane + one cross-half shuffle); the normalised
// values, packed, ARE the B operand of the second GEMM (accumulator registers 8hh .. 8hh+7 = the k slots of a 32x32x16 B fragment under
// the weights' permuted K order).  The second GEMM runs in segments of 384 output columns (192 accumulator registers) with the head-major
// scatter of gemm.hip's QKV epilogue after each.  Only weights go through LDS: ring of 7 x 20 KiB, LDS-DMA six stages ahead, one barrier
// per stage; the whole stage sequence is unrolled with a compile-time stage index, so every counted s_waitcnt vmcnt is an immediate.
#include "ops.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int LC = 320;
constexpr int LSEG = 384;          // columns of one segment of the second GEMM (8 heads x 48)
constexpr int LTM = 128;
constexpr int LDP = 48;            // padded head dim (8 heads)
constexpr int LSLOT = 20 * 1024;
constexpr int LNSLOT = 7;
constexpr int LDEPTH = 6;
constexpr int LBIAS_OFF = LNSLOT * LSLOT;     // fp32 tables behind the ring: b1 [320] | c2 [NSEG * 384]
constexpr int LW_BYTES = 3584;                // ... and one staging patch per wave for its stores (Q / K head tile: 32 x 112 B)
constexpr int LW_OFF = LBIAS_OFF + (LC + 3 * LSEG) * 4;

typedef __attribute__((ext_vector_type(16))) float f32x16;
// NATIVE vector types for every LDS read that is not an MFMA fragment: a load through one carries TBAA, and the waitcnt pass makes an LDS
// access WITHOUT alias info (float4 / uint4 are structs: none) wait for every LDS-DMA in flight — s_waitcnt vmcnt(0), the ring drained
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int N> __device__ __forceinline__ void lp_wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// ---- the stage sequence, as compile-time functions of (MODE, stage index g)
constexpr int lp_k1(int mode) { return mode == 0 ? 320 : 384; }
constexpr int lp_nseg(int mode) { return mode == 0 ? 3 : 1; }
constexpr int lp_g1(int mode) { return 2 * (lp_k1(mode) / 64); }                 // stages of the first GEMM: 2 row halves x K1 / 64 slabs
constexpr int lp_nst(int mode) { return lp_g1(mode) + lp_nseg(mode) * 15; }      // + segments x 3 row chunks x 5 slabs
constexpr int lp_pieces(int mode, int g) { return g < 0 || g >= lp_nst(mode) ? 0 : (g < lp_g1(mode) ? 5 : 4); }   // per wave
constexpr int lp_tiles(int mode, int g) { return lp_pieces(mode, g); }           // 160 rows = 5 tiles, 128 rows = 4
// pieces that may stay outstanding when stage g + 1 must have landed (in front of step 3 of stage g): all of stages g+2 .. g+5 and the
// three pieces of stage g + 6 issued in steps 0 .. 2
// ... and, vmcnt on gfx9 counting stores too (in issue order with the loads), the epilogue stores issued in that window: the 20 of the
// residual stream in front of stage G1, those of a segment's scatter in front of the next segment's first stage (6-bit counter: capped)
// segment computed in slot k of the second GEMM: mode 0 runs V, K, Q — the V^T tile is stored in 64-B runs (its rows are tokens), the Q / K
// tiles in 1-KiB runs; the scatter of the LAST segment has nothing to hide behind
constexpr int lp_seg(int mode, int k) { return mode == 0 ? 2 - k : 0; }
constexpr int lp_scatter_stores(int seg) { return seg == 2 ? 32 : 24; }
constexpr int lp_stores_before(int mode, int g) {
  if (g == lp_g1(mode)) return 20;
  if (g > lp_g1(mode) && g < lp_nst(mode) && (g - lp_g1(mode)) % 15 == 0) return lp_scatter_stores(lp_seg(mode, (g - lp_g1(mode)) / 15 - 1));
  return 0;
}
constexpr int lp_waitn(int mode, int g) {
  int n = lp_pieces(mode, g + 2) + lp_pieces(mode, g + 3) + lp_pieces(mode, g + 4) + lp_pieces(mode, g + 5) + (lp_pieces(mode, g + 6) ? 3 : 0);
  for (int h = g - 4; h <= g; ++h) n += lp_stores_before(mode, h);
  return n > 63 ? 63 : n;
}

// LDS writes of the per-wave staging patch as inline asm: for a compiler-visible LDS store the waitcnt pass assumes it may overwrite
// what an LDS-DMA in flight writes and drains the whole queue (s_waitcnt vmcnt(0)) in front of it.  LDS executes a wave's accesses in
// order, so the compiler-emitted reads behind these see the data; the "memory" clobber keeps them behind.
__device__ __forceinline__ void lds_write_b64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b16_pair(unsigned addr, unsigned v) {       // low half -> addr, high half -> addr + 64
  asm volatile("ds_write_b16 %0, %1\n\tds_write_b16_d16_hi %0, %1 offset:64" ::"v"(addr), "v"(v) : "memory");
}

struct Slab { const bf16_t* src; int ld; };

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lnproj_kernel(const LnProjArgs p) {
  kernarg_warm<sizeof(LnProjArgs)>();
  constexpr int K1 = lp_k1(MODE), NSEG = lp_nseg(MODE), G1 = lp_g1(MODE), NST = lp_nst(MODE), KS1 = K1 / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * LTM + w * 32 + l31;            // this lane's row (M % 128 == 0)

  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  auto slab_of = [&](int g) -> Slab {
    if (g < G1) { const int u = g / KS1, j = g - u * KS1; return Slab{p.W1 + (size_t)u * 160 * K1 + 64 * j, K1}; }
    const int h = g - G1, seg = h / 15, r = h - seg * 15, chunk = r / 5, j = r - chunk * 5;
    return Slab{p.W2p + (size_t)(lp_seg(MODE, seg) * LSEG + chunk * 128) * LC + 64 * j, LC};
  };
  auto issue_piece = [&](const Slab& st, int slot, int i) {
    const int pc = w + 4 * i;
    // uniform base (scalar arithmetic, SGPR pair) + one 32-bit per-lane offset: the saddr form of the load — no 64-bit per-piece address
    // registers to keep (the compiler hoists them: 60+ VGPRs) or to compute in the MFMA gaps
    // (readfirstlane pins the base in SGPRs: left alone, the compiler re-associates it into hoisted 64-bit per-lane addresses)
    const uint64_t b64 = (uint64_t)(uintptr_t)(st.src + (size_t)(8 * pc) * st.ld);
    const bf16_t* base = (const bf16_t*)(uintptr_t)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32)) << 32) |
                                                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b64));
    const bf16_t* src = (const bf16_t*)((const char*)base + (unsigned)((srow * st.ld + schunk * 8) * 2));     // (a 32-bit BYTE offset)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + slot * LSLOT + pc * 1024), 16, 0, 0);
  };
  auto wfrag = [&](const unsigned char* slot, int tile, int s) -> bf16x8 {
    const int row = tile * 32 + l31;
    return *reinterpret_cast<const bf16x8*>(slot + row * 128 + (((2 * s + hi) ^ (row & 7)) * 16));
  };

#pragma unroll
  for (int g = 0; g < LDEPTH; ++g) {
    const Slab st = slab_of(g);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < lp_pieces(MODE, g)) issue_piece(st, g % LNSLOT, i);
  }
  __builtin_amdgcn_sched_barrier(0);      // (pins the issue order the counted waits below rely on)
  // ---- operand loads BEHIND the first stages' LDS-DMA (one memory latency in front of the stream instead of two), oldest first in the
  // order they are needed: tables, residual rows, X fragments.
  // Branch-free (clamped index, the surplus threads rewrite the last entry): under a predicate the compiler sinks the load into the
  // branch, where it is the newest memory operation and its wait a full vmcnt(0).
  float tb1[2], tc2[(NSEG * LSEG + 255) / 256];
  // (mode 0 with the block's GroupNorm folded in: its scale / shift rows, 640 floats of this workgroup's sample — 128 rows, ntok % 128 == 0)
  float tgn[MODE == 0 ? 3 : 1];
  const bool gn_fold = MODE == 0 && p.gn_ss != nullptr;       // (uniform)
  if constexpr (MODE == 0) {
    if (gn_fold) {
      const float* ss = p.gn_ss + (size_t)((blockIdx.x * LTM) / p.ntok) * 2 * LC;
#pragma unroll
      for (int j = 0; j < 3; ++j) tgn[j] = ss[min(tid + 256 * j, 2 * LC - 1)];
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) tb1[j] = p.b1[min(tid + 256 * j, LC - 1)];
#pragma unroll
  for (int j = 0; j < (NSEG * LSEG + 255) / 256; ++j) tc2[j] = p.c2[min(tid + 256 * j, NSEG * LSEG - 1)];
  __builtin_amdgcn_sched_barrier(0);      // (the scheduler reorders independent loads freely: fences between the three groups)
  // MODE 1: the residual rows, 16 B per lane over 64-B runs (row lane / 4 (+ 16), part lane % 4 of each 32-column tile); they reach the
  // accumulator layout (lane (m, hi): columns 32 i + 8 q4 + 4 hi .. +3) through the wave's LDS patch
  u32x4 res[MODE == 1 ? 20 : 1];
  if constexpr (MODE == 1) {
    const bf16_t* tsrc = p.T + (size_t)(blockIdx.x * LTM + w * 32 + (lane >> 2)) * LC + (lane & 3) * 8;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      res[2 * i] = *reinterpret_cast<const u32x4*>(tsrc + 32 * i);
      res[2 * i + 1] = *reinterpret_cast<const u32x4*>(tsrc + 32 * i + 16 * LC);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // first GEMM's B operand: rows of X (n, or the attention output o), k16 step s -> X[m][16 s + 8 hi .. +8]
  bf16x8 xf[K1 / 16];
#pragma unroll
  for (int s = 0; s < K1 / 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(p.X + (size_t)m * K1 + 16 * s + 8 * hi);
  {
    float* bt = reinterpret_cast<float*>(smem + LBIAS_OFF);
    const unsigned bt_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)bt;
    // everything has landed.  (Starting the first stages with the X fragments still in flight would hide ~2 us, but with LDS-DMA
    // interleaved the compiler's own waits for loaded registers come out as vmcnt(0) instead of counted — measured in the ISA — and a full
    // wait inside stage 0 drains the ring.)
    lp_wait_vm<0>();
#pragma unroll
    for (int j = 0; j < 2; ++j) lds_write_b32(bt_lds + 4 * min(tid + 256 * j, LC - 1), tb1[j]);
#pragma unroll
    for (int j = 0; j < (NSEG * LSEG + 255) / 256; ++j) lds_write_b32(bt_lds + 4 * (LC + min(tid + 256 * j, NSEG * LSEG - 1)), tc2[j]);
    if constexpr (MODE == 0) {
      // the GroupNorm table sits in the LAST 4 KiB of the ring slot no stage has been issued into yet (slot LDEPTH): those bytes belong
      // to the fifth piece of stage LDEPTH, which every wave issues behind the first barrier of stage 0 — after all have read the table
      if (gn_fold) {
        const unsigned gt = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(smem + LDEPTH * LSLOT + 16 * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) lds_write_b32(gt + 4 * min(tid + 256 * j, 2 * LC - 1), tgn[j]);
      }
    }
  }
  const float* b1t = reinterpret_cast<const float*>(smem + LBIAS_OFF);
  const float* c2t = b1t + LC;

  bf16x8 wf[2][5];
  // one stage with a compile-time index G; mma(s, fragments) issues the MFMAs of k16 step s
  auto run_stage = [&](auto g_tag, auto&& mma) {
    constexpr int G = decltype(g_tag)::value;
    constexpr int NTL = lp_tiles(MODE, G), NTN = lp_tiles(MODE, G + 1), WAITN = lp_waitn(MODE, G), NXP = lp_pieces(MODE, G + LDEPTH);
    // last stage of a GEMM / segment: an epilogue follows — the next stage's first fragments are read after it (kept live across it, the
    // register allocator spills them to scratch, and a scratch access drains the whole LDS-DMA queue: vmcnt is in order)
    constexpr bool PHASE_END = G == G1 - 1 || (G >= G1 && (G - G1) % 15 == 14);
    const unsigned char* slot = smem + (G % LNSLOT) * LSLOT;
    const unsigned char* nslot = smem + ((G + 1) % LNSLOT) * LSLOT;
    const Slab nx = slab_of(NXP > 0 ? G + LDEPTH : G);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s == 3 && NTN > 0) {
        lp_wait_vm<WAITN>();
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
      if (NXP > 0) { issue_piece(nx, (G + LDEPTH) % LNSLOT, s); if (s == 3 && NXP == 5) issue_piece(nx, (G + LDEPTH) % LNSLOT, 4); }
      mma(s, wf[s & 1]);
#pragma unroll
      for (int q = 0; q < NTL; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q == 0 && (s + 1 < 4 || (NTN > 0 && !PHASE_END))) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
        if (q == 2 && NXP > 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (q == 3 && NXP == 5 && s == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };

  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the tables' ds_writes
  __builtin_amdgcn_s_barrier();
  if constexpr (MODE == 0) {
    if (gn_fold) {
      // x := bf16(x * scale[c] + shift[c]) on the fragments in registers: lane (m, hi) holds channels 16 s + 8 hi .. + 7 of k16 step s
      const float* gsc = reinterpret_cast<const float*>(smem + LDEPTH * LSLOT + 16 * 1024);
      const float* gsh = gsc + LC;
#pragma unroll
      for (int s = 0; s < K1 / 16; ++s) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(gsc + 16 * s + 8 * hi), s1 = *reinterpret_cast<const f32x4*>(gsc + 16 * s + 8 * hi + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(gsh + 16 * s + 8 * hi), h1 = *reinterpret_cast<const f32x4*>(gsh + 16 * s + 8 * hi + 4);
        union { bf16x8 v; uint32_t u[4]; } x;
        x.v = xf[s];
        x.u[0] = pack_bf2(fmaf(bf2f((bf16_t)(x.u[0] & 0xffff)), s0[0], h0[0]), fmaf(bf2f((bf16_t)(x.u[0] >> 16)), s0[1], h0[1]));
        x.u[1] = pack_bf2(fmaf(bf2f((bf16_t)(x.u[1] & 0xffff)), s0[2], h0[2]), fmaf(bf2f((bf16_t)(x.u[1] >> 16)), s0[3], h0[3]));
        x.u[2] = pack_bf2(fmaf(bf2f((bf16_t)(x.u[2] & 0xffff)), s1[0], h1[0]), fmaf(bf2f((bf16_t)(x.u[2] >> 16)), s1[1], h1[1]));
        x.u[3] = pack_bf2(fmaf(bf2f((bf16_t)(x.u[3] & 0xffff)), s1[2], h1[2]), fmaf(bf2f((bf16_t)(x.u[3] >> 16)), s1[3], h1[3]));
        xf[s] = x.v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the table reads have returned before this wave issues anything into the slot
    }
  }

  // per-wave 3.5 KiB of LDS behind the tables: every tile this wave writes goes through it, so that the global stores are 16 B per lane
  // over contiguous runs (the accumulator layout itself gives 8-B pieces of 32 different rows per instruction: measured, the memory
  // pipeline takes ~10 us per 25 MB written that way, and with one wave per SIMD nothing else runs meanwhile)
  unsigned char* wb = smem + LW_OFF + w * LW_BYTES;
  const unsigned wb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wb;      // its LDS byte address
  const int m0 = blockIdx.x * LTM + w * 32;                                  // the wave's first row (uniform)
  const int bidx = __builtin_amdgcn_readfirstlane(m0 / p.ntok);              // ntok % 32 == 0: one sample, 32 consecutive tokens
  const int tok0 = m0 - bidx * p.ntok;

  // ---- first GEMM: acc1[10 tiles of 32 columns] = b1 (+ residual) + W1 . X
  bf16x8 tf[20];
  {
    f32x16 acc1[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      if constexpr (MODE == 1) {
        lds_write_b128(wb_lds + (lane >> 2) * 80 + (lane & 3) * 16, res[2 * i]);
        lds_write_b128(wb_lds + (lane >> 2) * 80 + (lane & 3) * 16 + 16 * 80, res[2 * i + 1]);
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1t + 32 * i + 8 * q4 + 4 * hi);
        float v0 = bv.x, v1 = bv.y, v2 = bv.z, v3 = bv.w;
        if constexpr (MODE == 1) {
          const u32x2 r = *reinterpret_cast<const u32x2*>(wb + l31 * 80 + q4 * 16 + hi * 8);
          v0 += __uint_as_float(r.x << 16); v1 += __uint_as_float(r.x & 0xffff0000u);
          v2 += __uint_as_float(r.y << 16); v3 += __uint_as_float(r.y & 0xffff0000u);
        }
        acc1[i][4 * q4] = v0; acc1[i][4 * q4 + 1] = v1; acc1[i][4 * q4 + 2] = v2; acc1[i][4 * q4 + 3] = v3;
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) wf[0][q] = wfrag(smem, q, 0);
    auto stages1 = [&](auto... gs) {
      (run_stage(gs, [&](int s, const bf16x8* wfp) {
         constexpr int G = decltype(gs)::value;
         constexpr int U = G / KS1, J = G % KS1;
#pragma unroll
         for (int q = 0; q < 5; ++q)
           acc1[5 * U + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], xf[4 * J + s], acc1[5 * U + q], 0, 0, 0);
       }), ...);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { stages1(std::integral_constant<int, I>{}...); }(std::make_integer_sequence<int, G1>{});

    // ---- epilogue 1: round, store the residual stream (32 x 32 tiles through the wave's LDS patch: 64-B runs); LayerNorm of the row
    // (its 160 values of this lane + the other half wave's 160) on the ROUNDED values (what every later consumer of the stream reads);
    // the normalised values packed as the second GEMM's B fragments
    float sum = 0.f, sq = 0.f;
    const int t_wr = l31 * 80 + hi * 8;                                       // + 16 q4
    const int t_rd0 = (lane >> 2) * 80 + (lane & 3) * 16, t_rd1 = t_rd0 + 16 * 80;
    bf16_t* tdst = p.T + (size_t)(m0 + (lane >> 2)) * LC + (lane & 3) * 8;    // + 32 i (+ 16 rows for the second round)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const uint2 o = make_uint2(pack_bf2(acc1[i][4 * q4], acc1[i][4 * q4 + 1]), pack_bf2(acc1[i][4 * q4 + 2], acc1[i][4 * q4 + 3]));
        lds_write_b64(wb_lds + t_wr + 16 * q4, o);
        const float v0 = __uint_as_float(o.x << 16), v1 = __uint_as_float(o.x & 0xffff0000u);
        const float v2 = __uint_as_float(o.y << 16), v3 = __uint_as_float(o.y & 0xffff0000u);
        acc1[i][4 * q4] = v0; acc1[i][4 * q4 + 1] = v1; acc1[i][4 * q4 + 2] = v2; acc1[i][4 * q4 + 3] = v3;
        sum += (v0 + v1) + (v2 + v3);
        sq += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(wb + t_rd0), d1 = *reinterpret_cast<const u32x4*>(wb + t_rd1);
      {
        *reinterpret_cast<u32x4*>(tdst + 32 * i) = d0;
        *reinterpret_cast<u32x4*>(tdst + 32 * i + 16 * LC) = d1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    sum += __shfl_xor(sum, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float mean = sum * (1.f / LC);
    const float rstd = rsqrtf(fmaxf(sq * (1.f / LC) - mean * mean, 0.f) + p.ln_eps);
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int r = 8 * hh + 2 * jj;
          pk.u[jj] = pack_bf2((acc1[i][r] - mean) * rstd, (acc1[i][r + 1] - mean) * rstd);
        }
        tf[2 * i + hh] = pk.v;
      }
  }

  // ---- second GEMM, one segment of 384 columns at a time: acc2[12 tiles] = c2 + W2' segment . LN(t); head-major scatter
  // per-lane pieces of the scatter addresses: Q / K tile [32 tokens][48] (rows of 112 B in LDS), V^T tile [48][32 tokens] (rows of 64 B)
  const int qk_wr = l31 * 112 + hi * 8;                                       // + 16 g6
  const int v_wr = l31 * 2 + hi * 256;                                        // + 64 (8 g6 + e)
  int qk_rd[3], v_rd[3];
  unsigned v_go[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int c = lane + 64 * r;
    qk_rd[r] = (c / 6) * 112 + (c % 6) * 16;
    v_rd[r] = c * 16;                                                         // row c / 4, 16-B part c % 4
    v_go[r] = (unsigned)(c >> 2) * p.ntok_pad + (c & 3) * 8;
  }
  [&]<int... SG>(std::integer_sequence<int, SG...>) {
    ([&] {
      constexpr int SIDX = SG;               // position in the run order
      constexpr int SEG = lp_seg(MODE, SIDX);  // 0 -> Q (scaled), 1 -> K, 2 -> V^T   (dp == 48: head and element of a column are static)
      f32x16 acc2[12];
#pragma unroll
      for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(c2t + SEG * LSEG + 32 * i + 8 * q4 + 4 * hi);
          acc2[i][4 * q4] = c.x; acc2[i][4 * q4 + 1] = c.y; acc2[i][4 * q4 + 2] = c.z; acc2[i][4 * q4 + 3] = c.w;
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[0][q] = wfrag(smem + ((G1 + SIDX * 15) % LNSLOT) * LSLOT, q, 0);
      auto stages2 = [&](auto... gs) {
        (run_stage(gs, [&](int s, const bf16x8* wfp) {
           constexpr int H = decltype(gs)::value - G1 - SIDX * 15;
           constexpr int CH = H / 5, J = H % 5;
#pragma unroll
           for (int q = 0; q < 4; ++q)
             acc2[4 * CH + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfp[q], tf[4 * J + s], acc2[4 * CH + q], 0, 0, 0);
         }), ...);
      };
      [&]<int... I>(std::integer_sequence<int, I...>) {
        stages2(std::integral_constant<int, G1 + SIDX * 15 + I>{}...);
      }(std::make_integer_sequence<int, 15>{});
      // scatter, one head (48 columns = 6 groups of 8: group 6 h + g6 = tile (6h + g6) / 4, registers 4 ((6h + g6) % 4) .. +3) at a time
      {
        if constexpr (SEG < 2) {
          bf16_t* dst = (SEG == 0 ? p.Cq : p.Ck) + ((size_t)(bidx * 8) * p.ntok_pad + tok0) * LDP + lane * 8;
          const size_t head_stride = (size_t)p.ntok_pad * LDP;
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g6 = 0; g6 < 6; ++g6) {
              const int grp = 6 * h + g6, i = grp >> 2, q4 = grp & 3;
              float v0 = acc2[i][4 * q4], v1 = acc2[i][4 * q4 + 1], v2 = acc2[i][4 * q4 + 2], v3 = acc2[i][4 * q4 + 3];
              if constexpr (SEG == 0) { v0 *= p.qscale; v1 *= p.qscale; v2 *= p.qscale; v3 *= p.qscale; }
              lds_write_b64(wb_lds + qk_wr + 16 * g6, make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3)));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
              *reinterpret_cast<u32x4*>(dst + h * head_stride + r * 512) = *reinterpret_cast<const u32x4*>(wb + qk_rd[r]);
          }
        } else {
          bf16_t* dst = p.Cvt + (size_t)(bidx * 8) * p.dpv * p.ntok_pad + tok0;
          const size_t head_stride = (size_t)p.dpv * p.ntok_pad;
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g6 = 0; g6 < 6; ++g6) {
              const int grp = 6 * h + g6, i = grp >> 2, q4 = grp & 3;
              lds_write_b16_pair(wb_lds + v_wr + 64 * (8 * g6), pack_bf2(acc2[i][4 * q4], acc2[i][4 * q4 + 1]));
              lds_write_b16_pair(wb_lds + v_wr + 64 * (8 * g6 + 2), pack_bf2(acc2[i][4 * q4 + 2], acc2[i][4 * q4 + 3]));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
              *reinterpret_cast<u32x4*>(dst + h * head_stride + v_go[r]) = *reinterpret_cast<const u32x4*>(wb + v_rd[r]);
            // spare row dp of V^T := 1.0 (the attention kernel reads the softmax row sum off the PV MFMAs)
            if (lane < 4 && p.dpv > LDP)
              *reinterpret_cast<u32x4*>(dst + h * head_stride + (size_t)LDP * p.ntok_pad + lane * 8) = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
          }
        }
      }
    }(), ...);
  }(std::make_integer_sequence<int, NSEG>{});
}

// dst[n][16 g + 8 hi + j] := src[n][16 g + 8 (j >> 2) + 4 hi + (j & 3)]   (K = 320)
__global__ __launch_bounds__(256) void lnproj_kperm_kernel(const bf16_t* __restrict__ src, int N, bf16_t* __restrict__ dst) {
  const int64_t n = (int64_t)N * LC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % LC), row = (int)(i / LC);
    const int g16 = pos / 16, r = pos % 16, hi = r / 8, j = r % 8;
    dst[i] = src[(size_t)row * LC + 16 * g16 + 8 * (j >> 2) + 4 * hi + (j & 3)];
  }
}

template <int MODE>
int lnproj_launch_mode(const LnProjArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = LW_OFF + 4 * LW_BYTES;
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)lnproj_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(lnproj_kernel<MODE>, dim3(a.M / LTM), dim3(256), smem, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool lnproj_supported(int C, int M, int heads, int dp) { return C == LC && M > 0 && M % LTM == 0 && heads == 8 && dp == LDP; }

int lnproj_kperm_launch(const bf16_t* src, int N, bf16_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(lnproj_kperm_kernel, dim3(512), dim3(256), 0, s, src, N, dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

int lnproj_launch(const LnProjArgs& a, hipStream_t s) {
  GILL_REQUIRE(lnproj_supported(LC, a.M, a.heads, a.dp), "lnproj: C = 320, M % 128 == 0, 8 heads of padded dim 48 only");
  GILL_REQUIRE(a.X && a.T && a.W1 && a.b1 && a.W2p && a.c2 && a.Cq, "lnproj: null operand");
  GILL_REQUIRE(a.ntok > 0 && a.ntok % 32 == 0 && a.ntok_pad >= a.ntok && a.ntok_pad % 8 == 0, "lnproj: tokens per sample must be a multiple of 32");
  GILL_REQUIRE(a.gn_ss == nullptr || (a.mode == 0 && a.ntok % LTM == 0), "lnproj: the folded GroupNorm needs mode 0 and whole 128-row tiles per sample");
  if (a.mode == 0) {
    GILL_REQUIRE(a.Ck && a.Cvt && a.seg_base == 0 && a.dpv >= a.dp, "lnproj: QKV outputs missing");
    return lnproj_launch_mode<0>(a, s);
  }
  GILL_REQUIRE(a.seg_base == 0, "lnproj mode 1 writes the Q segment");
  return lnproj_launch_mode<1>(a, s);
}
