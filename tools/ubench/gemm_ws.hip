// Micro-benchmark: producer / consumer WAVE SPECIALISATION for the 256 x 160 conv tile.
//
// The shipped ping-pong loop (gemm.hip "PING-PONG", k_pp below) gives every wave both jobs: in its memory phase a wave issues 18
// ds_read_b128 and 6-7 LDS-DMA pieces, in its MMA phase 40 MFMAs.  The 26 pieces of a phase queue at the CU's ONE address / L1 pipe
// (~17 cycles per 1-KiB piece: 54-58 B/clk/CU, profiles/r04_lds_fill_ubench.log), behind them the fragment reads: the memory phase
// (~800 cycles) is longer than the 680-cycle MMA phase it is supposed to hide behind, so a K step costs ~1900-2200 cycles for 1360
// cycles of MFMA per SIMD.
//
// Here: waves 0-3 (one per SIMD) only multiply — wave tile 128 x 80, 80 MFMAs per K step, fragment reads interleaved one row block
// ahead (13 ds_read_b128 per 40 MFMAs), never a global_load_lds, an M0 write or a pointer; waves 4-7 (the SIMDs' second waves) only
// load — all 52 pieces of a K step, spread over the whole step, L stages ahead of the consumers.  One s_barrier per stage hands a
// landed stage over and a consumed slot back.
//   KT = 64, NS = 3, PF = 0: 128-B rows as in the shipped kernel, one barrier per K step, the consumers read a stage's first
//                            fragments after its barrier (one exposed LDS round trip per K step).
//   KT = 32, NS = 6, PF = 1: 64-B rows, six half-stage slots (same 156 KiB), a barrier per half step, the loaders certify one
//                            half stage further so the consumers prefetch the next half stage's first fragments across the barrier.
// Correctness: the full C is written once and 4096 random entries are checked against a CPU sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <type_traits>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define BN 160
#define BK 64
#define KW 320   // A is [M][KW]: the K walk wraps over it (cache behaviour of a 3x3 conv over an L2/MALL-resident input)
__device__ __forceinline__ int xcd_tile() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------------------------------------------------------- baseline
// the shipped ping-pong loop (copy of gemm_loop.hip k_gemm256pp<0>), for a same-session reference
__global__ __launch_bounds__(512, 1) void k_pp(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                               int M, int N, int K, int write_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int A_ELEMS = 256 * BK, B_ELEMS = BN * BK, BUF = A_ELEMS + B_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int srow = lane >> 3;
  const int nsteps = K / BK;
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[3];
  for (int i = 0; i < 4; ++i) {
    int row = (i * 8 + w) * 8 + srow;
    a_ptr[i] = A + (size_t)(m0 + row) * KW + ((lane & 7) ^ (row & 7)) * 8;
  }
  const int grp = w >> 2;
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ (row & 7)) * 8;
  }
  int kcol = 0;
  auto issue = [&](int buf) {
    bf16_t* As = smem + buf * BUF;
    bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)(As + (i * 8 + w) * 8 * BK), 16, 0, 0);
      a_ptr[i] += BK;
    }
    if (++kcol == KW / BK) { kcol = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_ptr[i] -= KW; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < 2 || grp == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 8 + w) * 8 * BK), 16, 0, 0);
        w_ptr[i] += BK;
      }
    }
  };
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[4][5];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const int frow = lane & 15, fkc = lane >> 4;
  bf16x8 af[2][4], bfr[2][5];
  auto wait_next = [&](int k) {
    if (k + 2 < nsteps) { if (grp == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto mem = [&](int k) {
    const bf16_t* As = smem + (k % 3) * BUF;
    const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[kk][i] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = wn * 80 + j * 16 + frow;
        bfr[kk][j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
      }
    }
    if (k + 2 < nsteps) issue((k + 2) % 3);
  };
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
  };
  issue(0);
  if (nsteps > 1) issue(1);
  if (nsteps > 1) { if (grp == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 0) {
    for (int k = 0; k < nsteps; ++k) {
      mem(k);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      wait_next(k);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nsteps; ++k) {
      mem(k);
      wait_next(k);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (write_c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(m0 + wm * 64 + i * 16 + frow) * N + n0 + wn * 80 + j * 16 + fkc * 4 + r] = acc[i][j][r];
  } else {
    float result = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    C[(size_t)blockIdx.x * 512 + tid] = result;
  }
}

// ------------------------------------------------------------------------------------------------------- wave-specialised
// ABL: timing-only ablations (results wrong): bit 0 loaders issue nothing, bit 1 consumers read no fragments after the first stage
template <int KT, int NS, int PF, int PRIO, int ABL>
__global__ __launch_bounds__(512, 1) void k_ws(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                               int M, int N, int K, int write_c) {
  static_assert((KT == 64 || KT == 32) && NS >= 3 && (PF == 0 || PF == 1) && NS - 2 - PF >= 0, "ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RB = KT * 2;                 // bytes per tile row of a stage (128 | 64)
  constexpr int CPR = RB / 16;               // 16-B chunks per row (8 | 4)
  constexpr int RPP = 64 / CPR;              // rows per 1-KiB piece (8 | 16)
  constexpr int A_BYTES = 256 * RB, W_BYTES = BN * RB, SLOT = A_BYTES + W_BYTES;
  constexpr int KK = KT / 32;                // 32-wide MFMA k steps per stage
  constexpr int AP = 256 / RPP, WP = BN / RPP;          // pieces per stage (32 + 20 | 16 + 10)
  constexpr int API = AP / 4, WPI = (WP + 3) / 4;       // per loader wave (8 + 5 | 4 + 3 (the last W piece on two of the four))
  constexpr int L = NS - 1;                  // stages the loaders run ahead
  constexpr int FLY = NS - 2 - PF;           // stages that may stay in flight across a barrier
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int nst = K / KT;                    // stages
  // bank swizzle of the 16-B chunk index, applied to the DMA source address and to the fragment read address.
  // 128-B rows: chunk ^= row & 7.  64-B rows: chunk ^= g[(row >> 2) & 3], g = {0, 3, 2, 1} — conflict-free for the ds_read_b128 lane groups
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): within a group the four lanes of one (row & 3) get four different chunks
#define SWZ(row) (KT == 64 ? ((row) & 7) : ((4 - (((row) >> 2) & 3)) & 3))

  if (w >= 4) {
    // ================================================================ loader waves
    const int lw = w - 4;
    const int srow = lane / CPR, sch = lane % CPR;
    const bf16_t* a_ptr[API];
    const bf16_t* w_ptr[WPI];
#pragma unroll
    for (int i = 0; i < API; ++i) {
      const int row = (i * 4 + lw) * RPP + srow;
      a_ptr[i] = A + (size_t)(m0 + row) * KW + (sch ^ SWZ(row)) * 8;
    }
    const bool w_last = ((WPI - 1) * 4 + lw) < WP;        // wave-uniform: does this wave own a piece in the last W round
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
      int p = i * 4 + lw; if (p > WP - 1) p = WP - 1;
      const int row = p * RPP + srow;
      w_ptr[i] = W + (size_t)(n0 + row) * K + (sch ^ SWZ(row)) * 8;
    }
    int kcol = 0;
    auto issue = [&](int slot) {
      if (ABL & 1) return;
      unsigned char* As = smem + slot * SLOT;
      unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int i = 0; i < API; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                         (__attribute__((address_space(3))) void*)(As + (i * 4 + lw) * 1024), 16, 0, 0);
        a_ptr[i] += KT;
      }
      if (++kcol == KW / KT) { kcol = 0;
#pragma unroll
        for (int i = 0; i < API; ++i) a_ptr[i] -= KW; }
#pragma unroll
      for (int i = 0; i < WPI; ++i) {
        if (i < WPI - 1 || w_last) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                           (__attribute__((address_space(3))) void*)(Bs + (i * 4 + lw) * 1024), 16, 0, 0);
          w_ptr[i] += KT;
        }
      }
    };
    // stages <= j have landed (own pieces).  Steady state: FLY younger stages stay in flight (immediate); near the ends: everything
    auto certify = [&](int j, int issued) {
      if (issued - 1 - j >= FLY && FLY > 0) {
        if (w_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY * (API + WPI)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY * (API + WPI - 1)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    int issued = 0;
#pragma unroll
    for (int s = 0; s < L; ++s)
      if (s < nst) { issue(s % NS); ++issued; }
    certify(PF, issued);
    __builtin_amdgcn_s_barrier();                       // B(0): stages 0 .. PF handed over
    for (int k = 0; k < nst; ++k) {
      if (k + L < nst) { issue((k + L) % NS); ++issued; }
      certify(k + 1 + PF, issued);
      __builtin_amdgcn_s_barrier();                     // B(k + 1): stages <= k + 1 + PF handed over, slot of stage k handed back
    }
    return;
  }

  // ================================================================== consumer (MMA) waves: 2 x 2, wave tile 128 x 80
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  const int wm = w >> 1, wn = w & 1;
  const int frow = lane & 15, fkc = lane >> 4;
  f32x4 acc[8][5];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // per-lane byte offsets inside a slot (row blocks are multiples of 16 rows: the swizzle depends on frow only)
  int aoff[KK], boff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int ch = (KT == 64 ? kk * 4 + fkc : fkc) ^ SWZ(frow);
    aoff[kk] = (wm * 128 + frow) * RB + ch * 16;
    boff[kk] = A_BYTES + (wn * 80 + frow) * RB + ch * 16;
  }
  auto rdA = [&](int slot, int kk, int i) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + aoff[kk] + i * 16 * RB);
  };
  auto rdB = [&](int slot, int kk, int j) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + boff[kk] + j * 16 * RB);
  };
  // B fragments: two sets, set = parity of the running 32-wide k step (the set not in use receives the next step's fragments, one per row
  // block 1..5); A fragments: four rotating registers, read TWO row blocks ahead of their MFMAs
  bf16x8 bs[2][5], af[4];
#pragma unroll
  for (int j = 0; j < 5; ++j) bs[0][j] = bs[1][j] = (bf16x8)(short)0x3c00;
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = (bf16x8)(short)0x3c00;
  __builtin_amdgcn_s_barrier();                         // B(0)
  if (PF) {
#pragma unroll
    for (int j = 0; j < 5; ++j) bs[0][j] = rdB(0, 0, j);
    af[0] = rdA(0, 0, 0);
    af[1] = rdA(0, 0, 1);
  }
  int slot = 0;
  // one stage; P0 = parity of its first k step (compile time: the fragment sets are register arrays)
  auto stage = [&](int k, auto p0_tag) {
    constexpr int P0 = decltype(p0_tag)::value;
    const int nslot = slot + 1 == NS ? 0 : slot + 1;
    const bool rd = !((ABL & 2) && k > 0);
    if (!PF && rd) {
      // the stage's first fragments, behind its barrier
#pragma unroll
      for (int j = 0; j < 5; ++j) bs[P0][j] = rdB(slot, 0, j);
      af[0] = rdA(slot, 0, 0);
      af[1] = rdA(slot, 0, 1);
    }
    const bool more_next = PF && (k + 1 < nst);          // the next stage's first fragments are prefetched across the barrier
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      constexpr int dummy = 0; (void)dummy;
      const int cur = (P0 + kk) & 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = kk * 8 + i + 2;                     // the A fragment read now (two row blocks ahead)
        if (rd) {
          if (f < KK * 8) af[f & 3] = rdA(slot, f >> 3, f & 7);
          else if (more_next) af[f & 3] = rdA(nslot, 0, f - KK * 8);
          if (i >= 1 && i <= 5) {
            if (kk + 1 < KK) bs[cur ^ 1][i - 1] = rdB(slot, kk + 1, i - 1);
            else if (more_next) bs[cur ^ 1][i - 1] = rdB(nslot, 0, i - 1);
          }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bs[cur][j], af[(kk * 8 + i) & 3], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_barrier();                       // B(k + 1)
    slot = nslot;
  };
  if constexpr (KK == 2) {
    for (int k = 0; k < nst; ++k) stage(k, std::integral_constant<int, 0>{});
  } else {
    // (nst even in this benchmark)
    for (int k = 0; k < nst; k += 2) {
      stage(k, std::integral_constant<int, 0>{});
      stage(k + 1, std::integral_constant<int, 1>{});
    }
  }
  if (write_c) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(m0 + wm * 128 + i * 16 + frow) * N + n0 + wn * 80 + j * 16 + fkc * 4 + r] = acc[i][j][r];
  } else {
    float result = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    C[(size_t)blockIdx.x * 512 + tid] = result;
  }
#undef SWZ
}

// ------------------------------------------------------------------------------------- wave-specialised, two barriers per K step
// KT = 64, 3 slots.  The consumers cross B_b(k) in the MIDDLE of stage k (it certifies stage k + 1, whose first fragments are then
// prefetched during the second half of stage k) and B_a(k + 1) at its end (it hands slot k back): no fragment read is ever waited
// for right behind a barrier, and the loaders keep one stage in flight across both (issued at B_a(k), certified at B_b(k + 1):
// 1.5 K steps to land).
// ABL: bit 0 loaders issue nothing, bit 1 no fragment reads, bit 2 the reads are issued but the MFMAs run on constant fragments
// (the reads' issue / LDS cost without their waits in front of the MFMAs)
template <int PRIO, int ABL, int DEEP = 0>
__global__ __launch_bounds__(512, 1) void k_ws2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                int M, int N, int K, int write_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KT = 64, NS = 3, RB = 128, A_BYTES = 256 * RB, W_BYTES = BN * RB, SLOT = A_BYTES + W_BYTES;
  constexpr int API = 8, WPI = 5, NPC = API + WPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int nst = K / KT;
  if (w >= 4) {
    const int lw = w - 4;
    const int srow = lane >> 3, sch = lane & 7;
    const bf16_t* a_ptr[API];
    const bf16_t* w_ptr[WPI];
#pragma unroll
    for (int i = 0; i < API; ++i) {
      const int row = (i * 4 + lw) * 8 + srow;
      a_ptr[i] = A + (size_t)(m0 + row) * KW + (sch ^ (row & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
      const int row = (i * 4 + lw) * 8 + srow;
      w_ptr[i] = W + (size_t)(n0 + row) * K + (sch ^ (row & 7)) * 8;
    }
    int kcol = 0;
    auto issue = [&](int slot) {
      if (ABL & 1) return;
      unsigned char* As = smem + slot * SLOT;
      unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int i = 0; i < API; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                         (__attribute__((address_space(3))) void*)(As + (i * 4 + lw) * 1024), 16, 0, 0);
        a_ptr[i] += KT;
      }
      if (++kcol == KW / KT) { kcol = 0;
#pragma unroll
        for (int i = 0; i < API; ++i) a_ptr[i] -= KW; }
#pragma unroll
      for (int i = 0; i < WPI; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 4 + lw) * 1024), 16, 0, 0);
        w_ptr[i] += KT;
      }
    };
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nst) { issue(s); ++issued; }
    // stage 0 landed
    if (issued >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // X0
    for (int k = 0; k < nst; ++k) {
      // stage k + 1 landed; stage k + 2 may stay in flight
      if (issued - 1 - (k + 1) >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // B_b(k)
      __builtin_amdgcn_s_barrier();                     // B_a(k + 1): slot of stage k is free
      if (k + NS < nst) { issue(k % NS); ++issued; }
    }
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  const int wm = w >> 1, wn = w & 1;
  const int frow = lane & 15, fkc = lane >> 4;
  f32x4 acc[8][5];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  int aoff[2], boff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = (kk * 4 + fkc) ^ (frow & 7);
    aoff[kk] = (wm * 128 + frow) * RB + ch * 16;
    boff[kk] = A_BYTES + (wn * 80 + frow) * RB + ch * 16;
  }
  auto rdA = [&](int slot, int kk, int i) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + aoff[kk] + i * 16 * RB);
  };
  auto rdB = [&](int slot, int kk, int j) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + boff[kk] + j * 16 * RB);
  };
  // DEEP = 0: four rotating A registers, read two row blocks (10 MFMAs) ahead.  DEEP = 1: eight A registers, A(h + 1, i) is read right
  // BEHIND the MFMAs of row block i of half stage h (the registers it frees): every fragment is in flight for a whole half stage
  bf16x8 bs[2][5], af[8];
  const bf16x8 cfrag = (bf16x8)(short)(0x3c00 + lane);
#pragma unroll
  for (int j = 0; j < 5; ++j) bs[0][j] = bs[1][j] = cfrag;
#pragma unroll
  for (int i = 0; i < 8; ++i) af[i] = cfrag;
  __builtin_amdgcn_s_barrier();                         // X0
#pragma unroll
  for (int j = 0; j < 5; ++j) bs[0][j] = rdB(0, 0, j);
  if (DEEP) {
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = rdA(0, 0, i);
  } else {
    af[0] = rdA(0, 0, 0);
    af[1] = rdA(0, 0, 1);
  }
  int slot = 0;
  for (int k = 0; k < nst; ++k) {
    const int nslot = slot + 1 == NS ? 0 : slot + 1;
    const bool rd = !((ABL & 2) && k > 0);
    const bool more = (k + 1 < nst);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = kk * 8 + i + 2;
        if (rd && !DEEP) {
          if (f < 16) af[f & 3] = rdA(slot, f >> 3, f & 7);
          else if (more) af[f & 3] = rdA(nslot, 0, f - 16);
        }
        if (rd && i >= 1 && i <= 5) {
          if (kk == 0) bs[1][i - 1] = rdB(slot, 1, i - 1);
          else if (more) bs[0][i - 1] = rdB(nslot, 0, i - 1);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (ABL & 4) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cfrag, cfrag, acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bs[kk][j], af[DEEP ? i : ((kk * 8 + i) & 3)], acc[i][j], 0, 0, 0);
        }
        if (rd && DEEP) {
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 0) af[i] = rdA(slot, 1, i);
          else if (more) af[i] = rdA(nslot, 0, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ABL & 4) {
#pragma unroll
        for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(bs[kk ^ 1][j]));
#pragma unroll
        for (int i = 0; i < (DEEP ? 8 : 4); ++i) asm volatile("" ::"v"(af[i]));
      }
      __builtin_amdgcn_s_barrier();                     // B_b(k) after kk = 0, B_a(k + 1) after kk = 1
    }
    slot = nslot;
  }
  if (write_c) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(m0 + wm * 128 + i * 16 + frow) * N + n0 + wn * 80 + j * 16 + fkc * 4 + r] = acc[i][j][r];
  } else {
    float result = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    C[(size_t)blockIdx.x * 512 + tid] = result;
  }
}

// --------------------------------------------------------------------------------- ping-pong consumers + dedicated loader waves
// 768 threads: waves 0-7 are the shipped ping-pong pair of wave groups (wave tile 64 x 80, fragment reads in the wave's memory phase, 40
// MFMAs in its MMA phase) minus every LDS-DMA piece, vmcnt wait and global pointer; waves 8-11 (a third wave per SIMD: <= 168 registers
// per lane) issue all 52 pieces of a K step, two K steps ahead, and certify "stage landed" in front of the even-phase barrier.
// Phase p: group A MEM(k) at p = 2k, MMA(k) at 2k + 1; group B MEM(k) at 2k + 1, MMA(k) at 2k + 2.  Stage k is last read (into
// registers) in phase 2k + 1, so its slot is refilled with stage k + 3 from phase 2k + 2 on; stage k must have landed when phase 2k starts.
// ABL bit 0: loaders issue nothing (timing only)
template <int LPRIO, int ABL>
__global__ __launch_bounds__(768, 1) void k_pp12(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                 int M, int N, int K, int write_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KT = 64, NS = 3, RB = 128, A_BYTES = 256 * RB, W_BYTES = BN * RB, SLOT = A_BYTES + W_BYTES;
  constexpr int API = 8, WPI = 5, NPC = API + WPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int nst = K / KT;
  if (w >= 8) {
    if (LPRIO) __builtin_amdgcn_s_setprio(LPRIO);
    const int lw = w - 8;
    const int srow = lane >> 3, sch = lane & 7;
    const bf16_t* a_ptr[API];
    const bf16_t* w_ptr[WPI];
#pragma unroll
    for (int i = 0; i < API; ++i) {
      const int row = (i * 4 + lw) * 8 + srow;
      a_ptr[i] = A + (size_t)(m0 + row) * KW + (sch ^ (row & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
      const int row = (i * 4 + lw) * 8 + srow;
      w_ptr[i] = W + (size_t)(n0 + row) * K + (sch ^ (row & 7)) * 8;
    }
    int kcol = 0;
    auto issue = [&](int slot) {
      if (ABL & 1) return;
      unsigned char* As = smem + slot * SLOT;
      unsigned char* Bs = As + A_BYTES;
#pragma unroll
      for (int i = 0; i < API; ++i) {
        if ((ABL & 2) && (i & 1)) continue;          // (timing: half the A pieces)
        if ((ABL & 8)) continue;                     // (timing: no A pieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                         (__attribute__((address_space(3))) void*)(As + (i * 4 + lw) * 1024), 16, 0, 0);
        a_ptr[i] += KT;
      }
      if (++kcol == KW / KT) { kcol = 0;
#pragma unroll
        for (int i = 0; i < API; ++i) a_ptr[i] -= KW; }
#pragma unroll
      for (int i = 0; i < WPI; ++i) {
        if (ABL & 4) continue;                       // (timing: no W pieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 4 + lw) * 1024), 16, 0, 0);
        w_ptr[i] += KT;
      }
    };
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nst) { issue(s); ++issued; }
    if (issued >= 3 && !ABL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // into phase 0: stage 0 landed
    for (int k = 0; k < nst; ++k) {
      // phase 2k: (k > 0) the slot of stage k - 1 is free -> stage k + 2
      if (k > 0 && k + 2 < nst) { issue((k + 2) % NS); ++issued; }
      __builtin_amdgcn_s_barrier();                     // into phase 2k + 1
      // stage k + 1 landed before phase 2k + 2; stage k + 2 may stay in flight
      if (ABL & 14) {
        // (ablations issue fewer pieces per stage: count those)
        constexpr int npc_abl = ((ABL & 8) ? 0 : ((ABL & 2) ? API / 2 : API)) + ((ABL & 4) ? 0 : WPI);
        if (issued - 1 - (k + 1) >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(npc_abl) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (issued - 1 - (k + 1) >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // into phase 2k + 2
    }
    __builtin_amdgcn_s_barrier();                       // (group B's last phase)
    return;
  }
  const int grp = w >> 2;
  const int wq = w & 3;
  const int wm = (grp << 1) | (wq >> 1), wn = wq & 1;   // group A: rows 0-127, group B: rows 128-255
  const int frow = lane & 15, fkc = lane >> 4;
  f32x4 acc[4][5];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  bf16x8 af[2][4], bfr[2][5];
  int aoff[2], boff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = (kk * 4 + fkc) ^ (frow & 7);
    aoff[kk] = (wm * 64 + frow) * RB + ch * 16;
    boff[kk] = A_BYTES + (wn * 80 + frow) * RB + ch * 16;
  }
  auto mem = [&](int slot) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + aoff[kk] + i * 16 * RB);
#pragma unroll
      for (int j = 0; j < 5; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + boff[kk] + j * 16 * RB);
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
  };
  __builtin_amdgcn_s_barrier();                         // into phase 0
  int slot = 0;
  if (grp == 0) {
    for (int k = 0; k < nst; ++k) {
      mem(slot);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nst; ++k) {
      mem(slot);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
  }
  if (write_c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(m0 + wm * 64 + i * 16 + frow) * N + n0 + wn * 80 + j * 16 + fkc * 4 + r] = acc[i][j][r];
  } else {
    float result = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    C[(size_t)blockIdx.x * 768 + tid] = result;
  }
}

static float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(const bf16_t*, const bf16_t*, float*, int, int, int, int);

static void run(kern_t kern, int smem, int nthr, const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what,
                const bf16_t* hA, const bf16_t* hW, bool check) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int grid = (M / 256) * (N / BN);
  double maxerr = -1.0;
  if (check) {
    hipMemset(C, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), smem, 0, A, W, C, M, N, K, 1);
    hipDeviceSynchronize();
    float* hC = (float*)malloc((size_t)M * N * 4);
    hipMemcpy(hC, C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    unsigned s = 777;
    maxerr = 0.0;
    for (int t = 0; t < 4096; ++t) {
      s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M;
      s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * KW + (k % KW)]) * (double)bf2f(hW[(size_t)n * K + k]);
      const double e = fabs(ref - (double)hC[(size_t)m * N + n]) / (fabs(ref) + 1.0);
      if (e > maxerr) maxerr = e;
    }
    free(hC);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), smem, 0, A, W, C, M, N, K, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), smem, 0, A, W, C, M, N, K, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R;
  printf("%-64s %7.1f us  %6.0f TFLOP/s  (%.3f us per K step of 64)  max rel err %.2e  [%s]\n", what, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         maxerr, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int M = argc > 2 ? atoi(argv[2]) : 32768, N = 320, K = argc > 1 ? atoi(argv[1]) : 2880;
  bf16_t *A, *W; float* C;
  const size_t na = (size_t)M * KW, nw = (size_t)N * K;
  hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&C, (size_t)M * N * 4 + (1 << 22));
  bf16_t* hA = (bf16_t*)malloc(na * 2);
  bf16_t* hW = (bf16_t*)malloc(nw * 2);
  unsigned s = 12345;
  for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
  for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; hW[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
  hipMemcpy(A, hA, na * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, hW, nw * 2, hipMemcpyHostToDevice);
  printf("M=%d N=%d K=%d, %d tiles of 256 x 160\n", M, N, K, (M / 256) * (N / BN));
  constexpr int SM3 = 3 * (256 * 128 + BN * 128);
  constexpr int SM6 = 6 * (256 * 64 + BN * 64);
  for (int rep = 0; rep < 3; ++rep) {
    const bool chk = rep == 0;
    run(k_pp, SM3, 512, A, W, C, M, N, K, "ping-pong (shipped structure)", hA, hW, chk);
    run(k_ws<64, 3, 0, 0, 0>, SM3, 512, A, W, C, M, N, K, "WS KT=64 3 slots, reads behind the barrier", hA, hW, chk);
    run(k_pp12<0, 0>, SM3, 768, A, W, C, M, N, K, "PP12: ping-pong consumers + 4 loader waves", hA, hW, chk);
    run(k_pp12<1, 0>, SM3, 768, A, W, C, M, N, K, "PP12, loaders prio 1", hA, hW, chk);
    run(k_ws2<0, 0>, SM3, 512, A, W, C, M, N, K, "WS2 two barriers per K step, prefetch across both", hA, hW, chk);
    run(k_ws2<0, 0, 1>, SM3, 512, A, W, C, M, N, K, "WS2 deep: A fragments a whole half stage ahead", hA, hW, chk);
    if (rep == 0) {
      run(k_ws<64, 3, 1, 0, 0>, SM3, 512, A, W, C, M, N, K, "WS KT=64 3 slots, prefetch across the barrier (no DMA in flight)", hA, hW, chk);
      run(k_ws<32, 6, 1, 0, 0>, SM6, 512, A, W, C, M, N, K, "WS KT=32 6 slots, prefetch across the barrier", hA, hW, chk);
    }
    if (rep == 2) {
      run(k_pp12<0, 1>, SM3, 768, A, W, C, M, N, K, "  ablation PP12: loaders issue nothing", hA, hW, false);
      run(k_pp12<0, 8>, SM3, 768, A, W, C, M, N, K, "  ablation PP12: W pieces only (20 of 52)", hA, hW, false);
      run(k_pp12<0, 4>, SM3, 768, A, W, C, M, N, K, "  ablation PP12: A pieces only (32 of 52)", hA, hW, false);
      run(k_pp12<0, 2>, SM3, 768, A, W, C, M, N, K, "  ablation PP12: half the A pieces + W (36 of 52)", hA, hW, false);
      run(k_pp12<0, 0>, SM3, 768, A, W, C, M, N, K, "  (PP12 again)", hA, hW, false);
      run(k_ws2<0, 1>, SM3, 512, A, W, C, M, N, K, "  ablation WS2: loaders issue nothing", hA, hW, false);
      run(k_ws2<0, 2>, SM3, 512, A, W, C, M, N, K, "  ablation WS2: no fragment reads", hA, hW, false);
      run(k_ws2<0, 3>, SM3, 512, A, W, C, M, N, K, "  ablation WS2: neither (MFMAs + barriers)", hA, hW, false);
      run(k_ws2<0, 4>, SM3, 512, A, W, C, M, N, K, "  ablation WS2: reads issued, MFMAs on constants", hA, hW, false);
      run(k_ws2<0, 5>, SM3, 512, A, W, C, M, N, K, "  ablation WS2: reads issued, MFMAs on constants, no loaders", hA, hW, false);
      run(k_ws2<0, 1, 1>, SM3, 512, A, W, C, M, N, K, "  ablation WS2 deep: loaders issue nothing", hA, hW, false);
    }
  }
  return 0;
}
