// Micro-benchmark of the GEMM main loop structure (128x160x64 tile, 256 threads, 2-stage LDS-DMA ring).
// Components can be switched off to find what bounds the loop.  M=32768 N=320 K=2880 (the L0 conv as a plain GEMM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define BM 128
#define BN 160
#define BK 64
#define KW 320   // A is [M][KW]: the K walk wraps over it (cache behaviour of a 3x3 conv over an L2/MALL-resident input)
__device__ __forceinline__ int xcd_tile() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// MODE 0: 16x16x32, waves 2x2 (64x80 each).  MODE 1: 32x32x16, waves 2(M) x 2(K halves), 64x160 each.
template <int MODE, int LOADS, int LDSR, int BARR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_gemm(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                   int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, BUF = A_ELEMS + B_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int srow = lane >> 3;
  const int nsteps = K / BK;
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[5];
  for (int i = 0; i < 4; ++i) {
    int row = (i * 4 + w) * 8 + srow;
    int sw = MODE == 0 ? (row & 7) : ((row >> 1) & 7);
    a_ptr[i] = A + (size_t)(m0 + row) * KW + ((lane & 7) ^ sw) * 8;
  }
  for (int i = 0; i < 5; ++i) {
    int row = (i * 4 + w) * 8 + srow;
    int sw = MODE == 0 ? (row & 7) : ((row >> 1) & 7);
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ sw) * 8;
  }
  int kcol = 0;
  int patch_step = 0;
  auto issue = [&](int buf) {
    if (!LOADS) return;
    bf16_t* As = smem + buf * BUF;
    bf16_t* Bs = LOADS == 2 ? smem + buf * B_ELEMS : As + A_ELEMS;     // patch mode: LDS = [W ring 2 x 20 KiB | patch 36 KiB]
    if (LOADS == 2) {
      // LDS-resident input patch: the A side is fetched once per 9 K steps (one 64-channel chunk of a (2+2) x (64+2) pixel
      // patch = 264 pixels = 33 KiB: 9 wave-instructions per wave), the W side every step
      if (patch_step == 0) {
        bf16_t* P = smem + 2 * B_ELEMS;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[i & 3] + (i >> 2) * 8 * KW),
                                           (__attribute__((address_space(3))) void*)(P + (i * 4 + w) * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a_ptr[i] += BK;
        if (++kcol == KW / BK) { kcol = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) a_ptr[i] -= KW; }
      }
      if (++patch_step == 9) patch_step = 0;
    } else if (LOADS != 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)(As + (i * 4 + w) * 8 * BK), 16, 0, 0);
      a_ptr[i] += BK;
    }
    if (++kcol == KW / BK) { kcol = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_ptr[i] -= KW; }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                       (__attribute__((address_space(3))) void*)(Bs + (i * 4 + w) * 8 * BK), 16, 0, 0);
      w_ptr[i] += BK;
    }
  };
  const int wm = w >> 1, wn = w & 1;
  float result = 0.f;
  if constexpr (MODE == 0) {
    f32x4 acc[4][5];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int frow = lane & 15, fkc = lane >> 4;
    bf16x8 af[4], bfr[5];
    for (int i = 0; i < 4; ++i) af[i] = (bf16x8)(short)(0x3c00 + lane);
    for (int j = 0; j < 5; ++j) bfr[j] = (bf16x8)(short)(0x3c10 + lane);
    // LOADS == 3: the A fragments come straight from global memory into registers (the MFMA fragment layout IS a 16-B-per-lane
    // row-major read), one K step ahead; only W goes through LDS-DMA
    bf16x8 aq[2][2][4];
    const bf16_t* arow[4];
    for (int i = 0; i < 4; ++i) arow[i] = A + (size_t)(m0 + wm * 64 + i * 16 + frow) * KW + fkc * 8;
    auto load_a = [&](int step, int slot) {
      const int kc = (step * BK) % KW;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) aq[slot][kk][i] = *reinterpret_cast<const bf16x8*>(arow[i] + kc + kk * 32);
    };
    issue(0);
    if (LOADS == 3) load_a(0, 0);
    int buf = 0;
    auto body = [&](int it, auto slot_tag) {
      constexpr int SLOT = decltype(slot_tag)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (BARR) __builtin_amdgcn_s_barrier();
      const bf16_t* As = LOADS == 2 ? smem + 2 * B_ELEMS + ((it % 9) / 3) * 66 * BK + ((it % 9) % 3) * BK : smem + buf * BUF;
      const bf16_t* Bs = LOADS == 2 ? smem + buf * B_ELEMS : As + A_ELEMS;
      if (LOADS == 3 && it + 1 < nsteps) load_a(it + 1, SLOT ^ 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (LOADS == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = aq[SLOT][kk][i];
        }
        if (LDSR) {
          if (LOADS != 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = wm * 64 + i * 16 + frow;
              af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
            }
          }
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int row = wn * 80 + j * 16 + frow;
            bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
          }
        }
        if (kk == 0 && it + 1 < nsteps) issue(buf ^ 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      buf ^= 1;
    };
    for (int it = 0; it < nsteps; it += 2) {
      body(it, std::integral_constant<int, 0>{});
      if (it + 1 < nsteps) body(it + 1, std::integral_constant<int, 1>{});
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f32x16 acc[2][5];
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    const int wk = wn;
    const int l31 = lane & 31, lh = lane >> 5;
    bf16x8 a2[2], b5[5];
    for (int i = 0; i < 2; ++i) a2[i] = (bf16x8)(short)(0x3c00 + lane);
    for (int j = 0; j < 5; ++j) b5[j] = (bf16x8)(short)(0x3c10 + lane);
    issue(0);
    int buf = 0;
    for (int it = 0; it < nsteps; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (BARR) __builtin_amdgcn_s_barrier();
      const bf16_t* As = smem + buf * BUF;
      const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int st = wk * 2 + ss;
        if (LDSR) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = wm * 64 + t * 32 + l31;
            a2[t] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((st * 2 + lh) ^ ((row >> 1) & 7)) * 8));
          }
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int row = j * 32 + l31;
            b5[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + (((st * 2 + lh) ^ ((row >> 1) & 7)) * 8));
          }
        }
        if (ss == 0 && it + 1 < nsteps) issue(buf ^ 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b5[j], a2[t], acc[t][j], 0, 0, 0);
      }
      buf ^= 1;
    }
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) result += acc[t][j][r];
  }
  C[(size_t)blockIdx.x * 256 + tid] = result;
}


// 8 waves: 2(M) x 2(N... none) -> wave (wm = w&1 .. ) layout: wm = (w >> 1) & 1 picks 64 rows, wk = w & 1 ... see below.
// Config K8: block 128x160, 512 threads; waves = 2 (M halves) x 4 (k16 steps of the BK=64 stage): each wave 64x160 via 32x32x16,
// one k16 step per stage -> 10 MFMAs + 7 ds_read_b128 per stage per wave.  STAGES-deep ring, counted vmcnt.
template <int STAGES>
__global__ __launch_bounds__(512, 1) void k_gemm8(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                  int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, BUF = A_ELEMS + B_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int srow = lane >> 3;
  const int nsteps = K / BK;
  // A: 16 row-groups of 8 -> 2 per wave; W: 20 row-groups -> waves 0..3 take 3, waves 4..7 take 2
  const bf16_t* a_ptr[2];
  const bf16_t* w_ptr[3];
  for (int i = 0; i < 2; ++i) {
    int row = (i * 8 + w) * 8 + srow;
    a_ptr[i] = A + (size_t)(m0 + row) * KW + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  const int nw = w < 4 ? 3 : 2;
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  int kcol = 0;
  auto issue = [&](int buf) {
    bf16_t* As = smem + buf * BUF;
    bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)(As + (i * 8 + w) * 8 * BK), 16, 0, 0);
      a_ptr[i] += BK;
    }
    if (++kcol == KW / BK) { kcol = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) a_ptr[i] -= KW; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < nw) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 8 + w) * 8 * BK), 16, 0, 0);
        w_ptr[i] += BK;
      }
    }
  };
  const int wm = w >> 2, st = w & 3;
  f32x16 acc[2][5];
  for (int t = 0; t < 2; ++t) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int s = 0; s < STAGES - 1; ++s) if (s < nsteps) issue(s);
  int buf = 0, buf_issue = STAGES - 1;
  for (int it = 0; it < nsteps; ++it) {
    // leave the youngest STAGES-2 stages in flight
    const int rem = nsteps - 1 - it;   // stages issued after stage `it`
    const int fly = rem < STAGES - 2 ? rem : STAGES - 2;
    if (w < 4) {
      if (fly >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (fly == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (fly >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (fly == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bf16_t* As = smem + buf * BUF;
    const bf16_t* Bs = As + A_ELEMS;
    bf16x8 a2[2], b5[5];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = wm * 64 + t * 32 + l31;
      a2[t] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((st * 2 + lh) ^ ((row >> 1) & 7)) * 8));
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int row = j * 32 + l31;
      b5[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + (((st * 2 + lh) ^ ((row >> 1) & 7)) * 8));
    }
    if (it + STAGES - 1 < nsteps) issue(buf_issue);
    buf_issue = (buf_issue + 1 == STAGES) ? 0 : buf_issue + 1;
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b5[j], a2[t], acc[t][j], 0, 0, 0);
  }
  float result = 0.f;
  for (int t = 0; t < 2; ++t) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) result += acc[t][j][r];
  C[(size_t)blockIdx.x * 512 + tid] = result;
}


// BM = 256: 512 threads, waves 4 (M) x 2 (N), each 64 x 80 via 16x16x32 (the production per-wave code), one block per CU.
template <int STAGES>
__global__ __launch_bounds__(512, 1) void k_gemm256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                    int M, int N, int K) {
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
  for (int i = 0; i < 4; ++i) {     // 32 row groups of 8 -> 4 per wave
    int row = (i * 8 + w) * 8 + srow;
    a_ptr[i] = A + (size_t)(m0 + row) * KW + ((lane & 7) ^ (row & 7)) * 8;
  }
  const int nw = w < 4 ? 3 : 2;      // 20 row groups of W
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
      if (i < nw) {
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
  for (int s = 0; s < STAGES - 1; ++s) if (s < nsteps) issue(s);
  int buf = 0, buf_issue = STAGES - 1;
  for (int it = 0; it < nsteps; ++it) {
    const int rem = nsteps - 1 - it;
    const int fly = rem < STAGES - 2 ? rem : STAGES - 2;
    if (w < 4) { if (fly >= 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else       { if (fly >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    const bf16_t* As = smem + buf * BUF;
    const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[5];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = wn * 80 + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + (((kk * 4 + fkc) ^ (row & 7)) * 8));
      }
      if (kk == 0) {
        if (it + STAGES - 1 < nsteps) issue(buf_issue);
        buf_issue = (buf_issue + 1 == STAGES) ? 0 : buf_issue + 1;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
  }
  float result = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  C[(size_t)blockIdx.x * 512 + tid] = result;
}


// 256 x 160 tile, 8 waves 4 x 2, PING-PONG: waves w and w + 4 share a SIMD and belong to different groups; the groups run half a
// K step apart so that in every phase ONE wave of each SIMD issues the step's 40 MFMAs while the other does the step's memory
// work (18 ds_read_b128 fragment loads + its 6-7 LDS-DMA pieces of stage k + 2).  One barrier per phase (two per K step),
// 3-deep ring (156 KiB).  Phase p: group A  MEM(k) at p = 2k, MMA(k) at p = 2k + 1;  group B  MEM(k) at 2k + 1, MMA(k) at 2k + 2.
// Before the barrier into phase 2k every wave has waited for its own stage-k pieces.
template <int PRIO>
__global__ __launch_bounds__(512, 1) void k_gemm256pp(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                      int M, int N, int K) {
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
  const int grp = w >> 2;            // 0: group A (7 pieces per stage), 1: group B (6)
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
  auto wait_next = [&](int k) {      // own pieces of stage k + 1 landed (stage k + 2, if issued, may stay in flight)
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
    if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
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
  float result = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  C[(size_t)blockIdx.x * 512 + tid] = result;
}

template <int PRIO>
__global__ __launch_bounds__(512, 1) void k_gemm256pp2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                      int M, int N, int K) {
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
  const int grp = w >> 2;            // 0: group A (7 pieces per stage), 1: group B (6)
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ (row & 7)) * 8;
  }
  int kcol = 0;
  auto issue_piece = [&](int buf, int pc) {     // piece 0..3: A rows, 4..6: W rows
    bf16_t* As = smem + buf * BUF;
    bf16_t* Bs = As + A_ELEMS;
    if (pc < 4) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[pc],
                                       (__attribute__((address_space(3))) void*)(As + (pc * 8 + w) * 8 * BK), 16, 0, 0);
      a_ptr[pc] += BK;
      if (pc == 3) { if (++kcol == KW / BK) { kcol = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_ptr[i] -= KW; } }
    } else {
      const int i = pc - 4;
      if (i < 2 || grp == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 8 + w) * 8 * BK), 16, 0, 0);
        w_ptr[i] += BK;
      }
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int pc = 0; pc < 7; ++pc) issue_piece(buf, pc);
  };
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[4][5];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const int frow = lane & 15, fkc = lane >> 4;
  bf16x8 af[2][4], bfr[2][5];
  auto wait_next = [&](int k) {      // own pieces of stage k + 1 landed (stage k + 2, if issued, may stay in flight)
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
  };
  // MMA(k) also issues this wave's LDS-DMA pieces of stage `st` (one piece after every 5th MFMA), st < 0: none
  auto mma = [&](int st) {
    const bool doit = st >= 0 && st < nsteps;
    const int buf = doit ? st % 3 : 0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        if (doit && kk * 4 + i < 7) { __builtin_amdgcn_sched_barrier(0); issue_piece(buf, kk * 4 + i); __builtin_amdgcn_sched_barrier(0); }
      }
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
      mma(k + 2);
      wait_next(k);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
  } else {
    if (2 < nsteps) issue(2);
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nsteps; ++k) {
      mem(k);
      // own pieces of stage k + 1 landed; stage k + 2 (issued during MMA(k - 1) / the prologue) may stay in flight
      if (k + 2 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma(k + 3);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  float result = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  C[(size_t)blockIdx.x * 512 + tid] = result;
}

template <int PRIO>
void run256pp2(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = 3 * (256 * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm256pp2<PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  int grid = (M / 256) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm256pp2<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm256pp2<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s prio=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)  [%s]\n", what, PRIO, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         hipGetErrorString(hipGetLastError()));
}


// Ping-pong as above with v_mfma_f32_32x32x16_bf16 (full rate from ONE wave, which is what a ping-pong phase has per SIMD; 16x16x32
// needs two): waves 8 (M) x 1, wave tile 32 x 160 = 1 x 5 tiles, 20 MFMAs + 24 ds_read_b128 per K step.
template <int PRIO>
void run256pp(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = 3 * (256 * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm256pp<PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  int grid = (M / 256) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm256pp<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm256pp<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s prio=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)  [%s]\n", what, PRIO, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         hipGetErrorString(hipGetLastError()));
}


// Ping-pong as above with v_mfma_f32_32x32x16_bf16 (full rate from ONE wave, which is what a ping-pong phase has per SIMD; 16x16x32
// needs two): waves 8 (M) x 1, wave tile 32 x 160 = 1 x 5 tiles, 20 MFMAs + 24 ds_read_b128 per K step.
template <int PRIO>
__global__ __launch_bounds__(512, 1) void k_gemm256pp32(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                        int M, int N, int K) {
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
    a_ptr[i] = A + (size_t)(m0 + row) * KW + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  const int grp = w >> 2;
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
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
  f32x16 acc[5];
  for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  bf16x8 af[4], bfr[4][5];
  auto wait_next = [&](int k) {
    if (k + 2 < nsteps) { if (grp == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto mem = [&](int k) {
    const bf16_t* As = smem + (k % 3) * BUF;
    const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int row = w * 32 + l31;
      af[st] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((st * 2 + lh) ^ ((row >> 1) & 7)) * 8));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int brow = j * 32 + l31;
        bfr[st][j] = *reinterpret_cast<const bf16x8*>(Bs + brow * BK + (((st * 2 + lh) ^ ((brow >> 1) & 7)) * 8));
      }
    }
    if (k + 2 < nsteps) issue((k + 2) % 3);
  };
  auto mma = [&]() {
    if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[st][j], af[st], acc[j], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
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
  float result = 0.f;
  for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) result += acc[j][r];
  C[(size_t)blockIdx.x * 512 + tid] = result;
}

template <int PRIO>
void run256pp32(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = 3 * (256 * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm256pp32<PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  int grid = (M / 256) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm256pp32<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm256pp32<PRIO><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s prio=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)  [%s]\n", what, PRIO, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         hipGetErrorString(hipGetLastError()));
}

template <int STAGES>
void run256(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = STAGES * (256 * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm256<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  int grid = (M / 256) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm256<STAGES><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm256<STAGES><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s stages=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)  [%s]\n", what, STAGES, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         hipGetErrorString(hipGetLastError()));
}

template <int STAGES>
void run8(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = STAGES * (BM * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm8<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm8<STAGES><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm8<STAGES><<<grid, 512, smem>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s stages=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)  [%s]\n", what, STAGES, us, 2.0 * M * N * K / us / 1e6, us / (K / BK),
         hipGetErrorString(hipGetLastError()));
}

template <int MODE, int LOADS, int LDSR, int BARR, int OCC>
void run(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what) {
  constexpr int smem = 2 * (BM * BK + BN * BK) * 2;
  hipFuncSetAttribute((const void*)k_gemm<MODE, LOADS, LDSR, BARR, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  int lds = OCC == 1 ? 100 * 1024 : (LOADS == 2 ? 2 * BN * BK * 2 + 36 * 1024 : smem);
  int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_gemm<MODE, LOADS, LDSR, BARR, OCC><<<grid, 256, lds>>>(A, W, C, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_gemm<MODE, LOADS, LDSR, BARR, OCC><<<grid, 256, lds>>>(A, W, C, M, N, K);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / R;
  printf("%-44s mode=%d loads=%d lds=%d barrier=%d occ=%d: %7.1f us  %6.0f TFLOP/s  (%.2f us per K step)\n", what, MODE, LOADS, LDSR, BARR, OCC, us,
         2.0 * M * N * K / us / 1e6, us / (K / BK));
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  int M = 32768, N = 320, K = argc > 1 ? atoi(argv[1]) : 2880;
  bf16_t *A, *W; float* C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)1024 * 512 * 4);
  // random bf16 in [-1, 1)
  {
    size_t na = (size_t)M * K, nw = (size_t)N * K;
    bf16_t* h = (bf16_t*)malloc((na > nw ? na : nw) * 2);
    unsigned s = 12345;
    for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; h[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
    hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; h[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
    hipMemcpy(W, h, nw * 2, hipMemcpyHostToDevice);
    free(h);
  }
  printf("M=%d N=%d K=%d, %d tiles\n", M, N, K, (M / BM) * (N / BN));
  run<0, 1, 1, 1, 2>(A, W, C, M, N, K, "16x16x32 full");
  run<0, 3, 1, 1, 2>(A, W, C, M, N, K, "16x16x32 A fragments direct to registers, W via LDS-DMA");
  run<0, 2, 1, 1, 2>(A, W, C, M, N, K, "16x16x32 A patch once per 9 steps");
  run<0, 0, 1, 1, 2>(A, W, C, M, N, K, "16x16x32 no global loads");
  run<0, 0, 1, 0, 2>(A, W, C, M, N, K, "16x16x32 no loads, no barrier");
  run<0, 0, 0, 0, 2>(A, W, C, M, N, K, "16x16x32 pure MFMA");
  run<0, 1, 0, 1, 2>(A, W, C, M, N, K, "16x16x32 loads+barrier, no LDS reads");
  run<0, 1, 1, 1, 1>(A, W, C, M, N, K, "16x16x32 full, 1 block/CU");
  run<0, 0, 0, 0, 1>(A, W, C, M, N, K, "16x16x32 pure MFMA, 1 block/CU");
  run<1, 1, 1, 1, 2>(A, W, C, M, N, K, "32x32x16 K-split full");
  run<1, 0, 1, 1, 2>(A, W, C, M, N, K, "32x32x16 no global loads");
  run<1, 0, 1, 0, 2>(A, W, C, M, N, K, "32x32x16 no loads, no barrier");
  run<1, 0, 0, 0, 2>(A, W, C, M, N, K, "32x32x16 pure MFMA");
  run<1, 1, 0, 1, 2>(A, W, C, M, N, K, "32x32x16 loads+barrier, no LDS reads");
  run<1, 1, 1, 1, 1>(A, W, C, M, N, K, "32x32x16 full, 1 block/CU");
  run<1, 0, 0, 0, 1>(A, W, C, M, N, K, "32x32x16 pure MFMA, 1 block/CU");
  run256<2>(A, W, C, M, N, K, "256x160 tile, 8 waves 4x2, 1 block/CU");
  run256<3>(A, W, C, M, N, K, "256x160 tile, 8 waves 4x2, 1 block/CU");
  run256pp<0>(A, W, C, M, N, K, "256x160 PING-PONG (2 groups of 4 waves)");
  run256pp<1>(A, W, C, M, N, K, "256x160 PING-PONG (2 groups of 4 waves)");
  run256pp2<0>(A, W, C, M, N, K, "256x160 PING-PONG, DMA issued inside MMA phase");
  run256pp32<0>(A, W, C, M, N, K, "256x160 PING-PONG 32x32x16, waves 8x1");
  run256pp32<1>(A, W, C, M, N, K, "256x160 PING-PONG 32x32x16, waves 8x1");
  run8<2>(A, W, C, M, N, K, "K8: 8 waves 128x160, 1 block/CU");
  run8<3>(A, W, C, M, N, K, "K8: 8 waves 128x160, 1 block/CU");
  run8<4>(A, W, C, M, N, K, "K8: 8 waves 128x160, 1 block/CU");
  return 0;
}
