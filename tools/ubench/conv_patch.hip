// Micro-benchmark (round 5): the ping-pong 3x3 convolution loop with the A operand read from a 9-TAP INPUT PATCH held in LDS instead of nine
// per-tap A tiles.  profiles/r05_wave_specialisation.md: the shipped loop pays ~7 cycles of wall time per 1-KiB LDS-DMA piece per K step whoever
// issues it, and 32 of its 52 pieces per K step are the A tile — every input pixel is fetched nine times, once per tap.  Here K runs chunk-major
// (64 input channels at a time, the nine taps inside): per chunk the tile's (4 + 2) x 64-pixel input patch is staged ONCE (49 pieces per 9 K steps
// instead of 288; double-buffered: 2 x 49 KiB next to the 3 x 20 KiB weight ring = 158 KiB), and a tap is a row offset into it:
//   LDS address = patch + ((wm + 1 + dy) * 64 + i * 16 + frow + dx) * 128 B + swizzle, swizzle = (chunk ^ ((frow + dx) & 7)) * 16  (three
//   precomputed lane values, kk = 1 is ^ 64), i * 2048 an immediate: two VALU adds per K step instead of 32 LDS-DMA pieces per workgroup.
// Image borders: rows above / below the image are staged from a zero page; x = -1 / 64 are ONE lane of one fragment read each (i = 0, frow = 0 for
// dx = -1; i = 3, frow = 15 for dx = +1), redirected to a zero row of the patch.
// Real convolution semantics (8 samples of 64 x 64 pixels, 320 channels, pad 1), checked against a CPU sum; the baseline k_pp is the shipped loop's
// copy from gemm_ws.hip (a GEMM over a wrapped A: same MFMA / LDS / DMA work per K step as the shipped convolution).
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


// ------------------------------------------------------------------------------------------------------------ 9-tap patch
__device__ __attribute__((aligned(256))) uint32_t g_zero[64];
template <int ABL>     // ABL bit 0: no patch pieces (timing only)
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_patch(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                  int M, int N, int K, int write_c) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  constexpr int PATCH_ROWS = 392;                       // 6 x 64 pixels + the zero row (384) + 7 filler rows of its LDS-DMA group
  constexpr int PATCH_B = PATCH_ROWS * 128;             // 50176 B
  constexpr int WST_B = BN * 128;                       // 20480 B per weight stage
  unsigned char* patch0 = smem_raw;
  unsigned char* wring = smem_raw + 2 * PATCH_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int b = m0 / 4096, y0 = (m0 % 4096) / 64;       // tile = image rows y0 .. y0 + 3 of sample b
  const int srow = lane >> 3;
  const int nsteps = K / BK;                             // 45 = 5 chunks x 9 taps
  const int grp = w >> 2;
  const bf16_t* zero_lane = (const bf16_t*)g_zero + (lane & 7) * 8;
  const bf16_t* w_ptr[3];
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ (row & 7)) * 8;
  }
  // patch group g (8 patch rows = 8 pixels of image row y0 - 1 + g / 8): source pointer of this lane for chunk 0, or the zero page
  auto patch_src = [&](int g, int chunk) __attribute__((always_inline)) -> const bf16_t* {
    const int yy = g >> 3;
    const int iy = y0 - 1 + yy;
    const bool ok = (g < 48) && iy >= 0 && iy < 64;
    const bf16_t* p = A + ((size_t)b * 4096 + (size_t)iy * 64 + (g & 7) * 8 + srow) * KW + chunk * 64 + ((lane & 7) ^ srow) * 8;
    return ok ? p : zero_lane;
  };
  auto issue_patch = [&](int g, int chunk) __attribute__((always_inline)) {
    if (ABL & 1) return;
    unsigned char* dst = patch0 + (chunk & 1) * PATCH_B + g * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)patch_src(g, chunk),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_w = [&](int stage) __attribute__((always_inline)) {
    unsigned char* Bs = wring + stage * WST_B;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < 2 || grp == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 8 + w) * 1024), 16, 0, 0);
        w_ptr[i] += BK;
      }
    }
  };
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[4][5];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const int frow = lane & 15, fkc = lane >> 4;
  // A fragment addressing (bytes, relative to the patch buffer): lane base + tap offset + swizzle(dx); i * 2048 is an immediate
  const int a_lane = ((wm + 1) * 64 + frow) * 128;
  // (the swizzle term is recomputed per K step — 4 VALU: a `swz[3]` array indexed by dx went to scratch, with a scratch_load + s_waitcnt vmcnt(0) per
  // K step that also waited for every LDS-DMA piece in flight: 57.6 us; three named registers behind selects sent ALL arrays to scratch)
  const int zrow = 384 * 128 + fkc * 16;                 // zero row (any chunk: all zeros)
  const bool lo_edge = frow == 0, hi_edge = frow == 15;
  bf16x8 af[2][4], bfr[2][5];
  // step k -> chunk, tap
  int had_prev = 0;
  auto wait_next = [&](int k, int had_patch) __attribute__((always_inline)) {
    // W(k + 1) landed: younger = this step's patch pieces + W(k + 2) (3 pieces in group 0, 2 in group 1).
    // ABL bit 1 (ORDER): the patch pieces go out BEHIND the step's weight pieces, so the previous step's pieces (first touches of the input: L2 misses)
    // are younger than W(k + 1) too and may stay in flight one more step.  ABL bit 2 (BURST, with ORDER): the whole next patch at tap 0, 6-7 pieces per wave.
    if (k + 2 < nsteps) {
      const int n = (grp == 0 ? 3 : 2) + had_patch + ((ABL & 2) ? had_prev : 0);
      switch (n) {
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;     // (over-waiting is always correct)
      }
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    had_prev = had_patch;
  };
  auto mem = [&](int k, int chunk, int tap) __attribute__((always_inline)) -> int {
    const unsigned char* P = patch0 + (chunk & 1) * PATCH_B;
    const unsigned char* Bs = wring + (k % 3) * WST_B;
    const int ty = tap / 3;
    const int dy = ty - 1, dx = tap - ty * 3 - 1;
    const int base = a_lane + (dy * 64 + dx) * 128 + ((fkc ^ ((frow + dx) & 7)) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int a0 = kk ? (base ^ 64) : base;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ad = a0 + i * 2048;
        if (i == 0) ad = (dx < 0 && lo_edge) ? zrow : ad;
        if (i == 3) ad = (dx > 0 && hi_edge) ? zrow : ad;
        af[kk][i] = *reinterpret_cast<const bf16x8*>(P + ad);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = wn * 80 + j * 16 + frow;
        bfr[kk][j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + (((kk * 4 + fkc) ^ (row & 7)) * 16));
      }
    }
    int had = 0;
    if ((ABL & 2) && k + 2 < nsteps) issue_w((k + 2) % 3);
    if (ABL & 4) {
      if (tap == 0 && chunk + 1 < nsteps / 9) {
#pragma unroll
        for (int t = 0; t < 7; ++t) { const int g = t * 8 + w; if (g <= 48) { issue_patch(g, chunk + 1); ++had; } }
      }
    } else if (tap <= 6 && chunk + 1 < nsteps / 9) {
      const int g = tap * 8 + w;
      if (g <= 48) { issue_patch(g, chunk + 1); had = (ABL & 1) ? 0 : 1; }
    }
    if (!(ABL & 2) && k + 2 < nsteps) issue_w((k + 2) % 3);
    return had;
  };
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
  };
  // prologue: the whole patch of chunk 0 (7 groups per wave), two weight stages
  for (int t = 0; t < 7; ++t) { const int g = t * 8 + w; if (g <= 48) issue_patch(g, 0); }
  issue_w(0);
  if (nsteps > 1) issue_w(1);
  if (nsteps > 1) { if (grp == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int chunk = 0, tap = 0;
  if (grp == 0) {
    for (int k = 0; k < nsteps; ++k) {
      const int had = mem(k, chunk, tap);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      wait_next(k, had);
      if (++tap == 9) { tap = 0; ++chunk; }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nsteps; ++k) {
      const int had = mem(k, chunk, tap);
      wait_next(k, had);
      if (++tap == 9) { tap = 0; ++chunk; }
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


// ------------------------------------------------------------------------------------------------- 9-tap patch, version 2
// The same loop with NOTHING but reads and LDS-DMA issues in the memory phase: the next K step's fragment addresses (three registers + their ^ 64
// twins), the next patch piece's source pointer and the (chunk, dy, dx) walk are computed in the MMA phase, one scalar / vector instruction behind
// each MFMA (sched_group_barrier), where the shipped kernel keeps its pointer bookkeeping.
template <int ABL>
__global__ __launch_bounds__(512, 1) void k_patch2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C,
                                                   int M, int N, int K, int write_c) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  constexpr int PATCH_ROWS = 392;
  constexpr int PATCH_B = PATCH_ROWS * 128;
  constexpr int WST_B = BN * 128;
  unsigned char* patch0 = smem_raw;
  unsigned char* wring = smem_raw + 2 * PATCH_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int tile = xcd_tile();
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const int b = m0 / 4096, y0 = (m0 % 4096) / 64;
  const int srow = lane >> 3;
  const int nsteps = K / BK, nchunks = nsteps / 9;
  const int grp = w >> 2;
  const bf16_t* zero_lane = (const bf16_t*)g_zero + (lane & 7) * 8;
  const bf16_t* w_ptr[3];
  for (int i = 0; i < 3; ++i) {
    int g = i * 8 + w; if (g > 19) g = 19;
    int row = g * 8 + srow;
    w_ptr[i] = W + (size_t)(n0 + row) * K + ((lane & 7) ^ (row & 7)) * 8;
  }
  auto patch_src = [&](int g, int chunk) __attribute__((always_inline)) -> const bf16_t* {
    const int yy = g >> 3;
    const int iy = y0 - 1 + yy;
    const bool ok = (g < 48) && iy >= 0 && iy < 64;
    const bf16_t* p = A + ((size_t)b * 4096 + (size_t)iy * 64 + (g & 7) * 8 + srow) * KW + chunk * 64 + ((lane & 7) ^ srow) * 8;
    return ok ? p : zero_lane;
  };
  auto issue_w = [&](int stage) __attribute__((always_inline)) {
    unsigned char* Bs = wring + stage * WST_B;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < 2 || grp == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                         (__attribute__((address_space(3))) void*)(Bs + (i * 8 + w) * 1024), 16, 0, 0);
        w_ptr[i] += BK;
      }
    }
  };
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[4][5];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const int frow = lane & 15, fkc = lane >> 4;
  const int a_lane = ((wm + 1) * 64 + frow) * 128;
  const int zrow = 384 * 128 + fkc * 16;
  const bool lo_edge = frow == 0, hi_edge = frow == 15;
  bf16x8 af[2][4], bfr[2][5];
  // state of the NEXT memory phase, prepared during the MMA phase before it
  int chunk = 0, dy = -1, dx = -1, tapi = 0;          // (chunk, tap) of the step whose addresses are in c_*
  int c_b0, c_e0, c_e3;                               // fragment addresses (bytes from smem_raw) of i = 1..2 (+ immediates), i = 0, i = 3
  const bf16_t* c_psrc; int c_pdst; bool c_has;       // patch piece of this step (for chunk + 1)
  auto prep = [&](int ch, int ty, int tx, int tp) __attribute__((always_inline)) {
    const int pb = (ch & 1) * PATCH_B;
    const int base = pb + a_lane + (ty * 64 + tx) * 128 + ((fkc ^ ((frow + tx) & 7)) << 4);
    c_b0 = base;
    c_e0 = (tx < 0 && lo_edge) ? pb + zrow : base;
    c_e3 = (tx > 0 && hi_edge) ? pb + zrow : base + 3 * 2048;
    const int g = tp * 8 + w;
    c_has = !(ABL & 1) && tp <= 6 && ch + 1 < nchunks && g <= 48;
    c_psrc = patch_src(g, ch + 1);
    c_pdst = ((ch + 1) & 1) * PATCH_B + g * 1024;
  };
  auto wait_next = [&](int k, bool had_patch) __attribute__((always_inline)) {
    if (k + 2 < nsteps) {
      const int n = (grp == 0 ? 3 : 2) + (had_patch ? 1 : 0);
      if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto mem = [&](int k) __attribute__((always_inline)) -> bool {
    const unsigned char* Bs = wring + (k % 3) * WST_B;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int x = kk ? 64 : 0;
      af[kk][0] = *reinterpret_cast<const bf16x8*>(smem_raw + (c_e0 ^ x));
      af[kk][1] = *reinterpret_cast<const bf16x8*>(smem_raw + (c_b0 ^ x) + 2048);
      af[kk][2] = *reinterpret_cast<const bf16x8*>(smem_raw + (c_b0 ^ x) + 4096);
      af[kk][3] = *reinterpret_cast<const bf16x8*>(smem_raw + (c_e3 ^ x));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = wn * 80 + j * 16 + frow;
        bfr[kk][j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + (((kk * 4 + fkc) ^ (row & 7)) * 16));
      }
    }
    const bool had = c_has;
    if (had)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)c_psrc,
                                       (__attribute__((address_space(3))) void*)(smem_raw + c_pdst), 16, 0, 0);
    if (k + 2 < nsteps) issue_w((k + 2) % 3);
    return had;
  };
  auto mma = [&]() __attribute__((always_inline)) {
    // advance the (chunk, dy, dx) walk and prepare the next step's addresses between the MFMAs
    int ndx = dx + 1, ndy = dy, nch = chunk, ntap = tapi + 1;
    if (ndx == 2) { ndx = -1; ndy = dy + 1; }
    if (ndy == 2) { ndy = -1; nch = chunk + 1; ntap = 0; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    dx = ndx; dy = ndy; chunk = nch; tapi = ntap;
    prep(chunk, dy, dx, tapi);
#pragma unroll
    for (int q = 0; q < 40; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 1, 0);
    }
  };
  // prologue: the whole patch of chunk 0 (7 groups per wave), two weight stages
  if (!(ABL & 1))
    for (int t = 0; t < 7; ++t) {
      const int g = t * 8 + w;
      if (g <= 48)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)patch_src(g, 0),
                                         (__attribute__((address_space(3))) void*)(patch0 + g * 1024), 16, 0, 0);
    }
  issue_w(0);
  if (nsteps > 1) issue_w(1);
  prep(0, -1, -1, 0);
  if (nsteps > 1) { if (grp == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 0) {
    for (int k = 0; k < nsteps; ++k) {
      const bool had = mem(k);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma();
      wait_next(k, had);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nsteps; ++k) {
      const bool had = mem(k);
      wait_next(k, had);
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

static float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
typedef void (*kern_t)(const bf16_t*, const bf16_t*, float*, int, int, int, int);

// conv = 1: reference is the 3x3 convolution (chunk-major K: k = (chunk * 9 + tap) * 64 + kk); conv = 0: the wrapped GEMM of k_pp
static void run(kern_t kern, int smem, int nthr, const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, const char* what,
                const bf16_t* hA, const bf16_t* hW, bool check, int conv) {
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
      s = s * 1664525u + 1013904223u; int m = (s >> 8) % M;
      s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
      if (t < 512) m = (m / 64) * 64 + ((t & 1) ? 63 : 0);          // image-border columns
      if (t >= 512 && t < 1024) m = (m / 4096) * 4096 + ((t & 1) ? 63 * 64 : 0) + (m % 64);   // first / last image rows
      double ref = 0.0;
      if (!conv) {
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * KW + (k % KW)]) * (double)bf2f(hW[(size_t)n * K + k]);
      } else {
        const int bb = m / 4096, y = (m % 4096) / 64, x = m % 64;
        for (int c = 0; c < KW / 64; ++c)
          for (int tap = 0; tap < 9; ++tap) {
            const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
            if (iy < 0 || iy >= 64 || ix < 0 || ix >= 64) continue;
            const size_t ar = (size_t)bb * 4096 + iy * 64 + ix;
            for (int kk = 0; kk < 64; ++kk)
              ref += (double)bf2f(hA[ar * KW + c * 64 + kk]) * (double)bf2f(hW[(size_t)n * K + (c * 9 + tap) * 64 + kk]);
          }
      }
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
  const int M = 32768, N = 320, K = 2880;
  bf16_t *A, *W; float* C;
  const size_t na = (size_t)M * KW, nw = (size_t)N * K;
  hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&C, (size_t)M * N * 4 + (1 << 22));
  void* z; hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero)); hipMemset(z, 0, 256);
  bf16_t* hA = (bf16_t*)malloc(na * 2);
  bf16_t* hW = (bf16_t*)malloc(nw * 2);
  unsigned s = 12345;
  for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
  for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; hW[i] = (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (bf16_t)((s >> 31) << 15); }
  hipMemcpy(A, hA, na * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, hW, nw * 2, hipMemcpyHostToDevice);
  printf("3x3 conv 320 -> 320 on 8 x 64 x 64 (M=%d N=%d K=%d), %d tiles of 256 x 160\n", M, N, K, (M / 256) * (N / BN));
  constexpr int SM3 = 3 * (256 * 128 + BN * 128);
  constexpr int SMP = 2 * 392 * 128 + 3 * BN * 128;
  for (int rep = 0; rep < 3; ++rep) {
    const bool chk = rep == 0;
    run(k_pp, SM3, 512, A, W, C, M, N, K, "ping-pong, per-tap A tiles (shipped structure; GEMM stand-in)", hA, hW, chk, 0);
    run(k_patch<0>, SMP, 512, A, W, C, M, N, K, "ping-pong, 9-tap input patch in LDS (real convolution)", hA, hW, chk, 1);
    run(k_patch2<0>, SMP, 512, A, W, C, M, N, K, "  version 2: addresses prepared in the MMA phase", hA, hW, chk, 1);
    run(k_patch<2>, SMP, 512, A, W, C, M, N, K, "  version 1, patch piece behind the weight pieces (one more step in flight)", hA, hW, chk, 1);
    run(k_patch<6>, SMP, 512, A, W, C, M, N, K, "  version 1, the whole next patch at tap 0, behind the weight pieces", hA, hW, chk, 1);
    if (rep == 2) run(k_patch<1>, SMP, 512, A, W, C, M, N, K, "  ablation: no patch pieces (timing only)", hA, hW, false, 1);
    if (rep == 2) run(k_patch2<1>, SMP, 512, A, W, C, M, N, K, "  ablation, version 2: no patch pieces (timing only)", hA, hW, false, 1);
  }
  return 0;
}
