// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue(alpha * A[M,K] . W[N,K]^T)
//
// Tile: 128 x BN x 64 per 256-thread workgroup (4 waves as 2x2, each wave 64 x BN/2 via
// v_mfma_f32_16x16x32_bf16), two workgroups per CU; the 3x3 convolutions run on a 256 x 160 tile of 8 waves, one workgroup
// per CU, 3-deep ring, with a ping-pong main loop (see "PING-PONG" in the kernel).  Both operands are K-contiguous, so both LDS tiles are
// [rows][64 bf16] = 128 B rows filled by direct-to-LDS DMA (global_load_lds_dwordx4: one
// wave instruction = 1 KiB = 8 tile rows) into a 2-deep ring: the DMA of K step i+1 flies while
// step i is multiplied.  The DMA destination is lane-linear, so the bank-conflict swizzle
// (16-B slot ^= row&7) is applied on the per-lane SOURCE address and again on the ds_read_b128
// address (conflict-free for the 16-lane read groups).
// The MFMA is issued "swapped" (W fragment as the A operand) so every lane ends up with 4
// consecutive N columns of one M row: bias/residual/activation run on float4s and the
// store is 8 B of bf16 per lane.
//
// The conv path computes the im2col row pointers on the fly: NHWC activations make each
// 64-wide K step a contiguous 128-B run inside one (tap, pixel); per K step the tap offset is a
// wave-uniform scalar added to a per-row centre offset, padding taps read a zero page.
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher places workgroup b on XCD b % 8, so ids are
// remapped (bijectively) to give every XCD a contiguous range of tiles — the N tiles of one M range
// and the 3x3 halo rows of neighbouring M ranges then hit the same 4 MiB L2.
#include "ops.h"
#include <stdlib.h>

#define BM_HOST 128       // rows of the 4-wave tile, for the host-side split-K heuristic; the kernel's BM is 32 * NWV
#define BK 64
#define GEMM_THREADS 256
#define LN_MAX_PLANES 20   // row-sum planes a folded-LayerNorm consumer can add: 2 per N tile of the producer, or one per 64 columns of a split-K reducer (C <= 1280)

// zeros read by padding taps; a row's pointer advances 128 B per K step within a (tap, source) segment, so the page
// covers the longest segment (ZERO_PAGE_STEPS K steps) plus one 128-B row
#define ZERO_PAGE_STEPS 127
__device__ __attribute__((aligned(256))) uint32_t g_zero_page_storage[(ZERO_PAGE_STEPS + 1) * 32];

const bf16_t* gill_zero_page() {
  static const bf16_t* p = nullptr;
  if (!p) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_zero_page_storage)) == hipSuccess) {
      (void)hipMemset(d, 0, sizeof(g_zero_page_storage));
      p = (const bf16_t*)d;
    }
  }
  return p;
}

struct GemmDev {
  GemmArgs a;
  const bf16_t* zero;
  int ksteps;        // K / 64
  int ksteps_per_split;
  int tiles_n;
  int npw;           // N tiles one workgroup walks back to back (plain GEMM, no split-K): the LDS ring keeps flowing
  int groups_n;      // cdiv(tiles_n, npw)
  int nwv;           // waves per workgroup: 4 (128-row tile) or 2 (64-row tile)
  int mi;            // 16-row M sub-tiles per wave: 4, or 2 (nwv = 4: the 64-row tile on four waves)
  int kt;            // K elements per ring stage: 64 (2-deep ring) or 32 (4-deep ring, 128-row tiles only)
  int n_major;       // 1: consecutive tile ids walk M first (an XCD's contiguous id range = a range of N tiles over every M tile)
  int tiles_m;
  int xb_m, xb_n;    // > 0: XCD BLOCK MAP — XCD x owns an xb_m x xb_n block of the (M tile, N group) grid (see xcd_block_pick()); 0: the range map above
};

__device__ __noinline__ float gelu_erf_call(float v) { return gelu_erf(v); }  // keeps erff out of the unrolled epilogue

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_GELU) return gelu_erf_call(v);
  if (act == ACT_QUICK_GELU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
  return v;
}

// Generic epilogue for 4 consecutive columns n..n+3 of row m (n % 4 == 0, n + 3 < N): used by the split-K reducer.
// (Loading bias / row vector / residual up front, in flight with the partials, measured SLOWER: loop 518.3 -> 520.1 ms.)
__device__ __forceinline__ void store4(const GemmArgs& p, int m, int n, float v0, float v1, float v2, float v3,
                                       float* fin = nullptr) {
  float v[4] = {v0 * p.alpha, v1 * p.alpha, v2 * p.alpha, v3 * p.alpha};
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (p.rowvec) {
    const float4 b = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_bstride + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (p.resid) {
    if (p.resid_f32) {
      const float4 r = *reinterpret_cast<const float4*>((const float*)p.resid + (size_t)m * p.ldr + n);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    } else {
      const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)p.resid + (size_t)m * p.ldr + n);
      v[0] += bf2f((bf16_t)(r.x & 0xffff)); v[1] += bf2f((bf16_t)(r.x >> 16));
      v[2] += bf2f((bf16_t)(r.y & 0xffff)); v[3] += bf2f((bf16_t)(r.y >> 16));
    }
  }
  if (p.act != ACT_NONE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], p.act);
  }
  if (fin) { fin[0] = v[0]; fin[1] = v[1]; fin[2] = v[2]; fin[3] = v[3]; }
  if (p.out_mode == OUT_BF16) {
    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
  } else if (p.out_mode == OUT_F32) {
    *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  } else {  // OUT_QKV
    const int hd = p.heads * p.dp;
    const int seg = p.seg_base + n / hd;
    const int within = n % hd;
    const int h = within / p.dp, dd = within % p.dp;
    const int b = m / p.ntok, t = m % p.ntok;
    if (seg == 0) {
      uint2 o; o.x = pack_bf2(v[0] * p.qscale, v[1] * p.qscale); o.y = pack_bf2(v[2] * p.qscale, v[3] * p.qscale);
      *reinterpret_cast<uint2*>(p.Cq + ((size_t)(b * p.heads + h) * p.ntok_pad_q + t) * p.dp + dd) = o;
    } else if (seg == 1) {
      uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(p.Ck + ((size_t)(b * p.heads + h) * p.ntok_pad_kv + t + p.kv_tok_offset) * p.dp + dd) = o;
    } else {
      bf16_t* base = p.Cvt + ((size_t)(b * p.heads + h) * p.dpv + dd) * p.ntok_pad_kv + t + p.kv_tok_offset;
#pragma unroll
      for (int i = 0; i < 4; ++i) base[(size_t)i * p.ntok_pad_kv] = f2bf(v[i]);
      if (dd + 4 == p.dp && p.dpv > p.dp) base[(size_t)4 * p.ntok_pad_kv] = (bf16_t)0x3F80;
    }
  }
}

// REDUCE + GROUPNORM.  Where a split-K GEMM's output feeds a single-source GroupNorm (every 3x3 convolution of UNet levels 2-3 at
// the 8-sample batch: conv1 -> norm2 inside a ResnetBlock2D, and a block's last GEMM -> the next block's first norm), the reducer is
// the first place that holds complete output values — and a unit that owns ALL rows of one sample for a run of whole groups holds
// complete GroupNorm statistics too.  Unit = (sample b, W output columns from col0 = whole groups of fn_cg channels): sums the split slices
// in slice order, applies the GEMM epilogue (bias, per-sample row vector, bf16 residual), optionally stores the raw tensor, totals
// {sum, sum of squares} per group in a fixed order (thread -> 16 row lanes -> the group's column quads), and writes the normalised
// (+ SiLU) tensor: the GroupNorm-apply launch of that norm (5.5-7.5 us each, 24 per forward) and its read of the raw tensor disappear.
// Also files the fused-statistics partials of the raw output (gn_stats: one slab = the whole sample) for a later two-source consumer.
// fn_Y == nullptr: the plain finish (raw tensor + partials only).
// RPT rows per thread: rows_per_batch = 16 * RPT (256 -> 16, 64 -> 4).  W columns (80 | 40): W / 4 column quads x 16 row lanes = 4 W threads
// do the work; a caller with more threads (the COOP finish inside gemm_kernel: 512) passes them all — they only take part in the barriers.
// Two callers, one body: the stand-alone reducer kernel below (one unit per block) and gemm_kernel's EPI 7 (one unit per workgroup, after the
// arrival counter of its group has filled) — so the in-kernel finish is bit-identical to the reducer launch it replaces.
// poison: the caller gave up waiting for its group (COOP_SPIN_LIMIT): every output of the unit becomes NaN.
template <int RPT, int W>
__device__ __forceinline__ void splitk_finish_unit(const GemmArgs& p, const int b, const int col0, const int tidx, float2 (*part)[W / 4], float2* quad,
                                                   float2* mr, const bool poison) {
  constexpr int QW = W / 4, ROWS = 16 * RPT;
  const bool act = tidx < 4 * W;
  const int tx = tidx % QW, ty = act ? tidx / QW : 0;
  const int n = col0 + tx * 4;
  const int mbase = b * ROWS;
  const bool norm = p.fn_Y != nullptr;
  // every global load of the unit goes out before the first use: the partials of the first slices, the epilogue's vectors, the
  // residual rows, the norm's parameters — one memory round trip in front of the reduction, not four
  float4 v[RPT];
  auto row_of = [&](int r) -> int { return mbase + ty + 16 * r; };     // row of the thread's r-th value
  float ps = 0.f, pq = 0.f;
  uint2 raw[RPT];
  float4 gam = make_float4(0, 0, 0, 0), bet = make_float4(0, 0, 0, 0);
  if (act) {
  float4 add = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
  const float4 rv = p.rowvec ? *reinterpret_cast<const float4*>(p.rowvec + (size_t)b * p.rowvec_bstride + n) : make_float4(0, 0, 0, 0);
  if (norm) { gam = *reinterpret_cast<const float4*>(p.fn_gamma + n); bet = *reinterpret_cast<const float4*>(p.fn_beta + n); }
  uint2 rr[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    rr[r] = p.resid ? *reinterpret_cast<const uint2*>((const bf16_t*)p.resid + (size_t)row_of(r) * p.ldr + n) : make_uint2(0u, 0u);
  const float* src = p.ws + (size_t)(mbase + ty) * p.N + n;
  const size_t zstride = (size_t)p.M * p.N, rstride = (size_t)16 * p.N;
  // slices are ADDED in slice order whatever the order their loads go out in.  Few rows per thread (RPT = 4: the 64-row samples of UNet level 3,
  // split up to eight ways): the first eight slices' loads all at once — 32 independent 16-B loads per thread, ONE memory round trip — where a loop
  // over the slices is one round trip per slice (round 6: inside gemm_kernel's COOP finish one workgroup per CU has no neighbour to hide them: the
  // level-3 finish took 21 us that way; clamped slice index + predicated add, so absent slices cost a redundant load and nothing else).
  constexpr int CH = (RPT <= 4) ? 8 : 2;
  {
    float4 t[CH][RPT];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int zz = (u < p.splitk) ? u : p.splitk - 1;
#pragma unroll
      for (int r = 0; r < RPT; ++r) t[u][r] = *reinterpret_cast<const float4*>(src + (size_t)zz * zstride + (size_t)r * rstride);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = t[0][r];
#pragma unroll
    for (int u = 1; u < CH; ++u) {
      if (u < p.splitk) {      // (uniform; splitk >= 2)
#pragma unroll
        for (int r = 0; r < RPT; ++r) { v[r].x += t[u][r].x; v[r].y += t[u][r].y; v[r].z += t[u][r].z; v[r].w += t[u][r].w; }
      }
    }
  }
  for (int z = CH; z < p.splitk; ++z) {
    float4 t[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) t[r] = *reinterpret_cast<const float4*>(src + (size_t)z * zstride + (size_t)r * rstride);
#pragma unroll
    for (int r = 0; r < RPT; ++r) { v[r].x += t[r].x; v[r].y += t[r].y; v[r].z += t[r].z; v[r].w += t[r].w; }
  }
  add.x += rv.x; add.y += rv.y; add.z += rv.z; add.w += rv.w;
  if (poison) add.x = add.y = add.z = add.w = __builtin_nanf("");
  // STATISTICS SOURCE: every FUSED producer of GroupNorm statistics in this file — the in-kernel epilogue (gs / gq), the plain reducer
  // (store4()'s `fin`) and this unit — sums the fp32 values BEFORE their rounding to bf16, and every consumer normalises the ROUNDED tensor (below:
  // raw[r], what a stand-alone apply pass would read).  The fused and unfused forms of one layer therefore see the same statistics source and differ
  // only in the (fixed) order of the fp32 additions, which is why the fused and unfused forms are held to closeness, not equality.  Only the stand-alone
  // statistics kernel of two-source (skip-concat) inputs reads rounded values — it has nothing else.  Variance: E[x^2] - E[x]^2 in fp32 over a
  // (sample, group) slab of <= 256 x 80 values of O(1): cancellation costs ~1e-6 relative at |mean| ~ sigma, far below the bf16 output step.
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int m = row_of(r);
    v[r].x += add.x; v[r].y += add.y; v[r].z += add.z; v[r].w += add.w;
    v[r].x += bf2f((bf16_t)(rr[r].x & 0xffff)); v[r].y += bf2f((bf16_t)(rr[r].x >> 16));        // (zeros without a residual)
    v[r].z += bf2f((bf16_t)(rr[r].y & 0xffff)); v[r].w += bf2f((bf16_t)(rr[r].y >> 16));
    ps += (v[r].x + v[r].y) + (v[r].z + v[r].w);
    pq += (v[r].x * v[r].x + v[r].y * v[r].y) + (v[r].z * v[r].z + v[r].w * v[r].w);
    raw[r].x = pack_bf2(v[r].x, v[r].y); raw[r].y = pack_bf2(v[r].z, v[r].w);
    if (p.C) *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = raw[r];
  }
  part[ty][tx] = make_float2(ps, pq);
  }
  __syncthreads();
  if (tidx < QW) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float2 t = part[r][tidx]; a += t.x; q += t.y; }
    quad[tidx] = make_float2(a, q);
  }
  __syncthreads();
  if (norm) {
    const int qpg = p.fn_cg / 4;             // column quads per group
    const int ngrp = W / p.fn_cg;
    if (tidx < ngrp) {
      float a = 0.f, q = 0.f;
      for (int t = 0; t < qpg; ++t) { const float2 u = quad[tidx * qpg + t]; a += u.x; q += u.y; }
      const float inv_n = 1.f / ((float)p.fn_cg * (float)ROWS);
      const float sm = a * inv_n, sq = q * inv_n;
      const float var = fmaxf(sq - sm * sm, 0.f);
      mr[tidx] = make_float2(sm, rsqrtf(var + p.fn_eps));
    }
  }
  if (p.gn_stats) {      // partials of the raw output for a later (two-source) GroupNorm: bins of gn_cg channels, one slab per sample
    const int qpb = p.gn_cg / 4, nbin = W / p.gn_cg;
    const int t2 = tidx - 32;
    if (t2 >= 0 && t2 < 2 * nbin) {
      const int which = t2 & 1, lb = t2 >> 1;
      float a = 0.f;
      for (int t = 0; t < qpb; ++t) { const float2 u = quad[lb * qpb + t]; a += which ? u.y : u.x; }
      p.gn_stats[((size_t)b * p.gn_groups + col0 / p.gn_cg + lb) * 2 + which] = a;
    }
  }
  __syncthreads();
  if (norm && act) {
    const float2 g = mr[(tx * 4) / p.fn_cg];
    const float s0 = g.y * gam.x, s1 = g.y * gam.y, s2 = g.y * gam.z, s3 = g.y * gam.w;
    const float h0 = bet.x - g.x * s0, h1 = bet.y - g.x * s1, h2 = bet.z - g.x * s2, h3 = bet.w - g.x * s3;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int m = row_of(r);
      // (the consumer of the unfused path reads the bf16 tensor: normalise the rounded values)
      float o0 = fmaf(bf2f((bf16_t)(raw[r].x & 0xffff)), s0, h0), o1 = fmaf(bf2f((bf16_t)(raw[r].x >> 16)), s1, h1);
      float o2 = fmaf(bf2f((bf16_t)(raw[r].y & 0xffff)), s2, h2), o3 = fmaf(bf2f((bf16_t)(raw[r].y >> 16)), s3, h3);
      if (p.fn_silu) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
      uint2 o; o.x = pack_bf2(o0, o1); o.y = pack_bf2(o2, o3);
      *reinterpret_cast<uint2*>(p.fn_Y + (size_t)m * p.N + n) = o;
    }
  }
}

// ---- COOP: hand-offs between the co-resident workgroups of ONE launch (cdna_hip_programming.md Guideline 16, recipe R1; prices in
// MI355X_MICROARCH.md "publish-large" / "splitk-seam").  Payload goes out WRITE-THROUGH (sc1: nothing left dirty, no release fence), every storing
// wave drains its own stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane arrives on the group's counter (agent-scope atomic)
// and polls it relaxed with s_sleep; readers then take ONE agent-scope acquire (split-K: bulk plain loads) or read the few words with sc1 loads
// (GroupNorm partials).  Counters are zeroed once per forward (unet.hip), every slot is used by exactly one launch: the target is the group size.
// The spin is bounded: a workgroup that gives up poisons its outputs with NaN (every test and bench.py check finiteness) instead of hanging the GPU.
__device__ __forceinline__ void st_wt_f32x4(float* p, const f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_wt_f32(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_wt_f32(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
#define COOP_SPIN_LIMIT (1u << 18)
// give-ups since the last gemm_coop_check(): a waiter that ran out of patience (the GPU is being shared with another process or stream that ALSO
// runs waiting workgroups — the co-residency contract of GemmArgs::coop_ctr is broken) counts itself here before it poisons its outputs
__device__ unsigned g_coop_giveups;
__device__ __forceinline__ void coop_note_giveup() { __hip_atomic_fetch_add(&g_coop_giveups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_arrive(unsigned* ctr) { __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool coop_wait(unsigned* ctr, unsigned target) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if (spins > COOP_SPIN_LIMIT) { coop_note_giveup(); return false; }
    __builtin_amdgcn_s_sleep(2);
  }
}
__device__ __forceinline__ bool coop_arrive_wait(unsigned* ctr, unsigned target) {
  __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if (spins > COOP_SPIN_LIMIT) { coop_note_giveup(); return false; }
    __builtin_amdgcn_s_sleep(2);
  }
}

// CONV: 0 plain GEMM, 1 conv3x3, 2 conv3x3 with fused nearest-2x upsample (9 taps gathered from the source grid),
//       3 nearest-2x upsample + conv3x3 as FOUR 2x2-tap convolutions of the source grid, one per output parity (see "UPS4")
// EPI : 0 row-major (bf16 / fp32) epilogue, 1 GEGLU, 2 raw fp32 split-K partials, 3 QKV head-major scatter,
//       4 = 0 without activation / fp32 output / fp32 residual (the UNet's plain GEMMs: half the epilogue's code and branches)
//       5 folded-LayerNorm scores + softmax over each wave's 80 columns (OUT_SOFTMAX80; BN = 160)
//       6 = 0 on the ping-pong conv tiles + the consuming GroupNorm finished in-kernel (COOP, splitk == 1: see GemmArgs::coop_ctr)
//       7 = 2 + the split-K finish in-kernel (COOP): partials published write-through, one (sample, 40-column) unit finished per workgroup
// STAGES: depth of the LDS ring.  2: two workgroups per CU hide each other's DMA waits.  3: one workgroup per CU,
//         the DMA of K steps i+1 AND i+2 is in flight while step i is multiplied (counted s_waitcnt vmcnt(N), raw
//         s_barrier) — for grids of ~one workgroup per CU where co-residency cannot do the hiding.
// folded LayerNorm: rstd and mean*rstd of A's row m from the producer's {sum, sum of squares}
__device__ __forceinline__ void ln_row_factors(const GemmArgs& p, int m, float& rr, float& rm) {
  // the producer's per-plane partial sums of this row, added in plane order (fixed order: bit-reproducible); the loads go out four
  // planes at a time (predicated), not one L2 round trip per plane — and not LN_MAX_PLANES predicated loads per row: a level-0
  // consumer reads 4 planes (the prologue of the GEGLU / QKV kernels was 80 load slots per lane for them)
  const int R = p.ln_rows ? p.ln_rows : p.M;
  if (m >= R) m -= R;
  const float* base = p.ln_stats + (size_t)m * 2;
  const size_t pstride = (size_t)R * 2;
  float2 st = make_float2(0.f, 0.f);
  for (int pl = 0; pl < p.ln_planes; pl += 4) {      // (wave-uniform trip count)
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = (pl + u < p.ln_planes) ? *reinterpret_cast<const float2*>(base + (size_t)(pl + u) * pstride) : make_float2(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) { st.x += v[u].x; st.y += v[u].y; }
  }
  const float invk = 1.f / (float)p.K;
  const float mean = st.x * invk;
  const float var = fmaxf(st.y * invk - mean * mean, 0.f);
  rr = rsqrtf(var + p.ln_eps);
  rm = mean * rr;
}

// NWV: waves per workgroup.  4 -> 128-row tile (waves 2 x 2).  2 -> 64-row tile (waves 1 x 2) for problems with so few
// 128-row tiles that half the CUs would idle (UNet level 2-3 projections: M = 2048 / 512).
// 8 -> 256-row tile (waves 4 x 2, one workgroup per CU).  BN = 160, STAGES = 3: the convolutions' ping-pong kernel.
// BN = 256 (GILL_GEMM_BIG only): a K step fetches 64 KiB for 8.4 MFLOP — half the
// bytes per FLOP of the 128 x 160 tile, whose 36.8 KiB per 2.6 MFLOP is 89 % of the 64 B/clk a CU can pull from L2 at the
// MFMA peak — for the wide plain GEMMs (GEGLU, QKV) with enough 256 x 256 tiles to fill the chip.
// KT: K elements per ring stage (64 | 32).  With a 2-deep ring the loads of stage t+1 can only be issued after the barrier of
// stage t (their buffer is free from there): prefetch distance = one stage.  KT = 32 cuts the SAME 73.7 KiB of LDS into a
// 4-deep ring of half-size stages: loads are issued three stages (1.5 K steps of 64) ahead and waited for with counted
// s_waitcnt vmcnt(N), at the price of a barrier per 32-wide stage.  Measured: that price is higher than the gain (the
// launcher keeps KT = 64 unless GILL_GEMM_KT = 32).
// MI: 16-row M sub-tiles per wave (4 | 2).  MI = 2 with NWV = 4 is a 64-row tile on FOUR waves (2 x 2, wave tile 32 x BN/2) for the
// small plain GEMMs: the same 48 KiB of LDS as the 2-wave 64 x 128 tile (three workgroups per CU), twice the waves per CU.
// (The timing-only ablation guards of round 4 — PP_ABL: no fragment reads / no LDS-DMA / no MFMAs / no bookkeeping / no epilogue / no barriers —
// are gone from this file; their table is profiles/r04_pingpong_ablations.md, the bare loops live on in tools/ubench/gemm_loop.hip and gemm_ws.hip.)
// The kernel body is a device function of the workgroup's coordinates (gemm_kernel below passes blockIdx / gridDim), so that a PERSISTENT launch can
// run several GEMMs back to back in one workgroup (tools/ubench/persist_resnet.hip, round 6: the go / no-go prototype of one launch per ResnetBlock2D).
// SEAM (prototype only; 0 in every product instantiation — all of it compiles away): the tile is one PHASE of such a launch.  On entry it waits on
// the arrival counter of the M-tile group whose rows it reads (seam_in) — after issuing the WEIGHT pieces of its first ring stages when
// `prefetch` is set (weights do not depend on the previous phase: MI355X_MICROARCH "prefetch-credit") — and on exit (EPI 7) it publishes its
// finished units and arrives on seam_out.
struct GemmTileCtx {
  int bx, by, bz, gdx, gdy;
  unsigned* seam_in = nullptr; unsigned seam_in_target = 0; int seam_in_mtg = 1; int prefetch = 0;
  unsigned* seam_out = nullptr;
};
template <int NWV, int BN, int CONV, int EPI, int STAGES, int KT, int MI, int SEAM = 0>
__device__ __forceinline__ void gemm_tile(const GemmDev& d, const GemmTileCtx cx) {
  constexpr int BM = (NWV / 2) * MI * 16;
  static_assert(MI == 4 || (MI == 2 && NWV == 4 && CONV == 0 && (STAGES == 2 || STAGES == 3) && KT == 64) ||
                    (MI == 2 && NWV == 8 && BN == 160 && STAGES == 3 && KT == 64) || (MI == 2 && NWV == 8 && BN == 64 && STAGES == 6 && CONV == 0 && KT == 64) ||
                    (MI == 2 && NWV == 8 && BN == 128 && STAGES == 2 && CONV == 0 && KT == 64),
                "MI = 2: the 4-wave 64-row plain tile, the 128 x 160 ping-pong tile (8 waves of 32 x 80), or the 128 x 64 streaming tile on 8 waves of 32 x 32");
  constexpr int NT = BN / 32;  // 16-wide N sub-tiles per wave (wave covers BN/2 columns)
  constexpr bool PP = (NWV == 8 && BN == 160 && STAGES == 3 && KT == 64);   // ping-pong main loop (see there)
  // STREAM64 (host side: gemm_stream64_weights()): W is stored as [N / 64][K / 64][64 rows][64 k] — a K step's 64 x 64 weight tile is ONE
  // contiguous 8 KiB run (whole DRAM pages) instead of 64 rows of 128 B at a stride of K — and is streamed with the non-temporal policy
  constexpr bool WBLK = (BN == 64 && STAGES == 6 && CONV == 0);
  static_assert(KT == 64 || KT == 32, "stage depth");
  constexpr int LPR = KT / 8;          // lanes (16-B chunks) per tile row
  constexpr int RPI = 64 / LPR;        // tile rows one 1-KiB LDS-DMA wave-instruction covers (8 | 16)
  constexpr int KK = KT / 32;          // 32-wide MFMA k steps per stage
  // XOR swizzle of the 16-B chunk index by the row: conflict-free ds_read_b128 of 16 consecutive rows
  // (KT = 64: rows are 128 B, chunk ^= row & 7; KT = 32: rows are 64 B, chunk ^= (row >> 2) & 3)
#define SWZ(row) (KT == 64 ? ((row) & 7) : (((row) >> 2) & 3))
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  // layout: [buf][A tile BM*KT | B tile BN*KT]
  constexpr int A_ELEMS = BM * KT;
  constexpr int B_ELEMS = BN * KT;
  constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;

  const GemmArgs& p = d.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware (bijective) remap of the workgroup id
  int tile;
  {
    const int nwg = cx.gdx, bid = cx.bx;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // weight-heavy problems (N*K > the activation matrix: UNet levels 2-3, the LM) give each XCD a range of N tiles, so every
  // weight is fetched into one L2 only; activation-heavy ones a range of M tiles (N tiles and 3x3 halo rows share the L2)
  int gn = d.n_major ? tile / d.tiles_m : tile % d.groups_n;
  int tm = d.n_major ? tile % d.tiles_m : tile / d.groups_n;
  if (d.xb_m > 0) {
    // XCD block map (round 6): the 8 XCDs tile the (M tile, N group) grid with xb_m x xb_n blocks, so that an XCD's L2 holds xb_m tiles' worth of
    // activations (read once, re-used by the 3x3 taps and by the block's N groups) and 1 / (groups_n / xb_n) of the weights — the two range maps
    // above are its extreme cases (all of M x 1/8 of N: every L2 pulls the whole activation tensor per tap; 1/8 of M x all of N: every L2 pulls
    // every weight).  gridDim.x == tiles_m * groups_n == 8 * xb_m * xb_n (xcd_block_pick()).
    const int bid = cx.bx, xcd = bid & 7, idx = bid >> 3;
    const int blocks_m = d.tiles_m / d.xb_m;
    const int bmi = xcd % blocks_m, bni = xcd / blocks_m;
    tm = bmi * d.xb_m + idx % d.xb_m;
    gn = bni * d.xb_n + idx / d.xb_m;
  }
  const int m0 = tm * BM;
  const int tn_first = gn * d.npw;
  // (convolutions and split-K partials: one N tile per workgroup — gemm_launch_bn() — known at compile time, so that the K walk's
  // state is dead by the epilogue instead of being carried around it for a next tile that never comes)
  const int ntl = (CONV != 0 || EPI == 2 || EPI == 7) ? 1 : ((d.tiles_n - tn_first < d.npw) ? d.tiles_n - tn_first : d.npw);   // N tiles of this workgroup
  int n0 = tn_first * BN;
  const int z = cx.by;
  const int kt_beg = z * d.ksteps_per_split;
  int kt_end = kt_beg + d.ksteps_per_split;
  if (kt_end > d.ksteps) kt_end = d.ksteps;
  const int nsteps = kt_end - kt_beg;
  // UPS4 (CONV = 3).  conv3x3(nearest_2x(x)) reads, for the output pixel (2y + py, 2x + px), only the 2 x 2 source pixels
  // (y + py - 1 + a, x + px - 1 + b), a, b in {0, 1}: the 9 taps fall onto them in groups whose weights can be summed once at load
  // (conv_weight_relayout_ups4_launch: W4[class][n][a*2+b][c]; a summed group is always in or out of the image as a whole).  So the
  // layer is four stride-1 2x2-tap convolutions of the SOURCE grid — K = 4 Cin instead of 9 Cin, the same linear map with 2.25x
  // fewer multiply-adds.  blockIdx.z = parity class; p.M = source pixels (rows of one class); output row of source row m:
  // (b * OH + 2y + py) * OW + 2x + px.
  const int cls = (CONV == 3) ? cx.bz : 0;
  const int ups_py = cls >> 1, ups_px = cls & 1;

  // ---- per-lane staging geometry: instruction i of wave w fills tile rows (i*NWV+w)*RPI .. +RPI
  const int srow = lane / LPR;                            // row within the instruction's row group
  const int schunk = (lane % LPR) ^ SWZ(srow);            // logical 16-B chunk this lane fetches (source-side swizzle;
                                                          // group bases are multiples of RPI, so SWZ(row) == SWZ(srow))

  // A rows (4 per lane): element offset of the row start (plain) / of the centre input pixel (conv) in each source,
  // plus conv flag bits {1: dy=-1 in range, 2: dy=+1, 4: dx=-1, 8: dx=+1, 16: oy odd, 32: ox odd}
  constexpr int AI = BM / (RPI * NWV);                    // A instructions per wave per stage (4 | 2)
  int a_off1[AI], a_off2[AI], a_fl[AI], a_pc[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int row = (i * NWV + w) * RPI + srow;
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    a_fl[i] = 0; a_pc[i] = 0;
    if constexpr (CONV == 0) {
      a_off1[i] = m * p.lda + schunk * 8;
      a_off2[i] = m * p.lda2 + schunk * 8;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw;
      const int r = m - b * ohw;
      const int oy = r / p.OW;
      const int ox = r - oy * p.OW;
      int fl = 0, pc;
      if constexpr (CONV == 3) {      // rows are SOURCE pixels: a stride-1 walk of the source grid
        const int shw = p.IH * p.IW;
        const int sb = m / shw;
        const int sr = m - sb * shw;
        const int sy = sr / p.IW, sx = sr - sy * p.IW;
        pc = m;
        fl |= (sy - 1 >= 0) ? 1 : 0; fl |= (sy + 1 < p.IH) ? 2 : 0;
        fl |= (sx - 1 >= 0) ? 4 : 0; fl |= (sx + 1 < p.IW) ? 8 : 0;
      } else if constexpr (CONV == 2) {
        pc = (b * p.IH + (oy >> 1)) * p.IW + (ox >> 1);
        fl |= (oy - 1 >= 0) ? 1 : 0; fl |= (oy + 1 < p.OH) ? 2 : 0;
        fl |= (ox - 1 >= 0) ? 4 : 0; fl |= (ox + 1 < p.OW) ? 8 : 0;
        fl |= (oy & 1) << 4; fl |= (ox & 1) << 5;
      } else {
        const int cy = oy * p.stride, cx = ox * p.stride;
        pc = (b * p.IH + cy) * p.IW + cx;
        fl |= (cy - 1 >= 0) ? 1 : 0; fl |= (cy + 1 < p.IH) ? 2 : 0;
        fl |= (cx - 1 >= 0) ? 4 : 0; fl |= (cx + 1 < p.IW) ? 8 : 0;
      }
      a_fl[i] = fl;
      a_pc[i] = m;          // output pixel == input pixel of the fused 1x1 shortcut segment (stride 1 only)
      a_off1[i] = pc * p.K1 + schunk * 8;
      a_off2[i] = pc * (p.Cin - p.K1) + schunk * 8;
    }
  }
  // W rows: groups of RPI rows dealt to the waves round robin; a wave's last instruction may have no group (KT = 32, BN = 160:
  // 10 groups over 4 waves)
  constexpr int WG = BN / RPI;                  // row groups of the W tile
  constexpr int WI = (WG + NWV - 1) / NWV;      // W instructions per wave per stage (the last one predicated)
  static_assert(BN % RPI == 0, "whole row groups");
  const bool w_last_ok = ((WI - 1) * NWV + w) < WG;       // wave-uniform
  const int lw = AI + (w_last_ok ? WI : WI - 1);           // LDS-DMA instructions this wave issues per stage
  const bf16_t* w_ptr[WI];
  int n_issue = n0;        // first column of the N tile being staged
  // WIDE EPILOGUE STORES.  The MFMA (W fragment as the A operand) leaves a lane with 4 consecutive output columns of one row per
  // 16-wide sub-tile j; stored as they are, a wave instruction writes 16 rows x 32 B.  Which weight row sits in which LDS tile row
  // is free (the DMA source address is per lane), so tile row (j, x) of a wave's half is loaded with column c(j, x) such that the
  // sub-tiles j, j + 1 of a lane hold 8 CONSECUTIVE columns: c = (j >> 1) * 32 + (x >> 2) * 8 + (j & 1) * 4 + (x & 3) — one 16-B
  // store per lane, 64 contiguous bytes per row per instruction, half the store instructions (a last unpaired sub-tile, NT odd,
  // keeps c = j * 16 + x).  GEGLU (EPI 1; weight rows come in [16 value | 16 gate] blocks of 16 outputs): value and gate
  // sub-tiles of two pairs are laid out so that the lane's two products are 8 consecutive OUTPUT columns.
  auto tile_col = [&](int R) -> int {    // LDS tile row R -> column of the tile whose weight row it holds
    const int half = R / (BN / 2), q = R - half * (BN / 2);
    const int j = q >> 4, x = q & 15;
    int c;
    if constexpr (EPI == 1) {
      static_assert(EPI != 1 || NT % 4 == 0, "GEGLU: whole value/gate quads per wave");
      const int o = (x >> 2) * 8 + ((j >> 1) & 1) * 4 + (x & 3);          // output column within the quad's 32
      c = (j >> 2) * 64 + (o >> 4) * 32 + (o & 15) + (j & 1) * 16;       // physical: [16 value | 16 gate] per 16 outputs
    } else {
      c = ((j | 1) < NT) ? (j >> 1) * 32 + (x >> 2) * 8 + (j & 1) * 4 + (x & 3) : q;
    }
    return half * (BN / 2) + c;
  };
  // 64 x 64-blocked weights on the GENERAL tiles too (plain GEMMs, GemmArgs::w_blk64 with more rows than the STREAM64 tile is for — ADVICE r05): the
  // tile's weight rows come out of N / 64 blocks, a K step ahead is the next 8 KiB block of each
  const bool wblk = WBLK || (CONV == 0 && p.w_blk64 != 0);
  const int w_step = wblk ? 4096 : KT;
  const bf16_t* Wb = p.W + (p.wb_rows ? (size_t)(m0 / p.wb_rows) * (size_t)p.wb_stride : (size_t)0);     // per-sample weights (GemmArgs::wb_rows)
  auto w_setup = [&]() {
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int row = tile_col((i * NWV + w) * RPI + srow);
      int n = n_issue + row;
      if (n > p.N - 1) n = p.N - 1;
      if (wblk) w_ptr[i] = Wb + ((size_t)(n >> 6) * d.ksteps + kt_beg) * 4096 + (n & 63) * 64 + schunk * 8;     // (N % 64 == 0: n is never clamped across a block)
      else w_ptr[i] = Wb + ((size_t)cls * p.N + n) * p.K + schunk * 8 + kt_beg * KT;
    }
  };
  w_setup();
  const bf16_t* zero_lane = d.zero + schunk * 8;

  // ---- K-step addressing is incremental: K is walked in (tap, source) segments inside which every row pointer just
  // advances by 64 elements; seg_setup() (wave-uniform control flow, once per segment) recomputes the 4 A row pointers
  const bf16_t* a_ptr[AI];
  int seg_left = 0;
  auto seg_setup = [&](int k0) {
    if constexpr (CONV == 0) {
      const bool first = (k0 < p.K1);
      const bf16_t* src = first ? p.A : p.A2;
      const int kk = first ? k0 : k0 - p.K1;
#pragma unroll
      for (int i = 0; i < AI; ++i) a_ptr[i] = src + (first ? a_off1[i] : a_off2[i]) + kk;
      seg_left = ((first ? p.K1 : p.K) - k0) / KT;
    } else {
      if (CONV == 1 && k0 >= 9 * p.Cin) {
        // fused 1x1 "conv_shortcut" segment: K continues over the channels of the raw block input X1 (++ X2) at the
        // centre pixel; the weight rows carry [conv taps | shortcut] back to back
        const int ke = k0 - 9 * p.Cin;
        const bool fx = (ke < p.KX1);
        const bf16_t* xs = fx ? p.X1 : p.X2;
        const int cx = fx ? p.KX1 : (p.KX - p.KX1);
        const int kc = (fx ? ke : ke - p.KX1) + schunk * 8;
#pragma unroll
        for (int i = 0; i < AI; ++i) a_ptr[i] = xs + (size_t)a_pc[i] * cx + kc;
        seg_left = ((fx ? p.KX1 : p.KX) - ke) / KT;
      } else {
        // all wave-uniform: chunk, tap, source tensor, channel offset, tap displacement.  K order: 64-channel chunks, the 9
        // taps inside a chunk (see conv_weight_relayout_chunked_launch): one (chunk, tap) per K step
        int tap, c0;
        if (p.k_chunked) {
          const int chunk = k0 / (9 * BK);
          const int r = k0 - chunk * (9 * BK);
          tap = r / BK;
          c0 = chunk * BK + (r - tap * BK);
        } else {
          tap = k0 / p.Cin;
          c0 = k0 - tap * p.Cin;
        }
        const int ty = tap / 3;
        int dy = ty - 1, dx = tap - ty * 3 - 1;
        if constexpr (CONV == 3) { dy = ups_py - 1 + (tap >> 1); dx = ups_px - 1 + (tap & 1); }    // tap = a * 2 + b of the 2 x 2 window
        const int need = (dy < 0 ? 1 : (dy > 0 ? 2 : 0)) | (dx < 0 ? 4 : (dx > 0 ? 8 : 0));
        const bool first = (c0 < p.K1);
        const bf16_t* src = first ? p.A : p.A2;
        const int cs = first ? p.K1 : (p.Cin - p.K1);
        const int cc = first ? c0 : c0 - p.K1;
        const int delta = (dy * p.IW + dx) * cs + cc;   // used when CONV == 1
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          int off = (first ? a_off1[i] : a_off2[i]);
          if constexpr (CONV == 2) {
            const int ddy = (dy + ((a_fl[i] >> 4) & 1)) >> 1, ddx = (dx + ((a_fl[i] >> 5) & 1)) >> 1;
            off += (ddy * p.IW + ddx) * cs + cc;
          } else {
            off += delta;
          }
          const bool ok = (a_fl[i] & need) == need;
          a_ptr[i] = ok ? src + off : zero_lane;
        }
        seg_left = p.k_chunked ? 1 : ((first ? p.K1 : p.Cin) - c0) / KT;
      }
    }
  };
  int k_issue = kt_beg * KT;     // K coordinate of the next step to stage
  int steps_in_tile = 0;
  // One stage = prepare (pointer set-up when the K walk enters a new N tile / (tap, source) segment) + dma (the LDS-DMA
  // instructions) + post (advance the walk).  The ping-pong loop runs the three in different phases.
  auto issue_prepare = [&]() {
    if (steps_in_tile == nsteps) {   // next N tile of this workgroup: same A rows from the top, next BN weight rows
      steps_in_tile = 0;
      k_issue = kt_beg * KT;
      seg_left = 0;
      n_issue += BN;
      w_setup();
    }
    if (seg_left == 0) seg_setup(k_issue);
  };
  auto issue_dma = [&](int buf) {
    bf16_t* As = smem + buf * BUF_ELEMS;
    bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      bf16_t* l = As + (i * NWV + w) * RPI * KT;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      if (i == WI - 1 && !w_last_ok) break;
      bf16_t* l = Bs + (i * NWV + w) * RPI * KT;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                       (__attribute__((address_space(3))) void*)l, 16, 0, WBLK ? 2 : 0);     // (aux 2 = nt)
    }
  };
  // (SEAM prologue only: the two halves of issue_dma on their own — the W pieces `koff` elements ahead of the walk's current position)
  auto issue_dma_a = [&](int buf) {
    bf16_t* As = smem + buf * BUF_ELEMS;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)(As + (i * NWV + w) * RPI * KT), 16, 0, 0);
  };
  auto issue_dma_w = [&](int buf, int koff) {
    bf16_t* Bs = smem + buf * BUF_ELEMS + A_ELEMS;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      if (i == WI - 1 && !w_last_ok) break;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[i] + koff),
                                       (__attribute__((address_space(3))) void*)(Bs + (i * NWV + w) * RPI * KT), 16, 0, 0);
    }
  };
  auto issue_post = [&]() {
    ++steps_in_tile;
    --seg_left;
#pragma unroll
    for (int i = 0; i < AI; ++i) a_ptr[i] += KT;
#pragma unroll
    for (int i = 0; i < WI; ++i) w_ptr[i] += WBLK ? 4096 : (CONV == 0 ? w_step : KT);
    k_issue += KT;
  };
  auto issue = [&](int buf) {
    issue_prepare();
    issue_dma(buf);
    issue_post();
  };

  f32x4 acc[MI][NT];

  const int wm = w >> 1, wn = w & 1;
  const int frow = lane & 15;       // fragment row within a 16-row sub-tile
  const int fkc = lane >> 4;        // 16-B k chunk within the 32-wide MFMA k step

  static_assert(STAGES >= 2 && STAGES <= 6, "ring depth");
  const int total_steps = ntl * nsteps;
  if constexpr (SEAM != 0) {
    static_assert(SEAM == 0 || (PP && !WBLK), "SEAM: the ping-pong tiles");
    if (cx.seam_in) {
      if (cx.prefetch) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
          if (s < total_steps) issue_dma_w(s, s * KT);
      }
      if (tid == 0) {
        unsigned* c = cx.seam_in + tm / cx.seam_in_mtg;
        for (unsigned spins = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cx.seam_in_target && spins < COOP_SPIN_LIMIT; ++spins)
          __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __builtin_amdgcn_s_barrier();      // (raw: a __syncthreads() here would drain the weight pieces in flight)
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < total_steps) { issue_prepare(); issue_dma_a(s); if (!cx.prefetch) issue_dma_w(s, 0); issue_post(); }
    } else {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < total_steps) issue(s);
    }
  } else {
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < total_steps) issue(s);
  }
  // folded LayerNorm (GEGLU / QKV epilogues): the row factors are loaded here, under the first tile's DMA latency
  float ln_rr[4] = {1.f, 1.f, 1.f, 1.f}, ln_rm[4] = {0.f, 0.f, 0.f, 0.f};   // (first MI entries used)
  if constexpr (EPI == 1 || EPI == 3 || EPI == 5) {
    if (p.ln_stats) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (MI * 16) + i * 16 + frow;
        if (m < p.M) ln_row_factors(p, m, ln_rr[i], ln_rm[i]);
      }
    }
  }
  int buf = 0, buf_issue = STAGES - 1;
  int flat = 0;          // K steps consumed over all N tiles of this workgroup
  // GEGLU / QKV GEMMs walk several N tiles per workgroup with a 2-deep ring: see DEFER in the main loop
  constexpr bool DEFER = (CONV == 0 && STAGES == 2 && !PP && (EPI == 1 || EPI == 3));
  int deferred_buf = -1;
  for (int t = 0; t < ntl; ++t, n0 += BN) {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (PP) {
    // ---- PING-PONG main loop (256 x 160 tile, 8 waves 4 x 2, one workgroup per CU, 3-deep ring).  Waves w and w + 4 share a
    // SIMD and sit in different groups; the groups run half a K step apart, so in every phase ONE wave of each SIMD issues its
    // step's 40 MFMAs while the other does its step's memory work (18 ds_read_b128 fragment loads, then its 6-7 LDS-DMA pieces of
    // stage k + 2 with their address bookkeeping).  Phase p (one s_barrier each):  group A  MEM(k) at p = 2k, MMA(k) at 2k + 1;
    // group B  MEM(k) at 2k + 1, MMA(k) at 2k + 2.  Invariant: before the barrier into phase 2k every wave has waited for its own
    // pieces of stage k (in-order vmcnt: at most the pieces of stage k + 1 stay in flight).  Stage k's buffer is last read in
    // phase 2k + 1 and refilled (as stage k + 3) from phase 2k + 2 on.  Bare loop 1352 vs 1058 TFLOP/s for two co-resident
    // 128 x 160 workgroups (tools/ubench/gemm_loop.hip).
    const int grp = w >> 2;
    int issued = total_steps < STAGES - 1 ? total_steps : STAGES - 1;     // stages this wave has issued so far
    bf16x8 af[KK][MI];
    bf16x8 bfr[KK][NT];
    // own pieces of stage k have landed.  With a 3-deep ring at most ONE younger stage is in flight at that point, so the count is 0 or
    // this wave's pieces per stage (AI + WI, or one fewer on the waves whose last W piece has no row group): immediates behind a
    // two-way test — wait_vm()'s general switch is a chain of ~13 compare-and-branch pairs, twice per K step, at the END of group A's MMA phase
    auto landed = [&](int k) {
      if (k >= nsteps) return;
      if constexpr (SEAM != 0) {
        // prefetched prologue: the pieces went out as [W0 W1 | A0 A1] — stage 0 is complete once only A1's pieces are in flight
        if (cx.seam_in && cx.prefetch && k == 0 && issued == STAGES - 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI) : "memory"); return; }
      }
      if (issued - 1 - k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (w_last_ok) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + WI) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + WI - 1) : "memory");
    };
    auto mem = [&](int k) {
      const int sb = k % STAGES;
      const bf16_t* As = smem + sb * BUF_ELEMS;
      const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * (MI * 16) + i * 16 + frow;
          af[kk][i] = *reinterpret_cast<const bf16x8*>(As + row * KT + (((kk * 4 + fkc) ^ SWZ(row)) * 8));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wn * (BN / 2) + j * 16 + frow;
          bfr[kk][j] = *reinterpret_cast<const bf16x8*>(Bs + row * KT + (((kk * 4 + fkc) ^ SWZ(row)) * 8));
        }
      }
      // (the pointers of this stage were prepared during the previous MMA phase: only the DMA instructions here)
      // (round 4: only the A rows' pieces here and the W rows' at the tail of the wave's MMA phase, behind its 40 MFMAs — to shorten the
      // memory phase, the longer of the two by the cycle model: loop 477.7 -> 492.6 ms, A/B of two builds.  Not kept.)
      // (the pieces BEFORE the fragment reads instead of behind them: loop 448.6 -> 449.7 ms.  Not kept.)
      if (k + STAGES - 1 < nsteps) { issue_dma((k + STAGES - 1) % STAGES); ++issued; }
    };
    auto mma = [&](int k) {
      // the K walk's pointer advance is straight-line code: it goes out BETWEEN the MFMAs (one scalar / vector instruction per MFMA: the
      // matrix pipe paces the MFMAs at 16 cycles each, an in-order wave that issues them back to back and the bookkeeping behind them
      // pays for the bookkeeping in full at the tail of the phase).  (Past the last issued stage the advanced pointers are never used.)
      issue_post();
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < KK * MI * NT; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 1, 0);
      }
      // ... the next segment's set-up (branches) behind them
      if (k + STAGES < nsteps) issue_prepare();
    };
    if (STAGES - 1 < nsteps) issue_prepare();
    landed(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 0) {
      for (int k = 0; k < nsteps; ++k) {
        mem(k);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(k);
        landed(k + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_s_barrier();
    } else {
      __builtin_amdgcn_s_barrier();
      for (int k = 0; k < nsteps; ++k) {
        mem(k);
        landed(k + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(k);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  } else
  for (int it = 0; it < nsteps; ++it, ++flat) {
    // wait for stage `flat` only: the up to STAGES - 2 younger stages (lw instructions each) stay in flight
    if constexpr (STAGES == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (STAGES == 3) {
      // at most one younger stage in flight: 0 or this wave's pieces per stage, as immediates (see the ping-pong loop's landed())
      if (total_steps - 1 - flat <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (w_last_ok) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + WI) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + WI - 1) : "memory");
    } else {
      // deeper rings (STREAM64: 6): in the steady state STAGES - 2 younger stages are in flight — this wave's pieces per stage times that, as an
      // immediate behind a two-way test; the last STAGES - 2 steps of the walk just wait for everything (wait_vm()'s run-time switch compiles to a
      // chain of ~15 compare-and-branch pairs per K step: an EMPTY step of this loop cost 0.34 us with it, profiles/r05_opt_stream64.md)
      if (total_steps - 1 - flat < STAGES - 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (w_last_ok) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (AI + WI)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (AI + WI - 1)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bf16_t* As = smem + buf * BUF_ELEMS;
    const bf16_t* Bs = As + A_ELEMS;
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    // Fragment reads are software-pipelined by hand: the reads of k step kk+1 are ISSUED before the MFMAs of k step kk (two
    // fragment sets live), and sched_barriers pin that order.  Left alone, the compiler recycles one 4-register A fragment
    // for the second k step (read -> s_waitcnt lgkmcnt(0) -> 5 MFMAs, three times over): every one of those waits exposes a
    // full LDS round trip.
    bf16x8 af[KK][MI];
    bf16x8 bfr[KK][NT];
    auto frag_load = [&](int kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * (MI * 16) + i * 16 + frow;
        const int slot = (kk * 4 + fkc) ^ SWZ(row);
        af[kk][i] = *reinterpret_cast<const bf16x8*>(As + row * KT + slot * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * (BN / 2) + j * 16 + frow;
        const int slot = (kk * 4 + fkc) ^ SWZ(row);
        bfr[kk][j] = *reinterpret_cast<const bf16x8*>(Bs + row * KT + slot * 8);
      }
    };
    frag_load(0);
    // stage the next tile while the first fragments are in flight from LDS (its ~30 instructions cover that latency)
    // (round 4: the stage's 8-9 LDS-DMA pieces issued one at a time BETWEEN this step's MFMAs, fenced, instead of in a bunch here — the
    // microarchitecture notes price a piece at ~60 cycles among bare MFMAs against 100-185 behind fragment reads: loop 447.5 -> 469.1 ms,
    // A/B of two builds in one session.  Not kept.)
    // (DEFER: the first stage of the NEXT N tile is issued from this tile's epilogue instead, after the epilogue's own parameter
    // loads — vmcnt returns in order, so loads issued behind a 1-KiB-per-lane-group DMA stage wait for the whole stage to land, and
    // the compiler fences the epilogue's first load with vmcnt(0) anyway: 1-1.5 us per tile with nothing to do)
    if (flat + STAGES - 1 < total_steps) {
      if (DEFER && it == nsteps - 1) deferred_buf = buf_issue;
      else issue(buf_issue);
    }
    buf_issue = (buf_issue + 1 == STAGES) ? 0 : buf_issue + 1;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) {
        __builtin_amdgcn_sched_barrier(0);
        frag_load(kk + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue.  acc[i][j][r]: m = m0 + wm*(MI*16) + i*16 + (lane&15); column: see tile_col() — sub-tiles (2g, 2g+1) of a lane
  // hold the 8 consecutive columns nbase + g*32 + fkc*8 .. +7 (registers r of sub-tile 2g, then of 2g+1); an unpaired last
  // sub-tile (NT odd) holds nbase + j*16 + fkc*4 .. +3
  const int mrow = m0 + wm * (MI * 16) + frow;
  const int nbase = n0 + wn * (BN / 2);
  constexpr int NG = (NT + 1) / 2;      // column groups per lane
  // UPS4: output row of source row m (divisions once per row, here only)
  auto out_row = [&](int m) -> int {
    if constexpr (CONV == 3) {
      const int shw = p.IH * p.IW;
      const int sb = m / shw;
      const int sr = m - sb * shw;
      const int sy = sr / p.IW, sx = sr - sy * p.IW;
      return (sb * p.OH + 2 * sy + ups_py) * p.OW + 2 * sx + ups_px;
    } else {
      return m;
    }
  };
  if constexpr (EPI == 2 || EPI == 7) {
    float* ws = p.ws + (size_t)z * (CONV == 3 ? 4 : 1) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (mrow + i * 16 >= p.M) continue;
      const int m = out_row(mrow + i * 16);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bool pair = (2 * g + 1 < NT);
        const int j1 = pair ? 2 * g + 1 : 2 * g;
        const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
        if (n >= p.N) continue;
        float* dst = ws + (size_t)m * p.N + n;
        if constexpr (EPI == 7) {      // COOP: write-through, nothing left dirty in this XCD's L2 for a release to flush
          st_wt_f32x4(dst, acc[i][2 * g]);
          if (pair && n + 4 < p.N) st_wt_f32x4(dst + 4, acc[i][j1]);
        } else {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[i][2 * g][0], acc[i][2 * g][1], acc[i][2 * g][2], acc[i][2 * g][3]);
          if (pair && n + 4 < p.N)
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[i][j1][0], acc[i][j1][1], acc[i][j1][2], acc[i][j1][3]);
        }
      }
    }
    if constexpr (EPI == 7) {
      // ---- COOP split-K finish (GemmArgs::coop_ctr).  Group = the workgroups whose tiles cover whole samples of one N tile, over all splits:
      // M tiles per group mtg = rows_per_batch / BM (level 2: 2) or 1 with spg = BM / rows_per_batch samples in the tile (level 3: 2); G = mtg x splits
      // workgroups arrive, then workgroup widx of the group finishes units widx, widx + G, ... of the group's spg x (BN / 40) (sample, 40-column) units.
      static_assert(EPI != 7 || (PP && BN % 40 == 0), "COOP split-K finish: the ping-pong tiles");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its own write-through stores ...
      __syncthreads();                                       // ... before the one arrival of the workgroup
      const int rpb = p.rows_per_batch;
      const int mtg = rpb > BM ? rpb / BM : 1, spg = rpb < BM ? BM / rpb : 1;
      const int G = mtg * cx.gdy;
      const int tg = tm / mtg;
      unsigned char* cs = smem_raw + 16384;                  // (the ring is dead; [0, 5 KiB) is the row-major epilogue's `red`, unused here)
      float2 (*part)[10] = reinterpret_cast<float2 (*)[10]>(cs);
      float2* quad = reinterpret_cast<float2*>(cs + 2048);
      float2* mr = reinterpret_cast<float2*>(cs + 2304);
      unsigned* okf = reinterpret_cast<unsigned*>(cs + 2560);
      if (tid == 0) {
        const bool ok = coop_arrive_wait(p.coop_ctr + (size_t)tg * d.tiles_n + n0 / BN, (unsigned)G);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire per workgroup: drops this CU's stale L1 lines; the slices are read with plain loads
        *okf = ok ? 1u : 0u;
      }
      __syncthreads();
      const bool poison = *okf == 0u;
      constexpr int UPT = BN / 40;                           // units per sample in this N tile
      const int widx = z * mtg + (tm - tg * mtg);
      for (int u = widx; u < spg * UPT; u += G) {            // (workgroup-uniform trip count)
        const int b = tg * spg + u / UPT, col0 = n0 + (u % UPT) * 40;
        if (rpb == 256) splitk_finish_unit<16, 40>(p, b, col0, tid, part, quad, mr, poison);
        else splitk_finish_unit<4, 40>(p, b, col0, tid, part, quad, mr, poison);
        __syncthreads();                                     // (the scratch is reused by the next unit)
      }
      if constexpr (SEAM != 0) {
        if (cx.seam_out) {      // this phase's units are finished: publish them (plain stores -> one agent-scope release) and arrive
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(cx.seam_out + tg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  } else if constexpr (EPI == 1) {
    // weight rows are interleaved in 16-row blocks (even block = value rows, odd block = gate rows of 16 outputs); tile_col() deals
    // a quad of sub-tiles (value, gate, value, gate) so that the lane's 2 x 4 products are 8 consecutive output columns
    const float* rr = ln_rr; const float* rm = ln_rm;
#pragma unroll
    for (int qd = 0; qd < NT / 4; ++qd) {
      const int ol = fkc * 8;                                          // output column within the quad's 32 (first of 8)
      const int pv = nbase + qd * 64 + (ol >> 4) * 32 + (ol & 15);     // physical column of the first value row (4 consecutive; +4: second pair)
      if (nbase + qd * 64 >= p.N) continue;            // (N % 128 == 0: a quad's 64 physical columns are all inside or all outside)
      const int no = (nbase + qd * 64) / 2 + ol;                       // logical output column (8 consecutive)
      float4 bv[2], bg[2], sv[2], sg[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bv[h] = p.bias ? *reinterpret_cast<const float4*>(p.bias + pv + 4 * h) : make_float4(0, 0, 0, 0);
        bg[h] = p.bias ? *reinterpret_cast<const float4*>(p.bias + pv + 16 + 4 * h) : make_float4(0, 0, 0, 0);
        sv[h] = p.ln_stats ? *reinterpret_cast<const float4*>(p.ln_colsum + pv + 4 * h) : make_float4(0, 0, 0, 0);
        sg[h] = p.ln_stats ? *reinterpret_cast<const float4*>(p.ln_colsum + pv + 16 + 4 * h) : make_float4(0, 0, 0, 0);
      }
      // (no branch between these loads and their uses, one explicit wait the compiler's counter tracking sees, stores predicated:
      // see the row-major epilogue)
      __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
      if (DEFER && qd == NT / 4 - 1 && deferred_buf >= 0) { issue(deferred_buf); deferred_buf = -1; }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow + i * 16;
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 av = acc[i][qd * 4 + 2 * h], ag = acc[i][qd * 4 + 2 * h + 1];
          o[4 * h + 0] = (av[0] * rr[i] - rm[i] * sv[h].x + bv[h].x) * gelu_erf(ag[0] * rr[i] - rm[i] * sg[h].x + bg[h].x);
          o[4 * h + 1] = (av[1] * rr[i] - rm[i] * sv[h].y + bv[h].y) * gelu_erf(ag[1] * rr[i] - rm[i] * sg[h].y + bg[h].y);
          o[4 * h + 2] = (av[2] * rr[i] - rm[i] * sv[h].z + bv[h].z) * gelu_erf(ag[2] * rr[i] - rm[i] * sg[h].z + bg[h].z);
          o[4 * h + 3] = (av[3] * rr[i] - rm[i] * sv[h].w + bv[h].w) * gelu_erf(ag[3] * rr[i] - rm[i] * sg[h].w + bg[h].w);
        }
        uint4 ov; ov.x = pack_bf2(o[0], o[1]); ov.y = pack_bf2(o[2], o[3]); ov.z = pack_bf2(o[4], o[5]); ov.w = pack_bf2(o[6], o[7]);
        if (m < p.M) *reinterpret_cast<uint4*>((bf16_t*)p.C + (size_t)m * p.ldc + no) = ov;
      }
    }
  } else if constexpr (EPI == 5) {
    // scores of one (sample, head) per wave half: the wave's 80 columns ARE the head's keys (77 + 3 pads whose bias is -1e30), so the
    // softmax of a row is a reduction over the lane's 20 values and the 4 lanes (fkc) that share the row.  exp2 domain (scale * log2(e)
    // is folded into the per-sample weights).  P is stored row-major: the A operand of the second GEMM.
    static_assert(EPI != 5 || BN == 160, "softmax epilogue: one 80-column head per wave half");
    const size_t vb = p.wb_rows ? (size_t)(m0 / p.wb_rows) * (size_t)p.vb_stride : (size_t)0;
    const float* bias = p.bias + vb;
    const float* csum = p.ln_colsum + vb;
    float4 bzs[NG][2], css[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
      bzs[g][0] = *reinterpret_cast<const float4*>(bias + n);
      css[g][0] = *reinterpret_cast<const float4*>(csum + n);
      bzs[g][1] = pair ? *reinterpret_cast<const float4*>(bias + n + 4) : make_float4(0, 0, 0, 0);
      css[g][1] = pair ? *reinterpret_cast<const float4*>(csum + n + 4) : make_float4(0, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): see the row-major epilogue
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow + i * 16;
      const float rr = ln_rr[i], rm = ln_rm[i];
      float v[NG][8];
      float mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bool pair = (2 * g + 1 < NT);
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][pair ? 2 * g + 1 : 2 * g];
        v[g][0] = a0[0] * rr - rm * css[g][0].x + bzs[g][0].x; v[g][1] = a0[1] * rr - rm * css[g][0].y + bzs[g][0].y;
        v[g][2] = a0[2] * rr - rm * css[g][0].z + bzs[g][0].z; v[g][3] = a0[3] * rr - rm * css[g][0].w + bzs[g][0].w;
        v[g][4] = pair ? a1[0] * rr - rm * css[g][1].x + bzs[g][1].x : -INFINITY; v[g][5] = pair ? a1[1] * rr - rm * css[g][1].y + bzs[g][1].y : -INFINITY;
        v[g][6] = pair ? a1[2] * rr - rm * css[g][1].z + bzs[g][1].z : -INFINITY; v[g][7] = pair ? a1[3] * rr - rm * css[g][1].w + bzs[g][1].w : -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[g][e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[g][e] = __builtin_amdgcn_exp2f(v[g][e] - mx); sum += v[g][e]; }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      if (m < p.M) {
        bf16_t* crow = (bf16_t*)p.C + (size_t)m * p.ldc;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool pair = (2 * g + 1 < NT);
          const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
          uint4 o; o.x = pack_bf2(v[g][0] * inv, v[g][1] * inv); o.y = pack_bf2(v[g][2] * inv, v[g][3] * inv);
          o.z = pack_bf2(v[g][4] * inv, v[g][5] * inv); o.w = pack_bf2(v[g][6] * inv, v[g][7] * inv);
          if (pair) *reinterpret_cast<uint4*>(crow + n) = o;
          else *reinterpret_cast<uint2*>(crow + n) = make_uint2(o.x, o.y);
        }
      }
    }
  } else if constexpr (EPI == 3) {
    // head-major scatter; all divisions hoisted: per-row (b, t) once, per-column-group (segment, head, dd) once (dp % 8 == 0: the 8
    // columns of a group lie in one head of one segment)
    int rq[MI], rk[MI], rv[MI]; bool mok[MI];
    float rr[MI], rm[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow + i * 16;
      mok[i] = m < p.M;
      rr[i] = p.ln_stats ? ln_rr[i] : p.alpha; rm[i] = ln_rm[i];
      const int b = m / p.ntok, t = m - b * p.ntok;
      rq[i] = (b * p.heads * p.ntok_pad_q + t) * p.dp;
      rk[i] = (b * p.heads * p.ntok_pad_kv + t + p.kv_tok_offset) * p.dp;
      rv[i] = b * p.heads * p.dpv * p.ntok_pad_kv + t + p.kv_tok_offset;
    }
    const int hd = p.heads * p.dp;
    // epilogue parameters of every column group first (then the deferred stage of the next N tile: see DEFER)
    float4 bzs[NG][2], css[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
      const bool hi = pair && (n + 4 < p.N);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = (n < p.N) && ((e == 0) || hi);
        bzs[g][e] = (p.bias && ok) ? *reinterpret_cast<const float4*>(p.bias + n + 4 * e) : make_float4(0, 0, 0, 0);
        css[g][e] = (p.ln_stats && ok) ? *reinterpret_cast<const float4*>(p.ln_colsum + n + 4 * e) : make_float4(0, 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): see the row-major epilogue
    if (DEFER && deferred_buf >= 0) { issue(deferred_buf); deferred_buf = -1; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool pair = (2 * g + 1 < NT);
      const int j1 = pair ? 2 * g + 1 : 2 * g;
      const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
      if (n >= p.N) continue;
      const bool hi = pair && (n + 4 < p.N);
      const int segl = n / hd;
      const int within = n - segl * hd;
      const int h = within / p.dp, dd = within - h * p.dp;
      const int seg = p.seg_base + segl;
      const float4* bz = bzs[g]; const float4* cs = css[g];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (!mok[i]) continue;
        float v[8];
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
        v[0] = a0[0] * rr[i] - rm[i] * cs[0].x + bz[0].x; v[1] = a0[1] * rr[i] - rm[i] * cs[0].y + bz[0].y;
        v[2] = a0[2] * rr[i] - rm[i] * cs[0].z + bz[0].z; v[3] = a0[3] * rr[i] - rm[i] * cs[0].w + bz[0].w;
        v[4] = a1[0] * rr[i] - rm[i] * cs[1].x + bz[1].x; v[5] = a1[1] * rr[i] - rm[i] * cs[1].y + bz[1].y;
        v[6] = a1[2] * rr[i] - rm[i] * cs[1].z + bz[1].z; v[7] = a1[3] * rr[i] - rm[i] * cs[1].w + bz[1].w;
        if (seg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.qscale;   // softmax scale * log2(e) folded into Q
        }
        if (seg <= 1) {
          bf16_t* dst = (seg == 0) ? p.Cq + (size_t)rq[i] + (size_t)h * p.ntok_pad_q * p.dp + dd
                                   : p.Ck + (size_t)rk[i] + (size_t)h * p.ntok_pad_kv * p.dp + dd;
          if (hi) {
            uint4 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
            *reinterpret_cast<uint4*>(dst) = o;
          } else {
            uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = o;
          }
        } else {
          bf16_t* base = p.Cvt + (size_t)rv[i] + (size_t)(h * p.dpv + dd) * p.ntok_pad_kv;
          const int ne = hi ? 8 : 4;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < ne) base[(size_t)e * p.ntok_pad_kv] = f2bf(v[e]);
          // spare row dp of Vt := 1.0: the attention kernel reads the softmax row sum off the PV MFMAs
          if (dd + ne == p.dp && p.dpv > p.dp) base[(size_t)ne * p.ntok_pad_kv] = (bf16_t)0x3F80;
        }
      }
    }
  } else {
    // row-major epilogue; per-row offsets (incl. the batch index of the row vector) hoisted out of the column loop
    // (32-bit element offsets: gemm_launch() checks M * max(ldc, ldr) < 2^31)
    // The epilogue is written WITHOUT branches between its loads and their uses: all of a column group's loads (bias, row vector,
    // residual, for every row) are issued back to back from addresses clamped into the tensors, then everything is computed, and only
    // the stores are predicated.  With a `continue` per row the compiler has to re-fence every row's block with s_waitcnt vmcnt(0) —
    // and on gfx9 that counter also holds the stores, so every row waited for the previous row's store round trip: 12 serialised
    // round trips made the 256 x 160 conv tile's epilogue 4.1 us (tools/conv_ksweep.py with stamps), 3.4 us per GEGLU tile.
    int crow[MI], rrow[MI]; bool mok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow + i * 16;
      mok[i] = m < p.M;
      const int mc = mok[i] ? m : p.M - 1;
      crow[i] = out_row(mc) * p.ldc;
      rrow[i] = mc * p.ldr;
    }
    // fused GroupNorm statistics of the OUTPUT tensor (consumed by the next GroupNorm: saves its whole stats pass).
    // The 64 rows of a wave belong to one slab of one sample (rows_per_batch % 64 == 0).  Fixed-order reduction: 4 rows in the
    // lane, 16 row lanes by xor-shuffles, one LDS cell per (row half, column) written once (the tile ring is dead by now),
    // then one thread per (row half, bin, moment) adds the bin's gn_cg columns and stores the partial: no atomics anywhere.
    float2* red = reinterpret_cast<float2*>(smem);   // [NWV / 2 wave rows][BN] {sum, sum of squares}: one 8-byte LDS write per column at
                                                     // base + immediate offset (separate moment planes cost one address register per column)
    float2* red_lane = red + wm * BN + wn * (BN / 2);
    if (p.gn_stats) __syncthreads();                 // every wave is done reading the ring
    float rws[4] = {0.f, 0.f, 0.f, 0.f}, rwq[4] = {0.f, 0.f, 0.f, 0.f};   // row sums for a following (folded) LayerNorm
    constexpr bool F32_PATHS = (CONV == 0 && EPI != 4);     // fp32 residual / fp32 output / activation: plain GEMMs on EPI 0 only
    // phase A: the loads.  Column vector cv = bias (+ the row vector, a per-sample bias — ResnetBlock2D's time embedding — which is the
    // same for all rows of the wave when they lie in one sample: the case in the engines); a wave that straddles samples, and the
    // fp32 residual (32 B per lane and row), take the slow path: loads inside the row loop.
    // The bf16-only kernels (convs, EPI 4) issue the loads of all NG column groups up front (at most 24 + 48 registers next to the
    // 80 accumulators) and wait once: one round trip per tile, and no load ever queues behind a store.  The generic EPI 0 kernel, which
    // also carries the fp32 / activation paths, goes group by group (loads, wait, compute, store: NG round trips).
    constexpr bool ALL_UPFRONT = !F32_PATHS;
    constexpr int NSLOT = ALL_UPFRONT ? NG : 1;
    const int wrow0 = m0 + wm * (MI * 16);
    const int wrow1 = (wrow0 + MI * 16 - 1 < p.M) ? wrow0 + MI * 16 - 1 : p.M - 1;
    const bool rv_pre = p.rowvec && wrow0 < p.M && (wrow0 / p.rows_per_batch == wrow1 / p.rows_per_batch);   // wave-uniform
    const bool rv_slow = p.rowvec && !rv_pre;
    const bool rf_slow = F32_PATHS && p.resid && p.resid_f32;
    const bool rb_pre = p.resid && !rf_slow;
    const float* colvec = rv_pre ? p.rowvec + (wrow0 / p.rows_per_batch) * p.rowvec_bstride : p.bias;    // first (or only) column vector
    const float* colvec2 = rv_pre ? p.bias : nullptr;                                                      // bias on top of a row vector
    float4 cv[NSLOT][2];
    uint2 rb[NSLOT][MI][2];
    auto load_group = [&](int g, int sl) {
      const bool pair = (2 * g + 1 < NT);
      const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
      const bool hi = pair && (n + 4 < p.N);
      const int nl = (n < p.N) ? n : 0;      // columns the loads use: always inside the row (lanes beyond N are never stored)
      const int nh = hi ? n + 4 : nl;
      cv[sl][0] = cv[sl][1] = make_float4(0, 0, 0, 0);
      if (colvec) {
        cv[sl][0] = *reinterpret_cast<const float4*>(colvec + nl);
        cv[sl][1] = *reinterpret_cast<const float4*>(colvec + nh);
      }
      if (colvec2) {
        const float4 t0 = *reinterpret_cast<const float4*>(colvec2 + nl);
        const float4 t1 = *reinterpret_cast<const float4*>(colvec2 + nh);
        cv[sl][0].x += t0.x; cv[sl][0].y += t0.y; cv[sl][0].z += t0.z; cv[sl][0].w += t0.w;
        cv[sl][1].x += t1.x; cv[sl][1].y += t1.y; cv[sl][1].z += t1.z; cv[sl][1].w += t1.w;
      }
    };
    auto load_group_resid = [&](int g, int sl) {
      const bool pair = (2 * g + 1 < NT);
      const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
      const bool hi = pair && (n + 4 < p.N);
      const int nl = (n < p.N) ? n : 0;
      const int nh = hi ? n + 4 : nl;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        rb[sl][i][0] = *reinterpret_cast<const uint2*>((const bf16_t*)p.resid + rrow[i] + nl);
        rb[sl][i][1] = *reinterpret_cast<const uint2*>((const bf16_t*)p.resid + rrow[i] + nh);
      }
    };
    if constexpr (ALL_UPFRONT) {
#pragma unroll
      for (int g = 0; g < NG; ++g) load_group(g, g);
      if (rb_pre) {
#pragma unroll
        for (int g = 0; g < NG; ++g) load_group_resid(g, g);
      }
      // phase B: an explicit s_waitcnt, which the compiler's own counter tracking sees — nothing is pending behind it, so the
      // predicated stores below (branches, i.e. control-flow joins) cannot draw conservative vmcnt(0) fences
      __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    }
    // phase C: compute and store
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int sl = ALL_UPFRONT ? g : 0;
      const int rs = sl;
      if constexpr (!ALL_UPFRONT) {
        load_group(g, 0);
        if (rb_pre) load_group_resid(g, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      }
      const bool pair = (2 * g + 1 < NT);
      const int j1 = pair ? 2 * g + 1 : 2 * g;
      const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;    // first column of the group within the wave's half
      const int n = nbase + cl;
      const bool nok = n < p.N;
      const bool hi = pair && (n + 4 < p.N);
      float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
        float v[8] = {a0[0] * p.alpha + cv[sl][0].x, a0[1] * p.alpha + cv[sl][0].y, a0[2] * p.alpha + cv[sl][0].z, a0[3] * p.alpha + cv[sl][0].w,
                      a1[0] * p.alpha + cv[sl][1].x, a1[1] * p.alpha + cv[sl][1].y, a1[2] * p.alpha + cv[sl][1].z, a1[3] * p.alpha + cv[sl][1].w};
        // (slow paths: guarded by wave-uniform flags ONLY and loading from clamped columns, so that they compile to scalar branches —
        // under a lane predicate a never-taken load still executes its s_waitcnt vmcnt(0), which queues behind the previous row's store)
        if (rv_slow) {
          const int vr = (min(mrow + i * 16, p.M - 1) / p.rows_per_batch) * p.rowvec_bstride;
          const int nl = nok ? n : 0, nh = hi ? n + 4 : nl;
          const float4 b0 = *reinterpret_cast<const float4*>(p.rowvec + vr + nl);
          const float4 b1 = *reinterpret_cast<const float4*>(p.rowvec + vr + nh);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (rb_pre) {
          v[0] += bf2f((bf16_t)(rb[rs][i][0].x & 0xffff)); v[1] += bf2f((bf16_t)(rb[rs][i][0].x >> 16));
          v[2] += bf2f((bf16_t)(rb[rs][i][0].y & 0xffff)); v[3] += bf2f((bf16_t)(rb[rs][i][0].y >> 16));
          v[4] += bf2f((bf16_t)(rb[rs][i][1].x & 0xffff)); v[5] += bf2f((bf16_t)(rb[rs][i][1].x >> 16));
          v[6] += bf2f((bf16_t)(rb[rs][i][1].y & 0xffff)); v[7] += bf2f((bf16_t)(rb[rs][i][1].y >> 16));
        }
        if (F32_PATHS && rf_slow) {
          const int nl = nok ? n : 0, nh = hi ? n + 4 : nl;
          const float4 r0 = *reinterpret_cast<const float4*>((const float*)p.resid + rrow[i] + nl);
          const float4 r1 = *reinterpret_cast<const float4*>((const float*)p.resid + rrow[i] + nh);
          v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
          v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        // (convolutions: no activation, bf16 output, no LayerNorm row sums — gemm_launch() checks — so their kernels carry none of it)
        if (F32_PATHS && p.act != ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
        }
        const bool st = mok[i] && nok;
        if (F32_PATHS && p.out_mode == OUT_F32) {
          if (st) *reinterpret_cast<float4*>((float*)p.C + crow[i] + n) = make_float4(v[0], v[1], v[2], v[3]);
          if (st && hi) *reinterpret_cast<float4*>((float*)p.C + crow[i] + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          uint4 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
          const bool stc = st && EPI != 6;     // (COOP: the raw tensor — where somebody else reads it — is stored AFTER the arrival, see below)
          if (stc && hi) *reinterpret_cast<uint4*>((bf16_t*)p.C + crow[i] + n) = o;
          else if (stc) *reinterpret_cast<uint2*>((bf16_t*)p.C + crow[i] + n) = make_uint2(o.x, o.y);
          if constexpr (EPI == 6) {      // the finished values stay in the accumulators until the group's statistics are complete
            acc[i][2 * g] = (f32x4){v[0], v[1], v[2], v[3]};
            if (pair) acc[i][j1] = (f32x4){v[4], v[5], v[6], v[7]};
          }
          if (CONV == 0 && p.row_stats) {   // statistics of what the consumer will read: the rounded values (columns beyond N: none)
            const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
            const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
            const float r4 = __uint_as_float(o.z << 16), r5 = __uint_as_float(o.z & 0xffff0000u);
            const float r6 = __uint_as_float(o.w << 16), r7 = __uint_as_float(o.w & 0xffff0000u);
            const float slo = (r0 + r1) + (r2 + r3), qlo = (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
            const float shi = (r4 + r5) + (r6 + r7), qhi = (r4 * r4 + r5 * r5) + (r6 * r6 + r7 * r7);
            rws[i] += nok ? slo : 0.f;
            rwq[i] += nok ? qlo : 0.f;
            rws[i] += hi ? shi : 0.f;      // (two separate additions, low half first: the order the sums have always been formed in)
            rwq[i] += hi ? qhi : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {       // (rows beyond M hold the clamped row's values: they must not reach the statistics)
          const float ve = mok[i] ? v[e] : 0.f;
          gs[e] += ve; gq[e] += ve * ve;
        }
      }
      if (p.gn_stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (e >= 4 && !pair) break;
          const float a = row16_sum(gs[e]), q = row16_sum(gq[e]);     // over the 16 row lanes (DPP: no LDS traffic)
          if (frow == 0) red_lane[cl + e] = make_float2(a, q);
        }
      }
    }
    if (CONV == 0 && p.row_stats) {
      // a row's BN/2 columns of this wave sit in the 4 lanes {frow, frow+16, frow+32, frow+48}; plane = (N tile, wave column)
      float* plane = p.row_stats + (size_t)((n0 / BN) * 2 + wn) * p.M * 2;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        rws[i] += __shfl_xor(rws[i], 16, 64); rwq[i] += __shfl_xor(rwq[i], 16, 64);
        rws[i] += __shfl_xor(rws[i], 32, 64); rwq[i] += __shfl_xor(rwq[i], 32, 64);
        if (fkc == 0 && mok[i]) *reinterpret_cast<float2*>(plane + (size_t)(mrow + i * 16) * 2) = make_float2(rws[i], rwq[i]);
      }
    }
    if (p.gn_stats) {
      __syncthreads();
      constexpr int HALVES = BM / 64;
      const int bins_tile = BN / p.gn_cg;            // gemm_fused_gn_ok(): BN % gn_cg == 0
      if (tid < 2 * HALVES * bins_tile) {
        const int which = tid & 1, half = (tid >> 1) % HALVES, lb = (tid >> 1) / HALVES;
        const int mfirst = m0 + half * 64;
        const int bin = n0 / p.gn_cg + lb;
        if (mfirst < p.M && bin < p.gn_groups) {
          // a 64-row slab holds 64 / (MI * 16) wave rows (1, or 2 on the MI = 2 tiles): added in wave-row order, then column order
          constexpr int GPS = 64 / (MI * 16);
          float a = 0.f;
#pragma unroll
          for (int g = 0; g < GPS; ++g) {
            const float* src = reinterpret_cast<const float*>(red + (half * GPS + g) * BN + lb * p.gn_cg) + which;
            for (int c = 0; c < p.gn_cg; ++c) a += src[2 * c];
          }
          // (UPS4: rows_per_batch = SOURCE pixels per sample; a slab = 64 source rows of one class — any fixed partition of a
          // sample's output pixels into 64-row slabs serves the consumer, which only adds the slabs up in index order)
          const int b = mfirst / p.rows_per_batch;
          const int slab = (mfirst - b * p.rows_per_batch) / GN_SLAB_ROWS + (CONV == 3 ? cls * (p.rows_per_batch / GN_SLAB_ROWS) : 0);
          const int nslab = (CONV == 3 ? 4 : 1) * (p.rows_per_batch / GN_SLAB_ROWS);
          float* dst = p.gn_stats + (((size_t)b * nslab + slab) * p.gn_groups + bin) * 2 + which;
          if constexpr (EPI == 6) st_wt_f32(dst, a);       // COOP: the other workgroups of the group read it inside this launch
          else *dst = a;
        }
      }
    }
    if constexpr (EPI == 6) {
      // ---- COOP GroupNorm finish (GemmArgs::coop_ctr, splitk == 1).  The rows_per_batch / BM workgroups that hold the M tiles of one (sample, N tile)
      // have each published their per-slab partials above; once all have arrived every one of them totals the sample's partials for its BN columns
      // EXACTLY as groupnorm_apply_kernel does (16-partial segments on a fixed tree, segments in order, bins in order: bit-identical statistics),
      // folds gamma / beta into per-column scale | shift and normalises the ROUNDED values it still holds in its accumulators — what the
      // stand-alone pass would have read back from HBM.
      static_assert(EPI != 6 || (PP && CONV == 1), "COOP GroupNorm finish: the ping-pong convolution tiles");
      // Order (round 6, second pass): nothing but the few partial stores is in flight at the drain — the raw tensor's 40-80 KiB of stores per
      // workgroup go out AFTER the arrival and land while the group fills up, and gamma / beta are in registers before the wait.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float* cs = reinterpret_cast<float*>(smem_raw + 16384);    // scratch behind `red`: seg [32 bins][2][4] | tot [32][2] | scale [BN] | shift [BN] | flag
      float* c_seg = cs; float* c_tot = cs + 256; float* c_sc = cs + 320; float* c_sh = cs + 320 + BN; unsigned* okf = reinterpret_cast<unsigned*>(cs + 320 + 2 * BN);
      const int bsmp = m0 / p.rows_per_batch;
      unsigned* gctr = p.coop_ctr + (size_t)bsmp * d.tiles_n + n0 / BN;
      if (tid == 0) coop_arrive(gctr);
      const bool colt = tid < BN && n0 + tid < p.N;
      const float gam_c = colt ? p.fn_gamma[n0 + tid] : 0.f, bet_c = colt ? p.fn_beta[n0 + tid] : 0.f;
      if (p.C) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool pair = (2 * g + 1 < NT);
          const int j1 = pair ? 2 * g + 1 : 2 * g;
          const int n = nbase + (pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4);
          const bool nok = n < p.N;
          const bool hi = pair && (n + 4 < p.N);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
            uint4 o; o.x = pack_bf2(a0[0], a0[1]); o.y = pack_bf2(a0[2], a0[3]); o.z = pack_bf2(a1[0], a1[1]); o.w = pack_bf2(a1[2], a1[3]);
            if (mok[i] && nok && hi) *reinterpret_cast<uint4*>((bf16_t*)p.C + crow[i] + n) = o;
            else if (mok[i] && nok) *reinterpret_cast<uint2*>((bf16_t*)p.C + crow[i] + n) = make_uint2(o.x, o.y);
          }
        }
      }
      if (tid == 0) {
        const bool ok = coop_wait(gctr, (unsigned)(p.rows_per_batch / BM));
        *okf = ok ? 1u : 0u;
      }
      __syncthreads();
      const bool ok = *okf != 0u;
      const int bins_tile = BN / p.gn_cg;                       // <= 32 (gemm_coop_ok)
      const int ns = p.rows_per_batch / GN_SLAB_ROWS;           // <= 64 partials per (sample, bin)
      if (tid < bins_tile * 8) {
        const int seg = tid & 3, which = (tid >> 2) & 1, lb = tid >> 3;
        const float* src = p.gn_stats + (((size_t)bsmp * ns + seg * 16) * p.gn_groups + n0 / p.gn_cg + lb) * 2 + which;
        const size_t step = (size_t)p.gn_groups * 2;
        float pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = (seg * 16 + i < ns) ? ld_wt_f32(src + (size_t)i * step) : 0.f;      // sc1 loads: the producers stored sc1
        c_seg[(lb * 2 + which) * 4 + seg] = (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
                                            (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
      }
      __syncthreads();
      if (tid < bins_tile * 2) { const float* g4 = c_seg + tid * 4; c_tot[tid] = (g4[0] + g4[1]) + (g4[2] + g4[3]); }
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        const int gl = tid / p.fn_cg, r1 = p.fn_cg / p.gn_cg;
        float a = 0.f, q = 0.f;
        for (int bin = gl * r1; bin < (gl + 1) * r1; ++bin) { a += c_tot[bin * 2]; q += c_tot[bin * 2 + 1]; }
        const float inv_n = 1.f / ((float)p.fn_cg * (float)p.rows_per_batch);
        const float sm = a * inv_n;         // E[x]
        const float sq = q * inv_n;         // E[x^2]
        const float var = fmaxf(sq - sm * sm, 0.f);
        const float rstd = rsqrtf(var + p.fn_eps);
        float sc = rstd * gam_c;
        if (!ok) sc = __builtin_nanf("");
        const float sh = bet_c - sm * sc;
        c_sc[tid] = sc; c_sh[tid] = sh;
        if (p.fn_ss && m0 == bsmp * p.rows_per_batch) {        // the scale | shift table, once per (sample, N tile): [B][2][N]
          p.fn_ss[((size_t)bsmp * 2 + 0) * p.N + n0 + tid] = sc;
          p.fn_ss[((size_t)bsmp * 2 + 1) * p.N + n0 + tid] = sh;
        }
      }
      __syncthreads();
      if (p.fn_Y) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool pair = (2 * g + 1 < NT);
          const int j1 = pair ? 2 * g + 1 : 2 * g;
          const int cl = pair ? g * 32 + fkc * 8 : (2 * g) * 16 + fkc * 4;
          const int n = nbase + cl;
          const bool nok = n < p.N;
          const bool hi = pair && (n + 4 < p.N);
          const float* scp = c_sc + wn * (BN / 2) + cl; const float* shp = c_sh + wn * (BN / 2) + cl;
          const float4 s0 = *reinterpret_cast<const float4*>(scp), h0 = *reinterpret_cast<const float4*>(shp);
          const float4 s1 = pair ? *reinterpret_cast<const float4*>(scp + 4) : s0, h1 = pair ? *reinterpret_cast<const float4*>(shp + 4) : h0;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const f32x4 a0 = acc[i][2 * g], a1 = acc[i][j1];
            float o[8];
            o[0] = fmaf(bf2f(f2bf(a0[0])), s0.x, h0.x); o[1] = fmaf(bf2f(f2bf(a0[1])), s0.y, h0.y);
            o[2] = fmaf(bf2f(f2bf(a0[2])), s0.z, h0.z); o[3] = fmaf(bf2f(f2bf(a0[3])), s0.w, h0.w);
            o[4] = fmaf(bf2f(f2bf(a1[0])), s1.x, h1.x); o[5] = fmaf(bf2f(f2bf(a1[1])), s1.y, h1.y);
            o[6] = fmaf(bf2f(f2bf(a1[2])), s1.z, h1.z); o[7] = fmaf(bf2f(f2bf(a1[3])), s1.w, h1.w);
            if (p.fn_silu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = silu_f(o[e]);
            }
            uint4 ov; ov.x = pack_bf2(o[0], o[1]); ov.y = pack_bf2(o[2], o[3]); ov.z = pack_bf2(o[4], o[5]); ov.w = pack_bf2(o[6], o[7]);
            bf16_t* yrow = p.fn_Y + (size_t)(mrow + i * 16) * p.N + n;
            if (mok[i] && nok && hi) *reinterpret_cast<uint4*>(yrow) = ov;
            else if (mok[i] && nok) *reinterpret_cast<uint2*>(yrow) = make_uint2(ov.x, ov.y);
          }
        }
      }
    }
  }
  }   // N tiles of this workgroup
#undef SWZ
}
template <int NWV, int BN, int CONV, int EPI, int STAGES, int KT, int MI>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? ((BN == 128 && STAGES == 2) ? 2 : 1) : (MI == 2 ? 3 : 2)) void gemm_kernel(const GemmDev d) {
  kernarg_warm<sizeof(GemmDev)>();
  GemmTileCtx cx;
  cx.bx = blockIdx.x; cx.by = blockIdx.y; cx.bz = blockIdx.z; cx.gdx = gridDim.x; cx.gdy = gridDim.y;
  gemm_tile<NWV, BN, CONV, EPI, STAGES, KT, MI, 0>(d, cx);
}

// split-K reduction + epilogue
// Block = 64 rows x W columns (W = 80 when the fused GroupNorm bins need it, else 64), 16 row lanes x W/4 column quads:
// thread = one float4 column quad of rows ty, ty+16, ty+32, ty+48.  The partials of every split are summed in split order.
// Fused statistics are fixed-order like the in-kernel epilogue's: per-row / per-column cells in LDS written once each, then
// one thread per output sum adds them in index order (row sums -> plane blockIdx.x of row_stats, bins -> this slab's
// GroupNorm partial).
template <int W, int R>
__global__ __launch_bounds__(4 * W) void gemm_splitk_reduce_kernel(const GemmArgs p) {
  kernarg_warm<sizeof(GemmArgs)>();
  constexpr int QW = W / 4;                     // column quads per row
  constexpr int ROWS = 16 * R;                  // rows per block: 64 for large outputs, 16 when blocks would be too few
  __shared__ float2 rowp[ROWS][QW];               // per (row, quad) {sum, sum of squares} of the bf16-rounded outputs
  __shared__ float2 colp[16][W];                // per (row lane, column) {sum, sum of squares} over the lane's 4 rows
  const int tx = threadIdx.x % QW, ty = threadIdx.x / QW;
  const int n = blockIdx.x * W + tx * 4;
  const int mbase = blockIdx.y * ROWS;
  float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  float rsum[R], rsq[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { rsum[r] = 0.f; rsq[r] = 0.f; }
  // row of the thread's r-th value
  auto row_of = [&](int r) -> int { return mbase + ty + 16 * r; };
  if (n < p.N) {
    // all partial loads of the thread's 4 rows are issued before any epilogue store (a store in between would fence the
    // next row's loads): 4 rows x 4 splits = 16 independent 16-B loads in flight per pass
    float4 s[R];
    const float* src[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int m = mbase + ty + 16 * r;
      if (m > p.M - 1) m = p.M - 1;
      s[r] = make_float4(0, 0, 0, 0);
      src[r] = p.ws + (size_t)m * p.N + n;
    }
    const size_t zstride = (size_t)p.M * p.N;
    constexpr int U = 16 / R;        // slices per pass: 16 independent 16-B loads in flight per thread either way
    int z = 0;
    for (; z + U <= p.splitk; z += U) {
      float4 v[R][U];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u) v[r][u] = *reinterpret_cast<const float4*>(src[r] + (size_t)(z + u) * zstride);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u) { s[r].x += v[r][u].x; s[r].y += v[r][u].y; s[r].z += v[r][u].z; s[r].w += v[r][u].w; }
    }
    for (; z < p.splitk; ++z) {
      float4 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const float4*>(src[r] + (size_t)z * zstride);
#pragma unroll
      for (int r = 0; r < R; ++r) { s[r].x += v[r].x; s[r].y += v[r].y; s[r].z += v[r].z; s[r].w += v[r].w; }
    }
    const float4 cs = p.ln_stats ? *reinterpret_cast<const float4*>(p.ln_colsum + n) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = row_of(r);
      if (m >= p.M) continue;
      if (p.ln_stats) {
        float rr, rm;
        ln_row_factors(p, m, rr, rm);
        s[r].x = s[r].x * rr - rm * cs.x; s[r].y = s[r].y * rr - rm * cs.y;
        s[r].z = s[r].z * rr - rm * cs.z; s[r].w = s[r].w * rr - rm * cs.w;
      }
      float fin[4];
      store4(p, m, n, s[r].x, s[r].y, s[r].z, s[r].w, fin);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gs[e] += fin[e]; gq[e] += fin[e] * fin[e]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float q = bf2f(f2bf(fin[e])); rsum[r] += q; rsq[r] += q * q; }
    }
  }
  if (p.row_stats) {
#pragma unroll
    for (int r = 0; r < R; ++r) rowp[ty + 16 * r][tx] = make_float2(rsum[r], rsq[r]);
    __syncthreads();
    if ((int)threadIdx.x < ROWS && mbase + (int)threadIdx.x < p.M) {
      int nq = (p.N - blockIdx.x * W + 3) / 4;
      if (nq > QW) nq = QW;
      float a = 0.f, q = 0.f;
      for (int t = 0; t < nq; ++t) { a += rowp[threadIdx.x][t].x; q += rowp[threadIdx.x][t].y; }
      *reinterpret_cast<float2*>(p.row_stats + ((size_t)blockIdx.x * p.M + mbase + threadIdx.x) * 2) = make_float2(a, q);
    }
  }
  if (p.gn_stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) colp[ty][tx * 4 + e] = make_float2(gs[e], gq[e]);
    __syncthreads();
    // two levels, each in index order: column totals over the 16 row lanes (one thread per column), then the bin's columns — a chain
    // of 16 + gn_cg additions instead of 16 x gn_cg on the 2 x bins threads that finish the block (640 at 40 channels per group)
    __shared__ float2 colt[W];
    if ((int)threadIdx.x < W) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float2 v = colp[r][threadIdx.x]; a += v.x; q += v.y; }
      colt[threadIdx.x] = make_float2(a, q);
    }
    __syncthreads();
    const int bins_blk = W / p.gn_cg;            // gemm_fused_gn_ok(): W % gn_cg == 0
    if ((int)threadIdx.x < 2 * bins_blk && mbase < p.M) {
      const int which = threadIdx.x & 1, lb = threadIdx.x >> 1;
      const int bin = blockIdx.x * bins_blk + lb;
      if (bin < p.gn_groups) {
        float a = 0.f;
        for (int c = 0; c < p.gn_cg; ++c) { const float2 v = colt[lb * p.gn_cg + c]; a += which ? v.y : v.x; }
        const int b = mbase / p.rows_per_batch;
        const int slab = (mbase - b * p.rows_per_batch) / ROWS;
        const int nslab = p.rows_per_batch / ROWS;
        p.gn_stats[(((size_t)b * nslab + slab) * p.gn_groups + bin) * 2 + which] = a;
      }
    }
  }
}

template <int RPT, int W>
__global__ __launch_bounds__(4 * W) void gemm_splitk_reduce_gn_kernel(const GemmArgs p) {
  kernarg_warm<sizeof(GemmArgs)>();
  __shared__ float2 part[16][W / 4];     // per (row lane, column quad) {sum, sum of squares} over the thread's rows
  __shared__ float2 quad[W / 4];         // per column quad, over the 16 row lanes
  __shared__ float2 mr[W / 4];           // per group of the block: {mean, rstd}   (fn_cg >= 4)
  splitk_finish_unit<RPT, W>(p, blockIdx.y, blockIdx.x * W, threadIdx.x, part, quad, mr, false);
}
// blocks of 40 columns where 80-column blocks would leave half the CUs without one and the groups / bins allow it
static inline int reduce_gn_width(const GemmArgs& a) {
  const bool ok40 = 40 % a.fn_cg == 0 && (a.gn_stats == nullptr || 40 % a.gn_cg == 0) && a.N % 40 == 0;
  if (!ok40) return 80;
  return ((int64_t)(a.N / 80) * (a.M / a.rows_per_batch) < 256) ? 40 : 80;
}
// geometries the fused reducer takes: whole samples of 64 or 256 rows, 80-column blocks of whole groups, the plain bf16 epilogue
bool gemm_fused_norm_ok(const GemmArgs& a) {
  const int rows = a.rows_per_batch;
  const int M = (a.conv && a.ups == 2) ? 0 : a.M;      // (not the 4-tap upsample form: its partials are filed per parity class)
  return (rows == 64 || rows == 256) && M > 0 && M % rows == 0 && a.N % 80 == 0 && a.fn_cg >= 4 && a.fn_cg % 4 == 0 && 80 % a.fn_cg == 0 &&
         a.out_mode == OUT_BF16 && a.act == ACT_NONE && !a.resid_f32 && !a.row_stats && !a.ln_stats && a.alpha == 1.f && a.ldc == a.N &&
         (a.gn_stats == nullptr || (a.gn_cg >= 4 && a.gn_cg % 4 == 0 && 80 % a.gn_cg == 0));
}

// width of the split-K reducer's blocks: 80 when the fused GroupNorm bins do not divide 64 (UNet: 5 / 10 / 20 / 40 channels)
// (160 where N allows: a block row is then 640 B = five whole 128-B lines of a partial slice; 80-wide blocks read 320-B pieces, every
// other one straddling a line: loop 568.3 -> 562.5 ms, profiles/r02_big_tile.md)
static inline int reduce_width(const GemmArgs& a) {
  if (a.gn_stats && 64 % a.gn_cg != 0) return (a.N % 160 == 0 && 160 % a.gn_cg == 0) ? 160 : 80;
  return 64;
}
// rows per block of the reducer: 64 for large outputs, 16 when there would be too few blocks to pull the partials
// (32-row blocks where they would still give every CU two blocks: measured equal, 536.2 vs 536.3 ms on the loop — not kept)
static inline int reduce_rows(const GemmArgs& a) {
  return ((int64_t)cdiv(a.N, reduce_width(a)) * cdiv(a.M, 64) >= 1024) ? 64 : 16;
}
// Long-K plain GEMMs whose width tiles by 160 (the feed-forward output GEMMs of UNet levels 1-3: K = 5 C) run on the convolutions'
// ping-pong kernel too: their 64 x 160 / 128 x 160 four-wave tiles fetch 22 / 14 KiB per MFLOP through a 64 B/clk L2 port, the 8-wave
// tiles 14 / 10 (round 4 A/B: 57 -> 43 us at level 1, loop -0.5 %)
// (round 6: ... and, from K = 640 up, the plain GEMMs whose ping-pong tiling is ONE full round of workgroups — proj_in / attn1.to_out of level 1 at the CFG batch 8,
// 8192 x 640 x 640: 256 ping-pong workgroups, one per CU, instead of 320 four-wave 128 x 128 tiles of which 64 CUs get two; 17.4 -> 14.3 us stand-alone,
// tools/pp_shortk_probe.py.  Fewer tiles than CUs (2048 x 1280 x 1280: 16.4 -> 18.3 us) and K = 320 (level 0: +-0) stay on the general tiles.)
// GILL_GEMM_PP = 0 (A/B switch, read once): no ping-pong tiles for the convolutions, the short-K plain GEMMs and the QKV GEMMs (the long-K plain GEMMs keep them)
static bool gemm_pp_switch() {
  static const int pp_env = [] { const char* v = getenv("GILL_GEMM_PP"); return v ? atoi(v) : 1; }();
  return pp_env != 0;
}
static bool gemm_plain_pingpong(int M, int N, int K) {
  if (N % 160 != 0 || M % 128 != 0 || K % 64 != 0) return false;
  if (K >= 2560) return true;
  // workgroups the launcher will make of it (gemm_launch_bn: 128-row tiles when 256-row ones give at most 128): ONE balanced round only — with
  // ten K steps a second, partial round costs more than the tile gains
  const int64_t wg256 = (int64_t)cdiv(M, 256) * (N / 160);
  const int64_t wg = (wg256 <= 128 || M % 256 != 0) ? (int64_t)(M / 128) * (N / 160) : wg256;
  return gemm_pp_switch() && K >= 640 && wg >= 200 && wg <= 256;
}

// ... and the launch's other conditions (tile_width() and gemm_launch_bn() used to test different things): the 8-wave plain kernel has
// the lean bf16 row-major epilogue only
// (per-sample weights — GemmArgs::wb_rows, the cross-attention values GEMM — where samples are whole 256-row tiles: round 6, loop 441.8 -> 440.0 ms)
static bool gemm_plain_pingpong_args(const GemmArgs& a) {
  return !a.conv && gemm_plain_pingpong(a.M, a.N, a.K) && a.act == ACT_NONE && a.out_mode == OUT_BF16 && !a.resid_f32 && a.wb_rows % 256 == 0 && !a.ln_stats;
}

// QKV GEMMs (head-major scatter epilogue, folded LayerNorm) on the 256 x 160 ping-pong tiles where those give a balanced round of one workgroup per CU
// (round 6): the workgroups fill >= 75 % of their last round of 256, and 128-row tiles would not fill theirs better.  Level 2 at the CFG batch 8
// (2048 x 3840 x 1280: 192 workgroups) 52.5 -> 38.5 us per launch, loop -0.9 %.  Returns the wave M sub-tile count (4) or 0.
// (Measured with it and not kept: the 128-row tiles where THEY are the balanced choice — level 1, 8192 x 1920 x 640, 768 workgroups = three rounds:
// 46.9 -> 42.7 us per launch, and the loop gains nothing because the card then clocks 15 MHz lower, three boxes, profiles/r06_skeleton_ab.log.
// Round 4 had these GEMMs on 128-wide ping-pong tiles at every level and lost 0.6-1 %: K = 320 at level 0 and the 1.5-round grids were in it.)
static int gemm_qkv_pingpong_mi(const GemmArgs& a) {
  if (!gemm_pp_switch() || a.conv || a.out_mode != OUT_QKV || a.N % 160 != 0 || a.M % 256 != 0 || a.K % 64 != 0 || a.K < 640 || a.splitk > 1 || a.wb_rows || a.w_blk64) return 0;
  auto fill = [](int64_t wg) { return (double)wg / (double)(((wg + 255) / 256) * 256); };
  const double f128 = fill((int64_t)(a.M / 128) * (a.N / 160)), f256 = fill((int64_t)(a.M / 256) * (a.N / 160));
  // ONE round only: several rounds of these tiles (level 1 at UNet batch 16: 768 workgroups) make the kernel faster and the card clock lower, like the
  // 128-row form at batch 8 — loop 764.6 ms without, 767.3 ms with, 15-35 MHz apart (profiles/r06_skeleton_ab.log, session r06_s33)
  if ((int64_t)(a.M / 256) * (a.N / 160) > 256) return 0;
  return (f256 >= 0.75 && f256 >= f128) ? 4 : 0;
}

// Tile width.  Tried and removed (numbers in profiles/r02_big_tile.md, profiles/r01_sweep_gemm_tiles.md): a 256 x 256 8-wave tile for
// the wide plain GEMMs (GEGLU 94 -> 100 us, QKV 46.6 -> 48.3 us: those kernels are bound by their epilogues), a 4-deep ring of
// 32-wide K stages (+7 % on the loop), a 3-deep ring on the 4-wave tiles.
static inline int tile_width(const GemmArgs& a) {
  if (a.act == ACT_GEGLU) return 128;
  // measured on MI355X (profiles/r01_sweep_gemm_tiles.md): the 128x160 tile (5 N sub-tiles per wave: more MFMAs per LDS
  // byte) beats 128x128 by 15-40 % on every UNet width, all of which are multiples of 160
  int bn = (a.N % 160 == 0) ? 160 : 128;
  // ... and except for the head-major scatter epilogue (QKV / to_q), whose columns fall into 48..160-wide heads: 128-wide tiles
  // measured 579.3 -> 576.3 ms on the loop
  if (!a.conv && a.out_mode == OUT_QKV && a.N % 128 == 0 && !gemm_qkv_pingpong_mi(a)) bn = 128;
  // ... except on the 64-row tile (plain GEMMs with few tiles, see gemm_launch_bn): 64 x 128 needs 48 KiB of LDS, i.e. three
  // workgroups (six waves) per CU instead of two (four) — loop 587.3 -> 584.7 ms
  if (!a.conv && a.splitk <= 1 && a.N % 128 == 0 && !a.gn_stats && (int64_t)cdiv(a.M, 128) * cdiv(a.N, 160) < 300 && a.M > 64 && a.out_mode != OUT_SOFTMAX80 &&
      !gemm_plain_pingpong_args(a)) bn = 128;
  return bn;
}
int gemm_row_planes(const GemmArgs& a) {
  if (a.splitk > 1) return cdiv(a.N, reduce_width(a));
  return 2 * cdiv(a.N, tile_width(a));
}
int gemm_gn_slab_rows(const GemmArgs& a) {
  if (a.splitk > 1 && (a.fn_Y || gemm_coop_ok(a))) return a.rows_per_batch;      // the fused reducer / the in-kernel finish files one partial per (sample, bin)
  return a.splitk > 1 ? reduce_rows(a) : GN_SLAB_ROWS;
}
bool gemm_fused_gn_ok(int N, int cg) {
  if (cg < 1 || N % cg != 0 || N / cg > 64) return false;
  const int bn = (N % 160 == 0) ? 160 : 128;
  // in-kernel epilogue: bins must not straddle N tiles; split-K reducer: nor its 64- or 80-column blocks
  return bn % cg == 0 && (64 % cg == 0 || (80 % cg == 0 && N % 80 == 0));
}


// 3x3 convolutions whose row count is a multiple of 256 and whose width tiles by 160 run on the 256 x 160 ping-pong kernel
// (GILL_GEMM_PP = 0: two co-resident 128 x 160 workgroups instead, the round-1 structure)
// (round 4: 256 x 128 ping-pong tiles for the widths 160 does not divide — the VAE decoder's 128 / 256 / 512-channel convolutions: 484.8 /
// 403.2 / 303.9 us against 464.8 / 406.1 / 301.0 us on the four-wave 128 x 128 tiles, VAE decode 12.74 vs 12.69 ms.  Not kept.)
bool gemm_conv_pingpong(int rows_multiple_of, int Cout) {
  return gemm_pp_switch() && rows_multiple_of % 256 == 0 && Cout % 160 == 0;
}

// ---- COOP (GemmArgs::coop_ctr): which launches may finish in-kernel.
static int device_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
  }();
  return n;
}
// rows of the ping-pong tile gemm_launch_bn() runs this launch on (0: it is not a ping-pong launch, or a form COOP does not take)
static int coop_tile_rows(const GemmArgs& a, int sk) {
  if (a.N % 160 != 0) return 0;
  const int tiles_n = a.N / 160;
  if (a.conv) {
    if (a.ups || !gemm_conv_pingpong(a.M, a.N)) return 0;      // (upsample forms: parity classes in blockIdx.z / a 9-tap gather — their finish stays a launch)
    return ((int64_t)cdiv(a.M, 256) * tiles_n * sk <= 128 && a.M % 128 == 0) ? 128 : 256;
  }
  if (!gemm_plain_pingpong_args(a)) return 0;
  return ((int64_t)cdiv(a.M, 256) * tiles_n * sk <= 128 || a.M % 256 != 0) ? 128 : 256;
}
// GILL_GEMM_COOP = 0: every finish as its own launch (split-K reducers, GroupNorm-apply), the round-5 dataflow (A/B and the on / off parity test);
// 1 (default): the GroupNorm finish in the 3x3 convolutions' epilogue (EPI 6: a measured go); 2: also the split-K finish (EPI 7: a measured no-go)
int gemm_coop_mode() {
  static const int mode = [] { const char* e = getenv("GILL_GEMM_COOP"); return e ? atoi(e) : 1; }();
  return mode;
}
bool gemm_coop_ok(const GemmArgs& a) {
  if (!a.coop_ctr || gemm_coop_mode() == 0) return false;
  const int sk = a.splitk > 1 ? a.splitk : 1;
  if (sk > 1 && !a.coop_splitk) return false;
  const int bm = coop_tile_rows(a, sk);
  const int rpb = a.rows_per_batch;
  if (bm == 0 || rpb <= 0 || a.M % bm != 0 || a.M % rpb != 0) return false;
  // co-residency: the waiting workgroups must all be on the chip — one per CU (the ping-pong tiles take a CU's whole LDS), so the grid may not
  // exceed the CU count (MI355X: 256; a device with fewer falls back to the launches)
  if ((int64_t)(a.M / bm) * (a.N / 160) * sk > device_cus()) return false;
  if (a.out_mode != OUT_BF16 || a.act != ACT_NONE || a.resid_f32 || a.row_stats || a.ln_stats || a.alpha != 1.f || a.wb_rows || a.partials_only || a.w_blk64)
    return false;
  if (sk > 1) {
    // split-K finish: (whole sample, 40 columns) units on the 128 x 160 tile, the reducer's geometries (64 / 256 rows per sample)
    if (bm != 128 || !(rpb == 64 || rpb == 256) || a.fn_ss) return false;
    if (a.fn_Y) { if (a.fn_cg < 4 || a.fn_cg % 4 != 0 || 40 % a.fn_cg != 0 || !a.fn_gamma || !a.fn_beta) return false; }
    else if (!a.C) return false;
    if (a.gn_stats && (a.gn_cg < 4 || a.gn_cg % 4 != 0 || 40 % a.gn_cg != 0)) return false;
    return true;
  }
  // GroupNorm finish in the convolution's epilogue: whole M tiles per sample, <= 64 slab partials per bin (the apply kernel's own limit),
  // <= 32 bins per N tile, groups made of whole bins inside one N tile
  if (!a.conv || (!a.fn_Y && !a.fn_ss) || !a.fn_gamma || !a.fn_beta) return false;
  if (!a.gn_stats || a.gn_cg <= 0 || 160 % a.gn_cg != 0 || 160 / a.gn_cg > 32 || a.gn_cg * a.gn_groups != a.N) return false;
  if (a.fn_cg <= 0 || a.fn_cg % a.gn_cg != 0 || 160 % a.fn_cg != 0) return false;
  if (rpb % bm != 0 || rpb % GN_SLAB_ROWS != 0 || rpb / GN_SLAB_ROWS > 64) return false;
  return true;
}
int gemm_coop_counters(const GemmArgs& a) { return cdiv(a.M, 128) * cdiv(a.N, 160); }     // (an upper bound: one per 128-row tile)
// Reads and clears the give-up count of the in-kernel finishes (a small synchronous copy: call it where the host waits anyway).  Non-zero means
// outputs of an earlier launch were NaN-poisoned because waiting workgroups were not co-resident: this process shared the GPU with another stream
// or process that ALSO ran waiting workgroups (two of them can starve each other for CUs).  The contract is exclusive use of the device by ONE
// handle's stream while it runs, or GILL_GEMM_COOP=0.
int gemm_coop_giveups(unsigned* count) {
  unsigned n = 0;
  *count = 0;
  if (gemm_coop_mode() == 0) return 0;
  GILL_CHECK_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_coop_giveups), sizeof(n)));
  if (n) {
    const unsigned zero = 0;
    GILL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_giveups), &zero, sizeof(zero)));
  }
  *count = n;
  return 0;
}

bool conv_k_chunked(int HW, int Cin, int Cout) {
  return HW >= 4096 && (int64_t)HW * Cin * 2 > (int64_t)4 << 20 && !gemm_conv_pingpong(HW, Cout);
}

// STREAM64.  Weight-streaming GEMMs (a few hundred rows against 33-134 MB matrices: OPT-6.7b at <= 8 prompts) ran at 2.2-3 TB/s of weights on
// the general tiles, whatever the ring depth (profiles/r05_opt_stream64.md: 128 x 64 tiles with a 6-deep ring streamed no faster than 64 x 128
// tiles with a 3-deep one): a K step fetched 64-128 weight ROWS of 128 B each at a stride of 2 K bytes, i.e. one DRAM page activation per 128 B.
// STREAM64 changes the layout, not only the tile: the matrix is stored at load time as [N / 64][K / 64][64][64] (gemm_stream64_weights() says
// which matrices; convert_to_bf16_blk64_launch() writes them), so the 128 x 64 tile's weight stage is one contiguous 8 KiB run, a workgroup's
// whole K walk one contiguous K * 128 B stream, fetched with the non-temporal policy (MI355X_MICROARCH "nt-weights") into a 6-deep ring
// (144 KiB, one workgroup of eight waves per CU, 40 KiB of weights + 80 KiB of L2-resident activations in flight per CU).  All rows of a <= 128-row problem
// sit in the one M tile (every weight is read once); more rows = more M tiles next to each other on one XCD (n_major), sharing the L2.
// No split for >= 160 workgroups; narrower matrices split K to ~256 workgroups.  (The A/B switch of round 5 is gone: row-major weights on the general
// tiles measured 5.95-6.08 ms for the OPT stage against 4.31 ms, profiles/r05_opt_stream64.md.)
bool gemm_stream64_weights(int N, int K) {
  return N % 64 == 0 && K % 64 == 0 && K >= 1024 && (int64_t)N * K >= ((int64_t)4 << 20);
}
int gemm_pick_splitk_blk64(int M, int N, int K) {
  if (M > GEMM_STREAM64_MAX_ROWS) return gemm_pick_splitk(M, N, K, ACT_NONE);     // (runs on the general tiles: gemm_launch())
  const int tiles = cdiv(M, 128) * (N / 64), ksteps = K / BK;
  if (tiles >= 160) return 1;
  int s = (256 + tiles / 2) / tiles;
  while (s > 1 && ksteps / s < 8) --s;
  if (s > 16) s = 16;
  // no empty trailing split (ksteps = 129, s = 16 -> 9 steps per split, split 15 would start past the end and write a zero plane)
  return cdiv(ksteps, cdiv(ksteps, s));
}

int gemm_pick_splitk(int M, int N, int K, int act, bool plain, bool generic) {
  if (act == ACT_GEGLU) return 1;
  const int bn = (N % 160 == 0) ? 160 : 128;
  const int tiles = cdiv(M, BM_HOST) * cdiv(N, bn);
  const int ksteps = K / BK;
  if (tiles >= 384 || ksteps < 8) return 1;
  {   // convs that gemm_launch_bn puts on 128 x 160 ping-pong tiles: one workgroup per CU, 256 slots
    if (!generic && (plain ? gemm_plain_pingpong(M, N, K) : gemm_conv_pingpong(M, N)) && tiles <= 256 && M % 128 == 0) {
      int s = (256 + tiles / 2) / tiles;
      const int min_steps = (M <= 256 || tiles < 64) ? 4 : 24;
      if (s > ksteps / min_steps) s = ksteps / min_steps;
      if (s > 16) s = 16;
      return s < 1 ? 1 : s;
    }
  }
  // (round 5, first session: an M-dependent rule for skinny weight-streaming GEMMs lived here.  The OPT matrices now take gemm_pick_splitk_blk64();
  // for everything else it made the split factor — the fp32 summation order — depend on the batch, and the tiny UNet's batch-invariance check went
  // from bit-identical to 2.4e-2 after 11 recurrent calls.  Removed.)
  // plain GEMMs with 150..383 128-row tiles run on 64-row tiles instead (gemm_launch: >= 300 workgroups, no partials):
  // measured 8192 x 640 x 3200 unsplit 49.1 us, two-way split + reducer 54.6 us
  if (plain && !generic && tiles >= 150 && tiles < 300 && M > 64) return 1;
  int s = (512 + tiles / 2) / tiles;   // aim for ~2 workgroups per CU (512 resident slots)
  // each split pays an fp32 partial write + a reduce pass: keep >= 24 K steps per split (measured: K = 1280 GEMMs lose
  // from any split, K >= 5120 convs win up to 4-8 ways), except for skinny weight-streaming GEMMs (OPT, M <= 256)
  // where filling every CU with HBM requests matters more than the tiny partials
  const int min_steps = (M <= 256 || tiles < 64) ? 4 : 24;
  if (s > ksteps / min_steps) s = ksteps / min_steps;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return s;
}

int gemm_splitk_reduce_launch(const GemmArgs& a, hipStream_t s) {
  GILL_REQUIRE(a.splitk > 1 && a.ws != nullptr, "split-K reducer: no partials");
  if (a.fn_Y) {
    GILL_REQUIRE(gemm_fused_norm_ok(a) && a.fn_gamma && a.fn_beta, "split-K reducer: unsupported fused GroupNorm geometry");
    const int w = reduce_gn_width(a);
    const dim3 rg(a.N / w, a.M / a.rows_per_batch);
    if (a.rows_per_batch == 256 && w == 80) hipLaunchKernelGGL((gemm_splitk_reduce_gn_kernel<16, 80>), rg, dim3(320), 0, s, a);
    else if (a.rows_per_batch == 256) hipLaunchKernelGGL((gemm_splitk_reduce_gn_kernel<16, 40>), rg, dim3(160), 0, s, a);
    else if (w == 80) hipLaunchKernelGGL((gemm_splitk_reduce_gn_kernel<4, 80>), rg, dim3(320), 0, s, a);
    else hipLaunchKernelGGL((gemm_splitk_reduce_gn_kernel<4, 40>), rg, dim3(160), 0, s, a);
    GILL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const int rw = reduce_width(a), rr = reduce_rows(a);
  const dim3 rg(cdiv(a.N, rw), cdiv(a.M, rr));
  if (rw == 160 && rr == 64) hipLaunchKernelGGL((gemm_splitk_reduce_kernel<160, 4>), rg, dim3(640), 0, s, a);
  else if (rw == 160) hipLaunchKernelGGL((gemm_splitk_reduce_kernel<160, 1>), rg, dim3(640), 0, s, a);
  else if (rw == 80 && rr == 64) hipLaunchKernelGGL((gemm_splitk_reduce_kernel<80, 4>), rg, dim3(320), 0, s, a);
  else if (rw == 80) hipLaunchKernelGGL((gemm_splitk_reduce_kernel<80, 1>), rg, dim3(320), 0, s, a);
  else if (rr == 64) hipLaunchKernelGGL((gemm_splitk_reduce_kernel<64, 4>), rg, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_splitk_reduce_kernel<64, 1>), rg, dim3(256), 0, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int NWV, int BN, int CONV, int EPI, int STAGES, int KT = BK, int MI = 4>
static int gemm_launch_inst(const GemmDev& d, dim3 grid, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = STAGES * ((NWV / 2) * MI * 16 * KT + BN * KT) * (int)sizeof(bf16_t);
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<NWV, BN, CONV, EPI, STAGES, KT, MI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<NWV, BN, CONV, EPI, STAGES, KT, MI>), grid, dim3(NWV * 64), smem, s, d);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// d.nwv = 2 selects the 64-row tile (plain GEMMs, 2-deep ring, no split-K)
template <int BN, int CONV, int EPI>
static int gemm_launch_stages(const GemmDev& d, dim3 grid, hipStream_t s) {
  if constexpr (BN == 64) {
    // STREAM64 (see gemm_stream64()): 128 x 64 tile, 6-deep ring = 144 KiB, one workgroup per CU
    static_assert(CONV == 0 && (EPI == 0 || EPI == 2 || EPI == 3), "the 64-wide streaming tile: plain GEMMs, row-major / partial / QKV epilogues");
    return gemm_launch_inst<8, 64, 0, EPI, 6, BK, 2>(d, grid, s);     // eight waves of 32 x 32: see gemm_launch_stream64()
  } else {
  if constexpr (CONV == 0 && EPI != 2) {
    if (d.nwv == 2) return gemm_launch_inst<2, BN, CONV, EPI, 2>(d, grid, s);
  }
  if constexpr (CONV == 0 && (EPI == 0 || EPI == 4) && BN == 160) {
    if (d.nwv == 4 && d.mi == 2) {
      // (2-deep ring: 56 KB, two workgroups per CU; 3-deep = 84 KB = one per CU measured slower: 495.7 vs 489.4 ms)
      return gemm_launch_inst<4, BN, CONV, EPI, 2, BK, 2>(d, grid, s);
    }
  }
  if constexpr (CONV == 0 && EPI == 5) {
    if (d.nwv == 8) return gemm_launch_inst<8, BN, CONV, EPI, 3, BK, 2>(d, grid, s);     // ping-pong 128 x 160 tile
    // 64 x 160 tile on four waves of 32 x 80.  Measured on the three UNet shapes of the 8-sample batch back to back (tools/xalg_bench.py, 20
    // launches each, operands warm): two waves of 64 x 80 1177 us, this with a 2-deep ring 1003-1013 us, 3-deep 1212 us, 3-deep only on
    // the <= 256-workgroup grids 1051 us.  In the UNet forward the per-sample weights are cold (13 MB per layer, read once per call) and
    // a workgroup's K walk is a chain of HBM round trips: there the 3-deep ring on the one-workgroup-per-CU grids (levels 2-3) wins,
    // loop 446.7 -> 444.1 ms (4-deep: 444.7).
    if (d.nwv == 4 && d.mi == 2) {
      if ((int)(grid.x * grid.y) <= 256) return gemm_launch_inst<4, BN, CONV, EPI, 3, BK, 2>(d, grid, s);
      return gemm_launch_inst<4, BN, CONV, EPI, 2, BK, 2>(d, grid, s);
    }
  }
  if constexpr (CONV == 0 && (EPI == 0 || EPI == 3 || EPI == 4) && BN == 128) {
    if (d.nwv == 4 && d.mi == 2) {
      // 64-row tile on four waves, 3-deep ring (72 KB: two workgroups per CU).  These grids are 1-2 workgroups per CU, so the ring depth IS
      // the number of K stages in flight on a CU, and a K step of these short-K GEMMs is one L2 round trip: 2-deep 487.3 ms, 3-deep
      // 483.4 ms, 4-deep (one workgroup per CU) 496.1 ms on the loop.
      return gemm_launch_inst<4, BN, CONV, EPI, 3, BK, 2>(d, grid, s);
    }
  }
  if constexpr (BN == 160 && (CONV != 0 || EPI == 2 || EPI == 4 || (CONV == 0 && EPI == 3))) {
    if (d.nwv == 8 && d.mi == 2) return gemm_launch_inst<8, BN, CONV, EPI, 3, BK, 2>(d, grid, s);   // ping-pong 128 x 160 tile
    if (d.nwv == 8) return gemm_launch_inst<8, BN, CONV, EPI, 3>(d, grid, s);     // ping-pong 256 x 160 tile
  }
  if constexpr (CONV == 0 && BN == 128 && (EPI == 1 || EPI == 3)) {
    // round 5: the 128 x 128 tile of the GEGLU / QKV GEMMs on eight waves of 32 x 64 (two workgroups = 16 waves per CU, 116 registers) instead of four of
    // 64 x 64: half the LDS-DMA pieces, fragment reads and MFMAs per wave and step, twice the waves to overlap them.  Loop 447.4 -> 445.8 ms (A/B x3,
    // profiles/r05_linears.md); level-1 GEGLU 82.1 -> 79.7 us stand-alone
    if (d.nwv == 4 && d.mi == 4) return gemm_launch_inst<8, BN, CONV, EPI, 2, BK, 2>(d, grid, s);
  }
  // (the same for the UNet's plain 128-row GEMMs — the lean row-major epilogue, EPI 4 — measured slower: loop 451.3 -> 456.0 ms, profiles/r05_linears.md)
  return gemm_launch_inst<4, BN, CONV, EPI, 2>(d, grid, s);
  }
}

// XCD BLOCK MAP.  Model of the fabric traffic of one launch under a map that gives every XCD an xm x xn block of the (M tile, N group) grid (all
// K splits of a tile sit on its XCD: blockIdx.y does not move the XCD when gridDim.x % 8 == 0):
//   activations: the block's rows x the channels a K walk reads, once if they fit next to the streaming weights in the 4 MiB L2 (budget 2 MiB),
//                else once per re-read (a tap-major 3x3 convolution re-reads its input rows 9 times, a plain GEMM once per N tile a workgroup walks);
//   weights:     the block's N columns x K, once (the block's M tiles walk K in step and share the stream).
// Picks the block with the least modelled traffic; the two range maps (n_major / m-major) are candidates like any other, and the block map is only
// used where it beats the range map in force by 15 % (profiles/r06_xcd_block_map.md: level-2 convolutions 417 -> ... MB per dispatch).
static void xcd_block_pick(const GemmArgs& a, GemmDev& d, int bm_rows, int bn_cols, int ncls) {
  d.xb_m = d.xb_n = 0;
  const int nwg = d.tiles_m * d.groups_n;
  // (the ping-pong tiles only — the 3x3 convolutions and the long-K plain GEMMs of levels 1-3, one workgroup per CU: the launches whose fetch
  // profiles/r05_fetch_by_kernel.md flagged; the model is not trusted on the multi-N-tile walks of the GEGLU / QKV kernels)
  if (ncls != 1 || nwg % 8 != 0 || nwg < 16 || a.wb_rows || d.nwv != 8 || bn_cols != 160) return;
  const double l2_budget = 2.0 * 1024 * 1024;
  const int taps = a.conv ? 9 : 1;
  const double row_bytes = 2.0 * (a.conv ? (double)(a.Cin + a.KX) * (a.stride == 2 ? 4 : 1) : (double)a.K);     // input bytes behind one output row
  const double col_bytes = 2.0 * (double)a.K * bn_cols * d.npw;                                                   // weight bytes of one N group
  auto cost = [&](double xm, double xn) {
    const double A = xm * bm_rows * row_bytes, W = xn * col_bytes;
    // (a plain GEMM's N groups of one M tile run side by side on the XCD and share the activation stream; only a workgroup's own walk over npw N
    // tiles comes back to the same rows later)
    const double rereads = (A <= l2_budget) ? 1.0 : (a.conv ? (a.k_chunked ? 1.0 : (double)taps) : (double)d.npw);
    return A * rereads + W;
  };
  double best = 1e30; int bxm = 0, bxn = 0;
  for (int blocks_m = 1; blocks_m <= 8; blocks_m *= 2) {
    const int blocks_n = 8 / blocks_m;
    if (d.tiles_m % blocks_m != 0 || d.groups_n % blocks_n != 0) continue;
    const int xm = d.tiles_m / blocks_m, xn = d.groups_n / blocks_n;
    const double c = cost(xm, xn);
    if (c < best) { best = c; bxm = xm; bxn = xn; }
  }
  if (bxm == 0) return;
  // the range map in force: n_major = (all of M, 1/8 of N) per XCD, else (1/8 of M, all of N) — priced with the same model
  const double cur = d.n_major ? cost(d.tiles_m, d.groups_n / 8.0) : cost(d.tiles_m / 8.0, d.groups_n);
  if (best <= 0.85 * cur) { d.xb_m = bxm; d.xb_n = bxn; }
}

template <int BN>
static int gemm_launch_bn(const GemmArgs& a, hipStream_t s) {
  GemmDev d;
  d.a = a;
  d.zero = gill_zero_page();
  GILL_REQUIRE(d.zero != nullptr, "zero page unavailable");
  const int sk = a.splitk > 1 ? a.splitk : 1;
  d.a.splitk = sk;
  GemmArgs red = d.a;       // what the split-K reducer sees: the output tensor, whatever the conv kernel's row space is
  // UPS4 (a.ups == 2): the kernel's rows are the SOURCE pixels of one parity class, blockIdx.z the class
  const int ncls = (a.conv && a.ups == 2) ? 4 : 1;
  const int Mk = a.M / ncls;
  d.a.M = Mk;
  d.a.rows_per_batch = a.rows_per_batch / ncls;
  d.tiles_n = cdiv(a.N, BN);
  // 64-row tiles for plain GEMMs whose 128-row tiling leaves CUs without a tile
  d.nwv = 4;
  // (tile_width() counts 128 x 160 tiles when it hands these GEMMs BN = 128, this rule counts 128 x 128 tiles: an 8192 x 640 GEMM
  // — 256 vs 320 — therefore lands on the 4-wave 128 x 128 tile, not the 64-row one; measured better that way: 567.8 vs 575.5 ms)
  // (re-measured with the 3-deep ring below: < 300 493.8 ms, < 400 495.6, < 520 496.3, < 700 496.5)
  if (!a.conv && sk == 1 && (int64_t)cdiv(a.M, 128) * d.tiles_n < 300 && a.M > 64) d.nwv = 2;   // measured: < 300 +0.3 %, < 520 -3 % end to end
  if (a.wb_rows % 128 != 0 && !a.conv && sk == 1) d.nwv = 2;      // per-sample weights on samples of 64 rows (GemmArgs::wb_rows): 64-row tiles
  d.mi = 4;
  // the 64-row tile on FOUR waves (2 x 2, wave tile 32 x 64) where its width is 128: same LDS, twice the waves per CU —
  // loop 580.4 -> 576.3 ms, 2048 x 1280 x 1280 19.8 -> 17.3 us
  if (d.nwv == 2 && BN == 128 && a.act != ACT_GEGLU && !a.gn_stats) { d.nwv = 4; d.mi = 2; }
  // ... and where it is 160 (the feed-forward output GEMMs of levels 1-3 with their fused GroupNorm partials: bins of 20 / 40 channels
  // need the 160-wide tile): loop 492.4 -> 489.4 ms against the two-wave tile.
  if (d.nwv == 2 && BN == 160 && a.act == ACT_NONE && a.out_mode == OUT_BF16) { d.nwv = 4; d.mi = 2; }
  if (a.out_mode == OUT_SOFTMAX80) { d.nwv = 4; d.mi = 2; }      // (see gemm_launch_stages)
  // ... or, where 128-row ping-pong tiles are exactly one balanced round (level 1 at the CFG batch 8: 8192 x 640 x 640 = 256 workgroups instead of 512
  // four-wave ones), those (round 6, like the plain GEMMs of that shape)
  if (BN == 160 && a.out_mode == OUT_SOFTMAX80 && gemm_pp_switch() && sk == 1 && a.M % 128 == 0 && a.wb_rows % 128 == 0 && a.K >= 640 &&
      (int64_t)(a.M / 128) * d.tiles_n >= 200 && (int64_t)(a.M / 128) * d.tiles_n <= 256) { d.nwv = 8; d.mi = 2; }
  // (round 4: the 64-row four-wave tile with its 3-deep ring — 96 instead of 64 KB in flight per CU — on ALL GEGLU / QKV / 128-wide plain
  // GEMMs, not only the few-tile ones: loop 460.4 -> 469.4 / 464.8 / 460.7 ms.  Not kept.)
  // 3x3 convolutions: 256 x 160 tile with the ping-pong main loop, one workgroup per CU (GILL_GEMM_PP = 0 keeps the two
  // co-resident 128 x 160 workgroups)
  if (BN == 160 && a.conv && gemm_conv_pingpong(Mk, a.N)) {
    d.nwv = 8;
    // Where the 256-row tiling x split-K gives at most 128 workgroups (UNet levels 1-3 at the 8-sample batch), run
    // 128 x 160 ping-pong tiles (eight waves of 32 x 80); gemm_pick_splitk() then aims at 256 workgroups of those, i.e. half the
    // split factor: half the fp32 partials (none at level 1)
    // (loop 561.1 -> 558.0 ms; GILL_GEMM_PP128 = 0 keeps the 256-row tile with twice the split)
    if ((int64_t)cdiv(Mk, 256) * ncls * d.tiles_n * sk <= 128 && Mk % 128 == 0) d.mi = 2;
  }
  if (BN == 160 && gemm_plain_pingpong_args(a)) {
    d.nwv = 8; d.mi = 4;
    if (((int64_t)cdiv(Mk, 256) * d.tiles_n * sk <= 128) || Mk % 256 != 0) d.mi = 2;
  }
  if (BN == 160 && gemm_qkv_pingpong_mi(a)) { d.nwv = 8; d.mi = gemm_qkv_pingpong_mi(a); }
  // (round 4: the short-K GEGLU / QKV GEMMs on 256 x 128 / 128 x 128 ping-pong tiles, one N tile per workgroup: loop 464.0 -> 478.0 / 480.4 ms
  // (GEGLU), 467.0 / 468.8 ms (QKV) — ten K steps do not amortise a prologue and an epilogue that no second workgroup covers.  Not kept.)
  d.kt = 64;
  d.ksteps = a.K / d.kt;
  d.ksteps_per_split = cdiv(d.ksteps, sk);
  const int tiles_m = cdiv(Mk, (d.nwv / 2) * d.mi * 16);
  // Short-K GEMMs with many N tiles (GEGLU, QKV: K = 320..1280, 6..80 N tiles) spend most of a tile's life in the first-load
  // latency and the epilogue.  Let one workgroup walk `npw` N tiles back to back instead: the ring is staged across tile
  // boundaries, so the loads of tile t+1 fly during the epilogue of tile t.  Keep ~2 workgroups per CU.
  d.npw = 1;
  if (!a.conv && sk == 1 && d.nwv != 8 && !a.gn_stats && d.tiles_n >= 2) {
    int groups = cdiv(512, tiles_m);   // ~2 workgroups per CU
    if (groups < 1) groups = 1;
    if (groups > d.tiles_n) groups = d.tiles_n;
    d.npw = cdiv(d.tiles_n, groups);
  }
  d.groups_n = cdiv(d.tiles_n, d.npw);
  d.tiles_m = tiles_m;
  {
    const int64_t wel = (int64_t)a.N * a.K;
    const int64_t ael = (int64_t)a.M * (a.conv ? a.Cin + a.KX : a.K);
    d.n_major = (wel > ael && d.groups_n >= 8) ? 1 : 0;
  }
  xcd_block_pick(a, d, (d.nwv / 2) * d.mi * 16, BN, ncls);
  GILL_REQUIRE(a.wb_rows == 0 || (a.wb_rows % ((d.nwv / 2) * d.mi * 16) == 0 && !a.conv), "per-sample weights: tiles must not straddle samples");
  dim3 grid(tiles_m * d.groups_n, sk, ncls);
  if (d.nwv == 8) GILL_REQUIRE(d.npw == 1, "internal: the ping-pong kernel walks one N tile per workgroup");
  // COOP: the finish inside this launch (EPI 6: GroupNorm in the conv epilogue; EPI 7: the split-K reduction) — no second launch
  bool coop = false;
  if constexpr (BN == 160) {
    coop = gemm_coop_ok(d.a);
    if (coop) GILL_REQUIRE(d.nwv == 8 && (d.nwv / 2) * d.mi * 16 == coop_tile_rows(d.a, sk) && (sk == 1 || d.mi == 2) && ncls == 1,
                           "internal: gemm_coop_ok() and the launcher disagree about the tile");
  }
  GILL_REQUIRE(coop || sk > 1 || (a.fn_Y == nullptr && a.fn_ss == nullptr), "fused GroupNorm output without a split-K reducer or an in-kernel finish");
  if (a.conv) {
    if (a.ups == 2) {
      if (sk > 1) GILL_TRY((gemm_launch_stages<BN, 3, 2>(d, grid, s)));
      else GILL_TRY((gemm_launch_stages<BN, 3, 0>(d, grid, s)));
    } else if (a.ups) {
      if (sk > 1) GILL_TRY((gemm_launch_stages<BN, 2, 2>(d, grid, s)));
      else GILL_TRY((gemm_launch_stages<BN, 2, 0>(d, grid, s)));
    } else if (coop) {
      if constexpr (BN == 160) {
        if (sk > 1) GILL_TRY((gemm_launch_inst<8, 160, 1, 7, 3, BK, 2>(d, grid, s)));
        else if (d.mi == 2) GILL_TRY((gemm_launch_inst<8, 160, 1, 6, 3, BK, 2>(d, grid, s)));
        else GILL_TRY((gemm_launch_inst<8, 160, 1, 6, 3, BK, 4>(d, grid, s)));
      }
    } else {
      if (sk > 1) GILL_TRY((gemm_launch_stages<BN, 1, 2>(d, grid, s)));
      else GILL_TRY((gemm_launch_stages<BN, 1, 0>(d, grid, s)));
    }
  } else {
    if (sk > 1 && coop) {
      if constexpr (BN == 160) GILL_TRY((gemm_launch_inst<8, 160, 0, 7, 3, BK, 2>(d, grid, s)));
    }
    else if (sk > 1) GILL_TRY((gemm_launch_stages<BN, 0, 2>(d, grid, s)));
    else if (a.act == ACT_GEGLU) {
      if constexpr (BN == 128) GILL_TRY((gemm_launch_stages<BN, 0, 1>(d, grid, s)));
      else GILL_REQUIRE(BN == 128, "internal: GEGLU runs on 128-wide tiles");
    }
    else if (a.out_mode == OUT_SOFTMAX80) {
      if constexpr (BN == 160) GILL_TRY((gemm_launch_stages<BN, 0, 5>(d, grid, s)));
      else GILL_REQUIRE(BN == 160, "internal: the softmax epilogue runs on 160-wide tiles");
    }
    else if (a.out_mode == OUT_QKV) GILL_TRY((gemm_launch_stages<BN, 0, 3>(d, grid, s)));
    else if (a.act == ACT_NONE && a.out_mode == OUT_BF16 && !a.resid_f32) GILL_TRY((gemm_launch_stages<BN, 0, 4>(d, grid, s)));
    else GILL_TRY((gemm_launch_stages<BN, 0, 0>(d, grid, s)));
  }
  if (sk > 1 && !a.partials_only && !coop) return gemm_splitk_reduce_launch(red, s);
  return 0;
}

static int gemm_launch_stream64(const GemmArgs& a, hipStream_t s) {
  GemmDev d;
  d.a = a;
  d.zero = gill_zero_page();
  GILL_REQUIRE(d.zero != nullptr, "zero page unavailable");
  const int sk = a.splitk > 1 ? a.splitk : 1;
  d.a.splitk = sk;
  d.tiles_n = a.N / 64; d.tiles_m = cdiv(a.M, 128); d.npw = 1; d.groups_n = d.tiles_n; d.n_major = 1; d.xb_m = d.xb_n = 0;
  d.kt = 64;
  d.ksteps = a.K / 64;
  d.ksteps_per_split = cdiv(d.ksteps, sk);
  // (round 5: every workgroup starting its K walk (t * 5 | 13 | 29) % steps into its slice, against HBM-channel camping of 256 lockstep streams
  // whose bases are K * 128 B apart: OPT stage 5.18 -> 5.24 ms, profiles/r05_opt_stream64.md.  Not that; removed.)
  // eight waves of 32 x 32 (three LDS-DMA pieces per wave and stage) rather than four of 64 x 32 (six): OPT stage 5.03 -> 4.70 ms — a wave issues its
  // pieces, fragment reads and MFMAs one after the other, and with one workgroup per CU only more waves overlap them
  d.nwv = 8; d.mi = 2;
  const dim3 grid(d.tiles_n * d.tiles_m, sk, 1);
  if (sk > 1) GILL_TRY((gemm_launch_stages<64, 0, 2>(d, grid, s)));
  else if (a.out_mode == OUT_QKV) GILL_TRY((gemm_launch_stages<64, 0, 3>(d, grid, s)));
  else GILL_TRY((gemm_launch_stages<64, 0, 0>(d, grid, s)));
  if (sk > 1 && !a.partials_only) return gemm_splitk_reduce_launch(d.a, s);
  return 0;
}

int gemm_launch(const GemmArgs& a, hipStream_t s) {
  GILL_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "empty GEMM");
  GILL_REQUIRE((int64_t)a.M * (a.lda > a.N ? a.lda : a.N) < (int64_t)1 << 31, "GEMM operand too large for 32-bit offsets");
  GILL_REQUIRE((int64_t)a.M * (a.ldc > a.ldr ? a.ldc : a.ldr) < (int64_t)1 << 31, "GEMM output / residual too large for 32-bit offsets");
  GILL_REQUIRE(a.K % BK == 0, "K must be a multiple of 64");
  GILL_REQUIRE(a.N % 4 == 0, "N must be a multiple of 4");
  GILL_REQUIRE(a.A != nullptr && a.W != nullptr, "null operand");
  if (a.conv) {
    if (a.ups == 2) {
      GILL_REQUIRE(a.Cin % BK == 0 && a.K == 4 * a.Cin && a.KX == 0, "upsample conv (4-tap form): Cin must be a multiple of 64 and K == 4*Cin");
      GILL_REQUIRE(!a.k_chunked && !a.resid && !a.rowvec && a.M % 4 == 0 && a.OH == 2 * a.IH && a.OW == 2 * a.IW,
                   "upsample conv (4-tap form): tap-major weights, no residual / row vector, output = 2x the source grid");
      GILL_REQUIRE(!a.gn_stats || a.rows_per_batch % 256 == 0, "upsample conv (4-tap form): fused GroupNorm statistics need whole 64-row slabs per class");
    } else {
      GILL_REQUIRE(a.Cin % BK == 0 && a.K == 9 * a.Cin + a.KX, "conv: Cin must be a multiple of 64 and K == 9*Cin + KX");
    }
    if (a.KX) {
      GILL_REQUIRE(a.stride == 1 && !a.ups && a.X1 != nullptr, "conv: the fused 1x1 segment needs stride 1, no upsample");
      GILL_REQUIRE(a.KX % BK == 0 && a.KX1 % BK == 0 && a.KX1 <= a.KX && (a.KX1 == a.KX || a.X2 != nullptr),
                   "conv: fused 1x1 segment channel counts must be multiples of 64");
    }
    GILL_REQUIRE(a.K1 % BK == 0 && a.K1 <= a.Cin, "conv: source split must be a multiple of 64");
    GILL_REQUIRE(a.Cin / BK <= ZERO_PAGE_STEPS, "conv: Cin too large for the zero page");
    GILL_REQUIRE(a.K1 == a.Cin || a.A2 != nullptr, "conv: second source missing");
    GILL_REQUIRE(!(a.ups && a.stride != 1), "conv: upsample needs stride 1");
    GILL_REQUIRE(a.out_mode == OUT_BF16 && a.act == ACT_NONE && !a.resid_f32 && !a.row_stats,
                 "conv: bf16 row-major epilogue without activation / fp32 residual / row statistics only");
    GILL_REQUIRE((int64_t)(a.M / (a.OH * a.OW)) * a.IH * a.IW * a.Cin < (int64_t)1 << 31, "conv input too large for 32-bit offsets");
  } else {
    GILL_REQUIRE(a.K1 % BK == 0 && a.K1 <= a.K, "K split must be a multiple of 64");
    GILL_REQUIRE(a.K1 == a.K || a.A2 != nullptr, "second A source missing");
  }
  if (a.gn_stats) {
    GILL_REQUIRE(a.out_mode != OUT_QKV && a.act != ACT_GEGLU, "fused GroupNorm statistics need the row-major epilogue");
    GILL_REQUIRE(a.rows_per_batch % 64 == 0 && a.gn_groups > 0 && a.gn_groups <= 64 && a.gn_cg * a.gn_groups == a.N,
                 "fused GroupNorm statistics: rows per sample must be a multiple of 64 and bins must tile N");
    GILL_REQUIRE(gemm_fused_gn_ok(a.N, a.gn_cg) && tile_width(a) % a.gn_cg == 0,
                 "fused GroupNorm statistics: bins must not straddle output tiles");
  }
  if (a.row_stats)
    GILL_REQUIRE(a.out_mode == OUT_BF16 && a.act != ACT_GEGLU, "row statistics need the bf16 row-major epilogue");
  if (a.ln_stats) {
    GILL_REQUIRE(a.ln_planes >= 1 && a.ln_planes <= LN_MAX_PLANES, "folded LayerNorm: 1 <= ln_planes <= 20");
    GILL_REQUIRE(a.ln_colsum != nullptr && !a.conv && a.K1 == a.K, "folded LayerNorm: column sums missing / single-source plain GEMM only");
    GILL_REQUIRE(a.act == ACT_GEGLU || a.out_mode == OUT_QKV || a.out_mode == OUT_SOFTMAX80, "folded LayerNorm is implemented in the GEGLU, QKV and softmax epilogues");
    GILL_REQUIRE(a.alpha == 1.f, "folded LayerNorm: alpha must be 1");
  }
  GILL_REQUIRE((a.fn_Y == nullptr && a.fn_ss == nullptr) || gemm_coop_ok(a) || (a.fn_ss == nullptr && a.splitk > 1 && gemm_fused_norm_ok(a)),
               "fused GroupNorm output: split-K GEMMs of a supported geometry, or an in-kernel finish (gemm_coop_ok), only");
  GILL_REQUIRE(a.C != nullptr || a.fn_Y != nullptr || a.out_mode == OUT_QKV, "no output tensor");
  if (a.splitk > 1) {
    GILL_REQUIRE(a.ws != nullptr, "split-K workspace missing");
    GILL_REQUIRE(a.act != ACT_GEGLU, "split-K cannot be combined with GEGLU");
  }
  if (a.act == ACT_GEGLU) {
    GILL_REQUIRE(a.N % 128 == 0 && a.out_mode == OUT_BF16, "GEGLU needs N % 128 == 0 and bf16 row-major output");
    return gemm_launch_bn<128>(a, s);
  }
  if (a.out_mode == OUT_SOFTMAX80)
    GILL_REQUIRE(a.N % 160 == 0 && a.splitk <= 1 && a.ln_stats && a.ln_colsum && a.bias && !a.conv && !a.gn_stats && !a.row_stats && a.C,
                 "softmax epilogue: N % 160 == 0, no split-K, folded-LayerNorm operands and a bias vector required");
  if (a.out_mode == OUT_QKV) GILL_REQUIRE(a.dp % 8 == 0 && a.heads > 0 && a.ntok > 0, "bad QKV scatter geometry (padded head dim must be a multiple of 8)");
  if (a.w_blk64) {
    GILL_REQUIRE(!a.conv && a.N % 64 == 0 && a.K1 == a.K && !a.gn_stats && !a.row_stats && !a.ln_stats && !a.wb_rows && !a.fn_Y &&
                     a.act != ACT_GEGLU && a.out_mode != OUT_SOFTMAX80,
                 "64 x 64-blocked weights (STREAM64): plain GEMMs with the row-major, QKV or split-K epilogue only");
    if (a.M <= GEMM_STREAM64_MAX_ROWS) return gemm_launch_stream64(a, s);
    // more rows than that: the general tiles read the blocked layout (the 128 x 64 tile is CU-bound per M tile: profiles/r05_opt_stream64.md)
  }
  const int bn = tile_width(a);
  if (bn == 160) return gemm_launch_bn<160>(a, s);
  return gemm_launch_bn<128>(a, s);
}
