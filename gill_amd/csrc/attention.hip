// Flash-style (online-softmax) attention on v_mfma_f32_32x32x16_bf16 for gfx950.
//
// Operands are head-major, produced by the QKV GEMM epilogue (gemm.hip, OUT_QKV):
//   Q  [B][H][nq_pad ][DP]      K [B][H][nkv_pad][DP]      Vt [B][H][dpv][nkv_pad]
// Output O is token-major [B*nq][ldo] so the out-projection GEMM consumes it directly.
//
// Workgroup = 8 waves = 256 queries of one (batch, head); each wave owns 32 query rows.  K/V are walked in tiles
// of 64 keys that all 8 waves share through LDS (coalesced 16-B global loads -> registers -> padded LDS rows,
// register-staged double buffer: the next tile's global loads are in flight while the current one is consumed;
// one barrier per tile).  Row strides are padded (+16 B for K and Vt rows: an odd number of 16-B units) so every
// ds_read_b128 lane group hits distinct banks.
//
// Everything is computed transposed so each lane owns ONE query column of the 32x32 tiles:
//   S^T[kv][q] = sum_d K[kv][d] Q[q][d]          (A = K fragment, B = Q fragment held in registers)
//   O^T[d][q]  = sum_kv Vt[d][kv] P[q][kv]       (A = Vt fragment, B = P in registers)
// so the running max / sum / rescale are lane-local (one xor-32 shuffle joins the two half waves that share a
// query).  P never leaves registers: the S^T accumulator registers r=0..7 / 8..15 of a lane are, as they are, the
// k-slots of the B operand of the PV MFMAs, provided the Vt fragment is gathered with the same kv permutation
//   kv(h, hi, j) = 16*h + 8*(j>>2) + 4*hi + (j&3)
// — and the Vt tile is stored in LDS in exactly that order (within every 16 keys the 4-key blocks sit as [0-3][8-11][4-7][12-15]:
// the staging store splits each 16-B chunk into its two 8-B halves anyway), so a fragment is ONE 16-B LDS read (256 B/clk; the two
// 8-B reads it replaces ran at half that, and at d = 40 the LDS pipe was as busy as the matrix pipe).
//
// Softmax cost (the bound at d = 40: one exp per 80 MACs): Q is pre-scaled by scale*log2(e) and the S accumulator is
// initialised to -m_run, so p = exp2(mfma result); the running max only moves when a row grows past m_run + 8 (rare,
// wave-uniform slow path), and for DP % 32 != 0 the row sum rides through the PV MFMAs on a ones-row of Vt.
//
// DP (padded head dim, multiple of 16): 48 (d=40), 64, 80, 128, 160.  Padding columns of Q/K are exact zeros
// (zero weight rows); padding rows of Vt (up to dpv = roundup(DP,32)) only feed output rows that are never stored.
#include "ops.h"
#include <stdlib.h>
#include <type_traits>

// threads per workgroup NTHR = 512 | 256 | 128 (template): one wave per 32 queries, NTHR / 2 queries per workgroup
#define ATT_KVT 64     // keys per tile

template <int DP, int ATT_THREADS>
struct AttCfg {
  static constexpr int KS = DP / 16;              // QK^T k-steps per 32-key half
  static constexpr int NDT = (DP + 31) / 32;      // 32-row d tiles of O^T
  static constexpr int DPV = NDT * 32;
  static constexpr int KSTR = DP + 8;             // K row stride in elements (+16 B)
  static constexpr int VSTR = ATT_KVT + 8;        // Vt row stride in elements (+16 B)
  static constexpr bool HAS_ONES = (DP % 32) != 0; // a spare Vt row (index DP) of ones carries the softmax row sum
  static constexpr int K_CHUNKS = ATT_KVT * DP / 8;       // 16-B chunks of a K tile
  static constexpr int V_CHUNKS = DPV * ATT_KVT / 8;      // 16-B chunks of a Vt tile
  static constexpr int K_PER_THR = (K_CHUNKS + ATT_THREADS - 1) / ATT_THREADS;
  static constexpr int V_PER_THR = (V_CHUNKS + ATT_THREADS - 1) / ATT_THREADS;
  // every thread loads AND stores K_PER_THR / V_PER_THR chunks unconditionally (branch-free staging keeps the
  // prefetch registers out of scratch); the LDS regions are over-allocated to absorb the surplus chunks
  static constexpr int K_ROWS = (K_PER_THR * ATT_THREADS + DP / 8 - 1) / (DP / 8);
  static constexpr int V_ROWS = K_PER_THR * 0 + V_PER_THR * ATT_THREADS / 8;
  static constexpr int K_ELEMS = K_ROWS * KSTR;
  static constexpr int V_ELEMS = V_ROWS * VSTR;
  static constexpr int BUF_ELEMS = K_ELEMS + V_ELEMS;
  static constexpr int SMEM_BYTES = 2 * BUF_ELEMS * 2;
};

// (A one-dim form of attention_dma_kernel's QF3 lived here in round 3 — q[40] := -m_run as ONE bf16 value: its offset loses the low bits at
// |score| >= 2^15, ADVICE r03.  The d = 40 shapes that matter run on the LDS-DMA kernel below; this kernel keeps the exact fp32 offset.)
template <int DP, int ATT_THREADS>
__global__ __launch_bounds__(ATT_THREADS, (DP <= 64 ? 4 : 2)) void attention_kernel(const AttnArgs p) {
  kernarg_warm<sizeof(AttnArgs)>();
  using Cfg = AttCfg<DP, ATT_THREADS>;
  constexpr int ATT_QB = ATT_THREADS / 2;
  constexpr int KS = Cfg::KS, NDT = Cfg::NDT, KSTR = Cfg::KSTR, VSTR = Cfg::VSTR;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(att_smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  // XCD-aware workgroup map: the dispatcher places workgroup L on XCD L % 8, and every query tile of one (sample, head)
  // streams that pair's whole K / V^T.  Pairs are dealt to XCDs round robin and all query tiles of a pair stay on the pair's
  // XCD, so K / V^T are fetched into ONE L2 instead of eight (level-0 self-attention: 484 MB -> ~130 MB of fabric reads).
  const int nqt = (p.nq + ATT_QB - 1) / ATT_QB;
  int pair, qtile;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = xcd + 8 * (slot / nqt);
    qtile = slot - (slot / nqt) * nqt;
  } else {
    pair = blockIdx.x / nqt;
    qtile = blockIdx.x - pair * nqt;
  }
  if (pair >= p.B * p.H) return;
  const int b = pair / p.H, h = pair - b * p.H;
  const int q0 = qtile * ATT_QB + w * 32;
  const bool wave_active = q0 < p.nq;
  const int lq = lane & 31;
  const int hi = lane >> 5;
  const int kvb = p.kv_bstride_zero ? 0 : b;

  const bf16_t* Qb = p.Q + (size_t)(b * p.H + h) * p.nq_pad * DP;
  const bf16_t* Kb = p.K + (size_t)(kvb * p.H + h) * p.nkv_pad * DP;
  const bf16_t* Vb = p.Vt + (size_t)(kvb * p.H + h) * p.dpv * p.nkv_pad;

  int qrow = q0 + lq;
  const bool qvalid = qrow < p.nq;
  if (qrow > p.nq - 1) qrow = p.nq - 1;

  bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * DP + s * 16 + hi * 8);

  f32x16 oacc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;

  // scores live in the log2 domain: Q was multiplied by scale * log2(e) by its producer (AttnArgs::scale is informative)
  float m_run = 0.f, l_run = 0.f;

  // causal: query i attends kv <= i + (nkv - nq).  The tile loop bound is uniform per workgroup.
  const int coff = p.nkv - p.nq;
  int kv_end_blk = p.nkv;
  if (p.causal) {
    int lim = qtile * ATT_QB + ATT_QB - 1 + coff + 1;
    if (lim < kv_end_blk) kv_end_blk = lim;
    if (kv_end_blk < 1) kv_end_blk = 1;
  }
  const int ntiles = (kv_end_blk + ATT_KVT - 1) / ATT_KVT;
  const int qidx = qrow + coff;
  const int wave_kv_end = p.causal ? min(p.nkv, q0 + 31 + coff + 1) : p.nkv;   // keys this wave can see at all

  // ---- tile staging (all 512 threads): global -> registers -> LDS
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's uint4 struct did not)
  u32x4 kreg[Cfg::K_PER_THR];
  u32x4 vreg[Cfg::V_PER_THR];
  auto gload = [&](int tile) {
    const int kv0 = tile * ATT_KVT;
#pragma unroll
    for (int i = 0; i < Cfg::K_PER_THR; ++i) {
      int c = tid + i * ATT_THREADS;
      if (c > Cfg::K_CHUNKS - 1) c = Cfg::K_CHUNKS - 1;   // surplus threads re-load the last chunk (never stored)
      // K tile rows are contiguous in global memory: 64 rows * DP elements
      int row = c / (DP / 8);
      const int col = c - row * (DP / 8);
      if (kv0 + row > p.nkv_pad - 1) row = p.nkv_pad - 1 - kv0;   // stay inside the allocation (masked anyway)
      kreg[i] = *reinterpret_cast<const u32x4*>(Kb + (size_t)(kv0 + row) * DP + col * 8);
    }
#pragma unroll
    for (int i = 0; i < Cfg::V_PER_THR; ++i) {
      int c = tid + i * ATT_THREADS;
      if (c > Cfg::V_CHUNKS - 1) c = Cfg::V_CHUNKS - 1;
      const int row = c >> 3;
      int col = (c & 7) * 8;
      if (kv0 + col > p.nkv_pad - 8) col = p.nkv_pad - 8 - kv0;   // nkv_pad is a multiple of 32, tiles are 64 wide
      vreg[i] = *reinterpret_cast<const u32x4*>(Vb + (size_t)row * p.nkv_pad + kv0 + col);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* Ks = smem + buf * Cfg::BUF_ELEMS;
    bf16_t* Vs = Ks + Cfg::K_ELEMS;
#pragma unroll
    for (int i = 0; i < Cfg::K_PER_THR; ++i) {
      const int c = tid + i * ATT_THREADS;
      const int row = c / (DP / 8);
      const int col = c - row * (DP / 8);
      *reinterpret_cast<u32x4*>(Ks + row * KSTR + col * 8) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < Cfg::V_PER_THR; ++i) {
      const int c = tid + i * ATT_THREADS;
      const int row = c >> 3;
      const int col = (c & 7) * 8;
      // keys col .. col+7 -> their fragment positions (see the header): [0-3] -> 0, [4-7] -> 8, [8-11] -> 4, [12-15] -> 12 of the 16-key group
      bf16_t* d = Vs + row * VSTR + (col & ~15) + ((col & 8) >> 1);
      *reinterpret_cast<uint2*>(d) = make_uint2(vreg[i][0], vreg[i][1]);
      *reinterpret_cast<uint2*>(d + 8) = make_uint2(vreg[i][2], vreg[i][3]);
    }
  };

  // The query tiles of one (sample, head) run side by side on one XCD and all stream the same K / V^T: each starts its walk
  // over the key tiles at a different tile (rotation; the online softmax does not care about the order, and the order of a
  // given query tile is fixed, so results stay bit-reproducible), so that they do not all ask the L2 for the same lines at
  // the same moment.  Causal attention keeps the natural order (its tile bound depends on the query tile).
  const int rot = (p.causal || !p.xcd_map) ? 0 : (qtile * 5) % ntiles;
  gload(rot);
  lstore(0);
  __syncthreads();

  for (int it = 0; it < ntiles; ++it) {
    int tile = it + rot;
    if (tile >= ntiles) tile -= ntiles;
    const int kv0 = tile * ATT_KVT;
    if (it + 1 < ntiles) gload(tile + 1 < ntiles ? tile + 1 : 0);
    const bf16_t* Ks = smem + (it & 1) * Cfg::BUF_ELEMS;
    const bf16_t* Vs = Ks + Cfg::K_ELEMS;

    if (wave_active && kv0 < wave_kv_end) {
      // ---- S^T for the two 32-key halves.  Q arrives pre-multiplied by scale*log2(e), and the accumulator starts at
      // -m_run, so the MFMA result IS the exponent s - m_run: no per-element scale / subtract VALU work.
      const bool first = (it == 0);
      const float acc0 = first ? 0.f : -m_run;
      f32x16 s[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[hh][r] = acc0;
        const bf16_t* krow = Ks + (hh * 32 + lq) * KSTR + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + ks * 16);
          s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[hh], 0, 0, 0);
        }
      }
      // ---- masking only where the tile is not entirely visible (wave-uniform test)
      const bool need_mask = (kv0 + ATT_KVT > p.nkv) || (p.causal && (kv0 + ATT_KVT - 1 > q0 + coff));
      if (need_mask) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool masked = (kv >= p.nkv) || (p.causal && kv > qidx);
            s[hh][r] = masked ? -INFINITY : s[hh][r];
          }
      }
      float mx = s[0][0];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[hh][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // Lazy running max: m_run moves only on the first tile or when a row's scores exceed it by > 8 (p <= 2^8: exact
      // in the bf16 exponent range, fp32 accumulators have the headroom).  The rescale of O (and of the row sum, which
      // rides in O as the ones-row of Vt) is then a rare wave-uniform slow path instead of per-tile work.
      // Masked entries are -inf (exp2 -> exactly 0).
      const bool grow = first || (mx > 8.0f);
      if (__any(grow)) {
        float delta;
        if (first) { m_run = (mx > -INFINITY) ? mx : 0.f; delta = m_run; }   // (a valid row always sees key 0)
        else { delta = fmaxf(mx, 0.f); m_run += delta; }                    // fully masked row: mx = -inf -> 0
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[hh][r] -= delta;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[hh][r] = __builtin_amdgcn_exp2f(s[hh][r]);
      if constexpr (!Cfg::HAS_ONES) {
        // no spare Vt row to carry the row sum (DP % 32 == 0): explicit sum
        float rs = 0.f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 16; ++r) rs += s[hh][r];
        rs += __shfl_xor(rs, 32, 64);
        l_run += rs;
      }

      bf16x8 pa[4];
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) {
        union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf2(s[h4 >> 1][(h4 & 1) * 8 + 2 * j], s[h4 >> 1][(h4 & 1) * 8 + 2 * j + 1]);
        pa[h4] = pk.v;
      }
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const bf16_t* vrow = Vs + (t * 32 + lq) * VSTR + 8 * hi;
#pragma unroll
        for (int h4 = 0; h4 < 4; ++h4) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + 16 * h4);
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pa[h4], oacc[t], 0, 0, 0);
        }
      }
    }
    if (it + 1 < ntiles) lstore((it + 1) & 1);
    __syncthreads();
  }

  if (!wave_active) return;
  if constexpr (Cfg::HAS_ONES) {
    // Vt row DP is all ones (written by the QKV epilogue), so O^T row DP accumulated sum_kv P: it sits in register 8
    // of tile DP/32 in the hi = 0 half wave (d = 32 t + 8 g + 4 hi + j with DP % 32 == 16 -> g = 2, hi = 0, j = 0)
    static_assert(DP % 32 == 16, "ones-row position");
    l_run = __shfl(oacc[DP / 32][8], lq, 64);
  }
  if (!qvalid) return;
  const float inv = 1.f / l_run;
  bf16_t* orow = p.O + (size_t)(b * p.nq + qrow) * p.ldo + h * DP;
#pragma unroll
  for (int t = 0; t < NDT; ++t) {
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


// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for d = 40 (DP = 48), non-causal, nq % (NTHR / 2) == 0, nkv_pad % 64 == 0: the level-0 attention of the UNet.
//
// The register-staged kernel above spends ~175 wave instructions per 64-key tile, of which only 14 are MFMAs: at d = 40 the
// SIMD's VALU issue port — not the matrix pipe (39.9 % busy) — is what a tile waits for (32 exp + 16 packs + ~27 for the running max incl.
// a cross-half ds_bpermute + ~30 for staging addresses, the staging stores and the loop).  This variant removes everything that is not
// softmax arithmetic from the VALU stream:
//   * K and V^T tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction): no staging registers, no
//     ds_write, no per-tile address arithmetic (uniform base per tile + one fixed 32-bit lane offset per piece).  Both tiles are
//     [64 rows][128 B] images (K rows padded from 96 to 128 B: the lanes of the two pad chunks re-fetch chunks 0 / 1 of the same row —
//     the same cache line — into slots nobody reads), XOR-swizzled on the SOURCE address and the ds_read address
//     (chunk ^= (row >> 1) & 7: conflict-free for the 16-lane groups of ds_read_b128 at a 128-B row stride).
//   * The K rows are DEALT to LDS rows with bits 2 and 3 of the row index swapped (the DMA source address is per lane, so any row
//     order is free).  The S^T accumulator registers 8 a .. 8 a + 7 of a lane are then 8 CONSECUTIVE keys (16 a + 8 hi + j), i.e. the
//     lane's P fragment pairs with ONE natural 16-byte chunk of the V^T row: no permuted V^T image (the register-staged kernel
//     builds one with split 8-byte stores).
//   * QF3: the running max rides in THREE padding dims, q[40..42] := the exact three-way bf16 split of -m_run (8 + 8 + 8 mantissa
//     bits = fp32), against k[40..42] = 1 — so S accumulates on an inline-constant zero AND the offset is exact at any magnitude (the
//     one-dim QF above carries a bf16-rounded offset).  The ones never come from memory: dims 40..47 of K are the
//     same for every key, so the half wave that holds them (hi = 1, k step 2) reads its fragment from ONE constant 16-byte LDS slot
//     (its lane address points there for every row; broadcast read).
//   * The lazy running max is decided on the lane's OWN 32 scores (wave-uniform `any`); the two half waves that share a query only
//     exchange their maxima in the rare slow path.
//   * The tile loop is unrolled by two so that the ring stage is a compile-time constant: every LDS address is a loop-invariant lane
//     offset + an immediate.
// Everything else (transposed S^T / O^T, P in registers, ones-row of V^T carrying the row sum, XCD map, rotated tile walk) is as above.
// Measured (8 x 8 heads x 4096^2, rocprofv3): 241-246 us against 290 us for the register-staged kernel; loop 479.8 -> 473.6 ms.
// Two software-pipelined forms (scores of TWO tiles live: QK^T of tile t + 1 beside the softmax of tile t) were built, parity-green,
// and measured no faster (profiles/r04_attention.md): left to hipcc's own order (which clusters the six QK^T MFMAs) 260-273 us and
// loop +0.8 %; with a sched_group_barrier order "1 MFMA, 1 LDS read, 7 VALU" over the whole tile (148 VGPRs) loop +-0.3 %.  Removed.
template <int DP, int ATT_THREADS>
__global__ __launch_bounds__(ATT_THREADS, 4) void attention_dma_kernel(const AttnArgs p) {
  kernarg_warm<sizeof(AttnArgs)>();
  static_assert(DP == 48, "QF3 needs three padding dims behind d = 40");
  constexpr int NW = ATT_THREADS / 64;            // waves per workgroup
  constexpr int ATT_QB = ATT_THREADS / 2;         // queries per workgroup
  constexpr int KS = DP / 16, NDT = (DP + 31) / 32, NCH = DP / 8;
  constexpr int PPW = 8 / NW;                     // K pieces (and V^T pieces) per wave and tile: 8 pieces of 8 rows each
  constexpr int K_BYTES = 64 * 128, STAGE = 2 * K_BYTES, CONST_OFF = 2 * STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem_raw[];
  unsigned char* smem = att_smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqt = p.nq / ATT_QB;
  int pair, qtile;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = xcd + 8 * (slot / nqt);
    qtile = slot - (slot / nqt) * nqt;
  } else {
    pair = blockIdx.x / nqt;
    qtile = blockIdx.x - pair * nqt;
  }
  if (pair >= p.B * p.H) return;
  const int b = pair / p.H, h = pair - b * p.H;
  const int q0 = qtile * ATT_QB + w * 32;
  const int lq = lane & 31;
  const int hi = lane >> 5;
  const int kvb = p.kv_bstride_zero ? 0 : b;

  const bf16_t* Qb = p.Q + (size_t)(b * p.H + h) * p.nq_pad * DP;
  const bf16_t* Kb = p.K + (size_t)(kvb * p.H + h) * p.nkv_pad * DP;
  const bf16_t* Vb = p.Vt + (size_t)(kvb * p.H + h) * p.dpv * p.nkv_pad;
  const int qrow = q0 + lq;          // (nq % ATT_QB == 0: always a valid row)

  // k[40..42] = 1, k[43..47] = 0 for every key: one constant fragment (visible after the first barrier below)
  if (tid == 0) *reinterpret_cast<uint4*>(smem + CONST_OFF) = make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u);

  bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * DP + s * 16 + hi * 8);

  // ---- LDS-DMA geometry: piece pc (0..7) of a tile image covers its rows 8 pc .. 8 pc + 7; wave w issues pieces w * PPW + j
  const int r8 = lane >> 3, pch = lane & 7;
  unsigned koff[PPW], voff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int i = (w * PPW + j) * 8 + r8;                       // LDS row of this lane's 16 bytes
    const int c = pch ^ ((i >> 1) & 7);                         // logical chunk held at physical position pch of row i
    const int key = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);   // K: LDS row i holds key swap23(i) of the tile
    const int ck = c >= NCH ? c - NCH : c;                      // (pad chunks: re-fetch chunk 0 / 1 — never read)
    koff[j] = (unsigned)((key * DP + ck * 8) * 2);
    voff[j] = (unsigned)((i * p.nkv_pad + c * 8) * 2);
  }
  auto issue_tile = [&](int tile, int stage) {
    const char* kbase = (const char*)(Kb + (size_t)tile * ATT_KVT * DP);     // (uniform)
    const char* vbase = (const char*)(Vb + (size_t)tile * ATT_KVT);
    unsigned char* dst = smem + stage * STAGE + (w * PPW) * 1024;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + koff[j]),
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + voff[j]),
                                       (__attribute__((address_space(3))) void*)(dst + K_BYTES + j * 1024), 16, 0, 0);
    }
  };
  // fragment reads: row (32 hh | 32 t) + lq, logical chunk 2 c4 + hi -> physical (2 c4 + hi) ^ ((lq >> 1) & 7): one lane offset per c4
  int fo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) fo[c4] = lq * 128 + (((2 * c4 + hi) ^ ((lq >> 1) & 7)) * 16);
  // k step 2 of QK^T: dims 32..39 (hi = 0) from the tile, dims 40..47 (hi = 1) from the constant slot — an address per (stage, half),
  // because the immediates of the other reads do not apply to the constant
  int k2a[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) k2a[st][hh] = hi ? CONST_OFF : st * STAGE + hh * 4096 + fo[2];

  f32x16 oacc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = 0.f;       // (q[40..42] = 0 so far)

  const int ntiles = (p.nkv + ATT_KVT - 1) / ATT_KVT;
  const int rot = p.xcd_map ? (qtile * 5) % ntiles : 0;
  issue_tile(rot, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one tile; STG (the ring stage) is a compile-time constant: the loop below is unrolled by two
  auto tile_body = [&](int it, auto stg) {
    constexpr int STG = decltype(stg)::value;
    int tile = it + rot;
    if (tile >= ntiles) tile -= ntiles;
    const int kv0 = tile * ATT_KVT;
    // the other stage was last read in iteration it - 1, and every wave has passed the barrier that closed it
    if (it + 1 < ntiles) issue_tile(tile + 1 < ntiles ? tile + 1 : 0, STG ^ 1);
    const unsigned char* Ks = smem + STG * STAGE;
    const unsigned char* Vs = Ks + K_BYTES;

    f32x16 s[2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const bf16x8 kf = (ks == 2) ? *reinterpret_cast<const bf16x8*>(smem + k2a[STG][hh])
                                    : *reinterpret_cast<const bf16x8*>(Ks + hh * 4096 + fo[ks]);
        if (ks == 0) s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
        else s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[hh], 0, 0, 0);
      }
    }
    // register r of s[hh] is key kv0 + 32 hh + 16 (r >> 3) + 8 hi + (r & 7) (rows dealt with bits 2 / 3 swapped, see the header)
    if (kv0 + ATT_KVT > p.nkv) {      // ragged last tile (wave-uniform)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + hh * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
          s[hh][r] = (kv >= p.nkv) ? -INFINITY : s[hh][r];
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[hh][r]);
    // Lazy running max (see the kernel above): decided on the lane's own 32 scores; only the slow path joins the two half waves
    const bool first = (it == 0);
    if (first || __any(mx > 8.0f)) {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float delta;
      if (first) delta = (mx > -INFINITY) ? mx : 0.f;      // (m_run = 0 so far, and O = 0: the common update below is exact)
      else delta = fmaxf(mx, 0.f);                          // fully masked row: mx = -inf -> 0
      m_run += delta;
      {   // q[40..42] := -m_run as an exact sum of three bf16 values (lane (q, hi = 1) holds dims 40 .. 47 of its query)
        const float x = -m_run;
        const bf16_t a0 = f2bf(x);
        const float r1 = x - bf2f(a0);
        const bf16_t a1 = f2bf(r1);
        const float r2 = r1 - bf2f(a1);
        const bf16_t a2 = f2bf(r2);
        union { bf16x8 v; uint32_t u[4]; } xq;
        xq.v = qf[2];
        if (hi) { xq.u[0] = (uint32_t)a0 | ((uint32_t)a1 << 16); xq.u[1] = (uint32_t)a2; }
        qf[2] = xq.v;
      }
      if (!first) {          // (first tile: O = 0, and exp2(-delta) may be inf)
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[hh][r] -= delta;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[hh][r] = __builtin_amdgcn_exp2f(s[hh][r]);
    bf16x8 pa[4];
#pragma unroll
    for (int h4 = 0; h4 < 4; ++h4) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf2(s[h4 >> 1][(h4 & 1) * 8 + 2 * j], s[h4 >> 1][(h4 & 1) * 8 + 2 * j + 1]);
      pa[h4] = pk.v;
    }
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + t * 4096 + fo[h4]);
        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pa[h4], oacc[t], 0, 0, 0);
      }
    }
    // own pieces of the next tile have landed, AND every LDS read of this tile has returned: hipcc sinks the last PV fragment read's
    // lgkmcnt wait (and its MFMA) below the barrier, i.e. the read could still sit in the LDS queue when another wave, released by
    // the barrier, refills this stage — with eight processes sharing the GPU (test_bench_eight_ranks_uneven_shards_gloo) that
    // window was hit: steps differed by 1.4e-3.  The explicit lgkmcnt(0) closes it (tools/sessions/r04_b8.sh is the bisect).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                       // (the next tile's LDS reads stay behind the barrier)
  };
  for (int it = 0; it < ntiles; it += 2) {
    tile_body(it, std::integral_constant<int, 0>{});
    if (it + 1 < ntiles) tile_body(it + 1, std::integral_constant<int, 1>{});
  }

  // V^T row DP is all ones (written by the QKV epilogue), so O^T row DP accumulated sum_kv P: register 8 of tile DP / 32, hi = 0 half
  static_assert(DP % 32 == 16, "ones-row position");
  const float l_run = __shfl(oacc[DP / 32][8], lq, 64);
  const float inv = 1.f / l_run;
  bf16_t* orow = p.O + (size_t)(b * p.nq + qrow) * p.ldo + h * DP;
#pragma unroll
  for (int t = 0; t < NDT; ++t) {
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

template <int DP, int NTHR>
static int attention_dma_launch(const AttnArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = 2 * 2 * 64 * 128 + 64;     // two stages of (K | V^T) images + the constant K fragment
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)attention_dma_kernel<DP, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  constexpr int xcd_on = 1;
  AttnArgs b = a;
  b.xcd_map = xcd_on;
  dim3 grid((xcd_on ? 8 * cdiv(a.H * a.B, 8) : a.H * a.B) * (a.nq / (NTHR / 2)), 1, 1);
  hipLaunchKernelGGL((attention_dma_kernel<DP, NTHR>), grid, dim3(NTHR), smem, s, b);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
// GILL_ATT_DMA=0: the register-staged kernel everywhere (A/B switch)
static bool attention_dma_ok(const AttnArgs& a, int nthr) {
  static const bool on = [] { const char* e = getenv("GILL_ATT_DMA"); return !(e && e[0] == '0'); }();
  return on && a.dp == 48 && a.d == 40 && !a.causal && a.nq % (nthr / 2) == 0 && a.nkv_pad % 64 == 0 && a.dpv >= 64 && a.nkv >= 1;
}

template <int DP, int NTHR>
static int attention_launch_inst(const AttnArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = AttCfg<DP, NTHR>::SMEM_BYTES;
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<DP, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  constexpr int xcd_on = 1;
  AttnArgs b = a;
  b.xcd_map = xcd_on;
  dim3 grid((xcd_on ? 8 * cdiv(a.H * a.B, 8) : a.H * a.B) * cdiv(a.nq, NTHR / 2), 1, 1);   // (XCD, pair slot, query tile): see the kernel's map
  hipLaunchKernelGGL((attention_kernel<DP, NTHR>), grid, dim3(NTHR), smem, s, b);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// 256 queries per workgroup when that still gives every CU a workgroup (measured: level 1, 256 workgroups, is 2 % faster
// this way; level 2, 64 workgroups, 15-22 % faster with 128-query workgroups); shorter query ranges (UNet levels 1-2, OPT, the
// mapper) get 128-query workgroups instead of leaving most CUs idle (each workgroup stages the whole K/V itself,
// which is L2-resident at those sizes)
template <int DP>
static int attention_launch_dp(const AttnArgs& a, hipStream_t s) {
  const int64_t bh = (int64_t)a.H * a.B;
  const bool big = cdiv(a.nq, 256) * bh >= 256;
  if constexpr (DP == 48) {
    if (attention_dma_ok(a, big ? 512 : 256)) return big ? attention_dma_launch<DP, 512>(a, s) : attention_dma_launch<DP, 256>(a, s);
  }
  if (big) return attention_launch_inst<DP, 512>(a, s);
  return attention_launch_inst<DP, 256>(a, s);   // (128-thread workgroups would stage 10+ chunks per thread: spills)
}

int attention_launch(const AttnArgs& a, hipStream_t s) {
  GILL_REQUIRE(a.B > 0 && a.H > 0 && a.nq > 0 && a.nkv > 0, "empty attention");
  GILL_REQUIRE(a.nkv_pad % 32 == 0 && a.nkv_pad >= a.nkv, "nkv_pad must be a multiple of 32 covering nkv");
  GILL_REQUIRE(a.nq_pad >= a.nq, "nq_pad must cover nq");
  GILL_REQUIRE(a.dpv >= round_up(a.dp, 32), "dpv must cover roundup(dp, 32)");
  GILL_REQUIRE(a.ldo >= a.H * a.dp, "ldo too small");
  switch (a.dp) {
    case 48:  return attention_launch_dp<48>(a, s);
    case 64:  return attention_launch_dp<64>(a, s);
    case 80:  return attention_launch_dp<80>(a, s);
    case 128: return attention_launch_dp<128>(a, s);
    case 160: return attention_launch_dp<160>(a, s);
    default:
      gill_set_error("attention: unsupported padded head dim (supported: 48, 64, 80, 128, 160)");
      return -2;
  }
}
