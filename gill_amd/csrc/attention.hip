// Flash-style (online-softmax) attention on v_mfma_f32_32x32x16_bf16 for gfx950.
//
// Operands are head-major, produced by the QKV GEMM epilogue (gemm.hip, OUT_QKV):
//   Q  [B][H][nq_pad ][DP]      K [B][H][nkv_pad][DP]      Vt [B][H][dpv][nkv_pad]
// Output O is token-major [B*nq][ldo] so the out-projection GEMM consumes it directly.
//
// One wave owns 32 query rows; everything is computed transposed so each lane owns ONE query
// column of the 32x32 tiles:
//   S^T[kv][q] = sum_d K[kv][d] Q[q][d]          (A = K fragment, B = Q fragment)
//   O^T[d][q]  = sum_kv Vt[d][kv] P[q][kv]       (A = Vt fragment, B = P in registers)
// so the running max / sum / rescale are lane-local (one xor-32 shuffle joins the two half
// waves that share a query).  P never leaves registers: the S^T accumulator registers r=0..7 /
// 8..15 of a lane are, as they are, the k-slots of the B operand of the two PV MFMAs, provided
// the Vt fragment is gathered with the same kv permutation
//   kv(h2, hi, j) = 16*h2 + 8*(j>>2) + 4*hi + (j&3)
// (two 8-byte loads per fragment from the kv-contiguous Vt rows).
//
// DP (padded head dim, multiple of 16): 48 (d=40), 64, 80, 128, 160.  Padding columns of Q/K are
// exact zeros (zero weight rows), padding rows of Vt (up to dpv = roundup(DP,32)) only feed
// output rows that are never stored.
#include "ops.h"

template <int DP>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16;          // QK^T k-steps
  constexpr int NDT = (DP + 31) / 32;  // 32-row d tiles of O^T
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + w * 32;
  if (q0 >= p.nq) return;
  const int lq = lane & 31;
  const int hi = lane >> 5;
  const int kvb = p.kv_bstride_zero ? 0 : b;

  const bf16_t* Qb = p.Q + (size_t)(b * p.H + h) * p.nq_pad * DP;
  const bf16_t* Kb = p.K + (size_t)(kvb * p.H + h) * p.nkv_pad * DP;
  const bf16_t* Vb = p.Vt + (size_t)(kvb * p.H + h) * p.dpv * p.nkv_pad;

  int qrow = q0 + lq;
  const bool qvalid = qrow < p.nq;
  if (!qvalid) qrow = p.nq - 1;

  bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)qrow * DP + s * 16 + hi * 8);

  f32x16 oacc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;

  const float sl2 = p.scale * 1.4426950408889634f;  // scores are tracked in the log2 domain
  float m_run = -1e30f, l_run = 0.f;

  // causal: query i attends kv <= i + (nkv - nq)
  const int coff = p.nkv - p.nq;
  int kv_end = p.nkv;
  if (p.causal) {
    int lim = q0 + 31 + coff + 1;
    if (lim < kv_end) kv_end = lim;
    if (kv_end < 1) kv_end = 1;
  }
  const int qidx = qrow + coff;

  for (int kv0 = 0; kv0 < kv_end; kv0 += 32) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const bf16_t* krow = Kb + (size_t)(kv0 + lq) * DP + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + ks * 16);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
    }
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = s[r] * sl2;
      const bool masked = (kv >= p.nkv) || (p.causal && kv > qidx);
      v = masked ? -1e30f : v;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[r] - m_new);
      s[r] = e;
      rs += e;
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;

    bf16x8 pa[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf2(s[h2 * 8 + 2 * j], s[h2 * 8 + 2 * j + 1]);
      pa[h2] = pk.v;
    }
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      const bf16_t* vrow = Vb + (size_t)(t * 32 + lq) * p.nkv_pad + kv0 + 4 * hi;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        union { bf16x8 v; uint2 u[2]; } vf;
        vf.u[0] = *reinterpret_cast<const uint2*>(vrow + 16 * h2);
        vf.u[1] = *reinterpret_cast<const uint2*>(vrow + 16 * h2 + 8);
        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pa[h2], oacc[t], 0, 0, 0);
      }
    }
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

int attention_launch(const AttnArgs& a, hipStream_t s) {
  GILL_REQUIRE(a.B > 0 && a.H > 0 && a.nq > 0 && a.nkv > 0, "empty attention");
  GILL_REQUIRE(a.nkv_pad % 32 == 0 && a.nkv_pad >= a.nkv, "nkv_pad must be a multiple of 32 covering nkv");
  GILL_REQUIRE(a.nq_pad >= a.nq, "nq_pad must cover nq");
  GILL_REQUIRE(a.dpv >= round_up(a.dp, 32), "dpv must cover roundup(dp, 32)");
  GILL_REQUIRE(a.ldo >= a.H * a.dp, "ldo too small");
  dim3 grid(cdiv(a.nq, 128), a.H, a.B);
  dim3 block(256);
  switch (a.dp) {
    case 48:  hipLaunchKernelGGL(attention_kernel<48>, grid, block, 0, s, a); break;
    case 64:  hipLaunchKernelGGL(attention_kernel<64>, grid, block, 0, s, a); break;
    case 80:  hipLaunchKernelGGL(attention_kernel<80>, grid, block, 0, s, a); break;
    case 128: hipLaunchKernelGGL(attention_kernel<128>, grid, block, 0, s, a); break;
    case 160: hipLaunchKernelGGL(attention_kernel<160>, grid, block, 0, s, a); break;
    default:
      gill_set_error("attention: unsupported padded head dim (supported: 48, 64, 80, 128, 160)");
      return -2;
  }
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
