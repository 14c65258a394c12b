// GEGLU projection of the UNet's transformer blocks with fp8 (OCP e4m3) activations AND weights on CDNA4's fp8 matrix instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 — the second piece of BASELINE.json configs[4] ("SD-1.5 fp8 weights on CDNA4 fp8 MFMA") after the
// ResnetBlock2D convolutions (conv_fp8.hip): ff.net.0 (GEGLU, 19 % of an image's FLOPs, the largest linear of a block) at levels 1-3.
// Not a reference function: the reference runs the UNet in fp16 (gill/models.py:550-551); opt-in with gill_unet_config.fp8_convs, compared
// against the fp32 oracle and the build's own bf16 path in tests/test_fp8_gpu.py.
//
//   out[m][o] = (x8[m] . Wv8[o] * sv[o] + bv[o]) * gelu(x8[m] . Wg8[o] * sg[o] + bg[o])
//
// x8 = fp8(ACT * LNhat(t)): the block's norm3 applied EXPLICITLY by ln_quant_fp8_kernel ((t - mean) * rstd from the producer's row-sum planes;
// gamma is folded into the weight rows and beta . W^T into the bias at load, as for the bf16 path — but the folded-LayerNorm trick itself,
// rstd (x W'^T - mean colsum), cannot take an fp8 x: it would quantise the un-normalised residual stream and subtract two large numbers).
// Weights: the engine's LayerNorm-folded, value / gate-interleaved rows (16 value rows | 16 gate rows per 16 outputs), quantised per row:
// w8 = fp8(w / s), s = max|w| / 448, colscale = s / ACT.
//
// Structure: conv_fp8.hip's — 128 x 128 tile, four waves 2 x 2, 2-deep LDS ring of 128-byte rows fed by 1-KiB LDS-DMA pieces (a ring row holds
// 128 K elements: half the pieces and LDS reads per FLOP of the bf16 tile at twice the MFMA rate), two workgroups per CU, XOR-swizzled rows,
// operands swapped so that a lane holds 4 consecutive columns.  A value sub-tile (even j) and its gate sub-tile (odd j) are the SAME 16 outputs
// in the same lane: the GEGLU product needs no exchange.
#include "ops.h"

#define L8_BM 128
#define L8_BN 128
#define L8_ROWB 128
#define L8_THREADS 256

typedef __attribute__((ext_vector_type(8))) int l8_i32x8;
typedef __attribute__((ext_vector_type(4))) int l8_i32x4;

struct LinF8Dev {
  LinF8Args a;
  int ksteps, tiles_n;
};

__global__ __launch_bounds__(L8_THREADS, 2) void geglu_fp8_kernel(const LinF8Dev d) {
  constexpr int NT = L8_BN / 32;       // 16-wide N sub-tiles per wave: 4 = two (value, gate) pairs
  constexpr int A_BYTES = L8_BM * L8_ROWB, B_BYTES = L8_BN * L8_ROWB, BUF_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LinF8Args& p = d.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile;
  {   // XCD-aware remap (see gemm.hip): every XCD a contiguous range of tiles — the N tiles of one M range share the activation rows in its L2
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = tile % d.tiles_n, tm = tile / d.tiles_n;
  const int m0 = tm * L8_BM, n0 = tn * L8_BN;

  // staging geometry: instruction i of wave w fills tile rows (i*4+w)*8 .. +8; lane -> (row in group, 16-B chunk); source-side swizzle
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const unsigned char* a_ptr[4];
  const unsigned char* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (i * 4 + w) * 8 + srow;
    if (m > p.M - 1) m = p.M - 1;
    a_ptr[i] = p.A8 + (size_t)m * p.K + schunk * 16;
    int n = n0 + (i * 4 + w) * 8 + srow;
    if (n > p.N - 1) n = p.N - 1;
    w_ptr[i] = p.W8 + (size_t)n * p.K + schunk * 16;
  }
  auto issue = [&](int buf) {
    unsigned char* As = smem + buf * BUF_BYTES;
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[i],
                                       (__attribute__((address_space(3))) void*)(As + (i * 4 + w) * 8 * L8_ROWB), 16, 0, 0);
      a_ptr[i] += L8_ROWB;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                       (__attribute__((address_space(3))) void*)(Bs + (i * 4 + w) * 8 * L8_ROWB), 16, 0, 0);
      w_ptr[i] += L8_ROWB;
    }
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wm = w >> 1, wn = w & 1;
  const int frow = lane & 15, fkb = lane >> 4;      // fragment row, 32-byte k block of the 128-wide step
  issue(0);
  int buf = 0;
  for (int it = 0; it < d.ksteps; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* As = smem + buf * BUF_BYTES;
    const unsigned char* Bs = As + A_BYTES;
    l8_i32x8 af[4], bfr[NT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + frow;
      const l8_i32x4 lo = *reinterpret_cast<const l8_i32x4*>(As + row * L8_ROWB + (((fkb * 2) ^ (row & 7)) * 16));
      const l8_i32x4 hi = *reinterpret_cast<const l8_i32x4*>(As + row * L8_ROWB + (((fkb * 2 + 1) ^ (row & 7)) * 16));
      af[i] = (l8_i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * (L8_BN / 2) + j * 16 + frow;
      const l8_i32x4 lo = *reinterpret_cast<const l8_i32x4*>(Bs + row * L8_ROWB + (((fkb * 2) ^ (row & 7)) * 16));
      const l8_i32x4 hi = *reinterpret_cast<const l8_i32x4*>(Bs + row * L8_ROWB + (((fkb * 2 + 1) ^ (row & 7)) * 16));
      bfr[j] = (l8_i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    if (it + 1 < d.ksteps) issue(buf ^ 1);      // stage the next step while the fragments are in flight from LDS
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)   // operands swapped (W rows as the A operand): a lane holds 4 consecutive N; cbsz = blgp = 0: both e4m3; scales 2^0
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);
    buf ^= 1;
  }

  // ---- epilogue.  acc[i][j][r]: m = m0 + wm*64 + i*16 + frow, physical column n = n0 + wn*64 + j*16 + fkb*4 + r; weight rows come in
  // [16 value | 16 gate] blocks of 16 outputs, so sub-tiles (2 jp, 2 jp + 1) are value and gate of outputs (n0 + wn*64) / 2 + jp*16 + fkb*4 + r
  const int mrow = m0 + wm * 64 + frow;
  const int inner = p.N / 2;
#pragma unroll
  for (int jp = 0; jp < NT / 2; ++jp) {
    const int pv = n0 + wn * (L8_BN / 2) + jp * 32 + fkb * 4;       // physical column of the value quad; the gate quad is 16 further
    if (pv >= p.N) continue;
    const int no = (n0 + wn * (L8_BN / 2)) / 2 + jp * 16 + fkb * 4;  // output column
    const float4 sv = *reinterpret_cast<const float4*>(p.colscale + pv), sg = *reinterpret_cast<const float4*>(p.colscale + pv + 16);
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + pv) : make_float4(0, 0, 0, 0);
    const float4 bg = p.bias ? *reinterpret_cast<const float4*>(p.bias + pv + 16) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mrow + i * 16;
      if (m >= p.M) continue;
      const f32x4 av = acc[i][2 * jp], ag = acc[i][2 * jp + 1];
      const float o0 = (av[0] * sv.x + bv.x) * gelu_erf(ag[0] * sg.x + bg.x), o1 = (av[1] * sv.y + bv.y) * gelu_erf(ag[1] * sg.y + bg.y);
      const float o2 = (av[2] * sv.z + bv.z) * gelu_erf(ag[2] * sg.z + bg.z), o3 = (av[3] * sv.w + bv.w) * gelu_erf(ag[3] * sg.w + bg.w);
      uint2 o; o.x = pack_bf2(o0, o1); o.y = pack_bf2(o2, o3);
      *reinterpret_cast<uint2*>(p.C + (size_t)m * inner + no) = o;
    }
  }
}

int geglu_fp8_launch(const LinF8Args& a, hipStream_t s) {
  GILL_REQUIRE(a.A8 && a.W8 && a.colscale && a.C, "geglu fp8: null operand");
  GILL_REQUIRE(a.M > 0 && a.K % L8_ROWB == 0 && a.N % 32 == 0, "geglu fp8: K must be a multiple of 128, the physical width a multiple of 32");
  GILL_REQUIRE((int64_t)a.M * a.K < (int64_t)1 << 31 && (int64_t)a.M * (a.N / 2) < (int64_t)1 << 31, "geglu fp8: operand too large for 32-bit offsets");
  LinF8Dev d;
  d.a = a;
  d.ksteps = a.K / L8_ROWB;
  d.tiles_n = cdiv(a.N, L8_BN);
  static bool attr_set = false;
  constexpr int smem = 2 * (L8_BM + L8_BN) * L8_ROWB;
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)geglu_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(geglu_fp8_kernel, dim3(cdiv(a.M, L8_BM) * d.tiles_n), dim3(L8_THREADS), smem, s, d);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------- weight quantisation
// bf16 rows [N][K] -> w8[n][K] fp8 e4m3, colscale[n] = max|w_n| / 448 / act_scale (the activation scale folded in: one multiply per output channel
// in the epilogue).  One block per row.
__global__ __launch_bounds__(256) void linear_w_quant_fp8_kernel(const bf16_t* __restrict__ w, int K, float act_scale, unsigned char* __restrict__ w8,
                                                                 float* __restrict__ colscale) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  const bf16_t* src = w + (size_t)n * K;
  float mx = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) mx = fmaxf(mx, fabsf(bf2f(src[i])));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  const float amax = red[0];
  const float sc = amax > 0.f ? amax / 448.f : 1.f;
  const float inv = 1.f / sc;
  if (threadIdx.x == 0) colscale[n] = sc / act_scale;
  unsigned char* dst = w8 + (size_t)n * K;
  for (int k = threadIdx.x * 2; k < K; k += 512) {
    const float v0 = clamp_fp8_keep_nan(bf2f(src[k]) * inv), v1 = clamp_fp8_keep_nan(bf2f(src[k + 1]) * inv);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false);
    *reinterpret_cast<unsigned short*>(dst + k) = (unsigned short)(pk & 0xffff);
  }
}
int linear_weight_quant_fp8_launch(const bf16_t* w, int N, int K, float act_scale, unsigned char* w8, float* colscale, hipStream_t s) {
  GILL_REQUIRE(w && w8 && colscale && K % 2 == 0, "linear fp8: null operand / odd K");
  hipLaunchKernelGGL(linear_w_quant_fp8_kernel, dim3(N), dim3(256), 0, s, w, K, act_scale, w8, colscale);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------- LayerNorm -> fp8
// y8[m][c] = fp8(act_scale * (t[m][c] - mean_m) * rstd_m), mean / rstd of row m from the producer's per-plane {sum, sum of squares} (GemmArgs::row_stats:
// planes added in plane order, exactly as the folded-LayerNorm consumers do — gemm.hip ln_row_factors()).  Rows m >= ln_rows read row m - ln_rows.
// 8 elements per thread; a row's C / 8 threads each total the planes themselves (L1-resident after the first).
__global__ __launch_bounds__(256) void ln_quant_fp8_kernel(const bf16_t* __restrict__ t, int M, int C, const float* __restrict__ ln_stats, int planes,
                                                           int ln_rows, float eps, float act_scale, unsigned char* __restrict__ y) {
  const int cpr = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * cpr) return;
  const int m = (int)(idx / cpr), c = (int)(idx - (int64_t)m * cpr) * 8;
  const int R = ln_rows ? ln_rows : M;
  const int ms = m >= R ? m - R : m;
  float sx = 0.f, sq = 0.f;
  for (int pl = 0; pl < planes; ++pl) {
    const float2 v = *reinterpret_cast<const float2*>(ln_stats + ((size_t)pl * R + ms) * 2);
    sx += v.x; sq += v.y;
  }
  const float invc = 1.f / (float)C;
  const float mean = sx * invc;
  const float var = fmaxf(sq * invc - mean * mean, 0.f);
  const float rs = rsqrtf(var + eps) * act_scale;
  const float off = -mean * rs;
  const uint4 u = *reinterpret_cast<const uint4*>(t + (size_t)m * C + c);
  const unsigned uu[4] = {u.x, u.y, u.z, u.w};
  float o[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[2 * e] = clamp_fp8_keep_nan(fmaf(bf2f((bf16_t)(uu[e] & 0xffff)), rs, off));
    o[2 * e + 1] = clamp_fp8_keep_nan(fmaf(bf2f((bf16_t)(uu[e] >> 16)), rs, off));
  }
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], hi, true);
  *reinterpret_cast<uint2*>(y + (size_t)m * C + c) = make_uint2((unsigned)lo, (unsigned)hi);
}
int ln_quant_fp8_launch(const bf16_t* t, int M, int C, const float* ln_stats, int planes, int ln_rows, float eps, float act_scale, unsigned char* y,
                        hipStream_t s) {
  GILL_REQUIRE(t && ln_stats && y && C % 8 == 0 && planes >= 1, "LayerNorm -> fp8: null operand / C % 8 / no statistics");
  const int64_t n = (int64_t)M * (C / 8);
  hipLaunchKernelGGL(ln_quant_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, t, M, C, ln_stats, planes, ln_rows, eps, act_scale, y);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
