// 3x3 convolution (pad 1, stride 1) with fp8 (OCP e4m3) activations AND weights on CDNA4's fp8 matrix instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 — BASELINE.json configs[4] ("fp8 weights on CDNA4 fp8 MFMA") for the ResnetBlock2D
// convolutions of the SD UNet, which are 52 % of an image's FLOPs.  Not a reference function: the
// reference runs the UNet in fp16 (gill/models.py:550-551); this is the build's opt-in variant (gill_unet_config.fp8_convs),
// compared against the build's own bf16 path in tests/test_fp8_gpu.py.
//
// Same structure as gemm.hip's implicit GEMM (128 x BN tile, 4 waves 2 x 2, 2-deep LDS ring fed by 1-KiB LDS-DMA pieces, two
// workgroups per CU, XOR-swizzled 128-byte rows), but a ring row holds 128 K ELEMENTS instead of 64: half the LDS-DMA pieces
// and half the LDS reads per FLOP — the two costs that bound the bf16 loop — at twice the MFMA rate.
//   K order: "halves" of 64 input channels, tap-major: half q = tap * (Cin / 64) + chunk; a K step is two consecutive halves,
//   so the two 64-byte halves of a ring row may belong to different taps (each lane fetches for the half its chunk is in);
//   9 * Cin / 64 may be odd: the weight rows are zero-padded to whole K steps and the missing half reads the zero page.
//   Quantisation: activations y = fp8(ACT_SCALE * SiLU(GroupNorm(x))) written by the GroupNorm-apply kernel (values in
//   [-0.28, ~8] -> scaled by 8 into e4m3's normal range); weights per output channel w8 = fp8(w / s_o), s_o = max|w_o| / 448;
//   the epilogue multiplies the fp32 accumulator by s_o / ACT_SCALE before bias / time-embedding row / residual.
#include "ops.h"
#include <hip/hip_fp16.h>

#define F8_BM 128
#define F8_ROWB 128          // bytes per ring row = K elements per step
#define F8_THREADS 256

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct ConvF8Dev {
  ConvF8Args a;
  const unsigned char* zero;
  int ksteps, ksteps_per_split, tiles_n, hpt, nhalves;   // hpt: halves per tap (Cin / 64); nhalves = 9 * hpt
};

template <int BN, int SPLIT>
__global__ __launch_bounds__(F8_THREADS, 2) void conv3x3_fp8_kernel(const ConvF8Dev d) {
  constexpr int NT = BN / 32;
  constexpr int A_BYTES = F8_BM * F8_ROWB, B_BYTES = BN * F8_ROWB, BUF_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvF8Args& p = d.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile;
  {   // XCD-aware remap (see gemm.hip): every XCD a contiguous range of tiles
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = tile % d.tiles_n, tm = tile / d.tiles_n;
  const int m0 = tm * F8_BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int kt_beg = z * d.ksteps_per_split;
  int kt_end = kt_beg + d.ksteps_per_split;
  if (kt_end > d.ksteps) kt_end = d.ksteps;
  const int nsteps = kt_end - kt_beg;

  // staging geometry: instruction i of wave w fills tile rows (i*4+w)*8 .. +8; lane -> (row in group, 16-B chunk)
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;         // logical chunk of the 128-B row this lane fetches (source-side swizzle)
  const int hf = schunk >> 2;                   // which 64-channel half of the K step the chunk lies in
  const int cb = (schunk & 3) * 16;             // byte offset inside that half's 64-channel run
  int a_pix[4], a_fl[4];                        // centre input pixel (element offset / Cin), border flags
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (i * 4 + w) * 8 + srow;
    if (m > p.M - 1) m = p.M - 1;
    const int hw = p.H * p.W;
    const int b = m / hw, r = m - b * hw;
    const int y = r / p.W, x = r - y * p.W;
    a_pix[i] = m;
    a_fl[i] = ((y - 1 >= 0) ? 1 : 0) | ((y + 1 < p.H) ? 2 : 0) | ((x - 1 >= 0) ? 4 : 0) | ((x + 1 < p.W) ? 8 : 0);
  }
  constexpr int WI = BN / 32;                   // W instructions per wave per step (BN rows / 8 per instr / 4 waves)
  const unsigned char* w_ptr[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int n = n0 + (i * 4 + w) * 8 + srow;
    if (n > p.N - 1) n = p.N - 1;
    w_ptr[i] = p.W8 + (size_t)n * p.Kpad + schunk * 16 + (size_t)kt_beg * F8_ROWB;
  }
  const unsigned char* zero_lane = d.zero + cb;

  int step_issue = kt_beg;
  auto issue = [&](int buf) {
    unsigned char* As = smem + buf * BUF_BYTES;
    unsigned char* Bs = As + A_BYTES;
    // the two halves of this K step (wave-uniform): q -> (tap, channel offset, displacement, border need); each lane uses
    // the half its chunk lies in
    int delta[2], need[2], valid[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = step_issue * 2 + h;
      const int tap = q / d.hpt;
      const int c0 = (q - tap * d.hpt) * 64;
      const int ty = tap / 3, dy = ty - 1, dx = tap - ty * 3 - 1;
      delta[h] = (dy * p.W + dx) * p.Cin + c0;
      need[h] = (dy < 0 ? 1 : (dy > 0 ? 2 : 0)) | (dx < 0 ? 4 : (dx > 0 ? 8 : 0));
      valid[h] = q < d.nhalves;
    }
    const int my_delta = hf ? delta[1] : delta[0];
    const int my_need = hf ? need[1] : need[0];
    const bool my_valid = hf ? valid[1] : valid[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = my_valid && ((a_fl[i] & my_need) == my_need);
      const unsigned char* src = ok ? p.A8 + (size_t)a_pix[i] * p.Cin + my_delta + cb : zero_lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (i * 4 + w) * 8 * F8_ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                       (__attribute__((address_space(3))) void*)(Bs + (i * 4 + w) * 8 * F8_ROWB), 16, 0, 0);
      w_ptr[i] += F8_ROWB;
    }
    ++step_issue;
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wm = w >> 1, wn = w & 1;
  const int frow = lane & 15, fkb = lane >> 4;      // fragment row, 32-byte k block of the 128-wide step
  if (nsteps > 0) issue(0);
  int buf = 0;
  for (int it = 0; it < nsteps; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* As = smem + buf * BUF_BYTES;
    const unsigned char* Bs = As + A_BYTES;
    i32x8 af[4], bfr[NT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + frow;
      const i32x4 lo = *reinterpret_cast<const i32x4*>(As + row * F8_ROWB + (((fkb * 2) ^ (row & 7)) * 16));
      const i32x4 hi = *reinterpret_cast<const i32x4*>(As + row * F8_ROWB + (((fkb * 2 + 1) ^ (row & 7)) * 16));
      af[i] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * (BN / 2) + j * 16 + frow;
      const i32x4 lo = *reinterpret_cast<const i32x4*>(Bs + row * F8_ROWB + (((fkb * 2) ^ (row & 7)) * 16));
      const i32x4 hi = *reinterpret_cast<const i32x4*>(Bs + row * F8_ROWB + (((fkb * 2 + 1) ^ (row & 7)) * 16));
      bfr[j] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    if (it + 1 < nsteps) issue(buf ^ 1);      // stage the next step while the fragments are in flight from LDS
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)   // operands swapped (W rows as the A operand) so that a lane holds 4 consecutive N, as in gemm.hip;
                                     // cbsz = blgp = 0: both fp8 e4m3; block scales 2^0 (E8M0 127)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);
    buf ^= 1;
  }

  // ---- epilogue.  acc[i][j][r]: m = m0 + wm*64 + i*16 + (lane&15), n = n0 + wn*BN/2 + j*16 + (lane>>4)*4 + r
  const int mrow = m0 + wm * 64 + frow;
  const int ncol = n0 + wn * (BN / 2) + fkb * 4;
  if constexpr (SPLIT) {
    // fp32 partials, already de-quantised (x s_o / ACT_SCALE): the bf16 path's reducer (gemm.hip) finishes them
    float* ws = p.ws + (size_t)z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = ncol + j * 16;
      if (n >= p.N) continue;
      const float4 sc = *reinterpret_cast<const float4*>(p.colscale + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mrow + i * 16;
        if (m >= p.M) continue;
        *reinterpret_cast<float4*>(ws + (size_t)m * p.N + n) =
            make_float4(acc[i][j][0] * sc.x, acc[i][j][1] * sc.y, acc[i][j][2] * sc.z, acc[i][j][3] * sc.w);
      }
    }
    return;
  } else {
    float* red = reinterpret_cast<float*>(smem);     // [2 moments][2 row halves][BN]: fixed-order GroupNorm partials (gemm.hip)
    if (p.gn_stats) __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = ncol + j * 16;
      if (n >= p.N) continue;
      const float4 sc = *reinterpret_cast<const float4*>(p.colscale + n);
      const float4 bz = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
      float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mrow + i * 16;
        if (m >= p.M) continue;
        float v[4] = {acc[i][j][0] * sc.x + bz.x, acc[i][j][1] * sc.y + bz.y, acc[i][j][2] * sc.z + bz.z, acc[i][j][3] * sc.w + bz.w};
        if (p.rowvec) {
          const float4 b = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_bstride + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.resid) {
          const uint2 r = *reinterpret_cast<const uint2*>(p.resid + (size_t)m * p.N + n);
          v[0] += bf2f((bf16_t)(r.x & 0xffff)); v[1] += bf2f((bf16_t)(r.x >> 16));
          v[2] += bf2f((bf16_t)(r.y & 0xffff)); v[3] += bf2f((bf16_t)(r.y >> 16));
        }
        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p.C + (size_t)m * p.N + n) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
      }
      if (p.gn_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = row16_sum(gs[e]), q = row16_sum(gq[e]);     // over the 16 row lanes (DPP)
          if (frow == 0) {
            const int c = wn * (BN / 2) + j * 16 + fkb * 4 + e;
            red[wm * BN + c] = a;
            red[2 * BN + wm * BN + c] = q;
          }
        }
      }
    }
    if (p.gn_stats) {
      __syncthreads();
      const int bins_tile = BN / p.gn_cg;
      if (tid < 2 * 2 * bins_tile) {
        const int which = tid & 1, half = (tid >> 1) & 1, lb = tid >> 2;
        const int mfirst = m0 + half * 64;
        const int bin = n0 / p.gn_cg + lb;
        if (mfirst < p.M && bin < p.gn_groups) {
          const float* src = red + which * 2 * BN + half * BN + lb * p.gn_cg;
          float a = 0.f;
          for (int c = 0; c < p.gn_cg; ++c) a += src[c];
          const int b = mfirst / p.rows_per_batch;
          const int slab = (mfirst - b * p.rows_per_batch) / GN_SLAB_ROWS;
          const int nslab = p.rows_per_batch / GN_SLAB_ROWS;
          p.gn_stats[(((size_t)b * nslab + slab) * p.gn_groups + bin) * 2 + which] = a;
        }
      }
    }
  }
}

template <int BN, int SPLIT>
static int conv_fp8_inst(const ConvF8Dev& d, dim3 grid, hipStream_t s) {
  static bool attr_set = false;
  constexpr int smem = 2 * (F8_BM + BN) * F8_ROWB;
  if (!attr_set) {
    GILL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_fp8_kernel<BN, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_fp8_kernel<BN, SPLIT>), grid, dim3(F8_THREADS), smem, s, d);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv_fp8_kpad(int Cin) { return (9 * (Cin / 64) + 1) / 2 * F8_ROWB; }

int conv3x3_fp8_launch(const ConvF8Args& a, hipStream_t s) {
  GILL_REQUIRE(a.A8 && a.W8 && a.colscale && a.C, "conv fp8: null operand");
  GILL_REQUIRE(a.Cin % 64 == 0 && a.N % 4 == 0 && a.M == a.B * a.H * a.W, "conv fp8: Cin % 64, N % 4, M = B*H*W");
  GILL_REQUIRE(a.Kpad == conv_fp8_kpad(a.Cin), "conv fp8: weight rows must be padded to whole K steps");
  GILL_REQUIRE((int64_t)a.M * a.Cin < (int64_t)1 << 31, "conv fp8: input too large for 32-bit offsets");
  ConvF8Dev d;
  d.a = a;
  d.zero = reinterpret_cast<const unsigned char*>(gill_zero_page());
  GILL_REQUIRE(d.zero != nullptr, "zero page unavailable");
  d.hpt = a.Cin / 64;
  d.nhalves = 9 * d.hpt;
  d.ksteps = a.Kpad / F8_ROWB;
  const int sk = a.splitk > 1 ? a.splitk : 1;
  d.a.splitk = sk;
  d.ksteps_per_split = cdiv(d.ksteps, sk);
  const int bn = (a.N % 160 == 0) ? 160 : 128;
  d.tiles_n = cdiv(a.N, bn);
  if (a.gn_stats) {
    GILL_REQUIRE(a.rows_per_batch % 64 == 0 && a.gn_groups > 0 && a.gn_groups <= 64 && a.gn_cg * a.gn_groups == a.N && bn % a.gn_cg == 0,
                 "conv fp8: fused GroupNorm statistics layout");
    GILL_REQUIRE(sk == 1, "conv fp8: fused statistics come from the reducer when split-K is on");
  }
  if (sk > 1) GILL_REQUIRE(a.ws != nullptr, "conv fp8: split-K workspace missing");
  dim3 grid(cdiv(a.M, F8_BM) * d.tiles_n, sk, 1);
  if (bn == 160) return sk > 1 ? conv_fp8_inst<160, 1>(d, grid, s) : conv_fp8_inst<160, 0>(d, grid, s);
  return sk > 1 ? conv_fp8_inst<128, 1>(d, grid, s) : conv_fp8_inst<128, 0>(d, grid, s);
}

// ---------------------------------------------------------------------------------------------- weight quantisation
// OIHW (3x3) of any supported dtype -> w8[o][Kpad] fp8 e4m3 in the kernel's K order (half q = tap * Cin/64 + chunk, then 64
// channels), zero-padded; scale[o] = max|w_o| / 448 / ACT_SCALE... the activation scale is folded in here so that the epilogue
// multiplies by one number per output channel.  One block per output channel.
__global__ __launch_bounds__(256) void conv_w_quant_fp8_kernel(const void* w, int dtype, int Cin, int Kpad, float act_scale,
                                                               unsigned char* w8, float* colscale) {
  __shared__ float red[256];
  const int o = blockIdx.x;
  const int n = Cin * 9;
  auto load = [&](int i) -> float {
    const int64_t j = (int64_t)o * n + i;
    if (dtype == 0) return bf2f(((const bf16_t*)w)[j]);
    if (dtype == 1) return ((const float*)w)[j];
    return (float)(((const __half*)w)[j]);
  };
  float mx = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, fabsf(load(i)));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  const float amax = red[0];
  const float sc = amax > 0.f ? amax / 448.f : 1.f;
  const float inv = 1.f / sc;
  if (threadIdx.x == 0) colscale[o] = sc / act_scale;
  const int hpt = Cin / 64;
  unsigned char* dst = w8 + (size_t)o * Kpad;
  for (int k = threadIdx.x * 2; k < Kpad; k += 512) {     // two K positions per thread: one cvt_pk
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = k + e;
      const int q = kk >> 6, cc = kk & 63;
      const int tap = q / hpt, c = (q - tap * hpt) * 64 + cc;
      v[e] = (q < 9 * hpt) ? load(c * 9 + tap) * inv : 0.f;       // OIHW: ((o*Cin + c)*9 + tap)
      v[e] = clamp_fp8_keep_nan(v[e]);
    }
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    *reinterpret_cast<unsigned short*>(dst + k) = (unsigned short)(pk & 0xffff);
  }
}

int conv_weight_quant_fp8_launch(const void* w_oihw, int dtype, int Cout, int Cin, float act_scale, unsigned char* w8, float* colscale,
                                 hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2 && Cin % 64 == 0, "conv fp8: unsupported weight dtype / Cin % 64");
  hipLaunchKernelGGL(conv_w_quant_fp8_kernel, dim3(Cout), dim3(256), 0, s, w_oihw, dtype, Cin, conv_fp8_kpad(Cin), act_scale, w8, colscale);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// bf16 -> fp8(scale * x), 8 elements per thread (tests / tools: the engines get their fp8 activations from the GroupNorm-apply kernel)
__global__ __launch_bounds__(256) void quant_bf16_fp8_kernel(const bf16_t* __restrict__ x, float scale, int64_t n8, unsigned char* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + i * 8);
    const unsigned uu[4] = {u.x, u.y, u.z, u.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = clamp_fp8_keep_nan(bf2f((bf16_t)(uu[e] & 0xffff)) * scale);
      o[2 * e + 1] = clamp_fp8_keep_nan(bf2f((bf16_t)(uu[e] >> 16)) * scale);
    }
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], hi, true);
    *reinterpret_cast<uint2*>(y + i * 8) = make_uint2((unsigned)lo, (unsigned)hi);
  }
}
int quant_bf16_fp8_launch(const bf16_t* x, float scale, int64_t n, unsigned char* y, hipStream_t s) {
  GILL_REQUIRE(n % 8 == 0, "fp8 quantisation: element count must be a multiple of 8");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(quant_bf16_fp8_kernel, dim3((int)blocks), dim3(256), 0, s, x, scale, n / 8, y);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
