// Small HBM-bound kernels of the generate_images path: embedding gathers, casts, timestep
// embedding, the tiny first/last UNet convolutions, the lm_head GEMV, the fused
// classifier-free-guidance + PLMS update, and the one-off weight re-layouts.
#include "ops.h"
#include <hip/hip_fp16.h>

static inline int grid_for(int64_t n, int per_block = 256, int cap = 8192) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- embeddings
// out[b][t][:] = table[ids[b][t]][:] + pos_table[t + pos_offset][:]   (fp32 residual stream)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                           int vocab, const bf16_t* __restrict__ pos, int pos_offset,
                                                           int T, int D, float* __restrict__ out) {
  const int row = blockIdx.x;  // b*T + t
  const int t = row % T;
  int64_t id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  const bf16_t* e = table + (size_t)id * D;
  const bf16_t* pe = pos ? pos + (size_t)(t + pos_offset) * D : nullptr;
  float* o = out + (size_t)row * D;
  for (int c = threadIdx.x * 2; c < D; c += blockDim.x * 2) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(e + c);
    float a = bf2f((bf16_t)(u & 0xffff)), bq = bf2f((bf16_t)(u >> 16));
    if (pe) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(pe + c);
      a += bf2f((bf16_t)(v & 0xffff)); bq += bf2f((bf16_t)(v >> 16));
    }
    *reinterpret_cast<float2*>(o + c) = make_float2(a, bq);
  }
}

int embed_tokens_launch(const int64_t* ids, const bf16_t* table, int vocab, const bf16_t* pos_table, int pos_offset,
                        int B, int T, int D, float* out_f32, hipStream_t s) {
  GILL_REQUIRE(D % 2 == 0, "embedding dim must be even");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(B * T), dim3(256), 0, s, ids, table, vocab, pos_table, pos_offset, T, D,
                     out_f32);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// out[row][:] = table[ids[row]][:]   (bf16 rows copied as they are: what nn.Embedding returns for a bf16 table)
__global__ __launch_bounds__(256) void embed_rows_bf16_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                              int vocab, int D, bf16_t* __restrict__ out) {
  const int row = blockIdx.x;
  int64_t id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  const uint32_t* e = reinterpret_cast<const uint32_t*>(table + (size_t)id * D);
  uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)row * D);
  for (int c = threadIdx.x; c < D / 2; c += blockDim.x) o[c] = e[c];
}
int embed_rows_bf16_launch(const int64_t* ids, const bf16_t* table, int vocab, int n, int D, bf16_t* out, hipStream_t s) {
  GILL_REQUIRE(D % 2 == 0, "embedding dim must be even");
  hipLaunchKernelGGL(embed_rows_bf16_kernel, dim3(n), dim3(256), 0, s, ids, table, vocab, D, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TS* __restrict__ src, const int32_t* __restrict__ idx, int D,
                                                          TD* __restrict__ dst) {
  const int r = blockIdx.x;
  const TS* sr = src + (size_t)idx[r] * D;
  TD* dr = dst + (size_t)r * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float v;
    if constexpr (sizeof(TS) == 2) v = bf2f(sr[c]); else v = sr[c];
    if constexpr (sizeof(TD) == 2) dr[c] = f2bf(v); else dr[c] = v;
  }
}

int gather_rows_launch(const void* src, int src_f32, const int32_t* row_idx, int nrows, int D, void* dst, int dst_f32,
                       hipStream_t s) {
  dim3 g(nrows), b(256);
  if (src_f32 && dst_f32) hipLaunchKernelGGL((gather_rows_kernel<float, float>), g, b, 0, s, (const float*)src, row_idx, D, (float*)dst);
  else if (src_f32 && !dst_f32) hipLaunchKernelGGL((gather_rows_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, row_idx, D, (bf16_t*)dst);
  else if (!src_f32 && dst_f32) hipLaunchKernelGGL((gather_rows_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, row_idx, D, (float*)dst);
  else hipLaunchKernelGGL((gather_rows_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, row_idx, D, (bf16_t*)dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void add_cast_kernel(const TA* __restrict__ a, const TB* __restrict__ b, int64_t n,
                                                       int64_t period, bf16_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x, y;
    if constexpr (sizeof(TA) == 2) x = bf2f(a[i]); else x = a[i];
    const int64_t j = i % period;
    if constexpr (sizeof(TB) == 2) y = bf2f(b[j]); else y = b[j];
    out[i] = f2bf(x + y);
  }
}

int add_cast_launch(const void* a, int a_f32, const void* b, int b_f32, int64_t n, int64_t b_period, bf16_t* out,
                    hipStream_t s) {
  dim3 g(grid_for(n)), blk(256);
  if (a_f32 && b_f32) hipLaunchKernelGGL((add_cast_kernel<float, float>), g, blk, 0, s, (const float*)a, (const float*)b, n, b_period, out);
  else if (a_f32 && !b_f32) hipLaunchKernelGGL((add_cast_kernel<float, bf16_t>), g, blk, 0, s, (const float*)a, (const bf16_t*)b, n, b_period, out);
  else if (!a_f32 && b_f32) hipLaunchKernelGGL((add_cast_kernel<bf16_t, float>), g, blk, 0, s, (const bf16_t*)a, (const float*)b, n, b_period, out);
  else hipLaunchKernelGGL((add_cast_kernel<bf16_t, bf16_t>), g, blk, 0, s, (const bf16_t*)a, (const bf16_t*)b, n, b_period, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}
int cast_f32_to_bf16_launch(const float* x, bf16_t* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int cast_bf16_to_f32_launch(const bf16_t* x, float* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- timestep embedding
// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out = [cos(t*f_i) | sin(t*f_i)],
// f_i = exp(-ln(10000) * i / half)
__global__ void timestep_embed_kernel(const float* __restrict__ t, int n, int dim, bf16_t* __restrict__ out) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  const int r = i / half, c = i - r * half;
  const float f = expf(-9.210340371976184f * (float)c / (float)half);
  const float a = t[r] * f;
  out[(size_t)r * dim + c] = f2bf(cosf(a));
  out[(size_t)r * dim + half + c] = f2bf(sinf(a));
}
int timestep_embed_launch(const float* t, int n, int dim, bf16_t* out, hipStream_t s) {
  GILL_REQUIRE(dim % 2 == 0, "timestep embedding dim must be even");
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(cdiv(n * dim / 2, 256)), dim3(256), 0, s, t, n, dim, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = f2bf(silu_f(bf2f(x[i])));
}
int silu_bf16_launch(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- conv_in / conv_out (direct)
// conv_in: thread = (pixel, 8 output channels); input latents are fp32 NCHW (tiny: Cin = 4).
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const float* __restrict__ bias, int B, int Cin, int H, int W,
                                                      int Cout, bf16_t* __restrict__ y) {
  const int groups = Cout / 8;
  const int64_t total = (int64_t)B * H * W * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const int64_t pix = i / groups;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bias ? bias[g * 8 + o] : 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = yh + tap / 3 - 1, ix = xw + tap % 3 - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      for (int c = 0; c < Cin; ++c) {
        // the oracle rounds the conv input to bf16 like every other activation entering an MFMA
        const float v = bf2f(f2bf(x[(((size_t)b * Cin + c) * H + iy) * W + ix]));
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] += v * bf2f(w[((size_t)(g * 8 + o) * 9 + tap) * Cin + c]);
      }
    }
    uint4 u;
    u.x = pack_bf2(acc[0], acc[1]); u.y = pack_bf2(acc[2], acc[3]); u.z = pack_bf2(acc[4], acc[5]); u.w = pack_bf2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(y + (size_t)pix * Cout + g * 8) = u;
  }
}
int conv_in_launch(const float* x, const bf16_t* w, const float* bias, int B, int Cin, int H, int W, int Cout, bf16_t* y,
                   hipStream_t s) {
  GILL_REQUIRE(Cout % 8 == 0, "conv_in: Cout must be a multiple of 8");
  const int64_t total = (int64_t)B * H * W * (Cout / 8);
  hipLaunchKernelGGL(conv_in_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, w, bias, B, Cin, H, W, Cout, y);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// zero / nzero: a range of 32-bit words this kernel also clears — the UNet forward's COOP arrival counters (GemmArgs::coop_ctr), reset by the
// forward's first kernel instead of a launch (or a memset node) of their own
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, int B, int Cin, int H, int W, int kpad,
                                                          bf16_t* __restrict__ out, unsigned* __restrict__ zero, int nzero) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) zero[i] = 0u;
  const int64_t total = (int64_t)B * H * W * (kpad / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int oct = (int)(i % (kpad / 8));
    const int64_t pix = i / (kpad / 8);
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = oct * 8 + j;
      const int tap = k / Cin, c = k - tap * Cin;
      const int iy = yh + tap / 3 - 1, ix = xw + tap % 3 - 1;
      const bool ok = (k < 9 * Cin) && iy >= 0 && iy < H && ix >= 0 && ix < W;
      v[j] = ok ? x[(((size_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
    }
    uint4 u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(out + pix * kpad + oct * 8) = u;
  }
}
int im2col_nchw_launch(const float* x, int B, int Cin, int H, int W, int kpad, bf16_t* out, hipStream_t s, unsigned* zero, int nzero) {
  GILL_REQUIRE(kpad % 64 == 0 && 9 * Cin <= kpad, "im2col: kpad must be a multiple of 64 covering 9*Cin");
  const int64_t total = (int64_t)B * H * W * (kpad / 8);
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, B, Cin, H, W, kpad, out, zero, zero ? nzero : 0);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// conv_out: one wave per output pixel; lanes stride over (tap, channel-chunk), shuffle-reduce the Cout (<=8) sums.
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const float* __restrict__ bias, int B, int Cin, int H, int W,
                                                       int Cout, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (int64_t)B * H * W) return;
  const int xw = (int)(pix % W);
  const int yh = (int)((pix / W) % H);
  const int b = (int)(pix / ((int64_t)W * H));
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  const int chunks = Cin / 8;
  for (int j = lane; j < 9 * chunks; j += 64) {
    const int tap = j / chunks, c = (j - tap * chunks) * 8;
    const int iy = yh + tap / 3 - 1, ix = xw + tap % 3 - 1;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const uint4 u = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * Cin + c);
    const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((bf16_t)(uu[i] & 0xffff)); v[2 * i + 1] = bf2f((bf16_t)(uu[i] >> 16)); }
    for (int o = 0; o < Cout; ++o) {
      const uint4 wv = *reinterpret_cast<const uint4*>(w + ((size_t)o * 9 + tap) * Cin + c);
      const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s += v[2 * i] * bf2f((bf16_t)(ww[i] & 0xffff)) + v[2 * i + 1] * bf2f((bf16_t)(ww[i] >> 16));
      acc[o] += s;
    }
  }
  for (int o = 0; o < Cout; ++o) {
    const float s = wave_sum(acc[o]);
    if (lane == 0) y[(((size_t)b * Cout + o) * H + yh) * W + xw] = s + (bias ? bias[o] : 0.f);
  }
}
// conv_out on the matrix pipe: one wave = 16 consecutive pixels of an image row x all (<= 16) output channels, one
// v_mfma_f32_16x16x32_bf16 per (tap, 32 input channels).  First operand = the weight rows (rows >= Cout are zero registers; the
// [Cout][9][Cin] weights sit in LDS, copied once per workgroup), second operand = the 16 pixels' input channels straight from
// global memory (16 B per lane; the 3x3 halo overlaps of neighbouring waves hit in L1/L2).  A lane ends up with output channels
// (lane >> 4) * 4 + r of pixel lane & 15: 64-byte fp32 NCHW row segments.  Two accumulators break the MFMA dependency chain.
// The one-wave-per-pixel kernel above spent 68 us on the UNet's 64 x 64 x 320 -> 4 conv_out (VALU + L1 bound); this one is fetch-bound.
template <int KS>     // KS = Cin / 32 when compile-time (all of a tap's loads in flight before its MFMAs), 0 = run-time loop
__global__ __launch_bounds__(256) void conv_out_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int Cin, int H, int W,
                                                            int Cout, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cw_smem[];
  bf16_t* ws = reinterpret_cast<bf16_t*>(cw_smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nw16 = Cout * 9 * Cin / 8;
  for (int i = tid; i < nw16; i += 256) reinterpret_cast<uint4*>(ws)[i] = reinterpret_cast<const uint4*>(w)[i];
  __syncthreads();
  const int gpr = W / 16;                                   // 16-pixel groups per image row
  const int64_t g = (int64_t)blockIdx.x * 4 + wv;
  if (g >= (int64_t)B * H * gpr) return;
  const int gx = (int)(g % gpr);
  const int yh = (int)((g / gpr) % H);
  const int b = (int)(g / ((int64_t)gpr * H));
  const int px = lane & 15, kg = lane >> 4;
  const int xw = gx * 16 + px;
  const bool wrow = px < Cout;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int ksteps = KS > 0 ? KS : Cin / 32;
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = yh + tap / 3 - 1, ix = xw + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const bf16_t* xp = x + (((size_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin + kg * 8;
    const bf16_t* wp = ws + ((size_t)(wrow ? px : 0) * 9 + tap) * Cin + kg * 8;
    if constexpr (KS > 0) {
      bf16x8 xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + ks * 32);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
        if (!wrow) wf = zero;
        const bf16x8 xv = ok ? xf[ks] : zero;
        if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xv, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xv, acc0, 0, 0, 0);
      }
    } else {
      for (int ks = 0; ks < ksteps; ++ks) {
        bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xp + ks * 32);
        bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
        if (!ok) x0 = zero;
        if (!wrow) w0 = zero;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc0, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = kg * 4 + r;
    if (o < Cout) y[(((size_t)b * Cout + o) * H + yh) * W + xw] = acc0[r] + acc1[r] + (bias ? bias[o] : 0.f);
  }
}

int conv_out_launch(const bf16_t* x, const bf16_t* w, const float* bias, int B, int Cin, int H, int W, int Cout, float* y,
                    hipStream_t s) {
  const size_t wbytes = (size_t)Cout * 9 * Cin * sizeof(bf16_t);
  // matrix-pipe kernel where its geometry holds; the one-wave-per-pixel kernel is the general path (rows not a multiple of 16 wide, ...)
  if (Cout <= 16 && Cin % 32 == 0 && W % 16 == 0 && wbytes <= 64 * 1024 && (((uintptr_t)w | (uintptr_t)x) & 15) == 0) {
    const int64_t groups = (int64_t)B * H * (W / 16);
    const dim3 grid((unsigned)cdiv64(groups, 4));
    if (Cin == 320) hipLaunchKernelGGL(conv_out_mfma_kernel<10>, grid, dim3(256), wbytes, s, x, w, bias, B, Cin, H, W, Cout, y);
    else if (Cin == 128) hipLaunchKernelGGL(conv_out_mfma_kernel<4>, grid, dim3(256), wbytes, s, x, w, bias, B, Cin, H, W, Cout, y);
    else if (Cin == 64) hipLaunchKernelGGL(conv_out_mfma_kernel<2>, grid, dim3(256), wbytes, s, x, w, bias, B, Cin, H, W, Cout, y);
    else hipLaunchKernelGGL(conv_out_mfma_kernel<0>, grid, dim3(256), wbytes, s, x, w, bias, B, Cin, H, W, Cout, y);
    GILL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  GILL_REQUIRE(Cout <= 8 && Cin % 8 == 0, "conv_out: Cout <= 8 and Cin % 8 == 0 required");
  const int64_t pix = (int64_t)B * H * W;
  hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)cdiv64(pix, 4)), dim3(256), 0, s, x, w, bias, B, Cin, H, W, Cout, y);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- skinny GEMM (lm_head): one wave per output column
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, int M,
                                                          int N, int K, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  const bf16_t* wr = W + (size_t)n * K;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    const uint4 wv = *reinterpret_cast<const uint4*>(wr + k);
    const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
    float wf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { wf[2 * i] = bf2f((bf16_t)(ww[i] & 0xffff)); wf[2 * i + 1] = bf2f((bf16_t)(ww[i] >> 16)); }
    for (int m = 0; m < M; ++m) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + (size_t)m * K + k);
      const uint32_t xx[4] = {xv.x, xv.y, xv.z, xv.w};
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sacc += wf[2 * i] * bf2f((bf16_t)(xx[i] & 0xffff)) + wf[2 * i + 1] * bf2f((bf16_t)(xx[i] >> 16));
      acc[m] += sacc;
    }
  }
  for (int m = 0; m < M; ++m) {
    const float s = wave_sum(acc[m]);
    if (lane == 0) out[(size_t)m * N + n] = s;
  }
}
int skinny_gemm_launch(const bf16_t* x, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  GILL_REQUIRE(M >= 1 && M <= 8 && K % 8 == 0, "skinny gemm: 1 <= M <= 8, K % 8 == 0");
  hipLaunchKernelGGL(skinny_gemm_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, x, W, M, N, K, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- CFG + PLMS
__global__ __launch_bounds__(256) void zero_bytes_kernel(uint4* dst, int64_t n16) {
  const uint4 z = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = z;
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(uint4* dst, const uint4* src, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the CFG pair's shared prefix ends: a[n16 ..] := a[0 .. n16) and b2[0 .. n16) = b2[n16 ..] := b[0 .. n16) in one launch (unet.hip)
__global__ __launch_bounds__(256) void dup_pair_kernel(uint4* a, const uint4* b, uint4* b2, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 va = a[i], vb = b[i];
    a[i + n16] = va;
    b2[i] = vb;
    b2[i + n16] = vb;
  }
}
int dup_pair_launch(void* a, const void* b, void* b2, size_t bytes, hipStream_t s) {
  GILL_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)b2) & 15) == 0 && (bytes & 15) == 0, "dup_pair: 16-byte alignment required");
  if (bytes == 0) return 0;
  hipLaunchKernelGGL(dup_pair_kernel, dim3(grid_for((int64_t)(bytes / 16))), dim3(256), 0, s, (uint4*)a, (const uint4*)b, (uint4*)b2,
                     (int64_t)(bytes / 16));
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int zero_bytes_launch(void* dst, size_t bytes, hipStream_t s) {
  GILL_REQUIRE(((uintptr_t)dst & 15) == 0 && (bytes & 15) == 0, "zero_bytes: 16-byte alignment required");
  if (bytes == 0) return 0;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid_for((int64_t)(bytes / 16))), dim3(256), 0, s, (uint4*)dst, (int64_t)(bytes / 16));
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int copy_bytes_launch(void* dst, const void* src, size_t bytes, hipStream_t s) {
  GILL_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0 && (bytes & 15) == 0, "copy_bytes: 16-byte alignment required");
  if (bytes == 0) return 0;
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for((int64_t)(bytes / 16))), dim3(256), 0, s, (uint4*)dst, (const uint4*)src,
                     (int64_t)(bytes / 16));
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void sd_stage_kernel(const SdLoopArgs a) {
  kernarg_warm<sizeof(SdLoopArgs)>();
  const int step = a.ctr[0];          // nobody writes ctr[0] while this kernel runs
  const int64_t total = (int64_t)a.B * a.n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < total; i += stride) {
    const float v = a.lat[i];
    a.lat2[i] = v;                    // scale_model_input is the identity for PNDM (custom_sd.py:630-631)
    if (a.cfg) a.lat2[total + i] = v;
  }
  const float* row = a.temb_table + (size_t)step * a.temb_total;
  for (int64_t i = t0; i < a.temb_total; i += stride) a.temb_cur[i] = row[i];
  if (t0 == 0) a.ctr[1] = step;
}
int sd_stage_launch(const SdLoopArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(sd_stage_kernel, dim3(grid_for((int64_t)a.B * a.n)), dim3(256), 0, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void plms_step_kernel(const SdLoopArgs a) {
  kernarg_warm<sizeof(SdLoopArgs)>();
  const int step = a.ctr[1];          // written by this step's stage kernel; nobody writes it while this kernel runs
  const PlmsRow r = a.rows[step];
  const int64_t total = (int64_t)a.B * a.n;
  const float guidance = a.cfg ? a.guidance[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float e = a.eps[i];
    if (a.cfg) {
      const float ec = a.eps[total + i];
      e = e + guidance * (ec - e);
    }
    float sample = a.lat[i];
    float ep;
    float* ets = a.ets;
    if (r.mode == 0) {            // counter 0: plain step, remember the sample
      ep = e;
      a.cur_sample[i] = sample;
      ets[(size_t)r.slot_new * total + i] = e;
    } else if (r.mode == 1) {     // counter 1: average with ets[-1], restart from the remembered sample
      ep = 0.5f * (e + ets[(size_t)r.s1 * total + i]);
      sample = a.cur_sample[i];
    } else {
      ets[(size_t)r.slot_new * total + i] = e;
      const float e1 = e;
      const float e2 = ets[(size_t)r.s1 * total + i];
      if (r.mode == 2) ep = (3.f * e1 - e2) * 0.5f;
      else {
        const float e3 = ets[(size_t)r.s2 * total + i];
        if (r.mode == 3) ep = (23.f * e1 - 16.f * e2 + 5.f * e3) * (1.f / 12.f);
        else {
          const float e4 = ets[(size_t)r.s3 * total + i];
          ep = (55.f * e1 - 59.f * e2 + 37.f * e3 - 9.f * e4) * (1.f / 24.f);
        }
      }
    }
    a.lat[i] = r.sample_coeff * sample - r.eps_coeff * ep;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.ctr[0] = step + 1;   // read only by the next step's stage kernel
}
int plms_step_launch(const SdLoopArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(plms_step_kernel, dim3(grid_for((int64_t)a.B * a.n)), dim3(256), 0, s, a);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- weight conversion / re-layout (creation time)
__device__ __forceinline__ float load_as_f32(const void* p, int dtype, int64_t i) {
  if (dtype == 0) return bf2f(((const bf16_t*)p)[i]);
  if (dtype == 1) return ((const float*)p)[i];
  return (float)(((const __half*)p)[i]);
}
__global__ __launch_bounds__(256) void convert_bf16_kernel(const void* src, int dtype, int64_t n, bf16_t* dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f2bf(load_as_f32(src, dtype, i));
}
__global__ __launch_bounds__(256) void convert_f32_kernel(const void* src, int dtype, int64_t n, float* dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = load_as_f32(src, dtype, i);
}
int convert_to_bf16_launch(const void* src, int dtype, int64_t n, bf16_t* dst, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  hipLaunchKernelGGL(convert_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dtype, n, dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int convert_to_f32_launch(const void* src, int dtype, int64_t n, float* dst, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  hipLaunchKernelGGL(convert_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dtype, n, dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
// [N][K] (any source dtype) -> bf16 [N / 64][K / 64][64][64]: the STREAM64 weight layout (gemm.hip).  One thread per 8 consecutive k: reads
// 16-32 B, writes 16 B; a wave covers 8 rows x 128 B of one block.
__global__ __launch_bounds__(256) void convert_bf16_blk64_kernel(const void* w, int dtype, int N, int K, bf16_t* out) {
  const int64_t n8 = (int64_t)N * K / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = i * 8;                       // destination element index
    const int64_t blk = o >> 12;
    const int r = (int)((o >> 6) & 63), c = (int)(o & 63);
    const int kb = (int)(blk % (K / 64));
    const int nb = (int)(blk / (K / 64));
    const int64_t src = ((int64_t)nb * 64 + r) * K + (int64_t)kb * 64 + c;
    uint4 v;
    v.x = pack_bf2(load_as_f32(w, dtype, src + 0), load_as_f32(w, dtype, src + 1));
    v.y = pack_bf2(load_as_f32(w, dtype, src + 2), load_as_f32(w, dtype, src + 3));
    v.z = pack_bf2(load_as_f32(w, dtype, src + 4), load_as_f32(w, dtype, src + 5));
    v.w = pack_bf2(load_as_f32(w, dtype, src + 6), load_as_f32(w, dtype, src + 7));
    *reinterpret_cast<uint4*>(out + o) = v;
  }
}
int convert_to_bf16_blk64_launch(const void* src, int dtype, int N, int K, bf16_t* dst, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  GILL_REQUIRE(N % 64 == 0 && K % 64 == 0, "64 x 64-blocked weights: N and K must be multiples of 64");
  hipLaunchKernelGGL(convert_bf16_blk64_kernel, dim3(grid_for((int64_t)N * K / 8)), dim3(256), 0, s, src, dtype, N, K, dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
// OIHW (3x3) -> [O][tap][I]
__global__ __launch_bounds__(256) void conv_w_relayout_kernel(const void* w, int dtype, int Cout, int Cin, bf16_t* out) {
  const int64_t n = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 9);
    const int o = (int)(i / ((int64_t)Cin * 9));
    out[i] = f2bf(load_as_f32(w, dtype, ((int64_t)o * Cin + c) * 9 + tap));
  }
}
// OIHW (3x3) -> [O][chunk][tap][64]: the K order of the implicit-GEMM convolution (gemm.hip, CONV != 0).  K walks the input
// channels in 64-wide chunks and, inside a chunk, the 9 taps: the 9 shifted reads of one chunk's input patch are 9
// consecutive K steps, so they hit a patch that is still in L2 (tap-major order re-streamed the whole input 9 times from the
// fabric once a sample's activations outgrew the 4 MiB L2: 398 MB fetched for a 42 MB input at level 0, Cin = 640).
__global__ __launch_bounds__(256) void conv_w_relayout_chunked_kernel(const void* w, int dtype, int Cout, int Cin, bf16_t* out) {
  const int64_t n = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % ((int64_t)Cin * 9));
    const int o = (int)(i / ((int64_t)Cin * 9));
    const int chunk = k / 576, r = k - chunk * 576;
    const int tap = r >> 6, c = chunk * 64 + (r & 63);
    out[i] = f2bf(load_as_f32(w, dtype, ((int64_t)o * Cin + c) * 9 + tap));
  }
}
// OIHW (3x3) of a convolution that follows a nearest-2x upsample -> [class = py*2+px][O][a*2+b][I]: the four 2x2-tap kernels on the
// SOURCE grid (gemm.hip, "UPS4").  Output pixel (2y+py, 2x+px), tap ky reads upsampled row 2y+py+ky-1 = source row
// y + floor((py+ky-1)/2): py = 0: ky 0 -> y-1 (a = 0), ky 1, 2 -> y (a = 1); py = 1: ky 0, 1 -> y (a = 0), ky 2 -> y+1 (a = 1).  The taps
// that share a source pixel are summed in fp32 (ascending ky, then kx) and rounded to bf16 once.
__global__ __launch_bounds__(256) void conv_w_relayout_ups4_kernel(const void* w, int dtype, int Cout, int Cin, bf16_t* out) {
  const int64_t n = (int64_t)4 * Cout * 4 * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int t4 = (int)((i / Cin) % 4);
    const int o = (int)((i / ((int64_t)Cin * 4)) % Cout);
    const int cls = (int)(i / ((int64_t)Cin * 4 * Cout));
    const int py = cls >> 1, px = cls & 1, a = t4 >> 1, b = t4 & 1;
    const int ky0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int kx0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += load_as_f32(w, dtype, ((int64_t)o * Cin + c) * 9 + ky * 3 + kx);
    out[i] = f2bf(acc);
  }
}
int conv_weight_relayout_ups4_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  GILL_REQUIRE(Cin % 64 == 0, "implicit-GEMM conv: input channels must be a multiple of 64");
  hipLaunchKernelGGL(conv_w_relayout_ups4_kernel, dim3(grid_for((int64_t)16 * Cout * Cin)), dim3(256), 0, s, w, dtype, Cout, Cin, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int conv_weight_relayout_chunked_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  GILL_REQUIRE(Cin % 64 == 0, "implicit-GEMM conv: input channels must be a multiple of 64");
  hipLaunchKernelGGL(conv_w_relayout_chunked_kernel, dim3(grid_for((int64_t)Cout * Cin * 9)), dim3(256), 0, s, w, dtype, Cout, Cin, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
int conv_weight_relayout_launch(const void* w, int dtype, int Cout, int Cin, bf16_t* out, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  hipLaunchKernelGGL(conv_w_relayout_kernel, dim3(grid_for((int64_t)Cout * Cin * 9)), dim3(256), 0, s, w, dtype, Cout, Cin, out);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* src, int cols, const int32_t* dst_rows, bf16_t* dst,
                                                           int dst_ld) {
  const int r = blockIdx.x;
  const bf16_t* sr = src + (size_t)r * cols;
  bf16_t* dr = dst + (size_t)dst_rows[r] * dst_ld;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) dr[c] = sr[c];
}
int scatter_rows_bf16_launch(const bf16_t* src, int rows, int cols, const int32_t* dst_rows, bf16_t* dst, int dst_ld,
                             hipStream_t s) {
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows), dim3(256), 0, s, src, cols, dst_rows, dst, dst_ld);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// dst[idx[i]] = src[i]  (fp32 vectors: bias permutations at creation time)
__global__ void permute_f32_kernel(const float* src, const int32_t* idx, int n, float* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}
int permute_f32_launch(const float* src, const int32_t* idx, int n, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(permute_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, idx, n, dst);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// token-major (B, n, H*d) -> head-major padded [B][H][n_pad][dp] (mode 0) or transposed [B][H][dpv][n_pad] (mode 1).
// Destination must be pre-zeroed (padding).
__global__ __launch_bounds__(256) void pack_heads_kernel(const bf16_t* __restrict__ src, int n, int H, int d, int n_pad,
                                                         int dp, int dpv, int mode, bf16_t* __restrict__ dst, int64_t total,
                                                         float mul) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(i % d);
    const int h = (int)((i / d) % H);
    const int t = (int)((i / ((int64_t)d * H)) % n);
    const int b = (int)(i / ((int64_t)d * H * n));
    const bf16_t v = (mul == 1.f) ? src[i] : f2bf(bf2f(src[i]) * mul);
    if (mode == 0) dst[((size_t)(b * H + h) * n_pad + t) * dp + dd] = v;
    else {
      dst[((size_t)(b * H + h) * dpv + dd) * n_pad + t] = v;
      if (dd == 0 && dpv > dp) dst[((size_t)(b * H + h) * dpv + dp) * n_pad + t] = (bf16_t)0x3F80;   // ones row (row sum)
    }
  }
}
int pack_heads_launch(const bf16_t* src, int B, int n, int H, int d, int n_pad, int dp, int dpv, int mode, bf16_t* dst,
                      hipStream_t s, float mul) {
  const int64_t total = (int64_t)B * n * H * d;
  hipLaunchKernelGGL(pack_heads_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, n, H, d, n_pad, dp, dpv, mode, dst, total, mul);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
// O (B*nq, H*dp) -> (B*nq, H*d): drop the per-head padding columns
__global__ __launch_bounds__(256) void unpad_heads_kernel(const bf16_t* __restrict__ src, int H, int d, int dp,
                                                          bf16_t* __restrict__ dst, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(i % d);
    const int h = (int)((i / d) % H);
    const int64_t row = i / ((int64_t)d * H);
    dst[i] = src[row * H * dp + h * dp + dd];
  }
}
int unpad_heads_launch(const bf16_t* src, int64_t rows, int H, int d, int dp, bf16_t* dst, hipStream_t s) {
  const int64_t total = rows * H * d;
  hipLaunchKernelGGL(unpad_heads_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, H, d, dp, dst, total);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// LayerNorm folded into the following Linear (see GemmArgs::ln_stats): one wave per weight row n:
//   W[n,k] <- bf16(W[n,k] * g[k]);  colsum[n] = sum_k W'[n,k] (of the ROUNDED products, what the MFMA multiplies);
//   bias[n] += sum_k beta[k] * W[n,k]
__global__ __launch_bounds__(256) void ln_fold_rows_kernel(bf16_t* __restrict__ W, int N, int K, const float* __restrict__ g,
                                                           const float* __restrict__ beta, float* __restrict__ colsum,
                                                           float* __restrict__ bias) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  bf16_t* row = W + (size_t)n * K;
  float s = 0.f, c = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = bf2f(row[k]);
    const bf16_t wp = f2bf(w * g[k]);
    row[k] = wp;
    s += bf2f(wp);
    c += beta[k] * w;
  }
  s = wave_sum(s); c = wave_sum(c);
  if (lane == 0) { colsum[n] = s; bias[n] += c; }
}

int ln_fold_rows_launch(bf16_t* W, int N, int K, const float* g, const float* beta, float* colsum, float* bias, hipStream_t s) {
  hipLaunchKernelGGL(ln_fold_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, N, K, g, beta, colsum, bias);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
