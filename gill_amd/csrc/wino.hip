// Winograd F(2x2, 3x3) for stride-1 3x3 convolutions whose GEMM is long-K and whose split-K partials already exist (UNet level 2 at the
// 8-sample batch: 16 x 16 maps, 640..2560 -> 1280 channels).  VERDICT r04 item 2; go / no-go numbers: profiles/r05_winograd.md.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d: 4 x 4 input patch at (2 ty - 1, 2 tx - 1), Y: the 2 x 2 output pixels of tile (ty, tx)
//
// The 16 per-position GEMMs  M_p [tiles x Cout] = V_p [tiles x Cin] . U_p [Cout x Cin]^T  ARE a split-K GEMM: with V stored as
// [tiles][16][Cin] and U as [Cout][16][Cin] (position-major inside a row), split z of a 16-way split of K = 16 Cin multiplies exactly
// position z's operands and writes its fp32 partial plane — no new matrix kernel: gemm_launch() with GemmArgs::wino on the plain
// ping-pong split-K path, 16/36 of the direct convolution's multiply-adds.  The split-K reducers (gemm.hip, WINO template flag) apply
// A^T M A to the 16 planes instead of summing them and then run their usual epilogue (bias, time-embedding row, residual, fused
// GroupNorm).  This file holds the two transforms around that GEMM:
//   wino_weight_transform_launch   U = G g G^T in fp32 from the checkpoint's OIHW weights, one rounding to bf16 (at load)
//   wino_input_transform_launch    V = B^T d B of the normalised NHWC activation (sums / differences of four bf16 values, one rounding)
// Numerics (tools/winograd/numerics_probe.py, full-size SD-1.5 forward, fp32 everywhere else): the 25 stride-1 convolutions with Cin >= 640
// cost 3.9e-3 relative L2 in this form against 1.9e-3 for the direct bf16 convolution; the engine's whole-forward distance is 1.1e-2.
#include "ops.h"

static inline int grid_for(int64_t n, int per_block = 256, int cap = 16384) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float wino_load_w(const void* p, int dtype, int64_t i) {
  if (dtype == 0) return bf2f(((const bf16_t*)p)[i]);
  if (dtype == 1) return ((const float*)p)[i];
  return (float)(((const _Float16*)p)[i]);
}

// U[o][p = i * 4 + j][c] = (G g G^T)[i][j],  G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]
__global__ __launch_bounds__(256) void wino_weight_transform_kernel(const void* __restrict__ w, int dtype, int Cout, int Cin, bf16_t* __restrict__ U) {
  const int64_t n = (int64_t)Cout * Cin;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin);
    const int o = (int)(idx / Cin);
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = wino_load_w(w, dtype, idx * 9 + t);
    float h[4][3];                      // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h[0][j] = g[0][j];
      h[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
      h[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
      h[3][j] = g[2][j];
    }
    bf16_t* dst = U + (size_t)o * 16 * Cin + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (G g) G^T
      dst[(size_t)(i * 4 + 0) * Cin] = f2bf(h[i][0]);
      dst[(size_t)(i * 4 + 1) * Cin] = f2bf(0.5f * (h[i][0] + h[i][1] + h[i][2]));
      dst[(size_t)(i * 4 + 2) * Cin] = f2bf(0.5f * (h[i][0] - h[i][1] + h[i][2]));
      dst[(size_t)(i * 4 + 3) * Cin] = f2bf(h[i][2]);
    }
  }
}

int wino_weight_transform_launch(const void* w_oihw, int dtype, int Cout, int Cin, bf16_t* U, hipStream_t s) {
  GILL_REQUIRE(dtype >= 0 && dtype <= 2, "unsupported source dtype");
  hipLaunchKernelGGL(wino_weight_transform_kernel, dim3(grid_for((int64_t)Cout * Cin)), dim3(256), 0, s, w_oihw, dtype, Cout, Cin, U);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}

// V[t][p = i * 4 + j][c] = (B^T d B)[i][j],  B^T = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
// thread = (tile t, 8 channels): 16 pixel loads of 16 B (zero outside the image), 16 stores of 16 B.  Tiles in (sample, ty, tx) order.
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, bf16_t* __restrict__ V) {
  const int TH = H >> 1, TW = W >> 1, C8 = C >> 3;
  const int64_t n = (int64_t)B * TH * TW * C8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % C8);
    const int t = (int)(idx / C8);
    const int b = t / (TH * TW);
    const int r = t - b * (TH * TW);
    const int ty = r / TW, tx = r - ty * TW;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    uint32_t wds[4][4][4];          // [i][j][word]: two channels per word; transformed in place, word by word
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = y0 + i, xx = x0 + j;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(x + ((size_t)(b * H + yy) * W + xx) * C + c8 * 8) : make_uint4(0u, 0u, 0u, 0u);
        wds[i][j][0] = v.x; wds[i][j][1] = v.y; wds[i][j][2] = v.z; wds[i][j][3] = v.w;
      }
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
      float lo[4][4], hi[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo[i][j] = bf2f((bf16_t)(wds[i][j][wd] & 0xffff)); hi[i][j] = bf2f((bf16_t)(wds[i][j][wd] >> 16)); }
      float tl[4][4], th[4][4];      // B^T d
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tl[0][j] = lo[0][j] - lo[2][j]; tl[1][j] = lo[1][j] + lo[2][j]; tl[2][j] = lo[2][j] - lo[1][j]; tl[3][j] = lo[1][j] - lo[3][j];
        th[0][j] = hi[0][j] - hi[2][j]; th[1][j] = hi[1][j] + hi[2][j]; th[2][j] = hi[2][j] - hi[1][j]; th[3][j] = hi[1][j] - hi[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // (B^T d) B
        wds[i][0][wd] = pack_bf2(tl[i][0] - tl[i][2], th[i][0] - th[i][2]);
        wds[i][1][wd] = pack_bf2(tl[i][1] + tl[i][2], th[i][1] + th[i][2]);
        wds[i][2][wd] = pack_bf2(tl[i][2] - tl[i][1], th[i][2] - th[i][1]);
        wds[i][3][wd] = pack_bf2(tl[i][1] - tl[i][3], th[i][1] - th[i][3]);
      }
    }
    bf16_t* dst = V + (size_t)t * 16 * C + c8 * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint4*>(dst + (size_t)(i * 4 + j) * C) = make_uint4(wds[i][j][0], wds[i][j][1], wds[i][j][2], wds[i][j][3]);
  }
}

int wino_input_transform_launch(const bf16_t* x, int B, int H, int W, int C, bf16_t* V, hipStream_t s) {
  GILL_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "Winograd input transform: even map sides, channels in eights");
  hipLaunchKernelGGL(wino_input_transform_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, s, x, B, H, W, C, V);
  GILL_CHECK_HIP(hipGetLastError());
  return 0;
}
