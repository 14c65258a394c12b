// Does the GEMM epilogue's store pattern (8 B per lane: a wave instruction writes 16 rows x 32 B) cost write bandwidth against
// row-contiguous 16-B-per-lane stores of the same 128 x 160 bf16 tile?   hipcc --offload-arch=gfx950 -O3 store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(uint16_t* C, int M, int N, unsigned v) {
  const int tiles_n = N / 160;
  const int tile = blockIdx.x, tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 128, n0 = tn * 160;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  if (MODE == 0) {
    const int frow = lane & 15, fkc = lane >> 4;
    for (int j = 0; j < 5; ++j)
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + frow, n = n0 + wn * 80 + j * 16 + fkc * 4;
        *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = make_uint2(v + i, v + j);
      }
  } else {
    // 128 rows x 20 chunks of 16 B = 2560 chunks over 256 threads: 10 per thread, consecutive lanes = consecutive chunks of a row
    for (int c = tid; c < 2560; c += 256) {
      const int r = c / 20, ch = c % 20;
      *reinterpret_cast<uint4*>(C + (size_t)(m0 + r) * N + n0 + ch * 8) = make_uint4(v, v + 1, v + 2, v + 3);
    }
  }
}
int main() {
  const int sizes[][2] = {{32768, 320}, {32768, 1280}, {8192, 640}, {131072, 320}};
  for (auto& sz : sizes) {
    const int M = sz[0], N = sz[1];
    uint16_t* C; hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
      const int grid = (M / 128) * (N / 160);
      for (int it = 0; it < 3; ++it) { if (mode == 0) k<0><<<grid, 256>>>(C, M, N, it); else k<1><<<grid, 256>>>(C, M, N, it); }
      hipEventRecord(e0);
      for (int it = 0; it < 50; ++it) { if (mode == 0) k<0><<<grid, 256>>>(C, M, N, it); else k<1><<<grid, 256>>>(C, M, N, it); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("M %d N %d mode %d (%s): %.1f us, %.2f TB/s\n", M, N, mode, mode ? "16 B/lane row-contiguous" : "8 B/lane, 16 rows x 32 B",
             ms * 1e3 / 50, (double)M * N * 2 / (ms * 1e-3 / 50) / 1e12);
    }
    hipFree(C);
  }
  return 0;
}
