// Micro-benchmark: back-to-back v_mfma_f32_16x16x32_bf16 issue rate and core clock under load.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, long long* cyc, int iters, int zero) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u;
    a[i] = zero ? 0 : (short)(0x3c00 | (seed >> 20 & 0x3ff));
    seed = seed * 1664525u + 1013904223u;
    b[i] = zero ? 0 : (short)(0xbc00 ^ (seed >> 20 & 0x83ff));
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float* out, long long* cyc, int iters, int zero) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u;
    a[i] = zero ? 0 : (short)(0x3c00 | (seed >> 20 & 0x3ff));
    seed = seed * 1664525u + 1013904223u;
    b[i] = zero ? 0 : (short)(0xbc00 ^ (seed >> 20 & 0x83ff));
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run32(int blocks, int threads, int iters, int zero) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mfma32<NACC><<<blocks, threads>>>(out, cyc, iters, zero);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma32<NACC><<<blocks, threads>>>(out, cyc, iters, zero);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double nm = (double)iters * NACC;
  double waves = (double)blocks * threads / 64;
  double tf = waves * nm * 32768.0 / (ms * 1e-3) / 1e12;
  printf("32x32x16 NACC=%2d blocks=%4d thr=%3d zero=%d: %.3f ms, %.1f clk/MFMA/wave (clock64), %.0f TFLOP/s\n", NACC, blocks,
         threads, zero, ms, (double)c / nm, tf);
  hipFree(out); hipFree(cyc);
}

template <int NACC>
void run(int blocks, int threads, int iters, int zero) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mfma<NACC><<<blocks, threads>>>(out, cyc, iters, zero);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma<NACC><<<blocks, threads>>>(out, cyc, iters, zero);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double nm = (double)iters * NACC;
  double waves = (double)blocks * threads / 64;
  double tf = waves * nm * 16384.0 / (ms * 1e-3) / 1e12;
  printf("NACC=%2d blocks=%4d thr=%3d zero=%d: %.3f ms, %.1f clk/MFMA/wave (clock64), %.0f TFLOP/s, implied clock64 rate %.2f GHz\n", NACC, blocks,
         threads, zero, ms, (double)c / nm, tf, (double)c / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int zero = 1; zero >= 0; --zero) {
    run<20>(1, 64, 20000, zero);       // one wave alone
    run<20>(256, 256, 20000, zero);    // one wave per SIMD on every CU
    run<20>(512, 256, 20000, zero);    // two waves per SIMD
    run<20>(1024, 256, 10000, zero);   // four waves per SIMD
    run<4>(256, 256, 100000, zero);    // 4 accumulators: dependent distance 4
    run<8>(256, 256, 50000, zero);
    run32<5>(1, 64, 40000, zero);
    run32<5>(256, 256, 40000, zero);
    run32<5>(512, 256, 40000, zero);
    run32<2>(256, 256, 100000, zero);
  }
  return 0;
}
