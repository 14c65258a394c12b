// Micro-benchmark: how fast can a CU fill LDS from L2-resident data on MI355X, by path and by number of issuing waves?
//   DMA : global_load_lds_dwordx4 (1 KiB per wave-instruction, M0 = LDS destination; what every GEMM ring of this library uses)
//   REG : global_load_dwordx4 into VGPRs, then ds_write_b128 (the round-1 staging)
//   MIX : alternate pieces between the two paths
// One 512-thread workgroup per CU (256 workgroups), `active` of its 8 waves issue; each issuing wave moves PIECES x 1 KiB per iteration from
// a 4 MiB window of a buffer (L2 hits after the first sweep) into a private 16 KiB LDS region, waits (vmcnt(0) / lgkmcnt(0)), repeats.
// Prints bytes per shader cycle per CU (s_memtime) and per microsecond.   hipcc --offload-arch=gfx950 -O3 -o lds_fill lds_fill.hip && ./lds_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int PIECES = 8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512, 1) void fill(const unsigned char* __restrict__ src, size_t window, int iters, int active, unsigned long long* cyc, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w >= active) return;
  unsigned char* mine = lds + w * (PIECES * 1024);
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)mine;
  size_t off = ((size_t)blockIdx.x * 8 + w) * (PIECES * 1024) % window;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* p = src + off + lane * 16;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024),
                                         (__attribute__((address_space(3))) void*)(mine + i * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
      u32x4 v[PIECES];
#pragma unroll
      for (int i = 0; i < PIECES; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + i * 1024);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) asm volatile("ds_write_b128 %0, %1 offset:0" ::"v"(lds_addr + i * 1024 + lane * 16), "v"(v[i]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      u32x4 v[PIECES / 2];
#pragma unroll
      for (int i = 0; i < PIECES / 2; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (2 * i) * 1024),
                                         (__attribute__((address_space(3))) void*)(mine + (2 * i) * 1024), 16, 0, 0);
        v[i] = *reinterpret_cast<const u32x4*>(p + (2 * i + 1) * 1024);
      }
#pragma unroll
      for (int i = 0; i < PIECES / 2; ++i) asm volatile("ds_write_b128 %0, %1 offset:0" ::"v"(lds_addr + (2 * i + 1) * 1024 + lane * 16), "v"(v[i]) : "memory");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    off += (size_t)256 * 8 * PIECES * 1024;
    if (off >= window) off -= window;
    if (off >= window) off %= window;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  acc += *reinterpret_cast<volatile unsigned*>(mine + lane * 4);
  if (acc == 0x12345u) sink[0] = acc;
  if (lane == 0 && w == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t window = (size_t)4 << 20, total = (size_t)64 << 20;
  unsigned char* src; unsigned long long* cyc; unsigned* sink;
  CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total));
  CHECK(hipMalloc(&cyc, 256 * 8)); CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  const char* names[3] = {"DMA (global_load_lds_dwordx4)", "REG (global_load_dwordx4 + ds_write_b128)", "MIX (alternating pieces)"};
  for (size_t win : {window, total}) {
    printf("source window %zu MiB (%s)\n", win >> 20, win == window ? "L2-resident" : "Infinity Cache / HBM");
    for (int mode = 0; mode < 3; ++mode)
      for (int active : {1, 2, 4, 8}) {
        auto launch = [&](int n) {
          if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(256), dim3(512), 8 * PIECES * 1024, 0, src, win, n, active, cyc, sink);
          else if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(256), dim3(512), 8 * PIECES * 1024, 0, src, win, n, active, cyc, sink);
          else hipLaunchKernelGGL(fill<2>, dim3(256), dim3(512), 8 * PIECES * 1024, 0, src, win, n, active, cyc, sink);
        };
        launch(50);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); launch(iters); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[256]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
        const double bytes_cu = (double)iters * active * PIECES * 1024;
        printf("  %-44s waves %d: %6.1f B/cycle/CU (s_memtime-class counter: %.0f ticks), %7.1f GB/s/CU, chip %5.2f TB/s, %6.1f ns per 1-KiB piece per wave\n", names[mode], active,
               bytes_cu / avg, avg, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * PIECES));
      }
  }
  return 0;
}
