// Micro-benchmark of a dependent kernel boundary on MI355X: what does the hand-off of a 21 MB activation tensor (the level-0 UNet
// tensors) cost between two kernels of one stream, by how the producer STORES it and by where the consumer READS it?
//
//   producer: 256 workgroups x 256 threads; workgroup b writes the contiguous chunk b (80 KiB) — plain 16-byte stores, or write-through
//             (sc1) 16-byte stores (MI355X_MICROARCH.md "publish-large": write-through leaves nothing dirty for the end-of-kernel release)
//   consumer: workgroup b reads chunk (b + shift) % 256 — shift 0: written by the same workgroup slot (same XCD, same CU if the
//             dispatcher repeats itself); shift 8: same XCD, another CU; shift 1: the next XCD
//   pair time = (P, C) x iters / iters from HIP events; P alone and C alone (warm) for reference.
//
// Answers VERDICT r03 item 1 (a) "cross-kernel XCD row affinity" and (c) "write-through epilogue stores" before any library kernel is touched.
//   hipcc --offload-arch=gfx950 -O3 -o boundary boundary.hip && ./boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NWG = 256, THREADS = 256, PER_THREAD = 20;            // 256 x 256 x 20 x 16 B = 20 MiB
constexpr size_t CHUNK16 = (size_t)THREADS * PER_THREAD;            // uint4 per workgroup chunk

template <int SC1, int WORK>
__global__ __launch_bounds__(THREADS) void producer(uint4* __restrict__ dst, unsigned seed) {
  uint4* p = dst + (size_t)blockIdx.x * CHUNK16 + threadIdx.x;
  uint4 v = make_uint4(seed + threadIdx.x, blockIdx.x, seed * 3u, 7u);
  if (WORK) {   // a "main loop" in front of the store burst (~10 us of dependent integer work), as a GEMM's epilogue has
    for (int i = 0; i < 4000; ++i) v.x = v.x * 1664525u + 1013904223u;
  }
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) {
    v.y += i;
    if (SC1) {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const u32x4 vv = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + (size_t)i * THREADS), "v"(vv) : "memory");
    }
    else p[(size_t)i * THREADS] = v;
  }
}

__global__ __launch_bounds__(THREADS) void consumer(const uint4* __restrict__ src, unsigned* __restrict__ out, int shift) {
  const int c = (blockIdx.x + shift) % NWG;
  const uint4* p = src + (size_t)c * CHUNK16 + threadIdx.x;
  uint4 v[PER_THREAD];
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) v[i] = p[(size_t)i * THREADS];
  unsigned a = 0;
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) a ^= v[i].x + v[i].y + v[i].z + v[i].w;
  if (a == 0x12345u) out[blockIdx.x * THREADS + threadIdx.x] = a;
}

template <typename F>
static float time_us(hipStream_t s, int iters, F body) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) body(i);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) body(i);
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const size_t bytes = (size_t)NWG * CHUNK16 * 16;
  // several buffers cycled so that a pair never finds its own previous output in a cache it would not have in the UNet (the forward
  // touches ~1.7 GB of weights + activations between two uses of one buffer): NBUF x 20 MiB > the 256 MiB Infinity Cache
  constexpr int NBUF = 16;
  std::vector<uint4*> buf(NBUF);
  for (auto& b : buf) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0, bytes)); }
  unsigned* out;
  CHECK(hipMalloc(&out, NWG * THREADS * 4));
  const int iters = 200;
  printf("hand-off of %.1f MiB between two dependent kernels (256 workgroups each), %d iterations, buffers cycled over %d x %.0f MiB\n",
         bytes / 1048576.0, iters, NBUF, bytes / 1048576.0);
  for (int round = 0; round < 3; ++round) {
    for (int work = 0; work < 2; ++work) {
      for (int sc1 = 0; sc1 < 2; ++sc1) {
        auto P = [&](int i) {
          uint4* d = buf[i % NBUF];
          if (work) { if (sc1) hipLaunchKernelGGL((producer<1, 1>), dim3(NWG), dim3(THREADS), 0, s, d, (unsigned)i);
                      else hipLaunchKernelGGL((producer<0, 1>), dim3(NWG), dim3(THREADS), 0, s, d, (unsigned)i); }
          else { if (sc1) hipLaunchKernelGGL((producer<1, 0>), dim3(NWG), dim3(THREADS), 0, s, d, (unsigned)i);
                 else hipLaunchKernelGGL((producer<0, 0>), dim3(NWG), dim3(THREADS), 0, s, d, (unsigned)i); }
        };
        const float tp = time_us(s, iters, [&](int i) { P(i); });
        printf("round %d  work %d  %-5s  P alone %6.2f us", round, work, sc1 ? "sc1" : "plain", tp);
        for (int shift : {0, 8, 1, 4}) {
          const float t = time_us(s, iters, [&](int i) {
            P(i);
            hipLaunchKernelGGL(consumer, dim3(NWG), dim3(THREADS), 0, s, (const uint4*)buf[i % NBUF], out, shift);
          });
          printf("   P+C(shift %d) %6.2f", shift, t);
        }
        printf("\n");
      }
    }
    const float tc = time_us(s, iters, [&](int i) { hipLaunchKernelGGL(consumer, dim3(NWG), dim3(THREADS), 0, s, (const uint4*)buf[0], out, 0); });
    const float tcc = time_us(s, iters, [&](int i) { hipLaunchKernelGGL(consumer, dim3(NWG), dim3(THREADS), 0, s, (const uint4*)buf[i % NBUF], out, 0); });
    printf("round %d  C alone, one warm buffer %6.2f us; C alone, cycled (cold) buffers %6.2f us\n", round, tc, tcc);
  }
  return 0;
}
