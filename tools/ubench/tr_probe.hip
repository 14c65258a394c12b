// Semantics probe for ds_read_b64_tr_b16 (gfx950): LDS holds lds[i] = i (16-bit); lane l reads at element address l * 4
// (its own 8 bytes); prints, for every lane, the four 16-bit values it receives.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe && tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int stride_elems) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane i of a 16-lane group: row (i >> 2) of a [4][16] block, 4-element column chunk (i & 3); rows `stride_elems` apart;
  // group g at g * 1024 elements
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const int addr = g * 1024 + (i >> 2) * stride_elems + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 24}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d elements; value = group*1024 + row*stride + col\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) { int v = h[l * 4 + j]; int g = v / 1024, r = (v % 1024) / stride, c = (v % 1024) % stride; printf("  (g%d r%d c%2d)", g, r, c); }
      printf("\n");
    }
  }
  return 0;
}
