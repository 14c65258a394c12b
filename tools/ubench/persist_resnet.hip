// Prototype (round 6): ONE persistent launch per ResnetBlock2D at UNet levels 2-3 — conv1 -> [+ time-embedding row, norm2, SiLU]
// -> conv2 [+ residual, the next block's GroupNorm] — against the launch chains it would replace.  Go / no-go: profiles/r06_persistent_phase.md.
//
// The phases ARE the product kernel: this file includes gill_amd/csrc/gemm.hip and calls its body, gemm_tile<8,160,1,7,3,64,2> (3x3 implicit GEMM on
// the 128 x 160 ping-pong tile, split-K, in-kernel finish "COOP" = splitk_finish_unit, the very code of the reducer launch), twice in one workgroup
// with a SEAM between them: a phase-2 workgroup may read the rows of its sample group only once every phase-1 workgroup that finishes units of that
// group has published them (one agent-scope release) and arrived on the group's counter; it polls that counter (relaxed, s_sleep), takes one
// agent-scope acquire, and goes on.  With `prefetch` it first issues the WEIGHT pieces of its first two ring stages — weights do not depend on the
// previous phase — so they fly while it waits (MI355X_MICROARCH.md "prefetch-credit").  All four variants produce bit-identical tensors:
//   A  conv1 | reducer+GN | conv2 | reducer+GN      four launches   (the round-5 dataflow, what the engine runs)
//   B  conv1+finish | conv2+finish                  two launches    (COOP split-K finish, EPI 7: GILL_GEMM_COOP=2)
//   C0 [conv1+finish -> seam -> conv2+finish]       one launch, the loader starts at the seam
//   C1 the same with the weight pieces run ahead across the seam
// Weights rotate over NW sets (> 256 MiB in all) so that every repetition streams them from HBM, as inside a UNet forward.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-c++20-extensions -o tools/ubench/persist_resnet tools/ubench/persist_resnet.hip
//   ./tools/ubench/persist_resnet [level: 3 | 2]
#include "../../gill_amd/csrc/gemm.hip"
#include <vector>
#include <string.h>

void gill_set_error(const std::string& msg) { fprintf(stderr, "gill error: %s\n", msg.c_str()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PF>
__global__ __launch_bounds__(512, 1) void resnet_persist_kernel(const GemmDev d1, const GemmDev d2, unsigned* seam, unsigned seam_target, int seam_mtg) {
  GemmTileCtx c1;
  c1.bx = blockIdx.x; c1.by = blockIdx.y; c1.bz = 0; c1.gdx = gridDim.x; c1.gdy = gridDim.y;
  c1.seam_out = seam;
  gemm_tile<8, 160, 1, 7, 3, 64, 2, 1>(d1, c1);
  __syncthreads();      // (phase 1's LDS scratch is dead before phase 2's first ring stage is filled)
  GemmTileCtx c2 = c1;
  c2.seam_out = nullptr; c2.seam_in = seam; c2.seam_in_target = seam_target; c2.seam_in_mtg = seam_mtg; c2.prefetch = PF;
  gemm_tile<8, 160, 1, 7, 3, 64, 2, 1>(d2, c2);
}

__global__ void fill_bf16_kernel(bf16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = f2bf(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
  }
}
__global__ void fill_f32_kernel(float* p, size_t n, uint32_t seed, float scale, float base) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = base + ((float)(x & 0xffff) / 32768.f - 1.f) * scale;
  }
}
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); return p; }
static void fillb(bf16_t* p, size_t n, uint32_t seed, float scale) { hipLaunchKernelGGL(fill_bf16_kernel, dim3(1024), dim3(256), 0, 0, p, n, seed, scale); }
static void fillf(float* p, size_t n, uint32_t seed, float scale, float base) { hipLaunchKernelGGL(fill_f32_kernel, dim3(256), dim3(256), 0, 0, p, n, seed, scale, base); }

int main(int argc, char** argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 3;
  const int B = 8, HW = level == 3 ? 8 : 16, C = 1280, S = level == 3 ? 8 : 2;
  const int M = B * HW * HW, K = 9 * C, rpb = HW * HW;
  const int NW = 10, R = 40;
  printf("ResnetBlock2D at UNet level %d: %d samples x %dx%d x %d channels (M = %d rows), 3x3 convs K = %d, split-K %d, 128 x 160 tiles -> %d workgroups\n",
         level, B, HW, HW, C, M, K, S, (M / 128) * (C / 160) * S);
  // operands
  bf16_t* n1 = dalloc<bf16_t>((size_t)M * C); bf16_t* x = dalloc<bf16_t>((size_t)M * C);
  fillb(n1, (size_t)M * C, 1, 1.0f); fillb(x, (size_t)M * C, 2, 1.0f);
  std::vector<bf16_t*> W1(NW), W2(NW);
  for (int i = 0; i < NW; ++i) {
    W1[i] = dalloc<bf16_t>((size_t)C * K); W2[i] = dalloc<bf16_t>((size_t)C * K);
    fillb(W1[i], (size_t)C * K, 10 + i, 0.01f); fillb(W2[i], (size_t)C * K, 30 + i, 0.01f);
  }
  float* b1 = dalloc<float>(C); float* b2 = dalloc<float>(C); float* temb = dalloc<float>((size_t)B * C);
  float* g2 = dalloc<float>(C); float* be2 = dalloc<float>(C); float* g3 = dalloc<float>(C); float* be3 = dalloc<float>(C);
  fillf(b1, C, 3, 0.1f, 0.f); fillf(b2, C, 4, 0.1f, 0.f); fillf(temb, (size_t)B * C, 5, 0.3f, 0.f);
  fillf(g2, C, 6, 0.2f, 1.f); fillf(be2, C, 7, 0.1f, 0.f); fillf(g3, C, 8, 0.2f, 1.f); fillf(be3, C, 9, 0.1f, 0.f);
  float* ws1 = dalloc<float>((size_t)S * M * C); float* ws2 = dalloc<float>((size_t)S * M * C);
  const int NV = 4;
  bf16_t* n2[NV]; bf16_t* out[NV]; bf16_t* y3[NV];
  for (int v = 0; v < NV; ++v) { n2[v] = dalloc<bf16_t>((size_t)M * C); out[v] = dalloc<bf16_t>((size_t)M * C); y3[v] = dalloc<bf16_t>((size_t)M * C); }
  const int nctr = gemm_coop_counters(GemmArgs{.M = M, .N = C});
  const int slots = 2 * nctr + 64;       // per repetition: conv1's counters, conv2's, the seam counters
  unsigned* ctr = dalloc<unsigned>((size_t)(R + 2) * slots);
  CK(hipDeviceSynchronize());

  auto conv_args = [&](int which, int v, int wset, unsigned* c) {
    GemmArgs g;
    g.conv = 1; g.IH = HW; g.IW = HW; g.OH = HW; g.OW = HW; g.Cin = C; g.stride = 1;
    g.M = M; g.N = C; g.K = K; g.K1 = C; g.rows_per_batch = rpb; g.splitk = S; g.ldc = C;
    g.fn_cg = C / 32; g.fn_eps = which == 0 ? 1e-5f : 1e-6f; g.fn_silu = which == 0 ? 1 : 0;
    if (which == 0) {      // conv1: norm1(x) -> + bias + time-embedding row -> norm2 + SiLU (the raw tensor has no other reader)
      g.A = n1; g.W = W1[wset]; g.bias = b1; g.rowvec = temb; g.rowvec_bstride = C; g.ws = ws1;
      g.fn_Y = n2[v]; g.fn_gamma = g2; g.fn_beta = be2;
    } else {               // conv2: -> + bias + residual x -> raw block output + the next block's GroupNorm
      g.A = n2[v]; g.W = W2[wset]; g.bias = b2; g.resid = x; g.ldr = C; g.ws = ws2; g.C = out[v]; g.ldc = C;
      g.fn_Y = y3[v]; g.fn_gamma = g3; g.fn_beta = be3;
    }
    g.coop_ctr = c; g.coop_splitk = c != nullptr;
    return g;
  };
  auto make_dev = [&](const GemmArgs& a) {      // gemm_launch_bn<160>()'s plan for this launch
    GemmDev d; d.a = a; d.zero = gill_zero_page();
    d.tiles_n = C / 160; d.npw = 1; d.groups_n = d.tiles_n; d.nwv = 8; d.mi = 2; d.kt = 64;
    d.ksteps = K / 64; d.ksteps_per_split = cdiv(d.ksteps, S); d.tiles_m = M / 128; d.n_major = 1;
    xcd_block_pick(a, d, 128, 160, 1);
    return d;
  };
  const int smem = 3 * (128 * 64 + 160 * 64) * (int)sizeof(bf16_t);
  CK(hipFuncSetAttribute((const void*)resnet_persist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipFuncSetAttribute((const void*)resnet_persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  const int mtg = rpb > 128 ? rpb / 128 : 1;
  const unsigned seam_target = (unsigned)(mtg * (C / 160) * S);
  const dim3 grid((M / 128) * (C / 160), S, 1);

  auto run = [&](int variant, int rep) -> int {
    const int wset = rep % NW;
    unsigned* c = ctr + (size_t)(rep % (R + 2)) * slots;
    if (variant == 0) {
      GemmArgs a1 = conv_args(0, 0, wset, nullptr), a2 = conv_args(1, 0, wset, nullptr);
      if (gemm_launch(a1, 0) || gemm_launch(a2, 0)) return 1;
    } else if (variant == 1) {
      GemmArgs a1 = conv_args(0, 1, wset, c), a2 = conv_args(1, 1, wset, c + nctr);
      if (gemm_launch(a1, 0) || gemm_launch(a2, 0)) return 1;
    } else {
      GemmArgs a1 = conv_args(0, variant, wset, c), a2 = conv_args(1, variant, wset, c + nctr);
      if (!gemm_coop_ok(a1) || !gemm_coop_ok(a2)) { fprintf(stderr, "geometry not taken by the in-kernel finish\n"); return 1; }
      GemmDev d1 = make_dev(a1), d2 = make_dev(a2);
      if (variant == 2) hipLaunchKernelGGL((resnet_persist_kernel<0>), grid, dim3(512), smem, 0, d1, d2, c + 2 * nctr, seam_target, mtg);
      else hipLaunchKernelGGL((resnet_persist_kernel<1>), grid, dim3(512), smem, 0, d1, d2, c + 2 * nctr, seam_target, mtg);
    }
    return 0;
  };
  const char* names[NV] = {"A  four launches (conv | reducer+GN | conv | reducer+GN)", "B  two launches (in-kernel split-K finish)",
                           "C0 one persistent launch, loader starts at the seam", "C1 one persistent launch, weight pieces run ahead across the seam"};
  // correctness first: every variant on weight set 0, outputs compared bit for bit with A
  std::vector<uint16_t> ref_out((size_t)M * C), ref_y3((size_t)M * C), got((size_t)M * C);
  for (int v = 0; v < NV; ++v) {
    CK(hipMemset(ctr, 0, sizeof(unsigned) * (size_t)(R + 2) * slots));
    CK(hipMemset(out[v], 0xff, sizeof(bf16_t) * (size_t)M * C)); CK(hipMemset(y3[v], 0xff, sizeof(bf16_t) * (size_t)M * C));
    if (run(v, 0)) return 1;
    CK(hipDeviceSynchronize());
    if (v == 0) {
      CK(hipMemcpy(ref_out.data(), out[0], sizeof(bf16_t) * (size_t)M * C, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ref_y3.data(), y3[0], sizeof(bf16_t) * (size_t)M * C, hipMemcpyDeviceToHost));
      size_t nan = 0; for (size_t i = 0; i < ref_y3.size(); ++i) nan += ((ref_y3[i] & 0x7f80) == 0x7f80);
      printf("reference (A): %zu non-finite outputs\n", nan);
    } else {
      CK(hipMemcpy(got.data(), out[v], sizeof(bf16_t) * (size_t)M * C, hipMemcpyDeviceToHost));
      const bool e1 = memcmp(got.data(), ref_out.data(), sizeof(bf16_t) * (size_t)M * C) == 0;
      CK(hipMemcpy(got.data(), y3[v], sizeof(bf16_t) * (size_t)M * C, hipMemcpyDeviceToHost));
      const bool e2 = memcmp(got.data(), ref_y3.data(), sizeof(bf16_t) * (size_t)M * C) == 0;
      printf("variant %d vs A: raw output %s, normalised output %s\n", v, e1 ? "BIT-IDENTICAL" : "DIFFERS", e2 ? "BIT-IDENTICAL" : "DIFFERS");
    }
  }
  // timing: R repetitions back to back per variant (counters zeroed outside the timed region), three rounds
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round) {
    for (int v = 0; v < NV; ++v) {
      CK(hipMemset(ctr, 0, sizeof(unsigned) * (size_t)(R + 2) * slots));
      for (int r = 0; r < 3; ++r) if (run(v, R + (r & 1))) return 1;      // warm-up (the two spare counter slots)
      CK(hipDeviceSynchronize());
      CK(hipMemset(ctr, 0, sizeof(unsigned) * (size_t)(R + 2) * slots));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < R; ++r) if (run(v, r)) return 1;
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("round %d  %-70s %7.1f us per resnet\n", round, names[v], ms * 1e3f / R);
    }
  }
  return 0;
}
