// Shared device/host helpers for the gill_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GILL_WAVE 64


// KERNARG WARM-UP.  A kernel's arguments are s_load'ed from the kernarg segment, which was written for this launch: the first touch of each of its
// 64-byte lines is a scalar-cache miss served from memory (~0.5-1 us under load), and hipcc issues those loads where the values are first needed —
// behind branches and integer divisions — so an argument struct of several hundred bytes costs several SERIALISED misses before the first memory
// instruction of the kernel is out (the "set-up" part of every launch's fixed cost, profiles/r05_opt_stream64.md).  First statement of a kernel:
// one dword of every line, all in flight at once, one wait — the compiler's own loads then hit the scalar cache.  BYTES = explicit argument bytes
// (only offsets inside them are touched; the hidden block behind them shares their last line or follows it); at most 10 lines.
template <int OFF>
__device__ __forceinline__ unsigned kernarg_touch(uint64_t ka) {
  unsigned t;
  asm volatile("s_load_dword %0, %1, %2" : "=s"(t) : "s"(ka), "i"(OFF));
  return t;
}
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int L = (BYTES + 63) / 64;
  static_assert(L >= 1 && L <= 10, "kernarg_warm: 1 .. 640 argument bytes");
  unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0, t9 = 0;
  t0 = kernarg_touch<0>(ka);
  if constexpr (L > 1) t1 = kernarg_touch<64>(ka);
  if constexpr (L > 2) t2 = kernarg_touch<128>(ka);
  if constexpr (L > 3) t3 = kernarg_touch<192>(ka);
  if constexpr (L > 4) t4 = kernarg_touch<256>(ka);
  if constexpr (L > 5) t5 = kernarg_touch<320>(ka);
  if constexpr (L > 6) t6 = kernarg_touch<384>(ka);
  if constexpr (L > 7) t7 = kernarg_touch<448>(ka);
  if constexpr (L > 8) t8 = kernarg_touch<512>(ka);
  if constexpr (L > 9) t9 = kernarg_touch<576>(ka);
  // (the destination registers stay allocated up to the wait: inputs of this statement)
  asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7), "s"(t8), "s"(t9) : "memory");
}

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16 through the native type: hipcc lowers these to gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even,
// NaN preserved) — one VALU op per PAIR instead of ~7 integer ops per element of a hand-rolled rounding.
typedef __attribute__((ext_vector_type(2))) float gill_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gill_bf16x2;

__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const gill_f32x2 v = {lo, hi};
  const gill_bf16x2 b = __builtin_convertvector(v, gill_bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k + 15), result in every lane, on the VALU (v_add_f32 with a DPP operand: quad
// permutes, then the row's half-mirror and mirror) instead of four ds_bpermute round trips through the LDS crossbar.  Fixed order.
template <int CTRL>
__device__ __forceinline__ float dpp_row_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_row_mov<0x4E>(v);    // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
  v += dpp_row_mov<0x141>(v);   // row_half_mirror: lane i <-> 7 - i inside each 8 lanes
  v += dpp_row_mov<0x140>(v);   // row_mirror: lane i <-> 15 - i
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-form GELU 0.5 x (1 + erf(x / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-level:
// two orders below the bf16 output resolution) — 1 rcp + 1 exp + a 5-term Horner chain instead of libm erff's
// branchy ~40-instruction path, which dominated the GEGLU GEMM epilogue (K = 320: epilogue ~ main loop).
__device__ __forceinline__ float erf_as(float z) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, z);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// The same function where its arithmetic runs alone on the SIMD (ffn.hip: one wave per SIMD, the GELUs of a chunk sit in front of the
// second GEMM): x * Phi(x) with the normal CDF as a logistic of an odd quintic, Phi(x) ~ 1 / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))),
// coefficients fitted (p = 8 norm of |x Phi(x) - gelu(x)| on [-8, 8]): |error| <= 2.7e-5 absolute over all x — two orders below the bf16
// resolution of the product it feeds.  x is clamped to +-7 inside the polynomial (c2 < 0 turns it over beyond |x| ~ 10; the logistic is
// saturated to 2e-11 by then).  8 VALU + exp2 + rcp instead of 14 + exp2 + rcp.  (In the GEGLU GEMM's epilogue, which a co-resident
// workgroup's main loop hides, it measured no gain: the engine keeps gelu_erf there.)
__device__ __forceinline__ float gelu_logistic(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
  const float x2 = xc * xc;
  float p = fmaf(0.0010187963489443064f, x2, -0.10680364072322845f);    // -log2(e) * {c2, c1, c0}
  p = fmaf(p, x2, -2.301090717315674f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xc));
}
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));   // v_exp_f32 + v_rcp_f32
}

// saturate to the e4m3 range but keep NaN a NaN (fminf / fmaxf return the non-NaN operand: a plain clamp would launder a
// non-finite activation into +-448 before the fp8 convolution, out of sight of every isfinite check downstream)
__device__ __forceinline__ float clamp_fp8_keep_nan(float x) { return (x != x) ? x : fminf(fmaxf(x, -448.f), 448.f); }

// ---- host side error plumbing (thread-local message, int status across the C ABI) ----
void gill_set_error(const std::string& msg);

#define GILL_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,         \
               hipGetErrorString(_e));                                                    \
      gill_set_error(_b);                                                                 \
      return -1;                                                                          \
    }                                                                                     \
  } while (0)

#define GILL_REQUIRE(cond, msg)                                                           \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s:%d: requirement failed (%s): %s", __FILE__, __LINE__,  \
               #cond, msg);                                                               \
      gill_set_error(_b);                                                                 \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

#define GILL_TRY(expr)                                                                    \
  do {                                                                                    \
    int _r = (expr);                                                                      \
    if (_r != 0) return _r;                                                               \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
