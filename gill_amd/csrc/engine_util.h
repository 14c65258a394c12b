// Host-side helpers shared by the stage engines: owned-HBM pool, named-weight lookup,
// creation-time weight conversion.
#pragma once
#include "../../include/gill_amd.h"
#include "ops.h"
#include <string>
#include <unordered_map>
#include <vector>

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int alloc(size_t n) {
    if (n == 0) n = 16;
    GILL_CHECK_HIP(hipMalloc(&p, n));
    bytes = n;
    return 0;
  }
  int alloc_zero(size_t n, hipStream_t s) {
    GILL_TRY(alloc(n));
    GILL_CHECK_HIP(hipMemsetAsync(p, 0, bytes, s));
    return 0;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// Every byte an engine owns in HBM comes from here and is released at destroy.
struct DevPool {
  std::vector<void*> ptrs;
  size_t total = 0;
  template <typename T>
  int alloc(T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    size_t n = count * sizeof(T);
    if (n == 0) n = 16;
    n = (n + 255) & ~(size_t)255;
    GILL_CHECK_HIP(hipMalloc(&p, n));
    if (zero) GILL_CHECK_HIP(hipMemset(p, 0, n));
    ptrs.push_back(p);
    total += n;
    *out = (T*)p;
    return 0;
  }
  ~DevPool() { for (void* p : ptrs) (void)hipFree(p); }
};

// An engine's private launch stream, fenced to the caller's stream with two events: enter() makes the private stream
// wait for everything the caller enqueued so far, leave() makes the caller's stream wait for the engine's work.
struct StreamFence {
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  int enter(hipStream_t caller) {
    if (!stream) {
      GILL_CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      GILL_CHECK_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
      GILL_CHECK_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
    }
    GILL_CHECK_HIP(hipEventRecord(ev_in, caller));
    GILL_CHECK_HIP(hipStreamWaitEvent(stream, ev_in, 0));
    return 0;
  }
  int leave(hipStream_t caller) {
    GILL_CHECK_HIP(hipEventRecord(ev_out, stream));
    GILL_CHECK_HIP(hipStreamWaitEvent(caller, ev_out, 0));
    return 0;
  }
  ~StreamFence() {
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (stream) (void)hipStreamDestroy(stream);
  }
  StreamFence() = default;
  StreamFence(const StreamFence&) = delete;
  StreamFence& operator=(const StreamFence&) = delete;
};

struct WeightTable {
  std::unordered_map<std::string, const gill_tensor*> map;
  WeightTable(const gill_tensor* w, int n) { for (int i = 0; i < n; ++i) if (w[i].name) map[w[i].name] = &w[i]; }
  const gill_tensor* find(const std::string& name) const {
    auto it = map.find(name);
    return it == map.end() ? nullptr : it->second;
  }
  static int64_t numel(const gill_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
  }
  // look up + check element count
  int get(const std::string& name, int64_t expect_numel, const gill_tensor** out) const {
    const gill_tensor* t = find(name);
    if (!t) { gill_set_error("missing weight tensor: " + name); return -3; }
    if (numel(t) != expect_numel) {
      gill_set_error("weight tensor " + name + " has " + std::to_string(numel(t)) + " elements, expected " +
                     std::to_string(expect_numel));
      return -3;
    }
    if (t->dtype < 0 || t->dtype > 2) { gill_set_error("weight tensor " + name + " has an unsupported dtype"); return -3; }
    *out = t;
    return 0;
  }
};

// dst row of each source row for the GEGLU projection weight: 16-row value blocks interleaved with
// the 16-row gate blocks of the same output columns (see gemm.hip ACT_GEGLU epilogue).
static inline std::vector<int32_t> geglu_row_permutation(int inner) {
  std::vector<int32_t> m(2 * (size_t)inner);
  for (int o = 0; o < inner; ++o) {
    m[o] = (o / 16) * 32 + (o % 16);
    m[inner + o] = (o / 16) * 32 + 16 + (o % 16);
  }
  return m;
}

// named tensor -> owned bf16 copy [numel]
static inline int load_bf16(const WeightTable& wt, DevPool& pool, const std::string& name, int64_t numel, bf16_t** out,
                            hipStream_t s) {
  const gill_tensor* t;
  GILL_TRY(wt.get(name, numel, &t));
  GILL_TRY(pool.alloc(out, (size_t)numel, false));
  return convert_to_bf16_launch(t->data, t->dtype, numel, *out, s);
}
static inline int load_f32(const WeightTable& wt, DevPool& pool, const std::string& name, int64_t numel, float** out,
                           hipStream_t s) {
  const gill_tensor* t;
  GILL_TRY(wt.get(name, numel, &t));
  GILL_TRY(pool.alloc(out, (size_t)numel, false));
  return convert_to_f32_launch(t->data, t->dtype, numel, *out, s);
}
