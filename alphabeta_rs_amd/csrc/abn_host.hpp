// Host-side infrastructure shared by the translation units of libabneutral_hip.so (abn_api.hip, abn_pairwise.hip): the
// context, its device-buffer pool, error reporting.  Not part of the C-ABI (include/abneutral.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/abneutral.h"

// ------------------------------------------------------------------------------------------------
// Device-buffer pool of a context.  The drop-in entry points (abn_ab_neutral_run, abn_boot_model_run, abn_cost_batch,
// abn_fit_batch: one call per window in the reference's loops) build and drop a plan per call — seventeen hipMalloc /
// hipFree pairs, several hundred microseconds next to a 1-3 ms fit.  Freed buffers of up to kPoolBufMax bytes are kept
// (at most kPoolTotalMax in all) and handed out again, best fit within 2x; everything is stream-ordered on the
// context's stream, and abn_shutdown frees the pool.
constexpr size_t kPoolBufMax = (size_t)64 << 20, kPoolTotalMax = (size_t)256 << 20;
struct BufPool {
  std::vector<std::pair<void*, size_t>> free_list;
  size_t held = 0;
  bool closed = false;  // abn_shutdown has run: buffers of plans that outlive their context are freed, not pooled
  void* take(size_t bytes, size_t* cap) {
    size_t best = free_list.size();
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].second >= bytes && free_list[i].second <= 2 * bytes + 256 &&
          (best == free_list.size() || free_list[i].second < free_list[best].second))
        best = i;
    if (best == free_list.size()) return nullptr;
    void* p = free_list[best].first;
    *cap = free_list[best].second;
    held -= *cap;
    free_list[best] = free_list.back();
    free_list.pop_back();
    return p;
  }
  bool give(void* p, size_t cap) {
    if (closed || cap > kPoolBufMax || held + cap > kPoolTotalMax) return false;
    free_list.emplace_back(p, cap);
    held += cap;
    return true;
  }
  void clear() {
    for (auto& e : free_list) (void)hipFree(e.first);
    free_list.clear();
    held = 0;
  }
};

struct abn_ctx {
  int device = -1;
  // read from hipDeviceProp at abn_init (MI355X: 256 CUs, 160 KiB of LDS per CU); every launch geometry below is a
  // multiple of the CU count, so a partitioned device (fewer CUs per logical GPU) gets proportionally smaller launches
  int cus = 256;
  size_t lds_per_cu = 160 * 1024;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<hipStream_t> side;  // lazily created: window groups of a plan run concurrently on these
  // shared with every DevBuf drawn from it: a plan destroyed after abn_shutdown (easy from Python: ctx.close() before
  // the Plan is collected) still finds its pool — closed, so its buffers are simply freed
  std::shared_ptr<BufPool> pool = std::make_shared<BufPool>();
  std::string err;
};

// the pool DevBuf allocations of the current call draw from (set by PoolScope around the entry points)
inline thread_local std::shared_ptr<BufPool> g_pool;
struct PoolScope {
  std::shared_ptr<BufPool> prev;
  explicit PoolScope(abn_ctx* c) : prev(g_pool) { g_pool = c ? c->pool : nullptr; }
  ~PoolScope() { g_pool = prev; }
};

inline int set_err(abn_ctx* c, int status, const std::string& msg) {
  if (c) c->err = msg;
  return status;
}

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return set_err((ctx), ABN_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));    \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;          // bytes of the allocation behind p
  std::shared_ptr<BufPool> pool;  // where it came from / goes back to
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && !(pool && pool->give(p, cap))) (void)hipFree(p);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    if (count == 0) return hipSuccess;
    const size_t bytes = count * sizeof(T);
    pool = g_pool;
    if (pool) {
      if (void* q = pool->take(bytes, &cap)) {
        p = (T*)q;
        n = count;
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc((void**)&p, bytes);
    if (e == hipErrorOutOfMemory && pool && pool->held) {  // the pool may be holding what this allocation needs
      (void)hipGetLastError();
      pool->clear();
      e = hipMalloc((void**)&p, bytes);
    }
    if (e == hipSuccess) {
      n = count;
      cap = bytes;
    }
    return e;
  }
  size_t bytes() const { return n * sizeof(T); }
};
