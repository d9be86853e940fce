// abn_multi_* (include/abneutral.h): one process, several MI355X.  Built on the public plan API only: one
// abn_ctx + abn_plan per device, each on a HIP stream of its own, so the devices run their shards
// concurrently from a single host thread (every abn_plan_* launch is asynchronous).  Every fit is independent
// and every random draw is a function of the GLOBAL (window, bootstrap) index, so the data path has no
// collective; the one exchange is the gather of the bootstrap tables (56 B per fit) over xGMI with RCCL:
// an in-place ncclAllGather when the shards are equal blocks, else one ncclBroadcast per (root, block).
//
// Replaces, for a host that owns the whole node, the serial window loop of src/cli/metaprofile.rs:50-72.
//
// RCCL is bound at run time (dlopen "librccl.so.1"; the symbols below) when more than one device is used: the
// single-GPU library has no link-time dependency on the 570 MB librccl, and a process that already holds
// RCCL (torch.distributed) shares its copy.  Signatures come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/abneutral.h"

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;

  bool load() {
    if (so) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) {
      why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
      return false;
    }
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(so, name);
      if (!p) {
        ok = false;
        why = std::string("librccl lacks ") + name;
      }
      return p;
    };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(so);
      so = nullptr;
    }
    return ok;
  }
};

struct Shard {
  int w0 = 0, wn = 0, b0 = 0, bn = 0;
};

// contiguous balanced split: the first (total % world) ranks get one extra item (as distributed.shard_range)
void split(int total, int world, int rank, int& start, int& count) {
  const int base = total / world, rem = total % world;
  start = rank * base + std::min(rank, rem);
  count = base + (rank < rem ? 1 : 0);
}

}  // namespace

struct abn_multi {
  int n = 0, N = 0, W = 0, S = 0, B = 0;
  bool by_boot = false;      // fewer windows than devices: every device repeats phase A, bootstraps are sharded
  bool uniform = false;      // window blocks of equal size: one in-place all-gather
  bool ran = false;
  bool gather = false;       // n > 1 (or ABN_MULTI_FORCE_RCCL=1: one device through the RCCL path, for the one-GPU test)
  std::vector<int> dev;
  std::vector<hipStream_t> stream;
  std::vector<abn_ctx*> ctx;
  std::vector<abn_plan*> plan;     // null for an empty shard
  std::vector<Shard> sh;
  std::vector<double*> full;       // [W x B x 7] on every device (n > 1), the gathered table
  std::vector<double*> local;      // by_boot: the plan's own [W x bn x 7]; else null (the plan writes into `full`)
  std::vector<ncclComm_t> comm;
  Rccl rccl;
  std::string err;
};

namespace {

int fail(abn_multi* m, int status, const std::string& msg) {
  if (m) m->err = msg;
  return status;
}

#define MHIP(m, call)                                                                      \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess) return fail((m), ABN_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

#define MNCCL(m, call)                                                                     \
  do {                                                                                     \
    ncclResult_t r__ = (call);                                                             \
    if (r__ != ncclSuccess)                                                                \
      return fail((m), ABN_ERR_HIP, std::string(#call) + ": " + (m)->rccl.GetErrorString(r__)); \
  } while (0)

int plan_fail(abn_multi* m, int i, int rc, const char* what) {
  return fail(m, rc, std::string(what) + " (device " + std::to_string(m->dev[(size_t)i]) + "): " +
                         abn_status_string(rc) + " — " + abn_last_error(m->ctx[(size_t)i]));
}

}  // namespace

extern "C" int abn_multi_rccl_available(int* ok) {
  if (!ok) return ABN_ERR_INVALID_ARG;
  Rccl r;
  *ok = r.load() ? 1 : 0;
  if (r.so) dlclose(r.so);
  return ABN_OK;
}

extern "C" const char* abn_multi_last_error(const abn_multi* m) { return m ? m->err.c_str() : "null handle"; }

extern "C" int abn_multi_destroy(abn_multi* m) {
  if (!m) return ABN_ERR_INVALID_ARG;
  for (size_t i = 0; i < m->plan.size(); ++i)
    if (m->plan[i]) abn_plan_destroy(m->plan[i]);
  for (size_t i = 0; i < m->comm.size(); ++i)
    if (m->comm[i] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comm[i]);
  for (size_t i = 0; i < m->dev.size(); ++i) {
    (void)hipSetDevice(m->dev[i]);
    if (i < m->full.size() && m->full[i]) (void)hipFree(m->full[i]);
    if (i < m->local.size() && m->local[i]) (void)hipFree(m->local[i]);
    if (i < m->ctx.size() && m->ctx[i]) abn_shutdown(m->ctx[i]);
    if (i < m->stream.size() && m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
  }
  if (m->rccl.so) dlclose(m->rccl.so);
  delete m;
  return ABN_OK;
}

// The partition of a job over the devices, host arithmetic only (no device needed): windows in contiguous balanced
// blocks when there are at least as many windows as devices, else every device takes all windows and a block of the
// bootstraps.  Same rule as alphabeta_rs_amd/distributed.py::plan_shard (tests/test_distributed_cpu.py compares them).
extern "C" int abn_multi_plan_shard(int32_t n_windows, int32_t n_boot, int32_t n_devices, int32_t device_index,
                                    int32_t* out4) {
  if (!out4 || n_windows <= 0 || n_boot <= 0 || n_devices <= 0 || device_index < 0 || device_index >= n_devices)
    return ABN_ERR_INVALID_ARG;
  int w0 = 0, wn = n_windows, b0 = 0, bn = n_boot;
  if (n_windows < n_devices) split(n_boot, n_devices, device_index, b0, bn);
  else split(n_windows, n_devices, device_index, w0, wn);
  out4[0] = w0;
  out4[1] = wn;
  out4[2] = b0;
  out4[3] = bn;
  return ABN_OK;
}

extern "C" int abn_multi_create(const int32_t* devices, int32_t n_devices, const abn_options* opts,
                                const double* generations, int32_t n_rows, int32_t n_windows, int32_t n_starts,
                                int32_t n_boot, abn_multi** out) {
  if (!out) return ABN_ERR_INVALID_ARG;
  *out = nullptr;
  if (!devices || n_devices <= 0 || !generations || n_rows <= 0 || n_windows <= 0 || n_starts <= 0 || n_boot <= 0)
    return ABN_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ABN_ERR_NO_DEVICE;
  for (int i = 0; i < n_devices; ++i) {
    if (devices[i] < 0 || devices[i] >= ndev) return ABN_ERR_INVALID_ARG;
    for (int j = 0; j < i; ++j)
      if (devices[j] == devices[i]) return ABN_ERR_INVALID_ARG;  // RCCL: one rank per GPU
  }
  abn_multi* m = new (std::nothrow) abn_multi();
  if (!m) return ABN_ERR_HIP;
  *out = m;  // handed out even on failure so that abn_multi_last_error can explain; the caller destroys it
  const size_t n = (size_t)n_devices;
  m->n = n_devices;
  m->N = n_rows;
  m->W = n_windows;
  m->S = n_starts;
  m->B = n_boot;
  m->dev.assign(devices, devices + n_devices);
  m->stream.assign(n, nullptr);
  m->ctx.assign(n, nullptr);
  m->plan.assign(n, nullptr);
  m->full.assign(n, nullptr);
  m->local.assign(n, nullptr);
  m->sh.resize(n);
  m->by_boot = n_windows < n_devices;
  m->uniform = !m->by_boot && (n_windows % n_devices == 0);
  const char* force = getenv("ABN_MULTI_FORCE_RCCL");
  m->gather = n_devices > 1 || (force && (force[0] == '1' || force[0] == '2'));
  if (force && force[0] == '2') m->uniform = false;  // ... and through the per-block broadcasts
  for (int i = 0; i < n_devices; ++i) {
    int32_t q[4];
    (void)abn_multi_plan_shard(n_windows, n_boot, n_devices, i, q);
    m->sh[(size_t)i] = Shard{q[0], q[1], q[2], q[3]};
  }
  const size_t table = (size_t)n_windows * (size_t)n_boot * 7;
  for (int i = 0; i < n_devices; ++i) {
    const Shard& s = m->sh[(size_t)i];
    MHIP(m, hipSetDevice(m->dev[(size_t)i]));
    MHIP(m, hipStreamCreateWithFlags(&m->stream[(size_t)i], hipStreamNonBlocking));
    int rc = abn_init(m->dev[(size_t)i], m->stream[(size_t)i], &m->ctx[(size_t)i]);
    if (rc) return fail(m, rc, std::string("abn_init: ") + abn_status_string(rc));
    if (m->gather) MHIP(m, hipMalloc((void**)&m->full[(size_t)i], table * sizeof(double)));
    if (s.wn == 0 || s.bn == 0) continue;  // nothing to fit here; the device still receives the table
    rc = abn_plan_create(m->ctx[(size_t)i], opts, generations, n_rows, s.wn, n_starts, s.bn, (uint32_t)s.w0,
                         (uint32_t)s.b0, &m->plan[(size_t)i]);
    if (rc) return plan_fail(m, i, rc, "abn_plan_create");
    if (m->gather) {
      if (m->by_boot) {
        MHIP(m, hipMalloc((void**)&m->local[(size_t)i], (size_t)s.wn * (size_t)s.bn * 7 * sizeof(double)));
        rc = abn_plan_bind_raw(m->plan[(size_t)i], m->local[(size_t)i]);
      } else {  // the kernels write this device's block of the gathered table in place
        rc = abn_plan_bind_raw(m->plan[(size_t)i], m->full[(size_t)i] + (size_t)s.w0 * (size_t)n_boot * 7);
      }
      if (rc) return plan_fail(m, i, rc, "abn_plan_bind_raw");
    }
  }
  if (m->gather) {
    if (!m->rccl.load()) return fail(m, ABN_ERR_HIP, m->rccl.why);
    m->comm.assign(n, nullptr);
    MNCCL(m, m->rccl.CommInitAll(m->comm.data(), n_devices, m->dev.data()));
  }
  return ABN_OK;
}

extern "C" int abn_multi_set_window_ids(abn_multi* m, const uint32_t* ids) {
  if (!m) return ABN_ERR_INVALID_ARG;
  for (int i = 0; i < m->n; ++i) {
    if (!m->plan[(size_t)i]) continue;
    const int rc = abn_plan_set_window_ids(m->plan[(size_t)i], ids ? ids + m->sh[(size_t)i].w0 : nullptr);
    if (rc) return plan_fail(m, i, rc, "abn_plan_set_window_ids");
  }
  return ABN_OK;
}

extern "C" int abn_multi_set_windows(abn_multi* m, const double* d_obs, const double* p0uu, const double* eqp,
                                     const double* eqp_weight) {
  if (!m || !d_obs || !p0uu) return m ? fail(m, ABN_ERR_INVALID_ARG, "null window data") : ABN_ERR_INVALID_ARG;
  for (int i = 0; i < m->n; ++i) {
    if (!m->plan[(size_t)i]) continue;
    const size_t w0 = (size_t)m->sh[(size_t)i].w0;
    const int rc = abn_plan_set_windows(m->plan[(size_t)i], d_obs + w0 * (size_t)m->N, p0uu + w0, eqp ? eqp + w0 : nullptr,
                                        eqp_weight ? eqp_weight + w0 : nullptr);
    if (rc) return plan_fail(m, i, rc, "abn_plan_set_windows");
  }
  return ABN_OK;
}

// the gather of the bootstrap tables, enqueued behind each device's kernels on that device's stream
static int enqueue_gather(abn_multi* m) {
  const size_t B7 = (size_t)m->B * 7;
  if (m->uniform) {  // equal window blocks, written in place: one all-gather
    const size_t count = (size_t)m->sh[0].wn * B7;
    MNCCL(m, m->rccl.GroupStart());
    for (int i = 0; i < m->n; ++i) {
      double* f = m->full[(size_t)i];
      const ncclResult_t r = m->rccl.AllGather(f + (size_t)m->sh[(size_t)i].w0 * B7, f, count, ncclDouble,
                                               m->comm[(size_t)i], m->stream[(size_t)i]);
      if (r != ncclSuccess) {
        (void)m->rccl.GroupEnd();
        return fail(m, ABN_ERR_HIP, std::string("ncclAllGather: ") + m->rccl.GetErrorString(r));
      }
    }
    MNCCL(m, m->rccl.GroupEnd());
    return ABN_OK;
  }
  // ragged window blocks, or bootstrap slices of every window: root r broadcasts each of its contiguous blocks
  for (int r = 0; r < m->n; ++r) {
    const Shard& s = m->sh[(size_t)r];
    if (s.wn == 0 || s.bn == 0) continue;
    const int nblocks = m->by_boot ? s.wn : 1;
    for (int k = 0; k < nblocks; ++k) {
      const size_t count = m->by_boot ? (size_t)s.bn * 7 : (size_t)s.wn * B7;
      const size_t dst = m->by_boot ? ((size_t)k * (size_t)m->B + (size_t)s.b0) * 7 : (size_t)s.w0 * B7;
      MNCCL(m, m->rccl.GroupStart());
      for (int i = 0; i < m->n; ++i) {
        double* recv = m->full[(size_t)i] + dst;
        const double* send = recv;  // receivers ignore it; the root's block is in place unless by_boot
        if (i == r && m->by_boot) send = m->local[(size_t)r] + (size_t)k * (size_t)s.bn * 7;
        const ncclResult_t rc = m->rccl.Broadcast(send, recv, count, ncclDouble, r, m->comm[(size_t)i], m->stream[(size_t)i]);
        if (rc != ncclSuccess) {
          (void)m->rccl.GroupEnd();
          return fail(m, ABN_ERR_HIP, std::string("ncclBroadcast: ") + m->rccl.GetErrorString(rc));
        }
      }
      MNCCL(m, m->rccl.GroupEnd());
    }
  }
  return ABN_OK;
}

extern "C" int abn_multi_run(abn_multi* m) {
  if (!m) return ABN_ERR_INVALID_ARG;
  for (int i = 0; i < m->n; ++i) {  // asynchronous: all devices work at once
    if (!m->plan[(size_t)i]) continue;
    const int rc = abn_plan_run(m->plan[(size_t)i]);
    if (rc) return plan_fail(m, i, rc, "abn_plan_run");
  }
  m->ran = true;
  if (m->gather) return enqueue_gather(m);
  return ABN_OK;
}

extern "C" int abn_multi_sync(abn_multi* m) {
  if (!m) return ABN_ERR_INVALID_ARG;
  for (int i = 0; i < m->n; ++i) {
    MHIP(m, hipSetDevice(m->dev[(size_t)i]));
    MHIP(m, hipStreamSynchronize(m->stream[(size_t)i]));
  }
  // a persistent launch that lost a chain is an error here too, not only at the download (the gathered table may be read
  // through abn_multi_raw_device_ptr)
  for (int i = 0; i < m->n; ++i)
    if (m->plan[(size_t)i])
      if (int rc = abn_plan_sync(m->plan[(size_t)i])) return plan_fail(m, i, rc, "abn_plan_sync");
  return ABN_OK;
}

extern "C" int abn_multi_shard(abn_multi* m, int32_t device_index, int32_t* out4) {
  if (!m || !out4 || device_index < 0 || device_index >= m->n) return ABN_ERR_INVALID_ARG;
  const Shard& s = m->sh[(size_t)device_index];
  out4[0] = s.w0;
  out4[1] = s.wn;
  out4[2] = s.b0;
  out4[3] = s.bn;
  return ABN_OK;
}

// HIP-event durations of one device's last run (abn_plan_kernel_ms of its plan): the launches of bench.py's roofline
extern "C" int abn_multi_kernel_ms(abn_multi* m, int32_t device_index, double* ms3) {
  if (!m || !ms3 || device_index < 0 || device_index >= m->n) return ABN_ERR_INVALID_ARG;
  const int rc = abn_plan_kernel_ms(m->plan[(size_t)device_index], ms3);
  return rc ? plan_fail(m, device_index, rc, "abn_plan_kernel_ms") : ABN_OK;
}

extern "C" int abn_multi_raw_device_ptr(abn_multi* m, int32_t device_index, void** dev_ptr) {
  if (!m || !dev_ptr || device_index < 0 || device_index >= m->n) return ABN_ERR_INVALID_ARG;
  if (m->gather) {
    *dev_ptr = m->full[(size_t)device_index];
    return ABN_OK;
  }
  return abn_plan_raw_device_ptr(m->plan[0], dev_ptr);
}

extern "C" int abn_multi_download(abn_multi* m, double* models, double* pred, double* resid, double* raw,
                                  abn_fit_info* info_a, abn_fit_info* info_b, int32_t* best_start) {
  if (!m) return ABN_ERR_INVALID_ARG;
  if (!m->ran) return fail(m, ABN_ERR_STATE, "abn_multi_run has not been called");
  int rc = abn_multi_sync(m);
  if (rc) return rc;
  const size_t N = (size_t)m->N, S = (size_t)m->S, B = (size_t)m->B;
  bool no_fit = false;
  std::vector<abn_fit_info> ib;
  for (int i = 0; i < m->n; ++i) {
    if (!m->plan[(size_t)i]) continue;
    const Shard& s = m->sh[(size_t)i];
    const size_t w0 = (size_t)s.w0;
    // by_boot: phase A is replicated (same inputs, same bits): device 0 answers for it; info_b comes in slices
    const bool lead = !m->by_boot || i == 0;
    abn_fit_info* ibp = nullptr;
    if (info_b) {
      if (m->by_boot) {
        ib.resize((size_t)s.wn * (size_t)s.bn);
        ibp = ib.data();
      } else {
        ibp = info_b + w0 * B;
      }
    }
    rc = abn_plan_download(m->plan[(size_t)i], lead && models ? models + w0 * 4 : nullptr,
                           lead && pred ? pred + w0 * N : nullptr, lead && resid ? resid + w0 * N : nullptr,
                           !m->gather ? raw : nullptr, lead && info_a ? info_a + w0 * S : nullptr, ibp,
                           lead && best_start ? best_start + w0 : nullptr);
    if (rc == ABN_ERR_NO_FINITE_FIT) {
      no_fit = true;
      rc = ABN_OK;
    }
    if (rc) return plan_fail(m, i, rc, "abn_plan_download");
    if (info_b && m->by_boot)
      for (int w = 0; w < s.wn; ++w)
        std::memcpy(info_b + (size_t)w * B + (size_t)s.b0, ib.data() + (size_t)w * (size_t)s.bn,
                    (size_t)s.bn * sizeof(abn_fit_info));
  }
  if (raw && m->gather) {  // the gathered table, from the first device
    MHIP(m, hipSetDevice(m->dev[0]));
    MHIP(m, hipMemcpyAsync(raw, m->full[0], (size_t)m->W * B * 7 * sizeof(double), hipMemcpyDeviceToHost, m->stream[0]));
    MHIP(m, hipStreamSynchronize(m->stream[0]));
  }
  if (no_fit) return fail(m, ABN_ERR_NO_FINITE_FIT, "a window has no finite start (best_start = -1): its rows are NaN");
  return ABN_OK;
}

extern "C" int abn_multi_counters(abn_multi* m, int64_t* out5) {
  if (!m || !out5) return ABN_ERR_INVALID_ARG;
  for (int k = 0; k < 5; ++k) out5[k] = 0;
  for (int i = 0; i < m->n; ++i) {
    if (!m->plan[(size_t)i]) continue;
    int64_t c[5];
    const int rc = abn_plan_counters(m->plan[(size_t)i], c);
    if (rc) return plan_fail(m, i, rc, "abn_plan_counters");
    for (int k = 0; k < 5; ++k) out5[k] += c[k];
  }
  return ABN_OK;
}
