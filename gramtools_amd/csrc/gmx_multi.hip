// gmx_multi.hip — several GPUs: reads shard, the index is replicated, ONE exchange at the end.
//
// The reference's unit of parallelism is one read inside an OpenMP loop (quasimap.cpp:90-118) over shared
// coverage structures. Here every GPU owns an engine with its own accumulators; the reads of a call are dealt
// out by global read index (the per-read seeds are the master stream's, so the result does not depend on the
// number of GPUs), and at the end the uint32 accumulator blocks are summed: one RCCL all-reduce over xGMI of the
// fused block (coverage + the five read counters as 16-bit limbs), plus the exchange of the grouped log of sites
// with more than 8 alleles (counted records: small). uint16 wrap / saturation are functions of the totals, so the
// result equals the single-thread reference.
//
// One exchange routine (gmx_exchange) serves both users:
//   gmx_group  N engines in ONE process (the `gram` executable: one host thread per GPU)       — ncclCommInitAll
//   gmx_comm   one engine per process (bench.py under torch.distributed.run, any launcher)    — ncclCommInitRank
// RCCL is loaded with dlopen at first use: a single-GPU run never touches it. Without RCCL (or when two engines
// of a group sit on the same device, as in the 1-GPU test) the group falls back to peer copies and an add kernel.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/gmx.h"
#include "gmx_internal.h"

// One feeder thread per GPU, each on a core of its own (the i-th CPU the process may run on): a feeder is a short burst of
// launches per batch and then sleeps on its slot event (gmx_engine.hip: blocking events); kept on one core it does not
// chase the parser threads around the socket. GMX_PIN_FEEDERS=0 switches the pinning off. The reference's unit of
// parallelism for context: one OpenMP loop over the reads of a batch (quasimap.cpp:90).
static void gmx_pin_feeder(size_t i) {
  static const bool on = !(getenv("GMX_PIN_FEEDERS") && getenv("GMX_PIN_FEEDERS")[0] == '0');
  if (!on) return;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
  const int n_allowed = CPU_COUNT(&allowed);
  if (n_allowed < 2) return;
  int want = (int)(i % (size_t)n_allowed), seen = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &allowed)) continue;
    if (seen++ == want) {
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(c, &one);
      (void)sched_setaffinity(0, sizeof(one), &one);
      return;
    }
  }
}

#define HIP_TRY(expr)                                                       \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      gmx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
      return GMX_EHIP;                                                      \
    }                                                                       \
  } while (0)

namespace {

struct Rccl {
  void *so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  // a process that already holds an RCCL (torch's) gets that one through the soname
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.so) break;
  }
  if (!r.so) return r;
  auto sym = [&](const char *n) { return dlsym(r.so, n); };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.GroupStart &&
         r.GroupEnd && r.GetErrorString;
  return r;
}

#define NCCL_TRY(expr)                                                                   \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess) {                                                             \
      gmx_set_error(std::string(#expr) + ": " + rccl().GetErrorString(_r));              \
      return GMX_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

__global__ void gmx_add_u32_kernel(uint32_t *dst, const uint32_t *src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// One member of an exchange: an engine, its communicator (null = no RCCL), the stream the exchange is enqueued on.
struct Member {
  gmx_engine *e = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  GmxEngineRaw raw{};
};

// a device allocation that goes away with its scope (the exchange's temporaries, on every error path too)
struct DevBuf {
  void *p = nullptr;
  int device = 0;
  ~DevBuf() {
    if (!p) return;
    (void)hipSetDevice(device);
    (void)hipFree(p);
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
};

// The grouped log, member by member: all-gathered in two steps (sizes, then the payloads padded to the largest), and
// every member ends with the sum of all logs in its engine (gmx_grouped_log_merge_gathered, shared with the tests).
// Everything that can fail on this rank alone (export, allocation) happens BEFORE the first collective, so a failing rank
// returns without leaving its peers inside an all-gather.
int exchange_logs_rccl(std::vector<Member> &ms, int world) {
  Rccl &r = rccl();
  const size_t n = ms.size();
  std::vector<std::vector<uint32_t>> mine(n);
  std::vector<DevBuf> d_sizes(n);
  for (size_t i = 0; i < n; ++i) {
    int rc = gmx_engine_log_export(ms[i].e, mine[i]);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ms[i].raw.device));
    d_sizes[i].device = ms[i].raw.device;
    HIP_TRY(hipMalloc(&d_sizes[i].p, (size_t)(world + 1) * 8));
    const uint64_t sz = mine[i].size();
    HIP_TRY(hipMemcpy(d_sizes[i].as<uint64_t>() + world, &sz, 8, hipMemcpyHostToDevice));
  }
  NCCL_TRY(r.GroupStart());
  for (size_t i = 0; i < n; ++i)
    NCCL_TRY(r.AllGather(d_sizes[i].as<uint64_t>() + world, d_sizes[i].as<uint64_t>(), 1, ncclUint64, ms[i].comm, ms[i].stream));
  NCCL_TRY(r.GroupEnd());
  std::vector<uint64_t> sizes(world);
  uint64_t pad = 0;
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(ms[i].raw.device));
    HIP_TRY(hipStreamSynchronize(ms[i].stream));
    HIP_TRY(hipMemcpy(sizes.data(), d_sizes[i].as<uint64_t>(), (size_t)world * 8, hipMemcpyDeviceToHost));
  }
  for (uint64_t s : sizes) pad = std::max(pad, s);
  if (pad == 0) return GMX_OK;  // (every rank sees the same sizes: all of them return here or none)
  std::vector<DevBuf> d_buf(n);
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(ms[i].raw.device));
    d_buf[i].device = ms[i].raw.device;
    HIP_TRY(hipMalloc(&d_buf[i].p, (size_t)(world + 1) * pad * 4));
    if (!mine[i].empty())
      HIP_TRY(hipMemcpy(d_buf[i].as<uint32_t>() + (size_t)world * pad, mine[i].data(), mine[i].size() * 4, hipMemcpyHostToDevice));
  }
  NCCL_TRY(r.GroupStart());
  for (size_t i = 0; i < n; ++i)
    NCCL_TRY(r.AllGather(d_buf[i].as<uint32_t>() + (size_t)world * pad, d_buf[i].as<uint32_t>(), pad, ncclUint32, ms[i].comm, ms[i].stream));
  NCCL_TRY(r.GroupEnd());
  std::vector<uint32_t> all((size_t)world * pad), merged;
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(ms[i].raw.device));
    HIP_TRY(hipStreamSynchronize(ms[i].stream));
    HIP_TRY(hipMemcpy(all.data(), d_buf[i].as<uint32_t>(), all.size() * 4, hipMemcpyDeviceToHost));
    const int64_t words = gmx_grouped_log_merge_gathered(all.data(), sizes.data(), world, pad, nullptr, 0);
    if (words < 0) return (int)words;
    merged.assign((size_t)words, 0);
    if (words && gmx_grouped_log_merge_gathered(all.data(), sizes.data(), world, pad, merged.data(), (uint64_t)words) < 0) return GMX_EINVAL;
    int rc = gmx_engine_log_import(ms[i].e, merged.data(), merged.size(), true);
    if (rc) return rc;
  }
  return GMX_OK;
}

// THE exchange. RCCL: counters -> limbs, one in-place all-reduce(sum) of the fused block per member, limbs -> counters;
// then the grouped logs when the index has sites that use them.
int gmx_exchange(std::vector<Member> &ms, int world) {
  Rccl &r = rccl();
  for (auto &m : ms) {
    int rc = gmx_coverage_reduce_begin(m.e, m.stream);
    if (rc) return rc;
  }
  NCCL_TRY(r.GroupStart());
  for (auto &m : ms) NCCL_TRY(r.AllReduce(m.raw.d_fused, m.raw.d_fused, m.raw.n_fused, ncclUint32, ncclSum, m.comm, m.stream));
  NCCL_TRY(r.GroupEnd());
  for (auto &m : ms) {
    int rc = gmx_coverage_reduce_end(m.e, m.stream);
    if (rc) return rc;
  }
  if (ms[0].raw.log_sites) return exchange_logs_rccl(ms, world);
  return GMX_OK;
}

// Without RCCL (single process only): gather onto member 0 with peer copies and an add kernel, copy the totals back.
int gmx_exchange_peer(std::vector<Member> &ms) {
  const size_t n = ms.size();
  for (auto &m : ms) {
    int rc = gmx_coverage_reduce_begin(m.e, m.stream);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(m.raw.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
  }
  const int root = ms[0].raw.device;
  HIP_TRY(hipSetDevice(root));
  DevBuf tmp_buf;
  tmp_buf.device = root;
  HIP_TRY(hipMalloc(&tmp_buf.p, ms[0].raw.n_fused * 4));
  uint32_t *tmp = tmp_buf.as<uint32_t>();
  for (size_t i = 1; i < n; ++i) {
    HIP_TRY(hipMemcpyPeer(tmp, root, ms[i].raw.d_fused, ms[i].raw.device, ms[0].raw.n_fused * 4));
    hipLaunchKernelGGL(gmx_add_u32_kernel, dim3(1024), dim3(256), 0, ms[0].stream, ms[0].raw.d_fused, tmp, ms[0].raw.n_fused);
    HIP_TRY(hipStreamSynchronize(ms[0].stream));
  }
  for (size_t i = 1; i < n; ++i) HIP_TRY(hipMemcpyPeer(ms[i].raw.d_fused, ms[i].raw.device, ms[0].raw.d_fused, root, ms[0].raw.n_fused * 4));
  for (auto &m : ms) {
    int rc = gmx_coverage_reduce_end(m.e, m.stream);
    if (rc) return rc;
  }
  if (ms[0].raw.log_sites) {  // host merge of the counted records
    std::vector<std::vector<uint32_t>> logs(n);
    for (size_t i = 0; i < n; ++i) {
      int rc = gmx_engine_log_export(ms[i].e, logs[i]);
      if (rc) return rc;
    }
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) {
        int rc = gmx_engine_log_import(ms[i].e, logs[j].data(), logs[j].size(), j == 0);
        if (rc) return rc;
      }
  }
  return GMX_OK;
}

}  // namespace

struct gmx_group {
  std::vector<Member> ms;
  bool use_rccl = false;
};

struct gmx_comm {
  Member m;
  int world = 1, rank = 0;
};

extern "C" {

int gmx_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
} GMX_GUARD_INT("gmx_device_count")

// (gmx.h) Brings the HIP runtime, the device's context and this library's code objects up — 150-250 ms that a caller can
// spend on another thread while it loads its index from disk.
__global__ void gmx_warmup_kernel(uint32_t *p) {
  if (p) *p = 1u;
}
int gmx_device_warmup(int device) try {
  if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) {
    gmx_set_error(std::string("gmx_device_warmup: device ") + std::to_string(device) + ": " + hipGetErrorString(hipGetLastError()));
    return GMX_ENODEV;
  }
  hipLaunchKernelGGL(gmx_warmup_kernel, dim3(1), dim3(1), 0, nullptr, (uint32_t *)nullptr);
  if (hipDeviceSynchronize() != hipSuccess) {
    gmx_set_error(std::string("gmx_device_warmup: ") + hipGetErrorString(hipGetLastError()));
    return GMX_EHIP;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_device_warmup")

int gmx_group_create(const gmx_index *ix, const gmx_engine_opts *opts_in, const int *devices, int n_devices, gmx_group **out) try {
  if (!ix || !devices || n_devices <= 0 || !out) {
    gmx_set_error("gmx_group_create: bad argument");
    return GMX_EINVAL;
  }
  gmx_group *g = new gmx_group();
  gmx_engine_opts opts;
  if (opts_in)
    opts = *opts_in;
  else
    gmx_engine_default_opts(&opts);
  bool distinct = true;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  // one host thread per device: every engine uploads the whole index (replicated), and at whole-genome scale that is
  // minutes per device when done one after the other. (Engines sharing a device are created in turn.)
  g->ms.resize(n_devices);
  std::vector<int> rcs(n_devices, GMX_OK);
  std::vector<std::string> errs(n_devices);
  auto create = [&](int i) {
    gmx_engine_opts o = opts;
    o.device = devices[i];
    rcs[i] = gmx_engine_create(ix, &o, &g->ms[i].e);
    try {
      if (rcs[i]) errs[i] = gmx_last_error();
      else gmx_engine_raw(g->ms[i].e, &g->ms[i].raw);
    } catch (...) {  // (the message could not be copied: the code stands)
    }
  };
  if (distinct && n_devices > 1) {
    GmxThreads th;
    for (int i = 0; i < n_devices; ++i) th.run([&create, i] { create(i); });
    th.join();
  } else {
    for (int i = 0; i < n_devices; ++i) create(i);
  }
  for (int i = 0; i < n_devices; ++i)
    if (rcs[i]) {
      gmx_set_error("device " + std::to_string(devices[i]) + ": " + errs[i]);
      const int rc = rcs[i];
      gmx_group_destroy(g);
      return rc;
    }
  if (n_devices > 1 && distinct && !getenv("GMX_NO_RCCL") && rccl().ok) {
    std::vector<ncclComm_t> comms(n_devices);
    if (rccl().CommInitAll(comms.data(), n_devices, devices) == ncclSuccess) {
      for (int i = 0; i < n_devices; ++i) g->ms[i].comm = comms[i];
      g->use_rccl = true;
    }
  }
  *out = g;
  return GMX_OK;
} GMX_GUARD_INT("gmx_group_create")

void gmx_group_destroy(gmx_group *g) try {
  if (!g) return;
  for (auto &m : g->ms) {
    if (m.comm) (void)rccl().CommDestroy(m.comm);
    if (m.e) gmx_engine_destroy(m.e);
  }
  delete g;
} GMX_GUARD_VOID("gmx_group_destroy")

int gmx_group_size(const gmx_group *g) { return g ? (int)g->ms.size() : 0; }
gmx_engine *gmx_group_engine(gmx_group *g, int i) { return g && i >= 0 && i < (int)g->ms.size() ? g->ms[i].e : nullptr; }
int gmx_group_uses_rccl(const gmx_group *g) { return g && g->use_rccl ? 1 : 0; }

int gmx_group_map_reads_host(gmx_group *g, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds, uint64_t n_reads) try {
  if (!g || g->ms.empty()) {
    gmx_set_error("null group");
    return GMX_EINVAL;
  }
  const size_t n = g->ms.size();
  if (n == 1) return gmx_map_reads_host(g->ms[0].e, reads, offsets, seeds, n_reads);
  std::vector<int> rcs(n, GMX_OK);
  std::vector<std::string> errs(n);
  GmxThreads th;
  for (size_t i = 0; i < n; ++i) {  // contiguous ranges of the global read index; the seeds are the global stream's
    const uint64_t base = n_reads / n, rem = n_reads % n;
    const uint64_t lo = i * base + std::min<uint64_t>(i, rem), cnt = base + (i < rem ? 1 : 0);
    th.run([=, &rcs, &errs]() {
      if (cnt == 0) return;
      rcs[i] = gmx_map_reads_host(g->ms[i].e, reads, offsets + lo, seeds + lo, cnt);
      try {
        if (rcs[i]) errs[i] = gmx_last_error();
      } catch (...) {
      }
    });
  }
  th.join();
  for (size_t i = 0; i < n; ++i)
    if (rcs[i]) {
      gmx_set_error("device " + std::to_string(g->ms[i].raw.device) + ": " + errs[i]);
      return rcs[i];
    }
  return GMX_OK;
} GMX_GUARD_INT("gmx_group_map_reads_host")

// The packed form, dealt the same way: engine i takes a contiguous range of the read index, its planes start at that
// range's first pair (gmx.h: a sub-range of a packed batch is a packed batch). Every engine's call returns once its
// chunks are enqueued (page-locked buffers), so the host threads are short-lived; gmx_group_sync_uploads waits for all.
int gmx_group_map_reads_packed_host(gmx_group *g, const uint64_t *planes, const uint64_t *offsets, uint32_t uniform_len,
                                    const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads) try {
  if (!g || g->ms.empty()) {
    gmx_set_error("null group");
    return GMX_EINVAL;
  }
  const size_t n = g->ms.size();
  if (n == 1) return gmx_map_reads_packed_host(g->ms[0].e, planes, offsets, uniform_len, seeds, skip, n_reads);
  if (!planes || !seeds || (!offsets && !uniform_len)) {
    gmx_set_error("gmx_group_map_reads_packed_host: null argument");
    return GMX_EINVAL;
  }
  std::vector<int> rcs(n, GMX_OK);
  std::vector<std::string> errs(n);
  GmxThreads th;
  const uint64_t ppr = (uniform_len + 31u) / 32u;
  auto pair_at = [&](uint64_t r) { return uniform_len ? r * ppr : ((offsets[r] >> 5) - (offsets[0] >> 5)) + r; };
  // Pageable buffers are registered with the runtime HERE, once and whole: the members' sub-ranges share pages, so a
  // registration per member can fail ("already registered") or make a neighbour's range look page-locked to a member that
  // then returns with its uploads in flight while the neighbour unregisters. Registered here, every member sees page-locked
  // memory and only queues its copies; the group waits for all of them before it unregisters.
  struct Reg { const void *p; uint64_t bytes; bool on; };
  Reg regs[4] = {{planes, pair_at(n_reads) * 8, false}, {offsets, (n_reads + 1) * 8, false}, {seeds, n_reads * 4, false}, {skip, n_reads, false}};
  bool registered = false;
  for (auto &r : regs) {
    if (!r.p || !r.bytes) continue;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, r.p) == hipSuccess && at.type == hipMemoryTypeHost) continue;
    (void)hipGetLastError();
    r.on = hipHostRegister(const_cast<void *>(r.p), r.bytes, hipHostRegisterPortable) == hipSuccess;
    (void)hipGetLastError();
    registered = registered || r.on;
  }
  for (size_t i = 0; i < n; ++i) {
    const uint64_t base = n_reads / n, rem = n_reads % n;
    const uint64_t lo = i * base + std::min<uint64_t>(i, rem), cnt = base + (i < rem ? 1 : 0);
    const uint64_t p0 = pair_at(lo);
    th.run([=, &rcs, &errs]() {
      if (cnt == 0) return;
      gmx_pin_feeder(i);
      rcs[i] = gmx_map_reads_packed_host(g->ms[i].e, planes + p0, offsets ? offsets + lo : nullptr, uniform_len, seeds + lo,
                                         skip ? skip + lo : nullptr, cnt);
      try {
        if (rcs[i]) errs[i] = gmx_last_error();
      } catch (...) {
      }
    });
  }
  th.join();
  if (registered) {
    for (auto &m : g->ms) (void)gmx_engine_sync_uploads(m.e);
    for (auto &r : regs)
      if (r.on) (void)hipHostUnregister(const_cast<void *>(r.p));
    (void)hipGetLastError();
  }
  for (size_t i = 0; i < n; ++i)
    if (rcs[i]) {
      gmx_set_error("device " + std::to_string(g->ms[i].raw.device) + ": " + errs[i]);
      return rcs[i];
    }
  return GMX_OK;
} GMX_GUARD_INT("gmx_group_map_reads_packed_host")

int gmx_group_sync_uploads(gmx_group *g) try {
  if (!g) {
    gmx_set_error("null group");
    return GMX_EINVAL;
  }
  for (auto &m : g->ms) {
    int rc = gmx_engine_sync_uploads(m.e);
    if (rc) return rc;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_group_sync_uploads")

int gmx_group_allreduce(gmx_group *g) try {
  if (!g || g->ms.empty()) {
    gmx_set_error("null group");
    return GMX_EINVAL;
  }
  if (g->ms.size() == 1) return GMX_OK;
  for (auto &m : g->ms) {
    int rc = gmx_engine_sync(m.e);
    if (rc) return rc;
  }
  int rc = g->use_rccl ? gmx_exchange(g->ms, (int)g->ms.size()) : gmx_exchange_peer(g->ms);
  if (rc) return rc;
  for (auto &m : g->ms) {
    HIP_TRY(hipSetDevice(m.raw.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_group_allreduce")

int gmx_comm_unique_id(uint8_t *out128) try {
  if (!rccl().ok) {
    gmx_set_error("RCCL (librccl.so) could not be loaded");
    return GMX_ENODEV;
  }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(rccl().GetUniqueId(&id));
  memcpy(out128, &id, 128);
  return GMX_OK;
} GMX_GUARD_INT("gmx_comm_unique_id")

int gmx_comm_create(const uint8_t *id128, int world, int rank, gmx_engine *e, gmx_comm **out) try {
  if (!id128 || !e || !out || world <= 0 || rank < 0 || rank >= world) {
    gmx_set_error("gmx_comm_create: bad argument");
    return GMX_EINVAL;
  }
  if (!rccl().ok) {
    gmx_set_error("RCCL (librccl.so) could not be loaded");
    return GMX_ENODEV;
  }
  gmx_comm *c = new gmx_comm();
  c->world = world;
  c->rank = rank;
  c->m.e = e;
  gmx_engine_raw(e, &c->m.raw);
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  if (hipSetDevice(c->m.raw.device) != hipSuccess) {
    delete c;
    gmx_set_error("hipSetDevice failed");
    return GMX_EHIP;
  }
  ncclResult_t r = rccl().CommInitRank(&c->m.comm, world, id, rank);
  if (r != ncclSuccess) {
    gmx_set_error(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    delete c;
    return GMX_EHIP;
  }
  *out = c;
  return GMX_OK;
} GMX_GUARD_INT("gmx_comm_create")

void gmx_comm_destroy(gmx_comm *c) try {
  if (!c) return;
  if (c->m.comm) (void)rccl().CommDestroy(c->m.comm);
  delete c;
} GMX_GUARD_VOID("gmx_comm_destroy")

int gmx_comm_allreduce_coverage(gmx_comm *c, void *hip_stream) try {
  if (!c) {
    gmx_set_error("null communicator");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(c->m.raw.device));
  c->m.stream = (hipStream_t)hip_stream;
  std::vector<Member> ms(1, c->m);
  return gmx_exchange(ms, c->world);
} GMX_GUARD_INT("gmx_comm_allreduce_coverage")

}  // extern "C"
