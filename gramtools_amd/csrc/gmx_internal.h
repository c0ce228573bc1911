// gmx_internal.h — glue shared by the two translation units of libgmx.so.
#pragma once
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_index.h"

void gmx_set_error(const std::string &msg) noexcept;  // (never throws: a message that cannot be stored is dropped)

// Nothing C++ leaves the library (include/gmx.h: "returns 0 or a negative GMX_E* code"). Every exported function is a
// function-try-block closed by one of these; gmx_guard_catch() names the exception in flight: std::bad_alloc -> GMX_ENOMEM,
// std::system_error (a thread that could not be started) -> GMX_ENOMEM, any other std::exception -> GMX_EINVAL with its what(),
// anything else -> GMX_EINVAL. The reference's process ends such a run with a message and a non-zero exit code
// (gramtools/commands/genotype/genotype.py:106-107), not with SIGABRT.
int gmx_guard_catch(const char *fn) noexcept;
#define GMX_GUARD_INT(fn) catch (...) { return gmx_guard_catch(fn); }
#define GMX_GUARD_VOID(fn) catch (...) { (void)gmx_guard_catch(fn); }
#define GMX_GUARD_PTR(fn) catch (...) { (void)gmx_guard_catch(fn); return nullptr; }
#define GMX_GUARD_ZERO(fn) catch (...) { (void)gmx_guard_catch(fn); return 0; }

// Joins what it holds when it goes out of scope, also on the way out of an exception (a joinable std::thread's destructor
// ends the process). The worker bodies catch for themselves: an exception that leaves a thread's function ends the process too.
struct GmxThreads {
  std::vector<std::thread> th;
  template <class F> void run(F &&f) { th.emplace_back(std::forward<F>(f)); }
  void join() { for (auto &t : th) if (t.joinable()) t.join(); }
  ~GmxThreads() { join(); }
};
const gmx::HostIndex &gmx_index_host(const gmx_index *ix);
uint64_t gmx_index_serial(const gmx_index *ix);  // unique within the process

// What the multi-GPU exchange (gmx_multi.hip) needs of an engine (gmx_engine.hip).
struct GmxEngineRaw {
  int device;
  uint32_t *d_fused;  // the accumulator block + 32 counter-limb words (gmx_coverage_device)
  size_t n_fused;
  bool log_sites;     // the index has sites that use the grouped log
};
void gmx_engine_raw(gmx_engine *e, GmxEngineRaw *out);
// the engine's grouped log as counted records (gmx.h: gmx_coverage_fetch_grouped_log); import adds records to the
// engine's host-side totals, after emptying them when `replace`
int gmx_engine_log_export(gmx_engine *e, std::vector<uint32_t> &out);
int gmx_engine_log_import(gmx_engine *e, const uint32_t *records, size_t n_words, bool replace);
