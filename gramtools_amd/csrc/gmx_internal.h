// gmx_internal.h — glue shared by the two translation units of libgmx.so.
#pragma once
#include <string>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_index.h"

void gmx_set_error(const std::string &msg);
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
