// gmx_engine.hip — HIP kernels (gfx950) and the device half of the C ABI.
//
// Execution model (HISTORY.md §2): one LANE per (read, orientation) task — 64 independent vBWT backward
// searches per wavefront. A lane carries one search state in registers and keeps the others on a small LIFO
// stack in LDS (gmx_dfs.h). A state that has narrowed to ONE suffix-array position is kept in text form and
// compares 32 read bases per step against a 16-byte record of the PRG itself; wide intervals use 64-byte rank
// blocks; a variant marker costs one 16-byte sub-record of its pre-resolved hit record. Final states go to
// the coverage kernels (gmx_cover.h): class selection with the seeded draw, then the coverage atomics.
//
// Kernels (launch order, HISTORY.md §2.4):
//   gmx_pack_kernel          flags reads with a non-ACGT byte (encode_dna_bases, utils.cpp:73-92), packs bases to bit planes
//   gmx_probe_kernel         seed look-up + the first steps of search_read_backwards (quasimap.cpp:227-256); survivors are parked
//   gmx_extend_kernel        the rest of the read for the compacted survivors
//   gmx_search_big_kernel    the same search for tasks that overflowed the per-lane pools (global-memory pools; side streams)
//   gmx_filter[_lds]_kernel  all_read_kmers_occur_in_index for tasks without final state (quasimap.cpp:212-225)
//   gmx_cover_single_kernel  coverage::record::search_states (coverage_common.cpp:179-197) for single-instance tasks
//   gmx_cover_kernel         the same in general (classes, seeded selection, hull), two scratch sizes
//   (QuasimapReadsStats, quasimap.hpp:17-24, are counted by the kernels that decide each task: SearchOut::stats)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_core.h"
#ifdef GMX_LOOP_STATS  // debug build: wall time of the coverage routine's phases (tools/cover_stats.py)
#define GMX_COVER_PROF(env, k) (env).prof(k)
#define GMX_COVER_WHY(env, k) (env).why(k)
#endif
#include "gmx_cover.h"
#include "gmx_dfs.h"
#include "gmx_index.h"
#include "gmx_internal.h"

#ifndef GMX_BLOCK
#define GMX_BLOCK 256
#endif
#ifndef GMX_FAST_STATES
#define GMX_FAST_STATES 8     // final / parked states kept per task by the fast pass
#endif
#ifndef GMX_STACK_DEPTH
#define GMX_STACK_DEPTH 6
#endif
// GMX_STACK_DEPTH: pending entries (sibling states, unresolved marker hits) per lane, in LDS
#define GMX_STACK_WORDS 5
#define GMX_CNT_STRIDE 32      // device counters sit 128 B apart: same-line atomics would serialise in one L2 channel
#define GMX_N_COUNTERS 40      // per-batch queue counters (SearchOut::counters)
#define GMX_CNT_LOG_RETRY 30u       // entries of log_retry_list: coverage queue entries whose task found the grouped log full
#define GMX_CNT_LOG_RETRY_RECS 31u  // ... compact records (log_retry_recs: index into cover_recs)
#define GMX_CNT_LOG_RETRY_HUGE 33u  // ... tasks the last tier has to search again (log_retry_huge)
#define GMX_CNT_GENERAL_REST 34u    // entries gmx_cover_one_kernel left to the general instances (general_rest_list)
#define GMX_CNT_ALIVE2 35u          // stragglers of the extend kernel, parked for its next pass: [35 + pass], pass 0 .. GMX_EXTRA_PASSES - 1
#define GMX_EXTRA_PASSES 3          // (counters 35, 36, 37; task lists GMX_TL_ALIVE2 ..)
#define GMX_CNT_SINGLE_REST 38u     // compact records gmx_cover_jump_kernel left to gmx_cover_single_rest_kernel (single_rest_list)
#define GMX_CNT_REPLAY_RECS 32u     // replay: number of compact records to redo (gmx_cover_single_replay_kernel)
#ifndef GMX_FAST_ARENA
#define GMX_FAST_ARENA 48     // path arena nodes per task (fast pass): a read through an MSA region of configs[2] needs 25-40
                              // (24 sent 14 k of a million such reads to the large-capacity pass: 5.1 -> 3.1 ms per batch)
#endif
#define GMX_STATUS_MISSING_KMER 5u  // refinement of GMX_TASK_UNMAPPED by the k-mer filter
#define GMX_STATUS_IGNORED 7u       // reverse-complement task of a forward_only engine: not mapped, not counted

// The rest of this translation unit, in three parts (one TU: the kernels share the contexts, the launch code names the kernels):
#include "gmx_engine_search.h"         // contexts, dfs_run_wave, search kernels, k-mer filter
#include "gmx_engine_cover_kernels.h"  // coverage kernels, pack kernel
#include "gmx_engine_host.h"           // engine object, launches, feeds, C ABI
