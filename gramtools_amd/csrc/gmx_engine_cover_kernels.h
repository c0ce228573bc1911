// gmx_engine_cover_kernels.h — part of the ONE translation unit gmx_engine.hip (included there): the coverage kernels (general
// instances, cooperative instance, single-instance kernels over compact records and geometry records, log replay), the read
// counters' limbs and the pack kernel.
#pragma once
// ---------------------------------------------------------------------------
// coverage kernel
// ---------------------------------------------------------------------------
struct CoverAcc {
  uint32_t *acc;        // the accumulator block (gmx_types.h: gmx_slot_*)
  uint32_t *log;        // grouped log words
  uint32_t *log_cursor; // [0] = words used
  uint32_t log_cap;
  uint32_t *scratch_big;
  uint32_t n_lanes_big;
  int rng_mode;
  uint32_t log_sites;   // the index has sites with more than 8 alleles (users of the log)
  uint32_t *heap;       // the last tier's memory (gmx_tail_stage)
  uint64_t heap_words;
  const uint32_t *status;      // per task, for the read counters tallied by the batch's last launch
  uint32_t n_tasks;
  unsigned long long *stats;   // QuasimapReadsStats counters
};

// The grouped log (sites without dense group counters): a task reserves ALL the words it will append with one atomic add,
// before it records anything (gmx_cover.h); a task that does not fit fails whole (GMX_TASK_LOGFULL), gives its words back
// and is redone after the host has drained the log (log_settle). GMX_LOG_PAD words (a reservation abandoned on an error)
// are skipped by every reader.
#define GMX_LOG_PAD 0xFFFFFFFFu
#ifdef GMX_LOOP_STATS
// per coverage instance (LIST): [0..7] wall time (10 ns units) per phase summed over tasks, [8..15] its maximum
__device__ unsigned long long gmx_cover_stats[6 * 16];
__device__ unsigned long long gmx_coop_stats[6 * 8];  // cooperative instances: wave-level wall time of the four phases, [7] rounds
extern "C" int gmx_debug_coop_stats(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_coop_stats), sizeof(gmx_coop_stats)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[6 * 8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_coop_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
__device__ unsigned long long gmx_cover_why[8 * 4];  // per coverage instance (6, 7: cooperative item / class scratch): tasks that exceeded loci / key sites / hull / items
extern "C" int gmx_debug_cover_why(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_cover_why), sizeof(gmx_cover_why)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8 * 4] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_cover_why), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int gmx_debug_cover_stats(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_cover_stats), sizeof(gmx_cover_stats)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[6 * 16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_cover_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
struct CoverLogPart {
#ifdef GMX_LOOP_STATS
  long long prof_t = 0;
  int prof_list = 0;
  __device__ void why(int k) { atomicAdd(&gmx_cover_why[(prof_list & 7) * 4 + (k & 3)], 1ull); }
  __device__ void prof(int k) {
    const long long t = wall_clock64();
    atomicAdd(&gmx_cover_stats[prof_list * 16 + k], (unsigned long long)(t - prof_t));
    atomicMax(&gmx_cover_stats[prof_list * 16 + 8 + k], (unsigned long long)(t - prof_t));
    prof_t = t;
  }
#endif
  uint32_t *acc, *log, *log_cursor;
  uint32_t log_cap;
  uint32_t status;
  uint32_t log_at;
  uint32_t log_end = 0;  // end of this task's reservation
  uint32_t log_sites;  // the index has sites that use the log
  __device__ __forceinline__ bool has_log_sites() const { return log_sites != 0; }
  __device__ __forceinline__ bool log_reserve(uint32_t words) {
    // compare-and-swap: the cursor moves only for a reservation that fits, so it never exceeds the capacity, the words
    // below it are exactly the successful reservations back to back, and a failing task leaves no trace (an add that is
    // taken back later opens a window in which another task's words end up beyond the cursor).
    uint32_t cur = __hip_atomic_load(log_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      if (cur > log_cap || words > log_cap - cur) {
        log_at = log_end = 0;
        status = GMX_TASK_LOGFULL;
        return false;
      }
      const uint32_t seen = atomicCAS(log_cursor, cur, cur + words);
      if (seen == cur) break;
      cur = seen;
    }
    log_at = cur;
    log_end = log_at + words;
    return true;
  }
  // a task that failed AFTER its reservation (a condition on which the reference throws) leaves no unwritten words behind
  __device__ __forceinline__ void log_abandon() {
    if (status != GMX_TASK_MAPPED && status != GMX_TASK_LOGFULL)
      for (uint32_t i = log_at; i < log_end && i < log_cap; ++i) log[i] = GMX_LOG_PAD;
    log_at = log_end = 0;
  }
  __device__ __forceinline__ bool log_grouped_begin(uint32_t site_index, uint32_t n_ids) {
    log[log_at++] = site_index;
    log[log_at++] = n_ids;
    return true;
  }
  __device__ __forceinline__ void log_grouped_id(int32_t a) { log[log_at++] = (uint32_t)a; }
  __device__ __forceinline__ void log_grouped_end() {}
  __device__ __forceinline__ uint32_t single_loci() const { return GMX_SINGLE_LOCI; }
  __device__ __forceinline__ void add_allele_sum(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_per_base(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_hit(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_grouped_dense(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_allele_and_group(uint32_t slot) {  // slot is even: both counters in one 64-bit add
    atomicAdd(reinterpret_cast<unsigned long long *>(acc + slot), 0x100000001ull);
  }
  __device__ __forceinline__ void fail(uint32_t s) {
    if (status == GMX_TASK_MAPPED || s == GMX_TASK_ERROR) status = s;
  }
};

template <uint32_t I_, uint32_t B_, uint32_t LOC_, uint32_t H_, uint32_t P_ = GMX_PATH_CACHE>
struct CoverEnvT : CoverLogPart {
  static constexpr uint32_t I_MAX = I_, B_MAX = B_, LOC_MAX = LOC_, H_MAX = H_, P_MAX = P_;
  __device__ __forceinline__ static constexpr uint32_t i_max() { return I_; }
  __device__ __forceinline__ static constexpr uint32_t b_max() { return B_; }
  __device__ __forceinline__ static constexpr uint32_t loc_max() { return LOC_; }
  __device__ __forceinline__ static constexpr uint32_t h_max() { return H_; }
  uint32_t *scratch;  // already offset by the lane
  uint32_t stride;
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[(size_t)w * stride]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[(size_t)w * stride] = v; }
};

// The last tier: capacities decided per task from what the heap slice holds (gmx_tail_stage).
struct CoverEnvDyn : CoverLogPart {
  uint32_t cap_i, cap_b, cap_loc, cap_h;
  __device__ __forceinline__ uint32_t i_max() const { return cap_i; }
  __device__ __forceinline__ uint32_t b_max() const { return cap_b; }
  __device__ __forceinline__ uint32_t loc_max() const { return cap_loc; }
  __device__ __forceinline__ uint32_t h_max() const { return cap_h; }
  uint32_t *scratch;
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[w]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[w] = v; }
};

typedef CoverEnvT<4, 12, 24, 24> CoverEnvLds;         // first tier of the general pass: per-lane scratch in the block's LDS
typedef CoverEnvT<12, 12, 48, 48> CoverEnvMid;         // the large-capacity pass's tasks (a read in a 10-copy repeat has ~11 items)
typedef CoverEnvT<24, 16, 64, 64> CoverEnv;           // per-lane scratch of the regular pass
typedef CoverEnvT<1024, 32, 1024, 1024> CoverEnvBig;  // reads with many mapping instances (repeats)

// ---------------------------------------------------------------------------
// The last tier. Every pool above has a fixed size per task; a task that exceeds one of them — a read with thousands of
// mapping instances, or through more nested sites than the large-capacity pools hold — ends up here, where the only
// limit is the engine's heap (gmx_engine_opts::huge_heap_bytes): the reference has no limit either
// (encapsulated_search.cpp:30-107 and coverage_common.cpp:85-146 simply iterate). Work items are
//   * tasks of huge_list: searched again from the seed with pools carved from a heap slice, then recorded with a scratch
//     sized for what the search produced (gmx_cover_task over CoverEnvDyn);
//   * entries of cover_huge_list: their final states are where the search left them, only the scratch was too small.
// One wave runs the stage (the last block of the batch's last coverage launch): first every lane takes work items with
// one 64th of the heap each, then lane 0 alone redoes, with the whole heap, what did not fit. Nothing is recorded for a
// task before all of its capacity checks have passed, so redoing is safe. A task that does not fit the whole heap is
// reported (GMX_ECAP: raise huge_heap_bytes). Common batches have no work item and pay one counter read.
// ---------------------------------------------------------------------------
// What a coverage queue entry stands for: a task finished by the fast pass (its id), a large-capacity slot, or the slot
// of an instance-searched task.
struct GmxTaskStates {
  uint32_t task, nf;
  const GmxFinalState *finals;
  const GmxPathNode *arena;
};
__device__ __forceinline__ GmxTaskStates gmx_entry_states(uint32_t entry, const SearchOut &o, const BigOut &g) {
  GmxTaskStates t;
  if ((entry & GMX_ENTRY_INST) == GMX_ENTRY_INST) {
    const uint32_t slot = entry & 0x3fffffffu, first = o.inst_first[slot];
    t.task = o.slot_task[slot];
    t.nf = o.slot_n_final[slot];
    t.finals = o.inst_states + (size_t)first * GMX_INST_STATES;
    t.arena = o.inst_arena + (size_t)first * GMX_FAST_ARENA;
  } else if (entry & GMX_ENTRY_BIG) {
    const uint32_t slot = entry & 0x7fffffffu;
    t.task = g.task_of_slot[slot];
    t.nf = g.n_final[slot];
    t.finals = g.states + (size_t)slot * g.max_states;
    t.arena = g.arena + (size_t)slot * g.max_path_nodes;
  } else {
    t.task = entry;
    t.nf = o.n_final[entry] & 0xFF;
    t.finals = o.finals + (size_t)entry * GMX_FAST_STATES;
    t.arena = o.arena + entry;  // handles are offsets from the task's base (FastCtx::alloc_node)
  }
  return t;
}

__device__ uint32_t gmx_count_items(const GmxIndexView &ix, const GmxFinalState *finals, uint32_t nf) {
  uint32_t n = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    const GmxFinalState st = finals[f];
    if (st.traversed != GMX_NIL || st.traversing != GMX_NIL) {
      ++n;
      continue;
    }
    for (uint32_t i = st.lo;; ++i) {
      n += ix.nodes[ix.pos_node[gmx_occ_pos(ix, st.hi, i)]].site != 0;
      if (gmx_text_form(st.hi) || i == st.hi) break;
    }
  }
  return n;
}

// returns the status of the work item: MAPPED (done), OVERFLOW (the slice was too small, nothing recorded), or an error
__device__ uint32_t gmx_tail_item(const GmxIndexView &ix, const BatchView &b, const SearchOut &o, const BigOut &g, const CoverAcc &acc,
                                  bool active, uint32_t item, uint32_t n_search, uint32_t *slice, uint64_t slice_words, bool whole_heap,
                                  uint32_t &task_out) {
  const bool is_search = active && item < n_search;
  uint32_t task = 0, nf = 0;
  const GmxFinalState *finals = nullptr;
  const GmxPathNode *arena = nullptr;
  uint32_t *scratch = slice;
  uint64_t scratch_words = slice_words;
  uint32_t status = GMX_TASK_MAPPED;
  // --- search (all lanes of the wave take part in the loop's ballots) ---
  BigCtx ctx;
  const uint64_t S = std::min<uint64_t>(slice_words / 30, 0x3FFFFFFFull);  // states; half of the slice is left for the scratch
  ctx.sp = 0;
  ctx.cap = (uint32_t)S;
  ctx.out = reinterpret_cast<GmxFinalState *>(slice);
  ctx.stack = slice + 4 * S;
  ctx.arena = reinterpret_cast<GmxPathNode *>(slice + 9 * S);
  ctx.arena_n = 0;
  ctx.arena_cap = (uint32_t)(2 * S);
  ctx.status = GMX_TASK_MAPPED;
  ctx.n_out = 0;
  ctx.out_cap = (uint32_t)S;
  ReadRef r;
  r.w = b.packed;
  r.len = 0;
  r.rc = false;
  r.cur_idx = 0xFFFFFFFFu;
  r.cur = make_uint2(0, 0);
  bool run = false;
  if (is_search) {
    task = o.huge_list[item];
    r = task_read(b, task);
    const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
    const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
    const uint32_t from = r.len - k;
    load_seed(ix, longer ? ix.seeds2 : ix.seeds, kmer_code(r, from, k), ctx,
              [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
                if (tvd == GMX_NIL && tvg == GMX_NIL && from > 0 && hi > lo && hi != GMX_TEXT_MARK) {  // position by position in text form (gmx_search_big_kernel)
                  bool ok = true;
                  for (uint32_t i = lo; ok; ++i) {
                    ok = ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
                    if (i == hi) break;
                  }
                  return ok;
                }
                return ctx.push(lo, hi, tvd, tvg, from, GMX_MODE_STATE);
              });
    run = ctx.status == GMX_TASK_MAPPED;
  }
  GmxLane ln;
  dfs_run_wave<2, false>(ix, ctx, r, 0, run, 0, ln);
  if (!active) return GMX_TASK_MAPPED;
  uint32_t len;
  if (is_search) {
    task_out = task;
    status = ctx.status;
    if (status != GMX_TASK_MAPPED) return status;
    nf = ctx.n_out;
    if (nf == 0) {
      // its read counter
      atomicAdd(&acc.stats[all_kmers_present(ix.kmer_bitmap, ix.kmer_size, r) ? 3 : 2], 1ull);
      o.n_final[task] = 0;
      return GMX_TASK_MAPPED;
    }
    finals = ctx.out;
    arena = ctx.arena;
    scratch = slice + 15 * S;
    scratch_words = slice_words - 15 * S;
    len = r.len;
  } else {
    const GmxTaskStates ts = gmx_entry_states(o.cover_huge_list[item - n_search], o, g);
    task = ts.task;
    nf = ts.nf;
    finals = ts.finals;
    arena = ts.arena;
    task_out = task;
    const uint32_t read = task >> 1;
    len = read_len(b, read);
  }
  // --- coverage with a scratch sized for this task ---
  CoverEnvDyn env;
  const uint64_t n_items = std::max<uint32_t>(gmx_count_items(ix, finals, nf), 1u);
  uint64_t cap_b = std::min<uint64_t>(std::max<uint64_t>(len + 8u, 32u), 4096u);
  if (whole_heap) cap_b = std::max<uint64_t>(cap_b, std::min<uint64_t>(65536u, scratch_words / (4 * n_items)));
  const uint64_t fixed = n_items * (GmxScratch<CoverEnvDyn>::ITEM_W + 2 + cap_b) + 2 * GMX_PATH_CACHE + 1;
  if (fixed + 5 * 64 > scratch_words) return GMX_TASK_OVERFLOW;
  const uint64_t rest = std::min<uint64_t>((scratch_words - fixed) / 5, 0x0FFFFFFFull);
  env.cap_i = (uint32_t)n_items;
  env.cap_b = (uint32_t)cap_b;
  env.cap_loc = env.cap_h = (uint32_t)rest;
  env.scratch = scratch;
  env.arena = arena;
  env.acc = acc.acc;
  env.log = acc.log;
  env.log_cursor = acc.log_cursor;
  env.log_cap = acc.log_cap;
  env.log_sites = acc.log_sites;
  env.status = GMX_TASK_MAPPED;
  env.log_at = 0;
  gmx_cover_task(ix, env, finals, nf, len, b.seeds[task >> 1], acc.rng_mode);
  env.log_abandon();
  if (env.status == GMX_TASK_MAPPED && is_search) {
    atomicAdd(&acc.stats[4], 1ull);  // exact_mapped
    o.n_final[task] = nf;
  }
  return env.status;
}

// a work item of the last tier that found the grouped log full: redone after the host has drained the log
__device__ __forceinline__ void gmx_tail_log_retry(const SearchOut &o, uint32_t item, uint32_t n_search) {
  if (item < n_search)
    o.log_retry_huge[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE], 1u)] = o.huge_list[item];
  else
    o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = o.cover_huge_list[item - n_search];
}

__device__ void gmx_tail_stage(const GmxIndexView &ix, const BatchView &b, const SearchOut &o, const BigOut &g, const CoverAcc &acc) {
  const uint32_t n_search = o.counters[11 * GMX_CNT_STRIDE], n_cover = o.counters[15 * GMX_CNT_STRIDE];
  const uint32_t total = n_search + n_cover;
  if (total == 0) return;
  __shared__ uint32_t n_retry;
  if (threadIdx.x == 0) n_retry = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t slice_words = acc.heap_words / 64;
  for (uint32_t base = 0; base < total; base += 64) {  // every lane: one work item, one 64th of the heap
    const uint32_t item = base + lane;
    uint32_t task = 0;
    const uint32_t st = gmx_tail_item(ix, b, o, g, acc, item < total, item, n_search, acc.heap + (size_t)lane * slice_words, slice_words,
                                      false, task);
    if (item < total && st == GMX_TASK_OVERFLOW)
      o.huge_retry[atomicAdd(&n_retry, 1u)] = item;
    else if (item < total && st == GMX_TASK_LOGFULL)
      gmx_tail_log_retry(o, item, n_search);
    else if (item < total && st != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, st) == 0u)
      o.error[1] = task;
  }
  __syncthreads();
  __threadfence();
  const uint32_t retries = n_retry;
  for (uint32_t i = 0; i < retries; ++i) {  // lane 0 alone, the whole heap
    uint32_t task = 0;
    const uint32_t st = gmx_tail_item(ix, b, o, g, acc, lane == 0, o.huge_retry[i], n_search, acc.heap, acc.heap_words, true, task);
    if (lane == 0 && st == GMX_TASK_LOGFULL)
      gmx_tail_log_retry(o, o.huge_retry[i], n_search);
    else if (lane == 0 && st != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, st) == 0u)
      o.error[1] = task;
  }
}

// Four instances over four device-side queues (LIST):
//   3  tasks finished by the probe / extend kernels that gmx_cover_single_kernel passed on; scratch sized for the
//      few instances and loci most such tasks have
//   0  those whose selection exceeded it, regular scratch
//   2  tasks finished by the large-capacity search (runs on the engine's side stream), regular scratch
//   1  entries of 0 and 2 whose selection exceeded the regular scratch, redone with the large one after both
// An entry that exceeds a scratch has recorded nothing yet.  The per-lane scratch of 3, 0 and 2 lives in the block's
// LDS (a dependent chain of scratch accesses per task: LDS latency, not L2 latency, sets the pace); a block runs
// gmx_cover_lds_lanes<Env>() lanes, as many as copies of the scratch fit 64 KB.  Instance 1 uses global memory.
template <class Env>
constexpr uint32_t gmx_cover_lds_lanes() {
  return GmxScratchFixed<Env>::total * 64 * sizeof(uint32_t) <= 64 * 1024   ? 64u
         : GmxScratchFixed<Env>::total * 32 * sizeof(uint32_t) <= 64 * 1024 ? 32u
                                                                       : 16u;
}
template <class Env, int LIST>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g,
                                                              CoverAcc acc, uint32_t lanes_rt, uint32_t after_coop) {
  constexpr bool BIG = LIST == 1;
  constexpr bool LDS = LIST != 1;
  const uint32_t LANES = LDS ? lanes_rt : 64u;  // active lanes of a block (blockDim.x is 64)
  // LIST 4 and 2 share the large-capacity pass's queue: 4 takes what its first instance mapped and leaves the length
  // in counter [10], 2 starts there
  // (instances 3, 5 and 2 after the cooperative kernel: only what that one left, reject lists and counters [27], [26], [28])
  uint32_t n_mapped = o.counters[(LIST == 3   ? (after_coop ? 27 : GMX_CNT_GENERAL_REST)
                                 : LIST == 0 ? 13
                                 : LIST == 1 ? 4
                                 : LIST == 5 ? (after_coop ? 26 : 25)
                                 : LIST == 2 ? (after_coop ? 28 : 7)
                                             : 7) * GMX_CNT_STRIDE];
  const uint32_t m_start = LIST == 2 && !after_coop ? o.counters[10 * GMX_CNT_STRIDE] : 0u;
  const uint32_t *list = LIST == 3   ? (after_coop ? o.general_serial_list : o.general_rest_list)
                         : LIST == 0 ? o.cover_mid_list
                         : LIST == 1 ? o.cover_overflow_list
                         : LIST == 5 ? (after_coop ? o.inst_serial_list : o.inst_mapped_list)
                         : LIST == 2 ? (after_coop ? o.big_serial_list : o.big_mapped_list)
                                     : o.big_mapped_list;
  if (LIST == 4 && blockIdx.x == 0 && threadIdx.x == 0) o.counters[10 * GMX_CNT_STRIDE] = n_mapped;  // read by LIST 2 only
  if (threadIdx.x >= LANES) return;
#ifdef GMX_LOOP_STATS
  long long t_kernel = wall_clock64();
#endif
  const uint32_t lane_id = blockIdx.x * LANES + threadIdx.x;
  const uint32_t work_blocks = gridDim.x;
  // interleaved: a short queue spreads over all waves (few diverging lanes each) instead of filling the first ones
  for (uint32_t m = m_start + threadIdx.x * work_blocks + blockIdx.x; m < n_mapped; m += work_blocks * LANES) {
    uint32_t entry = list[m];
    uint32_t task, nf;
    const GmxFinalState *finals;
    const GmxPathNode *arena;
    {
      const GmxTaskStates ts = gmx_entry_states(entry, o, g);
      task = ts.task;
      nf = ts.nf;
      finals = ts.finals;
      arena = ts.arena;
    }
    uint32_t read = task >> 1;
    uint32_t len = read_len(b, read);
    Env env;
    env.scratch = LDS ? gmx_lds + threadIdx.x : acc.scratch_big + lane_id;
    env.stride = LDS ? LANES : acc.n_lanes_big;
    env.arena = arena;
    env.acc = acc.acc;
    env.log = acc.log;
    env.log_cursor = acc.log_cursor;
    env.log_cap = acc.log_cap;
    env.log_sites = acc.log_sites;
    env.status = GMX_TASK_MAPPED;
    env.log_at = 0;
#ifdef GMX_LOOP_STATS
    env.prof_list = LIST;
    env.prof_t = t_kernel;
    env.prof(6);  // from the start of the kernel (first task of the lane) or the end of the lane's previous task
    atomicAdd(&gmx_cover_stats[LIST * 16 + 7], 1ull);
#endif
    gmx_cover_task(ix, env, finals, nf, len, b.seeds[read], acc.rng_mode);
    env.log_abandon();
#ifdef GMX_LOOP_STATS
    env.prof(5);
    t_kernel = env.prof_t;
#endif
    if (env.status == GMX_TASK_OVERFLOW && LIST == 3) {  // nothing has been recorded for it yet: next scratch size
      o.cover_mid_list[atomicAdd(&o.counters[13 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_OVERFLOW && !BIG) {
      o.cover_overflow_list[atomicAdd(&o.counters[4 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_OVERFLOW) {  // beyond the largest fixed scratch: the last tier sizes one from its heap
      o.cover_huge_list[atomicAdd(&o.counters[15 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_LOGFULL) {  // nothing recorded: again once the host has drained the log
      o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status != GMX_TASK_MAPPED) {
      if (atomicCAS(&o.error[0], 0u, env.status) == 0u) o.error[1] = task;
    }
  }
  if (BIG) {  // this instance is the batch's last search / coverage launch: whichever block finishes last serves the last tier
    __shared__ uint32_t ticket;
    __threadfence();
    if (threadIdx.x == 0) ticket = atomicAdd(&o.counters[14 * GMX_CNT_STRIDE], 1u);
    __syncthreads();
    if (ticket == work_blocks - 1) {
      __threadfence();
      gmx_tail_stage(ix, b, o, g, acc);
    }
  }
}

// ---------------------------------------------------------------------------
// Single-instance tasks the compact path could not take — a nested traversing path, traversed sites that are not
// consecutive (children inside an MSA region), or more loci than the register slots of gmx_cover_single_nested hold —
// one lane per task with the loci in LDS (gmx_cover_single_nested_wide): no keys, no sort, no class merge, no draw.
// The general instances (cooperative, then serial) spent 0.8 ms of wall time per round on such tasks at configs[2]
// (profiles/round3/coop_phases_config2.txt): a single state of width one has ONE item, hence one class, and the draw
// cannot change the outcome (coverage_common.cpp:166-177 with one class and no non-variant instance selects it whatever
// the number drawn). What does not fit (several final states, wide intervals, more than 32 loci) goes on to them.
// ---------------------------------------------------------------------------
#define GMX_ONE_THREADS 64
struct OneEnv : CoverLogPart {
  uint32_t *scratch;  // this lane's words, GMX_ONE_THREADS apart
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[w * GMX_ONE_THREADS]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[w * GMX_ONE_THREADS] = v; }
};
__global__ void __launch_bounds__(GMX_ONE_THREADS) gmx_cover_one_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, CoverAcc acc,
                                                                        uint32_t enabled) {
  const uint32_t n = o.counters[8 * GMX_CNT_STRIDE];
  // (entry m = lane * blocks + block: a short queue — a few thousand entries among a million reads on a flat PRG — is spread
  //  over all workgroups, a handful of lanes each, instead of filling the first few waves with 64 divergent dependent-load
  //  chains apiece: the kernel's duration is that of its slowest wave)
  for (uint32_t m = threadIdx.x * gridDim.x + blockIdx.x; m < n; m += gridDim.x * GMX_ONE_THREADS) {
    const uint32_t entry = o.cover_general_list[m];
    const GmxTaskStates ts = gmx_entry_states(entry, o, g);
    bool taken = false;
    if (enabled && ts.nf == 1) {
      const GmxFinalState st = ts.finals[0];
      if (gmx_text_form(st.hi) || st.lo == st.hi) {
        OneEnv env;
        env.scratch = gmx_lds + threadIdx.x;
        env.arena = ts.arena;
        env.acc = acc.acc;
        env.log = acc.log;
        env.log_cursor = acc.log_cursor;
        env.log_cap = acc.log_cap;
        env.log_sites = acc.log_sites;
        env.status = GMX_TASK_MAPPED;
        env.log_at = 0;
        const uint32_t len = read_len(b, ts.task >> 1);
        if (!taken) taken = gmx_cover_single_nested_wide(ix, env, st, len);
        if (env.status == GMX_TASK_LOGFULL) {
          o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
        } else if (env.status != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, env.status) == 0u) {
          o.error[1] = ts.task;
        }
        env.log_abandon();
      }
    }
    if (!taken) o.general_rest_list[atomicAdd(&o.counters[GMX_CNT_GENERAL_REST * GMX_CNT_STRIDE], 1u)] = entry;
  }
}

// ---------------------------------------------------------------------------
// The general routine, cooperatively: 16 lanes per task, one lane per item. The serial instances above spend one lane
// on a whole task — a read with ten mapping instances is ten items' worth of loci, keys, a sort and a class search in
// one lane, and a wave of such lanes executes the union of all their branches: the SIMDs, not memory, set the pace.
// Here the items of a task are spread over lanes that all run the same short code:
//   units    a path-bearing final state is one unit (an item); a pathless one has one unit per occurrence, each a
//            non-variant instance or an allele-encapsulated item (encapsulated_search.cpp:30-107). Lanes load one final
//            state each, a prefix sum of the widths assigns units to lanes.
//   keys     every item lane runs gmx_item_loci + gmx_item_key on its own LDS scratch (capacity one item).
//   classes  every item lane compares its key with the group's other keys: the lanes whose key has no equal at a lower
//            lane lead a class; the rank of a class = the number of leaders with smaller keys (std::map order,
//            coverage_common.hpp:133).
//   draw     one seeded draw over non-variant instances + classes (coverage_common.cpp:95-108).
//   record   the leader of the drawn class merges its members' loci and hulls in the group's class scratch
//            (gmx_class_add_item: a set union, the order of the members is immaterial) and records.
// Tasks with more than 16 units, or exceeding a scratch capacity (nothing recorded by then), go to the serial instance
// of the same queue through a reject list. One wave per block, four tasks per wave, persistent over the queue.
// ---------------------------------------------------------------------------
// Scratch sizes per instance. An item's key holds its level-0 sites: a 150-base read inside an MSA region of configs[2]
// (a site every ~20 bases) has 9-12 of them — with room for 6, two thirds of that workload's tasks fell through to the
// one-lane instances (whose keys were as short: the last, global-memory one then took 9 of the batch's 15 ms).
template <int LIST>
struct CoopSizes {  // instances 3 and 2: the regular tasks' general instance, and what the large-capacity search mapped
  typedef CoverEnvT<1, 16, 24, 1, 24> Item;    // one item: its record, key, loci window (and the copy of its traversed list)
  typedef CoverEnvT<1, 1, 48, 48, 8> Class;    // the drawn class: union of loci, hull (no keys; the loci come from the members' windows)
};
template <>
struct CoopSizes<5> {  // instance 5: the instance-searched reads in repeats (many items, short paths)
  typedef CoverEnvT<1, 8, 12, 1> Item;
  typedef CoverEnvT<1, 1, 24, 24> Class;
};
template <int LIST>
constexpr uint32_t gmx_coop_lds_words() {  // + per group: the drawn item's traversed list as path nodes (the walk's handles)
  return 64u * GmxScratchFixed<typename CoopSizes<LIST>::Item>::total + 4u * GmxScratchFixed<typename CoopSizes<LIST>::Class>::total +
         4u * 3u * CoopSizes<LIST>::Item::P_MAX;
}

template <int LIST>
__global__ void __launch_bounds__(64) gmx_cover_coop_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, CoverAcc acc) {
  typedef typename CoopSizes<LIST>::Item CoopItemEnv;
  typedef typename CoopSizes<LIST>::Class CoopClassEnv;
  typedef GmxScratch<CoopItemEnv> SI;
  typedef GmxScratch<CoopClassEnv> SC;
  const uint32_t n = o.counters[(LIST == 5 ? 25 : LIST == 2 ? 7 : GMX_CNT_GENERAL_REST) * GMX_CNT_STRIDE];
  const uint32_t n_first = LIST == 2 ? o.counters[10 * GMX_CNT_STRIDE] : 0u;  // instance 2 starts where instance 4 stopped
  const uint32_t *list = LIST == 5 ? o.inst_mapped_list : LIST == 2 ? o.big_mapped_list : o.general_rest_list;
  uint32_t *reject = LIST == 5 ? o.inst_serial_list : LIST == 2 ? o.big_serial_list : o.general_serial_list;
  uint32_t *reject_n = &o.counters[(LIST == 5 ? 26 : LIST == 2 ? 28 : 27) * GMX_CNT_STRIDE];
  const uint32_t lane = threadIdx.x, grp = lane >> 4, gl = lane & 15u, gbase = grp << 4;
  CoopItemEnv ie;
  ie.scratch = gmx_lds + lane;
  ie.stride = 64;
  CoopClassEnv ce;
  ce.scratch = gmx_lds + 64u * GmxScratchFixed<CoopItemEnv>::total + grp;
  ce.stride = 4;
  ie.acc = ce.acc = acc.acc;
  ie.log = ce.log = acc.log;
  ie.log_cursor = ce.log_cursor = acc.log_cursor;
  ie.log_cap = ce.log_cap = acc.log_cap;
  ie.log_sites = ce.log_sites = acc.log_sites;
  ie.log_at = ce.log_at = 0;
#ifdef GMX_LOOP_STATS
  ie.prof_list = 6;
  ce.prof_list = 7;
#endif
  const uint32_t kofs = SI::keys(ie);  // key word t of lane L: gmx_lds[(kofs + t) * 64 + L]
  for (uint32_t m0 = n_first + blockIdx.x * 4u; m0 < n; m0 += gridDim.x * 4u) {  // wave-uniform: every lane takes part in the shuffles
    const uint32_t m = m0 + grp;
    const bool have = m < n;
#ifdef GMX_LOOP_STATS  // wave-level phase times of this instance: [0] units, [1] loci + keys, [2] classes + draw, [3] class merge + record; [7] rounds
    long long tp = wall_clock64();
#define GMX_COOP_PHASE(k) do { const long long tq = wall_clock64(); if (lane == 0) atomicAdd(&gmx_coop_stats[LIST * 8 + (k)], (unsigned long long)(tq - tp)); tp = tq; } while (0)
    if (lane == 0) atomicAdd(&gmx_coop_stats[LIST * 8 + 7], 1ull);
#else
#define GMX_COOP_PHASE(k) do { } while (0)
#endif
    const uint32_t entry = have ? list[m] : 0u;
    GmxTaskStates ts{0u, 0u, nullptr, nullptr};
    if (have) ts = gmx_entry_states(entry, o, g);
    bool rejected = ts.nf > 16u;
    // --- units ---
    GmxFinalState st{0u, 0u, GMX_NIL, GMX_NIL};
    uint32_t w = 0;
    if (have && !rejected && gl < ts.nf) {
      st = ts.finals[gl];
      w = (st.traversed != GMX_NIL || st.traversing != GMX_NIL || gmx_text_form(st.hi)) ? 1u : min(st.hi - st.lo, 16u) + 1u;
    }
    uint32_t incl = w;
#pragma unroll
    for (uint32_t d = 1; d < 16; d <<= 1) {
      const uint32_t v = __shfl_up(incl, d, 16);
      if (gl >= d) incl += v;
    }
    const uint32_t start = incl - w, n_units = __shfl(incl, 15, 16);
    rejected = rejected || n_units > 16u;
#ifdef GMX_LOOP_STATS
    if (have && gl == 0 && rejected) ie.why(3);  // more than 16 units
#endif
    uint32_t f_lo = 0, f_hi = 0, f_tvd = GMX_NIL, f_tvg = GMX_NIL, f_start = 0;
    bool unit = false;
#pragma unroll 4
    for (uint32_t f = 0; f < 16; ++f) {
      const uint32_t s = __shfl(start, f, 16), ww = __shfl(w, f, 16);
      const uint32_t lo = __shfl(st.lo, f, 16), hi = __shfl(st.hi, f, 16), tvd = __shfl(st.traversed, f, 16), tvg = __shfl(st.traversing, f, 16);
      if (gl >= s && gl < s + ww) {
        unit = true;
        f_lo = lo;
        f_hi = hi;
        f_tvd = tvd;
        f_tvg = tvg;
        f_start = s;
      }
    }
    unit = unit && have && !rejected;
    bool is_item = false, nonvar = false;
    uint32_t i_lo = 0, i_hi = 0, enc_site = 0;
    int32_t enc_allele = -1;
    if (unit) {
      if (f_tvd != GMX_NIL || f_tvg != GMX_NIL) {
        is_item = true;
        i_lo = f_lo;
        i_hi = f_hi;
      } else {
        const uint32_t i = f_lo + (gl - f_start);
        const GmxNode &nd = ix.nodes[ix.pos_node[gmx_occ_pos(ix, f_hi, i)]];
        if (nd.site == 0) {
          nonvar = true;
        } else {
          is_item = true;
          i_lo = i;
          i_hi = gmx_text_form(f_hi) ? f_hi : i;
          enc_site = nd.site;
          enc_allele = nd.allele;
        }
      }
    }
    const uint32_t items16 = (uint32_t)(__ballot(is_item) >> gbase) & 0xFFFFu;
    const uint32_t nonvariant = __popc((uint32_t)(__ballot(nonvar) >> gbase) & 0xFFFFu);
    GMX_COOP_PHASE(0);
    // --- loci and key of the lane's item ---
    ie.arena = ts.arena;
    ie.status = GMX_TASK_MAPPED;
    if (is_item) {
      ie.sset(SI::items + 0, i_lo);
      ie.sset(SI::items + 1, i_hi);
      ie.sset(SI::items + 2, f_tvd);
      ie.sset(SI::items + 3, f_tvg);
      ie.sset(SI::items + 4, enc_site);
      ie.sset(SI::items + 5, (uint32_t)enc_allele);
      const uint32_t nl = gmx_item_loci(ix, ie, 0, 0);
      if (nl != 0xFFFFFFFFu) {
        gmx_item_key(ix, ie, 0, 0, nl);
        ie.sset(SI::order(ie), nl);  // (the order word is free with one item: the class's first lane reads the window's length here)
      }
    }
    uint32_t err = (is_item && ie.status != GMX_TASK_MAPPED && ie.status != GMX_TASK_OVERFLOW) ? ie.status : 0u;
    rejected = rejected || (((uint32_t)(__ballot(is_item && ie.status == GMX_TASK_OVERFLOW) >> gbase) & 0xFFFFu) != 0u);
    bool failed = (((uint32_t)(__ballot(err != 0u) >> gbase) & 0xFFFFu) != 0u);
    __syncthreads();  // the keys are in LDS
    GMX_COOP_PHASE(1);
    // --- classes ---
    uint32_t lt = 0, eq = 0;
    if (is_item && !rejected && !failed) {
      const uint32_t la = gmx_lds[kofs * 64u + lane];
      for (uint32_t rest = items16 & ~(1u << gl); rest; rest &= rest - 1u) {
        const uint32_t j = (uint32_t)__ffs(rest) - 1u, other = gbase + j;
        const uint32_t lb = gmx_lds[kofs * 64u + other];
        const uint32_t mlen = min(la, lb);
        int cmp = 0;  // sign of (other's key - mine)
        for (uint32_t t = 0; t < mlen && cmp == 0; ++t) {
          const uint32_t va = gmx_lds[(kofs + 1u + t) * 64u + lane], vb = gmx_lds[(kofs + 1u + t) * 64u + other];
          cmp = vb < va ? -1 : (vb > va ? 1 : 0);
        }
        if (cmp == 0) cmp = lb < la ? -1 : (lb > la ? 1 : 0);
        if (cmp < 0) lt |= 1u << j;
        if (cmp == 0) eq |= 1u << j;
      }
    }
    const bool leader = is_item && !rejected && !failed && (eq & ((1u << gl) - 1u)) == 0u;
    const uint32_t leaders16 = (uint32_t)(__ballot(leader) >> gbase) & 0xFFFFu;
    const uint32_t n_classes = __popc(leaders16), rank = __popc(lt & leaders16);
    // --- the draw ---
    bool member = false;
    if (have && !rejected && !failed && items16 != 0u) {
      uint32_t r = 0;
      if (!gmx_uniform_1_to_n(b.seeds[ts.task >> 1], nonvariant + n_classes, acc.rng_mode, r)) {
        err = GMX_TASK_ERROR;
      } else if (r > nonvariant) {
        member = is_item && rank == r - nonvariant - 1u;
      }
    }
    const uint32_t members16 = (uint32_t)(__ballot(member) >> gbase) & 0xFFFFu;
    GMX_COOP_PHASE(2);
    // --- the drawn class: its first lane merges the members and records ---
    bool class_overflow = false, class_logfull = false;
    if (member && gl == (uint32_t)__ffs(members16) - 1u) {
      const uint32_t read = ts.task >> 1;
      const uint32_t len = read_len(b, read);
      ce.arena = ts.arena;
      ce.status = GMX_TASK_MAPPED;
      ce.log_at = 0;
      uint32_t n_loci = 0, n_hull = 0;
      bool ok = true;
      for (uint32_t rest = members16; rest && ok; rest &= rest - 1u) {
        const uint32_t other = gbase + (uint32_t)__ffs(rest) - 1u;
#pragma unroll
        for (uint32_t t = 0; t < SI::ITEM_W; ++t) ce.sset(SC::items + t, gmx_lds[(SI::items + t) * 64u + other]);
        // the member's loci window as its lane left it (gmx_class_add_item would run gmx_item_loci again): set union
        const uint32_t nl_m = gmx_lds[SI::order(ie) * 64u + other], first = n_loci;
        for (uint32_t i = 0; i < nl_m && ok; ++i) {
          const uint32_t site = gmx_lds[(SI::loci(ie) + 2u * i) * 64u + other], al = gmx_lds[(SI::loci(ie) + 2u * i + 1u) * 64u + other];
          bool dup = false;
          for (uint32_t j = 0; j < first && !dup; ++j) dup = ce.sget(SC::loci(ce) + 2u * j) == site && ce.sget(SC::loci(ce) + 2u * j + 1u) == al;
          if (dup) continue;
          if (n_loci >= ce.loc_max()) {
            ce.fail(GMX_TASK_OVERFLOW);
            ok = false;
            break;
          }
          ce.sset(SC::loci(ce) + 2u * n_loci, site);
          ce.sset(SC::loci(ce) + 2u * n_loci + 1u, al);
          ++n_loci;
        }
        // The walk consumes the member's traversed list newest first, a dependent arena load per locus (the fast
        // pass's arena keeps a task's nodes n_tasks entries apart: every one a miss). The member's lane has copied
        // the list to its scratch: laid out as path nodes in LDS, handle = index, the walk never leaves the CU for it.
        const uint32_t nt_m = gmx_lds[(SI::path(ie) + 2u * CoopItemEnv::P_MAX) * 64u + other];
        if (nt_m != 0xFFFFFFFFu && nt_m != 0u && ce.sget(SC::items + 2) != GMX_NIL) {
          GmxPathNode *ln = reinterpret_cast<GmxPathNode *>(gmx_lds + 64u * GmxScratchFixed<CoopItemEnv>::total +
                                                            4u * GmxScratchFixed<CoopClassEnv>::total + grp * 3u * CoopItemEnv::P_MAX);
          for (uint32_t i = 0; i < nt_m; ++i)
            ln[i] = GmxPathNode{gmx_lds[(SI::path(ie) + 2u * i) * 64u + other], (int32_t)gmx_lds[(SI::path(ie) + 2u * i + 1u) * 64u + other],
                                i + 1u < nt_m ? i + 1u : GMX_NIL};
          ce.arena = ln;
          ce.sset(SC::items + 2, 0u);
        } else {
          ce.arena = ts.arena;
        }
        ok = ok && gmx_item_per_base(ix, ce, 0, len, n_hull);
      }
      ce.arena = ts.arena;
      if (ok) gmx_class_record(ix, ce, n_loci, n_hull);
      class_overflow = ce.status == GMX_TASK_OVERFLOW;
      class_logfull = ce.status == GMX_TASK_LOGFULL;
      if (ce.status != GMX_TASK_MAPPED && !class_overflow && !class_logfull) err = ce.status;
      ce.log_abandon();
    }
    rejected = rejected || (((uint32_t)(__ballot(class_overflow) >> gbase) & 0xFFFFu) != 0u);
    const bool logfull = (((uint32_t)(__ballot(class_logfull) >> gbase) & 0xFFFFu) != 0u);
    if (have && gl == 0 && logfull)
      o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
    if (have && gl == 0 && rejected) reject[atomicAdd(reject_n, 1u)] = entry;
    if (err != 0u && atomicCAS(&o.error[0], 0u, err) == 0u) o.error[1] = ts.task;
    __syncthreads();  // the scratch is reused by the next round
    GMX_COOP_PHASE(3);
  }
#undef GMX_COOP_PHASE
}

// Path handles of a GmxCoverRec: traversed loci are addressed by their index in the record (newest first), the
// traversing path is an inline handle (gmx_types.h) or nil.
struct CompactRec {  // a GmxCoverRec in scalars (with the array member the compiler kept the record in scratch memory and indexed it)
  uint32_t p, len_n, tvg, s0, s1, s2, a01, a2;
  __device__ __forceinline__ CompactRec &operator=(const GmxCoverRec &r) {
    p = r.p, len_n = r.len_n, tvg = r.tvg, s0 = r.site[0], s1 = r.site[1], s2 = r.site[2], a01 = r.a01, a2 = r.a2;
    return *this;
  }
};
struct CompactEnv : CoverLogPart {
  CompactRec rec;
  __device__ __forceinline__ uint32_t n_trav() const { return (rec.len_n >> 16) & 31u; }
  __device__ __forceinline__ bool run_form() const { return (rec.len_n & GMX_REC_RUN_FLAG) != 0; }
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return 5u + 2u * (h & ~GMX_INLINE_FLAG);
    if (run_form()) return rec.s0 + 2u * h;
    return h == 0 ? rec.s0 : (h == 1 ? rec.s1 : rec.s2);
  }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return -1;
    if (run_form()) {
      const uint32_t q = h >> 2, w = q == 0 ? rec.s1 : q == 1 ? rec.s2 : q == 2 ? rec.a01 : rec.a2;
      return (int32_t)((w >> (8u * (h & 3u))) & 0xFFu);
    }
    return (int32_t)(h == 0 ? (rec.a01 & 0xFFFFu) : (h == 1 ? (rec.a01 >> 16) : rec.a2));
  }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return GMX_NIL;
    return h + 1 < n_trav() ? h + 1 : GMX_NIL;
  }
};

// The common case, one lane per compact record and no scratch (gmx_cover_single, gmx_cover.h): a task with ONE
// final state of width one. Few registers, a coalesced queue, region-local tables.
template <bool NESTED>  // (two kernels: the nested routine's locus arrays would cost the flat one registers and scratch)
__device__ __forceinline__ void gmx_cover_single_rec(const GmxIndexView &ix, const SearchOut &o, const CoverAcc &acc, size_t rec_idx,
                                                     uint32_t *handoff_list, uint32_t handoff_counter) {
  CompactEnv env;
  env.rec = o.cover_recs[rec_idx];
  env.acc = acc.acc;
  env.log = acc.log;
  env.log_cursor = acc.log_cursor;
  env.log_cap = acc.log_cap;
  env.log_sites = acc.log_sites;
  env.status = GMX_TASK_MAPPED;
  env.log_at = 0;
  const GmxFinalState st{env.rec.p, GMX_TEXT_MARK, env.n_trav() ? 0u : GMX_NIL, env.rec.tvg};
  if constexpr (!NESTED) {
    gmx_cover_single(ix, env, st, env.rec.len_n & 0xFFFFu);
  } else if (!gmx_cover_single_nested(ix, env, st, env.rec.len_n & 0xFFFFu)) {  // many loci: the general instance next
    handoff_list[atomicAdd(&o.counters[handoff_counter * GMX_CNT_STRIDE], 1u)] = o.cover_rec_task[rec_idx];
  }
  if (env.status == GMX_TASK_LOGFULL) {  // nothing recorded: again once the host has drained the log
    o.log_retry_recs[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE], 1u)] = (uint32_t)rec_idx;
  } else if (env.status != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, env.status) == 0u) {
    o.error[1] = o.cover_rec_task[rec_idx];
  }
  env.log_abandon();
}

template <bool NESTED>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t region = blockIdx.x & (GMX_REGIONS - 1);  // = the XCD this workgroup runs on (round-robin dispatch)
  const uint32_t n_mapped = o.counters[(16 + region) * GMX_CNT_STRIDE];
  const uint32_t m = (blockIdx.x / GMX_REGIONS) * GMX_BLOCK + threadIdx.x;
  if (m >= n_mapped) return;
  gmx_cover_single_rec<NESTED>(ix, o, acc, (size_t)region * o.region_cap + m, o.cover_general_list, 8u);
}

// The same queue on a flat PRG whose sites have geometry records (GmxSiteGeo): gmx_cover_jump alone — no walk, no GmxSite, the
// increments staged in LDS between its check pass and the recording — and what it declines (a site of more than 8 alleles
// or an allele of 255+ bases on the path) goes to gmx_cover_single_rest_kernel, the routine above over a list. (Forced to 64
// registers for 8 waves per SIMD it spills and is no faster: GMX_JUMP_MIN_BLOCKS; profiles/round4/cover_jump_variants_config3.txt.)
struct StageLds {
  uint32_t *w;  // this lane's words, GMX_BLOCK apart
  __device__ __forceinline__ uint32_t cap() const { return GMX_STAGE_MAX; }
  __device__ __forceinline__ void put(uint32_t i, uint32_t v) { w[i * GMX_BLOCK] = v; }
  __device__ __forceinline__ uint32_t get(uint32_t i) const { return w[i * GMX_BLOCK]; }
};
#ifndef GMX_JUMP_MIN_BLOCKS
#define GMX_JUMP_MIN_BLOCKS 1  // (8 = 64 registers, 8 waves per SIMD with spills: measured slower, and its LDS crowds out the side streams' kernels)
#endif
__global__ void __launch_bounds__(GMX_BLOCK, GMX_JUMP_MIN_BLOCKS) gmx_cover_jump_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t region = blockIdx.x & (GMX_REGIONS - 1);
  const uint32_t n_mapped = o.counters[(16 + region) * GMX_CNT_STRIDE];
  const uint32_t m = (blockIdx.x / GMX_REGIONS) * GMX_BLOCK + threadIdx.x;
  if (m >= n_mapped) return;
  const size_t rec_idx = (size_t)region * o.region_cap + m;
  CompactEnv env;
  env.rec = o.cover_recs[rec_idx];
  env.acc = acc.acc;
  env.log = nullptr;
  env.log_cursor = nullptr;
  env.log_cap = 0;
  env.log_sites = 0;
  env.status = GMX_TASK_MAPPED;
  env.log_at = env.log_end = 0;
  const uint32_t p = env.rec.p, tvd = env.n_trav() ? 0u : GMX_NIL, tvg = env.rec.tvg;
  StageLds stage{gmx_lds + threadIdx.x};
  const bool done = gmx_cover_jump(ix, env, stage, p, tvd, tvg, env.rec.len_n & 0xFFFFu);
  if (!done) o.single_rest_list[atomicAdd(&o.counters[GMX_CNT_SINGLE_REST * GMX_CNT_STRIDE], 1u)] = (uint32_t)rec_idx;
}
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_rest_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t n = o.counters[GMX_CNT_SINGLE_REST * GMX_CNT_STRIDE];
  for (uint32_t i = blockIdx.x * GMX_BLOCK + threadIdx.x; i < n; i += gridDim.x * GMX_BLOCK)
    gmx_cover_single_rec<false>(ix, o, acc, o.single_rest_list[i], o.cover_general_list, 8u);
}

// ---- grouped log full: the batch's failed entries again, after the host has drained the log (launch_log_replay) ----
// moves the retry lists' lengths to where the replay kernels read them and empties the retry lists for this round
__global__ void gmx_log_replay_setup_kernel(SearchOut o, const uint32_t *retry_huge_in) {
  uint32_t *c = o.counters;
  const uint32_t n_entries = c[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], n_recs = c[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE],
                 n_huge = c[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE];
  for (uint32_t i = threadIdx.x; i < n_huge; i += blockDim.x) o.huge_list[i] = retry_huge_in[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    c[4 * GMX_CNT_STRIDE] = n_entries;                 // gmx_cover_kernel<CoverEnvBig, 1> reads its queue length here
    c[GMX_CNT_REPLAY_RECS * GMX_CNT_STRIDE] = n_recs;
    c[11 * GMX_CNT_STRIDE] = n_huge;                   // the last tier's search items
    c[15 * GMX_CNT_STRIDE] = 0;
    c[14 * GMX_CNT_STRIDE] = 0;                        // the ticket counter of the last-tier stage
    c[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE] = 0;
    c[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE] = 0;
    c[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE] = 0;
  }
}
template <bool NESTED>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_replay_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc,
                                                                            const uint32_t *recs_in) {
  const uint32_t n = o.counters[GMX_CNT_REPLAY_RECS * GMX_CNT_STRIDE];
  for (uint32_t i = blockIdx.x * GMX_BLOCK + threadIdx.x; i < n; i += gridDim.x * GMX_BLOCK)
    gmx_cover_single_rec<NESTED>(ix, o, acc, recs_in[i], o.cover_overflow_list, 4u);  // (nested, many loci: the large scratch, which runs next)
}

// The five uint64 read counters <-> 16-bit limbs in uint32 words, so that they travel inside the one uint32
// all-reduce(sum) of the coverage block: limb sums of up to 65536 ranks cannot overflow (gmx_coverage_reduce_*).
__global__ void gmx_stats_limbs_kernel(unsigned long long *stats, uint32_t *limbs, int recombine) {
  const uint32_t t = threadIdx.x;
  if (!recombine) {
    if (t < 20) limbs[t] = (uint32_t)((stats[t >> 2] >> (16 * (t & 3))) & 0xFFFFull);
    else if (t < 32) limbs[t] = 0;
  } else if (t < 5) {
    unsigned long long v = 0;
    for (int l = 3; l >= 0; --l) v = (v << 16) + limbs[4 * t + l];  // limb sums carry into the limbs above
    stats[t] = v;
  }
}

// Validation + packing, one lane per read. Reads holding a byte outside 1..4 are skipped as a whole
// (encode_dna_bases, utils.cpp:73-92). The packed form is two bit planes per 32 bases (uint2: low bits, high
// bits of the codes 0..3): the search kernels compare 32 bases per step against the PRG's planes (GmxTextRec),
// and a single base is two bit extracts.
//
// A block owns GMX_PACK_READS consecutive reads, whose bytes and whose packed pairs are both contiguous:
// the bytes are staged through LDS with coalesced 16-byte loads, packed from LDS (aligned dwords joined with
// v_alignbyte), and written back from LDS with coalesced stores. Blocks whose reads do not fit the LDS
// window (very long reads) take the direct per-lane path.
#define GMX_PACK_READS 128
#define GMX_PACK_IN_BYTES (24 * 1024)
#define GMX_PACK_OUT_PAIRS (GMX_PACK_IN_BYTES / 32 + GMX_PACK_READS + 8)
typedef uint32_t __attribute__((aligned(1))) gmx_u32_unaligned;
// four bytes -> four bits of each plane (bit i = byte i), flagging bytes outside 1..4
__device__ __forceinline__ void pack4(uint32_t x, uint32_t &lo, uint32_t &hi, uint32_t &bad) {
  uint32_t y = x - 0x01010101u;                         // per-byte code 0..3 when every byte is in 1..4
  bad |= ((y & ~x & 0x80808080u) | (y & 0xFCFCFCFCu));  // a zero byte, or a byte > 4
  lo = (((y & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
  hi = ((((y >> 1) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint2 pack_tail(const uint8_t *p, uint32_t rem, uint32_t &bad) {
  uint2 out = make_uint2(0, 0);
  for (uint32_t j = 0; j < rem; ++j) {
    uint32_t x = p[j];
    if (x < 1 || x > 4) bad = 1;
    out.x |= ((x - 1u) & 1u) << j;
    out.y |= (((x - 1u) >> 1) & 1u) << j;
  }
  return out;
}
#define GMX_PACK_THREADS (2 * GMX_PACK_READS)  // two threads per read: twice the loads and stores in flight per LDS window
__global__ void __launch_bounds__(GMX_PACK_THREADS) gmx_pack_kernel(BatchView b, uint8_t *skip, uint2 *packed, uint32_t *counters,
                                                                    uint32_t *zero, uint32_t zero_words) {
  __shared__ uint4 in4[GMX_PACK_IN_BYTES / 16 + 2];
  __shared__ uint2 outp[GMX_PACK_OUT_PAIRS];
  // the queue counters are per batch: this is the batch's first kernel and everything that counts comes after it
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < GMX_N_COUNTERS * GMX_CNT_STRIDE; i += GMX_PACK_THREADS) counters[i] = 0;
  // a reset queued just ahead of this batch (gmx_engine_reset_async): the accumulator block, read counters and log
  // cursor zeroed here instead of by a memset of their own (nothing in this kernel touches them otherwise)
  for (uint32_t i = blockIdx.x * GMX_PACK_THREADS + threadIdx.x; i < zero_words; i += gridDim.x * GMX_PACK_THREADS) zero[i] = 0;
  const uint32_t r0 = blockIdx.x * GMX_PACK_READS;
  const uint32_t r1 = min(r0 + GMX_PACK_READS, b.n_reads);
  const uint32_t read = r0 + (threadIdx.x >> 1), half = threadIdx.x & 1u;  // the two threads of a read are neighbours
  const uint64_t s0 = b.offsets[r0], s1 = b.offsets[r1];
  const uintptr_t g0 = reinterpret_cast<uintptr_t>(b.reads + s0);
  const uint32_t shift = (uint32_t)(g0 & 15u);
  const uint64_t span = (s1 - s0) + shift;
  if (span <= GMX_PACK_IN_BYTES) {  // block-uniform
    const uint4 *src = reinterpret_cast<const uint4 *>(g0 - shift);
    const uint32_t n16 = (uint32_t)((span + 15) >> 4);
    {  // independent 16-byte loads in flight per thread and round
      uint32_t i = threadIdx.x;
      for (; i + 2 * GMX_PACK_THREADS < n16; i += 3 * GMX_PACK_THREADS) {
        const uint4 v0 = src[i], v1 = src[i + GMX_PACK_THREADS], v2 = src[i + 2 * GMX_PACK_THREADS];
        in4[i] = v0;
        in4[i + GMX_PACK_THREADS] = v1;
        in4[i + 2 * GMX_PACK_THREADS] = v2;
      }
      for (; i < n16; i += GMX_PACK_THREADS) in4[i] = src[i];
    }
    const uint64_t po0 = pack_off(b, r0);
    const uint32_t n_out = (uint32_t)(pack_off(b, r1) - po0);
    for (uint32_t i = threadIdx.x; i < n_out; i += GMX_PACK_THREADS) outp[i] = make_uint2(0, 0);
    __syncthreads();
    if (read < r1) {
      const uint64_t s = b.offsets[read];
      const uint32_t len = (uint32_t)(b.offsets[read + 1] - s);
      const uint32_t q = shift + (uint32_t)(s - s0);
      const uint32_t *w = reinterpret_cast<const uint32_t *>(in4);
      const uint8_t *bytes = reinterpret_cast<const uint8_t *>(in4);
      uint2 *out = outp + (uint32_t)(pack_off(b, read) - po0);
      const uint32_t full = len >> 5, first_half = (full + 1u) >> 1;
      const uint32_t c0 = half ? first_half : 0u, c1 = half ? full : first_half;  // this thread's pairs
      uint32_t idx = (q >> 2) + 8u * c0;
      const uint32_t sh = q & 3u;
      uint32_t bad = 0;
      uint32_t carry = w[idx];
      for (uint32_t c = c0; c < c1; ++c) {
        uint2 pair = make_uint2(0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          uint32_t nxt = w[++idx], lo, hi;
          pack4(__builtin_amdgcn_alignbyte(nxt, carry, sh), lo, hi, bad);
          pair.x |= lo << (4 * j);
          pair.y |= hi << (4 * j);
          carry = nxt;
        }
        out[c] = pair;
      }
      const uint32_t rem = len & 31u;
      if (half && rem) out[full] = pack_tail(bytes + q + full * 32, rem, bad);
      bad |= (uint32_t)__shfl_xor((int)bad, 1);
      if (!half) skip[read] = bad ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_out; i += GMX_PACK_THREADS) packed[po0 + i] = outp[i];
    return;
  }
  if (read >= r1 || half) return;
  uint64_t s = b.offsets[read], e = b.offsets[read + 1];
  uint32_t len = (uint32_t)(e - s);
  const uint8_t *p = b.reads + s;
  uint2 *out = packed + pack_off(b, read);
  uint32_t bad = 0;
  uint32_t full = len >> 5;
  for (uint32_t c = 0; c < full; ++c) {
    uint2 pair = make_uint2(0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t lo, hi;
      pack4(*reinterpret_cast<const gmx_u32_unaligned *>(p + c * 32 + j * 4), lo, hi, bad);
      pair.x |= lo << (4 * j);
      pair.y |= hi << (4 * j);
    }
    out[c] = pair;
  }
  uint32_t rem = len & 31u;
  if (rem) out[full] = pack_tail(p + full * 32, rem, bad);
  skip[read] = bad ? 1 : 0;
}
