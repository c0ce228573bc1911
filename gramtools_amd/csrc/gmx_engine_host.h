// gmx_engine_host.h — part of the ONE translation unit gmx_engine.hip (included there): the engine object, its creation (index
// upload, shared per device), the batch pipeline's launches on three streams, the feeds of the C ABI (bytes, bit planes, 2-bit
// stream, planes already in HBM), the grouped log's accounting, coverage read-back.
#pragma once
// ===========================================================================
// engine (host side of the device half)
// ===========================================================================
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gmx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
      return GMX_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

// kernels timed one by one besides gmx_extend_kernel (gmx_timing::kernel_ms; include/gmx.h lists them)
enum : int { GMX_TK_SEED = 0, GMX_TK_FILTER0, GMX_TK_FILTER1, GMX_TK_SINGLE, GMX_TK_EXTEND2, GMX_TK_UNPACK, GMX_TK_N };
static_assert(GMX_TK_N <= GMX_TIMED_KERNELS, "gmx_timing::kernel_ms holds GMX_TIMED_KERNELS entries");

struct gmx_engine {
  gmx_engine_opts opts;
  GmxIndexView dview;  // device pointers
  std::vector<void *> allocs;
  uint64_t index_bytes = 0;
  // accumulators
  uint32_t *d_fused = nullptr, *d_limbs = nullptr;  // accumulator block (n_acc words, gmx_types.h) | 32 counter-limb words
  size_t n_fused = 0, n_acc = 0;
  std::vector<uint32_t> phys_allele, phys_pb, phys_grouped;  // logical slot -> slot of the block (gmx_coverage_fetch)
  std::vector<uint32_t> hit_fix;                             // hit counters and the logical slots they count for
  unsigned long long *d_stats = nullptr;  // with d_log_cursor behind the coverage block: one memset resets all of it
  uint32_t *d_error = nullptr;
  uint32_t *d_log = nullptr, *d_log_cursor = nullptr;
  uint32_t log_cap = 0;
  uint32_t n_allele = 0, n_pb = 0, n_grouped = 0;
  // batch workspace (sized for max_batch_reads)
  uint64_t cap_reads = 0;
  uint8_t *d_skip = nullptr;
  uint2 *d_packed = nullptr;
  uint64_t cap_packed = 0;
  uint32_t *d_status = nullptr, *d_n_final = nullptr, *d_mapped = nullptr, *d_overflow = nullptr, *d_counters = nullptr;
  uint32_t *d_general_rest = nullptr;
  uint32_t *d_single_rest = nullptr;
  bool cover_jump = false;  // gmx_cover_jump_kernel + gmx_cover_single_rest_kernel instead of gmx_cover_single_kernel<false>
  uint32_t *d_task_lists = nullptr;  // SearchOut::task_lists: d_overflow, d_overflow2, d_alive, d_dead, d_dead2, d_cover_general are its slices
  GmxSeed *d_alive_seed = nullptr;
  uint32_t *d_alive = nullptr, *d_dead = nullptr, *d_dead2 = nullptr;
  uint64_t *d_seed_cursor = nullptr;
  bool seed_cursor = false;  // the index has many multi-state k-mer entries: kernels instantiated with the seed cursor
  GmxFinalState *d_finals = nullptr;
  GmxPathNode *d_arena = nullptr;
  GmxCoverRec *d_cover_recs = nullptr;
  BigOut big{};
  uint32_t *d_scratch_big = nullptr, *d_cover_overflow = nullptr;
  uint32_t cover_big_lanes = 0;
  uint32_t *d_big_mapped = nullptr, *d_cover_general = nullptr, *d_cover_mid = nullptr, *d_overflow2 = nullptr;
  uint32_t *d_huge = nullptr, *d_cover_huge = nullptr, *d_huge_retry = nullptr;  // the last tier's queues (gmx_tail_stage)
  uint32_t *d_inst_list = nullptr, *d_inst_sa = nullptr, *d_inst_remaining = nullptr, *d_inst_mapped = nullptr;  // instance lanes (gmx_extend_inst_kernel)
  uint32_t inst_cap = 0;
  GmxPathNode *d_inst_arena = nullptr;
  GmxFinalState *d_inst_states = nullptr;
  uint32_t *d_inst_first = nullptr, *d_inst_width = nullptr;
  uint32_t *d_inst_serial = nullptr, *d_general_serial = nullptr, *d_big_serial = nullptr, *d_overflow3 = nullptr;  // what gmx_cover_coop_kernel leaves to the serial instances
  bool coop = true;  // GMX_NO_COOP=1 in the environment: serial coverage instances only (A/B runs)
  uint32_t *d_heap = nullptr;      // ... and its memory
  uint64_t heap_words = 0;
  bool log_sites = false;          // the index has sites with more than 8 alleles
  // grouped log: drained into `log_counts` (records with counts) whenever the device log may run full, and at fetch time
  std::map<std::vector<uint32_t>, uint64_t> log_counts;  // key = [site_index, ids...]
  // Exact accounting (round 3): after every batch of an engine whose index uses the log, the log cursor and the lengths
  // of the three retry lists are copied to page-locked words; before the next batch (and before any reader of the
  // coverage) log_settle() looks at them: entries that found the log full are redone after a drain (launch_log_replay),
  // and the log is drained once it is half full. No assumed bound on what a read appends.
  uint32_t *h_log_state = nullptr;     // [cursor, retry entries, retry records, retry last-tier tasks]
  hipEvent_t ev_log_state = nullptr;
  bool log_state_pending = false;
  uint32_t *d_log_retry[2] = {nullptr, nullptr}, *d_log_retry_recs[2] = {nullptr, nullptr}, *d_log_retry_huge[2] = {nullptr, nullptr};
  int log_retry_side = 0;              // which of the two sets the kernels append to
  BatchView last_b{};
  SearchOut last_o{};
  CoverAcc last_acc{};
  size_t last_big_lds = 0;
  uint64_t log_replays = 0, log_replayed_entries = 0;  // statistics (tests)
  uint64_t log_known = 0;          // log words in use after the last drain / look ...
  uint64_t log_reads_since = 0;    // (unused since round 3: the fill is read back after every batch)
  // gmx_engine_reset_async leaves its memset pending: the next batch's pack kernel zeroes the block when it is launched
  // on the same stream (one command and one dependent-launch gap less per job); every other reader of the accumulators
  // issues the memset first (flush_reset)
  bool reset_pending = false;
  hipStream_t reset_stream = nullptr;
  hipStream_t side2_stream = nullptr;
  hipEvent_t ev_fork2 = nullptr, ev_side1 = nullptr, ev_filter = nullptr;
  hipStream_t side_stream = nullptr;  // large-capacity search + its coverage run beside filter/cover
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_wait = nullptr;  // blocking event of gmx_quiesce
  struct GmxDeviceIndex *shared_index = nullptr;  // the device copy of the index tables, shared with the other engines of this index on this device
  uint32_t filter_lds_words = 0;  // > 0: the k-mer presence bitmap fits LDS (gmx_filter_lds_kernel)
  const uint32_t *d_kmer_planar = nullptr;  // that bitmap indexed by planar k-mer code (all_kmers_present_planar)
  const uint32_t *d_absent = nullptr;       // the k-mers that do NOT occur, when they are few (gmx_filter_absent_kernel)
  uint32_t n_absent = 0;
  bool use_absent = false;
  uint32_t n_cus = 256;
  uint32_t probe_iters = GMX_PROBE_ITERS;  // wave-loop iterations before the probe kernel parks what is left
  uint32_t extend_cap = 0;      // iterations of the LAST pass after which a task goes to the large-capacity route (0: runs to the end)
  uint32_t extend_budget = 8;   // wave-loop iterations of the extend kernel before a lane with work left is parked for the second pass
  bool seeds_in_place = false;  // gmx_engine_seeds_in_place
  // A second batch in flight (round 6). Every batch ends with a tail of few-lane kernels — on a nested PRG ~2 ms of a few straggler
  // tasks on a handful of CUs — during which the GPU is all but idle. The TWIN is a second workspace with streams of its own that
  // shares this engine's index and ACCUMULATORS (coverage is atomic adds: two batches may record side by side): the host feeds hand
  // consecutive launches to engine and twin in turn, so launch i + 1's full-GPU kernels run beside launch i's tail. Measured with
  // two engines per device in round 5 (tools/exp/engines_in_flight.py): configs[2] 388 -> 499 M reads/s at 1 M reads per launch,
  // configs[3] + 11 %, configs[1] +- 0 — so the twin exists for nested PRGs and indexes of 2 GB and more (GMX_TWIN=0 / 1 forces),
  // never for an index whose sites use the grouped log (its replay is per batch).
  gmx_engine *twin = nullptr;
  bool is_twin = false, twin_off = false;
  hipStream_t main_stream = nullptr;  // the twin's main chain (the engine's own: the NULL stream, or the caller's)
  uint32_t twin_toggle = 0;
  hipEvent_t ev_zeroed = nullptr, ev_twin_done = nullptr;  // a queued reset has executed | the twin's last batch has ended
  uint64_t zero_epoch = 0, seen_zero_epoch = 0;
  bool twin_in_flight = false;   // the twin has launched since the last join
  bool last_on_twin = false;     // the last launch went to the twin (gmx_engine_queue_counts reads that workspace's counters)
  uint32_t extend_passes = 1;   // launches over the stragglers (<= GMX_EXTRA_PASSES); all but the last with a budget of their own
  uint32_t extend_budget2[GMX_EXTRA_PASSES] = {24, 96, 0};
                                // (GMX_EXTEND_BUDGET in the environment; 0 = one pass)
  GmxParked *d_park2 = nullptr;
  uint32_t *d_park2_n = nullptr;
  // test hook (gmx_engine_debug_keep_states, gmx_debug_final_states): the last batch's per-task search results stay readable
  bool keep_states = false;
  uint64_t keep_reads = 0;            // reads of that batch
  std::vector<uint32_t> debug_isa;    // inverse suffix array (text position -> SA index), fetched on first use
  uint32_t fuse = 1;  // fused transitions in the extend kernel's wave loop (GMX_NO_FUSE=1 in the environment: off, for A/B runs)
  // host staging for the _host entry point
  uint8_t *d_reads = nullptr;
  uint64_t *d_offsets = nullptr;
  uint32_t *d_seeds = nullptr;
  uint64_t cap_bases = 0, cap_stage_reads = 0;
  // gmx_map_reads_host, pipelined: two staging slots (device buffers + pinned offsets/seeds), a copy stream
  struct StageSlot {
    uint8_t *d_reads = nullptr;
    uint64_t *d_offsets = nullptr, *h_offsets = nullptr;
    uint32_t *d_seeds = nullptr, *h_seeds = nullptr;
    uint64_t cap_bases = 0, cap_reads = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
  } stage[2];
  hipStream_t copy_stream = nullptr;
  hipStream_t copy_stream2 = nullptr;  // the packed feeds' uploads alternate between the two (gmx_map_reads_packed_host / _2bit_host)
  uint32_t copy_toggle = 0;
  hipStream_t last_stream = nullptr;
  // gmx_map_reads_packed_host: three slots of device buffers for bit planes, offsets, seeds and skip flags; the upload of
  // a chunk (copy stream, straight from the caller's page-locked buffers) runs beside the kernels of the chunks before
  struct PackSlot {
    uint2 *d_planes = nullptr;
    uint64_t *d_offsets = nullptr;
    uint32_t *d_seeds = nullptr;
    uint8_t *d_skip = nullptr;
    uint64_t cap_pairs = 0, cap_reads = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
  } pslot[3];
  uint32_t pslot_next = 0;
  // releases a device buffer obtained from alloc() before the engine is destroyed (superseded staging buffers)
  void release(void *q) {
    if (!q) return;
    for (size_t i = 0; i < allocs.size(); ++i)
      if (allocs[i] == q) {
        allocs[i] = allocs.back();
        allocs.pop_back();
        (void)hipFree(q);
        return;
      }
  }
  // optional HIP-event timing of the kernels (bench.py roofline leg)
  bool timing = false;
  struct EvTriple { hipEvent_t s, a, b, c; uint64_t reads; hipEvent_t k[GMX_TIMED_KERNELS][2]; uint32_t timed; };
  std::vector<EvTriple> pending;
  double search_ms = 0, cover_ms = 0;
  uint64_t search_launches = 0, cover_launches = 0, timed_reads = 0;
  double kernel_ms[GMX_TIMED_KERNELS] = {0};        // gmx_timing::kernel_ms (GMX_TK_*)
  uint64_t kernel_launches[GMX_TIMED_KERNELS] = {0};

  template <class T>
  int alloc(T **p, size_t count, bool zero) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    allocs.reserve(allocs.size() + 1);  // (may throw BEFORE there is device memory to lose)
    HIP_TRY(hipMalloc(&q, bytes));
    if (zero) HIP_TRY(hipMemset(q, 0, bytes));
    allocs.push_back(q);
    *p = (T *)q;
    return GMX_OK;
  }
  template <class T, class A>
  int upload(const T **dst, const std::vector<T, A> &src) {
    T *q = nullptr;
    int rc = alloc(&q, src.size(), false);
    if (rc) return rc;
    if (!src.empty()) HIP_TRY(hipMemcpy(q, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    index_bytes += src.size() * sizeof(T);
    *dst = q;
    return GMX_OK;
  }
  int upload(const uint32_t **dst, const gmx::WordBuf &src) {
    uint32_t *q = nullptr;
    int rc = alloc(&q, src.size(), false);
    if (rc) return rc;
    const size_t piece = (size_t)1 << 28;  // (pageable memory, tens of GB at whole-genome scale: 1 GB per staged copy)
    for (size_t at = 0; at < src.size(); at += piece)
      HIP_TRY(hipMemcpy(q + at, src.data() + at, std::min(piece, src.size() - at) * sizeof(uint32_t), hipMemcpyHostToDevice));
    index_bytes += src.size() * sizeof(uint32_t);
    *dst = q;
    return GMX_OK;
  }
};

// Host-side wait for an event that costs no core: query, nap, query. hipEventSynchronize — also on an event created with
// hipEventBlockingSync — kept the calling thread AND a thread of the runtime at 100 % of a core each on the GPU boxes
// (tools/exp/host_call_cost.py: 1500 back-to-back calls, 1.03 s of wall time, 1.03 s of CPU in each of the two threads), so a
// feeder that runs ahead of its GPU cost two cores: eight of them, sixteen — the whole container. The nap (50 us) is far
// below a batch (0.4-3 ms) and three batches are in flight per engine, so the GPU never waits for the host's wake-up.
// GMX_WAIT_SPIN=1: hipEventSynchronize as before (A/B runs).
static hipError_t gmx_event_wait(hipEvent_t ev) {
  static const bool spin = getenv("GMX_WAIT_SPIN") != nullptr;
  if (spin) return hipEventSynchronize(ev);
  // (the first 60 us by querying alone: an event about to complete — the end of a job, the last of several streams — is
  //  not paid for with a nap's wake-up latency; a feeder ahead of its GPU waits ~0.5 ms per batch and naps through it)
  const auto t0 = std::chrono::steady_clock::now();
  for (bool napping = false;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady) return q;
    (void)hipGetLastError();  // (hipErrorNotReady is sticky for hipGetLastError)
    if (!napping) {
      napping = std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(60);
      continue;
    }
    struct timespec ts = {0, 50 * 1000};
    nanosleep(&ts, nullptr);
  }
}

// Wait for the engine's own streams WITHOUT spinning: a blocking event per stream (the thread sleeps until the interrupt).
// hipStreamSynchronize / hipDeviceSynchronize poll — one core per waiting thread; a node's eight feeder threads, each ahead
// of its GPU, cost eight cores that way (profiles/round4/feed_x8.txt: 2.5 ns of host CPU per read, 27 cores' worth at
// 8 x 1.34 G reads/s). The callers still issue their hipDeviceSynchronize afterwards: it then returns at once.
static int gmx_quiesce(gmx_engine *e) {
  if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming | hipEventBlockingSync));
  hipStream_t streams[5] = {e->last_stream, e->copy_stream, e->side_stream, e->side2_stream, e->copy_stream2};
  for (int i = 0; i < 5; ++i) {
    if (i > 0 && !streams[i]) continue;  // ([0]: the null stream counts)
    bool seen = false;
    for (int j = 0; j < i; ++j) seen = seen || streams[j] == streams[i];
    if (seen) continue;
    HIP_TRY(hipEventRecord(e->ev_wait, streams[i]));
    HIP_TRY(gmx_event_wait(e->ev_wait));
  }
  return GMX_OK;
}

// the accumulators have been zeroed by work on `stream` (a queued reset): the twin's next batch must come behind it
static int note_zeroed(gmx_engine *e, hipStream_t stream) {
  if (!e->twin) return GMX_OK;
  if (!e->ev_zeroed) HIP_TRY(hipEventCreateWithFlags(&e->ev_zeroed, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(e->ev_zeroed, stream));
  ++e->zero_epoch;
  return GMX_OK;
}
// `stream` waits for the twin's last batch (its main stream joins its side streams at the end of every batch)
static int twin_join(gmx_engine *e, hipStream_t stream) {
  gmx_engine *tw = e->twin;
  if (!tw || !e->twin_in_flight) return GMX_OK;
  if (!e->ev_twin_done) HIP_TRY(hipEventCreateWithFlags(&e->ev_twin_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(e->ev_twin_done, tw->main_stream));
  HIP_TRY(hipStreamWaitEvent(stream, e->ev_twin_done, 0));
  return GMX_OK;
}
static int flush_reset(gmx_engine *e) {
  if (!e->reset_pending) return GMX_OK;
  e->reset_pending = false;
  HIP_TRY(hipMemsetAsync(e->d_fused, 0, (e->n_fused + 32) * 4, e->reset_stream));
  return note_zeroed(e, e->reset_stream);
}

// Grouped log -> host. Waits for the device, adds the log's records to e->log_counts when more than `keep_below` words are in
// use (and empties the device log), and notes how full it is. Records: [site_index, n_ids, ids...], each worth +1;
// GMX_LOG_PAD words are padding (CoverLogPart::log_reserve).
static int log_settle(gmx_engine *e);
static int gmx_log_drain(gmx_engine *e, uint64_t keep_below) {
  int frc = flush_reset(e);
  if (frc) return frc;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t used = 0;
  HIP_TRY(hipMemcpy(&used, e->d_log_cursor, 4, hipMemcpyDeviceToHost));
  used = std::min(used, e->log_cap);
  e->log_reads_since = 0;
  e->log_known = used;
  if (used <= keep_below) return GMX_OK;
  std::vector<uint32_t> w(used);
  HIP_TRY(hipMemcpy(w.data(), e->d_log, (size_t)used * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> key;
  for (size_t i = 0; i < w.size();) {
    if (w[i] == GMX_LOG_PAD) {
      ++i;
      continue;
    }
    if (i + 2 > w.size() || i + 2 + (size_t)w[i + 1] > w.size()) {  // reservations are exact (log_reserve): never expected
      gmx_set_error("grouped-allele-count log: malformed record at word " + std::to_string(i) + " of " + std::to_string(w.size()));
      return GMX_EREF;
    }
    key.assign(1, w[i]);
    key.insert(key.end(), w.begin() + i + 2, w.begin() + i + 2 + w[i + 1]);
    e->log_counts[key] += 1;
    i += 2 + w[i + 1];
  }
  HIP_TRY(hipMemset(e->d_log_cursor, 0, 4));
  e->log_known = 0;
  return GMX_OK;
}

// A coverage instance with its scratch in LDS: one wave per block, as many blocks per CU as scratch copies fit its LDS.
template <class Env, int LIST>
static void launch_cover_lds(gmx_engine *e, hipStream_t stream, const BatchView &b, const SearchOut &o, const CoverAcc &acc,
                             bool after_coop = false) {
  static const uint32_t lanes_env = getenv("GMX_COVER_LANES") ? (uint32_t)atoi(getenv("GMX_COVER_LANES")) : 0u;
  const uint32_t lanes = lanes_env ? std::min(lanes_env, gmx_cover_lds_lanes<Env>()) : gmx_cover_lds_lanes<Env>();
  const size_t lds = (size_t)GmxScratchFixed<Env>::total * lanes * sizeof(uint32_t);
  const uint32_t per_cu = std::min<uint32_t>((uint32_t)(160 * 1024 / lds), 32u);
  hipLaunchKernelGGL((gmx_cover_kernel<Env, LIST>), dim3(e->n_cus * per_cu), dim3(64), lds, stream, e->dview, b, o, e->big,
                     acc, lanes, after_coop ? 1u : 0u);
}

template <int LIST>
static void launch_cover_coop(gmx_engine *e, hipStream_t stream, const BatchView &b, const SearchOut &o, const CoverAcc &acc) {
  const size_t lds = (size_t)gmx_coop_lds_words<LIST>() * sizeof(uint32_t);
  const uint32_t per_cu = std::min<uint32_t>((uint32_t)(160 * 1024 / lds), 16u);
  hipLaunchKernelGGL((gmx_cover_coop_kernel<LIST>), dim3(e->n_cus * per_cu), dim3(64), lds, stream, e->dview, b, o, e->big, acc);
}

extern "C" {

void gmx_engine_default_opts(gmx_engine_opts *o) try {
  o->device = 0;
  o->rng_mode = GMX_RNG_LEMIRE;
  o->max_states = 1024;
  o->max_path_nodes = 2048;
  o->max_batch_reads = 4u << 20;
  o->forward_only = 0;
  o->huge_heap_bytes = 512ull << 20;
  o->log_cap_words = 0;
} GMX_GUARD_VOID("gmx_engine_default_opts")

static int ensure_batch_capacity(gmx_engine *e, uint64_t n_reads) {
  if (n_reads <= e->cap_reads) return GMX_OK;
  // (re)allocate: old buffers stay in `allocs` until destroy; growth is rare (first call sizes it)
  // (an eighth of headroom when the workspace GROWS: chunks of a decoded reads file differ by a few reads — the record a chunk's end
  // cuts — and every growth is two dozen allocations that wait for the device: 11 ms in the middle of `gram`'s second chunk)
  uint64_t cap = std::max<uint64_t>(e->cap_reads ? n_reads + n_reads / 8 : n_reads, 1024);
  uint64_t n_tasks = cap * 2;
  int rc;
  if ((rc = e->alloc(&e->d_skip, cap, true))) return rc;
  if ((rc = e->alloc(&e->d_status, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_n_final, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_cover_recs, n_tasks * GMX_REGIONS, false))) return rc;
  if ((rc = e->alloc(&e->d_mapped, n_tasks * GMX_REGIONS, false))) return rc;
  if ((rc = e->alloc(&e->d_task_lists, (size_t)GMX_TL_N * n_tasks, false))) return rc;
  e->d_overflow = e->d_task_lists + (size_t)GMX_TL_OVERFLOW * n_tasks;
  e->d_overflow2 = e->d_task_lists + (size_t)GMX_TL_OVERFLOW2 * n_tasks;
  e->d_alive = e->d_task_lists + (size_t)GMX_TL_ALIVE * n_tasks;
  e->d_dead = e->d_task_lists + (size_t)GMX_TL_DEAD * n_tasks;
  e->d_dead2 = e->d_task_lists + (size_t)GMX_TL_DEAD2 * n_tasks;
  e->d_cover_general = e->d_task_lists + (size_t)GMX_TL_GENERAL * n_tasks;
  if ((rc = e->alloc(&e->d_cover_overflow, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_general_rest, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_single_rest, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_park2, (size_t)n_tasks * GMX_STACK_DEPTH, false))) return rc;
  if ((rc = e->alloc(&e->d_park2_n, n_tasks, false))) return rc;
  if (e->log_sites)
    for (int side = 0; side < 2; ++side) {
      if ((rc = e->alloc(&e->d_log_retry[side], n_tasks, false))) return rc;
      if ((rc = e->alloc(&e->d_log_retry_recs[side], n_tasks, false))) return rc;
      if ((rc = e->alloc(&e->d_log_retry_huge[side], n_tasks, false))) return rc;
    }
  if ((rc = e->alloc(&e->d_cover_mid, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_seed_cursor, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_finals, n_tasks * GMX_FAST_STATES, false))) return rc;
  if ((rc = e->alloc(&e->d_arena, n_tasks * GMX_FAST_ARENA, false))) return rc;
  if ((rc = e->alloc(&e->d_alive_seed, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_huge, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_cover_huge, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_huge_retry, 2 * n_tasks, false))) return rc;
  // large-capacity pass: one slot (~60 KB of pools at the default capacities) per task it may have to take; a 1 M-read
  // batch with 5 % of the genome in 10-copy repeats sends 59 k of its 2 M tasks there
  e->big.max_slots = (uint32_t)std::min<uint64_t>(n_tasks, std::min<uint64_t>(std::max<uint64_t>(n_tasks / 16, 4096), 262144));
  if ((rc = e->alloc(&e->big.states, (size_t)e->big.max_slots * e->big.max_states, false))) return rc;
  if ((rc = e->alloc(&e->big.stack, (size_t)e->big.max_slots * e->big.max_states * GMX_STACK_WORDS, false))) return rc;
  if ((rc = e->alloc(&e->big.arena, (size_t)e->big.max_slots * e->big.max_path_nodes, false))) return rc;
  if ((rc = e->alloc(&e->big.n_final, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->big.task_of_slot, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_big_mapped, e->big.max_slots, false))) return rc;
  e->inst_cap = (uint32_t)std::min<uint64_t>(n_tasks, 1u << 23);  // instance lanes of reads in short repeats (320 B of pools each)
  if ((rc = e->alloc(&e->d_inst_list, e->inst_cap, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_sa, e->inst_cap, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_remaining, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_mapped, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_first, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_width, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_serial, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_general_serial, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_big_serial, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_overflow3, 2 * n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_arena, (size_t)e->inst_cap * GMX_FAST_ARENA, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_states, (size_t)e->inst_cap * GMX_INST_STATES, false))) return rc;
  e->cap_reads = cap;
  return GMX_OK;
}

// The device copy of an index is shared by the engines made of it on one device (round 5): several engines per GPU keep batches
// in flight side by side — a nested PRG's batch is a 0.4 ms burst and then 2 ms of a few straggler tasks on a handful of CUs —
// and the second one must not cost a second upload and a second copy in HBM. Reference counted; GMX_NO_INDEX_SHARE=1: off.
struct GmxDeviceIndex {
  uint64_t serial = 0;  // gmx_index_serial of the index it was uploaded from
  int device = 0;
  GmxIndexView view{};
  std::vector<void *> allocs;
  uint64_t bytes = 0;
  int refs = 0;
};
static std::mutex g_dev_index_mu;
static std::vector<GmxDeviceIndex *> g_dev_indexes;
static void gmx_dev_index_release(GmxDeviceIndex *d) {
  if (!d) return;
  std::lock_guard<std::mutex> lk(g_dev_index_mu);
  if (--d->refs > 0) return;
  for (void *p : d->allocs) (void)hipFree(p);
  g_dev_indexes.erase(std::remove(g_dev_indexes.begin(), g_dev_indexes.end(), d), g_dev_indexes.end());
  delete d;
}

static int engine_create(const gmx_index *ixh, const gmx_engine_opts *opts_in, gmx_engine *primary, gmx_engine **out);
int gmx_engine_create(const gmx_index *ixh, const gmx_engine_opts *opts_in, gmx_engine **out) try {
  int rc = engine_create(ixh, opts_in, nullptr, out);
  if (rc) return rc;
  gmx_engine *e = *out;
  // the twin (a second batch in flight; gmx_engine::twin): nested PRGs and indexes of 2 GB and more
  const gmx::HostIndex &h = gmx_index_host(ixh);
  const bool can = !e->log_sites && e->shared_index && !getenv("GMX_NO_INDEX_SHARE");  // (the twin reads the engine's copy of the index, never one of its own)
  bool want = can && (h.is_nested || e->index_bytes >= (2ull << 30));
  if (const char *tw = getenv("GMX_TWIN")) want = can && atoi(tw) != 0;
  if (want) {
    gmx_engine *t = nullptr;
    if (engine_create(ixh, &e->opts, e, &t) == GMX_OK) {
      e->twin = t;
    } else {
      (void)hipGetLastError();  // (no room for a second workspace: one batch at a time, as before)
    }
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_create")

static int engine_create(const gmx_index *ixh, const gmx_engine_opts *opts_in, gmx_engine *primary, gmx_engine **out) {
  if (!ixh || !out) {
    gmx_set_error("gmx_engine_create: null argument");
    return GMX_EINVAL;
  }
  gmx_engine_opts opts;
  if (opts_in)
    opts = *opts_in;
  else
    gmx_engine_default_opts(&opts);
  if (opts.max_states == 0) opts.max_states = 1024;
  if (opts.max_path_nodes == 0) opts.max_path_nodes = 2048;
  if (opts.max_batch_reads == 0) opts.max_batch_reads = 4u << 20;
  if (opts.huge_heap_bytes == 0) opts.huge_heap_bytes = 512ull << 20;
  if (const char *hb = getenv("GMX_HUGE_HEAP_BYTES")) opts.huge_heap_bytes = strtoull(hb, nullptr, 10);
  opts.huge_heap_bytes = std::max<uint64_t>(opts.huge_heap_bytes, 64 * 1024);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    gmx_set_error("no HIP device available: the quasimap engine has no CPU fallback");
    return GMX_ENODEV;
  }
  if (opts.device < 0 || opts.device >= ndev) {
    gmx_set_error("device ordinal out of range");
    return GMX_ENODEV;
  }
  HIP_TRY(hipSetDevice(opts.device));
  const gmx::HostIndex &h = gmx_index_host(ixh);
  if (h.kmer_size == 0) {
    gmx_set_error("the index was built without a k-mer seed table (kmer_size = 0)");
    return GMX_EINVAL;
  }
  gmx_engine *e = new gmx_engine();
  e->opts = opts;
  GmxIndexView v = h.view();
  int rc = 0;
  {
    std::lock_guard<std::mutex> share_lock(g_dev_index_mu);  // (engines of one group are created side by side: the second waits for the first one's upload)
    GmxDeviceIndex *found = nullptr;
    if (!getenv("GMX_NO_INDEX_SHARE"))
      for (GmxDeviceIndex *d : g_dev_indexes)
        if (d->serial == gmx_index_serial(ixh) && d->device == opts.device) found = d;
    if (found) {
      ++found->refs;
      v = found->view;
      e->index_bytes = found->bytes;
      e->shared_index = found;
    } else {
      rc |= e->upload(&v.blocks, h.blocks);
      rc |= e->upload(&v.hits, h.hits);
      rc |= e->upload(&v.hit_perm, h.hit_perm);
      rc |= e->upload(&v.hit_prog, h.hit_prog);
      rc |= e->upload(&v.text, h.text);
      rc |= e->upload(&v.prog, h.prog);
      rc |= e->upload(&v.sa, h.sa);
      rc |= e->upload(&v.pos_node, h.pos_node);
      rc |= e->upload(&v.nodes, h.nodes);
      rc |= e->upload(&v.edges, h.edges);
      rc |= e->upload(&v.sites, h.sites);
      rc |= e->upload(&v.site_geo, h.site_geo);
      rc |= e->upload(&v.seeds, h.seeds);
      if (h.kmer_size2) rc |= e->upload(&v.seeds2, h.seeds2);
      else v.seeds2 = nullptr;
      rc |= e->upload(&v.seed_words, h.seed_words);
      if (!rc) {  // flags in the multi-state entries of the device copies (GMX_SEEDF_*)
        if (((uint64_t)h.seed_words.size() >> h.seed_shift) >= (1u << 30)) {
          gmx_set_error("the seed tables hold more than 2^30 units of multi-state entries");
          rc = GMX_ECAP;
        } else {
          hipLaunchKernelGGL(gmx_seed_mark_kernel, dim3(4096), dim3(256), 0, nullptr, const_cast<GmxSeed *>(v.seeds), (uint64_t)h.seeds.size(), const_cast<uint32_t *>(v.seed_words), v.seed_shift, v.sa, v.text);
          if (h.kmer_size2)
            hipLaunchKernelGGL(gmx_seed_mark_kernel, dim3(4096), dim3(256), 0, nullptr, const_cast<GmxSeed *>(v.seeds2), (uint64_t)h.seeds2.size(), const_cast<uint32_t *>(v.seed_words), v.seed_shift, v.sa, v.text);
          rc |= hipDeviceSynchronize() != hipSuccess;
        }
      }
      rc |= e->upload(&v.kmer_bitmap, h.kmer_bitmap);

      if (!rc) {  // everything allocated so far is the index: it moves to the shared object
        GmxDeviceIndex *d = new GmxDeviceIndex();
        d->serial = gmx_index_serial(ixh);
        d->device = opts.device;
        d->view = v;
        d->allocs = std::move(e->allocs);
        e->allocs.clear();
        d->bytes = e->index_bytes;
        d->refs = 1;
        g_dev_indexes.push_back(d);
        e->shared_index = d;
      }
    }
  }
  e->dview = v;
  e->n_allele = h.n_allele_slots;
  e->n_pb = h.n_pb_slots;
  e->n_grouped = h.n_grouped_slots;
  {  // one contiguous block: a single all-reduce covers the whole coverage (gmx_coverage_device)
    e->n_acc = ((size_t)h.n_acc_slots + 63) / 64 * 64;
    e->n_fused = e->n_acc + 32;
    if (primary)
      e->d_fused = primary->d_fused;  // (the twin records into the engine's accumulators)
    else
      rc |= e->alloc(&e->d_fused, e->n_fused + 32, true);  // + 16 words of read counters + log cursor
    e->d_limbs = e->d_fused ? e->d_fused + e->n_acc : nullptr;
    e->d_stats = e->d_fused ? reinterpret_cast<unsigned long long *>(e->d_fused + e->n_fused) : nullptr;
    e->d_log_cursor = e->d_fused ? e->d_fused + e->n_fused + 16 : nullptr;
    rc |= e->alloc(&e->d_error, 2, true);
    e->phys_allele = h.phys_allele;
    e->phys_pb = h.phys_pb;
    e->phys_grouped = h.phys_grouped;
    e->hit_fix = h.hit_fix;
  }
  // The grouped log is used only by sites with more alleles than get dense group counters (gmx_index.cpp: 8). Between
  // batches the engine looks at its real fill (log_settle): drained when half full; entries that found it full are redone.
  for (const GmxSite &st : h.sites) e->log_sites = e->log_sites || st.grouped_off == GMX_GROUPED_LOG;
  {  // the lean single-instance coverage kernel where most sites have geometry records (GMX_NO_COVER_JUMP: A/B runs)
    uint64_t n_jump = 0;
    for (const GmxSiteGeo &g : h.site_geo) n_jump += (g.flags & GMX_SITE_JUMP) ? 1u : 0u;
    e->cover_jump = !h.is_nested && 2 * n_jump > h.site_geo.size() && !getenv("GMX_NO_COVER_JUMP");
    // GMX_NO_COVER_JUMP=1 (the escape hatch, and the walk side of tests/test_cover_jump_ab.py): no kernel
    // sees the geometry records, every single-instance read is recorded by the walk as the reference walks it
    if (getenv("GMX_NO_COVER_JUMP")) e->dview.site_geo = nullptr;
  }
  {
    uint64_t cap = opts.log_cap_words ? opts.log_cap_words
                   : e->log_sites     ? (1ull << 26)  // 256 MB; a batch that fills it is settled by drain + replay (log_settle)
                                      : 64;
    e->log_cap = (uint32_t)std::min<uint64_t>(cap, 0xFFFFFF00ull);
  }
  if (primary)
    e->d_log = primary->d_log;  // (never written: an index with log sites has no twin)
  else
    rc |= e->alloc(&e->d_log, e->log_cap, false);
  e->heap_words = opts.huge_heap_bytes / 4 / 64 * 64;
  rc |= e->alloc(&e->d_heap, e->heap_words, false);
  rc |= e->alloc(&e->d_counters, GMX_N_COUNTERS * GMX_CNT_STRIDE, true);
  // large-capacity pass
  e->big.max_states = opts.max_states;
  e->big.max_path_nodes = opts.max_path_nodes;
  e->big.max_slots = 0;  // its pools are sized with the batch (ensure_batch_capacity)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, opts.device) == hipSuccess && prop.multiProcessorCount > 0)
      e->n_cus = (uint32_t)prop.multiProcessorCount;
    const size_t words = h.kmer_bitmap.size();
    if (!getenv("GMX_FORCE_ABSENT_FILTER") && words >= 4 && words % 4 == 0 && words * 4 <= 128 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(gmx_filter_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(words * 4)) == hipSuccess)
      e->filter_lds_words = (uint32_t)words;
    (void)hipGetLastError();
    if (primary) {  // (the k-mer filter's tables: the engine's)
      e->filter_lds_words = primary->filter_lds_words;
      e->d_kmer_planar = primary->d_kmer_planar;
      e->d_absent = primary->d_absent;
      e->n_absent = primary->n_absent;
      e->use_absent = primary->use_absent;
    } else if (e->filter_lds_words) {  // re-index the presence bitmap: table index (base j from the left in bit pair j) -> planar
      const uint32_t k = h.kmer_size;
      std::vector<uint32_t> planar(words, 0);
      for (uint64_t code = 0; code < (1ull << (2 * k)); ++code) {
        if (!((h.kmer_bitmap[code >> 5] >> (code & 31)) & 1u)) continue;
        uint32_t lo = 0, hi = 0;
        for (uint32_t j = 0; j < k; ++j) {
          const uint32_t base = (uint32_t)(code >> (2 * j)) & 3u;
          lo |= (base & 1u) << j;
          hi |= (base >> 1) << j;
        }
        const uint32_t p = (hi << k) | lo;
        planar[p >> 5] |= 1u << (p & 31);
      }
      rc |= e->upload(&e->d_kmer_planar, planar);
    } else if (!getenv("GMX_NO_ABSENT_FILTER")) {  // a bitmap too large for LDS: few absent k-mers? (whole-genome PRGs)
      const uint64_t n_k = 1ull << (2 * h.kmer_size);
      if (n_k - std::min<uint64_t>(n_k, h.n_seed_kmers_present) <= GMX_ABSENT_MAX) {
        std::vector<uint32_t> absent;
        for (size_t w = 0; w < words && absent.size() <= GMX_ABSENT_MAX; ++w) {
          uint32_t zeros = ~h.kmer_bitmap[w];
          while (zeros) {
            const uint64_t code = (uint64_t)w * 32 + (uint32_t)__builtin_ctz(zeros);
            zeros &= zeros - 1;
            if (code < n_k) absent.push_back((uint32_t)code);
          }
        }
        if (absent.size() <= GMX_ABSENT_MAX) {
          e->n_absent = (uint32_t)absent.size();
          e->use_absent = true;
          if (absent.empty()) absent.push_back(0);
          rc |= e->upload(&e->d_absent, absent);
        }
      }
    }
  }
  if (const char *pi = getenv("GMX_PROBE_ITERS")) e->probe_iters = (uint32_t)std::max(0, atoi(pi));
  if (getenv("GMX_NO_FUSE")) e->fuse = 0;
  if (const char *eb = getenv("GMX_EXTEND_BUDGET")) e->extend_budget = (uint32_t)std::max(0, atoi(eb));
  e->extend_cap = 0u;  // (GMX_EXTEND_CAP: off by default — at configs[2] a cap of 40 iterations sent 45 k tasks per batch to the
                       //  large-capacity route and the step took 6.4 ms instead of 2.5; see profiles/round4/config2_cap_sweep.txt)
  if (const char *ec = getenv("GMX_EXTEND_CAP")) e->extend_cap = (uint32_t)std::max(0, atoi(ec));
  // passes over the stragglers and the iteration budgets of all but the last. ONE pass by default: at configs[2] (nested
  // MSA regions) three passes — budgets 24 and 96 — take 230 + 528 + 494 us where the single pass takes 901: what is left
  // after the first budget is a few tasks with hundreds of general iterations each (~5 us per iteration: dependent fetches
  // of jump programs and path nodes), and packing them into full waves again does not shorten any of them.
  // GMX_EXTEND_PASSES = "b0,b1": three passes, budgets b0 and b1 (experiments).
  e->extend_passes = 1u;
  e->extend_budget2[0] = 24;
  e->extend_budget2[1] = 96;
  if (const char *ep = getenv("GMX_EXTEND_PASSES")) {
    e->extend_passes = 1;
    for (const char *q = ep; *q && e->extend_passes < GMX_EXTRA_PASSES;) {
      e->extend_budget2[e->extend_passes - 1] = (uint32_t)std::max(1l, strtol(q, const_cast<char **>(&q), 10));
      ++e->extend_passes;
      if (*q == ',') ++q; else break;
    }
  }
  if (getenv("GMX_NO_COOP")) e->coop = false;
  // k-mer entries with many states (small k on a large or dense PRG) do not fit the per-lane stack: when they carry
  // more than 10 % of the seed states the kernels take them one state at a time (seed cursor, a few % slower), else
  // the rare large entry goes to the large-capacity pass
  e->seed_cursor = h.n_seed_states_large * 10 > h.n_seed_states;
  if (const char *sc = getenv("GMX_SEED_CURSOR")) e->seed_cursor = atoi(sc) != 0;
  if (primary) {  // (the twin reads the engine's own copies)
    e->seed_cursor = primary->seed_cursor;
    e->dview.sa_ctx = primary->dview.sa_ctx;
    e->dview.seed_side = primary->dview.seed_side;
  }
  if (!primary && e->seed_cursor && !rc && !getenv("GMX_NO_SA_CTX")) {  // left-context word per suffix-array position (GmxIndexView::sa_ctx): + 4 B per symbol
    uint32_t *sc = nullptr;
    if (e->alloc(&sc, h.sa.size(), false) == GMX_OK) {
      hipLaunchKernelGGL(gmx_sa_ctx_kernel, dim3(8192), dim3(256), 0, nullptr, e->dview.sa, e->dview.text, (uint64_t)h.sa.size(), sc);
      if (hipDeviceSynchronize() == hipSuccess) {
        e->dview.sa_ctx = sc;
        e->index_bytes += h.sa.size() * sizeof(uint32_t);
      }
    } else {
      (void)hipGetLastError();  // (no room: the occurrences are screened through the suffix array and the text, as before)
    }
  }
  // ... and the screening side table of the multi-state entries (GmxIndexView::seed_side; gmx_seed_side_kernel): + a word per
  // four seed words (20 GB at configs[4]). GMX_NO_SEED_SIDE=1: entries are walked header by header as until round 5 (A/B runs).
  if (!primary && e->seed_cursor && !rc && !getenv("GMX_NO_SEED_SIDE") && h.seed_words.size() > 1) {
    uint32_t *side = nullptr;
    const size_t n_side = h.seed_words.size() / 4 + 2;
    if (e->alloc(&side, n_side, false) == GMX_OK) {
      hipLaunchKernelGGL(gmx_seed_side_kernel, dim3(8192), dim3(256), 0, nullptr, e->dview.seeds, (uint64_t)h.seeds.size(), e->dview.seed_words,
                         e->dview.seed_shift, side);
      if (h.kmer_size2)
        hipLaunchKernelGGL(gmx_seed_side_kernel, dim3(8192), dim3(256), 0, nullptr, e->dview.seeds2, (uint64_t)h.seeds2.size(), e->dview.seed_words,
                           e->dview.seed_shift, side);
      if (hipDeviceSynchronize() == hipSuccess) {
        e->dview.seed_side = side;
        e->index_bytes += n_side * sizeof(uint32_t);
      }
    } else {
      (void)hipGetLastError();  // (no room: the screen walks the entries themselves)
    }
  }
  // (stream priorities for the side streams — the few-task kernels first — were measured in round 4: no difference, the
  //  chains there wait for memory, not for wave slots)
  // On a NESTED PRG the twin's streams are created at the HIGHEST stream priority: the runtime keeps a pool of hardware queues per
  // priority (4 each by default), so they get queues of their own. At one priority the engine's and the twin's seven streams share
  // four queues and a twin's main chain sits in a queue behind the engine's straggler kernels: configs[2]'s packed feed 372 M
  // reads/s with one workspace, 436 M with two at one priority, 472 M with the twin's streams on queues of their own
  // (profiles/round6/twin_ab.txt). On a flat PRG every kernel of a batch fills the GPU and precedence for one workspace only
  // delays the other: configs[3] 856 M -> 892 M at one priority, 833 M with the pool. GMX_TWIN_PRIORITY=0 / 1 forces.
  auto make_stream = [&](hipStream_t *st) -> bool {
    int least = 0, greatest = 0;
    bool want_prio = h.is_nested;
    if (const char *tp = getenv("GMX_TWIN_PRIORITY")) want_prio = atoi(tp) != 0;
    const bool prio = primary && want_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
    (void)hipGetLastError();
    return (prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
  };
  rc |= !make_stream(&e->side_stream);
  rc |= !make_stream(&e->side2_stream);
  rc |= hipEventCreateWithFlags(&e->ev_fork2, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_side1, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_filter, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess;
  e->cover_big_lanes = 64 * 32;
  rc |= e->alloc(&e->d_scratch_big, (size_t)GmxScratchFixed<CoverEnvBig>::total * e->cover_big_lanes, false);
  if (primary) {
    e->is_twin = true;
    rc |= !make_stream(&e->main_stream);
    e->last_stream = e->main_stream;
  }
  if (rc) {
    gmx_engine_destroy(e);
    return GMX_EHIP;
  }
  *out = e;
  return GMX_OK;
}

void gmx_engine_destroy(gmx_engine *e) try {
  if (!e) return;
  (void)hipSetDevice(e->opts.device);
  (void)hipDeviceSynchronize();
  if (e->twin) gmx_engine_destroy(e->twin);
  e->twin = nullptr;
  if (e->main_stream) (void)hipStreamDestroy(e->main_stream);
  if (e->ev_zeroed) (void)hipEventDestroy(e->ev_zeroed);
  if (e->ev_twin_done) (void)hipEventDestroy(e->ev_twin_done);
  if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
  if (e->side2_stream) (void)hipStreamDestroy(e->side2_stream);
  if (e->ev_fork2) (void)hipEventDestroy(e->ev_fork2);
  if (e->ev_side1) (void)hipEventDestroy(e->ev_side1);
  if (e->ev_filter) (void)hipEventDestroy(e->ev_filter);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_wait) (void)hipEventDestroy(e->ev_wait);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->copy_stream2) (void)hipStreamDestroy(e->copy_stream2);
  if (e->h_log_state) (void)hipHostFree(e->h_log_state);
  if (e->ev_log_state) (void)hipEventDestroy(e->ev_log_state);
  for (auto &sl : e->pslot) {
    if (sl.copied) (void)hipEventDestroy(sl.copied);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  for (auto &sl : e->stage) {
    if (sl.copied) (void)hipEventDestroy(sl.copied);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
    if (sl.h_seeds) (void)hipHostFree(sl.h_seeds);
  }
  for (void *p : e->allocs) (void)hipFree(p);
  gmx_dev_index_release(e->shared_index);
  delete e;
} GMX_GUARD_VOID("gmx_engine_destroy")

int gmx_engine_reset(gmx_engine *e) try {
  HIP_TRY(hipSetDevice(e->opts.device));
  e->reset_pending = false;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(e->d_fused, 0, (e->n_fused + 32) * 4));
  HIP_TRY(hipMemset(e->d_error, 0, 8));
  HIP_TRY(hipMemset(e->d_counters, 0, GMX_N_COUNTERS * GMX_CNT_STRIDE * 4));
  if (e->twin) {
    HIP_TRY(hipMemset(e->twin->d_error, 0, 8));
    HIP_TRY(hipMemset(e->twin->d_counters, 0, GMX_N_COUNTERS * GMX_CNT_STRIDE * 4));
    e->twin_in_flight = false;
  }
  e->log_counts.clear();
  e->log_known = e->log_reads_since = 0;
  e->log_state_pending = false;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_reset")

int gmx_engine_reset_async(gmx_engine *e, void *hip_stream) try {
  HIP_TRY(hipSetDevice(e->opts.device));
  hipStream_t st = (hipStream_t)hip_stream;
  int frc = flush_reset(e);  // (an earlier one still pending, on whatever stream it named)
  if (frc) return frc;
  if ((frc = twin_join(e, st))) return frc;  // (the zeroing comes behind whatever the twin still records)
  e->twin_in_flight = false;
  e->reset_pending = true;
  e->reset_stream = st;
  e->log_counts.clear();  // what earlier batches left in the device log goes with the cursor
  e->log_known = e->log_reads_since = 0;
  e->log_state_pending = false;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_reset_async")

static bool gmx_ab_no_filter() {  // A/B runs ONLY (tools/kab.py): no k-mer filter, the two counters it decides stay zero
  static const bool off = getenv("GMX_AB_NO_FILTER") != nullptr;
  return off;
}
static void launch_filter(gmx_engine *e, hipStream_t st, dim3 task_grid, const BatchView &b, const SearchOut &o, int pass,
                          hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
  if (gmx_ab_no_filter()) return;
  if (e->filter_lds_words)
    hipExtLaunchKernelGGL(gmx_filter_lds_kernel, dim3(e->n_cus), dim3(GMX_FILTER_LDS_THREADS), e->filter_lds_words * 4,
                          st, t0, t1, 0u, e->dview, b, o, e->d_kmer_planar, e->filter_lds_words, pass);
  else if (e->use_absent)
    hipExtLaunchKernelGGL(gmx_filter_absent_kernel, task_grid, dim3(GMX_BLOCK), 0, st, t0, t1, 0u, e->dview, b, o, e->d_absent, e->n_absent, pass);
  else
    hipExtLaunchKernelGGL(gmx_filter_kernel, task_grid, dim3(GMX_BLOCK), 0, st, t0, t1, 0u, e->dview, b, o, pass);
}

// ---- grouped log: exact accounting between batches (engines whose index has sites with more than 8 alleles) ----------
static int log_state_enqueue(gmx_engine *e, hipStream_t stream) {
  if (!e->h_log_state) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_log_state), 4 * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_log_state, hipEventDisableTiming));
  }
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 0, e->d_log_cursor, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 1, e->d_counters + GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 2, e->d_counters + GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 3, e->d_counters + GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(e->ev_log_state, stream));
  e->log_state_pending = true;
  return GMX_OK;
}

// The entries of the last batch that found the log full, again: setup (list lengths where the kernels read them), the
// compact records, then the large-scratch coverage instance, whose last block also serves the last tier.
static int launch_log_replay(gmx_engine *e, hipStream_t stream) {
  const int in = e->log_retry_side, out = in ^ 1;
  SearchOut o = e->last_o;
  o.log_retry_list = e->d_log_retry[out];
  o.log_retry_recs = e->d_log_retry_recs[out];
  o.log_retry_huge = e->d_log_retry_huge[out];
  o.cover_overflow_list = e->d_log_retry[in];  // the queue of gmx_cover_kernel<CoverEnvBig, 1>: this round's entries
  hipLaunchKernelGGL(gmx_log_replay_setup_kernel, dim3(1), dim3(1024), 0, stream, o, e->d_log_retry_huge[in]);
  if (e->dview.is_nested)
    hipLaunchKernelGGL(gmx_cover_single_replay_kernel<true>, dim3(e->n_cus * 4), dim3(GMX_BLOCK), 0, stream, e->dview, e->last_b, o, e->last_acc,
                     e->d_log_retry_recs[in]);
  else
    hipLaunchKernelGGL(gmx_cover_single_replay_kernel<false>, dim3(e->n_cus * 4), dim3(GMX_BLOCK), 0, stream, e->dview, e->last_b, o, e->last_acc,
                     e->d_log_retry_recs[in]);
  hipLaunchKernelGGL((gmx_cover_kernel<CoverEnvBig, 1>), dim3(e->cover_big_lanes / 64), dim3(64), e->last_big_lds, stream, e->dview,
                     e->last_b, o, e->big, e->last_acc, 64u, 0u);
  HIP_TRY(hipGetLastError());
  e->log_retry_side = out;
  e->last_o.log_retry_list = o.log_retry_list;
  e->last_o.log_retry_recs = o.log_retry_recs;
  e->last_o.log_retry_huge = o.log_retry_huge;
  return log_state_enqueue(e, stream);
}

static int log_settle(gmx_engine *e) {
  if (!e->log_state_pending) return GMX_OK;
  uint64_t before = ~0ull;
  for (int round = 0;; ++round) {
    HIP_TRY(hipEventSynchronize(e->ev_log_state));
    e->log_state_pending = false;
    const uint32_t used = std::min(e->h_log_state[0], e->log_cap);
    const uint64_t retries = (uint64_t)e->h_log_state[1] + e->h_log_state[2] + e->h_log_state[3];
    if (retries == 0) {
      if (used > e->log_cap / 2) return gmx_log_drain(e, 0);
      e->log_known = used;
      return GMX_OK;
    }
    if (round > 0 && retries >= before) {  // (every round starts with an empty log: each must get at least one entry through)
      // an emptied log did not hold one task's records: only more memory helps (gmx_engine_sync reports the read)
      HIP_TRY(hipDeviceSynchronize());
      const uint32_t err[2] = {GMX_TASK_LOGFULL, 0};
      HIP_TRY(hipMemcpy(e->d_error, err, 8, hipMemcpyHostToDevice));
      return GMX_OK;
    }
    before = retries;
    int rc = gmx_log_drain(e, 0);
    if (rc) return rc;
    e->log_replays++;
    e->log_replayed_entries += retries;
    if ((rc = launch_log_replay(e, e->last_stream))) return rc;
  }
}

// One batch as the kernels see it: reads as bytes (d_reads + d_offsets: gmx_pack_kernel makes the bit planes) or as bit
// planes already (d_planes; gmx_map_reads_packed_host).
struct BatchInput {
  const uint8_t *d_reads = nullptr;
  const uint64_t *d_offsets = nullptr;  // null with uniform_len
  const uint32_t *d_seeds = nullptr;
  const uint2 *d_planes = nullptr;      // non-null: packed input, no pack kernel
  const uint8_t *d_skip = nullptr;      // packed input: per-read skip flags, or null
  const uint32_t *d_twobit = nullptr;   // non-null: the reads as a 2-bit stream (gmx_map_reads_2bit_host); unpacked into d_packed
  uint32_t twobit_base0 = 0;            // ... whose first base sits at this base index of d_twobit (< 32)
  uint32_t uniform_len = 0;
  uint64_t n_reads = 0, total_bases = 0;
};

// first kernel of a batch whose reads arrive packed: what gmx_pack_kernel does besides packing (queue counters, a queued reset)
__global__ void gmx_batch_begin_kernel(uint32_t *counters, uint32_t *zero, uint32_t zero_words) {
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < GMX_N_COUNTERS * GMX_CNT_STRIDE; i += blockDim.x) counters[i] = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x) zero[i] = 0;
}

// Reads that arrive as a 2-bit stream (include/gmx.h, gmx_pack_reads_2bit: base j of the batch in bits 2j, 2j + 1) -> the bit
// planes the kernels read, in gmx_pack_kernel's layout. One thread per pair of planes (32 bases): three words of the
// stream, funnel-shifted to the pair's first base, even bits -> low plane, odd bits -> high plane.
__device__ __forceinline__ uint32_t gmx_even_bits(unsigned long long x) {  // bits 0, 2, 4, .. 62 of x, compacted
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (uint32_t)x;
}
__global__ void __launch_bounds__(256) gmx_unpack2_kernel(BatchView b, const uint32_t *stream, uint32_t base0, uint2 *packed) {
  const uint32_t ppr_uniform = b.pairs_per_read;
  for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;; t += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t read, pair;
    uint64_t first_base;  // of the read, in the batch's stream
    uint32_t len;
    if (b.uniform_len) {
      read = (uint32_t)(t / ppr_uniform);
      if (read >= b.n_reads) break;
      pair = (uint32_t)(t - (uint64_t)read * ppr_uniform);
      first_base = (uint64_t)read * b.uniform_len;
      len = b.uniform_len;
    } else {  // ragged: one thread per read walks its pairs (the plane layout needs the offsets anyway)
      read = (uint32_t)t;
      if (read >= b.n_reads) break;
      pair = 0;
      first_base = b.offsets[read] - b.offsets[0];
      len = (uint32_t)(b.offsets[read + 1] - b.offsets[read]);
    }
    uint2 *out = packed + pack_off(b, read);
    const uint32_t n_pairs = b.uniform_len ? pair + 1 : (len + 31u) / 32u;
    for (uint32_t p = pair; p < n_pairs; ++p) {
      const uint64_t j = base0 + first_base + 32ull * p;  // base index in the stream of the pair's first base
      const uint64_t w = j >> 4;                         // 16 bases per word
      const uint32_t sh = (uint32_t)(j & 15u) * 2u;
      const uint32_t w0 = stream[w], w1 = stream[w + 1], w2 = stream[w + 2];
      const unsigned long long bits = (unsigned long long)__builtin_amdgcn_alignbit(w1, w0, sh) |
                                      ((unsigned long long)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
      out[p] = make_uint2(gmx_even_bits(bits), gmx_even_bits(bits >> 1));
    }
  }
}

static int launch_batch(gmx_engine *e, const BatchInput &in, hipStream_t stream) {
  const uint64_t n_reads = in.n_reads, total_bases = in.total_bases;
  if (n_reads == 0) return GMX_OK;
  if (n_reads > (0x7fffffffull / GMX_FAST_ARENA) / 2) {  // path-node handles (offsets into the arena table) stay below 2^31
    gmx_set_error("batch too large: at most 44 M reads per launch (lower gmx_engine_opts.max_batch_reads)");
    return GMX_EINVAL;
  }
  // the batch before: redo what found the log full, drain when half full. FIRST: a replay reads that batch's queues and
  // retry lists, which a growing workspace (ensure_batch_capacity) replaces with fresh, uninitialised buffers.
  int rc = e->log_sites ? log_settle(e) : GMX_OK;
  if (rc) return rc;
  if ((rc = ensure_batch_capacity(e, n_reads))) return rc;
  const bool fold_reset = e->reset_pending && e->reset_stream == stream;
  if (e->reset_pending && !fold_reset && (rc = flush_reset(e))) return rc;
  e->reset_pending = false;
  if (!in.d_planes) {
    uint64_t need = total_bases / 32 + n_reads + 16;  // pairs; the slack covers the one-pair look-ahead of planes()
    if (need > e->cap_packed) {
      rc = e->alloc(&e->d_packed, need, false);
      if (rc) return rc;
      e->cap_packed = need;
    }
  }
  BatchView b{in.d_reads, in.d_offsets, in.d_seeds, (in.d_planes || in.d_twobit) ? in.d_skip : e->d_skip, in.d_planes ? in.d_planes : e->d_packed,
              (uint32_t)n_reads, (uint32_t)(e->opts.forward_only ? 1 : 0), in.uniform_len, (in.uniform_len + 31u) / 32u,
              e->keep_states ? 1u : 0u};
  const uint32_t region_inv = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (((uint64_t)GMX_REGIONS << 32) + e->dview.n_prg - 1) / std::max<uint32_t>(e->dview.n_prg, 1u));
  SearchOut o{};  // (member by member: the struct's order is not part of any contract)
  o.status = e->d_status;
  o.n_final = e->d_n_final;
  o.finals = e->d_finals;
  o.arena = e->d_arena;
  o.cover_recs = e->d_cover_recs;
  o.cover_rec_task = e->d_mapped;
  o.region_cap = (uint32_t)(e->cap_reads * 2);
  o.region_inv = region_inv;
  o.task_lists = e->d_task_lists;
  o.list_stride = (uint32_t)(e->cap_reads * 2);
  o.overflow_list = e->d_overflow;
  o.overflow2_list = e->d_overflow2;
  o.cover_overflow_list = e->d_cover_overflow;
  o.big_mapped_list = e->d_big_mapped;
  o.cover_mid_list = e->d_cover_mid;
  o.cover_general_list = e->d_cover_general;
  o.alive_list = e->d_alive;
  o.dead_list = e->d_dead;
  o.dead2_list = e->d_dead2;
  o.seed_cursor = e->d_seed_cursor;
  o.error = e->d_error;
  o.counters = e->d_counters;
  o.alive_seed = e->d_alive_seed;
  o.huge_list = e->d_huge;
  o.cover_huge_list = e->d_cover_huge;
  o.huge_retry = e->d_huge_retry;
  o.arena_stride = (uint32_t)(e->cap_reads * 2);
  o.inst_list = e->d_inst_list;
  o.inst_sa = e->d_inst_sa;
  o.inst_remaining = e->d_inst_remaining;
  o.inst_cap = e->inst_cap;
  o.inst_slots = !getenv("GMX_NO_INST") ? e->big.max_slots : 0u;
  o.slot_n_final = e->big.n_final;
  o.slot_task = e->big.task_of_slot;
  o.inst_mapped_list = e->d_inst_mapped;
  o.inst_arena = e->d_inst_arena;
  o.inst_states = e->d_inst_states;
  o.inst_first = e->d_inst_first;
  o.inst_remaining_width = e->d_inst_width;
  o.inst_serial_list = e->d_inst_serial;
  o.general_serial_list = e->d_general_serial;
  o.big_serial_list = e->d_big_serial;
  o.overflow3_list = e->d_overflow3;
  o.split_twice = getenv("GMX_NO_SPLIT2") ? 0u : 1u;
  o.general_rest_list = e->d_general_rest;
  o.single_rest_list = e->d_single_rest;
  o.park2 = e->d_park2;
  o.park2_n = e->d_park2_n;
  o.log_retry_list = e->d_log_retry[e->log_retry_side];
  o.log_retry_recs = e->d_log_retry_recs[e->log_retry_side];
  o.log_retry_huge = e->d_log_retry_huge[e->log_retry_side];
  o.stats = e->d_stats;
  uint32_t n_tasks = (uint32_t)n_reads * 2;
  if (e->keep_states) {  // test hook: a task that never reaches a kernel that writes its state count reads as "no state"
    HIP_TRY(hipMemsetAsync(e->d_n_final, 0, (size_t)n_tasks * sizeof(uint32_t), stream));
    e->keep_reads = n_reads;
  }
  gmx_engine::EvTriple ev{};
  if (e->timing) {
    HIP_TRY(hipEventCreate(&ev.s));
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventCreate(&ev.c));
    for (int k = 0; k < GMX_TK_N; ++k) {
      HIP_TRY(hipEventCreate(&ev.k[k][0]));
      HIP_TRY(hipEventCreate(&ev.k[k][1]));
    }
    ev.reads = n_reads;
    HIP_TRY(hipEventRecord(ev.s, stream));
  }
  // (timing leg: events attached to the dispatches themselves — their own start and end, as a kernel trace sees them)
  auto t_ev = [&](int k, int side) -> hipEvent_t {
    if (!e->timing) return nullptr;
    ev.timed |= 1u << k;
    return ev.k[k][side];
  };
  if (in.d_planes || in.d_twobit) {
    hipLaunchKernelGGL(gmx_batch_begin_kernel, dim3(fold_reset ? 256 : 1), dim3(1024), 0, stream, e->d_counters,
                       fold_reset ? e->d_fused : nullptr, fold_reset ? (uint32_t)(e->n_fused + 32) : 0u);
    if (in.d_twobit) {
      const uint64_t threads = in.uniform_len ? n_reads * ((in.uniform_len + 31u) / 32u) : n_reads;
      hipExtLaunchKernelGGL(gmx_unpack2_kernel, dim3((unsigned)std::min<uint64_t>((threads + 255) / 256, 1u << 20)), dim3(256), 0, stream,
                            t_ev(GMX_TK_UNPACK, 0), t_ev(GMX_TK_UNPACK, 1), 0u, b, in.d_twobit, in.twobit_base0, e->d_packed);
    }
  } else
    hipLaunchKernelGGL(gmx_pack_kernel, dim3((unsigned)((n_reads + GMX_PACK_READS - 1) / GMX_PACK_READS)), dim3(GMX_PACK_THREADS), 0, stream, b,
                       e->d_skip, e->d_packed, e->d_counters, fold_reset ? e->d_fused : nullptr, fold_reset ? (uint32_t)(e->n_fused + 32) : 0u);
  if (fold_reset && (rc = note_zeroed(e, stream))) return rc;  // (the batch's first kernel zeroed the accumulators: the twin's next batch waits for it)
  size_t lds = (size_t)GMX_STACK_DEPTH * GMX_STACK_WORDS * GMX_BLOCK * sizeof(uint32_t);
  const size_t big_lds = (size_t)GMX_BIG_LDS_DEPTH * GMX_STACK_WORDS * 64 * sizeof(uint32_t);
  dim3 task_grid((n_tasks + GMX_BLOCK - 1) / GMX_BLOCK);
  const bool seeded = e->dview.kmer_size2 != 0 && !getenv("GMX_NO_SEEDED");  // longer seed table: no probe phase (gmx_seed_kernel)
  if (seeded)
    hipExtLaunchKernelGGL(gmx_seed_kernel, dim3((n_tasks + GMX_SEED_THREADS * GMX_SEED_CHUNKS - 1) / (GMX_SEED_THREADS * GMX_SEED_CHUNKS)), dim3(GMX_SEED_THREADS), 0, stream,
                          t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o);
  else if (e->seed_cursor)
    hipExtLaunchKernelGGL(gmx_probe_kernel<true>, task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o,
                          e->probe_iters);
  else
    hipExtLaunchKernelGGL(gmx_probe_kernel<false>, task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o,
                          e->probe_iters);
  // fork 1: the probe kernel's overflow queue (few, long-running tasks) is served by the large-capacity kernel on a
  // side stream while the extend kernel runs, and so is the k-mer filter of the tasks the probe kernel found dead
  // (most reverse-complement tasks; the extend kernel queues its own dead tasks separately)
  HIP_TRY(hipEventRecord(e->ev_fork, stream));
  HIP_TRY(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
  CoverAcc acc{e->d_fused, e->d_log, e->d_log_cursor, e->log_cap, e->d_scratch_big, e->cover_big_lanes, e->opts.rng_mode,
               e->log_sites ? 1u : 0u, e->d_heap, e->heap_words, e->d_status, (uint32_t)n_reads * 2u, e->d_stats};
  if (seeded) {  // what gmx_seed_kernel sent to the large-capacity pass (reads in repeats), and its coverage: on side 2
    HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_fork, 0));
    const InstPools pools{0};
    // one lane per mapping instance of the reads in short repeats, then their coverage. (On a stream of its own this pair
    // gained nothing: the runtime then put two of the four streams on one hardware queue, and filter and extend kernel
    // ran one after the other.)
    hipLaunchKernelGGL(gmx_extend_inst_kernel, dim3(e->n_cus * 2), dim3(GMX_BLOCK), lds, e->side2_stream, e->dview, b, o, pools);
    if (e->coop) launch_cover_coop<5>(e, e->side2_stream, b, o, acc);
    launch_cover_lds<CoverEnvMid, 5>(e, e->side2_stream, b, o, acc, e->coop);
    hipLaunchKernelGGL(gmx_search_split_kernel, dim3(4096), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, 0);
    launch_cover_lds<CoverEnvMid, 4>(e, e->side2_stream, b, o, acc);
  } else {
    hipLaunchKernelGGL(gmx_search_big_kernel, dim3(1024), dim3(64), big_lds, e->side_stream, e->dview, b, o, e->big, 0);
  }
  HIP_TRY(hipEventRecord(e->ev_side1, e->side_stream));
  if (!gmx_ab_no_filter()) launch_filter(e, e->side_stream, task_grid, b, o, 0, t_ev(GMX_TK_FILTER0, 0), t_ev(GMX_TK_FILTER0, 1));
  // (timing leg: the events are attached to this very dispatch — its own start and end, as a kernel trace sees them —
  // instead of being recorded around it, where they add the gap to the kernel before and two barrier packets)
  hipEvent_t k0 = e->timing ? ev.a : nullptr, k1 = e->timing ? ev.b : nullptr;
  const uint32_t budget = e->extend_budget;
  if (seeded && e->seed_cursor)
    hipExtLaunchKernelGGL((gmx_extend_kernel<true, 1>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else if (seeded)
    hipExtLaunchKernelGGL((gmx_extend_kernel<false, 1>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else if (e->seed_cursor)
    hipExtLaunchKernelGGL((gmx_extend_kernel<true, 0>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else
    hipExtLaunchKernelGGL((gmx_extend_kernel<false, 0>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  if (budget) {  // the stragglers, compacted (a block that finds its part of the queue empty returns at once)
    for (uint32_t pass = 0; pass < e->extend_passes; ++pass) {
      const bool last = pass + 1 >= e->extend_passes;
      // (the last pass runs to the end, or — nested PRGs — to its cap, beyond which a task goes to the split search)
      const uint32_t budget2 = last ? e->extend_cap : e->extend_budget2[pass];
      const uint32_t pass_arg = pass | (last && e->extend_cap ? 0x80000000u : 0u);
      hipEvent_t p0 = pass == 0 ? t_ev(GMX_TK_EXTEND2, 0) : nullptr, p1 = pass == 0 ? t_ev(GMX_TK_EXTEND2, 1) : nullptr;
      if (e->seed_cursor)
        hipExtLaunchKernelGGL((gmx_extend_kernel<true, 2>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, p0, p1, 0u, e->dview, b, o, e->fuse, budget2, pass_arg);
      else
        hipExtLaunchKernelGGL((gmx_extend_kernel<false, 2>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, p0, p1, 0u, e->dview, b, o, e->fuse, budget2, pass_arg);
    }
  }
  // fork 2: the extend kernel's overflow queue, then the coverage of everything the large-capacity kernel mapped,
  // beside filter + coverage of the regular tasks
  HIP_TRY(hipEventRecord(e->ev_fork2, stream));
  HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_fork2, 0));
  HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_side1, 0));
  // the second filter pass (the tasks the extend kernel found dead) comes first here: side 1 is busy with the first pass
  // for most of the batch, and behind the few-lane kernels below it would end after the main stream's last kernel
  if (!gmx_ab_no_filter()) launch_filter(e, e->side2_stream, task_grid, b, o, 1, t_ev(GMX_TK_FILTER1, 0), t_ev(GMX_TK_FILTER1, 1));
  // the extend kernel's overflow queue (and the tasks whose instances ran out of their pools): the 16-lane split search
  // first, one lane with a whole slot for what that leaves
  static const bool split2 = getenv("GMX_NO_SPLIT2") == nullptr;
  if (split2) hipLaunchKernelGGL(gmx_search_split_kernel, dim3(1024), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, 1);
  hipLaunchKernelGGL(gmx_search_big_kernel, dim3(1024), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, split2 ? 2 : 1);
  if (e->coop) launch_cover_coop<2>(e, e->side2_stream, b, o, acc);
  launch_cover_lds<CoverEnvMid, 2>(e, e->side2_stream, b, o, acc, e->coop);
  // the general instances of the regular tasks: their queue is complete after the extend kernel unless the PRG is
  // nested (there gmx_cover_single_kernel hands tasks over), so they run on side 1, off the main stream (and not behind
  // the large-capacity pass's chain of few-lane kernels: with reads in repeats that chain is the batch's longest path)
  const bool general_on_side = !e->dview.is_nested;
  HIP_TRY(hipStreamWaitEvent(e->side_stream, e->ev_fork2, 0));
  const size_t one_lds = (size_t)4 * GMX_WIDE_LOCI * GMX_ONE_THREADS * sizeof(uint32_t);
  const bool one = !getenv("GMX_NO_COVER_ONE");
  auto launch_one = [&](hipStream_t st) {  // (with GMX_NO_COVER_ONE the kernel only passes its queue on: A/B runs)
    hipLaunchKernelGGL(gmx_cover_one_kernel, dim3(e->n_cus * 4), dim3(GMX_ONE_THREADS), one_lds, st, e->dview, b, o, e->big, acc, one ? 1u : 0u);
  };
  if (general_on_side) {
    launch_one(e->side_stream);
    if (e->coop) launch_cover_coop<3>(e, e->side_stream, b, o, acc);
    launch_cover_lds<CoverEnvLds, 3>(e, e->side_stream, b, o, acc, e->coop);
    launch_cover_lds<CoverEnv, 0>(e, e->side_stream, b, o, acc);
  }
  HIP_TRY(hipEventRecord(e->ev_filter, e->side_stream));
  // (Round 4 measured the records in PRG order — a radix sort of (position, record) pairs in front of this kernel, for the
  //  locality of the accumulator and table lines: at configs[3] the kernel took 508 us instead of 436 plus 120 us of sorting, at
  //  configs[4] 646 instead of 611: neighbouring lanes then hit the SAME accumulator words and their atomics serialise. Dropped.)
  if (e->dview.is_nested) {
    hipExtLaunchKernelGGL(gmx_cover_single_kernel<true>, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), 0, stream, t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u,
                          e->dview, b, o, acc);
  } else if (e->cover_jump) {  // most sites have geometry records: the lean kernel, then the few records it declined
    // (Measured and dropped: this kernel over the records queued by then BESIDE the extend kernel's passes over the stragglers,
    //  those moved to side 1 — at configs[3] the passes then took 335 us instead of 165 and the batch 1.13 ms instead of 1.07:
    //  the two kernels wait for the same thing, the memory system's rate of scattered accesses.)
    hipExtLaunchKernelGGL(gmx_cover_jump_kernel, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), (uint32_t)(GMX_STAGE_MAX * GMX_BLOCK * sizeof(uint32_t)), stream,
                          t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u, e->dview, b, o, acc);
    hipLaunchKernelGGL(gmx_cover_single_rest_kernel, dim3(e->n_cus * 2), dim3(GMX_BLOCK), 0, stream, e->dview, b, o, acc);
  } else {
    hipExtLaunchKernelGGL(gmx_cover_single_kernel<false>, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), 0, stream, t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u,
                          e->dview, b, o, acc);
  }
  // The batch's last coverage instance (1: what exceeded the regular scratch; its last block also serves the last tier,
  // whose search keeps its first pending entries in LDS) needs every other instance done except gmx_cover_single_kernel,
  // which queues nothing on a non-nested PRG: there it runs at the end of side 2, beside that kernel.
  hipStream_t last = general_on_side ? e->side2_stream : stream;
  if (!general_on_side) {
    launch_one(stream);
    if (e->coop) launch_cover_coop<3>(e, stream, b, o, acc);
    launch_cover_lds<CoverEnvLds, 3>(e, stream, b, o, acc, e->coop);
    launch_cover_lds<CoverEnv, 0>(e, stream, b, o, acc);
    HIP_TRY(hipEventRecord(e->ev_join, e->side2_stream));
    HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
  }
  HIP_TRY(hipStreamWaitEvent(last, e->ev_filter, 0));
  hipLaunchKernelGGL((gmx_cover_kernel<CoverEnvBig, 1>), dim3(e->cover_big_lanes / 64), dim3(64), big_lds, last, e->dview,
                     b, o, e->big, acc, 64u, 0u);
  if (general_on_side) {
    HIP_TRY(hipEventRecord(e->ev_join, e->side2_stream));
    HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
  }
  // (no pass over per-task status words: the read counters are added where each task's fate is decided, SearchOut::stats)
  if (e->timing) {
    HIP_TRY(hipEventRecord(ev.c, stream));
    e->pending.push_back(ev);
  }
  HIP_TRY(hipGetLastError());
  e->last_stream = stream;
  if (e->log_sites) {  // what log_settle() looks at before the next batch, and what a replay needs of this one
    e->last_b = b;
    e->last_o = o;
    e->last_acc = acc;
    e->last_big_lds = big_lds;
    if ((rc = log_state_enqueue(e, stream))) return rc;
  }
  return GMX_OK;
}

int gmx_map_reads_device(gmx_engine *e, const uint8_t *d_reads, const uint64_t *d_offsets, const uint32_t *d_seeds,
                         uint64_t n_reads, uint64_t total_bases, void *hip_stream) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  hipStream_t stream = (hipStream_t)hip_stream;
  uint64_t done = 0;
  while (done < n_reads) {
    uint64_t n = std::min<uint64_t>(e->opts.max_batch_reads, n_reads - done);
    BatchInput in;
    in.d_reads = d_reads;
    in.d_offsets = d_offsets + done;
    in.d_seeds = d_seeds + done;
    in.n_reads = n;
    in.total_bases = total_bases;
    int rc = launch_batch(e, in, stream);
    if (rc) return rc;
    done += n;
  }
  // An index with log sites: a batch that found the grouped log full is replayed from ITS inputs (read lengths, seeds), and
  // the caller may reuse its device buffers in stream order after this call: settle now (waits for the batch; engines
  // without log sites — every dense-count index — stay asynchronous).
  if (e->log_sites) return log_settle(e);
  return GMX_OK;
} GMX_GUARD_INT("gmx_map_reads_device")

// Is [p, p + bytes) page-locked memory the runtime can DMA from asynchronously (gmx_host_alloc, hipHostMalloc, registered)?
static bool gmx_is_pinned(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Large calls: chunks of <= 1 M reads through two staging slots; the upload of a chunk (copy stream, from the caller's
// buffer registered with the runtime for the duration of the call) runs beside the kernels of the one before.
static int map_reads_host_pipelined(gmx_engine *e, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                                    uint64_t n_reads, uint64_t chunk) {
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  const uint64_t first = offsets[0], total = offsets[n_reads] - first;
  const bool registered = !gmx_is_pinned(reads + first) &&
                          hipHostRegister(const_cast<uint8_t *>(reads + first), total, hipHostRegisterDefault) == hipSuccess;
  (void)hipGetLastError();
  int rc = GMX_OK;
  auto hip_ok = [&](hipError_t err, const char *what) {  // (no early return: the epilogue below always runs)
    if (err == hipSuccess) return true;
    gmx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    rc = GMX_EHIP;
    return false;
  };
  uint64_t done = 0;
  try {  // (an exception — host memory — must not skip the epilogue: the caller's buffer is registered, copies are in flight)
  for (uint32_t i = 0; done < n_reads && rc == GMX_OK; ++i) {
    gmx_engine::StageSlot &sl = e->stage[i & 1];
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    const uint64_t b0 = offsets[done], bases = offsets[done + n] - b0;
    if (sl.busy) {  // the chunk that used this slot two rounds ago
      if (!hip_ok(gmx_event_wait(sl.done), "hipEventSynchronize")) break;
      sl.busy = false;
    }
    if (!sl.copied) {
      if (!hip_ok(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate") ||
          !hip_ok(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate"))
        break;
    }
    if (bases > sl.cap_bases) {  // (the slot is idle: its superseded buffer can go at once)
      const uint64_t cb = std::max<uint64_t>(bases + bases / 8, 1 << 16);
      e->release(sl.d_reads);
      sl.d_reads = nullptr;
      sl.cap_bases = 0;
      if ((rc = e->alloc(&sl.d_reads, cb + 16, false))) break;
      sl.cap_bases = cb;
    }
    if (n > sl.cap_reads) {
      const uint64_t cr = std::max<uint64_t>(n, 1024);
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, cr + 1, false)) || (rc = e->alloc(&sl.d_seeds, cr, false))) break;
      if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
      if (sl.h_seeds) (void)hipHostFree(sl.h_seeds);
      sl.h_offsets = nullptr;
      sl.h_seeds = nullptr;
      if (!hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sl.h_offsets), (cr + 1) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc") ||
          !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sl.h_seeds), cr * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc"))
        break;
      sl.cap_reads = cr;
    }
    for (uint64_t j = 0; j <= n; ++j) sl.h_offsets[j] = offsets[done + j] - b0;
    memcpy(sl.h_seeds, seeds + done, n * sizeof(uint32_t));
    if (!hip_ok(hipMemcpyAsync(sl.d_reads, reads + b0, bases, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(reads)") ||
        !hip_ok(hipMemcpyAsync(sl.d_offsets, sl.h_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(offsets)") ||
        !hip_ok(hipMemcpyAsync(sl.d_seeds, sl.h_seeds, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(seeds)") ||
        !hip_ok(hipEventRecord(sl.copied, e->copy_stream), "hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(nullptr, sl.copied, 0), "hipStreamWaitEvent"))
      break;
    {
      BatchInput in;
      in.d_reads = sl.d_reads;
      in.d_offsets = sl.d_offsets;
      in.d_seeds = sl.d_seeds;
      in.n_reads = n;
      in.total_bases = bases;
      rc = launch_batch(e, in, nullptr);
    }
    if (rc) break;
    if (!hip_ok(hipEventRecord(sl.done, nullptr), "hipEventRecord")) break;
    sl.busy = true;
    done += n;
  }
  } catch (...) {
    rc = gmx_guard_catch("gmx_map_reads_host");
  }
  // common epilogue, error or not: nothing in flight reads the caller's buffer, the slots are idle, the buffer is unregistered
  (void)hipStreamSynchronize(e->copy_stream);
  (void)hipDeviceSynchronize();
  e->stage[0].busy = e->stage[1].busy = false;
  if (registered) (void)hipHostUnregister(const_cast<uint8_t *>(reads + first));
  (void)hipGetLastError();
  return rc ? rc : gmx_engine_sync(e);
}

static uint64_t gmx_feed_chunk(const gmx_engine *e);  // reads per launch of the host feeds (below)
int gmx_map_reads_host(gmx_engine *e, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                       uint64_t n_reads) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  {
    const uint64_t chunk = gmx_feed_chunk(e);
    if (n_reads > chunk && !getenv("GMX_HOST_SERIAL")) return map_reads_host_pipelined(e, reads, offsets, seeds, n_reads, chunk);
  }
  uint64_t done = 0;
  while (done < n_reads) {
    uint64_t n = std::min<uint64_t>(e->opts.max_batch_reads, n_reads - done);
    uint64_t b0 = offsets[done], b1 = offsets[done + n];
    uint64_t bases = b1 - b0;
    if (bases > e->cap_bases) {
      uint64_t cb = std::max<uint64_t>(bases, 1 << 16);
      int rc = e->alloc(&e->d_reads, cb + 16, false);
      if (rc) return rc;
      e->cap_bases = cb;
    }
    if (n > e->cap_stage_reads) {
      uint64_t cr = std::max<uint64_t>(n, 1024);
      int rc = e->alloc(&e->d_offsets, cr + 1, false);
      if (rc) return rc;
      rc = e->alloc(&e->d_seeds, cr, false);
      if (rc) return rc;
      e->cap_stage_reads = cr;
    }
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t i = 0; i <= n; ++i) rel[i] = offsets[done + i] - b0;
    HIP_TRY(hipMemcpy(e->d_reads, reads + b0, bases, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_offsets, rel.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_seeds, seeds + done, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    BatchInput in;
    in.d_reads = e->d_reads;
    in.d_offsets = e->d_offsets;
    in.d_seeds = e->d_seeds;
    in.n_reads = n;
    in.total_bases = bases;
    int rc = launch_batch(e, in, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));  // staging buffers are reused by the next batch
    if (e->log_sites && (rc = log_settle(e))) return rc;  // ... and a replay of this batch reads them: before they are overwritten
    done += n;
  }
  return gmx_engine_sync(e);
} GMX_GUARD_INT("gmx_map_reads_host")

// Which workspace and main stream the next launch of a host feed takes: the engine's own (NULL stream) and its twin's in turn
// (gmx_engine::twin). The launch that carries a queued reset goes to the engine itself (its first kernel zeroes the accumulators);
// the twin's next launch waits for that kernel.
struct BatchTarget {
  gmx_engine *eng;
  hipStream_t stream;
};
static int pick_target(gmx_engine *e, BatchTarget *out) {
  out->eng = e;
  out->stream = nullptr;
  e->last_on_twin = false;
  gmx_engine *tw = e->twin;
  if (!tw || e->twin_off || e->keep_states) return GMX_OK;
  if (e->reset_pending) {
    e->twin_toggle = 1;
    return GMX_OK;
  }
  if ((e->twin_toggle++ & 1u) == 0) return GMX_OK;
  if (tw->seen_zero_epoch != e->zero_epoch) {
    HIP_TRY(hipStreamWaitEvent(tw->main_stream, e->ev_zeroed, 0));
    tw->seen_zero_epoch = e->zero_epoch;
  }
  tw->timing = e->timing;
  out->eng = tw;
  out->stream = tw->main_stream;
  e->twin_in_flight = true;
  e->last_on_twin = true;
  return GMX_OK;
}

// planes: the bit planes (twobit = false) or the 2-bit stream as 32-bit words (twobit = true; gmx_map_reads_2bit_host)
static int map_reads_packed_impl(gmx_engine *e, const uint64_t *planes, bool twobit, const uint64_t *offsets, uint32_t uniform_len,
                                 const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads) {
  if (!e || !planes || !seeds || (!offsets && !uniform_len)) {
    gmx_set_error("gmx_map_reads_packed_host / gmx_map_reads_2bit_host: null argument (offsets may be null only with uniform_len)");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  static const bool two_copy_streams = getenv("GMX_TWO_COPY_STREAMS") != nullptr;
  if (two_copy_streams && !e->copy_stream2) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream2, hipStreamNonBlocking));
  const uint32_t ppr = (uniform_len + 31u) / 32u;
  auto base_at = [&](uint64_t r) -> uint64_t { return uniform_len ? r * uniform_len : offsets[r] - offsets[0]; };
  auto pair_at = [&](uint64_t r) -> uint64_t {  // 8-byte units from the call's first read to read r (gmx.h: layout of `planes`;
    if (twobit) return (base_at(r) + 31) >> 5;  //  a 2-bit stream: 32 bases per unit, rounded up)
    return uniform_len ? r * ppr : ((offsets[r] >> 5) - (offsets[0] >> 5)) + r;
  };
  // buffers the runtime cannot DMA from are registered for the duration of the call, which then waits for its uploads
  struct Reg { const void *p; bool on; };
  Reg regs[4] = {{planes, false}, {offsets, false}, {seeds, false}, {skip, false}};
  const uint64_t reg_bytes[4] = {pair_at(n_reads) * 8, (n_reads + 1) * 8, n_reads * 4, n_reads};
  bool all_pinned = true;
  for (int i = 0; i < 4; ++i) {
    if (!regs[i].p || gmx_is_pinned(regs[i].p)) continue;
    regs[i].on = hipHostRegister(const_cast<void *>(regs[i].p), reg_bytes[i], hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    all_pinned = false;
  }
  // seeds in place (gmx_engine_seeds_in_place): the kernels read the few seeds they need — a read draws only when it has
  // several equally good mapping classes — from the caller's page-locked buffer over PCIe; nothing is uploaded
  const uint32_t *d_seeds_host = nullptr;
  if (e->seeds_in_place && gmx_is_pinned(seeds)) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<uint32_t *>(seeds), 0) == hipSuccess && dp) d_seeds_host = static_cast<const uint32_t *>(dp);
    else (void)hipGetLastError();
  }
  const uint64_t chunk = gmx_feed_chunk(e);
  int rc = GMX_OK;
  auto hip_ok = [&](hipError_t err, const char *what) {
    if (err == hipSuccess) return true;
    gmx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    rc = GMX_EHIP;
    return false;
  };
  try {  // (as above: the epilogue unregisters the caller's buffers and waits for the copies that read them)
  for (uint64_t done = 0; done < n_reads && rc == GMX_OK;) {
    gmx_engine::PackSlot &sl = e->pslot[e->pslot_next];
    e->pslot_next = (e->pslot_next + 1) % 3;
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    uint64_t p0 = pair_at(done), pairs = pair_at(done + n) - p0;
    uint32_t twobit_base0 = 0;
    if (twobit) {  // the chunk's bases from the 8-byte unit holding its first one
      const uint64_t b0 = base_at(done), b1 = base_at(done + n);
      p0 = b0 >> 5;
      pairs = ((b1 + 31) >> 5) - p0;
      twobit_base0 = (uint32_t)(b0 & 31u);
    }
    if (sl.busy) {  // the batch that used this slot three chunks ago
      if (!hip_ok(gmx_event_wait(sl.done), "hipEventSynchronize")) break;
      sl.busy = false;
    }
    if (!sl.copied) {
      if (!hip_ok(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate") ||
          !hip_ok(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate"))
        break;
    }
    if (pairs + 16 > sl.cap_pairs) {  // (+ slack: the kernels fetch whole 16-byte pieces and one pair ahead)
      const uint64_t cp = pairs + pairs / 8 + 64;
      e->release(sl.d_planes);
      sl.d_planes = nullptr;
      sl.cap_pairs = 0;
      if ((rc = e->alloc(&sl.d_planes, cp, false))) break;
      sl.cap_pairs = cp;
    }
    if (n > sl.cap_reads) {
      const uint64_t cr = std::max<uint64_t>(n, 1024);
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      e->release(sl.d_skip);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.d_skip = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, cr + 1, false)) || (rc = e->alloc(&sl.d_seeds, cr, false)) || (rc = e->alloc(&sl.d_skip, cr, false))) break;
      sl.cap_reads = cr;
    }
    // One copy stream. (Round 5: the box's link carries 56-57 GB/s with nothing beside it, tools/exp/h2d_rate.py, this loop 51; with
    // consecutive batches' uploads alternating between TWO streams — a copy queued behind the one in flight — the median job is 2.5 %
    // faster, 1.395 against 1.365 G reads/s, but one job in five is 20-40 % slower — two copies share the link and both batches' kernels
    // start late — where one stream's jobs lie within 0.5 % of each other. GMX_TWO_COPY_STREAMS=1 selects it. Three and four streams:
    // worse. ONE batch's planes split over two streams reached 31-37 GB/s, and a kernel pulling the stream out of the caller's
    // page-locked memory itself 34 GB/s — both measured in round 3 and removed.)
    const hipStream_t cs = two_copy_streams && ((e->copy_toggle++) & 1u) ? e->copy_stream2 : e->copy_stream;
    if (!hip_ok(hipMemcpyAsync(sl.d_planes, planes + p0, pairs * 8, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(planes)")) break;
    if (!uniform_len &&
        !hip_ok(hipMemcpyAsync(sl.d_offsets, offsets + done, (n + 1) * 8, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(offsets)"))
      break;
    if (!d_seeds_host &&
        !hip_ok(hipMemcpyAsync(sl.d_seeds, seeds + done, n * 4, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(seeds)"))
      break;
    if (skip && !hip_ok(hipMemcpyAsync(sl.d_skip, skip + done, n, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(skip)")) break;
    BatchTarget tg;
    if ((rc = pick_target(e, &tg))) break;
    if (!hip_ok(hipEventRecord(sl.copied, cs), "hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(tg.stream, sl.copied, 0), "hipStreamWaitEvent"))
      break;
    BatchInput in;
    in.d_offsets = uniform_len ? nullptr : sl.d_offsets;
    in.d_seeds = d_seeds_host ? d_seeds_host + done : sl.d_seeds;
    in.d_planes = twobit ? nullptr : sl.d_planes;
    in.d_twobit = twobit ? reinterpret_cast<const uint32_t *>(sl.d_planes) : nullptr;
    in.twobit_base0 = twobit_base0;
    in.d_skip = skip ? sl.d_skip : nullptr;
    in.uniform_len = uniform_len;
    in.n_reads = n;
    in.total_bases = uniform_len ? n * (uint64_t)uniform_len : offsets[done + n] - offsets[done];
    if ((rc = launch_batch(tg.eng, in, tg.stream))) break;
    if (!hip_ok(hipEventRecord(sl.done, tg.stream), "hipEventRecord")) break;
    sl.busy = true;
    done += n;
  }
  } catch (...) {
    rc = gmx_guard_catch(twobit ? "gmx_map_reads_2bit_host" : "gmx_map_reads_packed_host");
  }
  // common epilogue: a failed call, or one that registered memory, leaves nothing in flight that reads the caller's buffers
  if (rc != GMX_OK || !all_pinned) {
    (void)hipStreamSynchronize(e->copy_stream);
    if (e->copy_stream2) (void)hipStreamSynchronize(e->copy_stream2);
    if (rc != GMX_OK) {
      (void)hipDeviceSynchronize();
      for (auto &sl : e->pslot) sl.busy = false;
    }
  }
  for (int i = 0; i < 4; ++i)
    if (regs[i].on) (void)hipHostUnregister(const_cast<void *>(regs[i].p));
  (void)hipGetLastError();
  return rc;
}

int gmx_map_reads_packed_host(gmx_engine *e, const uint64_t *planes, const uint64_t *offsets, uint32_t uniform_len,
                              const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads) try {
  return map_reads_packed_impl(e, planes, false, offsets, uniform_len, seeds, skip, n_reads);
} GMX_GUARD_INT("gmx_map_reads_packed_host")

// Reads per launch of the host / device-plane feeds: the whole call, up to max_batch_reads (4 M). Every batch ends with a tail
// of few-lane kernels — on a NESTED PRG with ~2 ms of a few straggler tasks (reads inside MSA regions: hundreds of dependent
// general iterations each) whatever its size —, and the tail is paid per launch (round 5, tools/exp/engines_in_flight.py,
// kernel pipeline): configs[2] maps 138 M reads/s in launches of 250 k reads, 388 M at 1 M, 640 M at 4 M; configs[3] 914 M ->
// 1 173 M, configs[4] 350 -> 412 M, configs[1] 2.30 -> 2.63 G from 1 M to 4 M. (Until round 5 a launch took at most 2^20 reads;
// a call's first upload is now up to four times as long, the uploads behind it still hide behind the kernels.)
static uint64_t gmx_feed_chunk(const gmx_engine *e) {
  static const char *env = getenv("GMX_FEED_CHUNK");
  if (env) return std::max<uint64_t>(1, std::min<uint64_t>(e->opts.max_batch_reads, strtoull(env, nullptr, 10)));
  return e->opts.max_batch_reads;
}

// bit planes already in HBM (gmx_ingest_*): nothing to upload; seeds in device memory, or page-locked and read in place
int gmx_map_reads_packed_device(gmx_engine *e, const uint64_t *d_planes, const uint64_t *d_offsets, uint32_t uniform_len,
                                const uint32_t *seeds, const uint8_t *d_skip, uint64_t n_reads) try {
  if (!e || !d_planes || !seeds || (!d_offsets && !uniform_len)) {
    gmx_set_error("gmx_map_reads_packed_device: null argument (d_offsets may be null only with uniform_len)");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  const uint32_t *d_seeds = seeds;
  if (gmx_is_pinned(seeds)) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<uint32_t *>(seeds), 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      gmx_set_error("gmx_map_reads_packed_device: the page-locked seeds have no device address");
      return GMX_EHIP;
    }
    d_seeds = static_cast<const uint32_t *>(dp);
  }
  const uint64_t chunk = gmx_feed_chunk(e);
  if (!uniform_len && n_reads > chunk) {
    gmx_set_error("gmx_map_reads_packed_device: with d_offsets a call takes at most 2^20 reads (and at most max_batch_reads)");
    return GMX_EINVAL;
  }
  const uint64_t ppr = (uniform_len + 31u) / 32u;
  for (uint64_t done = 0; done < n_reads;) {
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    BatchInput in;
    in.d_planes = reinterpret_cast<const uint2 *>(d_planes + done * ppr);
    in.d_offsets = uniform_len ? nullptr : d_offsets;
    in.d_seeds = d_seeds + done;
    in.d_skip = d_skip ? d_skip + done : nullptr;
    in.uniform_len = uniform_len;
    in.n_reads = n;
    in.total_bases = uniform_len ? n * (uint64_t)uniform_len : 0;  // (sizes the pack buffer of byte input only)
    BatchTarget tg;
    int rc = pick_target(e, &tg);
    if (rc) return rc;
    // (the planes were written by work on the NULL stream or on streams the caller has ordered before it — gmx_ingest_wait has
    //  returned —: the twin's stream needs no extra wait for them)
    if ((rc = launch_batch(tg.eng, in, tg.stream))) return rc;
    done += n;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_map_reads_packed_device")

int gmx_map_reads_2bit_host(gmx_engine *e, const uint64_t *stream, const uint64_t *offsets, uint32_t uniform_len, const uint32_t *seeds,
                            const uint8_t *skip, uint64_t n_reads) try {
  return map_reads_packed_impl(e, stream, true, offsets, uniform_len, seeds, skip, n_reads);
} GMX_GUARD_INT("gmx_map_reads_2bit_host")

void *gmx_engine_second_stream(gmx_engine *e) { return e && e->twin ? (void *)e->twin->main_stream : nullptr; }

int gmx_engine_seeds_in_place(gmx_engine *e, int on) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  e->seeds_in_place = on != 0;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_seeds_in_place")

int gmx_engine_sync_uploads(gmx_engine *e) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  if (e->copy_stream) {  // (sleeping, not polling: gmx_quiesce)
    if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventRecord(e->ev_wait, e->copy_stream));
    HIP_TRY(gmx_event_wait(e->ev_wait));
  }
  if (e->copy_stream2) {
    if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventRecord(e->ev_wait, e->copy_stream2));
    HIP_TRY(gmx_event_wait(e->ev_wait));
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_sync_uploads")

// page-locked allocations are remembered so that gmx_host_free knows which call returns them. Freed page-locked blocks
// of 1 MB or more are kept (up to 16 of them) and handed out again: pinning and unpinning 100 MB costs 10-20 ms each
// way, which a reads feed would otherwise pay at its start and again at its end.
static std::mutex g_host_mu;
struct HostBlock {
  uint64_t bytes;
  bool pinned;
};
static std::map<void *, HostBlock> g_host_live;
static std::vector<std::pair<void *, uint64_t>> g_host_spare;  // pinned blocks waiting for reuse
void *gmx_host_alloc(uint64_t bytes) try {
  bytes = std::max<uint64_t>(bytes, 1);
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t best = g_host_spare.size();
    for (size_t i = 0; i < g_host_spare.size(); ++i)
      if (g_host_spare[i].second >= bytes && g_host_spare[i].second <= 2 * bytes + (1u << 20) &&
          (best == g_host_spare.size() || g_host_spare[i].second < g_host_spare[best].second))
        best = i;
    if (best != g_host_spare.size()) {
      void *p = g_host_spare[best].first;
      g_host_live[p] = HostBlock{g_host_spare[best].second, true};
      g_host_spare.erase(g_host_spare.begin() + (long)best);
      return p;
    }
  }
  void *p = nullptr;
  static const unsigned alloc_flags = getenv("GMX_HOST_ALLOC_FLAGS") ? (unsigned)strtoul(getenv("GMX_HOST_ALLOC_FLAGS"), nullptr, 0) : hipHostMallocDefault;
  bool pinned = hipHostMalloc(&p, bytes, alloc_flags) == hipSuccess && p;
  if (!pinned) {
    (void)hipGetLastError();
    p = malloc(bytes);
  }
  if (p) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_live[p] = HostBlock{bytes, pinned};
  }
  return p;
} GMX_GUARD_PTR("gmx_host_alloc")
void gmx_host_free(void *p) try {
  if (!p) return;
  HostBlock blk{0, false};
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_live.find(p);
    if (it == g_host_live.end()) return;
    blk = it->second;
    g_host_live.erase(it);
    if (blk.pinned && blk.bytes >= (1u << 20) && g_host_spare.size() < 16) {
      g_host_spare.emplace_back(p, blk.bytes);
      return;
    }
  }
  if (blk.pinned)
    (void)hipHostFree(p);
  else
    free(p);
} GMX_GUARD_VOID("gmx_host_free")

// Sizes the batch workspace and the staging buffers of the _host entry point ahead of the first call (otherwise the first
// call allocates them, and a later, larger call allocates them again).
int gmx_engine_reserve(gmx_engine *e, uint64_t n_reads, uint64_t n_bases) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  n_reads = std::min<uint64_t>(n_reads, e->opts.max_batch_reads);
  int rc = ensure_batch_capacity(e, n_reads);
  if (rc) return rc;
  const uint64_t need = n_bases / 32 + n_reads + 16;
  if (need > e->cap_packed) {
    if ((rc = e->alloc(&e->d_packed, need, false))) return rc;
    e->cap_packed = need;
  }
  if (n_bases > e->cap_bases) {
    if ((rc = e->alloc(&e->d_reads, n_bases + 16, false))) return rc;
    e->cap_bases = n_bases;
  }
  if (n_reads > e->cap_stage_reads) {
    if ((rc = e->alloc(&e->d_offsets, n_reads + 1, false)) || (rc = e->alloc(&e->d_seeds, n_reads, false))) return rc;
    e->cap_stage_reads = n_reads;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_reserve")

// The same for gmx_map_reads_packed_host: the batch workspace, the copy stream and the three upload slots (bit planes,
// offsets, seeds, skip flags) for chunks of up to n_reads reads / n_pairs plane pairs.
int gmx_engine_reserve_packed(gmx_engine *e, uint64_t n_reads, uint64_t n_pairs) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  n_reads = std::min<uint64_t>(n_reads, gmx_feed_chunk(e));
  int rc = ensure_batch_capacity(e, n_reads);
  if (rc) return rc;
  if (e->twin && (rc = ensure_batch_capacity(e->twin, n_reads))) return rc;
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (auto &sl : e->pslot) {
    if (sl.busy) continue;
    if (!sl.copied) {
      HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync));
      HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
    }
    if (n_pairs + 16 > sl.cap_pairs) {
      e->release(sl.d_planes);
      sl.d_planes = nullptr;
      sl.cap_pairs = 0;
      if ((rc = e->alloc(&sl.d_planes, n_pairs + 64, false))) return rc;
      sl.cap_pairs = n_pairs + 64;
    }
    if (n_reads > sl.cap_reads) {
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      e->release(sl.d_skip);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.d_skip = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, n_reads + 1, false)) || (rc = e->alloc(&sl.d_seeds, n_reads, false)) ||
          (rc = e->alloc(&sl.d_skip, n_reads, false)))
        return rc;
      sl.cap_reads = n_reads;
    }
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_reserve_packed")

int gmx_engine_sync(gmx_engine *e) try {
  HIP_TRY(hipSetDevice(e->opts.device));
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  {
    int qrc = gmx_quiesce(e);
    if (qrc) return qrc;
    if (e->twin && (qrc = gmx_quiesce(e->twin))) return qrc;
    e->twin_in_flight = false;
  }
  HIP_TRY(hipStreamSynchronize(e->last_stream));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t c[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(c + 2, e->d_error, 8, hipMemcpyDeviceToHost));
  if (c[2] == 0 && e->twin) {  // (the twin's batches report in its own words)
    HIP_TRY(hipMemcpy(c + 2, e->twin->d_error, 8, hipMemcpyDeviceToHost));
    if (c[2] != 0) HIP_TRY(hipMemset(e->twin->d_error, 0, 8));
  }
  if (c[2] != 0) {
    HIP_TRY(hipMemset(e->d_error, 0, 8));
    char msg[256];
    if (c[2] == GMX_TASK_LOGFULL) {
      snprintf(msg, sizeof(msg),
               "a read's records exceed the whole grouped-allele-count log (sites without dense group counters; %u words): "
               "nothing of it was recorded: raise gmx_engine_opts.log_cap_words",
               e->log_cap);
      gmx_set_error(msg);
      return GMX_ECAP;
    }
    if (c[2] == GMX_TASK_OVERFLOW) {
      snprintf(msg, sizeof(msg),
               "read %u (orientation %u) needs more memory for its search states or its mapping instances than the whole "
               "last-tier heap holds (%llu bytes); nothing of this read was recorded: raise gmx_engine_opts.huge_heap_bytes "
               "(GMX_HUGE_HEAP_BYTES)",
               c[3] >> 1, c[3] & 1, (unsigned long long)e->heap_words * 4);
      gmx_set_error(msg);
      return GMX_ECAP;
    }
    snprintf(msg, sizeof(msg),
             "read %u (orientation %u): inconsistent variant path (the reference throws/asserts here: a site "
             "traversed twice or an exit that does not match the entered site)",
             c[3] >> 1, c[3] & 1);
    gmx_set_error(msg);
    return GMX_EREF;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_sync")

int gmx_engine_enable_timing(gmx_engine *e, int on) try {
  e->timing = on != 0;
  if (e->twin) e->twin->timing = e->timing;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_enable_timing")

int gmx_engine_timing(gmx_engine *e, gmx_timing *out) try {
  HIP_TRY(hipSetDevice(e->opts.device));
  if (e->twin) {  // (the twin's batches: its events, added to this engine's sums)
    e->pending.insert(e->pending.end(), e->twin->pending.begin(), e->twin->pending.end());
    e->twin->pending.clear();
  }
  for (auto &ev : e->pending) {
    HIP_TRY(hipEventSynchronize(ev.c));
    float ms0 = 0, ms1 = 0, ms2 = 0;
    HIP_TRY(hipEventElapsedTime(&ms0, ev.s, ev.a));
    HIP_TRY(hipEventElapsedTime(&ms1, ev.a, ev.b));
    HIP_TRY(hipEventElapsedTime(&ms2, ev.b, ev.c));
    e->search_ms += ms1;
    e->cover_ms += ms0 + ms2;
    e->search_launches++;
    e->cover_launches++;
    e->timed_reads += ev.reads;
    for (int k = 0; k < GMX_TK_N; ++k) {
      if (ev.timed & (1u << k)) {  // (events of other streams: complete, ev.c is behind the batch's join)
        float ms = 0;
        HIP_TRY(hipEventSynchronize(ev.k[k][1]));
        HIP_TRY(hipEventElapsedTime(&ms, ev.k[k][0], ev.k[k][1]));
        e->kernel_ms[k] += ms;
        e->kernel_launches[k]++;
      }
      (void)hipEventDestroy(ev.k[k][0]);
      (void)hipEventDestroy(ev.k[k][1]);
    }
    (void)hipEventDestroy(ev.s);
    (void)hipEventDestroy(ev.a);
    (void)hipEventDestroy(ev.b);
    (void)hipEventDestroy(ev.c);
  }
  e->pending.clear();
  out->search_ms = e->search_ms;
  out->search_launches = e->search_launches;
  out->cover_ms = e->cover_ms;
  out->cover_launches = e->cover_launches;
  out->reads = e->timed_reads;
  for (int k = 0; k < GMX_TIMED_KERNELS; ++k) {
    out->kernel_ms[k] = e->kernel_ms[k];
    out->kernel_launches[k] = e->kernel_launches[k];
    e->kernel_ms[k] = 0;
    e->kernel_launches[k] = 0;
  }
  e->search_ms = e->cover_ms = 0;
  e->search_launches = e->cover_launches = e->timed_reads = 0;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_timing")

int gmx_engine_queue_counts(gmx_engine *e, gmx_queue_counts *out) try {
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t raw[GMX_N_COUNTERS * GMX_CNT_STRIDE];
  HIP_TRY(hipMemcpy(raw, (e->twin && e->last_on_twin ? e->twin : e)->d_counters, sizeof(raw), hipMemcpyDeviceToHost));  // (the workspace of the LAST launch)
  auto c = [&](int i) { return (uint64_t)raw[i * GMX_CNT_STRIDE]; };
  out->mapped = 0;
  for (int r = 0; r < GMX_REGIONS; ++r) out->mapped += c(16 + r);
  out->mapped += c(8);
  out->alive = c(5);
  out->dead = c(6) + c(12);
  out->overflow_probe = c(1);
  out->overflow_extend = c(9);
  out->big_mapped = c(7);
  out->cover_general = c(8);
  out->cover_mid = c(13);
  out->cover_overflow = c(4);
  out->seed_cursor = e->seed_cursor ? 1 : 0;
  out->inst_mapped = c(25);
  out->huge_search = c(11);
  out->huge_cover = c(15);
  out->log_replays = e->log_replays;
  out->log_replayed_entries = e->log_replayed_entries;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_queue_counts")

int gmx_coverage_device(gmx_engine *e, gmx_device_coverage *out) try {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  if (e->twin && e->twin_in_flight) {  // (the caller is about to read the block: the twin's batches first)
    int qrc = gmx_quiesce(e->twin);
    if (qrc) return qrc;
    e->twin_in_flight = false;
  }
  out->allele_sum = out->per_base = out->grouped = nullptr;  // interleaved in the block: use `fused`, or gmx_coverage_fetch
  out->n_allele_sum = e->n_allele;
  out->n_per_base = e->n_pb;
  out->n_grouped = e->n_grouped;
  out->stats = e->d_stats;
  out->n_stats = 5;
  out->fused = e->d_fused;
  out->n_fused = e->n_fused;
  return GMX_OK;
} GMX_GUARD_INT("gmx_coverage_device")

int gmx_coverage_reduce_begin(gmx_engine *e, void *hip_stream) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  {
    int trc = twin_join(e, (hipStream_t)hip_stream);  // (the exchange reads what the twin's batches record)
    if (trc) return trc;
  }
  hipLaunchKernelGGL(gmx_stats_limbs_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, e->d_stats, e->d_limbs, 0);
  HIP_TRY(hipGetLastError());
  return GMX_OK;
} GMX_GUARD_INT("gmx_coverage_reduce_begin")

int gmx_coverage_reduce_end(gmx_engine *e, void *hip_stream) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  hipLaunchKernelGGL(gmx_stats_limbs_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, e->d_stats, e->d_limbs, 1);
  HIP_TRY(hipGetLastError());
  return GMX_OK;
} GMX_GUARD_INT("gmx_coverage_reduce_end")

int gmx_coverage_fetch(gmx_engine *e, uint32_t *allele_sum, uint32_t *per_base, uint32_t *grouped, gmx_stats *stats) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<uint32_t> block(std::max<size_t>(e->n_acc, 1));
  if (e->n_acc) HIP_TRY(hipMemcpy(block.data(), e->d_fused, e->n_acc * 4, hipMemcpyDeviceToHost));
  if (allele_sum) for (size_t i = 0; i < e->phys_allele.size(); ++i) allele_sum[i] = block[e->phys_allele[i]];
  if (per_base) for (size_t i = 0; i < e->phys_pb.size(); ++i) per_base[i] = block[e->phys_pb[i]];
  if (grouped) for (size_t i = 0; i < e->phys_grouped.size(); ++i) grouped[i] = block[e->phys_grouped[i]];
  for (size_t i = 0; i + 3 < e->hit_fix.size(); i += 4) {  // a hit = one each of allele-sum, group {allele} and the base
    const uint32_t hits = block[e->hit_fix[i]];
    if (allele_sum) allele_sum[e->hit_fix[i + 1]] += hits;
    if (grouped) grouped[e->hit_fix[i + 2]] += hits;
    if (per_base) per_base[e->hit_fix[i + 3]] += hits;
  }
  if (stats) {
    unsigned long long s[5];
    HIP_TRY(hipMemcpy(s, e->d_stats, sizeof(s), hipMemcpyDeviceToHost));
    stats->all_reads_count = s[0];
    stats->skipped_reads_count = s[1];
    stats->missing_kmer_reads_count = s[2];
    stats->no_extension_reads_count = s[3];
    stats->exact_mapped_reads_count = s[4];
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_coverage_fetch")

int64_t gmx_coverage_fetch_grouped_log(gmx_engine *e, uint32_t *out, uint64_t cap_words) try {
  if (!e) return GMX_EINVAL;
  if (hipSetDevice(e->opts.device) != hipSuccess) return GMX_EHIP;
  if (log_settle(e)) return GMX_EHIP;
  if (gmx_log_drain(e, 0)) return GMX_EHIP;
  uint64_t n = 0;
  for (auto const &kv : e->log_counts) {  // [site_index, n_ids | GMX_LOG_COUNTED, count lo, count hi, ids...]
    const uint64_t words = 4 + (kv.first.size() - 1);
    if (out && n + words <= cap_words) {
      out[n] = kv.first[0];
      out[n + 1] = (uint32_t)(kv.first.size() - 1) | GMX_LOG_COUNTED;
      out[n + 2] = (uint32_t)kv.second;
      out[n + 3] = (uint32_t)(kv.second >> 32);
      for (size_t j = 1; j < kv.first.size(); ++j) out[n + 3 + j] = kv.first[j];
    }
    n += words;
  }
  return (int64_t)n;
} GMX_GUARD_INT("gmx_coverage_fetch_grouped_log")

int gmx_coverage_import_grouped_log(gmx_engine *e, const uint32_t *records, uint64_t n_words, int replace) try {
  if (!e || (!records && n_words)) {
    gmx_set_error("gmx_coverage_import_grouped_log: null argument");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  return gmx_engine_log_import(e, records, (size_t)n_words, replace != 0);
} GMX_GUARD_INT("gmx_coverage_import_grouped_log")

}  // extern "C"

void gmx_engine_raw(gmx_engine *e, GmxEngineRaw *out) {
  (void)flush_reset(e);
  out->device = e->opts.device;
  out->d_fused = e->d_fused;
  out->n_fused = e->n_fused;
  out->log_sites = e->log_sites;
}

int gmx_engine_log_export(gmx_engine *e, std::vector<uint32_t> &out) {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  const int64_t n = gmx_coverage_fetch_grouped_log(e, nullptr, 0);
  if (n < 0) return (int)n;
  out.assign((size_t)n, 0);
  if (n && gmx_coverage_fetch_grouped_log(e, out.data(), (uint64_t)n) < 0) return GMX_EHIP;
  return GMX_OK;
}

int gmx_engine_log_import(gmx_engine *e, const uint32_t *w, size_t n_words, bool replace) {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  if (replace) {
    int rc = gmx_log_drain(e, 0);  // whatever is still on the device belongs to the totals being replaced
    if (rc) return rc;
    e->log_counts.clear();
  }
  std::vector<uint32_t> key;
  for (size_t i = 0; i < n_words;) {
    if (w[i] == GMX_LOG_PAD) {
      ++i;
      continue;
    }
    if (i + 2 > n_words) break;
    const uint32_t n = w[i + 1] & ~GMX_LOG_COUNTED;
    const size_t head = (w[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
    if (i + head + n > n_words) {
      gmx_set_error("corrupt grouped log");
      return GMX_EINVAL;
    }
    const uint64_t count = head == 4 ? ((uint64_t)w[i + 2] | ((uint64_t)w[i + 3] << 32)) : 1;
    key.assign(1, w[i]);
    key.insert(key.end(), w + i + head, w + i + head + n);
    e->log_counts[key] += count;
    i += head + n;
  }
  return GMX_OK;
}

#include "gmx_engine_debug.h"  // test hooks (final SearchStates of a task, the search loop on given states)
