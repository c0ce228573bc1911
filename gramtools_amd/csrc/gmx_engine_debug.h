// gmx_engine_debug.h — test hooks of the engine; part of gmx_engine.hip's translation unit (included at its end).
//
// SURVEY §7 step 3 asks for "per-read final SearchStates against the oracle". The reference's unit tests pin the search at
// the level of SearchStates (tests/genotype/quasimap/search/test_vBWT_jump.cpp:55-405, test_encapsulated_search.cpp:28-254,
// test_BWT_search.cpp, the search_read_backwards cases of test_quasimap.cpp); these hooks let tests/ run those vectors on the
// HIP kernels' own code instead of only through the coverage they lead to:
//   gmx_debug_final_states   the final states of one task of the LAST batch, wherever the production pipeline left them
//                            (fast tier, instance lanes, a large-capacity slot)
//   gmx_debug_search         the device's search loop (dfs_run_wave over BigCtx: gmx_search_big_kernel's) on one read from
//                            caller-given states, read positions `from` down to `stop`
//   gmx_debug_encapsulate    gmx_final_items (gmx_cover.h) — the device's handle_allele_encapsulated_states — on given states
// States travel as words in the k-mer index's own serialisation (gmx_types.h, GmxSeed):
//   [n_states, {lo, hi, n_traversed, n_traversing, (site, allele) x n_traversed, site x n_traversing}*]      (push order)
// with SA intervals as the reference has them (text-form states are converted back through the inverse suffix array).
// Nothing here is on the mapping path.

struct GmxDebugPools {  // one task's worth of large-capacity pools
  GmxFinalState *states = nullptr;
  uint32_t *stack = nullptr;
  GmxPathNode *arena = nullptr;
  uint32_t *result = nullptr;  // [status, n_out, arena_n]
  uint32_t *in = nullptr;
  uint2 *planes = nullptr;
  uint32_t cap_states = 0, cap_nodes = 0, cap_in = 0, cap_pairs = 0;
};

// lane 0 of one wave: push the given states at read position `from`, run the search loop down to `stop`
__global__ void __launch_bounds__(64) gmx_debug_search_kernel(GmxIndexView ix, BatchView b, const uint32_t *in, uint32_t from, uint32_t stop,
                                                              uint32_t mode, uint32_t from_seed_table, BigOut g, uint32_t *result) {
  const bool active = threadIdx.x == 0;
  BigCtx ctx;
  ctx.sp = 0;
  ctx.cap = g.max_states;
  ctx.stack = g.stack;
  ctx.arena = g.arena;
  ctx.arena_n = 0;
  ctx.arena_cap = g.max_path_nodes;
  ctx.status = GMX_TASK_MAPPED;
  ctx.out = g.states;
  ctx.n_out = 0;
  ctx.out_cap = g.max_states;
  ReadRef r;
  r.w = b.packed;
  r.len = 0;
  r.rc = false;
  r.cur_idx = 0xFFFFFFFFu;
  r.cur = make_uint2(0, 0);
  bool run = false;
  if (active) {
    r = task_read(b, 0);
    if (from_seed_table) {  // as the large-capacity pass seeds a task (quasimap.cpp:235-241), from the table of k
      from = r.len - ix.kmer_size;
      load_seed(ix, ix.seeds, kmer_code(r, from, ix.kmer_size), ctx,
                [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) { return ctx.push(lo, hi, tvd, tvg, from, GMX_MODE_STATE); });
    } else {
      const uint32_t *p = in;
      const uint32_t ns = *p++;
      for (uint32_t s = 0; s < ns && ctx.status == GMX_TASK_MAPPED; ++s) {
        const uint32_t lo = p[0], hi = p[1], nt = p[2], ng = p[3];
        p += 4;
        uint32_t tvd = GMX_NIL, tvg = GMX_NIL;
        bool ok = true;
        for (uint32_t j = 0; j < nt; ++j, p += 2)
          if (ok) ok = (tvd = ctx.arena_new(p[0], (int32_t)p[1], tvd)) != GMX_NIL;
        for (uint32_t j = 0; j < ng; ++j, ++p)
          if (ok) ok = (tvg = ctx.arena_new(p[0], -1, tvg)) != GMX_NIL;
        if (!ok || !ctx.push(lo, hi, tvd, tvg, from, mode)) ctx.fail(GMX_TASK_OVERFLOW);
      }
    }
    run = ctx.status == GMX_TASK_MAPPED;
  }
  GmxLane ln;
  dfs_run_wave<2, false>(ix, ctx, r, stop, run, 0, ln);
  if (active) {
    result[0] = ctx.status;
    result[1] = ctx.n_out;
    result[2] = ctx.arena_n;
  }
}

// gmx_final_items on given final states: out = [n_items, n_nonvariant, {lo, hi, tvd, tvg, site, allele} x n_items, position index x n_nonvariant]
__global__ void gmx_debug_items_kernel(GmxIndexView ix, const GmxFinalState *finals, uint32_t n_final, uint32_t *out, uint32_t cap_items) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint32_t n_items = 0, n_nonvar = 0;
  uint32_t *items = out + 2, *nonvar = out + 2 + 6 * (size_t)cap_items;
  const bool ok = gmx_final_items(
      ix, finals, n_final,
      [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg, uint32_t es, int32_t ea) -> bool {
        if (n_items >= cap_items) return false;
        uint32_t *it = items + 6 * (size_t)n_items++;
        it[0] = lo, it[1] = hi, it[2] = tvd, it[3] = tvg, it[4] = es, it[5] = (uint32_t)ea;
        return true;
      },
      [&](uint32_t i) {
        if (n_nonvar < cap_items) nonvar[n_nonvar] = i;
        ++n_nonvar;
      });
  out[0] = ok ? n_items : 0xFFFFFFFFu;
  out[1] = n_nonvar;
}

namespace {

struct DebugWriter {  // serialises states into the caller's buffer; counts the words even when they do not fit
  uint32_t *out;
  uint64_t cap, n = 0;
  void put(uint32_t w) {
    if (out && n < cap) out[n] = w;
    ++n;
  }
};

int debug_isa(gmx_engine *e) {
  if (!e->debug_isa.empty()) return GMX_OK;
  std::vector<uint32_t> sa(e->dview.n);
  HIP_TRY(hipMemcpy(sa.data(), e->dview.sa, sa.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  e->debug_isa.assign(sa.size(), 0u);
  for (size_t i = 0; i < sa.size(); ++i)
    if (sa[i] < sa.size()) e->debug_isa[sa[i]] = (uint32_t)i;
  return GMX_OK;
}

// one state -> words; `node(h)` fetches a path node by handle
template <class NodeAt>
int debug_put_state(gmx_engine *e, DebugWriter &w, const GmxFinalState &st, NodeAt node) {
  uint32_t lo = st.lo, hi = st.hi;
  if (hi == GMX_TEXT_MARK) {  // text form: a = PRG position -> its suffix-array index
    if (lo >= e->debug_isa.size()) {
      gmx_set_error("debug states: text-form state outside the PRG");
      return GMX_EREF;
    }
    lo = hi = e->debug_isa[lo];
  }
  std::vector<std::pair<uint32_t, int32_t>> tvd;
  std::vector<uint32_t> tvg;
  for (uint32_t h = st.traversed; h != GMX_NIL;) {
    GmxPathNode nd;
    int rc = node(h, nd);
    if (rc) return rc;
    tvd.emplace_back(nd.site, nd.allele);
    h = nd.next;
    if (tvd.size() > (1u << 20)) {
      gmx_set_error("debug states: path list does not end");
      return GMX_EREF;
    }
  }
  for (uint32_t h = st.traversing; h != GMX_NIL;) {
    if (gmx_h_inline(h)) {  // a single entered site, inline in the handle
      tvg.push_back(5u + 2u * (h & ~GMX_INLINE_FLAG));
      break;
    }
    GmxPathNode nd;
    int rc = node(h, nd);
    if (rc) return rc;
    tvg.push_back(nd.site);
    h = nd.next;
    if (tvg.size() > (1u << 20)) {
      gmx_set_error("debug states: path list does not end");
      return GMX_EREF;
    }
  }
  w.put(lo);
  w.put(hi);
  w.put((uint32_t)tvd.size());
  w.put((uint32_t)tvg.size());
  for (size_t i = tvd.size(); i-- > 0;) {  // the lists are newest first; the reference's vectors are in push order
    w.put(tvd[i].first);
    w.put((uint32_t)tvd[i].second);
  }
  for (size_t i = tvg.size(); i-- > 0;) w.put(tvg[i]);
  return GMX_OK;
}

int debug_pools(gmx_engine *e, GmxDebugPools &p, uint32_t n_in_words, uint32_t n_pairs) {
  const uint32_t S = std::max<uint32_t>(e->big.max_states, 64u), N = std::max<uint32_t>(e->big.max_path_nodes, 64u);
  int rc;
  if ((rc = e->alloc(&p.states, S, false))) return rc;
  if ((rc = e->alloc(&p.stack, (size_t)S * GMX_STACK_WORDS, false))) return rc;
  if ((rc = e->alloc(&p.arena, N, false))) return rc;
  if ((rc = e->alloc(&p.result, 4, true))) return rc;
  if ((rc = e->alloc(&p.in, std::max<uint32_t>(n_in_words, 1u), false))) return rc;
  if ((rc = e->alloc(&p.planes, n_pairs + 16, true))) return rc;
  p.cap_states = S;
  p.cap_nodes = N;
  return GMX_OK;
}
void debug_pools_free(gmx_engine *e, GmxDebugPools &p) {
  e->release(p.states);
  e->release(p.stack);
  e->release(p.arena);
  e->release(p.result);
  e->release(p.in);
  e->release(p.planes);
}

}  // namespace

extern "C" {

int gmx_engine_debug_keep_states(gmx_engine *e, int on) try {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  e->keep_states = on != 0;
  return GMX_OK;
} GMX_GUARD_INT("gmx_engine_debug_keep_states")

int gmx_debug_final_states(gmx_engine *e, uint64_t task, uint32_t *out, uint64_t cap_words, uint64_t *n_words, int *tier) try {
  if (!e || !n_words) {
    gmx_set_error("gmx_debug_final_states: null argument");
    return GMX_EINVAL;
  }
  if (!e->keep_states || task >= 2 * e->keep_reads) {
    gmx_set_error("gmx_debug_final_states: no such task in the last batch (gmx_engine_debug_keep_states first)");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  int rc = debug_isa(e);
  if (rc) return rc;
  uint32_t raw[GMX_N_COUNTERS * GMX_CNT_STRIDE];
  HIP_TRY(hipMemcpy(raw, e->d_counters, sizeof(raw), hipMemcpyDeviceToHost));
  auto c = [&](int i) { return raw[i * GMX_CNT_STRIDE]; };
  DebugWriter w{out, cap_words};
  // the last tier keeps nothing (its pools are slices of a heap the next work item reuses)
  {
    const uint32_t n_huge = c(11);
    std::vector<uint32_t> huge(n_huge);
    if (n_huge) HIP_TRY(hipMemcpy(huge.data(), e->d_huge, n_huge * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t t : huge)
      if (t == task) {
        gmx_set_error("gmx_debug_final_states: the task was searched by the last tier, which keeps no states");
        return GMX_ECAP;
      }
  }
  // large-capacity slots: instance 0 of the split search / the probe kernel's queue [0, c1), the extend kernel's queue
  // [c1, c1 + c9), the one-lane search of what the split search left [c1 + c9, c1 + c9 + c29). A task redone by a later
  // tier owns a later slot, and the slots of its failed attempts say "no state": the last slot with states counts.
  const uint32_t used = (uint32_t)std::min<uint64_t>((uint64_t)c(1) + c(9) + c(29), e->big.max_slots);
  if (used) {
    std::vector<uint32_t> owner(used), nf(used), over(std::min<uint32_t>(c(1), used));
    HIP_TRY(hipMemcpy(owner.data(), e->big.task_of_slot, used * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nf.data(), e->big.n_final, used * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (!over.empty()) HIP_TRY(hipMemcpy(over.data(), e->d_overflow, over.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // which of the first c(1) slots were really written this batch: the queue entry names the task
    bool in_slots = false;
    for (uint32_t s = used; s-- > 0;) {
      const bool first_q = s < over.size();
      if (first_q && (over[s] & ~GMX_INST_FLAG) != task) continue;
      if (!first_q && owner[s] != task) continue;
      in_slots = true;
      if (first_q && owner[s] != task) continue;  // (queued, never reached: the slot's content is another batch's)
      if (nf[s] == 0) continue;
      const bool inst = first_q && (over[s] & GMX_INST_FLAG) != 0;
      const uint32_t n = nf[s];
      std::vector<GmxFinalState> st(n);
      if (inst) {
        uint32_t first = 0, width = 0;
        HIP_TRY(hipMemcpy(&first, e->d_inst_first + s, 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&width, e->d_inst_width + s, 4, hipMemcpyDeviceToHost));
        if (n > width * GMX_INST_STATES) continue;  // its instances ran out of their pools: redone elsewhere
        HIP_TRY(hipMemcpy(st.data(), e->d_inst_states + (size_t)first * GMX_INST_STATES, n * sizeof(GmxFinalState), hipMemcpyDeviceToHost));
        const GmxPathNode *arena = e->d_inst_arena + (size_t)first * GMX_FAST_ARENA;
        w.put(n);
        for (uint32_t i = 0; i < n; ++i) {
          rc = debug_put_state(e, w, st[i], [&](uint32_t h, GmxPathNode &nd) {
            HIP_TRY(hipMemcpy(&nd, arena + h, sizeof(nd), hipMemcpyDeviceToHost));
            return GMX_OK;
          });
          if (rc) return rc;
        }
        if (tier) *tier = 2;
      } else {
        if (n > e->big.max_states) continue;
        HIP_TRY(hipMemcpy(st.data(), e->big.states + (size_t)s * e->big.max_states, n * sizeof(GmxFinalState), hipMemcpyDeviceToHost));
        const GmxPathNode *arena = e->big.arena + (size_t)s * e->big.max_path_nodes;
        w.put(n);
        for (uint32_t i = 0; i < n; ++i) {
          rc = debug_put_state(e, w, st[i], [&](uint32_t h, GmxPathNode &nd) {
            HIP_TRY(hipMemcpy(&nd, arena + h, sizeof(nd), hipMemcpyDeviceToHost));
            return GMX_OK;
          });
          if (rc) return rc;
        }
        if (tier) *tier = 1;
      }
      *n_words = w.n;
      return w.n > cap_words && out ? GMX_ECAP : GMX_OK;
    }
    if (in_slots) {  // searched by the large-capacity route, no state anywhere: unmapped
      w.put(0);
      if (tier) *tier = 1;
      *n_words = w.n;
      return GMX_OK;
    }
  }
  // the fast tier: finals[task * GMX_FAST_STATES ..], node k of the task at arena[k * stride + task]
  uint32_t packed = 0;
  HIP_TRY(hipMemcpy(&packed, e->d_n_final + task, 4, hipMemcpyDeviceToHost));
  const uint32_t n = std::min<uint32_t>(packed & 0xFFu, GMX_FAST_STATES);
  std::vector<GmxFinalState> st(n);
  if (n) HIP_TRY(hipMemcpy(st.data(), e->d_finals + (size_t)task * GMX_FAST_STATES, n * sizeof(GmxFinalState), hipMemcpyDeviceToHost));
  const GmxPathNode *arena = e->d_arena + task;
  w.put(n);
  for (uint32_t i = 0; i < n; ++i) {
    rc = debug_put_state(e, w, st[i], [&](uint32_t h, GmxPathNode &nd) {
      HIP_TRY(hipMemcpy(&nd, arena + h, sizeof(nd), hipMemcpyDeviceToHost));
      return GMX_OK;
    });
    if (rc) return rc;
  }
  if (tier) *tier = 0;
  *n_words = w.n;
  return w.n > cap_words && out ? GMX_ECAP : GMX_OK;
} GMX_GUARD_INT("gmx_debug_final_states")

int gmx_debug_search(gmx_engine *e, const uint8_t *read, uint32_t read_len, int from_seed_table, const uint32_t *states, uint64_t n_state_words,
                     uint32_t from, uint32_t stop, int lf_only, uint32_t *out, uint64_t cap_words, uint64_t *n_words) try {
  if (!e || !read || !n_words || read_len == 0 || (!from_seed_table && (!states || n_state_words == 0 || from > read_len || stop > from))) {
    gmx_set_error("gmx_debug_search: bad argument");
    return GMX_EINVAL;
  }
  for (uint32_t i = 0; i < read_len; ++i)
    if (read[i] < 1 || read[i] > 4) {
      gmx_set_error("gmx_debug_search: the read must hold bases 1..4");
      return GMX_EINVAL;
    }
  if (from_seed_table && read_len < e->dview.kmer_size) {
    gmx_set_error("gmx_debug_search: read shorter than the k-mer size");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  int rc = debug_isa(e);
  if (rc) return rc;
  const uint32_t n_pairs = (read_len + 31u) / 32u;
  GmxDebugPools p;
  if ((rc = debug_pools(e, p, (uint32_t)n_state_words, n_pairs))) return rc;
  std::vector<uint2> planes(n_pairs + 16, make_uint2(0, 0));
  for (uint32_t i = 0; i < read_len; ++i) {
    const uint32_t code = read[i] - 1u;
    planes[i >> 5].x |= (code & 1u) << (i & 31u);
    planes[i >> 5].y |= (code >> 1) << (i & 31u);
  }
  HIP_TRY(hipMemcpy(p.planes, planes.data(), planes.size() * sizeof(uint2), hipMemcpyHostToDevice));
  if (!from_seed_table) HIP_TRY(hipMemcpy(p.in, states, n_state_words * sizeof(uint32_t), hipMemcpyHostToDevice));
  BatchView b{};
  b.packed = p.planes;
  b.n_reads = 1;
  b.forward_only = 1;
  b.uniform_len = read_len;
  b.pairs_per_read = n_pairs;
  BigOut g{};
  g.states = p.states;
  g.stack = p.stack;
  g.arena = p.arena;
  g.max_states = p.cap_states;
  g.max_path_nodes = p.cap_nodes;
  g.max_slots = 1;
  const size_t big_lds = (size_t)GMX_BIG_LDS_DEPTH * GMX_STACK_WORDS * 64 * sizeof(uint32_t);
  hipLaunchKernelGGL(gmx_debug_search_kernel, dim3(1), dim3(64), big_lds, 0, e->dview, b, p.in, from, stop,
                     lf_only ? GMX_MODE_LF : GMX_MODE_STATE, from_seed_table ? 1u : 0u, g, p.result);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  uint32_t res[3] = {0, 0, 0};
  HIP_TRY(hipMemcpy(res, p.result, sizeof(res), hipMemcpyDeviceToHost));
  if (res[0] != GMX_TASK_MAPPED) {
    debug_pools_free(e, p);
    gmx_set_error(res[0] == GMX_TASK_OVERFLOW ? "gmx_debug_search: the states exceed the engine's large-capacity pools (gmx_engine_opts.max_states / max_path_nodes)"
                                              : "gmx_debug_search: the search reports a condition on which the reference throws");
    return res[0] == GMX_TASK_OVERFLOW ? GMX_ECAP : GMX_EREF;
  }
  const uint32_t n = res[1];
  std::vector<GmxFinalState> st(n);
  std::vector<GmxPathNode> arena(res[2]);
  if (n) HIP_TRY(hipMemcpy(st.data(), p.states, n * sizeof(GmxFinalState), hipMemcpyDeviceToHost));
  if (res[2]) HIP_TRY(hipMemcpy(arena.data(), p.arena, res[2] * sizeof(GmxPathNode), hipMemcpyDeviceToHost));
  debug_pools_free(e, p);
  DebugWriter w{out, cap_words};
  w.put(n);
  for (uint32_t i = 0; i < n; ++i) {
    rc = debug_put_state(e, w, st[i], [&](uint32_t h, GmxPathNode &nd) {
      if (h >= arena.size()) {
        gmx_set_error("gmx_debug_search: path handle outside the arena");
        return GMX_EREF;
      }
      nd = arena[h];
      return GMX_OK;
    });
    if (rc) return rc;
  }
  *n_words = w.n;
  return w.n > cap_words && out ? GMX_ECAP : GMX_OK;
} GMX_GUARD_INT("gmx_debug_search")

int gmx_debug_encapsulate(gmx_engine *e, const uint32_t *states, uint64_t n_state_words, uint32_t *out, uint64_t cap_words, uint64_t *n_words,
                          uint32_t *nonvariant_sa, uint64_t cap_nonvariant, uint64_t *n_nonvariant) try {
  if (!e || !states || n_state_words == 0 || !n_words || !n_nonvariant) {
    gmx_set_error("gmx_debug_encapsulate: bad argument");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  // the given states as the search kernels leave them: final-state records + a path arena (plain node indices)
  std::vector<GmxFinalState> st;
  std::vector<GmxPathNode> arena;
  const uint32_t *p = states, *end = states + n_state_words;
  const uint32_t ns = *p++;
  uint64_t positions = 0;
  for (uint32_t s = 0; s < ns; ++s) {
    if (p + 4 > end || p + 4 + 2 * (uint64_t)p[2] + p[3] > end) {
      gmx_set_error("gmx_debug_encapsulate: malformed state words");
      return GMX_EINVAL;
    }
    const uint32_t lo = p[0], hi = p[1], nt = p[2], ng = p[3];
    p += 4;
    if (lo > hi || hi >= e->dview.n) {
      gmx_set_error("gmx_debug_encapsulate: interval outside the suffix array");
      return GMX_EINVAL;
    }
    uint32_t tvd = GMX_NIL, tvg = GMX_NIL;
    for (uint32_t j = 0; j < nt; ++j, p += 2) {
      arena.push_back(GmxPathNode{p[0], (int32_t)p[1], tvd});
      tvd = (uint32_t)arena.size() - 1;
    }
    for (uint32_t j = 0; j < ng; ++j, ++p) {
      arena.push_back(GmxPathNode{p[0], -1, tvg});
      tvg = (uint32_t)arena.size() - 1;
    }
    st.push_back(GmxFinalState{lo, hi, tvd, tvg});
    positions += (uint64_t)hi - lo + 1;
  }
  if (positions > (1u << 24)) {
    gmx_set_error("gmx_debug_encapsulate: more than 2^24 positions");
    return GMX_EINVAL;
  }
  const uint32_t cap_items = (uint32_t)std::max<uint64_t>(positions + ns, 1);
  GmxFinalState *d_st = nullptr;
  uint32_t *d_out = nullptr;
  int rc;
  if ((rc = e->alloc(&d_st, st.size(), false))) return rc;
  if ((rc = e->alloc(&d_out, 2 + 7 * (size_t)cap_items, true))) return rc;
  if (!st.empty()) HIP_TRY(hipMemcpy(d_st, st.data(), st.size() * sizeof(GmxFinalState), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gmx_debug_items_kernel, dim3(1), dim3(64), 0, 0, e->dview, d_st, (uint32_t)st.size(), d_out, cap_items);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  std::vector<uint32_t> res(2 + 7 * (size_t)cap_items);
  HIP_TRY(hipMemcpy(res.data(), d_out, res.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  e->release(d_st);
  e->release(d_out);
  if (res[0] == 0xFFFFFFFFu || res[1] > cap_items) {
    gmx_set_error("gmx_debug_encapsulate: item buffer exceeded");
    return GMX_EREF;
  }
  DebugWriter w{out, cap_words};
  w.put(res[0]);
  for (uint32_t i = 0; i < res[0]; ++i) {
    const uint32_t *it = res.data() + 2 + 6 * (size_t)i;
    GmxFinalState f{it[0], it[1], it[2], it[3]};
    if (it[4] != 0) {  // a position inside an allele: the state [i, i] with that locus as its path
      w.put(f.lo);
      w.put(f.lo);
      w.put(1);
      w.put(0);
      w.put(it[4]);
      w.put(it[5]);
      continue;
    }
    // (debug_put_state handles text form; the given states are in SA form, so none arises)
    rc = debug_put_state(e, w, f, [&](uint32_t h, GmxPathNode &nd) {
      if (h >= arena.size()) {
        gmx_set_error("gmx_debug_encapsulate: path handle outside the arena");
        return GMX_EREF;
      }
      nd = arena[h];
      return GMX_OK;
    });
    if (rc) return rc;
  }
  *n_words = w.n;
  *n_nonvariant = res[1];
  for (uint32_t i = 0; i < res[1] && nonvariant_sa && i < cap_nonvariant; ++i) nonvariant_sa[i] = res[2 + 6 * (size_t)cap_items + i];
  return (w.n > cap_words && out) || (nonvariant_sa && res[1] > cap_nonvariant) ? GMX_ECAP : GMX_OK;
} GMX_GUARD_INT("gmx_debug_encapsulate")

}  // extern "C"
