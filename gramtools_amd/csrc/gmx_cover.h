// gmx_cover.h — per-(read, orientation) coverage recording on the flat index.
//
// What the reference does in coverage::record::search_states
// (libgramtools/src/genotype/quasimap/coverage/coverage_common.cpp:166-197) after
// handle_allele_encapsulated_states (search/encapsulated_search.cpp:90-107):
//   encapsulation split -> LocusFinder per state -> equivalence classes over level-0
//   sites -> one seeded uniform draw -> allele-sum, grouped allele counts and per-base
//   (hull per node) increments for the chosen class.
//
// Written as GMX_HD functions over an `Env` so that the HIP coverage kernel runs them
// one lane per task, and the test-only host build (tests/hostemu) can run the very
// same logic against the oracle without a GPU. The product library only ever calls
// them from kernels.
//
// Env interface:
//   uint32_t sget(uint32_t word) / void sset(uint32_t word, uint32_t v)   per-task scratch words
//   uint32_t i_max(), b_max(), loc_max(), h_max()                          capacities (compile-time constants for the fixed
//                                                                          tiers, runtime values for the last, heap-backed tier)
//   bool has_log_sites() / bool log_reserve(uint32_t words)                  grouped log of sites with more than 8 alleles: ALL
//                                                                          words a task will append are reserved at once, before
//                                                                          anything is recorded (a full log fails the task whole)
//   void add_allele_sum(uint32_t slot), add_per_base(uint32_t slot), add_grouped_dense(uint32_t slot): +1 on a slot of the
//        accumulator block (gmx_types.h: gmx_slot_*); add_allele_and_group(slot): +1 on slot and slot + 1 (one 64-bit add)
//   void add_hit(uint32_t slot)                                             hit counter of a one-base allele (gmx_types.h)
//   bool log_grouped_begin(uint32_t site_index, uint32_t n_ids) / void log_grouped_id(int32_t) / void log_grouped_end()
//   void fail(uint32_t status)
//   uint32_t h_site(h) / int32_t h_allele(h) / uint32_t h_next(h)            path-list handles (arena nodes, inline handles, or a
//                                                                            compact record's entries: gmx_engine.hip CompactEnv)
#pragma once
#include <type_traits>
#include "gmx_types.h"

#define GMX_RNG_LEMIRE 0    // libstdc++ >= 11 uniform_int_distribution (default: the toolchain of this image)
#define GMX_RNG_DIVISION 1  // libstdc++ <= 10

// ---------------------------------------------------------------------------
// std::mt19937 outputs #0.. without materialising the 624-word state: output j (< 227)
// depends only on the seeded words s[j], s[j+1], s[j+397] (random.hpp:14-25; the per-read
// generator is freshly seeded and almost always drawn once, coverage_common.cpp:169,102).
// ---------------------------------------------------------------------------
struct GmxMt {
  uint32_t a;   // s[j]
  uint32_t b;   // s[j + 397]
  uint32_t j;
};
GMX_HD uint32_t gmx_mt_next_seed_word(uint32_t prev, uint32_t i) { return 1812433253u * (prev ^ (prev >> 30)) + i; }
GMX_HD void gmx_mt_init(GmxMt &g, uint32_t seed) {
  g.a = seed;
  uint32_t x = seed;
  for (uint32_t i = 1; i <= 397; ++i) x = gmx_mt_next_seed_word(x, i);
  g.b = x;
  g.j = 0;
}
// returns false once the cheap window (227 outputs) is exhausted
GMX_HD bool gmx_mt_next(GmxMt &g, uint32_t &out) {
  if (g.j >= 226) return false;
  uint32_t a1 = gmx_mt_next_seed_word(g.a, g.j + 1);
  uint32_t y = (g.a & 0x80000000u) | (a1 & 0x7fffffffu);
  uint32_t v = g.b ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  v ^= (v >> 11);
  v ^= (v << 7) & 0x9d2c5680u;
  v ^= (v << 15) & 0xefc60000u;
  v ^= (v >> 18);
  out = v;
  g.a = a1;
  g.b = gmx_mt_next_seed_word(g.b, g.j + 398);
  g.j++;
  return true;
}
// std::uniform_int_distribution<uint32_t>(1, n)(mt19937(seed)), n >= 1 (random.cpp:15-18).
GMX_HD bool gmx_uniform_1_to_n(uint32_t seed, uint32_t n, int mode, uint32_t &result) {
  if (n == 1) {  // range of size one: the draw is consumed but cannot change the result
    result = 1;
    return true;
  }
  GmxMt g;
  gmx_mt_init(g, seed);
  uint32_t x;
  if (mode == GMX_RNG_LEMIRE) {  // bits/uniform_int_dist.h (_S_nd): multiply-shift with rejection
    if (!gmx_mt_next(g, x)) return false;
    uint64_t product = (uint64_t)x * (uint64_t)n;
    uint32_t low = (uint32_t)product;
    if (low < n) {
      uint32_t threshold = (0u - n) % n;
      while (low < threshold) {
        if (!gmx_mt_next(g, x)) return false;
        product = (uint64_t)x * (uint64_t)n;
        low = (uint32_t)product;
      }
    }
    result = (uint32_t)(product >> 32) + 1u;
    return true;
  }
  const uint32_t scaling = 0xffffffffu / n;
  const uint64_t past = (uint64_t)n * scaling;
  do {
    if (!gmx_mt_next(g, x)) return false;
  } while ((uint64_t)x >= past);
  result = x / scaling + 1u;
  return true;
}

// ---------------------------------------------------------------------------
// scratch layout (words)
// ---------------------------------------------------------------------------
#ifndef GMX_COVER_ROUTE  // test build: which routine recorded a single-instance task (tests/hostemu)
#define GMX_COVER_ROUTE(k) do { } while (0)
#endif
#ifndef GMX_COVER_PROF
#define GMX_COVER_PROF(env, k) do { } while (0)
#endif
#ifndef GMX_COVER_WHY  // debug build: which capacity a task exceeded (0 loci, 1 key sites, 2 hull, 3 items)
#define GMX_COVER_WHY(env, k) do { } while (0)
#endif
#define GMX_PATH_CACHE 8u  // (site, allele) pairs of an item's traversed list kept in scratch (gmx_item_loci) ...
// ... unless the Env says otherwise (Env::P_MAX): a read inside an MSA region of configs[2] has traversed 10-15 sites
template <class Env, class = void>
struct GmxPathMax {
  static constexpr uint32_t value = GMX_PATH_CACHE;
};
template <class Env>
struct GmxPathMax<Env, std::void_t<decltype(Env::P_MAX)>> {
  static constexpr uint32_t value = Env::P_MAX;
};
template <class Env>
struct GmxScratch {
  // item i: lo, hi, tvd, tvg, enc_site, enc_allele
  static constexpr uint32_t ITEM_W = 6;
  static constexpr uint32_t items = 0;
  GMX_HD static uint32_t order(const Env &e) { return items + e.i_max() * ITEM_W; }            // item indices sorted by key
  GMX_HD static uint32_t keys(const Env &e) { return order(e) + e.i_max(); }                  // per item: len, b_max sites
  GMX_HD static uint32_t loci(const Env &e) { return keys(e) + e.i_max() * (1 + e.b_max()); }  // (site, allele)
  GMX_HD static uint32_t hull(const Env &e) { return loci(e) + e.loc_max() * 2; }             // (node, start, end)
  GMX_HD static uint32_t path(const Env &e) { return hull(e) + e.h_max() * 3; }              // copy of an item's traversed list
  GMX_HD static uint32_t total_of(const Env &e) { return path(e) + 2 * GmxPathMax<Env>::value + 1; }  // + the copy's length
};
template <class Env>
struct GmxScratchFixed {  // the fixed tiers: Env::I_MAX .. are compile-time constants
  static constexpr uint32_t total = Env::I_MAX * (GmxScratch<Env>::ITEM_W + 1) + Env::I_MAX * (1 + Env::B_MAX) + Env::LOC_MAX * 2 + Env::H_MAX * 3 + 2 * GmxPathMax<Env>::value + 1;
};

GMX_HD bool gmx_in_bubble(const GmxNode &n) { return n.allele != -1 && n.site != 0; }
// A final state is an SA interval [lo, hi], or — text form (gmx_types.h, GmxTextRec) — the single PRG position lo.
GMX_HD bool gmx_text_form(uint32_t hi) { return hi == GMX_TEXT_MARK; }
GMX_HD uint32_t gmx_occ_pos(const GmxIndexView &ix, uint32_t hi, uint32_t i) { return gmx_text_form(hi) ? i : ix.sa[i]; }

// The loci of item `it` appended to loci[n_loci..] (a LocusFinder run whose sets are merged into the
// class's set, coverage_common.cpp:110-122). check_site_uniqueness (:17-32) is enforced.
template <class Env>
GMX_HD uint32_t gmx_item_loci(const GmxIndexView &ix, Env &env, uint32_t it, uint32_t n_loci_in) {
  typedef GmxScratch<Env> S;
  uint32_t base = S::items + it * S::ITEM_W;
  uint32_t lo = env.sget(base), hi = env.sget(base + 1), tvd = env.sget(base + 2), tvg = env.sget(base + 3);
  uint32_t enc_site = env.sget(base + 4);
  int32_t enc_allele = (int32_t)env.sget(base + 5);
  // A LocusFinder starts with empty used_sites: loci found by *this* item are searched from `first`.
  // The merge into the class set is a set union, so de-duplicating against earlier items' loci is
  // equivalent as long as the per-item `used_sites` short-circuit is evaluated on this item only.
  // We therefore run the finder on a private window [first, n) and merge afterwards.
  uint32_t first = n_loci_in;
  uint32_t n = n_loci_in;
  auto window_used = [&](uint32_t site) {
    for (uint32_t i = first; i < n; ++i)
      if (env.sget(S::loci(env) + 2 * i) == site) return true;
    return false;
  };
  auto window_add = [&](uint32_t site, int32_t allele) -> bool {
    for (uint32_t i = first; i < n; ++i)
      if (env.sget(S::loci(env) + 2 * i) == site && (int32_t)env.sget(S::loci(env) + 2 * i + 1) == allele) return true;
    if (n >= env.loc_max()) {
      GMX_COVER_WHY(env, 0);
      env.fail(GMX_TASK_OVERFLOW);
      return false;
    }
    env.sset(S::loci(env) + 2 * n, site);
    env.sset(S::loci(env) + 2 * n + 1, (uint32_t)allele);
    ++n;
    return true;
  };
  auto nested = [&](uint32_t site, int32_t allele) -> bool {
    for (;;) {
      if (window_used(site)) return true;
      if (!window_add(site, allele)) return false;
      if (!ix.is_nested) return true;  // every site is a level-0 site
      const GmxSite &s = ix.sites[(site - 5) >> 1];
      if (s.parent_site == 0) return true;
      allele = s.parent_allele;
      site = s.parent_site;
    }
  };
  if (enc_site != 0) {
    if (!nested(enc_site, enc_allele)) return 0xFFFFFFFFu;
  } else {
    // The traversed list is copied to scratch once (newest first, up to GMX_PATH_CACHE entries): every handle step is a
    // dependent load from the arena, and the checks and the oldest-first pass below would repeat them many times over.
    // Longer lists are walked by handle.
    const uint32_t pc = S::path(env);
    uint32_t nt = 0;
    bool cached = true;
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) {
      if (nt == GmxPathMax<Env>::value) {
        cached = false;
        break;
      }
      env.sset(pc + 2 * nt, env.h_site(x));
      env.sset(pc + 2 * nt + 1, (uint32_t)env.h_allele(x));
      ++nt;
    }
    env.sset(pc + 2 * GmxPathMax<Env>::value, cached ? nt : 0xFFFFFFFFu);  // (read by the cooperative instance's class phase)
    // check_site_uniqueness over traversed + traversing
    if (cached) {
      for (uint32_t i = 0; i < nt; ++i) {
        const uint32_t sx = env.sget(pc + 2 * i);
        for (uint32_t j = i + 1; j < nt; ++j)
          if (env.sget(pc + 2 * j) == sx) {
            env.fail(GMX_TASK_ERROR);
            return 0xFFFFFFFFu;
          }
      }
      for (uint32_t y = tvg; y != GMX_NIL; y = env.h_next(y)) {
        const uint32_t sy = env.h_site(y);
        for (uint32_t i = 0; i < nt; ++i)
          if (env.sget(pc + 2 * i) == sy) {
            env.fail(GMX_TASK_ERROR);
            return 0xFFFFFFFFu;
          }
      }
    } else {
      for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) {
        uint32_t sx = env.h_site(x);
        for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
          if (env.h_site(y) == sx) {
            env.fail(GMX_TASK_ERROR);
            return 0xFFFFFFFFu;
          }
        for (uint32_t y = tvg; y != GMX_NIL; y = env.h_next(y))
          if (env.h_site(y) == sx) {
            env.fail(GMX_TASK_ERROR);
            return 0xFFFFFFFFu;
          }
      }
    }
    for (uint32_t x = tvg; x != GMX_NIL; x = env.h_next(x)) {
      uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) {
          env.fail(GMX_TASK_ERROR);
          return 0xFFFFFFFFu;
        }
    }
    if (tvg != GMX_NIL) {  // assign_traversing_loci, coverage_common.cpp:53-76
      uint32_t parent_seed = env.h_site(tvg);
      int32_t last_allele = -1;
      for (uint32_t i = lo;; ++i) {
        uint32_t p = gmx_occ_pos(ix, hi, i);
        last_allele = ix.nodes[ix.pos_node[p]].allele;
        if (!window_add(parent_seed, last_allele)) return 0xFFFFFFFFu;
        if (gmx_text_form(hi) || i == hi) break;
      }
      // assign_nested_locus(new_locus): the seed site itself is not yet in used_sites (unique_loci and
      // used_sites are separate sets in the reference), so the walk continues with its parent chain.
      if (ix.is_nested) {
        const GmxSite &ps = ix.sites[(parent_seed - 5) >> 1];
        if (ps.parent_site != 0 && !nested(ps.parent_site, ps.parent_allele)) return 0xFFFFFFFFu;
      }
    }
    // assign_traversed_loci (:78-83): push order = oldest first; the list head is the newest.
    if (cached) {
      for (uint32_t d = nt; d-- > 0;)
        if (!nested(env.sget(pc + 2 * d), (int32_t)env.sget(pc + 2 * d + 1))) return 0xFFFFFFFFu;
    } else {  // by walking to each depth
      uint32_t len = 0;
      for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) ++len;
      for (uint32_t d = len; d-- > 0;) {
        uint32_t x = tvd;
        for (uint32_t s = 0; s < d; ++s) x = env.h_next(x);
        if (!nested(env.h_site(x), env.h_allele(x))) return 0xFFFFFFFFu;
      }
    }
  }
  return n;
}

// key of item (sorted level-0 sites of its loci window [first, n)) -> keys[it]
template <class Env>
GMX_HD bool gmx_item_key(const GmxIndexView &ix, Env &env, uint32_t it, uint32_t first, uint32_t n) {
  typedef GmxScratch<Env> S;
  uint32_t kb = S::keys(env) + it * (1 + env.b_max());
  uint32_t len = 0;
  for (uint32_t i = first; i < n; ++i) {
    uint32_t site = env.sget(S::loci(env) + 2 * i);
    if (ix.is_nested && ix.sites[(site - 5) >> 1].parent_site != 0) continue;
    // insertion sort, distinct
    uint32_t pos = 0;
    bool dup = false;
    while (pos < len) {
      uint32_t v = env.sget(kb + 1 + pos);
      if (v == site) {
        dup = true;
        break;
      }
      if (v > site) break;
      ++pos;
    }
    if (dup) continue;
    if (len >= env.b_max()) {
      GMX_COVER_WHY(env, 1);
      env.fail(GMX_TASK_OVERFLOW);
      return false;
    }
    for (uint32_t q = len; q > pos; --q) env.sset(kb + 1 + q, env.sget(kb + q));
    env.sset(kb + 1 + pos, site);
    ++len;
  }
  env.sset(kb, len);
  return true;
}
// lexicographic comparison of two keys (std::set<Marker> ordering inside std::map, coverage_common.hpp:133)
template <class Env>
GMX_HD int gmx_key_cmp(Env &env, uint32_t a, uint32_t b) {
  typedef GmxScratch<Env> S;
  uint32_t ka = S::keys(env) + a * (1 + env.b_max()), kb = S::keys(env) + b * (1 + env.b_max());
  uint32_t la = env.sget(ka), lb = env.sget(kb);
  uint32_t m = la < lb ? la : lb;
  for (uint32_t i = 0; i < m; ++i) {
    uint32_t va = env.sget(ka + 1 + i), vb = env.sget(kb + 1 + i);
    if (va != vb) return va < vb ? -1 : 1;
  }
  return la == lb ? 0 : (la < lb ? -1 : 1);
}

// ---------------------------------------------------------------------------
// Traverser on the flat graph (allele_base.cpp:137-219). The current node's record travels in
// registers: a single-edge hop costs one dependent load (GmxNode::edge0), a bubble entry two.
// ---------------------------------------------------------------------------
struct GmxWalk {
  uint32_t node;
  GmxNode rec;           // nodes[node]
  uint32_t remaining;
  uint32_t cursor;       // arena node of the next locus to consume (newest first); GMX_NIL = exhausted
  uint32_t enc_site;     // encapsulated item: single locus, consumed when enc_left
  int32_t enc_allele;
  bool enc_left;
  bool first;
  uint32_t start, end;
  bool bad;
  // how the current node was reached: by consuming a traversed locus (via = its handle; n_consumed counts them), as
  // the walk's first node (GMX_VIA_FIRST), or otherwise (GMX_VIA_OTHER)
  uint32_t via;
  uint32_t n_consumed;
};
#define GMX_NO_NODE 0xFFFFFFFFu
#define GMX_VIA_OTHER 0xFFFFFFFFu
#define GMX_VIA_FIRST 0xFFFFFFFEu

GMX_HD void gmx_walk_init(const GmxIndexView &ix, GmxWalk &w, uint32_t p, uint32_t node, const GmxNode &rec,
                          uint32_t read_len, uint32_t tvd, uint32_t enc_site, int32_t enc_allele) {
  w.node = node;
  w.rec = rec;
  w.remaining = read_len;
  w.cursor = tvd;
  w.enc_site = enc_site;
  w.enc_allele = enc_allele;
  w.enc_left = enc_site != 0;
  w.first = true;
  w.start = p - rec.first_pos;
  w.end = 0;
  w.bad = false;
  w.via = GMX_VIA_OTHER;
  w.n_consumed = 0;
}
GMX_HD void gmx_walk_update(GmxWalk &w) {  // update_coordinates :189-204
  uint32_t len = w.rec.seq_len;
  w.end = 0;
  if (len > 0) {
    if (w.remaining == 0) {  // never reached by the reference on valid mappings (would wrap a size_t)
      w.bad = true;
      return;
    }
    uint64_t e = (uint64_t)w.start + w.remaining - 1;
    w.end = e < (uint64_t)(len - 1) ? (uint32_t)e : len - 1;
    w.remaining -= (w.end - w.start + 1);
  }
}
template <class Env>
GMX_HD void gmx_walk_next_site(const GmxIndexView &ix, Env &env, GmxWalk &w) {  // go_to_next_site :168-187
  w.start = 0;
  w.via = GMX_VIA_OTHER;
  while (w.rec.n_edges == 1) {
    if (w.remaining == 0) {
      w.node = GMX_NO_NODE;
      return;
    }
    w.node = w.rec.edge0;
    w.rec = ix.nodes[w.node];
    gmx_walk_update(w);
    if (w.bad) return;
    if (gmx_in_bubble(w.rec)) return;
  }
  uint32_t ne = w.rec.n_edges;
  int32_t allele;
  if (w.enc_site != 0) {
    if (!w.enc_left) {
      w.bad = true;
      return;
    }
    allele = w.enc_allele;
    w.enc_left = false;
  } else {
    if (w.cursor == GMX_NIL) {
      w.bad = true;
      return;
    }
    allele = env.h_allele(w.cursor);
    w.via = w.cursor;
    w.n_consumed++;
    w.cursor = env.h_next(w.cursor);
  }
  if (allele < 0 || (uint32_t)allele >= ne) {
    w.bad = true;
    return;
  }
  w.node = allele == 0 ? w.rec.edge0 : ix.edges[w.rec.edge_begin + (uint32_t)allele];
  w.rec = ix.nodes[w.node];
  gmx_walk_update(w);
}
// next_Node :149-166. Returns GMX_NO_NODE at the end.
template <class Env>
GMX_HD uint32_t gmx_walk_next(const GmxIndexView &ix, Env &env, GmxWalk &w) {
  if (w.first) {
    w.first = false;
    gmx_walk_update(w);
    if (w.bad) return GMX_NO_NODE;
    w.via = GMX_VIA_FIRST;
    if (!gmx_in_bubble(w.rec)) gmx_walk_next_site(ix, env, w);
    if (w.node == GMX_NO_NODE) w.bad = true;  // the reference would dereference a null node here
    return w.bad ? GMX_NO_NODE : w.node;
  }
  if (w.remaining == 0) return GMX_NO_NODE;
  gmx_walk_next_site(ix, env, w);
  if (w.bad) return GMX_NO_NODE;
  return w.node;
}

// process_Node + DummyCovNode hull (allele_base.cpp:109-135,282-296)
template <class Env>
GMX_HD bool gmx_hull_add(Env &env, uint32_t &n_hull, uint32_t node, uint32_t seq_len, uint32_t s, uint32_t e) {
  typedef GmxScratch<Env> S;
  if (seq_len == 0) return true;
  for (uint32_t i = 0; i < n_hull; ++i) {
    if (env.sget(S::hull(env) + 3 * i) != node) continue;
    uint32_t hs = env.sget(S::hull(env) + 3 * i + 1), he = env.sget(S::hull(env) + 3 * i + 2);
    if (s < hs) env.sset(S::hull(env) + 3 * i + 1, s);
    if (e > he) env.sset(S::hull(env) + 3 * i + 2, e);
    return true;
  }
  if (n_hull >= env.h_max()) {
    GMX_COVER_WHY(env, 2);
    env.fail(GMX_TASK_OVERFLOW);
    return false;
  }
  env.sset(S::hull(env) + 3 * n_hull, node);
  env.sset(S::hull(env) + 3 * n_hull + 1, s);
  env.sset(S::hull(env) + 3 * n_hull + 2, e);
  ++n_hull;
  return true;
}

// PbCovRecorder::process_SearchState (allele_base.cpp:246-280) for item `it`
template <class Env>
GMX_HD bool gmx_item_per_base(const GmxIndexView &ix, Env &env, uint32_t it, uint32_t read_len, uint32_t &n_hull) {
  typedef GmxScratch<Env> S;
  uint32_t base = S::items + it * S::ITEM_W;
  uint32_t lo = env.sget(base), hi = env.sget(base + 1), tvd = env.sget(base + 2);
  uint32_t enc_site = env.sget(base + 4);
  int32_t enc_allele = (int32_t)env.sget(base + 5);
  bool first = true;
  for (uint32_t occ = lo;; ++occ) {
    uint32_t p = gmx_occ_pos(ix, hi, occ);
    GmxWalk w;
    {
      uint32_t node0 = ix.pos_node[p];
      gmx_walk_init(ix, w, p, node0, ix.nodes[node0], read_len, tvd, enc_site, enc_allele);
    }
    if (first) {
      first = false;
      for (;;) {
        uint32_t node = gmx_walk_next(ix, env, w);
        if (w.bad) {
          env.fail(GMX_TASK_ERROR);
          return false;
        }
        if (node == GMX_NO_NODE) break;
        if (!gmx_hull_add(env, n_hull, node, w.rec.seq_len, w.start, w.end)) return false;
      }
    } else {
      uint32_t node = gmx_walk_next(ix, env, w);
      if (w.bad || node == GMX_NO_NODE) {
        env.fail(GMX_TASK_ERROR);
        return false;
      }
      if (!gmx_hull_add(env, n_hull, node, w.rec.seq_len, w.start, w.end)) return false;
    }
    if (gmx_text_form(hi) || occ == hi) break;
  }
  return true;
}

// ---------------------------------------------------------------------------
// The common case, without scratch: ONE final state of interval width one (this routine: on a non-nested PRG).
// There is one item, hence one equivalence class and total == 1 (or only a non-variant
// instance): the draw cannot change the outcome, the loci are the state's own path plus the
// allele under SA[lo], and one DAG walk visits each node once, so the hull is the walk itself.
// Results are identical to the general routine below (tests/hostemu runs both).
// ---------------------------------------------------------------------------
template <class Env>
GMX_HD bool gmx_record_locus(const GmxIndexView &ix, Env &env, uint32_t site, int32_t allele) {
  const GmxSite &s = ix.sites[(site - 5) >> 1];
  if (allele < 0 || (uint32_t)allele >= s.n_alleles) {
    env.fail(GMX_TASK_ERROR);
    return false;
  }
  if (s.grouped_off != GMX_GROUPED_LOG) {
    env.add_allele_and_group(gmx_slot_allele(s, (uint32_t)allele));  // allele-sum and group {allele}: one 64-bit word
  } else {
    env.add_allele_sum(gmx_slot_allele(s, (uint32_t)allele));
    if (!env.log_grouped_begin((site - 5) >> 1, 1)) return false;
    env.log_grouped_id(allele);
    env.log_grouped_end();
  }
  return true;
}
// ---------------------------------------------------------------------------
// Recording a single-instance read on a flat PRG WITHOUT the walk, from the sites' geometry (GmxSiteGeo, gmx_types.h):
// every site of the path has a 32-byte record that says where it lies in the PRG, how long its alleles are and where
// their counters are, so the walk's bookkeeping — bases left, first and last base covered in each allele — is position
// arithmetic on ONE sector per site instead of ~5 dependent node / edge records per site crossed, and GmxSite is not
// looked at. The records of up to four loci are fetched side by side (the walk is one chain of dependent loads: a wave
// of these tasks waits for memory four fifths of its time).
// Nothing is recorded before the whole path has been checked. What the arithmetic cannot vouch for (a site without
// geometry, a gap that is not the stretch between two neighbouring sites, a read that ends early or runs on into another
// site) returns false with nothing recorded, and the caller takes the routine that decides as the reference does. The
// check pass leaves the increments it found as a short list of operations in `stage` (the kernel: LDS); a path with more
// of them than the stage holds is gone through a second time.
// ---------------------------------------------------------------------------
struct GmxStageNone {
  GMX_HD uint32_t cap() const { return 0u; }
  GMX_HD void put(uint32_t, uint32_t) {}
  GMX_HD uint32_t get(uint32_t) const { return 0u; }
};
struct GmxStageTest {
  uint32_t w[16], n;
  GMX_HD uint32_t cap() const { return n; }
  GMX_HD void put(uint32_t i, uint32_t v) { w[i] = v; }
  GMX_HD uint32_t get(uint32_t i) const { return w[i]; }
};
#ifndef GMX_JUMP_BATCH
#define GMX_JUMP_BATCH 4u  // geometry records fetched side by side
#endif
#define GMX_STAGE_MAX 16u  // operations a stage may hold (two bits of kind each in one register)
template <class Env, class Stage>
GMX_HD bool gmx_cover_jump(const GmxIndexView &ix, Env &env, Stage &stage, uint32_t p, uint32_t tvd, uint32_t tvg, uint32_t read_len) {
  // NO multi-word state crosses from the check pass to the recording pass: each pass derives everything from the scalars
  // (p, tvd, tvg, read_len) and fetches the node of the read's first base itself, when it matters (the read starts inside an
  // allele) — BEHIND its first four geometry loads: pos_node -> node -> its site's geometry is a chain of three dependent
  // loads that nearly every wave of 64 tasks has a lane for, and the loci's records need not wait for it.
  // Why (round 5, DESIGN.md §4, profiles/round5/jump_ptr_form_*): the round-4 form kept that node in a local captured by
  // reference, copied from the caller through a pointer and re-fetched under `if (!have)` inside the lambda. The compiler
  // inlines the lambda twice; behind the first body the node's fields meet in a ten-way phi (early returns), and the AMDGPU
  // back end lowered that join by parking three of the five live fields in temporaries copied under a narrower exec mask than
  // the one they are copied back under — the lanes in between entered the recording pass with first_pos = 3 and recorded
  // nothing: 16 914 increments instead of 22 196. The optimized IR is right (every edge defined), the machine code is not.
  const uint32_t cap = stage.cap() < GMX_STAGE_MAX ? stage.cap() : GMX_STAGE_MAX;
  uint32_t n_ops = 0, kinds = 0;  // staged operations: 0 hit counter, 1 allele-sum + group pair, 2 per-base range (slot), 3 its length
  auto pass = [&](const bool direct) -> bool {
    auto op = [&](uint32_t kind, uint32_t word) {
      if (n_ops < cap) {
        stage.put(n_ops, word);
        kinds |= kind << (2u * n_ops);
      }
      ++n_ops;
    };
    auto hit = [&](uint32_t slot) {
      if (direct) env.add_hit(slot);
      else op(0u, slot);
    };
    auto pair = [&](uint32_t slot) {
      if (direct) env.add_allele_and_group(slot);
      else op(1u, slot);
    };
    auto range = [&](uint32_t slot, uint32_t n) {
      if (direct) {
        for (uint32_t i = 0; i < n; ++i) env.add_per_base(slot + i);
      } else {
        op(2u, slot);
        op(3u, n);
      }
    };
    uint32_t remaining = read_len, cur = p, tail = 0xFFFFFFFFu;
    if (remaining == 0) return false;
    uint32_t x = tvd, m = 0, hx[GMX_JUMP_BATCH];
    GmxSiteGeo g[GMX_JUMP_BATCH];
    auto fetch = [&]() {  // the next GMX_JUMP_BATCH loci: their records are independent loads
      m = 0;
#pragma unroll
      for (uint32_t j = 0; j < GMX_JUMP_BATCH; ++j) {
        hx[j] = x;
        if (x != GMX_NIL) {
          ++m;
          x = env.h_next(x);
        }
      }
#pragma unroll
      for (uint32_t j = 0; j < GMX_JUMP_BATCH; ++j)
        if (j < m) g[j] = ix.site_geo[(env.h_site(hx[j]) - 5) >> 1];
    };
    fetch();
    if (tvg != GMX_NIL || tvd == GMX_NIL) {  // the read starts inside an allele: of the traversing site, or of the one it never leaves
      const GmxNode rec0 = ix.nodes[ix.pos_node[p]];
      if (tvg == GMX_NIL && rec0.site == 0) return true;  // a non-variant instance only: nothing to record
      if (!gmx_in_bubble(rec0) || rec0.seq_len == 0 || rec0.cov_off == GMX_NO_COV || p < rec0.first_pos || p - rec0.first_pos >= rec0.seq_len)
        return false;
      if (tvg != GMX_NIL && rec0.site != env.h_site(tvg)) return false;
      const GmxSiteGeo g0 = ix.site_geo[(rec0.site - 5) >> 1];
      if (!(g0.flags & GMX_SITE_JUMP) || (uint32_t)rec0.allele >= gmx_geo_alleles(g0)) return false;
      const uint32_t start = p - rec0.first_pos;
      const uint32_t n = remaining < rec0.seq_len - start ? remaining : rec0.seq_len - start;
      remaining -= n;
      if (gmx_node_has_hit_counter(rec0)) {
        hit(rec0.cov_off + 1);
      } else {
        range(rec0.cov_off + start, n);
        pair(g0.allele_sum_off + 2u * (uint32_t)rec0.allele);
      }
      if (tvg == GMX_NIL) return remaining == 0;  // (no path at all: the whole read inside the allele)
      cur = gmx_geo_exit_pos(g0) + 1u;
      tail = g0.tail_len;
    }
    auto locus = [&](const GmxSiteGeo &g, const uint32_t allele) -> bool {
      if (!(g.flags & GMX_SITE_JUMP) || allele >= gmx_geo_alleles(g) || g.entry_pos < cur) return false;
      const uint32_t gap = g.entry_pos - cur;  // base symbols in front of the site
      if (remaining <= gap || (tail != 0xFFFFFFFFu && gap != tail)) return false;
      remaining -= gap;
      const uint32_t kind = (g.flags >> (2u * allele)) & 3u;
      if (kind == GMX_ALLELE_HIT) {
        remaining -= 1u;
        hit(gmx_geo_cov_off(g, allele) + 1);
      } else {
        if (kind == GMX_ALLELE_LONG) {
          const uint32_t len = gmx_geo_allele_len(g, allele);
          const uint32_t n = remaining < len ? remaining : len;
          remaining -= n;
          range(gmx_geo_cov_off(g, allele), n);
        }
        pair(g.allele_sum_off + 2u * allele);
      }
      cur = gmx_geo_exit_pos(g) + 1u;
      tail = g.tail_len;
      return true;
    };
    while (m != 0) {
#pragma unroll
      for (uint32_t j = 0; j < GMX_JUMP_BATCH; ++j)
        if (j < m && !locus(g[j], (uint32_t)env.h_allele(hx[j]))) return false;
      fetch();
    }
    return tail != 0xFFFFFFFFu && remaining <= tail;  // (the rest of the read lies in the stretch behind the last site)
  };
  if (!pass(false)) return false;
  if (n_ops > cap) {  // more than the stage holds: once more, recording
    pass(true);
    return true;
  }
  for (uint32_t i = 0; i < n_ops; ++i) {
    const uint32_t kind = (kinds >> (2u * i)) & 3u, w = stage.get(i);
    if (kind == 0u) {
      env.add_hit(w);
    } else if (kind == 1u) {
      env.add_allele_and_group(w);
    } else {
      const uint32_t n = stage.get(++i);
      for (uint32_t k = 0; k < n; ++k) env.add_per_base(w + k);
    }
  }
  return true;
}

template <class Env>
GMX_HD void gmx_cover_single(const GmxIndexView &ix, Env &env, const GmxFinalState &st, uint32_t read_len) {
  const uint32_t tvd = st.traversed, tvg = st.traversing;
  const uint32_t p = gmx_occ_pos(ix, st.hi, st.lo);
  // The first node matters when the read starts inside an allele (traversing locus, or encapsulated: no path at all).
  const bool first_in_play = tvg != GMX_NIL || tvd == GMX_NIL;
  uint32_t node0 = 0;
  GmxNode rec0;
  if (first_in_play) {
    node0 = ix.pos_node[p];
    rec0 = ix.nodes[node0];
  }
  uint32_t enc_site = 0;
  int32_t enc_allele = -1;
  if (tvd == GMX_NIL && tvg == GMX_NIL) {
    if (rec0.site == 0) return;  // a non-variant instance only: nothing to record
    enc_site = rec0.site;
    enc_allele = rec0.allele;
  } else {  // check_site_uniqueness (coverage_common.cpp:17-32)
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) {
      uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR);
      for (uint32_t y = tvg; y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR);
    }
    for (uint32_t x = tvg; x != GMX_NIL; x = env.h_next(x)) {
      uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR);
    }
  }
  if (env.has_log_sites()) {  // reserve the task's words of the grouped log before anything is recorded
    uint32_t words = 0;
    const uint32_t fs = enc_site != 0 ? enc_site : tvg != GMX_NIL ? env.h_site(tvg) : 0u;
    if (fs != 0 && ix.sites[(fs - 5) >> 1].grouped_off == GMX_GROUPED_LOG) words += 3;
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x))
      if (ix.sites[(env.h_site(x) - 5) >> 1].grouped_off == GMX_GROUPED_LOG) words += 3;
    if (words && !env.log_reserve(words)) return;
  }
  // gmx_cover_jump first — host and device alike since round 5 (the general coverage instances reach this routine with
  // paths of more than 16 loci; in round 4 the device kept the walk here because of the wrong-result build explained at
  // gmx_cover_jump). What it declines takes the routines below, which decide as the reference does.
  {
#ifdef GMX_COVER_TEST_STAGE  // test build (tests/hostemu): the staged form, with the capacity the test asks for
    GmxStageTest none;
    none.n = GMX_COVER_TEST_STAGE;
#else
    GmxStageNone none;
#endif
    if (ix.site_geo && gmx_cover_jump(ix, env, none, p, tvd, tvg, read_len)) {
      GMX_COVER_ROUTE(1);
      return;
    }
  }
  // Without the walk: every traversed site is walk-free (gmx_types.h: a one-base allele is its hit counter, an empty
  // one its allele-sum/group pair) and the first node, if in play, has a hit counter. The site records are
  // independent loads; the walk below is a chain of dependent ones.
  {
    const uint32_t first_site = enc_site != 0 ? enc_site : tvg != GMX_NIL ? env.h_site(tvg) : 0u;
    bool walk_free = !first_in_play || (gmx_node_has_hit_counter(rec0) && rec0.site == first_site);
    for (uint32_t x = tvd; x != GMX_NIL && walk_free; x = env.h_next(x))
      walk_free = (ix.sites[(env.h_site(x) - 5) >> 1].snp_kinds & GMX_SITE_WALK_FREE) != 0;
    if (walk_free) {
      GMX_COVER_ROUTE(0);
      if (first_in_play) env.add_hit(rec0.cov_off + 1);
      for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) {
        const GmxSite &s = ix.sites[(env.h_site(x) - 5) >> 1];
        const int32_t allele = env.h_allele(x);
        if (allele < 0 || (uint32_t)allele >= s.n_alleles) return env.fail(GMX_TASK_ERROR);
        if (((s.snp_kinds >> (2 * allele)) & 3u) == GMX_ALLELE_HIT)
          env.add_hit(gmx_slot_hit(s, (uint32_t)allele));
        else
          env.add_allele_and_group(gmx_slot_allele(s, (uint32_t)allele));
      }
      return;
    }
  }
  GMX_COVER_ROUTE(2);
  if (!first_in_play) {
    node0 = ix.pos_node[p];
    rec0 = ix.nodes[node0];
  }
  // A locus whose allele node carries a hit counter (gmx_types.h) is recorded by that one counter during the walk;
  // `hit` has bit k set for the k-th traversed locus (newest first, the order the walk consumes them), bit 31 for the
  // locus of the first node.
  uint32_t hit = 0;
  const uint32_t first_site = enc_site != 0 ? enc_site : tvg != GMX_NIL ? env.h_site(tvg) : 0u;
  GmxWalk w;
  gmx_walk_init(ix, w, p, node0, rec0, read_len, tvd, enc_site, enc_allele);
  for (;;) {
    uint32_t node = gmx_walk_next(ix, env, w);
    if (w.bad) return env.fail(GMX_TASK_ERROR);
    if (node == GMX_NO_NODE) break;
    if (w.rec.seq_len == 0) continue;
    if (w.rec.cov_off == GMX_NO_COV) return env.fail(GMX_TASK_ERROR);
    if (gmx_node_has_hit_counter(w.rec)) {
      uint32_t bit = 0;
      if (w.via == GMX_VIA_FIRST)
        bit = w.rec.site == first_site ? 0x80000000u : 0u;
      else if (w.via != GMX_VIA_OTHER && w.n_consumed <= 31 && env.h_site(w.via) == w.rec.site)
        bit = 1u << (w.n_consumed - 1);
      if (bit) {
        hit |= bit;
        env.add_hit(w.rec.cov_off + 1);
        continue;
      }
    }
    for (uint32_t i = w.start; i <= w.end; ++i) env.add_per_base(w.rec.cov_off + i);
  }
  if (enc_site != 0) {
    if (!(hit >> 31)) gmx_record_locus(ix, env, enc_site, enc_allele);
    return;
  }
  if (tvg != GMX_NIL && !(hit >> 31) && !gmx_record_locus(ix, env, env.h_site(tvg), rec0.allele)) return;
  uint32_t k = 0;
  for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x), ++k)
    if (!(k < 31 && ((hit >> k) & 1u)) && !gmx_record_locus(ix, env, env.h_site(x), env.h_allele(x))) return;
}

// The same on a nested PRG. The item's loci are the sites of its path AND their ancestors (assign_nested_locus,
// coverage_common.cpp:34-51): every chain climbs until it meets a site the item already has, so each site appears
// once, with the allele of the first chain that reached it — one allele per site, hence the same 64-bit update per
// locus as above. The loci are gathered in registers BEFORE anything is recorded; more than GMX_SINGLE_LOCI of them
// returns false with nothing recorded and the task takes the general routine.
#define GMX_SINGLE_LOCI 8
template <class Env>
GMX_HD bool gmx_cover_single_nested(const GmxIndexView &ix, Env &env, const GmxFinalState &st, uint32_t read_len) {
  const uint32_t tvd = st.traversed, tvg = st.traversing;
  const uint32_t p = gmx_occ_pos(ix, st.hi, st.lo);
  const uint32_t node0 = ix.pos_node[p];
  const GmxNode rec0 = ix.nodes[node0];
  uint32_t l_site[GMX_SINGLE_LOCI];
  int32_t l_allele[GMX_SINGLE_LOCI];
  uint32_t n = 0;
  bool full = false;
  auto used = [&](uint32_t site) {
    bool hit = false;
#pragma unroll
    for (uint32_t i = 0; i < GMX_SINGLE_LOCI; ++i) hit = hit || (i < n && l_site[i] == site);
    return hit;
  };
  auto add = [&](uint32_t site, int32_t allele) {
    if (n >= env.single_loci()) {  // GMX_SINGLE_LOCI, or less in tests
      full = true;
      return;
    }
#pragma unroll
    for (uint32_t i = 0; i < GMX_SINGLE_LOCI; ++i)
      if (i == n) {
        l_site[i] = site;
        l_allele[i] = allele;
      }
    ++n;
  };
  auto climb = [&](uint32_t site, int32_t allele) {
    while (!full && !used(site)) {
      add(site, allele);
      const GmxSite &s = ix.sites[(site - 5) >> 1];
      if (s.parent_site == 0) break;
      allele = s.parent_allele;
      site = s.parent_site;
    }
  };
  uint32_t enc_site = 0;
  int32_t enc_allele = -1;
  if (tvd == GMX_NIL && tvg == GMX_NIL) {
    if (rec0.site == 0) return true;  // a non-variant instance only: nothing to record
    enc_site = rec0.site;
    enc_allele = rec0.allele;
    climb(enc_site, enc_allele);
  } else {
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) {  // check_site_uniqueness (coverage_common.cpp:17-32)
      uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR), true;
      for (uint32_t y = tvg; y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR), true;
    }
    for (uint32_t x = tvg; x != GMX_NIL; x = env.h_next(x)) {
      uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR), true;
    }
    if (tvg != GMX_NIL) {  // assign_traversing_loci (:53-76): the innermost site with the allele under p, then its ancestors
      const uint32_t seed_site = env.h_site(tvg);
      add(seed_site, rec0.allele);
      const GmxSite &ps = ix.sites[(seed_site - 5) >> 1];
      if (ps.parent_site != 0) climb(ps.parent_site, ps.parent_allele);
    }
    uint32_t len = 0;  // assign_traversed_loci (:78-83): oldest first; the list head is the newest
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x)) ++len;
    for (uint32_t d = len; d-- > 0;) {
      uint32_t x = tvd;
      for (uint32_t i = 0; i < d; ++i) x = env.h_next(x);
      climb(env.h_site(x), env.h_allele(x));
    }
  }
  if (full) return false;
  if (env.has_log_sites()) {  // reserve the task's words of the grouped log before anything is recorded
    uint32_t words = 0;
#pragma unroll
    for (uint32_t i = 0; i < GMX_SINGLE_LOCI; ++i)
      if (i < n && ix.sites[(l_site[i] - 5) >> 1].grouped_off == GMX_GROUPED_LOG) words += 3;
    if (words && !env.log_reserve(words)) return true;
  }
  uint32_t hit = 0;  // loci recorded by a hit counter during the walk (gmx_types.h)
  GmxWalk w;
  gmx_walk_init(ix, w, p, node0, rec0, read_len, tvd, enc_site, enc_allele);
  for (;;) {
    uint32_t node = gmx_walk_next(ix, env, w);
    if (w.bad) return env.fail(GMX_TASK_ERROR), true;
    if (node == GMX_NO_NODE) break;
    if (w.rec.seq_len == 0) continue;
    if (w.rec.cov_off == GMX_NO_COV) return env.fail(GMX_TASK_ERROR), true;
    if (gmx_node_has_hit_counter(w.rec)) {  // the node is its allele: its locus, if the item has it, is recorded here
      uint32_t bit = 0;
#pragma unroll
      for (uint32_t i = 0; i < GMX_SINGLE_LOCI; ++i)
        if (i < n && l_site[i] == w.rec.site && l_allele[i] == w.rec.allele) bit = 1u << i;
      if (bit & ~hit) {
        hit |= bit;
        env.add_hit(w.rec.cov_off + 1);
        continue;
      }
    }
    for (uint32_t i = w.start; i <= w.end; ++i) env.add_per_base(w.rec.cov_off + i);
  }
#pragma unroll
  for (uint32_t i = 0; i < GMX_SINGLE_LOCI; ++i)
    if (i < n && !((hit >> i) & 1u) && !gmx_record_locus(ix, env, l_site[i], l_allele[i])) return true;
  return true;
}

// The same for single-instance tasks with MORE loci than the register slots hold (a read through an MSA region crosses
// ten to fifteen nested sites): loci and a copy of the traversed list live in the env's per-lane scratch (LDS: sget/sset),
// at most GMX_WIDE_LOCI each. One item, hence one class and no draw that could change the outcome — none of the general
// routine's keys, sort, class search and merge. Returns false, nothing recorded, when the task does not fit.
//   scratch words: [0, 2 C) loci (site, allele) | [2 C, 4 C) traversed list, newest first (site, allele);  C = GMX_WIDE_LOCI
#define GMX_WIDE_LOCI 32u
template <class Env>
GMX_HD bool gmx_cover_single_nested_wide(const GmxIndexView &ix, Env &env, const GmxFinalState &st, uint32_t read_len) {
  constexpr uint32_t C = GMX_WIDE_LOCI, L0 = 0, P0 = 2 * C;
  const uint32_t tvd = st.traversed, tvg = st.traversing;
  const uint32_t p = gmx_occ_pos(ix, st.hi, st.lo);
  const uint32_t node0 = ix.pos_node[p];
  const GmxNode rec0 = ix.nodes[node0];
  uint32_t n = 0;
  bool full = false;
  auto used = [&](uint32_t site) {
    for (uint32_t i = 0; i < n; ++i)
      if (env.sget(L0 + 2 * i) == site) return true;
    return false;
  };
  auto climb = [&](uint32_t site, int32_t allele) {  // assign_nested_locus, coverage_common.cpp:34-51
    while (!full && !used(site)) {
      if (n >= C) {
        full = true;
        return;
      }
      env.sset(L0 + 2 * n, site);
      env.sset(L0 + 2 * n + 1, (uint32_t)allele);
      ++n;
      const GmxSite &s = ix.sites[(site - 5) >> 1];
      if (s.parent_site == 0) break;
      allele = s.parent_allele;
      site = s.parent_site;
    }
  };
  uint32_t enc_site = 0;
  int32_t enc_allele = -1;
  if (tvd == GMX_NIL && tvg == GMX_NIL) {
    if (rec0.site == 0) return true;  // a non-variant instance only: nothing to record
    enc_site = rec0.site;
    enc_allele = rec0.allele;
    climb(enc_site, enc_allele);
  } else {
    uint32_t nt = 0;  // the traversed list, copied once (every handle step is a dependent load from the arena)
    for (uint32_t x = tvd; x != GMX_NIL; x = env.h_next(x), ++nt) {
      if (nt >= C) return false;
      env.sset(P0 + 2 * nt, env.h_site(x));
      env.sset(P0 + 2 * nt + 1, (uint32_t)env.h_allele(x));
    }
    // check_site_uniqueness (coverage_common.cpp:17-32)
    for (uint32_t i = 0; i < nt; ++i) {
      const uint32_t sx = env.sget(P0 + 2 * i);
      for (uint32_t j = i + 1; j < nt; ++j)
        if (env.sget(P0 + 2 * j) == sx) return env.fail(GMX_TASK_ERROR), true;
      for (uint32_t y = tvg; y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR), true;
    }
    for (uint32_t x = tvg; x != GMX_NIL; x = env.h_next(x)) {
      const uint32_t sx = env.h_site(x);
      for (uint32_t y = env.h_next(x); y != GMX_NIL; y = env.h_next(y))
        if (env.h_site(y) == sx) return env.fail(GMX_TASK_ERROR), true;
    }
    if (tvg != GMX_NIL) {  // assign_traversing_loci (:53-76): the innermost site with the allele under p, then its ancestors
      const uint32_t seed_site = env.h_site(tvg);
      env.sset(L0, seed_site);
      env.sset(L0 + 1, (uint32_t)rec0.allele);
      n = 1;
      const GmxSite &ps = ix.sites[(seed_site - 5) >> 1];
      if (ps.parent_site != 0) climb(ps.parent_site, ps.parent_allele);
    }
    for (uint32_t d = nt; d-- > 0;)  // assign_traversed_loci (:78-83): oldest first; the copy's head is the newest
      climb(env.sget(P0 + 2 * d), (int32_t)env.sget(P0 + 2 * d + 1));
  }
  if (full) return false;
  if (env.has_log_sites()) {  // reserve the task's words of the grouped log before anything is recorded
    uint32_t words = 0;
    for (uint32_t i = 0; i < n; ++i)
      if (ix.sites[(env.sget(L0 + 2 * i) - 5) >> 1].grouped_off == GMX_GROUPED_LOG) words += 3;
    if (words && !env.log_reserve(words)) return true;
  }
  uint32_t hit = 0;  // loci recorded by a hit counter during the walk (gmx_types.h); n <= 32
  GmxWalk w;
  gmx_walk_init(ix, w, p, node0, rec0, read_len, tvd, enc_site, enc_allele);
  for (;;) {
    uint32_t node = gmx_walk_next(ix, env, w);
    if (w.bad) return env.fail(GMX_TASK_ERROR), true;
    if (node == GMX_NO_NODE) break;
    if (w.rec.seq_len == 0) continue;
    if (w.rec.cov_off == GMX_NO_COV) return env.fail(GMX_TASK_ERROR), true;
    if (gmx_node_has_hit_counter(w.rec)) {  // the node is its allele: its locus, if the item has it, is recorded here
      uint32_t bit = 0;
      for (uint32_t i = 0; i < n; ++i)
        if (env.sget(L0 + 2 * i) == w.rec.site && (int32_t)env.sget(L0 + 2 * i + 1) == w.rec.allele) bit = 1u << i;
      if (bit & ~hit) {
        hit |= bit;
        env.add_hit(w.rec.cov_off + 1);
        continue;
      }
    }
    for (uint32_t i = w.start; i <= w.end; ++i) env.add_per_base(w.rec.cov_off + i);
  }
  for (uint32_t i = 0; i < n; ++i)
    if (!((hit >> i) & 1u) && !gmx_record_locus(ix, env, env.sget(L0 + 2 * i), (int32_t)env.sget(L0 + 2 * i + 1))) return true;
  return true;
}

// One member item of the chosen class: its loci merged into the class's set [0, n_loci) (a set union: the order of the
// members does not matter), its per-base hull into [0, n_hull). False: a capacity was exceeded or an error found
// (env.status), nothing recorded.
template <class Env>
GMX_HD bool gmx_class_add_item(const GmxIndexView &ix, Env &env, uint32_t it, uint32_t read_len, uint32_t &n_loci, uint32_t &n_hull) {
  typedef GmxScratch<Env> S;
  uint32_t first = n_loci;
  uint32_t n = gmx_item_loci(ix, env, it, first);
  if (n == 0xFFFFFFFFu) return false;
  // merge window [first, n) into [0, first): drop duplicates
  uint32_t w = first;
  for (uint32_t i = first; i < n; ++i) {
    uint32_t site = env.sget(S::loci(env) + 2 * i), al = env.sget(S::loci(env) + 2 * i + 1);
    bool dup = false;
    for (uint32_t j = 0; j < first && !dup; ++j)
      dup = env.sget(S::loci(env) + 2 * j) == site && env.sget(S::loci(env) + 2 * j + 1) == al;
    if (dup) continue;
    env.sset(S::loci(env) + 2 * w, site);
    env.sset(S::loci(env) + 2 * w + 1, al);
    ++w;
  }
  n_loci = w;
  return gmx_item_per_base(ix, env, it, read_len, n_hull);
}

// The chosen class's loci and hull recorded (allele_base.cpp:230-244, allele_sum.cpp:31-43, grouped_allele_counts.cpp:17-49).
template <class Env>
GMX_HD void gmx_class_record(const GmxIndexView &ix, Env &env, uint32_t n_loci, uint32_t n_hull) {
  typedef GmxScratch<Env> S;
  // --- every capacity check lies behind us except the grouped log: reserve all of this task's words at once ---
  if (env.has_log_sites()) {
    uint32_t words = 0;
    for (uint32_t i = 0; i < n_loci; ++i) {
      const uint32_t site = env.sget(S::loci(env) + 2 * i);
      if (ix.sites[(site - 5) >> 1].grouped_off != GMX_GROUPED_LOG) continue;
      bool first_of_site = true;
      for (uint32_t j = 0; j < i && first_of_site; ++j) first_of_site = env.sget(S::loci(env) + 2 * j) != site;
      words += first_of_site ? 3u : 1u;  // [site, n_ids, id] + one word per further id
    }
    if (words && !env.log_reserve(words)) return;
  }
  // --- record (allele_base.cpp:230-244, allele_sum.cpp:31-43, grouped_allele_counts.cpp:17-49) ---
  // Two passes. First every table lookup and every check, the accumulator slots noted in scratch (the words of the item
  // records, orders and keys are dead by now); then the increments, back to back. Interleaved, each lookup waits for
  // the increments issued before it, and a device-scope atomic completes at the memory side: microseconds under load.
  const uint32_t pend = S::items, pend_cap = S::loci(env) - S::items;
  uint32_t n_as = 0, n_gr = 0;  // allele-sum slots from the front, grouped slots from the back
  auto flush = [&]() {
    for (uint32_t i = 0; i < n_as; ++i) env.add_allele_sum(env.sget(pend + i));
    for (uint32_t i = 0; i < n_gr; ++i) env.add_grouped_dense(env.sget(pend + pend_cap - 1 - i));
    n_as = n_gr = 0;
  };
  for (uint32_t h = 0; h < n_hull; ++h) {
    const uint32_t off = ix.nodes[env.sget(S::hull(env) + 3 * h)].cov_off;
    if (off == GMX_NO_COV) {
      env.fail(GMX_TASK_ERROR);
      return;
    }
    env.sset(S::hull(env) + 3 * h, off);  // the entry is now (first slot of the node, start, end)
  }
  for (uint32_t i = 0; i < n_loci; ++i) {
    uint32_t site = env.sget(S::loci(env) + 2 * i);
    int32_t allele = (int32_t)env.sget(S::loci(env) + 2 * i + 1);
    const GmxSite &s = ix.sites[(site - 5) >> 1];
    if (allele < 0 || (uint32_t)allele >= s.n_alleles) {
      env.fail(GMX_TASK_ERROR);
      return;
    }
    if (n_as + n_gr == pend_cap) flush();
    env.sset(pend + n_as++, gmx_slot_allele(s, (uint32_t)allele));
  }
  for (uint32_t i = 0; i < n_loci; ++i) {
    uint32_t site = env.sget(S::loci(env) + 2 * i);
    bool first_of_site = true;
    for (uint32_t j = 0; j < i && first_of_site; ++j) first_of_site = env.sget(S::loci(env) + 2 * j) != site;
    if (!first_of_site) continue;
    const GmxSite &s = ix.sites[(site - 5) >> 1];
    if (s.grouped_off != GMX_GROUPED_LOG) {
      uint32_t mask = 0;
      for (uint32_t j = i; j < n_loci; ++j)
        if (env.sget(S::loci(env) + 2 * j) == site) mask |= 1u << env.sget(S::loci(env) + 2 * j + 1);
      if (n_as + n_gr == pend_cap) flush();
      env.sset(pend + pend_cap - 1 - n_gr++, gmx_slot_grouped(s, mask));
    } else {
      uint32_t cnt = 0;
      for (uint32_t j = i; j < n_loci; ++j)
        if (env.sget(S::loci(env) + 2 * j) == site) ++cnt;
      if (!env.log_grouped_begin((site - 5) >> 1, cnt)) return;
      // ascending allele ids (std::set<AlleleId> order, grouped_allele_counts.cpp:25-37)
      int32_t prev = -1;
      for (uint32_t k = 0; k < cnt; ++k) {
        int32_t best = 0x7fffffff;
        for (uint32_t j = i; j < n_loci; ++j)
          if (env.sget(S::loci(env) + 2 * j) == site) {
            int32_t a = (int32_t)env.sget(S::loci(env) + 2 * j + 1);
            if (a > prev && a < best) best = a;
          }
        env.log_grouped_id(best);
        prev = best;
      }
      env.log_grouped_end();
    }
  }
  for (uint32_t h = 0; h < n_hull; ++h) {
    const uint32_t off = env.sget(S::hull(env) + 3 * h), s = env.sget(S::hull(env) + 3 * h + 1), e = env.sget(S::hull(env) + 3 * h + 2);
    for (uint32_t i = s; i <= e; ++i) env.add_per_base(off + i);
  }
  flush();
}

// ---------------------------------------------------------------------------
// The items of a task's final states (handle_allele_encapsulated_states, encapsulated_search.cpp:30-107): a path-bearing
// state as it is; a path-less one position by position — inside an allele it becomes a state of its own with that locus
// as its path (add_item(lo, hi, traversed, traversing, site, allele)), outside every site it only counts
// (nonvariant(position index): count_nonvar_search_states, coverage_common.cpp:130-141). The reference merges neighbouring
// positions of one allele into an interval; position by position is the same set of mapping instances.
// false: add_item refused (scratch full). Used by gmx_cover_task and by the test hook gmx_debug_encapsulate.
// ---------------------------------------------------------------------------
template <class AddItem, class NonVariant>
GMX_HD bool gmx_final_items(const GmxIndexView &ix, const GmxFinalState *finals, uint32_t n_final, AddItem add_item, NonVariant nonvariant) {
  for (uint32_t f = 0; f < n_final; ++f) {
    GmxFinalState st = finals[f];
    if (st.traversed != GMX_NIL || st.traversing != GMX_NIL) {
      if (!add_item(st.lo, st.hi, st.traversed, st.traversing, 0, -1)) return false;
      continue;
    }
    for (uint32_t i = st.lo;; ++i) {
      const GmxNode &nd = ix.nodes[ix.pos_node[gmx_occ_pos(ix, st.hi, i)]];
      if (nd.site == 0)
        nonvariant(i);
      else if (!add_item(i, gmx_text_form(st.hi) ? st.hi : i, GMX_NIL, GMX_NIL, nd.site, nd.allele))
        return false;
      if (gmx_text_form(st.hi) || i == st.hi) break;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------
// The whole recording step for one mapped task.
// ---------------------------------------------------------------------------
template <class Env>
GMX_HD void gmx_cover_task(const GmxIndexView &ix, Env &env, const GmxFinalState *finals, uint32_t n_final,
                           uint32_t read_len, uint32_t seed, int rng_mode) {
  typedef GmxScratch<Env> S;
  if (n_final == 1 && (finals[0].lo == finals[0].hi || gmx_text_form(finals[0].hi))) {
    if (!ix.is_nested) {
      gmx_cover_single(ix, env, finals[0], read_len);
      return;
    }
    if (gmx_cover_single_nested(ix, env, finals[0], read_len)) return;
  }
  // --- items: path-bearing states + allele-encapsulated positions (encapsulated_search.cpp:30-107) ---
  uint32_t n_items = 0;
  uint32_t nonvariant = 0;  // count_nonvar_search_states, coverage_common.cpp:130-141 (uint32 arithmetic)
  auto add_item = [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg, uint32_t es, int32_t ea) -> bool {
    if (n_items >= env.i_max()) {
      GMX_COVER_WHY(env, 3);
      env.fail(GMX_TASK_OVERFLOW);
      return false;
    }
    uint32_t b = S::items + n_items * S::ITEM_W;
    env.sset(b, lo);
    env.sset(b + 1, hi);
    env.sset(b + 2, tvd);
    env.sset(b + 3, tvg);
    env.sset(b + 4, es);
    env.sset(b + 5, (uint32_t)ea);
    ++n_items;
    return true;
  };
  if (!gmx_final_items(ix, finals, n_final, add_item, [&](uint32_t) { nonvariant += 1; })) return;
  GMX_COVER_PROF(env, 0);
  if (n_items == 0) return;  // usps.size() == 0: nothing recorded, no draw (coverage_common.cpp:96-97)

  // --- class keys ---
  for (uint32_t it = 0; it < n_items; ++it) {
    uint32_t n = gmx_item_loci(ix, env, it, 0);
    if (n == 0xFFFFFFFFu) return;
    if (!gmx_item_key(ix, env, it, 0, n)) return;
  }
  GMX_COVER_PROF(env, 1);
  // Classes = runs of equal keys among the items sorted by key (the reference's std::map over level-0 site sets,
  // coverage_common.hpp:133): heap sort of the item indices, O(n log n) key comparisons — a read inside a many-copy
  // repeat has thousands of items, and every comparison is a chain of scratch loads.
  const uint32_t ord = S::order(env);
  for (uint32_t i = 0; i < n_items; ++i) env.sset(ord + i, i);
  if (n_items > 1) {
    auto sift = [&](uint32_t root, uint32_t end) {  // max-heap on [0, end)
      const uint32_t moving = env.sget(ord + root);
      for (;;) {
        uint32_t child = 2 * root + 1;
        if (child >= end) break;
        uint32_t cv = env.sget(ord + child);
        if (child + 1 < end) {
          const uint32_t rv = env.sget(ord + child + 1);
          if (gmx_key_cmp(env, cv, rv) < 0) {
            ++child;
            cv = rv;
          }
        }
        if (gmx_key_cmp(env, moving, cv) >= 0) break;
        env.sset(ord + root, cv);
        root = child;
      }
      env.sset(ord + root, moving);
    };
    for (uint32_t i = n_items / 2; i-- > 0;) sift(i, n_items);
    for (uint32_t end = n_items - 1; end > 0; --end) {
      const uint32_t top = env.sget(ord), last = env.sget(ord + end);
      env.sset(ord + end, top);
      env.sset(ord, last);
      sift(0, end);
    }
  }
  uint32_t n_classes = 1;
  for (uint32_t i = 1; i < n_items; ++i)
    if (gmx_key_cmp(env, env.sget(ord + i - 1), env.sget(ord + i)) != 0) ++n_classes;
  // --- selection (random_select_entry, coverage_common.cpp:95-108) ---
  uint32_t total = nonvariant + n_classes;
  uint32_t r;
  if (!gmx_uniform_1_to_n(seed, total, rng_mode, r)) {
    env.fail(GMX_TASK_ERROR);
    return;
  }
  GMX_COVER_PROF(env, 2);
  if (r <= nonvariant) return;
  const uint32_t want = r - nonvariant - 1;  // 0-based index in the ordered map
  uint32_t run_begin = 0, run_end = n_items;  // the want-th run of equal keys
  {
    uint32_t cls = 0;
    for (uint32_t i = 1; i < n_items; ++i)
      if (gmx_key_cmp(env, env.sget(ord + i - 1), env.sget(ord + i)) != 0) {
        ++cls;
        if (cls == want) run_begin = i;
        if (cls == want + 1) {
          run_end = i;
          break;
        }
      }
    if (want >= n_classes) {
      env.fail(GMX_TASK_ERROR);
      return;
    }
  }
  // --- loci of the class (union) + per-base hull ---
  uint32_t n_loci = 0, n_hull = 0;
  for (uint32_t ri = run_begin; ri < run_end; ++ri)
    if (!gmx_class_add_item(ix, env, env.sget(ord + ri), read_len, n_loci, n_hull)) return;
  GMX_COVER_PROF(env, 3);
  gmx_class_record(ix, env, n_loci, n_hull);
}
