// gmx_capi.cpp — host half of the C ABI: index construction, introspection, seeds, u16 finalisation.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <system_error>
#include <string>
#include <thread>
#include <vector>

#include "gmx_core.h"
#include "gmx_internal.h"

static std::atomic<uint64_t> g_index_serial{1};
struct gmx_index {
  gmx::HostIndex h;
  const uint64_t serial = g_index_serial.fetch_add(1);  // identity within the process (gmx_engine.hip shares device copies by it: an address can come back)
};

static thread_local std::string g_error;
void gmx_set_error(const std::string &msg) noexcept {
  try {
    g_error = msg;
  } catch (...) {
    g_error.clear();  // (clear() keeps the capacity and cannot throw)
  }
}

int gmx_guard_catch(const char *fn) noexcept {
  auto set = [&](const char *what) noexcept {  // "<function>: <what>" without a temporary that could throw again
    try {
      g_error.assign(fn);
      g_error.append(": ");
      g_error.append(what);
    } catch (...) {
      g_error.clear();
    }
  };
  try {
    throw;
  } catch (std::bad_alloc const &) {
    set("out of host memory");
    return GMX_ENOMEM;
  } catch (std::system_error const &e) {  // std::thread could not start a thread (EAGAIN)
    set(e.what());
    return GMX_ENOMEM;
  } catch (std::exception const &e) {
    set(e.what());
    return GMX_EINVAL;
  } catch (...) {
    set("unknown C++ exception");
    return GMX_EINVAL;
  }
}

// ---- test hook: make the library's n-th host allocation fail -------------------------------------------------------------
// GMX_TEST_FAIL_ALLOC=n in the environment, or gmx_debug_fail_alloc(n) (tests/test_alloc_failure.py): the n-th call of
// operator new made BY THIS LIBRARY from now on throws std::bad_alloc (n = 0: off; every later one succeeds again). The
// replacement functions are LOCAL symbols of libgmx.so (version script libgmx.map): they serve the library's own translation units only — the HIP runtime, RCCL,
// libstdc++'s own code and the host program keep theirs (an exception thrown into the runtime's frames would prove nothing) —
// and both sides end in malloc / free, so memory may cross. Cost when off: one relaxed load of a read-only flag per allocation.
static std::atomic<int64_t> g_fail_alloc{[] {
  const char *e = getenv("GMX_TEST_FAIL_ALLOC");
  return e ? (int64_t)atoll(e) : (int64_t)0;
}()};
static std::atomic<uint64_t> g_alloc_calls{0};
// Counting is ON only in a process that uses the hook (the environment variable, or a first gmx_debug_fail_alloc call): a counter
// every thread increments on every allocation is one contended cache line — the index builder's 64 threads allocate per k-mer, and
// with the count unconditional the whole-genome build took 471 s instead of 262 (round 6, found by the configs[4] test's time).
static std::atomic<bool> g_alloc_counting{getenv("GMX_TEST_FAIL_ALLOC") != nullptr};
static inline void *gmx_new_impl(size_t n, size_t align) {
  if (g_alloc_counting.load(std::memory_order_relaxed)) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    if (g_fail_alloc.load(std::memory_order_relaxed) > 0 && g_fail_alloc.fetch_sub(1, std::memory_order_relaxed) == 1) throw std::bad_alloc();
  }
  void *p = align > alignof(max_align_t) ? aligned_alloc(align, (n + align - 1) / align * align) : malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new(size_t n) { return gmx_new_impl(n, 0); }
void *operator new[](size_t n) { return gmx_new_impl(n, 0); }
void *operator new(size_t n, std::align_val_t a) { return gmx_new_impl(n, (size_t)a); }
void *operator new[](size_t n, std::align_val_t a) { return gmx_new_impl(n, (size_t)a); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, std::align_val_t) noexcept { free(p); }
void operator delete[](void *p, std::align_val_t) noexcept { free(p); }
void operator delete(void *p, size_t, std::align_val_t) noexcept { free(p); }
void operator delete[](void *p, size_t, std::align_val_t) noexcept { free(p); }
extern "C" uint64_t gmx_debug_fail_alloc(int64_t nth) {  // returns the library's allocation count so far
  // nth > 0: arm (and count); nth < 0: count only; nth == 0: disarm and STOP counting — a test session that has used the hook must
  // not leave the contended counter on for the tests behind it (the full-size configs[4] test runs last: 627 s instead of 310)
  g_alloc_counting.store(nth != 0, std::memory_order_relaxed);
  g_fail_alloc.store(nth > 0 ? nth : 0, std::memory_order_relaxed);
  return g_alloc_calls.load(std::memory_order_relaxed);
}
const gmx::HostIndex &gmx_index_host(const gmx_index *ix) { return ix->h; }
uint64_t gmx_index_serial(const gmx_index *ix) { return ix->serial; }

namespace {
// host context used only by gmx_index_jump_states (introspection of the pre-resolved jump programs)
struct ProbeCtx {
  struct St {
    uint32_t lo, hi, tvd, tvg;
  };
  std::vector<St> st;
  uint32_t n = 0;
  std::vector<GmxPathNode> arena;
  uint32_t status = GMX_TASK_MAPPED;
  uint32_t n_states() const { return n; }
  void set_n_states(uint32_t v) { n = v; }
  void get(uint32_t s, uint32_t &lo, uint32_t &hi, uint32_t &tvd, uint32_t &tvg) const {
    lo = st[s].lo; hi = st[s].hi; tvd = st[s].tvd; tvg = st[s].tvg;
  }
  void put(uint32_t s, uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) { st[s] = St{lo, hi, tvd, tvg}; }
  bool push(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
    if (n == st.size()) st.resize(st.size() ? st.size() * 2 : 16);
    st[n++] = St{lo, hi, tvd, tvg};
    return true;
  }
  uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) {
    arena.push_back(GmxPathNode{site, allele, next});
    return (uint32_t)arena.size() - 1;
  }
  uint32_t arena_site(uint32_t node) const { return arena[node].site; }
  uint32_t arena_next(uint32_t node) const { return arena[node].next; }
  void fail(uint32_t s) { status = s; }
};
}  // namespace

extern "C" {

const char *gmx_last_error(void) { return g_error.c_str(); }

// the symbols are handed over, not copied (12.4 GB at whole-genome scale; round 3 held them three times)
static int index_build_owned(std::vector<uint32_t> &&v, uint32_t kmer_size, int threads, gmx_index **out) {
  try {
    gmx_index *ix = new gmx_index();
    try {
      int k2 = -1;  // GMX_SEED_K2 in the environment: 0 disables the longer seed table, n forces its length
      if (const char *e = getenv("GMX_SEED_K2")) k2 = atoi(e);
      gmx::build_index(std::move(v), kmer_size, ix->h, threads, k2);
    } catch (...) {
      delete ix;
      throw;
    }
    *out = ix;
    return GMX_OK;
  } catch (std::bad_alloc const &) {
    gmx_set_error("out of memory while building the index");
    return GMX_ENOMEM;
  } catch (std::exception const &e) {
    gmx_set_error(e.what());
    return GMX_EINVAL;
  }
}

int gmx_index_build(const uint32_t *prg, uint64_t n, uint32_t kmer_size, int threads, gmx_index **out) try {
  if (!prg || !out) {
    gmx_set_error("gmx_index_build: null argument");
    return GMX_EINVAL;
  }
  try {
    return index_build_owned(std::vector<uint32_t>(prg, prg + n), kmer_size, threads, out);
  } catch (std::bad_alloc const &) {
    gmx_set_error("out of memory while building the index");
    return GMX_ENOMEM;
  }
} GMX_GUARD_INT("gmx_index_build")

int gmx_index_build_from_file(const char *path, uint32_t kmer_size, int threads, gmx_index **out) try {
  if (!path || !out) {
    gmx_set_error("gmx_index_build_from_file: null argument");
    return GMX_EINVAL;
  }
  try {
    return index_build_owned(gmx::read_prg_file(path), kmer_size, threads, out);
  } catch (std::bad_alloc const &) {
    gmx_set_error("out of memory while reading the PRG");
    return GMX_ENOMEM;
  } catch (std::exception const &e) {
    gmx_set_error(e.what());
    return GMX_EINVAL;
  }
} GMX_GUARD_INT("gmx_index_build_from_file")

int gmx_index_save(const gmx_index *ix, const char *path) try {
  if (!ix || !path) {
    gmx_set_error("gmx_index_save: null argument");
    return GMX_EINVAL;
  }
  try {
    gmx::save_index(ix->h, path);
    return GMX_OK;
  } catch (std::exception const &e) {
    gmx_set_error(e.what());
    return GMX_EINVAL;
  }
} GMX_GUARD_INT("gmx_index_save")

int gmx_index_load(const char *cache_path, const char *prg_path, uint32_t kmer_size, gmx_index **out) try {
  if (!cache_path || !prg_path || !out) {
    gmx_set_error("gmx_index_load: null argument");
    return GMX_EINVAL;
  }
  try {
    auto prg = gmx::read_prg_file(prg_path);
    gmx_index *ix = new gmx_index();
    try {
      gmx::load_index(cache_path, prg, kmer_size, ix->h);
    } catch (...) {
      delete ix;
      throw;
    }
    *out = ix;
    return GMX_OK;
  } catch (std::bad_alloc const &) {
    gmx_set_error("out of memory while loading the index cache");
    return GMX_ENOMEM;
  } catch (std::exception const &e) {
    gmx_set_error(e.what());
    return GMX_EINVAL;
  }
} GMX_GUARD_INT("gmx_index_load")

void gmx_index_destroy(gmx_index *ix) { delete ix; }

int gmx_index_get_info(const gmx_index *ix, gmx_index_info *o) try {
  const gmx::HostIndex &h = ix->h;
  memset(o, 0, sizeof(*o));
  o->n_text = h.prg.size() + 1;
  o->kmer_size = h.kmer_size;
  o->n_sites = (uint32_t)h.sites.size();
  o->is_nested = h.is_nested ? 1 : 0;
  o->n_allele_slots = h.n_allele_slots;
  o->n_per_base_slots = h.n_pb_slots;
  o->n_grouped_slots = h.n_grouped_slots;
  o->n_nodes = h.nodes.empty() ? 0 : (uint32_t)h.nodes.size() - 1;
  o->n_kmers_present = h.n_seed_kmers_present;
  o->kmer_size2 = h.kmer_size2;
  o->seed_shift = h.seed_shift;
  o->n_jump_sites = 0;
  for (const GmxSiteGeo &g : h.site_geo) o->n_jump_sites += (g.flags & GMX_SITE_JUMP) ? 1u : 0u;
  o->n_seed_words = h.seed_words.size();
  {
    uint64_t n_inline = 0;
    for (const GmxTextRec &r : h.text) n_inline += (uint64_t)__builtin_popcountll(r.mk & r.hi);
    o->n_inline_sites = (uint32_t)n_inline;
  }
  o->index_bytes = h.blocks.size() * sizeof(GmxRankBlock) + h.hits.size() * sizeof(GmxHit) + h.text.size() * sizeof(GmxTextRec) + (h.hit_perm.size() + h.hit_prog.size() + h.prog.size() + h.sa.size() + h.pos_node.size() +
                   h.edges.size() + h.seed_words.size() + h.kmer_bitmap.size()) * 4 + h.seeds2.size() * sizeof(GmxSeed) + h.nodes.size() * sizeof(GmxNode) +
                   h.sites.size() * (sizeof(GmxSite) + sizeof(GmxSiteGeo)) + h.seeds.size() * sizeof(GmxSeed);
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_get_info")

int gmx_index_site_layout(const gmx_index *ix, uint32_t *n_alleles, uint32_t *allele_sum_off, uint32_t *grouped_off,
                          uint32_t *parent_site, int32_t *parent_allele) try {
  const auto &s = ix->h.sites;
  for (size_t i = 0; i < s.size(); ++i) {
    if (n_alleles) n_alleles[i] = s[i].n_alleles;
    if (allele_sum_off) allele_sum_off[i] = ix->h.l_allele_off[i];
    if (grouped_off) grouped_off[i] = ix->h.l_grouped_off[i];
    if (parent_site) parent_site[i] = s[i].parent_site;
    if (parent_allele) parent_allele[i] = s[i].parent_allele;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_site_layout")

int64_t gmx_index_per_base_layout(const gmx_index *ix, uint32_t *out, uint64_t cap) try {
  const auto &h = ix->h;
  uint64_t n = 0;
  for (size_t i = 0; i + 1 < h.nodes.size(); ++i) {
    const GmxNode &nd = h.nodes[i];
    if (nd.cov_off == GMX_NO_COV) continue;
    if (out && n < cap) {
      const uint32_t l_cov = h.l_cov_off[i];
      out[5 * n + 0] = (nd.site - 5) / 2;
      out[5 * n + 1] = (uint32_t)nd.allele;
      out[5 * n + 2] = nd.first_pos;
      out[5 * n + 3] = l_cov;
      out[5 * n + 4] = nd.seq_len;
    }
    ++n;
  }
  return (int64_t)n;
} GMX_GUARD_INT("gmx_index_per_base_layout")

// allele_base_non_nested (allele_base.cpp:10-38): one slice per (site, allele); direct deletions have length 0.
int gmx_index_allele_base_layout(const gmx_index *ix, uint32_t *pb_off, uint32_t *len) try {
  const auto &h = ix->h;
  if (h.is_nested) {
    gmx_set_error("allele_base_non_nested is empty by convention for nested PRGs");
    return GMX_EINVAL;
  }
  for (size_t s = 0; s < h.sites.size(); ++s) {
    const GmxSite &site = h.sites[s];
    const GmxNode &entry = h.nodes[site.entry_node];
    for (uint32_t a = 0; a < site.n_alleles; ++a) {
      uint32_t tgt = h.edges[entry.edge_begin + a];
      uint32_t slot = h.l_allele_off[s] + a;
      if (tgt == site.exit_node) {
        pb_off[slot] = 0;
        len[slot] = 0;
      } else {
        pb_off[slot] = h.l_cov_off[tgt];
        len[slot] = h.nodes[tgt].seq_len;
      }
    }
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_allele_base_layout")

static std::vector<uint32_t> bubble_order(const gmx::HostIndex &h) {
  std::vector<uint32_t> order(h.sites.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (h.site_ref_pos[a] != h.site_ref_pos[b]) return h.site_ref_pos[a] > h.site_ref_pos[b];
    return a > b;
  });
  return order;
}

int gmx_index_bubble_order(const gmx_index *ix, uint32_t *out) try {
  auto o = bubble_order(ix->h);
  for (size_t i = 0; i < o.size(); ++i) out[i] = 5 + 2 * o[i];
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_bubble_order")

int gmx_compute_coverage_depth(const gmx_index *ix, const uint32_t *per_base_raw, const uint32_t *grouped_raw,
                               const uint32_t *glog, uint64_t n_log, gmx_depth_stats *out) try {
  const gmx::HostIndex &h = ix->h;
  // per-site haplogroup totals in uint16 arithmetic (get_max_cov_haplogroup, read_stats.cpp:72-92)
  std::vector<std::map<int32_t, uint16_t>> hap(h.sites.size());
  std::vector<std::map<std::vector<int32_t>, uint16_t>> logged(h.sites.size());
  for (uint64_t i = 0; i < n_log;) {  // records worth +1 or +count (gmx.h: gmx_coverage_fetch_grouped_log)
    if (glog[i] == 0xFFFFFFFFu) {  // padding
      ++i;
      continue;
    }
    if (i + 2 > n_log) break;
    const uint32_t s = glog[i], n = glog[i + 1] & ~GMX_LOG_COUNTED;
    const uint64_t head = (glog[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
    if (s >= h.sites.size() || i + head + n > n_log) {
      gmx_set_error("corrupt grouped log");
      return GMX_EINVAL;
    }
    const uint64_t count = head == 4 ? ((uint64_t)glog[i + 2] | ((uint64_t)glog[i + 3] << 32)) : 1;
    std::vector<int32_t> ids(glog + i + head, glog + i + head + n);
    logged[s][ids] = (uint16_t)(logged[s][ids] + count);
    i += head + n;
  }
  for (size_t s = 0; s < h.sites.size(); ++s) {
    const GmxSite &site = h.sites[s];
    if (h.l_grouped_off[s] != GMX_GROUPED_LOG) {
      uint32_t nm = (1u << site.n_alleles) - 1u;
      for (uint32_t m = 0; m < nm; ++m) {
        uint32_t tot = grouped_raw[h.l_grouped_off[s] + m];
        if (!tot) continue;
        uint16_t c = (uint16_t)(tot & 0xFFFFu);
        for (uint32_t a = 0; a < site.n_alleles; ++a)
          if (((m + 1) >> a) & 1u) hap[s][(int32_t)a] = (uint16_t)(hap[s][(int32_t)a] + c);
      }
    }
    for (auto &e : logged[s])
      for (auto a : e.first) hap[s][a] = (uint16_t)(hap[s][a] + e.second);
  }
  auto max_hap = [&](size_t s) -> std::pair<int32_t, uint16_t> {
    std::pair<int32_t, uint16_t> best{0, 0};
    bool any = false;
    for (auto &e : hap[s])
      if (!any || e.second > best.second) {
        best = {e.first, e.second};
        any = true;
      }
    return best;
  };
  std::vector<double> coverages;
  double total = 0;
  uint64_t no_cov = 0;
  for (uint32_t s : bubble_order(h)) {
    const GmxSite &site = h.sites[s];
    if (site.parent_site != 0) continue;  // nested sites are skipped (read_stats.cpp:128-132)
    auto mx = max_hap(s);
    uint16_t allele_cov = mx.second;
    double sum = 0;
    uint64_t n_bases = 0;
    uint32_t cur = site.entry_node;
    size_t guard = 0;
    while (cur != site.exit_node) {  // extract_max_coverage_allele, read_stats.cpp:94-117
      if (++guard > h.nodes.size() + 8) {
        gmx_set_error("coverage graph walk does not terminate");
        return GMX_EINVAL;
      }
      const GmxNode &nd = h.nodes[cur];
      uint32_t ne = h.nodes[cur + 1].edge_begin - nd.edge_begin;
      if (ne > 1 && nd.seq_len == 0) {
        auto m2 = max_hap((nd.site - 5) / 2);
        if (m2.first < 0 || (uint32_t)m2.first >= ne) {
          gmx_set_error("haplogroup out of range");
          return GMX_EINVAL;
        }
        cur = h.edges[nd.edge_begin + (uint32_t)m2.first];
        continue;
      }
      if (nd.seq_len > 0 && nd.cov_off != GMX_NO_COV)
        for (uint32_t i = 0; i < nd.seq_len; ++i) {
          uint32_t v = per_base_raw[h.l_cov_off[cur] + i];
          sum += (double)(v > 65535u ? 65535u : v);
          ++n_bases;
        }
      cur = h.edges[nd.edge_begin];
    }
    double site_cov = n_bases ? sum / (double)n_bases : (double)allele_cov;
    total += site_cov;
    coverages.push_back(site_cov);
    if (allele_cov == 0) no_cov++;
  }
  double mean = total / (double)coverages.size();
  double tv = 0;
  for (double c : coverages) tv += std::pow(c - mean, 2);
  out->mean_cov_depth = mean;
  out->variance_cov_depth = tv / (double)coverages.size();
  out->num_sites_noCov = no_cov;
  out->num_sites_total = coverages.size();
  return GMX_OK;
} GMX_GUARD_INT("gmx_compute_coverage_depth")

int gmx_debug_suffix_array_u16(const uint16_t *text, uint64_t n, uint16_t *out) try {
  try {
    gmx::debug_suffix_array_u16(text, (size_t)n, out);
    return GMX_OK;
  } catch (std::exception const &ex) {
    gmx_set_error(ex.what());
    return GMX_EINVAL;
  }
} GMX_GUARD_INT("gmx_debug_suffix_array_u16")

int gmx_index_copy_sa(const gmx_index *ix, uint32_t *out) try {
  memcpy(out, ix->h.sa.data(), ix->h.sa.size() * 4);
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_copy_sa")
int gmx_index_copy_bwt(const gmx_index *ix, uint32_t *out) try {
  if (ix->h.bwt.empty()) {  // (dropped by the builder on PRGs of 2^28 symbols and more: GMX_INDEX_INTROSPECTION=1 keeps it)
    gmx_set_error("this index was built without its introspection tables (GMX_INDEX_INTROSPECTION=1 keeps them)");
    return GMX_EINVAL;
  }
  memcpy(out, ix->h.bwt.data(), ix->h.bwt.size() * 4);
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_copy_bwt")
uint32_t gmx_index_rank(const gmx_index *ix, uint32_t upper, uint32_t base) try {
  if (base < 1 || base > 4) return 0;
  GmxIndexView v = ix->h.view();
  return gmx_rank(v, upper, base);
} GMX_GUARD_ZERO("gmx_index_rank")
int gmx_index_copy_pos_info(const gmx_index *ix, int64_t *out) try {
  const auto &h = ix->h;
  if (h.pos_target.size() != h.prg.size()) {
    gmx_set_error("this index was built without its introspection tables (GMX_INDEX_INTROSPECTION=1 keeps them)");
    return GMX_EINVAL;
  }
  for (size_t p = 0; p < h.prg.size(); ++p) {
    const GmxNode &nd = h.nodes[h.pos_node[p]];
    out[5 * p + 0] = nd.site;
    out[5 * p + 1] = nd.allele;
    out[5 * p + 2] = h.prg[p] <= 4 ? (int64_t)p - nd.first_pos : 0;
    out[5 * p + 3] = h.pos_target[p].first;
    out[5 * p + 4] = h.pos_target[p].second;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_index_copy_pos_info")
int64_t gmx_index_copy_target_map(const gmx_index *ix, int64_t *out, uint64_t cap) try {
  std::vector<int64_t> v{(int64_t)ix->h.target_map.size()};
  for (auto &e : ix->h.target_map) {
    v.push_back(e.first);
    v.push_back((int64_t)e.second.size());
    for (auto &t : e.second) {
      v.push_back(t.id);
      v.push_back(t.deletion_allele);
    }
  }
  if (v.size() > cap) return -(int64_t)v.size();
  std::copy(v.begin(), v.end(), out);
  return (int64_t)v.size();
} GMX_GUARD_INT("gmx_index_copy_target_map")
int64_t gmx_index_seed_states(const gmx_index *ix, const uint8_t *kmer, int64_t *out, uint64_t cap) try {
  return gmx_index_seed_states_k(ix, kmer, ix->h.kmer_size, out, cap);
} GMX_GUARD_INT("gmx_index_seed_states")
int64_t gmx_index_seed_states_k(const gmx_index *ix, const uint8_t *kmer, uint32_t len, int64_t *out, uint64_t cap) try {
  const auto &h = ix->h;
  if (len == 0 || (len != h.kmer_size && len != h.kmer_size2)) return GMX_EINVAL;
  uint32_t code = 0;
  for (uint32_t j = 0; j < len; ++j) {
    if (kmer[j] < 1 || kmer[j] > 4) return GMX_EINVAL;
    code |= (uint32_t)(kmer[j] - 1) << (2 * j);  // table index: rightmost base most significant (gmx_types.h)
  }
  auto v = gmx::seed_states_of(h, code, len != h.kmer_size);
  if (v.size() > cap) return -(int64_t)v.size();
  std::copy(v.begin(), v.end(), out);
  return (int64_t)v.size();
} GMX_GUARD_INT("gmx_index_seed_states_k")
int64_t gmx_index_jump_states(const gmx_index *ix, uint32_t lo, uint32_t hi, int64_t *out, uint64_t cap) try {
  const auto &h = ix->h;
  GmxIndexView v = h.view();
  ProbeCtx ctx;
  const GmxRankBlock b = v.blocks[lo >> GMX_BLK_SHIFT];
  gmx_marker_pass(v, lo, hi, GMX_NIL, GMX_NIL, b, ctx);
  std::vector<int64_t> r{(int64_t)ctx.n};
  for (uint32_t s = 0; s < ctx.n; ++s) {
    r.push_back(ctx.st[s].lo);
    r.push_back(ctx.st[s].hi);
    std::vector<std::pair<uint32_t, int32_t>> tmp;
    for (uint32_t x = ctx.st[s].tvd; x != GMX_NIL; x = ctx.arena[x].next) tmp.push_back({ctx.arena[x].site, ctx.arena[x].allele});
    r.push_back((int64_t)tmp.size());
    for (size_t i = tmp.size(); i-- > 0;) {
      r.push_back(tmp[i].first);
      r.push_back(tmp[i].second);
    }
    tmp.clear();
    for (uint32_t x = ctx.st[s].tvg; x != GMX_NIL; x = ctx.arena[x].next) tmp.push_back({ctx.arena[x].site, -1});
    r.push_back((int64_t)tmp.size());
    for (size_t i = tmp.size(); i-- > 0;) {
      r.push_back(tmp[i].first);
      r.push_back(-1);
    }
  }
  if (r.size() > cap) return -(int64_t)r.size();
  std::copy(r.begin(), r.end(), out);
  return (int64_t)r.size();
} GMX_GUARD_INT("gmx_index_jump_states")

// quasimap.cpp:120-141 + random.hpp:20: raw mt19937 outputs, 5000 per batch of <= 5000 reads
int gmx_master_seeds(uint32_t master_seed, const uint64_t *reads_per_file, uint64_t n_files, uint32_t *out) try {
  uint32_t mt[624];
  mt[0] = master_seed;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int idx = 624;
  auto next = [&]() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  };
  uint64_t o = 0;
  for (uint64_t f = 0; f < n_files; ++f)
    for (uint64_t start = 0; start < reads_per_file[f]; start += 5000)
      for (uint64_t i = 0; i < 5000; ++i) {
        uint32_t s = next();
        if (start + i < reads_per_file[f]) out[o++] = s;
      }
  return GMX_OK;
} GMX_GUARD_INT("gmx_master_seeds")

// ---- bit planes of encoded reads on the host (the form gmx_map_reads_packed_host uploads as it is) -------------------
uint64_t gmx_packed_pairs(const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads) try {
  if (uniform_len) return n_reads * (uint64_t)((uniform_len + 31u) / 32u);
  if (!offsets) return 0;
  return ((offsets[n_reads] >> 5) - (offsets[0] >> 5)) + n_reads;
} GMX_GUARD_ZERO("gmx_packed_pairs")

namespace {
// 8 encoded bases (bytes 1..4, base j in byte j) -> 8 bits of each plane; flags bytes outside 1..4 (gmx_pack_kernel's pack4)
inline void pack8(uint64_t x, uint32_t &lo, uint32_t &hi, uint64_t &bad) {
  const uint64_t y = x - 0x0101010101010101ull;
  bad |= (y & ~x & 0x8080808080808080ull) | (y & 0xFCFCFCFCFCFCFCFCull);
  lo = (uint32_t)(((y & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
  hi = (uint32_t)((((y >> 1) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}
// one read: ceil(len / 32) pairs at out; returns true when it holds a byte outside 1..4
inline bool pack_read(const uint8_t *p, uint32_t len, uint64_t *out) {
  uint64_t bad = 0;
  uint32_t i = 0, pair = 0;
  for (; i + 32 <= len; i += 32, ++pair) {
    uint32_t lo = 0, hi = 0;
    for (int j = 0; j < 4; ++j) {
      uint64_t x;
      memcpy(&x, p + i + 8 * j, 8);
      uint32_t l, h;
      pack8(x, l, h, bad);
      lo |= l << (8 * j);
      hi |= h << (8 * j);
    }
    out[pair] = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  if (i < len) {
    uint32_t lo = 0, hi = 0;
    for (uint32_t j = 0; i + j < len; ++j) {
      const uint32_t x = p[i + j];
      if (x < 1 || x > 4) bad = 1;
      lo |= ((x - 1u) & 1u) << j;
      hi |= (((x - 1u) >> 1) & 1u) << j;
    }
    out[pair] = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  return bad != 0;
}
}  // namespace

int gmx_pack_reads(const uint8_t *reads, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads, uint64_t *planes,
                   uint8_t *skip, int threads) try {
  if ((!reads && n_reads) || !offsets || !planes) {
    gmx_set_error("gmx_pack_reads: null argument");
    return GMX_EINVAL;
  }
  if (uniform_len)
    for (uint64_t r = 0; r < n_reads; ++r)
      if (offsets[r + 1] - offsets[r] != uniform_len) {
        gmx_set_error("gmx_pack_reads: uniform_len given but read " + std::to_string(r) + " has another length");
        return GMX_EINVAL;
      }
  const uint64_t n_pairs = gmx_packed_pairs(offsets, uniform_len, n_reads);
  const uint32_t ppr = (uniform_len + 31u) / 32u;
  const unsigned T = (unsigned)std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 256));
  auto work = [&](unsigned t) {
    const uint64_t r0 = n_reads * t / T, r1 = n_reads * (t + 1) / T;
    for (uint64_t r = r0; r < r1; ++r) {
      const uint64_t at = uniform_len ? r * ppr : ((offsets[r] >> 5) - (offsets[0] >> 5)) + r;
      const uint64_t next = uniform_len ? (r + 1) * ppr : ((offsets[r + 1] >> 5) - (offsets[0] >> 5)) + r + 1;
      const uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
      const bool bad = pack_read(reads + offsets[r], len, planes + at);
      for (uint64_t q = at + (len + 31u) / 32u; q < next; ++q) planes[q] = 0;  // the gap pairs of the offsets form
      if (skip) skip[r] = bad ? 1 : 0;
    }
  };
  if (T == 1 || n_reads < 4096) {
    const unsigned keep = T;
    (void)keep;
    for (unsigned t = 0; t < T; ++t) work(t);
  } else {
    GmxThreads th;  // (the workers only move bits between the caller's buffers: nothing in them throws)
    for (unsigned t = 1; t < T; ++t) th.run([&work, t] { work(t); });
    work(0);
    th.join();
  }
  (void)n_pairs;
  return GMX_OK;
} GMX_GUARD_INT("gmx_pack_reads")

// ---- reads as a 2-bit stream (gmx_map_reads_2bit_host): base j of the batch in bits 2j, 2j + 1 of the stream -------------
uint64_t gmx_twobit_units(const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads) try {
  const uint64_t bases = uniform_len ? n_reads * (uint64_t)uniform_len : (offsets ? offsets[n_reads] - offsets[0] : 0);
  return (bases + 31) / 32 + 1;  // (+ 1: the unit the device reads ahead of a chunk's last base)
} GMX_GUARD_ZERO("gmx_twobit_units")

int gmx_pack_reads_2bit(const uint8_t *reads, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads, uint64_t *stream,
                        uint8_t *skip, int threads) try {
  if ((!reads && n_reads) || !offsets || !stream) {
    gmx_set_error("gmx_pack_reads_2bit: null argument");
    return GMX_EINVAL;
  }
  if (uniform_len)
    for (uint64_t r = 0; r < n_reads; ++r)
      if (offsets[r + 1] - offsets[r] != uniform_len) {
        gmx_set_error("gmx_pack_reads_2bit: uniform_len given but read " + std::to_string(r) + " has another length");
        return GMX_EINVAL;
      }
  const uint64_t bases = offsets[n_reads] - offsets[0], units = (bases + 31) / 32 + 1;
  const uint8_t *src = reads + offsets[0];  // the batch's bases lie back to back: base j of the stream is src[j]
  const unsigned T = (unsigned)std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 256));
  auto work = [&](unsigned t) {
    for (uint64_t u = units * t / T; u < units * (t + 1) / T; ++u) {  // 32 bases -> one 8-byte unit
      uint64_t w = 0;
      const uint64_t j0 = u * 32, j1 = std::min(bases, j0 + 32);
      for (uint64_t j = j0; j < j1; ++j) w |= (uint64_t)((src[j] - 1u) & 3u) << (2 * (j - j0));
      stream[u] = w;
    }
    if (skip)
      for (uint64_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) {  // encode_dna_bases: any byte outside 1..4
        uint8_t bad = 0;
        for (uint64_t j = offsets[r]; j < offsets[r + 1]; ++j) bad |= (uint8_t)(reads[j] - 1u) > 3u;
        skip[r] = bad;
      }
  };
  if (T == 1 || n_reads < 4096) {
    for (unsigned t = 0; t < T; ++t) work(t);
  } else {
    GmxThreads th;  // (the workers only move bits between the caller's buffers: nothing in them throws)
    for (unsigned t = 1; t < T; ++t) th.run([&work, t] { work(t); });
    work(0);
    th.join();
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_pack_reads_2bit")

// ---- grouped logs as values: what every rank does with the all-gathered logs of an exchange (gmx_multi.hip) -----------
// `gathered` holds `world` slices of `pad` words, slice r carrying sizes[r] words of rank r's log (either record form,
// GMX_LOG_PAD words skipped); the result is one counted record per distinct (site, ids), in key order.
int64_t gmx_grouped_log_merge_gathered(const uint32_t *gathered, const uint64_t *sizes, int world, uint64_t pad, uint32_t *out,
                                       uint64_t cap_words) try {
  if ((!gathered && pad) || !sizes || world < 0) {
    gmx_set_error("gmx_grouped_log_merge_gathered: null argument");
    return GMX_EINVAL;
  }
  std::map<std::vector<uint32_t>, uint64_t> counts;
  std::vector<uint32_t> key;
  for (int r = 0; r < world; ++r) {
    if (sizes[r] > pad) {
      gmx_set_error("gmx_grouped_log_merge_gathered: a rank's log is longer than the padded slice");
      return GMX_EINVAL;
    }
    const uint32_t *w = gathered + (size_t)r * pad;
    const uint64_t n = sizes[r];
    for (uint64_t i = 0; i < n;) {
      if (w[i] == 0xFFFFFFFFu) {  // GMX_LOG_PAD
        ++i;
        continue;
      }
      if (i + 2 > n) {
        gmx_set_error("corrupt grouped log");
        return GMX_EINVAL;
      }
      const uint32_t n_ids = w[i + 1] & ~GMX_LOG_COUNTED;
      const uint64_t head = (w[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
      if (i + head + n_ids > n) {
        gmx_set_error("corrupt grouped log");
        return GMX_EINVAL;
      }
      const uint64_t count = head == 4 ? ((uint64_t)w[i + 2] | ((uint64_t)w[i + 3] << 32)) : 1;
      key.assign(1, w[i]);
      key.insert(key.end(), w + i + head, w + i + head + n_ids);
      counts[key] += count;
      i += head + n_ids;
    }
  }
  uint64_t at = 0;
  for (auto const &kv : counts) {
    const uint64_t words = 4 + (kv.first.size() - 1);
    if (out && at + words <= cap_words) {
      out[at] = kv.first[0];
      out[at + 1] = (uint32_t)(kv.first.size() - 1) | GMX_LOG_COUNTED;
      out[at + 2] = (uint32_t)kv.second;
      out[at + 3] = (uint32_t)(kv.second >> 32);
      for (size_t j = 1; j < kv.first.size(); ++j) out[at + 3 + j] = kv.first[j];
    }
    at += words;
  }
  return (int64_t)at;
} GMX_GUARD_INT("gmx_grouped_log_merge_gathered")

void gmx_finalize_u16(uint32_t *values, uint64_t n, int saturate) try {
  for (uint64_t i = 0; i < n; ++i) values[i] = saturate ? (values[i] > 65535u ? 65535u : values[i]) : (values[i] & 0xFFFFu);
} GMX_GUARD_VOID("gmx_finalize_u16")

}  // extern "C"
