// gmx_index.cpp — host-side index builder (see gmx_index.h).
#include "gmx_index.h"

#include <algorithm>
#include <array>
#include <limits>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <memory>
#include <cstring>
#include <fstream>
#include <sys/stat.h>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "gmx_core.h"
#include "gmx_internal.h"

namespace gmx {

// ===========================================================================
// Suffix array: SA-IS (Nong, Zhang, Chan 2009) on a compacted integer alphabet.
// The SA of a text ending in a unique smallest sentinel is unique, hence equal
// to the one SDSL builds for the reference (make_data_structures.cpp:9-33).
// ===========================================================================
namespace {

// Index type I is unsigned: uint32_t for real texts — the reference's SA_Index (search/types.hpp:19), good for every
// text shorter than 2^32 - 1, which whole-human PRGs (3.46 G symbols) need — and uint16_t in the test hook
// gmx_debug_suffix_array_u16, which runs the same code on texts longer than 2^15 so that every comparison and loop is
// exercised with the top bit of I in use. EMPTY (all ones) marks a free slot; valid entries are < n <= EMPTY - 1.
struct TypeBits {
  std::vector<uint8_t> b;
  explicit TypeBits(size_t n) : b((n + 7) / 8, 0) {}
  bool get(size_t i) const { return (b[i >> 3] >> (i & 7)) & 1; }
  void set(size_t i, bool v) {
    if (v)
      b[i >> 3] |= (uint8_t)(1u << (i & 7));
    else
      b[i >> 3] &= (uint8_t)~(1u << (i & 7));
  }
};

inline bool is_lms(const TypeBits &t, size_t i) { return i > 0 && t.get(i) && !t.get(i - 1); }

template <class I>
void get_buckets(const I *s, std::vector<I> &bkt, size_t n, size_t K, bool end) {
  std::fill(bkt.begin(), bkt.end(), (I)0);
  for (size_t i = 0; i < n; ++i) bkt[s[i]]++;
  size_t sum = 0;
  for (size_t i = 0; i <= K; ++i) {
    sum += bkt[i];
    bkt[i] = (I)(end ? sum : sum - bkt[i]);
  }
}

template <class I>
void induce_l(const TypeBits &t, I *SA, const I *s, std::vector<I> &bkt, size_t n, size_t K) {
  const I EMPTY = std::numeric_limits<I>::max();
  get_buckets(s, bkt, n, K, false);
  for (size_t i = 0; i < n; ++i) {
    const I v = SA[i];
    if (v == EMPTY || v == 0) continue;
    const size_t j = (size_t)v - 1;
    if (!t.get(j)) SA[bkt[s[j]]++] = (I)j;
  }
}

template <class I>
void induce_s(const TypeBits &t, I *SA, const I *s, std::vector<I> &bkt, size_t n, size_t K) {
  const I EMPTY = std::numeric_limits<I>::max();
  get_buckets(s, bkt, n, K, true);
  for (size_t i = n; i-- > 0;) {
    const I v = SA[i];
    if (v == EMPTY || v == 0) continue;
    const size_t j = (size_t)v - 1;
    if (t.get(j)) SA[--bkt[s[j]]] = (I)j;
  }
}

// s[n-1] must be 0 and unique smallest. K = largest symbol.
template <class I>
void sais(const I *s, I *SA, size_t n, size_t K) {
  const I EMPTY = std::numeric_limits<I>::max();
  if (n == 1) {
    SA[0] = 0;
    return;
  }
  TypeBits t(n);
  t.set(n - 1, true);
  t.set(n - 2, false);
  for (size_t i = n - 2; i-- > 0;) t.set(i, s[i] < s[i + 1] || (s[i] == s[i + 1] && t.get(i + 1)));

  std::vector<I> bkt(K + 1);
  get_buckets(s, bkt, n, K, true);
  for (size_t i = 0; i < n; ++i) SA[i] = EMPTY;
  for (size_t i = 1; i < n; ++i)
    if (is_lms(t, i)) SA[--bkt[s[i]]] = (I)i;
  induce_l(t, SA, s, bkt, n, K);
  induce_s(t, SA, s, bkt, n, K);

  size_t n1 = 0;
  for (size_t i = 0; i < n; ++i)
    if (is_lms(t, SA[i])) SA[n1++] = SA[i];  // after both passes every slot holds a suffix
  for (size_t i = n1; i < n; ++i) SA[i] = EMPTY;
  size_t name = 0, prev = 0;
  bool have_prev = false;
  for (size_t i = 0; i < n1; ++i) {
    const size_t pos = SA[i];
    bool diff = false;
    for (size_t d = 0; d < n; ++d) {
      if (!have_prev || s[pos + d] != s[prev + d] || t.get(pos + d) != t.get(prev + d)) {
        diff = true;
        break;
      } else if (d > 0 && (is_lms(t, pos + d) || is_lms(t, prev + d)))
        break;
    }
    if (diff) {
      name++;
      prev = pos;
      have_prev = true;
    }
    SA[n1 + pos / 2] = (I)(name - 1);
  }
  for (size_t i = n, j = n; i-- > n1;)
    if (SA[i] != EMPTY) SA[--j] = SA[i];

  I *SA1 = SA, *s1 = SA + n - n1;
  if (name < n1)
    sais<I>(s1, SA1, n1, name - 1);
  else
    for (size_t i = 0; i < n1; ++i) SA1[s1[i]] = (I)i;

  get_buckets(s, bkt, n, K, true);
  for (size_t i = 1, j = 0; i < n; ++i)
    if (is_lms(t, i)) s1[j++] = (I)i;
  for (size_t i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
  for (size_t i = n1; i < n; ++i) SA[i] = EMPTY;
  for (size_t i = n1; i-- > 0;) {
    const I j = SA[i];
    SA[i] = EMPTY;
    SA[--bkt[s[j]]] = j;
  }
  induce_l(t, SA, s, bkt, n, K);
  induce_s(t, SA, s, bkt, n, K);
}

}  // namespace

void debug_suffix_array_u16(const uint16_t *text, size_t n, uint16_t *sa) {
  if (n == 0 || n >= 0xFFFFu || text[n - 1] != 0) throw std::runtime_error("u16 suffix array: 0 < n < 65535 and a final sentinel 0");
  size_t K = 0;
  for (size_t i = 0; i < n; ++i) K = std::max<size_t>(K, text[i]);
  sais<uint16_t>(text, sa, n, K);
}

// fn(i) for i in [0, n_items) on `threads` threads, items handed out in order (dynamic)
template <class F>
static void par_for(size_t n_items, unsigned threads, F fn) {
  threads = (unsigned)std::min<size_t>(std::max(1u, threads), std::max<size_t>(n_items, 1));
  if (threads <= 1) {
    for (size_t i = 0; i < n_items; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  // an exception in a worker (std::bad_alloc at whole-genome scale) ends the loop and is thrown again on the caller's thread:
  // out of a thread's function it would end the process, and so would the destructor of a thread not yet joined
  std::exception_ptr first;
  std::mutex first_mu;
  {
    GmxThreads pool;
    for (unsigned w = 0; w < threads; ++w)
      pool.run([&]() {
        try {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_items) break;
            fn(i);
          }
        } catch (...) {
          next.store(n_items);
          std::lock_guard<std::mutex> lk(first_mu);
          if (!first) first = std::current_exception();
        }
      });
  }
  if (first) std::rethrow_exception(first);
}

// Parallel suffix sort for the PRG texts of this engine (round 3): suffixes are bucketed by the class of their first eight
// symbols — A, C, G, T or "marker" (after which the key stops; the sentinel counts as the smallest digit) — with a parallel
// counting sort, a key that never contradicts the suffix order, and every bucket is sorted with a plain suffix comparison.
// On DNA with variant markers the comparisons end after a few symbols (markers are all but unique), so the work is
// n log(bucket) short comparisons spread over all threads; SA-IS (one thread, O(n)) took 8 of the 40 s of a chr20-scale
// build and ten minutes at 3.46 G symbols. Long repeats make comparisons long: a budget of symbol comparisons guards
// against that — when it is spent the function gives up and the caller runs SA-IS. The result is THE suffix array either
// way (unique for a text with a unique smallest sentinel).
static bool parallel_suffix_sort(const uint32_t *text, size_t n, uint32_t *sa, unsigned threads) {
  constexpr unsigned L = 8;
  constexpr uint32_t NB = 390625;  // 5^8
  auto key_of = [&](size_t i) -> uint32_t {
    uint32_t key = 0;
    unsigned j = 0;
    for (; j < L && i + j < n; ++j) {
      const uint32_t c = text[i + j];
      if (c == 0 || c > 4) {  // the sentinel: smallest digit; a marker: the largest, and the key ends here
        key = key * 5u + (c == 0 ? 0u : 4u);
        ++j;
        break;
      }
      key = key * 5u + (c - 1u);
    }
    for (; j < L; ++j) key *= 5u;
    return key;
  };
  const unsigned T = std::max(1u, std::min(threads, 256u));
  const size_t per = (n + T - 1) / T;
  std::vector<std::vector<uint32_t>> hist(T);
  par_for(T, T, [&](size_t t) {
    hist[t].assign(NB, 0);
    const size_t lo = t * per, hi = std::min(n, lo + per);
    for (size_t i = lo; i < hi; ++i) hist[t][key_of(i)]++;
  });
  std::vector<uint64_t> bucket_start(NB + 1, 0);
  {
    uint64_t acc = 0;
    for (uint32_t b = 0; b < NB; ++b) {
      bucket_start[b] = acc;
      for (unsigned t = 0; t < T; ++t) {
        const uint32_t c = hist[t][b];
        hist[t][b] = (uint32_t)acc;  // where thread t writes its first suffix of bucket b (n < 2^32)
        acc += c;
      }
    }
    bucket_start[NB] = acc;
  }
  par_for(T, T, [&](size_t t) {
    const size_t lo = t * per, hi = std::min(n, lo + per);
    for (size_t i = lo; i < hi; ++i) sa[hist[t][key_of(i)]++] = (uint32_t)i;
  });
  hist.clear();
  std::atomic<uint64_t> budget_left{(uint64_t)n * 400ull + (1ull << 24)};
  std::atomic<bool> gave_up{false};
  const uint32_t chunk = 64;
  par_for((NB + chunk - 1) / chunk, T, [&](size_t ci) {
    if (gave_up.load(std::memory_order_relaxed)) return;
    uint64_t used = 0;
    auto less = [&](uint32_t a, uint32_t b) {
      size_t j = 0;
      const size_t lim = n - std::max(a, b);  // the shorter suffix ends with the sentinel: a difference comes first
      while (j < lim && text[a + j] == text[b + j]) ++j;
      used += j + 1;
      return j < lim ? text[a + j] < text[b + j] : a > b;
    };
    for (uint32_t b = (uint32_t)ci * chunk; b < std::min<uint32_t>(NB, ((uint32_t)ci + 1) * chunk); ++b) {
      const uint64_t lo = bucket_start[b], hi = bucket_start[b + 1];
      if (hi - lo < 2) continue;
      std::sort(sa + lo, sa + hi, less);
      if (used > (1ull << 22)) {
        if (budget_left.fetch_sub(used) < used) gave_up.store(true);
        used = 0;
        if (gave_up.load(std::memory_order_relaxed)) return;
      }
    }
    if (used && budget_left.fetch_sub(used) < used) gave_up.store(true);
  });
  return !gave_up.load();
}

// What the device pre-sort (gmx_suffixsort.hip) leaves: runs of suffixes that agree on their first 24 symbols (bit p of the
// mask: sa[p] ties with sa[p - 1]) — repeats. Each run is sorted by plain suffix comparison, runs spread over the threads,
// with parallel_suffix_sort's budget of symbol comparisons (false when it is spent: the caller sorts another way).
static bool finish_tied_runs(const uint32_t *text, size_t n, uint32_t *sa, const std::vector<uint32_t> &mask, unsigned threads) {
  std::vector<std::pair<size_t, size_t>> runs;  // [first, last]
  for (size_t w = 0; w < mask.size(); ++w) {
    uint32_t bits = mask[w];
    while (bits) {
      const size_t p = w * 32 + (size_t)__builtin_ctz(bits);
      bits &= bits - 1;
      if (p == 0 || p >= n) continue;
      if (!runs.empty() && runs.back().second == p - 1) runs.back().second = p;
      else runs.push_back({p - 1, p});
    }
  }
  if (runs.empty()) return true;
  std::atomic<uint64_t> budget_left{(uint64_t)n * 400ull + (1ull << 24)};
  std::atomic<bool> gave_up{false};
  const size_t chunk = 256;
  par_for((runs.size() + chunk - 1) / chunk, threads, [&](size_t ci) {
    if (gave_up.load(std::memory_order_relaxed)) return;
    uint64_t used = 0;
    auto less = [&](uint32_t a, uint32_t b) {
      size_t j = 0;
      const size_t lim = n - std::max(a, b);
      while (j < lim && text[a + j] == text[b + j]) ++j;
      used += j + 1;
      return j < lim ? text[a + j] < text[b + j] : a > b;
    };
    for (size_t r = ci * chunk; r < std::min(runs.size(), (ci + 1) * chunk); ++r) {
      std::sort(sa + runs[r].first, sa + runs[r].second + 1, less);
      if (used > (1ull << 22)) {
        if (budget_left.fetch_sub(used) < used) gave_up.store(true);
        used = 0;
        if (gave_up.load(std::memory_order_relaxed)) return;
      }
    }
    if (used && budget_left.fetch_sub(used) < used) gave_up.store(true);
  });
  return !gave_up.load();
}

void build_suffix_array(const std::vector<uint32_t> &text, std::vector<uint32_t> &sa, int threads) {
  size_t n = text.size();
  if (n == 0) {
    sa.clear();
    return;
  }
  // indices are uint32 (SA_Index, search/types.hpp:19); 0xFFFFFFFF marks a free slot, 0xFFFFFFFE a text-form state
  if (n >= (size_t)0xFFFFFFFEull) throw std::runtime_error("text too long for 32-bit suffix array indices (2^32 - 2 symbols at most)");
  if (text[n - 1] != 0) throw std::runtime_error("text must end with the sentinel 0");
  for (size_t i = 0; i + 1 < n; ++i)
    if (text[i] == 0) throw std::runtime_error("sentinel 0 inside the text");
  sa.assign(n, 0);
  const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
  size_t min_n = 1u << 16;  // (below that SA-IS takes milliseconds; GMX_PSORT_MIN: the tests run the parallel sort on small texts)
  if (const char *mn = getenv("GMX_PSORT_MIN")) min_n = (size_t)atoll(mn);
  // on the GPU when there is one and the text is large (GMX_DEVICE_BUILD=1: whatever the size, an error if it cannot; 0: never)
  if (g_device_suffix_presort && !getenv("GMX_SAIS") && !getenv("GMX_NO_DEVICE_SORT")) {
    const char *db = getenv("GMX_DEVICE_BUILD");
    const bool forced = db && atoi(db) != 0;
    if (forced || (!db && n >= ((size_t)1 << 22))) {
      uint32_t max_sym = 0;
      {
        std::vector<uint32_t> part_max(64, 0);
        par_for(64, hw, [&](size_t c) {
          uint32_t m = 0;
          for (size_t i = n * c / 64; i < n * (c + 1) / 64; ++i) m = std::max(m, text[i]);
          part_max[c] = m;
        });
        for (uint32_t m : part_max) max_sym = std::max(max_sym, m);
      }
      try {
        std::vector<uint32_t> mask;
        if (max_sym < (1u << 28) && g_device_suffix_presort(text.data(), n, sa.data(), mask) && finish_tied_runs(text.data(), n, sa.data(), mask, hw))
          return;
        if (forced && max_sym < (1u << 28) && !getenv("GMX_DEVICE_SORT_MAY_FALL_BACK"))
          throw std::runtime_error("GMX_DEVICE_BUILD=1: the device suffix sort could not be used (no device, or a repetitive text)");
      } catch (std::exception const &e) {
        if (forced) throw;
        fprintf(stderr, "gmx: the device suffix sort failed (%s): sorting on the host\n", e.what());
      }
    }
  }
  if (hw > 1 && n >= min_n && !getenv("GMX_SAIS") && parallel_suffix_sort(text.data(), n, sa.data(), hw)) return;
  // SA-IS on the compacted alphabet: rank of every symbol among the symbols present (no sorted copy of the text: 14 GB at 3.46 G)
  uint32_t max_sym = 0;
  for (size_t i = 0; i < n; ++i) max_sym = std::max(max_sym, text[i]);
  std::vector<uint32_t> rank((size_t)max_sym + 2, 0);
  for (size_t i = 0; i < n; ++i) rank[(size_t)text[i] + 1] = 1;
  for (size_t c = 1; c < rank.size(); ++c) rank[c] += rank[c - 1];  // rank[c] = symbols present below c
  const size_t K = rank.back() - 1;
  {
    std::vector<uint32_t> s(n);
    for (size_t i = 0; i < n; ++i) s[i] = rank[text[i]];
    sais<uint32_t>(s.data(), sa.data(), n, K);
  }
}

std::vector<uint32_t> read_prg_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("PRG String file not found: " + path);
  fseek(f, 0, SEEK_END);
  const long long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const size_t n = bytes > 0 ? (size_t)bytes / 4 : 0;  // a trailing partial word is ignored, as the reference's read loop does
  std::vector<uint32_t> prg(n);
  // little-endian uint32 per symbol (linearised_prg.cpp:8-45): read in place (12.5 GB at whole-genome scale: a byte vector and
  // a conversion loop doubled the memory and took most of a minute), bytes swapped afterwards on a big-endian host
  size_t got = 0;
  while (got < n) {
    const size_t r = fread(prg.data() + got, 4, std::min<size_t>(n - got, (size_t)1 << 26), f);
    if (r == 0) break;
    got += r;
  }
  fclose(f);
  if (got != n) throw std::runtime_error("PRG String file could not be read: " + path);
  const uint32_t probe = 1;
  if (*reinterpret_cast<const unsigned char *>(&probe) != 1)
    for (auto &x : prg) x = __builtin_bswap32(x);
  return prg;
}

// ===========================================================================
// Graph flattening + marker adjacency (coverage_graph.cpp:82-379)
// ===========================================================================
namespace {

enum class MType : uint8_t { sequence, site_entry, allele_end, site_end };  // (one byte per PRG position: 3.46 GB at whole-genome scale)

struct BuildNode {
  uint32_t site = 0;
  int32_t allele = -1;
  uint32_t seq_len = 0;
  uint32_t first_pos = 0;
  uint32_t n_next = 0;             // outgoing edges (kept in GraphBuild::links: a vector per node is a heap block per node)
  uint32_t next0 = 0xFFFFFFFFu;    // the first one
};

struct OpenSite {
  uint32_t site;
  uint32_t entry, exit;
  int32_t allele;
  uint64_t entry_pos, exit_pos;  // coverage_Node::pos bookkeeping (coverage_graph.cpp:97-110,174-258)
};

struct GraphBuild {
  std::vector<BuildNode> nodes;
  std::vector<std::pair<uint32_t, uint32_t>> links;  // edges (from, to) in creation order: a node's edges keep that order
  std::vector<uint32_t> pos_node;
  std::vector<MType> mtype;
  std::unordered_map<uint32_t, std::pair<uint32_t, int32_t>> parent;  // par_map
  std::map<uint32_t, std::vector<TargetedMarker>> target_map;
  std::vector<std::pair<uint32_t, int32_t>> pos_target;
  struct Bubble {
    uint32_t site, entry, exit;  // site marker; its entry and exit nodes
    uint64_t ref_pos;            // pos of its bubble start
  };
  std::vector<Bubble> bubbles;  // ascending site marker (sorted at the end of build_graph: a map insert per site costs seconds at 10^7 sites)
};

void build_graph(const std::vector<uint32_t> &prg, GraphBuild &g) {
  const size_t N = prg.size();
  // linearised_prg.cpp:52-80: last position of each allele marker; duplicate site markers are an error
  // (flat tables over the marker values: two hash maps with an entry per marker took a third of this function)
  uint32_t max_marker = 0;
  for (size_t p = 0; p < N; ++p) max_marker = std::max(max_marker, prg[p]);
  if ((uint64_t)max_marker > 2ull * N + 16)  // (N symbols cannot hold that many sites: numbered with a gap; no 16 GB table for it)
    throw std::runtime_error("site markers must be numbered 5,7,9,... without gaps (found " + std::to_string(max_marker - (max_marker & 1 ? 0u : 1u)) + ")");
  std::vector<uint32_t> end_pos((size_t)max_marker + 1, 0);  // position + 1 of the marker's last occurrence (0: none)
  {
    std::vector<uint8_t> seen((size_t)max_marker + 1, 0);
    for (size_t p = 0; p < N; ++p) {
      uint32_t m = prg[p];
      if (m == 0) throw std::runtime_error("PRG symbols must be >= 1");
      if (m <= 4) continue;
      if (m & 1) {
        if (seen[m])
          throw std::runtime_error("PRG consistency error: site marker " + std::to_string(m) + " used for two different sites");
        seen[m] = 1;
      } else
        end_pos[m] = (uint32_t)p + 1u;
    }
  }
  g.pos_node.assign(N, 0);
  g.mtype.assign(N, MType::sequence);
  g.pos_target.assign(N, {0u, -1});
  std::vector<OpenSite> stack;
  auto new_node = [&](uint32_t site, int32_t allele, uint32_t first_pos) {
    BuildNode n;
    n.site = site;
    n.allele = allele;
    n.first_pos = first_pos;
    g.nodes.push_back(n);
    return (uint32_t)g.nodes.size() - 1;
  };
  uint32_t back = new_node(0, -1, 0);  // root
  int64_t cur = -1;                    // open sequence node
  auto link = [&](uint32_t from, uint32_t to) {
    g.links.push_back({from, to});
    if (g.nodes[from].n_next++ == 0) g.nodes[from].next0 = to;
  };
  auto wire = [&](uint32_t target) {   // coverage_graph.cpp:260-266
    if (cur >= 0) {
      link(back, (uint32_t)cur);
      link((uint32_t)cur, target);
      cur = -1;
    } else
      link(back, target);
  };
  MType prev_t = MType::sequence;
  uint32_t prev_m = 0;
  uint64_t cur_pos = 0;  // position along the all-first-alleles path
  for (size_t p = 0; p < N; ++p) {
    uint32_t m = prg[p];
    MType t;
    if (m <= 4) {
      t = MType::sequence;
      if (cur < 0) {
        uint32_t site = stack.empty() ? 0 : stack.back().site;
        int32_t allele = stack.empty() ? -1 : stack.back().allele;
        cur = new_node(site, allele, (uint32_t)p);
      }
      g.nodes[cur].seq_len++;
      cur_pos++;
      g.pos_node[p] = (uint32_t)cur;
      if (prev_t != MType::sequence) {  // map_targets, coverage_graph.cpp:280-284
        int32_t cur_allele = stack.empty() ? -1 : stack.back().allele;
        g.pos_target[p] = {prev_m, cur_allele};
      }
    } else if (m & 1) {
      t = MType::site_entry;
      uint32_t entry = new_node(m, -1, (uint32_t)p);
      wire(entry);
      uint32_t exit = new_node(m, -1, (uint32_t)p);
      if (!stack.empty()) g.parent[m] = {stack.back().site, stack.back().allele};
      if (prev_t != MType::sequence) {  // make_site_entry_target, coverage_graph.cpp:313-328
        uint32_t target = prev_t == MType::allele_end ? prev_m - 1 : prev_m;
        if (!g.target_map.count(m)) g.target_map[m] = {TargetedMarker{target, -1}};
      }
      stack.push_back(OpenSite{m, entry, exit, 0, cur_pos, cur_pos});
      g.bubbles.push_back(GraphBuild::Bubble{m, entry, exit, cur_pos});
      back = entry;
      g.pos_node[p] = entry;
    } else {
      if (stack.empty() || stack.back().site + 1 != m)
        throw std::runtime_error("PRG consistency error: allele marker " + std::to_string(m) + " does not close the open site");
      OpenSite &top = stack.back();
      bool last = end_pos[m] == (uint32_t)p + 1u;
      t = last ? MType::site_end : MType::allele_end;
      int32_t ending_allele = top.allele;
      if (prev_t != MType::sequence) {
        if (last) {  // make_site_exit_target, coverage_graph.cpp:330-350
          if (prev_t == MType::site_entry)
            throw std::runtime_error("PRG consistency error: site number " + std::to_string(m) + " is empty");
          if (prev_t == MType::site_end)
            g.target_map[m].push_back(TargetedMarker{prev_m, -1});
          else
            g.target_map[m].push_back(TargetedMarker{prev_m - 1, ending_allele});
        } else {  // make_allele_end_target, coverage_graph.cpp:352-369
          if (prev_t == MType::site_entry)
            g.target_map[m].push_back(TargetedMarker{prev_m, ending_allele});
          else if (prev_t == MType::site_end)
            g.target_map[m].push_back(TargetedMarker{prev_m, -1});
          else
            g.target_map[m].push_back(TargetedMarker{prev_m - 1, ending_allele});
        }
      }
      wire(top.exit);
      if (top.allele == 0) top.exit_pos = cur_pos;  // the exit takes the end coordinate of the FIRST allele
      if (!last) {
        cur_pos = top.entry_pos;
        top.allele++;
        back = top.entry;
        g.pos_node[p] = top.entry;
      } else {
        if (top.allele == 0)  // coverage_graph.cpp:220-222
          throw std::runtime_error("Site numbered " + std::to_string(m) + " has only one allele");
        uint32_t exit = top.exit;
        g.nodes[exit].first_pos = (uint32_t)p;
        cur_pos = top.exit_pos;
        stack.pop_back();
        back = exit;
        g.pos_node[p] = exit;
      }
    }
    g.mtype[p] = t;
    prev_t = t;
    prev_m = m;
  }
  if (!stack.empty()) throw std::runtime_error("PRG consistency error: site " + std::to_string(stack.back().site) + " is never closed");
  uint32_t sink = new_node(0, -1, (uint32_t)N);
  wire(sink);
  auto by_site = [](const GraphBuild::Bubble &a, const GraphBuild::Bubble &b) { return a.site < b.site; };
  if (!std::is_sorted(g.bubbles.begin(), g.bubbles.end(), by_site)) std::sort(g.bubbles.begin(), g.bubbles.end(), by_site);
}

// ---------------------------------------------------------------------------
// Seed-table enumeration (the k-mer index, build/kmer_index/build.cpp:18-131)
// ---------------------------------------------------------------------------
// The context gmx_marker_pass / gmx_run_program (gmx_core.h) append to: a list of states and the path nodes they point at.
// `base` > 0: a piece of a marker pass run beside others — nodes below `base` are the shared ones (read only), this
// piece's own are numbered from `base` on and renumbered when the pieces are joined in order.
struct WalkCtx {
  std::vector<WalkState> *list;
  std::vector<GmxPathNode> *arena;
  const std::vector<GmxPathNode> *shared = nullptr;
  uint32_t base = 0;
  uint32_t status = GMX_TASK_MAPPED;
  bool push(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
    list->push_back(WalkState{lo, hi, tvd, tvg});
    return true;
  }
  uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) {
    arena->push_back(GmxPathNode{site, allele, next});
    return base + (uint32_t)arena->size() - 1;
  }
  const GmxPathNode &node(uint32_t n) const { return n < base ? (*shared)[n] : (*arena)[n - base]; }
  uint32_t arena_site(uint32_t n) const { return node(n).site; }
  uint32_t arena_next(uint32_t n) const { return node(n).next; }
  void fail(uint32_t s) { status = s; }
};

// What one task of the enumeration produces beside the entries it writes straight into the tables: the words of its
// multi-state entries (both tables, in enumeration order) and where each such entry's words start.
struct alignas(128) SeedTask : SeedPart {  // (own cache lines: neighbouring tasks run at the same time)
  std::vector<std::pair<uint32_t, int32_t>> tmp;
  std::vector<std::array<std::vector<WalkState>, 4>> kids;  // per depth: the four children's states (no allocation per node)
};

struct SeedTables {
  uint32_t k, k2;  // k2 = 0: no longer table
  GmxSeed *table, *table2;
  uint32_t *bitmap;  // presence bits of the k table
};

// Table index of a k-mer (gmx_types.h GmxSeed): the base at distance d from the k-mer's RIGHT end sits in bit pair
// K - 1 - d — the rightmost base is the most significant. The enumeration shares suffixes (the reference shares them
// through a cache of prefix diffs, build.cpp:55-86), so a task — the k-mers with given rightmost bases — owns one
// contiguous range of each table: it fills and writes its range alone, no line is shared between threads, nothing is
// copied afterwards. The walk carries `rev`, the index for K = 16; the index for K is rev >> 2 (16 - K).
inline uint32_t seed_index(uint32_t rev, uint32_t K) { return K >= 16 ? rev : rev >> (2 * (16 - K)); }

// the entry of index `code` in table `t` from the states of `list`
void seed_emit(const std::vector<WalkState> &list, const std::vector<GmxPathNode> &arena, uint32_t code, uint32_t t,
               const SeedTables &tb, SeedTask &task) {
  GmxSeed *table = t ? tb.table2 : tb.table;
  task.n_present[t]++;
  static const bool no_emit = getenv("GMX_WALK_NO_EMIT") != nullptr;  // (experiment: the walk without its stores)
  if (no_emit) return;
  if (t == 0) tb.bitmap[code >> 5] |= 1u << (code & 31);  // (whole words belong to the task)
  const uint32_t n = (uint32_t)list.size();
  const bool simple = n == 1 && list[0].tvd == GMX_NIL && list[0].tvg == GMX_NIL;
  if (simple) {
    table[code] = GmxSeed{list[0].lo, list[0].hi};
    task.n_states_all[t] += 1;
    return;
  }
  table[code] = GmxSeed{GMX_SEED_COMPLEX, 0};  // (the word offset follows when the tasks' words are joined)
  task.complex.push_back(SeedEntryRef{code, t, (uint64_t)task.words.size()});
  task.n_states_all[t] += n;
  if (n > 4) task.n_states_large[t] += n;
  std::vector<uint32_t> &w = task.words;
  w.push_back(n);
  for (uint32_t s = 0; s < n; ++s) {
    auto const &st = list[s];
    w.push_back(st.lo);
    w.push_back(st.hi);
    size_t at = w.size();
    w.push_back(0);
    w.push_back(0);
    task.tmp.clear();
    for (uint32_t x = st.tvd; x != GMX_NIL; x = arena[x].next) task.tmp.push_back({arena[x].site, arena[x].allele});
    w[at] = (uint32_t)task.tmp.size();
    for (size_t i = task.tmp.size(); i-- > 0;) {
      w.push_back(task.tmp[i].first);
      w.push_back((uint32_t)task.tmp[i].second);
    }
    task.tmp.clear();
    for (uint32_t x = st.tvg; x != GMX_NIL; x = arena[x].next) task.tmp.push_back({arena[x].site, -1});
    w[at + 1] = (uint32_t)task.tmp.size();
    for (size_t i = task.tmp.size(); i-- > 0;) w.push_back(task.tmp[i].first);
  }
}

// Depth-first enumeration of all k-mers sharing suffixes. `list`: the states after `depth` bases (the rightmost ones).
// One extension step (process_read_char_search_states, quasimap.cpp:258-268; gmx_extend in gmx_core.h) is the marker
// pass over the states — which does not depend on the base — followed by the LF step of every state, old and new: the
// marker pass runs ONCE per node and each state's rank block is read once for the four bases (round 3; four calls of
// gmx_extend per node before: 2230 CPU seconds at chr20 scale). One walk serves both tables: the states after k bases
// are the k table's entry, the walk goes on to k2. The children's lists have gmx_extend's order: the survivors among
// the node's states, then those among the states the marker pass added.
void seed_walk(const GmxIndexView &ix, const SeedTables &tb, uint32_t depth, uint32_t rev, std::vector<WalkState> &list,
               std::vector<GmxPathNode> &arena, SeedTask &task) {
  if (list.empty()) return;  // every longer k-mer with this suffix is absent too
  if (depth == tb.k) seed_emit(list, arena, seed_index(rev, tb.k), 0, tb, task);
  if (depth == tb.k2 && tb.k2 > tb.k) seed_emit(list, arena, seed_index(rev, tb.k2), 1, tb, task);
  if (depth >= (tb.k2 > tb.k ? tb.k2 : tb.k)) return;
  const size_t n0 = list.size(), arena0 = arena.size();
  if (depth > 0) {  // (the first base of a k-mer: no marker pass, build.cpp:23-27)
    WalkCtx ctx{&list, &arena};
    for (size_t s = 0; s < n0; ++s) {
      const WalkState st = list[s];
      const GmxRankBlock b = ix.blocks[st.lo >> GMX_BLK_SHIFT];
      gmx_marker_pass(ix, st.lo, st.hi, st.tvd, st.tvg, b, ctx);
    }
    if (ctx.status != GMX_TASK_MAPPED) throw std::runtime_error("seed table: inconsistent variant path while indexing k-mers");
  }
  auto &kids = task.kids[depth];
  for (auto &kd : kids) kd.clear();
  for (size_t s = 0; s < list.size(); ++s) {
    const WalkState st = list[s];
    const GmxRankBlock b = ix.blocks[st.lo >> GMX_BLK_SHIFT];
    for (uint32_t c = 1; c <= 4; ++c) {
      uint32_t lo = st.lo, hi = st.hi;
      if (gmx_lf(ix, c, lo, hi, b)) kids[c - 1].push_back(WalkState{lo, hi, st.tvd, st.tvg});
    }
  }
  list.resize(n0);
  for (uint32_t c = 1; c <= 4; ++c) seed_walk(ix, tb, depth + 1, rev | ((c - 1) << (2 * (15 - depth))), kids[c - 1], arena, task);
  arena.resize(arena0);
}

// One step of the walk for a node near the root, on all threads: the states after a few bases are millions (every marker
// of a quarter of the BWT after the second base), one thread per node leaves most of the host idle for seconds. The
// states are cut into units — runs of states, pieces of a wide interval — whose outputs are joined in order.
void seed_step_parallel(const GmxIndexView &ix, const WalkNode &parent, bool marker_pass, WalkNode kids[4], unsigned threads) {
  std::vector<WalkState> list = parent.list;
  std::vector<GmxPathNode> arena = parent.arena;
  if (marker_pass) {
    struct Unit {
      size_t s0, s1;    // states [s0, s1) ...
      uint32_t lo, hi;  // ... or (part) this part of state s0's interval
      bool part;
    };
    std::vector<Unit> units;
    const uint32_t wide = 1u << 16;
    for (size_t s = 0; s < list.size();) {
      if (list[s].hi - list[s].lo >= wide) {
        for (uint64_t a = list[s].lo; a <= list[s].hi; a += wide)
          units.push_back(Unit{s, s + 1, (uint32_t)a, (uint32_t)std::min<uint64_t>(list[s].hi, a + wide - 1), true});
        ++s;
        continue;
      }
      size_t e = s;
      while (e < list.size() && e - s < 2048 && list[e].hi - list[e].lo < wide) ++e;
      units.push_back(Unit{s, e, 0, 0, false});
      s = e;
    }
    struct Out {
      std::vector<WalkState> list;
      std::vector<GmxPathNode> arena;
      bool failed = false;
    };
    std::vector<Out> outs(units.size());
    const uint32_t base = (uint32_t)arena.size();
    par_for(units.size(), threads, [&](size_t u) {
      WalkCtx ctx{&outs[u].list, &outs[u].arena, &arena, base};
      for (size_t s = units[u].s0; s < units[u].s1; ++s) {
        const WalkState st = list[s];
        const bool part = units[u].part;
        const uint32_t lo = part ? units[u].lo : st.lo, hi = part ? units[u].hi : st.hi;
        const GmxRankBlock b = ix.blocks[lo >> GMX_BLK_SHIFT];
        gmx_marker_pass(ix, lo, hi, st.tvd, st.tvg, b, ctx);
      }
      outs[u].failed = ctx.status != GMX_TASK_MAPPED;
    });
    for (auto &o : outs) {
      if (o.failed) throw std::runtime_error("seed table: inconsistent variant path while indexing k-mers");
      const uint32_t shift = (uint32_t)arena.size() - base;  // this unit's node `base + i` becomes `arena.size() + i`
      auto fix = [&](uint32_t x) { return x != GMX_NIL && x >= base ? x + shift : x; };
      for (auto nd : o.arena) {
        nd.next = fix(nd.next);
        arena.push_back(nd);
      }
      for (auto st : o.list) list.push_back(WalkState{st.lo, st.hi, fix(st.tvd), fix(st.tvg)});
    }
  }
  // LF step of every state for the four bases, in chunks joined in order
  const size_t chunk = 4096, n_chunks = (list.size() + chunk - 1) / chunk;
  std::vector<std::array<std::vector<WalkState>, 4>> parts(n_chunks);
  par_for(n_chunks, threads, [&](size_t ci) {
    for (size_t s = ci * chunk; s < std::min(list.size(), (ci + 1) * chunk); ++s) {
      const WalkState st = list[s];
      const GmxRankBlock b = ix.blocks[st.lo >> GMX_BLK_SHIFT];
      for (uint32_t c = 1; c <= 4; ++c) {
        uint32_t lo = st.lo, hi = st.hi;
        if (gmx_lf(ix, c, lo, hi, b)) parts[ci][c - 1].push_back(WalkState{lo, hi, st.tvd, st.tvg});
      }
    }
  });
  // each child keeps the path nodes its states reach, renumbered (most nodes belong to states another base continued)
  par_for(4, threads, [&](size_t c) {
    WalkNode &kid = kids[c];
    kid.list.clear();
    kid.arena.clear();
    for (auto &pt : parts) kid.list.insert(kid.list.end(), pt[c].begin(), pt[c].end());
    std::vector<uint32_t> renamed(arena.size(), GMX_NIL);
    std::vector<uint32_t> chain;
    auto keep = [&](uint32_t x) -> uint32_t {
      if (x == GMX_NIL) return x;
      chain.clear();
      uint32_t y = x;
      for (; y != GMX_NIL && renamed[y] == GMX_NIL; y = arena[y].next) chain.push_back(y);
      uint32_t below = y == GMX_NIL ? GMX_NIL : renamed[y];
      for (size_t i = chain.size(); i-- > 0;) {
        kid.arena.push_back(GmxPathNode{arena[chain[i]].site, arena[chain[i]].allele, below});
        below = renamed[chain[i]] = (uint32_t)kid.arena.size() - 1;
      }
      return renamed[x];
    };
    for (auto &st : kid.list) {
      st.tvd = keep(st.tvd);
      st.tvg = keep(st.tvg);
    }
  });
}

}  // namespace

DeviceSeedWalk g_device_seed_walk = nullptr;
DeviceSuffixPresort g_device_suffix_presort = nullptr;

GmxIndexView HostIndex::view() const {
  GmxIndexView v;
  memset(&v, 0, sizeof(v));
  v.n = (uint32_t)prg.size() + 1;
  v.n_prg = (uint32_t)prg.size();
  v.sentinel_pos = sentinel_pos;
  v.kmer_size = kmer_size;
  for (int i = 0; i < 8; ++i) v.C[i] = C[i];
  v.n_blocks = (uint32_t)blocks.size();
  v.n_hits = (uint32_t)hits.size();
  v.n_nodes = nodes.empty() ? 0 : (uint32_t)nodes.size() - 1;
  v.n_sites = (uint32_t)sites.size();
  v.n_allele_slots = n_allele_slots;
  v.n_pb_slots = n_pb_slots;
  v.n_grouped_slots = n_grouped_slots;
  v.n_acc_slots = n_acc_slots;
  v.is_nested = is_nested ? 1 : 0;
  v.blocks = blocks.data();
  v.hits = hits.data();
  v.hit_perm = hit_perm.data();
  v.hit_prog = hit_prog.data();
  v.text = text.data();
  v.prog = prog.data();
  v.sa = sa.data();
  v.pos_node = pos_node.data();
  v.nodes = nodes.data();
  v.edges = edges.data();
  v.sites = sites.data();
  v.site_geo = site_geo.data();
  v.seeds = seeds.data();
  v.seeds2 = seeds2.data();
  v.kmer_size2 = kmer_size2;
  v.seed_words = seed_words.data();
  v.seed_shift = seed_shift;
  v.kmer_bitmap = kmer_bitmap.data();
  return v;
}

// GMX_BUILD_TRACE=1 in the environment: the builder's phases with their wall times on stderr
static void build_trace(const char *what) {
  static const bool on = getenv("GMX_BUILD_TRACE") != nullptr;
  static auto t_last = std::chrono::steady_clock::now();
  if (!on) return;
  const auto now = std::chrono::steady_clock::now();
  double rss_gb = 0, peak_gb = 0;  // resident memory now and its high-water mark (the container's limit is what bounds a whole-genome build)
  if (FILE *f = fopen("/proc/self/status", "r")) {
    char line[256];
    while (fgets(line, sizeof(line), f)) {
      unsigned long long kb = 0;
      if (sscanf(line, "VmRSS: %llu", &kb) == 1) rss_gb = kb / 1048576.0;
      if (sscanf(line, "VmHWM: %llu", &kb) == 1) peak_gb = kb / 1048576.0;
    }
    fclose(f);
  }
  fprintf(stderr, "[build %8.2f s, RSS %6.1f GiB, peak %6.1f GiB] %s\n", std::chrono::duration<double>(now - t_last).count(), rss_gb, peak_gb, what);
  t_last = now;
}

static void build_index_impl(HostIndex &out, uint32_t kmer_size, int threads, int seed_k2);
void build_index(const std::vector<uint32_t> &prg, uint32_t kmer_size, HostIndex &out, int threads, int seed_k2) {
  out = HostIndex();
  out.prg = prg;
  build_index_impl(out, kmer_size, threads, seed_k2);
}
void build_index(std::vector<uint32_t> &&prg, uint32_t kmer_size, HostIndex &out, int threads, int seed_k2) {  // (12.4 GB at whole-genome scale: no copy)
  out = HostIndex();
  out.prg = std::move(prg);
  build_index_impl(out, kmer_size, threads, seed_k2);
}
static void build_index_impl(HostIndex &out, uint32_t kmer_size, int threads, int seed_k2) {
  build_trace("start");
  const std::vector<uint32_t> &prg = out.prg;
  out.kmer_size = kmer_size;
  const size_t N = prg.size();
  if (N == 0) throw std::runtime_error("empty PRG");
  if (N >= 0xFFFFFFF0ull) throw std::runtime_error("PRG too long: positions are 32-bit (SA_Index, search/types.hpp:19), 2^32 - 16 symbols at most");

  // --- graph ---------------------------------------------------------------
  GraphBuild g;
  build_graph(prg, g);
  out.is_nested = !g.parent.empty();
  out.pos_node = std::move(g.pos_node);  // (moved, not copied: 12 + 25 GB at whole-genome scale)
  out.pos_target = std::move(g.pos_target);
  for (auto &e : g.target_map) out.target_map.push_back({e.first, e.second});

  // sites: markers must be 5,7,9,... contiguous (siteID_to_index indexing, data_types.hpp:78-81)
  uint32_t n_sites = (uint32_t)g.bubbles.size();
  out.sites.assign(n_sites, GmxSite{});
  {
    uint32_t expect = 5;
    for (auto &b : g.bubbles) {
      if (b.site != expect)
        throw std::runtime_error("site markers must be numbered 5,7,9,... without gaps (found " + std::to_string(b.site) + ")");
      expect += 2;
    }
  }
  // nodes + edges
  out.nodes.resize(g.nodes.size() + 1);
  uint32_t pb = 0;
  uint32_t edge_at = 0;
  for (size_t i = 0; i < g.nodes.size(); ++i) {
    auto const &bn = g.nodes[i];
    GmxNode &n = out.nodes[i];
    n.site = bn.site;
    n.allele = bn.allele;
    n.seq_len = bn.seq_len;
    n.first_pos = bn.first_pos;
    n.edge_begin = edge_at;
    edge_at += bn.n_next;
    bool in_bubble = bn.allele != -1 && bn.site != 0;  // is_in_bubble, coverage_graph.hpp:60-62
    if (in_bubble && bn.seq_len > 0) {
      n.cov_off = pb;
      pb += bn.seq_len;
    } else
      n.cov_off = GMX_NO_COV;
    n.n_edges = bn.n_next;
    n.edge0 = bn.next0;
  }
  {  // every node's edges side by side, in the order they were made
    out.edges.assign(edge_at, 0);
    std::vector<uint32_t> fill(g.nodes.size(), 0);
    for (auto const &lk : g.links) out.edges[out.nodes[lk.first].edge_begin + fill[lk.first]++] = lk.second;
    g.links = std::vector<std::pair<uint32_t, uint32_t>>();
  }
  {
    GmxNode &closing = out.nodes[g.nodes.size()];
    memset(&closing, 0, sizeof(closing));
    closing.allele = -1;
    closing.cov_off = GMX_NO_COV;
    closing.edge0 = 0xFFFFFFFFu;
    closing.edge_begin = (uint32_t)out.edges.size();
  }
  out.n_pb_slots = pb;
  uint32_t as = 0, gs = 0;
  // sites of up to `dense_max` alleles get 2^A - 1 dense group counters; wider ones use the engine's append log, which costs
  // a drain to the host now and then. GMX_DENSE_MAX_ALLELES (1 .. the default) lowers it: the tests force the log with 5.
  uint32_t dense_max = GMX_GROUPED_DENSE_MAX_ALLELES;
  if (const char *dm = getenv("GMX_DENSE_MAX_ALLELES")) dense_max = (uint32_t)std::min(GMX_GROUPED_DENSE_MAX_ALLELES, std::max(1, atoi(dm)));
  out.site_ref_pos.assign(out.sites.size(), 0);
  for (auto &b : g.bubbles) {
    uint32_t idx = (b.site - 5) / 2;
    GmxSite &s = out.sites[idx];
    auto pit = g.parent.find(b.site);
    s.parent_site = pit == g.parent.end() ? 0 : pit->second.first;
    s.parent_allele = pit == g.parent.end() ? -1 : pit->second.second;
    s.n_alleles = g.nodes[b.entry].n_next;
    s.allele_sum_off = as;
    as += s.n_alleles;
    if (s.n_alleles <= dense_max) {
      s.grouped_off = gs;
      gs += (1u << s.n_alleles) - 1u;
    } else
      s.grouped_off = GMX_GROUPED_LOG;
    s.entry_node = b.entry;
    s.exit_node = b.exit;
    s.snp_kinds = 0;
    out.site_ref_pos[idx] = (uint32_t)b.ref_pos;
  }
  out.n_allele_slots = as;
  out.n_grouped_slots = gs;
  build_trace("graph + sites");
  // --- logical layout -> accumulator block (gmx_types.h) ---------------------------------
  {
    const size_t n_sites = out.sites.size(), n_nodes = g.nodes.size();
    out.l_allele_off.resize(n_sites);
    out.l_grouped_off.resize(n_sites);
    out.l_cov_off.assign(n_nodes + 1, GMX_NO_COV);
    out.phys_allele.assign(as, 0);
    out.phys_grouped.assign(gs, 0);
    out.phys_pb.assign(pb, 0);
    out.hit_fix.clear();
    std::vector<std::vector<uint32_t>> nodes_of_site(n_sites);
    for (size_t i = 0; i < n_nodes; ++i) {
      out.l_cov_off[i] = out.nodes[i].cov_off;
      if (out.nodes[i].cov_off != GMX_NO_COV) nodes_of_site[(out.nodes[i].site - 5) / 2].push_back((uint32_t)i);
    }
    uint32_t at = 0;
    for (size_t i = 0; i < n_sites; ++i) {
      GmxSite &s = out.sites[i];
      out.l_allele_off[i] = s.allele_sum_off;
      out.l_grouped_off[i] = s.grouped_off;
      const uint32_t A = s.n_alleles;
      at += at & 1u;  // even: (allele-sum, single-allele group) pairs are 64-bit words
      const uint32_t base = at;
      for (uint32_t a = 0; a < A; ++a) out.phys_allele[s.allele_sum_off + a] = base + 2 * a;
      at += 2 * A;
      uint32_t multi = GMX_GROUPED_LOG;
      if (s.grouped_off != GMX_GROUPED_LOG) {
        multi = at;
        GmxSite probe = s;
        probe.allele_sum_off = base;
        probe.grouped_off = multi;
        for (uint32_t mask = 1; mask < (1u << A); ++mask) out.phys_grouped[s.grouped_off + mask - 1] = gmx_slot_grouped(probe, mask);
        at += (1u << A) - 1u - A;
      }
      const GmxNode &entry = out.nodes[s.entry_node];
      for (uint32_t nd : nodes_of_site[i]) {
        GmxNode &n = out.nodes[nd];
        // the whole allele is this one base: entry -> node -> exit
        const bool one_base_allele = n.seq_len == 1 && multi != GMX_GROUPED_LOG && n.allele >= 0 && (uint32_t)n.allele < A &&
                                     entry.n_edges == A && out.edges[entry.edge_begin + (uint32_t)n.allele] == nd &&
                                     n.n_edges == 1 && n.edge0 == s.exit_node;
        if (one_base_allele) {
          at |= 1u;  // odd cov_off marks the node; the hit counter follows
          out.hit_fix.push_back(at + 1);
          out.hit_fix.push_back(out.l_allele_off[i] + (uint32_t)n.allele);
          out.hit_fix.push_back(out.l_grouped_off[i] + (1u << n.allele) - 1u);
          out.hit_fix.push_back(n.cov_off);
          out.phys_pb[n.cov_off] = at;
          n.cov_off = at;
          at += 2;
          continue;
        }
        at += at & 1u;
        for (uint32_t j = 0; j < n.seq_len; ++j) out.phys_pb[n.cov_off + j] = at + j;
        n.cov_off = at;
        at += n.seq_len;
      }
      s.allele_sum_off = base;
      s.grouped_off = multi;
      if (multi != GMX_GROUPED_LOG && entry.n_edges == A) {  // walk-free? (gmx_types.h)
        uint32_t kinds = 0;
        bool ok = true;
        for (uint32_t a = 0; a < A && ok; ++a) {
          const uint32_t tgt = out.edges[entry.edge_begin + a];
          if (tgt == s.exit_node)
            kinds |= GMX_ALLELE_EMPTY << (2 * a);
          else if (gmx_node_has_hit_counter(out.nodes[tgt]) && out.nodes[tgt].site == 5 + 2 * i && out.nodes[tgt].allele == (int32_t)a)
            kinds |= GMX_ALLELE_HIT << (2 * a);
          else
            ok = false;
        }
        if (ok) {
          s.snp_kinds = kinds;
          for (uint32_t a = 0; a < A && ok; ++a)
            if (((kinds >> (2 * a)) & 3u) == GMX_ALLELE_HIT)
              ok = gmx_slot_hit(s, a) == out.nodes[out.edges[entry.edge_begin + a]].cov_off + 1;
          s.snp_kinds = ok ? (kinds | GMX_SITE_WALK_FREE) : 0u;
        }
      }
    }
    out.n_acc_slots = at;
  }
  // --- site geometry (GmxSiteGeo, gmx_types.h): flat PRGs, dense sites whose alleles are one node each ------------------
  out.site_geo.assign(out.sites.size(), GmxSiteGeo{});
  if (!out.is_nested && !getenv("GMX_NO_SITE_JUMP")) {
    const unsigned hw_geo = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
    const size_t n_sites_geo = out.sites.size(), piece = 1u << 16;
    par_for((n_sites_geo + piece - 1) / piece, hw_geo, [&](size_t c) {
      for (size_t i = c * piece; i < std::min(n_sites_geo, (c + 1) * piece); ++i) {
        const GmxSite &s = out.sites[i];
        const uint32_t A = s.n_alleles;
        const GmxNode &entry = out.nodes[s.entry_node], &exitn = out.nodes[s.exit_node];
        if (A < 2 || A > 8 || s.grouped_off == GMX_GROUPED_LOG || entry.n_edges != A) continue;
        GmxSiteGeo geo{};
        geo.allele_sum_off = s.allele_sum_off;
        geo.entry_pos = entry.first_pos;
        geo.flags = A << 16;
        bool ok = entry.first_pos < N && prg[entry.first_pos] == 5 + 2 * i;
        uint32_t at_pos = entry.first_pos + 1;
        for (uint32_t a = 0; a < A && ok; ++a) {  // lengths first: the layout rule reads them
          const uint32_t tgt = out.edges[entry.edge_begin + a];
          if (tgt != s.exit_node) {
            const GmxNode &n = out.nodes[tgt];
            ok = n.site == 5 + 2 * i && n.allele == (int32_t)a && n.n_edges == 1 && n.edge0 == s.exit_node && n.seq_len >= 1 &&
                 n.seq_len <= 254 && n.first_pos == at_pos && n.cov_off != GMX_NO_COV;
            if (!ok) break;
            geo.allele_lens |= (uint64_t)n.seq_len << (8 * a);
            at_pos += n.seq_len;
          }
          at_pos += 1;  // the separator behind the allele, or the end marker
        }
        for (uint32_t a = 0; a < A && ok; ++a) {  // the counters where the rule puts them; kinds as the recording takes them
          const uint32_t tgt = out.edges[entry.edge_begin + a];
          if (tgt == s.exit_node) {
            geo.flags |= GMX_ALLELE_EMPTY << (2 * a);
            continue;
          }
          const GmxNode &n = out.nodes[tgt];
          ok = gmx_geo_cov_off(geo, a) == n.cov_off && (n.seq_len == 1) == gmx_node_has_hit_counter(n);
          geo.flags |= (n.seq_len == 1 ? GMX_ALLELE_HIT : GMX_ALLELE_LONG) << (2 * a);
        }
        // the end marker where the lengths put it, and base symbols only up to the next marker
        ok = ok && exitn.first_pos == at_pos - 1 && gmx_geo_exit_pos(geo) == exitn.first_pos && exitn.first_pos < N &&
             prg[exitn.first_pos] == 6 + 2 * i;
        if (ok && (s.snp_kinds & GMX_SITE_WALK_FREE)) ok = (s.snp_kinds & 0xFFFFu) == (geo.flags & 0xFFFFu);
        if (!ok) continue;
        size_t j = (size_t)exitn.first_pos + 1;
        while (j < N && prg[j] <= 4) ++j;
        geo.tail_len = (uint32_t)(j - exitn.first_pos - 1);
        geo.flags |= GMX_SITE_JUMP | (s.snp_kinds & GMX_SITE_WALK_FREE);
        out.site_geo[i] = geo;
      }
    });
  }
  build_trace("accumulator layout");
  // --- suffix array, BWT, rank blocks -----------------------------------------
  std::vector<uint32_t> text(prg);
  text.push_back(0);
  const size_t n = text.size();
  build_suffix_array(text, out.sa, threads);
  out.bwt.resize(n);
  {
    const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
    const size_t piece = 1u << 22;
    std::atomic<uint32_t> sentinel{0};
    par_for((n + piece - 1) / piece, hw, [&](size_t c) {  // (a random look-up per position: spread over the threads)
      for (size_t i = c * piece; i < std::min(n, (c + 1) * piece); ++i) {
        uint32_t p = out.sa[i];
        out.bwt[i] = p == 0 ? 0u : text[p - 1];
        if (p == 0) sentinel.store((uint32_t)i);
      }
    });
    out.sentinel_pos = sentinel.load();
  }
  // symbol -> first SA index (FM-index C array over the compacted alphabet)
  // symbol -> first SA index (FM-index C array over the compacted alphabet) and count: flat arrays (a map look-up per symbol
  // costs minutes at 10^9 symbols, and a map of 170 M marker symbols takes a minute to build); counted on all threads — the
  // four bases in thread-local counters, the markers (under 1 % of the symbols, each a few times) with relaxed atomics.
  const unsigned hw_sym = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
  std::vector<uint32_t> sym_count, sym_first;
  {
    uint32_t max_sym = 0;
    {
      std::vector<uint32_t> part_max(64, 0);
      par_for(64, hw_sym, [&](size_t c) {
        uint32_t m = 0;
        for (size_t i = n * c / 64; i < n * (c + 1) / 64; ++i) m = std::max(m, text[i]);
        part_max[c] = m;
      });
      for (uint32_t m : part_max) max_sym = std::max(max_sym, m);
    }
    std::unique_ptr<std::atomic<uint32_t>[]> cnt(new std::atomic<uint32_t>[(size_t)max_sym + 1]);
    par_for(64, hw_sym, [&](size_t c) {
      for (size_t s = ((size_t)max_sym + 1) * c / 64; s < ((size_t)max_sym + 1) * (c + 1) / 64; ++s) cnt[s].store(0, std::memory_order_relaxed);
    });
    std::vector<std::array<uint64_t, 8>> base_cnt(64);
    par_for(64, hw_sym, [&](size_t c) {
      std::array<uint64_t, 8> local{};
      for (size_t i = n * c / 64; i < n * (c + 1) / 64; ++i) {
        const uint32_t s = text[i];
        if (s <= 4) local[s]++;
        else cnt[s].fetch_add(1, std::memory_order_relaxed);
      }
      base_cnt[c] = local;
    });
    sym_count.assign((size_t)max_sym + 1, 0);
    par_for(64, hw_sym, [&](size_t c) {
      for (size_t s = ((size_t)max_sym + 1) * c / 64; s < ((size_t)max_sym + 1) * (c + 1) / 64; ++s) sym_count[s] = cnt[s].load(std::memory_order_relaxed);
    });
    for (uint32_t s = 0; s <= 4 && s <= max_sym; ++s) {
      uint64_t t = 0;
      for (auto const &bc : base_cnt) t += bc[s];
      sym_count[s] = (uint32_t)t;
    }
    sym_first.assign((size_t)max_sym + 2, 0);  // exclusive sums of the counts (one pass: a dependent chain, 4 B per symbol value)
    uint32_t acc = 0;
    for (size_t c = 0; c < sym_count.size(); ++c) {
      sym_first[c] = acc;
      acc += sym_count[c];
    }
  }
  for (uint32_t c = 1; c <= 4; ++c) {
    // char2comp of an absent symbol is 0 in SDSL, C[0] = 0; an absent base never yields a valid interval
    out.C[c] = c < sym_count.size() && sym_count[c] ? sym_first[c] : 0;
  }
  size_t n_blocks = (n >> GMX_BLK_SHIFT) + 1;
  out.blocks.assign(n_blocks, GmxRankBlock{});
  {  // in pieces of blocks on all threads: bit planes and the piece's own counts, then the running counts added
    const size_t piece = 1u << 15, n_pieces = (n_blocks + piece - 1) / piece;
    std::vector<std::array<uint32_t, 4>> piece_cnt(n_pieces + 1, std::array<uint32_t, 4>{0, 0, 0, 0});
    par_for(n_pieces, hw_sym, [&](size_t pc) {
      uint32_t cA = 0, cC = 0, cG = 0, cM = 0;
      for (size_t b = pc * piece; b < std::min(n_blocks, (pc + 1) * piece); ++b) {
        GmxRankBlock &blk = out.blocks[b];
        blk.cnt[0] = cA;  // (within the piece: the pieces before are added below)
        blk.cnt[1] = cC;
        blk.cnt[2] = cG;
        blk.cnt[3] = cM;
        for (size_t j = 0; j < 128; ++j) {
          size_t i = (b << GMX_BLK_SHIFT) + j;
          if (i >= n) break;
          uint32_t c = out.bwt[i];
          size_t w = j >> 6, bit = j & 63;
          if (c > 4) {
            blk.mk[w] |= 1ull << bit;
            cM++;
          } else if (c == 0 || c == 1) {
            cA++;  // sentinel counted with A (raw count), corrected at query time
          } else if (c == 2) {
            blk.lo[w] |= 1ull << bit;
            cC++;
          } else if (c == 3) {
            blk.hi[w] |= 1ull << bit;
            cG++;
          } else {
            blk.lo[w] |= 1ull << bit;
            blk.hi[w] |= 1ull << bit;
          }
        }
      }
      piece_cnt[pc + 1] = {cA, cC, cG, cM};
    });
    for (size_t pc = 0; pc < n_pieces; ++pc)
      for (int q = 0; q < 4; ++q) piece_cnt[pc + 1][q] += piece_cnt[pc][q];
    par_for(n_pieces, hw_sym, [&](size_t pc) {
      if (pc == 0) return;
      for (size_t b = pc * piece; b < std::min(n_blocks, (pc + 1) * piece); ++b)
        for (int q = 0; q < 4; ++q) out.blocks[b].cnt[q] += piece_cnt[pc][q];
    });
  }

  build_trace("suffix array, BWT, rank blocks");
  // --- jump programs --------------------------------------------------------------
  // marker SA intervals: site marker -> single index; allele marker -> [C[m], C[next symbol] - 1]
  auto marker_first = [&](uint32_t m) -> uint32_t {
    if (m >= sym_count.size() || sym_count[m] == 0) throw std::runtime_error("marker " + std::to_string(m) + " absent from the PRG");
    return sym_first[m];
  };
  auto marker_last = [&](uint32_t m) -> uint32_t { return marker_first(m) + sym_count[m] - 1; };
  struct Work {
    uint32_t marker;
    int32_t allele;
    std::vector<uint32_t> ops;  // flat (op, site, allele)
  };
  out.prog.clear();
  out.prog.push_back(0);  // program 0: no outputs (marker positions that are never scanned)
  // (programs are built chunk by chunk of the marker list, on all threads: `prog` and `prog_of` are a chunk's own, its
  // offsets are rebased when the chunks are joined; a program shared by markers of two chunks is simply built twice)
  typedef std::map<std::pair<uint32_t, int32_t>, uint32_t> ProgMemo;
  auto make_program = [&](std::vector<uint32_t> &prog, ProgMemo &prog_of, uint32_t marker0, int32_t allele0) -> uint32_t {
    auto key = std::make_pair(marker0, allele0);
    auto f = prog_of.find(key);
    if (f != prog_of.end()) return f->second;
    std::vector<std::pair<std::vector<uint32_t>, std::pair<uint32_t, uint32_t>>> outputs;
    std::vector<Work> stack;
    stack.push_back(Work{marker0, allele0, {}});
    size_t guard = 0;
    while (!stack.empty()) {
      if (++guard > 1000000) throw std::runtime_error("marker jump closure does not terminate");
      Work w = std::move(stack.back());
      stack.pop_back();
      if (w.marker & 1) {  // extend_targets_site_exit, vBWT_jump.cpp:185-228
        std::vector<uint32_t> ops = w.ops;
        ops.insert(ops.end(), {GMX_OP_EXIT, w.marker, (uint32_t)w.allele});
        uint32_t site = w.marker;
        uint32_t idx = marker_first(site);
        bool commit = true;
        bool has_next = false;
        uint32_t next_marker = 0;
        while (g.target_map.count(site)) {
          auto const &tm = g.target_map.at(site);
          if (tm.size() != 1) throw std::runtime_error("site entry point with more than one target");
          uint32_t nm = tm.back().id;
          if ((nm & 1) == 0) {
            has_next = true;
            next_marker = nm;
            commit = false;
            break;
          }
          auto par = g.parent.find(site);
          if (par == g.parent.end() || par->second.first != nm) throw std::runtime_error("double exit not recorded in the parental map");
          ops.insert(ops.end(), {GMX_OP_EXIT, nm, (uint32_t)par->second.second});
          idx = marker_first(nm);
          site = nm;
        }
        if (commit) outputs.push_back({ops, {idx, idx}});
        if (has_next) stack.push_back(Work{next_marker, 0, ops});
      } else {  // extend_targets_site_entry, vBWT_jump.cpp:230-265
        std::vector<uint32_t> ops = w.ops;
        ops.insert(ops.end(), {GMX_OP_ENTER, w.marker - 1, (uint32_t)-1});
        outputs.push_back({ops, {marker_first(w.marker), marker_last(w.marker)}});
        auto tm = g.target_map.find(w.marker);
        if (tm != g.target_map.end())
          for (auto const &t : tm->second) {
            if (t.id & 1)
              stack.push_back(Work{t.id, t.deletion_allele, ops});
            else
              stack.push_back(Work{t.id, -1, ops});
          }
      }
    }
    uint32_t off = (uint32_t)prog.size();
    prog.push_back((uint32_t)outputs.size());
    for (auto &o : outputs) {
      prog.push_back((uint32_t)(o.first.size() / 3));
      prog.insert(prog.end(), o.first.begin(), o.first.end());
      prog.push_back(o.second.first);
      prog.push_back(o.second.second);
    }
    prog_of[key] = off;
    return off;
  };
  out.hits.clear();
  std::vector<uint32_t> progs_bwt;
  {
    GmxIndexView hv = out.view();  // blocks are in place: the LF steps below only need them and C[]
    const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
    // the BWT positions holding a marker, in BWT order
    const size_t piece = 1u << 22, n_pieces = (n + piece - 1) / piece;
    std::vector<size_t> piece_count(n_pieces + 1, 0);
    par_for(n_pieces, hw, [&](size_t c) {
      size_t k = 0;
      for (size_t i = c * piece; i < std::min(n, (c + 1) * piece); ++i) k += out.bwt[i] > 4;
      piece_count[c + 1] = k;
    });
    for (size_t c = 0; c < n_pieces; ++c) piece_count[c + 1] += piece_count[c];
    const size_t n_markers = piece_count[n_pieces];
    std::vector<uint32_t> marker_at(n_markers);
    par_for(n_pieces, hw, [&](size_t c) {
      size_t k = piece_count[c];
      for (size_t i = c * piece; i < std::min(n, (c + 1) * piece); ++i)
        if (out.bwt[i] > 4) marker_at[k++] = (uint32_t)i;
    });
    struct Chunk {
      std::vector<uint32_t> prog;  // local offsets start at 1 (0 = no program, as in the joined array)
      std::string error;
    };
    const size_t chunk_markers = 1u << 14, n_chunks = (n_markers + chunk_markers - 1) / chunk_markers;
    std::vector<Chunk> chunks(n_chunks);
    out.hits.resize(n_markers);
    progs_bwt.assign(n_markers, 0);
    par_for(n_chunks, hw, [&](size_t ci) {
      Chunk &ck = chunks[ci];
      ck.prog.push_back(0);
      ProgMemo memo;
      try {
        for (size_t mi = ci * chunk_markers; mi < std::min(n_markers, (ci + 1) * chunk_markers); ++mi) {
          const size_t i = marker_at[mi];
          uint32_t p = out.sa[i];
          GmxHit hit;
          memset(&hit, 0, sizeof(hit));
          uint32_t prog_off = 0;
          if (p < N && prg[p] <= 4) {
            // left_markers_search, vBWT_jump.cpp:94-117
            uint32_t m = out.pos_target[p].first;
            int32_t a = out.pos_target[p].second;
            if ((m & 1) == 0 && g.mtype[p - 1] != MType::site_end) m -= 1;  // allele separator: a site exit going backwards
            prog_off = make_program(ck.prog, memo, m, a);
          }
          for (auto &sub : hit.sub) {
            sub.head = GMX_HIT_PROG;
            sub.site = prog_off;  // (chunk-local: rebased below)
          }
          const uint32_t *pw = ck.prog.data() + prog_off;
          if (prog_off != 0 && pw[0] == 1 && pw[1] == 1) {  // one output, one op: pre-resolve it together with its LF step
            uint32_t op = pw[2], site = pw[3], lo = pw[5], hi = pw[6];
            if (op == GMX_OP_EXIT && lo == hi) {
              uint32_t b0 = out.bwt[lo];
              for (uint32_t c = 1; c <= 4; ++c) {
                GmxHitSub &sub = hit.sub[c - 1];
                sub.head = GMX_HIT_EXIT;
                sub.site = site;
                sub.y = pw[4];
                if (b0 == c) {
                  uint32_t l2 = lo, h2 = hi;
                  const GmxRankBlock blk = hv.blocks[l2 >> GMX_BLK_SHIFT];
                  if (!gmx_lf(hv, c, l2, h2, blk) || l2 != h2) throw std::runtime_error("internal: exit LF precomputation failed");
                  sub.head |= GMX_HITF_ALIVE | GMX_HITF_TEXT;
                  sub.x = out.sa[l2];
                }
              }
            } else if (op == GMX_OP_ENTER) {
              for (uint32_t c = 1; c <= 4; ++c) {
                GmxHitSub &sub = hit.sub[c - 1];
                sub.head = GMX_HIT_ENTER;
                sub.site = site;
                uint32_t l2 = lo, h2 = hi;
                const GmxRankBlock blk = hv.blocks[l2 >> GMX_BLK_SHIFT];
                if (gmx_lf(hv, c, l2, h2, blk)) {
                  sub.head |= GMX_HITF_ALIVE;
                  if (l2 == h2) {
                    sub.head |= GMX_HITF_TEXT;
                    sub.x = out.sa[l2];
                  } else {
                    sub.x = l2;
                    sub.y = h2;
                  }
                }
              }
            }
          }
          progs_bwt[mi] = prog_off;
          out.hits[mi] = hit;
        }
      } catch (std::bad_alloc const &) {
        throw;  // (par_for hands it to the caller's thread: GMX_ENOMEM, not a message in a runtime_error)
      } catch (std::exception const &e) {
        ck.error = e.what();
      }
    });
    // join the chunks' programs and rebase the offsets (a chunk's offset o > 0 becomes base + o - 1)
    std::vector<uint64_t> base(n_chunks + 1, 0);
    base[0] = out.prog.size();
    for (size_t ci = 0; ci < n_chunks; ++ci) {
      if (!chunks[ci].error.empty()) throw std::runtime_error(chunks[ci].error);
      base[ci + 1] = base[ci] + chunks[ci].prog.size() - 1;
    }
    if (base[n_chunks] >= 0xFFFFFFFFull) throw std::runtime_error("the jump programs exceed 2^32 words");
    out.prog.resize(base[n_chunks]);
    par_for(n_chunks, hw, [&](size_t ci) {
      const Chunk &ck = chunks[ci];
      if (ck.prog.size() > 1) memcpy(out.prog.data() + base[ci], ck.prog.data() + 1, (ck.prog.size() - 1) * sizeof(uint32_t));
      for (size_t mi = ci * chunk_markers; mi < std::min(n_markers, (ci + 1) * chunk_markers); ++mi) {
        if (progs_bwt[mi] == 0) continue;
        const uint32_t off = (uint32_t)(base[ci] + progs_bwt[mi] - 1);
        for (auto &sub : out.hits[mi].sub)
          if ((sub.head & 3u) == GMX_HIT_PROG && sub.site == progs_bwt[mi]) sub.site = off;
        progs_bwt[mi] = off;
      }
    });
  }
  build_trace("jump programs + hit records");
  // --- PRG text records; hit records re-ordered from BWT order to text order ---------------
  {
    out.text.assign(N / 64 + 1, GmxTextRec{0, 0, 0, 0, 0});
    uint32_t markers = 0, opens = 0;
    for (uint32_t q = 0; q < N; ++q) {
      GmxTextRec &r = out.text[q >> GMX_TEXT_SHIFT];
      if ((q & GMX_TEXT_MASK) == 0) {
        r.mrank = markers;
        r.srank = opens;
      }
      uint32_t sym = prg[q];
      if (sym > 4) {
        r.mk |= 1ull << (q & GMX_TEXT_MASK);
        ++markers;
        if (sym & 1u) {  // opens a site: flagged in the low plane (gmx_types.h)
          r.lo |= 1ull << (q & GMX_TEXT_MASK);
          ++opens;
        }
      } else {
        r.lo |= (uint64_t)((sym - 1u) & 1u) << (q & GMX_TEXT_MASK);
        r.hi |= (uint64_t)(((sym - 1u) >> 1) & 1u) << (q & GMX_TEXT_MASK);
      }
    }
    if ((N & GMX_TEXT_MASK) == 0) {
      out.text[N >> GMX_TEXT_SHIFT].mrank = markers;
      out.text[N >> GMX_TEXT_SHIFT].srank = opens;
    }
    if (markers != out.hits.size()) throw std::runtime_error("internal: marker count mismatch");
    std::vector<GmxHit> by_text(out.hits.size());
    out.hit_perm.assign(out.hits.size(), 0);
    out.hit_prog.assign(out.hits.size(), 0);
    uint32_t rank = 0;
    for (size_t i = 0; i < n; ++i) {
      if (out.bwt[i] <= 4) continue;
      uint32_t q = out.sa[i] - 1;  // the marker's PRG position
      const GmxTextRec &r = out.text[q >> GMX_TEXT_SHIFT];
      uint32_t t = r.mrank + (uint32_t)__builtin_popcountll(r.mk & ((1ull << (q & GMX_TEXT_MASK)) - 1ull));
      by_text[t] = out.hits[rank];
      out.hit_prog[t] = progs_bwt[rank];
      out.hit_perm[rank] = t;
      ++rank;
    }
    out.hits.swap(by_text);
    // ENTER + EXIT of a one-base allele -> FUSED (gmx_types.h)
    auto text_rank = [&](uint32_t q) {
      const GmxTextRec &r = out.text[q >> GMX_TEXT_SHIFT];
      return r.mrank + (uint32_t)__builtin_popcountll(r.mk & ((1ull << (q & GMX_TEXT_MASK)) - 1ull));
    };
    for (auto &hit : out.hits)
      for (auto &sub : hit.sub) {
        if ((sub.head & 3u) != GMX_HIT_ENTER || !(sub.head & GMX_HITF_ALIVE) || !(sub.head & GMX_HITF_TEXT)) continue;
        const uint32_t p1 = sub.x;  // the allele's last base
        if (p1 == 0 || prg[p1 - 1] <= 4) continue;
        const GmxHitSub &ex = out.hits[text_rank(p1 - 1)].sub[0];
        if ((ex.head & 3u) != GMX_HIT_EXIT || ex.site != sub.site) continue;
        const uint32_t open_pos = out.nodes[out.sites[(sub.site - 5) >> 1].entry_node].first_pos;
        if (open_pos >= p1 || prg[open_pos] != sub.site || (p1 - open_pos) >= (1u << 28)) continue;
        sub.head = GMX_HIT_FUSED | GMX_HITF_ALIVE | GMX_HITF_TEXT | ((p1 - open_pos) << 4);
        sub.y = ex.y;
      }
    // Inline sites (gmx_types.h): one-base alleles with distinct bases, the whole site inside one text record, the site's
    // id = its ordinal among the opening markers — and the closing marker's four sub-records say exactly what the
    // in-register resolution does (FUSED into the allele with that base, dead for every other base).
    if (!getenv("GMX_NO_INLINE_SITES")) {
      const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
      std::atomic<uint64_t> n_inline_all{0};
      const size_t n_recs = (N + GMX_TEXT_MASK) >> GMX_TEXT_SHIFT, recs_per_piece = 1u << 14;
      // (by text record: a site that qualifies lies inside ONE record, so the pieces write disjoint records; the ordinal of
      // an opening marker = the record's count of opening markers before it + those below it in the record)
      par_for((n_recs + recs_per_piece - 1) / recs_per_piece, hw, [&](size_t piece) {
        uint64_t n_inline = 0;
        const size_t q0 = piece * recs_per_piece << GMX_TEXT_SHIFT, q1 = std::min<size_t>(N, (piece + 1) * recs_per_piece << GMX_TEXT_SHIFT);
        uint32_t ordinal = out.text[q0 >> GMX_TEXT_SHIFT].srank;
        for (uint32_t q = (uint32_t)q0; q < q1; ++q) {
          const uint32_t sym = prg[q];
          if (sym <= 4 || !(sym & 1u)) continue;
          const uint32_t my_ordinal = ordinal++;
          if (sym != 5u + 2u * my_ordinal) continue;
          uint32_t A = 0, base_of[4] = {0, 0, 0, 0};  // allele whose base is c (index c - 1), + 1
          bool ok = true;
          for (uint32_t p2 = q + 1; p2 + 1 < N && prg[p2] <= 4 && prg[p2 + 1] == sym + 1u; p2 += 2) {  // `base marker` pairs
            if (A >= 4 || base_of[prg[p2] - 1u]) {
              ok = false;
              break;
            }
            base_of[prg[p2] - 1u] = ++A;
          }
          // all of the site's alleles are among them (the last pair's marker closes the site), at least two
          if (!ok || A < 2 || out.sites[(sym - 5u) >> 1].n_alleles != A) continue;
          const uint32_t q_close = q + 2u * A;
          if ((q_close >> GMX_TEXT_SHIFT) != (q >> GMX_TEXT_SHIFT)) continue;
          const GmxHit &hit = out.hits[text_rank(q_close)];
          for (uint32_t c = 1; c <= 4 && ok; ++c) {
            const GmxHitSub &sub = hit.sub[c - 1];
            if (base_of[c - 1]) {
              const uint32_t j = base_of[c - 1] - 1u, x = q + 1u + 2u * j;
              ok = (sub.head & 3u) == GMX_HIT_FUSED && (sub.head & GMX_HITF_ALIVE) && (sub.head & GMX_HITF_TEXT) && sub.site == sym &&
                   sub.y == j && sub.x == x && (sub.head >> 4) == x - q;
            } else {
              ok = (sub.head & 3u) == GMX_HIT_ENTER && !(sub.head & GMX_HITF_ALIVE);
            }
          }
          if (!ok) continue;
          out.text[q_close >> GMX_TEXT_SHIFT].hi |= 1ull << (q_close & GMX_TEXT_MASK);
          ++n_inline;
        }
        n_inline_all += n_inline;
      });
      const uint64_t n_inline = n_inline_all.load();
      build_trace(("inline sites: " + std::to_string(n_inline)).c_str());
    }
  }

  build_trace("text records");
  // The BWT and the per-position target table have served the builder (jump programs, hit records); past this point only
  // the tests' introspection calls read them (gmx_index_copy_bwt / _copy_pos_info). On a whole-genome PRG they are 12 + 25 GB
  // that the seed tables need: dropped from 2^28 symbols on (GMX_INDEX_INTROSPECTION=1 keeps, =0 drops them at any size).
  {
    const char *keep = getenv("GMX_INDEX_INTROSPECTION");
    if (keep ? atoi(keep) == 0 : N >= ((size_t)1 << 28)) {
      std::vector<uint32_t>().swap(out.bwt);
      std::vector<std::pair<uint32_t, int32_t>>().swap(out.pos_target);
      build_trace("introspection tables dropped (BWT, per-position targets)");
    }
  }
  // --- seed tables -------------------------------------------------------------------
  // The k-mer index of the reference (k = kmer_size) and, when it pays, the same construction continued to a longer
  // k-mer (kmer_size2): the states after k2 matched bases are the states after k bases extended by k2 - k ordinary
  // steps, so seeding the search from the longer table skips those steps — and a reverse-complement task whose
  // last k2-mer does not occur in the PRG ends at the look-up. The final states of a read do not depend on k.
  out.kmer_size2 = 0;
  out.seeds2.clear();
  if (kmer_size > 0) {
    if (kmer_size > 15) throw std::runtime_error("kmer_size > 15 is not supported");
    // longer seeds: the smallest k2 > k whose k-mer space holds >= 8 x the PRG (a k2-mer then occurs ~0.1 times on
    // average), if its direct-addressed table stays within 8 GB (k2 <= 15: 288 GB of HBM per GPU make that cheap;
    // chr20 scale, k = 14: k2 = 15 is +8 % reads/s for +10 GB)
    uint32_t k2 = seed_k2 < 0 ? kmer_size : (uint32_t)seed_k2;
    if (seed_k2 < 0) {
      while (k2 < 15 && (1ull << (2 * k2)) < 8ull * (uint64_t)N) ++k2;
      if ((1ull << (2 * k2)) < 8ull * (uint64_t)N) k2 = kmer_size;  // still too dense to thin the tasks out: no second table
    }
    if (!(k2 > kmer_size && k2 <= 15)) k2 = 0;
    const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
    const uint64_t n_k = 1ull << (2 * kmer_size), n_k2 = k2 ? 1ull << (2 * k2) : 0;
    // the enumeration is split by the rightmost `split` bases — the HIGH bits of the table index (seed_index): 256 tasks,
    // 1024 on a host with more than 64 threads (finer than the threads: the tasks' sizes differ by a quarter), and a
    // task's range of the presence bitmap is whole words (>= 64 entries)
    // GMX_DEVICE_BUILD: 1 = the walk below the first levels runs on the GPU (gmx_seedwalk.hip; an error if there is none),
    // 0 = never; unset = on the GPU when one is present and the PRG is large enough for that to pay
    bool on_device = false;
    {
      const char *db = getenv("GMX_DEVICE_BUILD");
      if (db ? atoi(db) != 0 : N >= (1u << 22)) on_device = g_device_seed_walk != nullptr;
      if (db && atoi(db) != 0 && !on_device) throw std::runtime_error("GMX_DEVICE_BUILD=1: this build has no device walk");
    }
    uint32_t split = on_device ? 4u : (hw > 64 ? 5u : 4u);
    while (split > 0 && split + 3 > kmer_size) --split;
    if (split == 0) on_device = false;
    const uint32_t n_tasks = 1u << (2 * split);
    const uint64_t per_task = n_k >> (2 * split), per_task2 = n_k2 >> (2 * split);
    out.seeds.resize(n_k);  // (not touched here: every task fills its range, on its own thread and memory node)
    out.seeds2.resize(n_k2);
    out.kmer_bitmap.assign((n_k + 31) / 32, 0);
    GmxIndexView ix = out.view();
    std::vector<SeedTask> tasks(n_tasks);
    std::vector<std::string> errors(n_tasks);
    auto parallel = [&](uint32_t n_items, std::function<void(uint32_t)> fn) { par_for(n_items, hw, [&](size_t i) { fn((uint32_t)i); }); };
    std::vector<double> t_fill(n_tasks, 0), t_dfs(n_tasks, 0);
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // The first `split` bases, level by level, every node on all threads (seed_step_parallel): node i of level d has
    // the bases b_0 .. b_(d-1) from the right end, b_0 in the two highest bits of i.
    std::vector<WalkNode> level(1);
    level[0].list.push_back(WalkState{0, (uint32_t)n - 1, GMX_NIL, GMX_NIL});  // get_initial_cache_element, build.cpp:35-46
    const double t_first0 = now();
    for (uint32_t d = 0; d < split; ++d) {
      std::vector<WalkNode> next((size_t)1 << (2 * (d + 1)));
      if (level.size() < 64) {  // few nodes, millions of states each: one after the other, each on all threads
        for (size_t i = 0; i < level.size(); ++i) {
          if (level[i].list.empty()) continue;
          seed_step_parallel(ix, level[i], d > 0, &next[4 * i], hw);
          level[i] = WalkNode();
        }
      } else {  // (a thread per node from here on: starting 256 threads three times per node costs more than the node)
        std::string first_error;
        std::mutex mu;
        par_for(level.size(), hw, [&](size_t i) {
          if (level[i].list.empty()) return;
          try {
            seed_step_parallel(ix, level[i], d > 0, &next[4 * i], 1);
          } catch (std::bad_alloc const &) {
            throw;  // (par_for hands it to the caller's thread: GMX_ENOMEM, not a message in a runtime_error)
          } catch (std::exception const &e) {
            std::lock_guard<std::mutex> lock(mu);
            if (first_error.empty()) first_error = e.what();
          }
          level[i] = WalkNode();
        });
        if (!first_error.empty()) throw std::runtime_error(first_error);
      }
      level.swap(next);
    }
    const double t_first = now() - t_first0;
    std::vector<const SeedPart *> parts;
    std::vector<SeedPart> device_parts;
    if (on_device) {
      const bool forced = getenv("GMX_DEVICE_BUILD") != nullptr;
      try {
        on_device = g_device_seed_walk(out, kmer_size, k2, split, level, out.seeds.data(), out.seeds2.data(), out.kmer_bitmap.data(), device_parts, out.seed_words);
      } catch (std::exception const &e) {
        if (forced) throw;
        fprintf(stderr, "gmx: the device walk of the index build failed (%s): walking on the host\n", e.what());
        on_device = false;
      }
      if (!on_device && forced) throw std::runtime_error("GMX_DEVICE_BUILD=1: no usable device");
      if (on_device) {
        for (auto &pt : device_parts) parts.push_back(&pt);
        level.clear();
        build_trace("  k-mers enumerated (device walk)");
        if (getenv("GMX_BUILD_TRACE")) fprintf(stderr, "    first %u bases on the host (every node on all threads): %.2f s\n", split, t_first);
      } else {
        std::fill(out.kmer_bitmap.begin(), out.kmer_bitmap.end(), 0u);
        device_parts.clear();
        out.seed_words.clear();
      }
    }
    auto run_task = [&](uint32_t task) {
      try {
        const double ta = now();
        SeedTask &tk = tasks[task];
        tk.kids.resize((k2 ? k2 : kmer_size) + 1);
        const SeedTables tb{kmer_size, k2, out.seeds.data(), out.seeds2.data(), out.kmer_bitmap.data()};
        std::fill(tb.table + task * per_task, tb.table + (task + 1) * per_task, GmxSeed{1, 0});
        if (k2) std::fill(tb.table2 + task * per_task2, tb.table2 + (task + 1) * per_task2, GmxSeed{1, 0});
        const double tc = now();
        t_fill[task] = tc - ta;
        WalkNode node;
        node.list.swap(level[task].list);
        node.arena.swap(level[task].arena);
        seed_walk(ix, tb, split, split ? task << (2 * (16 - split)) : 0u, node.list, node.arena, tk);
        t_dfs[task] = now() - tc;
      } catch (std::bad_alloc const &) {
        throw;  // (par_for hands it to the caller's thread: GMX_ENOMEM, not a message in a runtime_error)
      } catch (std::exception const &e) {
        errors[task] = e.what();
      }
    };
    if (!on_device) {
      parallel(n_tasks, run_task);
      level.clear();
      for (auto &e : errors)
        if (!e.empty()) throw std::runtime_error(e);
      {  // the tasks' words one after the other in the index's buffer; a task's own copy goes as soon as it is appended
        uint64_t total = 0;
        for (auto &tk : tasks) total += tk.words.size();
        out.seed_words.reserve(total + 1);
        for (auto &tk : tasks) {
          tk.word_base = out.seed_words.size();
          tk.n_words = tk.words.size();
          uint32_t *dst = out.seed_words.grow(tk.n_words);
          const size_t piece = (size_t)1 << 22;
          const std::vector<uint32_t> &src = tk.words;
          par_for((tk.n_words + piece - 1) / piece, hw, [&](size_t c) {
            memcpy(dst + c * piece, src.data() + c * piece, std::min(piece, (size_t)tk.n_words - c * piece) * sizeof(uint32_t));
          });
          std::vector<uint32_t>().swap(tk.words);
          parts.push_back(&tk);
        }
      }
      build_trace("  k-mers enumerated");
    }
    if (!on_device && getenv("GMX_BUILD_TRACE")) {
      auto stat = [&](const char *name, const std::vector<double> &v) {
        double sum = 0, mx = 0;
        for (double x : v) {
          sum += x;
          mx = std::max(mx, x);
        }
        fprintf(stderr, "    %s: sum %.2f s over %u tasks, max %.2f s\n", name, sum, n_tasks, mx);
      };
      fprintf(stderr, "    first %u bases (every node on all threads): %.2f s\n", split, t_first);
      stat("fill", t_fill);
      stat("walk", t_dfs);
    }
    // multi-state entries: the parts' words lie back to back in out.seed_words (compact); the entries are pointed at theirs.
    // An entry's offset has 30 bits (the device copies keep two flags beside it): from 2^30 words on — whole-genome PRGs —
    // the entries start on units of 2^seed_shift words, the smallest shift that fits (GMX_SEED_SHIFT in the environment:
    // at least this shift, for tests). The padding is made IN PLACE, from the last entry backwards: a padded entry never
    // starts below its compact position, so moving the last slice first never overwrites words still to be moved; slices
    // whose destination lies wholly above the sources of their whole run move side by side on all threads.
    struct Slice {
      const SeedPart *part;
      size_t c0, c1;      // its entries [c0, c1)
      uint64_t src, len;  // compact words of the slice: [src, src + len) of out.seed_words
      uint64_t dst;       // where the slice's first entry goes
    };
    auto entry_at = [](const SeedPart &pt, size_t i) { return pt.word_base + (i < pt.complex.size() ? pt.complex[i].off : pt.n_words); };
    size_t slice_entries = (size_t)1 << 16;
    if (const char *env = getenv("GMX_SEED_SLICE")) slice_entries = (size_t)std::max(1, atoi(env));  // (tests: many small slices)
    std::vector<Slice> slices;
    uint64_t compact_total = 0;
    for (const SeedPart *pt : parts) {
      if (pt->word_base != compact_total) throw std::runtime_error("seed tables: the parts' words are not back to back");
      if (!pt->complex.empty() && pt->complex[0].off != 0) throw std::runtime_error("seed tables: a part's first entry does not start its words");
      if (pt->complex.empty() && pt->n_words) throw std::runtime_error("seed tables: words without an entry");
      for (size_t c0 = 0; c0 < pt->complex.size(); c0 += slice_entries) {
        const size_t c1 = std::min(pt->complex.size(), c0 + slice_entries);
        slices.push_back(Slice{pt, c0, c1, entry_at(*pt, c0), entry_at(*pt, c1) - entry_at(*pt, c0), 0});
      }
      compact_total += pt->n_words;
    }
    if (compact_total != out.seed_words.size()) throw std::runtime_error("seed tables: the parts do not add up to the words");
    const uint32_t n_slices = (uint32_t)slices.size();
    std::vector<uint64_t> padded_len(n_slices, 0);
    uint32_t shift = 0;
    if (const char *env = getenv("GMX_SEED_SHIFT")) shift = (uint32_t)std::min(8, std::max(0, atoi(env)));
    uint64_t padded_total = 0;
    for (;; ++shift) {
      const uint64_t unit = 1ull << shift;
      parallel(n_slices, [&](uint32_t t) {
        const Slice &sl = slices[t];
        uint64_t sum = sl.len;
        if (shift) {
          sum = 0;
          for (size_t i = sl.c0; i < sl.c1; ++i) sum += (entry_at(*sl.part, i + 1) - entry_at(*sl.part, i) + unit - 1) >> shift << shift;
        }
        padded_len[t] = sum;
      });
      padded_total = 0;
      for (uint32_t t = 0; t < n_slices; ++t) {
        slices[t].dst = padded_total;
        padded_total += padded_len[t];
      }
      if ((padded_total >> shift) < (1ull << 30)) break;
      if (shift >= 8) throw std::runtime_error("seed tables: the multi-state entries do not fit 2^30 units of 256 words");
    }
    out.seed_shift = shift;
    out.seed_words.resize(padded_total + 1);
    uint32_t *const words = out.seed_words.data();
    words[padded_total] = 0;
    // one slice: its entries to their padded places (last first when source and destination overlap), the table entries
    auto place_slice = [&](const Slice &sl, bool backwards) {
      const SeedPart &pt = *sl.part;
      const uint64_t unit = 1ull << shift;
      auto put = [&](size_t i, uint64_t at) {
        const uint64_t src = entry_at(pt, i), len = entry_at(pt, i + 1) - src;
        if (at != src) memmove(words + at, words + src, len * sizeof(uint32_t));
        const uint64_t padded = (len + unit - 1) >> shift << shift;
        for (uint64_t z = len; z < padded; ++z) words[at + z] = 0;
        auto const &c = pt.complex[i];
        (c.table ? out.seeds2 : out.seeds)[c.code] = GmxSeed{GMX_SEED_COMPLEX, (uint32_t)(at >> shift)};
        return padded;
      };
      if (!backwards) {
        uint64_t at = sl.dst;
        for (size_t i = sl.c0; i < sl.c1; ++i) at += put(i, at);
      } else {  // (an entry's padded place is at or above its compact one and below the next entry's padded place)
        uint64_t end = sl.dst;
        for (size_t i = sl.c0; i < sl.c1; ++i) end += (entry_at(pt, i + 1) - entry_at(pt, i) + unit - 1) >> shift << shift;
        for (size_t i = sl.c1; i-- > sl.c0;) {
          const uint64_t len = entry_at(pt, i + 1) - entry_at(pt, i);
          end -= (len + unit - 1) >> shift << shift;
          put(i, end);
        }
      }
    };
    if (shift == 0) {
      parallel(n_slices, [&](uint32_t t) { place_slice(slices[t], false); });
    } else {
      for (uint32_t hi = n_slices; hi > 0;) {  // runs of slices [lo, hi), from the end of the words
        const uint64_t src_end = slices[hi - 1].src + slices[hi - 1].len;
        uint32_t lo = hi;
        while (lo > 0 && slices[lo - 1].dst >= src_end) --lo;
        if (lo == hi) {  // the last slice's own destination overlaps its source: alone, last entry first
          place_slice(slices[hi - 1], true);
          --hi;
          continue;
        }
        parallel(hi - lo, [&](uint32_t t) { place_slice(slices[lo + t], false); });
        hi = lo;
      }
    }
    out.n_seed_kmers_present = 0;
    uint64_t all[2] = {0, 0}, large[2] = {0, 0};
    for (const SeedPart *pt : parts) {
      out.n_seed_kmers_present += pt->n_present[0];
      for (int t = 0; t < 2; ++t) {
        all[t] += pt->n_states_all[t];
        large[t] += pt->n_states_large[t];
      }
    }
    out.kmer_size2 = k2;
    out.n_seed_states = all[k2 ? 1 : 0];  // of the table the kernels mostly use
    out.n_seed_states_large = large[k2 ? 1 : 0];
    build_trace("seed tables");
  }
}

std::vector<int64_t> seed_states_of(const HostIndex &ix, uint32_t code, bool longer_table) {
  std::vector<int64_t> v;
  GmxSeed s = (longer_table ? ix.seeds2 : ix.seeds).at(code);
  if (s.a == 1 && s.b == 0) return {-1};
  if (s.a != GMX_SEED_COMPLEX) return {1, s.a, s.b, 0, 0};
  const uint32_t *p = ix.seed_words.data() + ((size_t)s.b << ix.seed_shift);
  uint32_t ns = *p++;
  v.push_back(ns);
  for (uint32_t i = 0; i < ns; ++i) {
    v.push_back(p[0]);
    v.push_back(p[1]);
    uint32_t nt = p[2], ng = p[3];
    p += 4;
    v.push_back(nt);
    for (uint32_t j = 0; j < nt; ++j) {
      v.push_back(p[0]);
      v.push_back((int32_t)p[1]);
      p += 2;
    }
    v.push_back(ng);
    for (uint32_t j = 0; j < ng; ++j) {
      v.push_back(*p++);
      v.push_back(-1);
    }
  }
  return v;
}


// ---------------------------------------------------------------------------------------
// index cache
// ---------------------------------------------------------------------------------------
namespace {
const uint64_t kCacheMagic = 0x31584449584d47ull;  // "GMXIDX1"
const uint32_t kCacheVersion = 11;                  // bump on any change of the tables' layout or meaning

uint64_t fnv1a_u32(const std::vector<uint32_t> &v) {
  uint64_t h = 1469598103934665603ull;
  for (uint32_t x : v) {
    h ^= x;
    h *= 1099511628211ull;
  }
  return h;
}

// Checksum of everything written / read (four interleaved multiplicative lanes over 64-bit words: memory speed): a
// damaged cache whose table sizes still match would otherwise hand unchecked indices to the host and the device.
struct Checksum {
  uint64_t lane[4] = {1, 2, 3, 4};
  uint64_t n = 0;
  void add(const void *p, size_t bytes) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
      uint64_t w[4];
      memcpy(w, c + i, 32);
      for (int k = 0; k < 4; ++k) lane[k] = lane[k] * 0x9E3779B97F4A7C15ull + w[k];
    }
    for (; i < bytes; ++i) lane[i & 3] = lane[i & 3] * 0x100000001B3ull + c[i];
    n += bytes;
  }
  uint64_t value() const { return (lane[0] ^ (lane[1] << 1) ^ (lane[2] << 2) ^ (lane[3] << 3)) + n; }
};

struct Writer {
  FILE *f;
  Checksum sum;
  void raw(const void *p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("index cache: write failed");
    sum.add(p, n);
  }
  template <class T>
  void pod(const T &v) {
    raw(&v, sizeof(T));
  }
  template <class T, class A>
  void vec(const std::vector<T, A> &v) {
    pod<uint64_t>(v.size());
    raw(v.data(), v.size() * sizeof(T));
  }
  void vec(const WordBuf &v) {  // (the same bytes as a vector of words)
    pod<uint64_t>(v.size());
    const size_t piece = (size_t)1 << 28;  // 1 GB writes: a single fwrite of 90 GB is one system call's worth of trouble
    for (size_t at = 0; at < v.size(); at += piece) raw(v.data() + at, std::min(piece, v.size() - at) * sizeof(uint32_t));
  }
};
struct Reader {
  FILE *f;
  Checksum sum;
  uint64_t left = ~0ull;  // bytes of the file not read yet: no table can be larger (a damaged count must not map 256 GB first)
  void raw(void *p, size_t n) {
    if (n > left) throw std::runtime_error("index cache: truncated file");
    if (n && fread(p, 1, n, f) != n) throw std::runtime_error("index cache: truncated file");
    left -= n;
    sum.add(p, n);
  }
  template <class T>
  void pod(T &v) {
    raw(&v, sizeof(T));
  }
  template <class T, class A>
  void vec(std::vector<T, A> &v, uint64_t max_elems = (1ull << 36)) {
    uint64_t n = 0;
    pod(n);
    if (n > max_elems || n > left / sizeof(T)) throw std::runtime_error("index cache: implausible table size");
    v.resize(n);
    raw(v.data(), n * sizeof(T));
  }
  void vec(WordBuf &v, uint64_t max_elems = (1ull << 36)) {
    uint64_t n = 0;
    pod(n);
    if (n > max_elems || n > left / sizeof(uint32_t)) throw std::runtime_error("index cache: implausible table size");
    v.resize(n);
    const size_t piece = (size_t)1 << 28;
    for (size_t at = 0; at < n; at += piece) raw(v.data() + at, std::min<size_t>(piece, n - at) * sizeof(uint32_t));
  }
};

template <class IO, class H>
void index_tables(IO &io, H &h) {  // one list of tables for both directions
  io.vec(h.blocks);
  io.vec(h.sa);
  io.vec(h.hits);
  io.vec(h.hit_perm);
  io.vec(h.hit_prog);
  io.vec(h.text);
  io.vec(h.prog);
  io.vec(h.pos_node);
  io.vec(h.nodes);
  io.vec(h.edges);
  io.vec(h.sites);
  io.vec(h.site_geo);
  io.vec(h.seeds);
  io.vec(h.seeds2);
  io.vec(h.seed_words);
  io.vec(h.kmer_bitmap);
  io.vec(h.l_allele_off);
  io.vec(h.l_grouped_off);
  io.vec(h.l_cov_off);
  io.vec(h.phys_allele);
  io.vec(h.phys_pb);
  io.vec(h.phys_grouped);
  io.vec(h.hit_fix);
  io.vec(h.site_ref_pos);
  io.vec(h.bwt);
  io.vec(h.pos_target);
}
}  // namespace

void save_index(const HostIndex &h, const std::string &path) {
  const std::string tmp = path + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) throw std::runtime_error("index cache: cannot write " + tmp);
  try {
    Writer w{f};
    w.pod(kCacheMagic);
    w.pod(kCacheVersion);
    w.pod<uint64_t>(h.prg.size());
    w.pod<uint64_t>(fnv1a_u32(h.prg));
    w.pod(h.kmer_size);
    w.pod(h.kmer_size2);
    w.pod(h.seed_shift);
    w.pod(h.sentinel_pos);
    w.raw(h.C, sizeof(h.C));
    w.pod<uint32_t>(h.is_nested ? 1u : 0u);
    w.pod(h.n_allele_slots);
    w.pod(h.n_pb_slots);
    w.pod(h.n_grouped_slots);
    w.pod(h.n_acc_slots);
    w.pod(h.n_seed_kmers_present);
    w.pod(h.n_seed_states);
    w.pod(h.n_seed_states_large);
    index_tables(w, h);
    // target_map: flattened (key, count, (id, deletion_allele) x count)
    w.pod<uint64_t>(h.target_map.size());
    for (auto const &e : h.target_map) {
      w.pod(e.first);
      w.vec(e.second);
    }
    w.pod<uint64_t>(w.sum.value());  // of every byte before it
    w.pod(kCacheMagic);              // end mark
  } catch (...) {
    fclose(f);
    remove(tmp.c_str());
    throw;
  }
  if (fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0) {
    remove(tmp.c_str());
    throw std::runtime_error("index cache: cannot finish " + path);
  }
}

void load_index(const std::string &path, const std::vector<uint32_t> &prg, uint32_t kmer_size, HostIndex &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("index cache: cannot open " + path);
  try {
    Reader r{f};
    {
      struct stat st;
      if (fstat(fileno(f), &st) == 0 && st.st_size >= 0) r.left = (uint64_t)st.st_size;
    }
    uint64_t magic = 0, n_prg = 0, hash = 0;
    uint32_t version = 0, nested = 0;
    r.pod(magic);
    r.pod(version);
    if (magic != kCacheMagic || version != kCacheVersion) throw std::runtime_error("index cache: not a version-" + std::to_string(kCacheVersion) + " cache file");
    r.pod(n_prg);
    r.pod(hash);
    r.pod(out.kmer_size);
    r.pod(out.kmer_size2);
    r.pod(out.seed_shift);
    if (out.seed_shift > 8) throw std::runtime_error("index cache: corrupt header");
    if (n_prg != prg.size() || hash != fnv1a_u32(prg)) throw std::runtime_error("index cache: built from a different PRG");
    if (out.kmer_size != kmer_size) throw std::runtime_error("index cache: built for a different kmer_size");
    r.pod(out.sentinel_pos);
    r.raw(out.C, sizeof(out.C));
    r.pod(nested);
    out.is_nested = nested != 0;
    r.pod(out.n_allele_slots);
    r.pod(out.n_pb_slots);
    r.pod(out.n_grouped_slots);
    r.pod(out.n_acc_slots);
    r.pod(out.n_seed_kmers_present);
    r.pod(out.n_seed_states);
    r.pod(out.n_seed_states_large);
    index_tables(r, out);
    uint64_t n_tm = 0;
    r.pod(n_tm);
    if (n_tm > prg.size() + 1) throw std::runtime_error("index cache: implausible table size");
    out.target_map.resize(n_tm);
    for (auto &e : out.target_map) {
      r.pod(e.first);
      r.vec(e.second);
    }
    const uint64_t computed = r.sum.value();
    uint64_t stored = 0, end = 0;
    r.pod(stored);
    r.pod(end);
    if (end != kCacheMagic || stored != computed) throw std::runtime_error("index cache: damaged file (checksum)");
    out.prg = prg;
    // cheap structural checks against damage that keeps the sizes
    if (out.sa.size() != prg.size() + 1 || out.pos_node.size() != prg.size() || out.text.size() != prg.size() / 64 + 1 ||
        out.seeds.size() != (kmer_size ? (1ull << (2 * kmer_size)) : 0) ||
        out.seeds2.size() != (out.kmer_size2 ? (1ull << (2 * out.kmer_size2)) : 0) || out.kmer_size2 > 15 || out.nodes.empty() || out.phys_allele.size() != out.n_allele_slots ||
        out.phys_pb.size() != out.n_pb_slots || out.phys_grouped.size() != out.n_grouped_slots || out.hit_fix.size() % 4 != 0 || out.site_ref_pos.size() != out.sites.size() ||
        out.site_geo.size() != out.sites.size() ||  // (the kernels index site_geo by site: a shorter table is an out-of-bounds device read)
        (out.seed_words.size() >> out.seed_shift) >= (1ull << 30))  // (entry offsets are 30-bit units of 2^seed_shift words)
      throw std::runtime_error("index cache: inconsistent tables");
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
}

}  // namespace gmx
