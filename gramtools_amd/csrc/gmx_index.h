// gmx_index.h — host-side index builder: gram_dir/prg -> flat tables (see gmx_types.h).
//
// The reference derives the same information in `gram build`
// (libgramtools/src/build/build.cpp:8-72: coverage graph, SDSL FM-index, BWT masks,
// k-mer index) and reloads it in `gram genotype` (src/prg/prg_info.cpp:6-29,
// src/build/kmer_index/load.cpp:161-173). All of it is a pure function of the
// integer PRG and k, so this engine derives it from `prg` directly.
#pragma once
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "gmx_types.h"

namespace gmx {

// Are transparent huge pages worth asking for on this host? With `defrag=madvise` a fault in a MADV_HUGEPAGE region may
// compact memory synchronously: 16 s per GB on one of the hosts this was developed on, against 0.6 s per GB with 4 KB
// pages — while on the MI355X boxes 2 GB are touched in 7 ms instead of 120. So it is measured once, on 16 MB each way
// (GMX_HUGEPAGES=0/1 in the environment overrides).
inline bool gmx_huge_pages_pay() {
  static const bool ok = [] {
    if (const char *e = getenv("GMX_HUGEPAGES")) return atoi(e) != 0;
    const size_t bytes = (size_t)16 << 20, huge = (size_t)2 << 20;
    auto touch = [&](bool advise) -> double {
      void *p = aligned_alloc(huge, bytes);
      if (!p) return 1e9;
      if (advise && madvise(p, bytes, MADV_HUGEPAGE) != 0) {
        free(p);
        return 1e9;
      }
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < bytes; i += 4096) static_cast<volatile char *>(p)[i] = 1;
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      free(p);
      return dt;
    };
    const double plain = touch(false), advised = touch(true);
    return advised < plain;
  }();
  return ok;
}

// Allocator of the direct-addressed seed tables (2 GB for k = 14, 8.6 GB for k2 = 15) and of the multi-state entries'
// words: a resize default-initialises — touches nothing — so that every builder task (or the copy from the device)
// writes its own range first, on its own thread and memory node; the first touch of fresh memory is the expensive part
// on the virtualised hosts this runs on (279 CPU seconds of the chr20-scale build in round 2, when a second copy of the
// tables was value-initialised besides). Large blocks are 2 MB-aligned and advised as huge pages where that pays.
template <class T>
struct BigAlloc {
  typedef T value_type;
  BigAlloc() = default;
  template <class U>
  BigAlloc(const BigAlloc<U> &) {}
  T *allocate(size_t n) {
    size_t bytes = n ? n * sizeof(T) : 1;
    void *p;
    const size_t huge = (size_t)2 << 20;
    if (bytes >= ((size_t)64 << 20) && gmx_huge_pages_pay()) {
      bytes = (bytes + huge - 1) / huge * huge;
      p = aligned_alloc(huge, bytes);
      if (p) madvise(p, bytes, MADV_HUGEPAGE);
    } else {
      p = malloc(bytes);
    }
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t) { free(p); }
  template <class U>
  void construct(U *p) { ::new ((void *)p) U; }
  template <class U, class A0, class... A>
  void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
  bool operator==(const BigAlloc &) const { return true; }
  bool operator!=(const BigAlloc &) const { return false; }
};
typedef std::vector<GmxSeed, BigAlloc<GmxSeed>> SeedTable;

// The words of the multi-state seed entries: tens of GB for a whole-genome PRG (configs[4]: ~2.7 G path-bearing states),
// produced group by group. An anonymous mapping grown with mremap — which moves page-table entries, not pages — so that
// appending never holds a second copy (a std::vector's reallocation does, and round 3's builder kept the parts AND their
// joined copy: its host peak was what kept the 85 M-site PRG from being built inside the container's 300 GiB). Words are
// not initialised (the writer writes every one); mapped but untouched capacity costs nothing.
// Failure mode (documented, not hidden): the mapping is MAP_NORESERVE — capacity is address space, not a promise of memory —
// so a host that runs out of memory while the words are WRITTEN kills the process (SIGBUS / the OOM killer), it does not
// raise bad_alloc / GMX_ENOMEM as the vectors around it do. The whole-genome build sizes this buffer at 80+ GB inside a
// 300 GiB container (212 GiB peak); reserving it up front would double-count against overcommit limits that
// count reservations. The cache reader bounds a table by the bytes left in the file before it maps anything (gmx_index.cpp).
class WordBuf {
 public:
  WordBuf() = default;
  WordBuf(const WordBuf &) = delete;
  WordBuf &operator=(const WordBuf &) = delete;
  WordBuf(WordBuf &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr, o.n_ = o.cap_ = 0; }
  WordBuf &operator=(WordBuf &&o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_, n_ = o.n_, cap_ = o.cap_;
      o.p_ = nullptr, o.n_ = o.cap_ = 0;
    }
    return *this;
  }
  ~WordBuf() { release(); }
  uint32_t *data() { return p_; }
  const uint32_t *data() const { return p_; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  uint32_t &operator[](size_t i) { return p_[i]; }
  const uint32_t &operator[](size_t i) const { return p_[i]; }
  void reserve(size_t words) {
    if (words <= cap_) return;
    const size_t page = (size_t)2 << 20;
    const size_t bytes = (words * sizeof(uint32_t) + page - 1) / page * page;
    void *q = p_ ? mremap(p_, cap_ * sizeof(uint32_t), bytes, MREMAP_MAYMOVE)
                 : mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) throw std::bad_alloc();
    if (bytes >= ((size_t)64 << 20) && gmx_huge_pages_pay()) madvise(q, bytes, MADV_HUGEPAGE);
    p_ = static_cast<uint32_t *>(q);
    cap_ = bytes / sizeof(uint32_t);
  }
  void resize(size_t words) {  // (new words are whatever the mapping holds: zero pages the first time)
    if (words > cap_) reserve(std::max(words, cap_ + cap_ / 4));
    n_ = words;
  }
  void clear() { n_ = 0; }
  void release() {
    if (p_) munmap(p_, cap_ * sizeof(uint32_t));
    p_ = nullptr;
    n_ = cap_ = 0;
  }
  // room for `words` more at the end; returns where they go (the caller writes all of them)
  uint32_t *grow(size_t words) {
    const size_t at = n_;
    resize(n_ + words);
    return p_ + at;
  }

 private:
  uint32_t *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

struct TargetedMarker {
  uint32_t id;
  int32_t deletion_allele;
};

struct HostIndex {
  std::vector<uint32_t> prg;  // PRG symbols (no sentinel)
  uint32_t kmer_size = 0;
  uint32_t sentinel_pos = 0;
  uint32_t C[8] = {0};
  bool is_nested = false;

  std::vector<GmxRankBlock> blocks;
  std::vector<uint32_t> sa;
  std::vector<GmxHit> hits;
  std::vector<uint32_t> hit_perm, hit_prog;
  std::vector<GmxTextRec> text;
  std::vector<uint32_t> prog;
  std::vector<uint32_t> pos_node;
  std::vector<GmxNode> nodes;  // + 1 closing record
  std::vector<uint32_t> edges;
  std::vector<GmxSite> sites;
  std::vector<GmxSiteGeo> site_geo;  // per site; flags = 0 where the site has no geometry (gmx_types.h)
  SeedTable seeds;              // direct-addressed by the k-mer's table index (gmx_types.h GmxSeed)
  uint32_t kmer_size2 = 0;      // longer seed table (0 = none): the same construction continued to k2 > kmer_size
  SeedTable seeds2;             // its 4^k2 entries; multi-state records share seed_words
  WordBuf seed_words;           // multi-state entries (every word written by the walks / the join: not value-initialised)
  uint32_t seed_shift = 0;      // multi-state entries start at (GmxSeed::b << seed_shift); > 0 from 2^30 words on
  std::vector<uint32_t> kmer_bitmap;
  uint32_t n_allele_slots = 0, n_pb_slots = 0, n_grouped_slots = 0;  // lengths of the logical arrays
  uint32_t n_acc_slots = 0;                                            // length of the accumulator block
  // logical layout (what the C ABI and the dumps use) and where each logical slot lives in the accumulator block
  std::vector<uint32_t> l_allele_off, l_grouped_off;  // per site (l_grouped_off = GMX_GROUPED_LOG for > 8 alleles)
  std::vector<uint32_t> l_cov_off;                    // per node (GMX_NO_COV if none)
  std::vector<uint32_t> phys_allele, phys_pb, phys_grouped;
  // hit counters (gmx_types.h): {slot, logical allele-sum index, logical grouped index, logical per-base index} each
  std::vector<uint32_t> hit_fix;
  std::vector<uint32_t> site_ref_pos;  // per site: coverage_Node::pos of the bubble start (first-allele coordinate; orders bubble_map)

  // introspection used by the tests (not needed on the device)
  std::vector<uint32_t> bwt;
  std::vector<std::pair<uint32_t, std::vector<TargetedMarker>>> target_map;  // ascending key
  std::vector<std::pair<uint32_t, int32_t>> pos_target;                        // per PRG position (0,-1) if none
  uint64_t n_seed_kmers_present = 0;
  uint64_t n_seed_states = 0, n_seed_states_large = 0;  // states of all entries / of entries with more than 4 states

  GmxIndexView view() const;  // host-pointer view
};

// Suffix array of `text` (last symbol must be the unique smallest symbol 0). SA-IS, O(n).
void build_suffix_array(const std::vector<uint32_t> &text, std::vector<uint32_t> &sa, int threads = 1);
// The same code with 16-bit indices (test hook: texts longer than 2^15 use the index type's top bit, as texts longer
// than 2^31 do with the 32-bit indices of the product).
void debug_suffix_array_u16(const uint16_t *text, size_t n, uint16_t *sa);

// Builds everything. Throws std::runtime_error on an inconsistent PRG (same conditions as
// PRG_String / cov_Graph_Builder: linearised_prg.cpp:52-80, coverage_graph.cpp:220-222,338-341).
// kmer_size == 0 skips the seed table. threads <= 0 uses all hardware threads for the seed table.
// seed_k2: length of the longer seed table; -1 = choose from the PRG size, 0 (or <= kmer_size) = none.
void build_index(const std::vector<uint32_t> &prg, uint32_t kmer_size, HostIndex &out, int threads = 0, int seed_k2 = -1);
void build_index(std::vector<uint32_t> &&prg, uint32_t kmer_size, HostIndex &out, int threads = 0, int seed_k2 = -1);  // takes the symbols over

// Index cache: everything build_index derives, as one flat file, so that `gram genotype` starts with
// a read + H2D instead of SA construction and the seed-table enumeration. The file is tied to the PRG it was built
// from (length + FNV-1a of the symbols) and to k; load_index throws std::runtime_error on any mismatch or damage.
void save_index(const HostIndex &h, const std::string &path);
void load_index(const std::string &path, const std::vector<uint32_t> &prg, uint32_t kmer_size, HostIndex &out);

// gram_dir/prg reader: little-endian uint32 per symbol (linearised_prg.cpp:8-45).
std::vector<uint32_t> read_prg_file(const std::string &path);

// ---- seed-table enumeration: what the host walk and the device walk (gmx_seedwalk.hip) share -----------------------
struct WalkState {  // one SearchState of the k-mer index walk: SA interval + handles of its traversed / traversing paths
  uint32_t lo, hi, tvd, tvg;
};
struct WalkNode {  // the states after some rightmost bases of a k-mer and the path nodes they point at
  std::vector<WalkState> list;
  std::vector<GmxPathNode> arena;
};
struct SeedEntryRef {  // a multi-state entry: its table index, its table (0: k, 1: k2), where its words start
  uint32_t code, table;
  uint64_t off;
};
// What a part of the enumeration hands to the join (gmx_index.cpp): the words of its multi-state entries in enumeration
// order — by the k-mers' right-to-left base order, a k entry before the k2 entries it is a suffix of — and the entries.
// The words live in the index's seed_words (HostIndex::seed_words) from word_base on, back to back (the device walk copies
// every group's words straight there; a host task's `words` are appended and freed when the task is done): no second copy.
struct SeedPart {
  std::vector<uint32_t> words;        // host walk only, until appended
  uint64_t word_base = 0, n_words = 0;  // this part's words in HostIndex::seed_words; SeedEntryRef::off is relative to word_base
  std::vector<SeedEntryRef> complex;
  uint64_t n_present[2] = {0, 0}, n_states_all[2] = {0, 0}, n_states_large[2] = {0, 0};
};
// The walk below the nodes `roots` (all of depth `depth0`; node i has the bases b_0 .. b_(depth0-1) from the k-mer's right
// end, b_0 in the two highest bits of i) on the device: fills both tables and the presence bitmap (host memory, every
// entry written) and returns the parts in order. Registered by gmx_seedwalk.hip when it is linked in (libgmx.so); null in
// host-only builds (tests/hostemu). Throws std::runtime_error; returns false when no device can be used.
typedef bool (*DeviceSeedWalk)(const HostIndex &ix, uint32_t k, uint32_t k2, uint32_t depth0, std::vector<WalkNode> &roots,
                               GmxSeed *table, GmxSeed *table2, uint32_t *bitmap, std::vector<SeedPart> &parts, WordBuf &words);
extern DeviceSeedWalk g_device_seed_walk;

// The suffixes of `text` (n symbols, the last one the unique sentinel 0) ordered by their first 24 symbols on the device
// (gmx_suffixsort.hip); tie_mask: bit p set = sa[p] still ties with sa[p - 1] — the host finishes those runs by comparison.
// Null in host-only builds; returns false when no device can be used; throws std::runtime_error.
typedef bool (*DeviceSuffixPresort)(const uint32_t *text, size_t n, uint32_t *sa, std::vector<uint32_t> &tie_mask);
extern DeviceSuffixPresort g_device_suffix_presort;

// Collects the k-mer index states of one k-mer from the seed table (test / debug helper).
// Output format: [n_states, {lo, hi, n_traversed, (site, allele)*, n_traversing, (site, -1)*}*] or {-1} if absent.
std::vector<int64_t> seed_states_of(const HostIndex &ix, uint32_t kmer_code, bool longer_table = false);

}  // namespace gmx
