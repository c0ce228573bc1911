// quasimap_oracle.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A dependency-free CPU restatement of the reference's quasimap path
// (gramtools v1.10.0, libgramtools/src/genotype/quasimap/** and the data
// structures it reads). It exists so that the HIP path can be checked for
// bit-identical results. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this file's shared object. The product
// (gramtools_amd/) never links, imports or calls it.
//
// Parity status: PINNED. The reference itself cannot be compiled here (needs
// SDSL-lite 2.1.1, Boost, htslib — absent, no network), so this restatement is
// pinned against the known-answer vectors of the reference's own tests,
// transcribed as data under tests/golden/ (see tests/test_oracle_golden.py).
// The third-party arithmetic it stands in for: SDSL-lite v2.1.1 csa_wt
// construction (suffix array of the integer text with a unique smallest
// sentinel 0 appended, BWT, C[] / char2comp / sigma) and rank_support_v<1>;
// libstdc++ std::mt19937 + std::uniform_int_distribution<uint32_t>.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/libgramtools/). Data-structure choices deliberately mirror
// the reference (std::list of states, std::map of equivalence classes) so that
// list ORDER is reproduced — the reference's unit tests compare list equality.

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <numeric>
#include <optional>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace gmo {

// ---------------------------------------------------------------------------
// Types: include/common/data_types.hpp:11-30,52-81; search/types.hpp:13-57
// ---------------------------------------------------------------------------
using Marker = uint32_t;
using AlleleId = int32_t;
constexpr AlleleId FIRST_ALLELE = 0;
constexpr AlleleId ALLELE_UNKNOWN = -1;
using VariantLocus = std::pair<Marker, AlleleId>;
using VariantSitePath = std::vector<VariantLocus>;
using SA_Index = uint32_t;
using SA_Interval = std::pair<SA_Index, SA_Index>;
using int_Base = uint8_t;
using Sequence = std::vector<int_Base>;
using CovCount = uint16_t;

static inline bool is_site_marker(Marker m) {
  if (!(m > 4)) throw std::invalid_argument("The given marker is not a variant marker (>4)");
  return m % 2 == 1;
}
static inline bool is_allele_marker(Marker m) { return !is_site_marker(m); }
static inline std::size_t siteID_to_index(Marker site_ID) {
  if (!is_site_marker(site_ID)) throw std::invalid_argument("The given marker is not a site ID");
  return (site_ID - 5) / 2;
}

struct SearchState {
  SA_Interval sa_interval = {};
  VariantSitePath traversed_path = {};
  VariantSitePath traversing_path = {};
  bool operator==(const SearchState &o) const {
    return sa_interval == o.sa_interval && traversed_path == o.traversed_path &&
           traversing_path == o.traversing_path;
  }
  bool has_path() const { return !traversed_path.empty() || !traversing_path.empty(); }
};
using SearchStates = std::list<SearchState>;

// ---------------------------------------------------------------------------
// PRG string: src/prg/linearised_prg.cpp:47-80 (map_ends_and_check_for_duplicates)
// ---------------------------------------------------------------------------
struct PRG_String {
  std::vector<Marker> prg;
  std::unordered_map<Marker, int> end_positions;

  explicit PRG_String(std::vector<Marker> v) : prg(std::move(v)) {
    std::set<Marker> seen_sites;
    for (std::size_t pos = 0; pos < prg.size(); ++pos) {
      Marker marker = prg[pos];
      if (marker == 0) throw std::runtime_error("PRG symbols must be >= 1");
      if (marker <= 4) continue;
      if (is_site_marker(marker)) {
        if (seen_sites.count(marker))
          throw std::runtime_error("PRG consistency error: site marker " + std::to_string(marker) +
                                   " used for two different sites");
        seen_sites.insert(marker);
      } else {
        end_positions[marker] = (int)pos;
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Coverage graph: src/prg/coverage_graph.cpp, include/prg/coverage_graph.hpp
// Nodes are held in a vector and referred to by index instead of shared_ptr.
// ---------------------------------------------------------------------------
struct coverage_Node {
  std::string sequence;
  Marker site_ID = 0;
  AlleleId allele_ID = ALLELE_UNKNOWN;
  std::size_t pos = 0;
  std::vector<CovCount> coverage;
  bool is_site_boundary = false;
  std::vector<int> next;

  bool has_sequence() const { return !sequence.empty(); }
  bool is_in_bubble() const { return allele_ID != ALLELE_UNKNOWN && site_ID != 0; }  // hpp:60-62
  bool is_bubble_start() const { return next.size() > 1 && sequence.empty(); }
  bool is_bubble_end() const { return next.size() == 1 && sequence.empty(); }
};

enum class marker_type { sequence, site_entry, allele_end, site_end };

struct node_access {
  int node = -1;
  std::size_t offset = 0;
  VariantLocus target = {0, ALLELE_UNKNOWN};
};

struct targeted_marker {
  Marker ID = 0;
  AlleleId direct_deletion_allele = ALLELE_UNKNOWN;
};

struct coverage_Graph {
  std::vector<coverage_Node> nodes;
  int root = -1;
  // bubble_map: ordered by std::greater on (pos, site_ID) — coverage_graph.cpp:381-389.
  std::vector<std::pair<int, int>> bubble_map;  // (site entry node, site exit node), in map order
  std::unordered_map<Marker, VariantLocus> par_map;
  std::vector<node_access> random_access;
  std::unordered_map<Marker, std::vector<targeted_marker>> target_map;
  bool is_nested = false;
};

static std::string decode_dna_base(Marker m) {
  switch (m) {
    case 1: return "A";
    case 2: return "C";
    case 3: return "G";
    case 4: return "T";
  }
  throw std::runtime_error("not a base");
}

// cov_Graph_Builder: coverage_graph.cpp:82-379
struct cov_Graph_Builder {
  std::vector<coverage_Node> nodes;
  std::vector<Marker> linear_prg;
  std::unordered_map<Marker, int> end_positions;
  int root = -1, backWire = -1, cur_Node = -1;
  std::size_t cur_pos = 0;
  bool first_allele = false;
  VariantLocus cur_Locus;
  std::unordered_map<Marker, int> bubble_starts, bubble_ends;
  std::vector<std::pair<int, int>> bubbles;  // insertion order; sorted at the end
  std::unordered_map<Marker, VariantLocus> par_map;
  std::vector<node_access> random_access;
  std::unordered_map<Marker, std::vector<targeted_marker>> target_map;

  int new_node(std::size_t pos) {  // coverage_Node(std::size_t pos), cpp:12-18
    coverage_Node n;
    n.pos = pos;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  }
  int new_node(const std::string &seq, int pos, int site_ID, int allele_ID) {  // cpp:20-31
    coverage_Node n;
    n.sequence = seq;
    n.pos = (std::size_t)pos;
    n.site_ID = (Marker)site_ID;
    n.allele_ID = allele_ID;
    if (n.is_in_bubble()) n.coverage.assign(seq.size(), 0);
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  }

  explicit cov_Graph_Builder(PRG_String const &prg_string) {  // cpp:82-95
    linear_prg = prg_string.prg;
    random_access.assign(linear_prg.size(), node_access());
    end_positions = prg_string.end_positions;
    make_root();
    cur_Locus = {0, ALLELE_UNKNOWN};
    for (uint32_t i = 0; i < linear_prg.size(); ++i) {
      process_marker(i);
      setup_random_access(i);
    }
    make_sink();
    map_targets();
  }

  void make_root() {  // cpp:97-103
    cur_pos = (std::size_t)-1;
    root = new_node(cur_pos);
    backWire = root;
    cur_pos++;
    cur_Node = new_node(cur_pos);
  }
  void make_sink() {  // cpp:105-110
    int sink = new_node(cur_pos + 1);
    wire(sink);
    cur_Node = -1;
    backWire = -1;
  }
  marker_type find_marker_type(uint32_t pos) {  // cpp:146-164
    Marker m = linear_prg[pos];
    if (m <= 4) return marker_type::sequence;
    if (m % 2 == 1) return marker_type::site_entry;
    auto end_pos = end_positions.at(m);
    if ((int)pos < end_pos) return marker_type::allele_end;
    return marker_type::site_end;
  }
  void process_marker(uint32_t pos) {  // cpp:112-129
    Marker m = linear_prg[pos];
    switch (find_marker_type(pos)) {
      case marker_type::sequence: add_sequence(m); break;
      case marker_type::site_entry: enter_site(m); break;
      case marker_type::allele_end: end_allele(m); break;
      case marker_type::site_end: exit_site(m); break;
    }
  }
  void setup_random_access(uint32_t pos) {  // cpp:131-144
    marker_type t = find_marker_type(pos);
    int target = (t == marker_type::sequence) ? cur_Node : backWire;
    auto seq_size = nodes[target].sequence.size();
    if (seq_size <= 1)
      random_access[pos] = node_access{target, 0, VariantLocus{0, ALLELE_UNKNOWN}};
    else
      random_access[pos] = node_access{target, seq_size - 1, VariantLocus{0, ALLELE_UNKNOWN}};
  }
  void add_sequence(Marker m) {  // cpp:166-172, coverage_Node::add_sequence cpp:33-38
    std::string c = decode_dna_base(m);
    coverage_Node &n = nodes[cur_Node];
    n.sequence += c;
    if (n.is_in_bubble()) n.coverage.emplace_back(0);
    cur_pos++;
  }
  void enter_site(Marker m) {  // cpp:174-197
    int site_entry = new_node("", (int)cur_pos, (int)m, ALLELE_UNKNOWN);
    nodes[site_entry].is_site_boundary = true;
    wire(site_entry);
    cur_Node = new_node("", (int)cur_pos, (int)m, FIRST_ALLELE);
    first_allele = true;
    backWire = site_entry;
    int site_exit = new_node("", (int)cur_pos, (int)m, ALLELE_UNKNOWN);
    nodes[site_exit].is_site_boundary = true;
    bubbles.emplace_back(site_entry, site_exit);
    bubble_starts.insert({m, site_entry});
    bubble_ends.insert({m, site_exit});
    if (cur_Locus.first != 0) par_map.insert({m, cur_Locus});
    cur_Locus = {m, FIRST_ALLELE};
  }
  void end_allele(Marker m) {  // cpp:199-213
    Marker site_ID = m - 1;
    reach_allele_end(m);
    int site_entry = bubble_starts.at(site_ID);
    backWire = site_entry;
    cur_pos = nodes[site_entry].pos;
    cur_Locus.second++;
    cur_Node = new_node("", (int)cur_pos, (int)site_ID, cur_Locus.second);
  }
  void exit_site(Marker m) {  // cpp:215-236
    Marker site_ID = m - 1;
    int site_exit = reach_allele_end(m);
    if (cur_Locus.second == FIRST_ALLELE)
      throw std::runtime_error("Site numbered " + std::to_string(m) + " has only one allele");
    if (par_map.find(site_ID) != par_map.end()) {
      cur_Locus = par_map.at(site_ID);
      if (cur_Locus.second == FIRST_ALLELE) first_allele = true;
    } else
      cur_Locus = {0, ALLELE_UNKNOWN};
    backWire = site_exit;
    cur_pos = nodes[site_exit].pos;
    cur_Node = new_node("", (int)cur_pos, (int)cur_Locus.first, cur_Locus.second);
  }
  int reach_allele_end(Marker m) {  // cpp:238-258
    Marker site_ID = m - 1;
    if (cur_Locus.first != site_ID) throw std::runtime_error("PRG consistency error: unbalanced site markers");
    int site_exit = bubble_ends.at(site_ID);
    wire(site_exit);
    if (first_allele) {
      nodes[site_exit].pos = cur_pos;
      first_allele = false;
    }
    return site_exit;
  }
  void wire(int target) {  // cpp:260-266
    if (nodes[cur_Node].has_sequence()) {
      nodes[backWire].next.emplace_back(cur_Node);
      nodes[cur_Node].next.emplace_back(target);
    } else
      nodes[backWire].next.emplace_back(target);
  }

  void map_targets() {  // cpp:268-311
    marker_type prev_t = marker_type::sequence;
    Marker prev_m = 0;
    AlleleId cur_allele_ID = ALLELE_UNKNOWN;
    for (std::size_t pos = 0; pos < linear_prg.size(); ++pos) {
      Marker cur_m = linear_prg[pos];
      marker_type cur_t = find_marker_type((uint32_t)pos);
      switch (cur_t) {
        case marker_type::sequence:
          if (prev_t != marker_type::sequence) random_access[pos].target = VariantLocus{prev_m, cur_allele_ID};
          break;
        case marker_type::site_entry:
          cur_allele_ID = FIRST_ALLELE;
          if (prev_t != marker_type::sequence) make_site_entry_target(prev_t, prev_m, cur_m);
          break;
        case marker_type::site_end:
          if (prev_t != marker_type::sequence) make_site_exit_target(prev_t, prev_m, cur_m, cur_allele_ID);
          if (par_map.find(cur_m - 1) != par_map.end())
            cur_allele_ID = par_map.at(cur_m - 1).second;
          else
            cur_allele_ID = ALLELE_UNKNOWN;
          break;
        case marker_type::allele_end:
          if (prev_t != marker_type::sequence) make_allele_end_target(prev_t, prev_m, cur_m, cur_allele_ID);
          cur_allele_ID++;
          break;
      }
      prev_m = cur_m;
      prev_t = cur_t;
    }
  }
  void make_site_entry_target(marker_type prev_t, Marker prev_m, Marker cur_m) {  // cpp:313-328
    Marker marker_target = prev_m;
    if (prev_t == marker_type::allele_end) marker_target -= 1;
    target_map.insert({cur_m, {targeted_marker{marker_target, ALLELE_UNKNOWN}}});
  }
  void make_site_exit_target(marker_type prev_t, Marker prev_m, Marker cur_m, AlleleId cur_allele_ID) {  // cpp:330-350
    Marker marker_target = prev_m;
    AlleleId dd = ALLELE_UNKNOWN;
    switch (prev_t) {
      case marker_type::site_entry:
        throw std::runtime_error("PRG consistency error: site number " + std::to_string(cur_m) + " is empty");
      case marker_type::site_end: break;
      case marker_type::allele_end:
        marker_target -= 1;
        dd = cur_allele_ID;
        break;
      default: break;
    }
    add_exit_target(cur_m, targeted_marker{marker_target, dd});
  }
  void make_allele_end_target(marker_type prev_t, Marker prev_m, Marker cur_m, AlleleId cur_allele_ID) {  // cpp:352-369
    Marker marker_target = prev_m;
    AlleleId dd = cur_allele_ID;
    switch (prev_t) {
      case marker_type::site_entry: break;
      case marker_type::site_end: dd = ALLELE_UNKNOWN; break;
      case marker_type::allele_end: marker_target -= 1; break;
      default: break;
    }
    add_exit_target(cur_m, targeted_marker{marker_target, dd});
  }
  void add_exit_target(Marker cur_m, targeted_marker t) {  // cpp:371-379
    target_map[cur_m].emplace_back(t);
  }
};

static coverage_Graph make_coverage_graph(PRG_String const &p) {  // cpp:59-68
  cov_Graph_Builder b(p);
  coverage_Graph g;
  g.nodes = std::move(b.nodes);
  g.root = b.root;
  g.par_map = std::move(b.par_map);
  g.random_access = std::move(b.random_access);
  g.target_map = std::move(b.target_map);
  g.is_nested = !g.par_map.empty();
  // std::map<covG_ptr, covG_ptr, std::greater<covG_ptr>> ordering (cpp:381-389): larger pos first,
  // then larger site_ID first. Two bubble starts never share both.
  g.bubble_map = std::move(b.bubbles);
  std::sort(g.bubble_map.begin(), g.bubble_map.end(), [&](auto const &l, auto const &r) {
    auto const &a = g.nodes[l.first];
    auto const &c = g.nodes[r.first];
    if (a.pos != c.pos) return a.pos > c.pos;
    return a.site_ID > c.site_ID;
  });
  return g;
}

// ---------------------------------------------------------------------------
// FM-index (stands in for sdsl::csa_wt<wt_int,1,...>, make_data_structures.cpp:9-33)
// ---------------------------------------------------------------------------
struct FM_Index {
  std::vector<uint32_t> text;  // prg + sentinel 0
  std::vector<uint32_t> sa;    // full suffix array (SA sampling density 1, data_types.hpp:35-37)
  std::vector<uint32_t> bwt;
  std::vector<uint32_t> alphabet;  // sorted distinct symbols (comp -> char)
  std::unordered_map<uint32_t, uint32_t> char2comp_map;
  std::vector<uint64_t> C;  // C[comp] = number of symbols smaller; C[sigma] = size
  uint32_t sigma = 0;
  std::size_t size() const { return text.size(); }
  uint32_t char2comp(uint32_t c) const {
    auto it = char2comp_map.find(c);
    return it == char2comp_map.end() ? 0 : it->second;  // SDSL: unseen symbol -> 0
  }
  uint32_t operator[](std::size_t i) const { return sa[i]; }
};

// Suffix array by prefix doubling (O(n log^2 n)); the SA of a text ending in a
// unique smallest sentinel is unique, so any correct construction equals SDSL's.
static std::vector<uint32_t> build_sa(std::vector<uint32_t> const &text) {
  std::size_t n = text.size();
  std::vector<uint32_t> sa(n), rnk(n), tmp(n);
  std::iota(sa.begin(), sa.end(), 0u);
  {
    std::vector<uint32_t> sorted(text);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    for (std::size_t i = 0; i < n; ++i)
      rnk[i] = (uint32_t)(std::lower_bound(sorted.begin(), sorted.end(), text[i]) - sorted.begin());
  }
  for (std::size_t k = 1;; k <<= 1) {
    auto key = [&](uint32_t i) -> uint64_t {
      uint64_t second = (i + k < n) ? (uint64_t)rnk[i + k] + 1 : 0;
      return ((uint64_t)rnk[i] << 32) | second;
    };
    std::sort(sa.begin(), sa.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
    tmp[sa[0]] = 0;
    for (std::size_t i = 1; i < n; ++i) tmp[sa[i]] = tmp[sa[i - 1]] + (key(sa[i - 1]) < key(sa[i]) ? 1 : 0);
    rnk = tmp;
    if (rnk[sa[n - 1]] == n - 1) break;
  }
  return sa;
}

static FM_Index build_fm_index(std::vector<Marker> const &prg) {
  FM_Index fm;
  fm.text = prg;
  fm.text.push_back(0);
  fm.sa = build_sa(fm.text);
  std::size_t n = fm.text.size();
  fm.bwt.resize(n);
  for (std::size_t i = 0; i < n; ++i) fm.bwt[i] = fm.sa[i] == 0 ? fm.text[n - 1] : fm.text[fm.sa[i] - 1];
  fm.alphabet = fm.text;
  std::sort(fm.alphabet.begin(), fm.alphabet.end());
  fm.alphabet.erase(std::unique(fm.alphabet.begin(), fm.alphabet.end()), fm.alphabet.end());
  fm.sigma = (uint32_t)fm.alphabet.size();
  for (uint32_t c = 0; c < fm.sigma; ++c) fm.char2comp_map[fm.alphabet[c]] = c;
  fm.C.assign(fm.sigma + 1, 0);
  for (auto s : fm.text) fm.C[fm.char2comp_map[s] + 1]++;
  for (uint32_t c = 0; c < fm.sigma; ++c) fm.C[c + 1] += fm.C[c];
  return fm;
}

// A bit vector with rank support (stands in for sdsl::bit_vector + rank_support_v<1>):
// rank(i) = number of set bits in [0, i).
struct RankBitVector {
  std::vector<uint64_t> words;
  std::vector<uint64_t> cum;  // set bits before each word
  std::size_t nbits = 0;
  void init(std::size_t n) {
    nbits = n;
    words.assign(n / 64 + 1, 0);
  }
  void set(std::size_t i) { words[i >> 6] |= (1ull << (i & 63)); }
  bool get(std::size_t i) const { return (words[i >> 6] >> (i & 63)) & 1ull; }
  void finalise() {
    cum.assign(words.size() + 1, 0);
    for (std::size_t w = 0; w < words.size(); ++w) cum[w + 1] = cum[w] + (uint64_t)__builtin_popcountll(words[w]);
  }
  uint64_t rank(std::size_t i) const {
    std::size_t w = i >> 6, r = i & 63;
    uint64_t res = cum[w];
    if (r) res += (uint64_t)__builtin_popcountll(words[w] & ((1ull << r) - 1));
    return res;
  }
};

// ---------------------------------------------------------------------------
// PRG_Info: include/prg/prg_info.hpp:22-59; load_prg_info src/prg/prg_info.cpp:6-29;
// masks src/prg/make_data_structures.cpp:78-95,158-163
// ---------------------------------------------------------------------------
struct PRG_Info {
  FM_Index fm_index;
  std::vector<Marker> encoded_prg;
  std::unordered_map<Marker, int> last_allele_positions;
  mutable coverage_Graph coverage_graph;
  RankBitVector bwt_markers_mask;
  RankBitVector mask[5];  // [1..4] = a,c,g,t
  uint64_t num_variant_sites = 0;
};

static PRG_Info generate_prg_info(std::vector<Marker> const &prg_raw) {  // submods/submod_resources.cpp:21-62
  PRG_String ps{prg_raw};
  PRG_Info info;
  info.encoded_prg = ps.prg;
  info.fm_index = build_fm_index(ps.prg);
  info.coverage_graph = make_coverage_graph(ps);
  info.last_allele_positions = ps.end_positions;
  std::size_t n = info.fm_index.bwt.size();
  info.bwt_markers_mask.init(n);
  for (int b = 1; b <= 4; ++b) info.mask[b].init(n);
  for (std::size_t i = 0; i < n; ++i) {
    uint32_t c = info.fm_index.bwt[i];
    if (c > 4) info.bwt_markers_mask.set(i);           // make_data_structures.cpp:158-163
    if (c >= 1 && c <= 4) info.mask[c].set(i);         // make_data_structures.cpp:78-95
  }
  info.bwt_markers_mask.finalise();
  for (int b = 1; b <= 4; ++b) info.mask[b].finalise();
  info.num_variant_sites = info.coverage_graph.bubble_map.size();
  return info;
}

// ---------------------------------------------------------------------------
// BWT search: src/genotype/quasimap/search/BWT_search.cpp
// ---------------------------------------------------------------------------
static uint64_t dna_bwt_rank(uint64_t upper_index, Marker dna_base, PRG_Info const &prg_info) {  // :8-22
  if (dna_base >= 1 && dna_base <= 4) return prg_info.mask[dna_base].rank(upper_index);
  return 0;
}

static SA_Interval base_next_sa_interval(Marker next_char, SA_Index next_char_first_sa_index,
                                         SA_Interval const &cur, PRG_Info const &prg_info) {  // :45-76
  SA_Index sa_start_offset;
  if (cur.first <= 0)
    sa_start_offset = 0;
  else
    sa_start_offset = (SA_Index)dna_bwt_rank(cur.first, next_char, prg_info);
  SA_Index sa_end_offset = (SA_Index)dna_bwt_rank((uint64_t)cur.second + 1, next_char, prg_info);
  SA_Index new_start = next_char_first_sa_index + sa_start_offset;
  SA_Index new_end = next_char_first_sa_index + sa_end_offset - 1;  // uint32 arithmetic
  return SA_Interval{new_start, new_end};
}

static std::optional<SearchState> search_fm_index_base_backwards(int_Base pattern_char, uint64_t char_first_sa_index,
                                                                 SearchState const &ss, PRG_Info const &prg_info) {  // :28-43
  auto next = base_next_sa_interval(pattern_char, (SA_Index)char_first_sa_index, ss.sa_interval, prg_info);
  bool valid = (SA_Index)(next.first - 1) != next.second;
  if (!valid) return {};
  SearchState ns = ss;
  ns.sa_interval = next;
  return ns;
}

static SearchStates search_base_backwards(int_Base pattern_char, SearchStates const &search_states,
                                          PRG_Info const &prg_info) {  // :78-94
  auto char_alphabet_rank = prg_info.fm_index.char2comp(pattern_char);
  auto char_first_sa_index = prg_info.fm_index.C[char_alphabet_rank];
  SearchStates out;
  for (auto const &ss : search_states) {
    auto ns = search_fm_index_base_backwards(pattern_char, char_first_sa_index, ss, prg_info);
    if (ns) out.push_back(std::move(*ns));
  }
  return out;
}

// ---------------------------------------------------------------------------
// vBWT jumps: src/genotype/quasimap/search/vBWT_jump.cpp
// ---------------------------------------------------------------------------
static SA_Interval get_allele_marker_sa_interval(Marker allele_marker_char, PRG_Info const &prg_info) {  // :3-21
  const auto alphabet_rank = prg_info.fm_index.char2comp(allele_marker_char);
  const auto start_sa_index = (SA_Index)prg_info.fm_index.C[alphabet_rank];
  SA_Index end_sa_index;
  if (alphabet_rank < prg_info.fm_index.sigma - 1)
    end_sa_index = (SA_Index)(prg_info.fm_index.C[alphabet_rank + 1] - 1);
  else
    end_sa_index = (SA_Index)(prg_info.fm_index.size() - 1);
  return SA_Interval{start_sa_index, end_sa_index};
}

static SearchState entering_site_search_state(Marker allele_marker, SearchState const &cur, PRG_Info const &prg_info) {  // :29-44
  auto iv = get_allele_marker_sa_interval(allele_marker, prg_info);
  SearchState ns = cur;
  ns.sa_interval = iv;
  ns.traversing_path.push_back(VariantLocus{allele_marker - 1, ALLELE_UNKNOWN});
  return ns;
}

static void update_variant_site_path(SearchState &s, AlleleId allele_id, Marker site_ID) {  // :51-69
  bool started_in_site = s.traversing_path.empty();
  if (started_in_site) {
    s.traversed_path.push_back(VariantLocus{site_ID, allele_id});
  } else {
    auto existing_locus = s.traversing_path.back();
    if (existing_locus.first != site_ID || existing_locus.second != ALLELE_UNKNOWN)
      throw std::logic_error("update_variant_site_path: traversing path tail does not match exited site");
    existing_locus.second = allele_id;
    s.traversed_path.push_back(existing_locus);
    s.traversing_path.pop_back();
  }
}

static SearchState exiting_site_search_state(VariantLocus const &locus, SearchState const &cur, PRG_Info const &prg_info) {  // :76-92
  SearchState ns = cur;
  Marker site_marker = locus.first;
  AlleleId allele_id = locus.second;
  update_variant_site_path(ns, allele_id, site_marker);
  auto alphabet_rank = prg_info.fm_index.char2comp(site_marker);
  SA_Index site_index = (SA_Index)prg_info.fm_index.C[alphabet_rank];
  ns.sa_interval = SA_Interval{site_index, site_index};
  return ns;
}

using MarkersSearchResults = std::vector<VariantLocus>;

static MarkersSearchResults left_markers_search(SearchState const &ss, PRG_Info const &prg_info) {  // :94-117
  MarkersSearchResults res;
  auto const &iv = ss.sa_interval;
  for (int64_t index = iv.first; index <= (int64_t)iv.second; index++) {
    if (!prg_info.bwt_markers_mask.get((std::size_t)index)) continue;
    auto prg_index = prg_info.fm_index[(std::size_t)index];
    VariantLocus target_locus = prg_info.coverage_graph.random_access[prg_index].target;
    if (is_allele_marker(target_locus.first)) {
      if (prg_info.last_allele_positions.at(target_locus.first) != (int)prg_index - 1) target_locus.first--;
    }
    res.push_back(target_locus);
  }
  return res;
}

struct Locus_and_SearchState {
  VariantLocus locus;
  SearchState search_state;
  bool commit_me = false;
};
using Locus_and_SearchStates = std::vector<Locus_and_SearchState>;

static Locus_and_SearchState extend_targets_site_exit(VariantLocus const &target_locus, SearchState const &ss,
                                                      PRG_Info const &prg_info) {  // :185-228
  VariantLocus next_target = target_locus;
  auto site_marker = next_target.first;
  bool commit_me = true;
  auto &target_map = prg_info.coverage_graph.target_map;
  auto ns = exiting_site_search_state(target_locus, ss, prg_info);
  next_target = VariantLocus{0, 0};
  while (target_map.find(site_marker) != target_map.end()) {
    auto target_markers = target_map.at(site_marker);
    if (target_markers.size() != 1) throw std::logic_error("site entry point with more than one target");
    auto next_site_marker = target_markers.back().ID;
    if (is_allele_marker(next_site_marker)) {
      next_target = VariantLocus{next_site_marker, 0};
      commit_me = false;
      break;
    } else {
      auto parent_site = prg_info.coverage_graph.par_map.at(site_marker);
      if (parent_site.first != next_site_marker) throw std::logic_error("double exit not in parental map");
      auto allele_id = parent_site.second;
      ns = exiting_site_search_state(VariantLocus{next_site_marker, allele_id}, ns, prg_info);
      site_marker = next_site_marker;
    }
  }
  return Locus_and_SearchState{next_target, ns, commit_me};
}

static Locus_and_SearchStates extend_targets_site_entry(VariantLocus const &target_locus, SearchState const &ss,
                                                        PRG_Info const &prg_info) {  // :230-265
  Locus_and_SearchStates extensions;
  auto variant_marker = target_locus.first;
  auto ns = entering_site_search_state(target_locus.first, ss, prg_info);
  extensions.push_back({VariantLocus{0, 0}, ns, true});
  auto &target_map = prg_info.coverage_graph.target_map;
  if (target_map.find(variant_marker) == target_map.end()) return extensions;
  for (auto &mapped_target : target_map.at(variant_marker)) {
    if (is_site_marker(mapped_target.ID)) {
      VariantLocus site_exit_locus{mapped_target.ID, mapped_target.direct_deletion_allele};
      extensions.push_back({site_exit_locus, ns, false});
    } else {
      VariantLocus site_entry_locus{mapped_target.ID, ALLELE_UNKNOWN};
      extensions.push_back({site_entry_locus, ns, false});
    }
  }
  return extensions;
}

static SearchStates search_state_vBWT_jumps(SearchState const &cur, PRG_Info const &prg_info) {  // :134-183
  auto marker_targets = left_markers_search(cur, prg_info);
  if (marker_targets.empty()) return SearchStates{};
  SearchStates markers_search_states;
  Locus_and_SearchStates extension_targets;
  Locus_and_SearchStates to_process_targets;
  for (auto &mt : marker_targets) to_process_targets.push_back({mt, cur, false});
  while (!to_process_targets.empty()) {
    auto const to_process_target = to_process_targets.back();
    to_process_targets.pop_back();
    auto const &target_locus = to_process_target.locus;
    auto const &search_state = to_process_target.search_state;
    if (is_site_marker(target_locus.first)) {
      auto new_target = extend_targets_site_exit(target_locus, search_state, prg_info);
      extension_targets = Locus_and_SearchStates{new_target};
    } else {
      extension_targets = extend_targets_site_entry(target_locus, search_state, prg_info);
    }
    for (auto &new_target : extension_targets) {
      if (new_target.commit_me) markers_search_states.push_back(new_target.search_state);
      auto const &site_ID = new_target.locus.first;
      if (site_ID != 0) to_process_targets.push_back(new_target);
    }
  }
  return markers_search_states;
}

static void process_markers_search_states(SearchStates &current, PRG_Info const &prg_info) {  // :119-132
  SearchStates all_markers_search_states;
  for (auto const &ss : current) {
    auto ms = search_state_vBWT_jumps(ss, prg_info);
    if (!ms.empty()) all_markers_search_states.splice(all_markers_search_states.end(), ms);
  }
  current.splice(current.end(), all_markers_search_states);
}

// ---------------------------------------------------------------------------
// Encapsulated search: src/genotype/quasimap/search/encapsulated_search.cpp
// ---------------------------------------------------------------------------
static SearchStates handle_allele_encapsulated_state(SearchState const &ss, PRG_Info const &prg_info) {  // :30-88
  SearchStates out;
  SearchState cache;
  bool cache_empty = true;
  auto flush = [&]() {
    if (cache_empty) return;
    out.emplace_back(cache);
    cache_empty = true;
  };
  for (uint64_t sa_index = ss.sa_interval.first; sa_index <= ss.sa_interval.second; ++sa_index) {
    auto prg_index = prg_info.fm_index[sa_index];
    auto const &cov_node = prg_info.coverage_graph.nodes[prg_info.coverage_graph.random_access[prg_index].node];
    auto site_marker = cov_node.site_ID;
    auto allele_id = cov_node.allele_ID;
    bool within_site = site_marker != 0;
    if (!within_site) {
      flush();
      cache = SearchState{SA_Interval{(SA_Index)sa_index, (SA_Index)sa_index}, {}, {}};
      cache_empty = false;
      flush();
      continue;
    }
    if (cache_empty) {
      cache = SearchState{SA_Interval{(SA_Index)sa_index, (SA_Index)sa_index},
                          VariantSitePath{VariantLocus{site_marker, allele_id}}, {}};
      cache_empty = false;
      continue;
    }
    VariantSitePath current_path = {VariantLocus{site_marker, allele_id}};
    if (current_path == cache.traversed_path) {
      cache.sa_interval.second = (SA_Index)sa_index;
      continue;
    } else {
      flush();
      cache = SearchState{SA_Interval{(SA_Index)sa_index, (SA_Index)sa_index}, current_path, {}};
      cache_empty = false;
    }
  }
  flush();
  return out;
}

static SearchStates handle_allele_encapsulated_states(SearchStates const &states, PRG_Info const &prg_info) {  // :90-107
  SearchStates out;
  for (auto const &ss : states) {
    if (ss.has_path()) {
      out.emplace_back(ss);
      continue;
    }
    for (auto const &s : handle_allele_encapsulated_state(ss, prg_info)) out.emplace_back(s);
  }
  return out;
}

// ---------------------------------------------------------------------------
// k-mer index: src/build/kmer_index/{build.cpp,kmers.cpp}
// ---------------------------------------------------------------------------
struct SeqHash {
  std::size_t operator()(Sequence const &s) const {
    std::size_t h = 0;
    for (auto b : s) h ^= (std::size_t)b + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h;
  }
};
using KmerIndex = std::unordered_map<Sequence, SearchStates, SeqHash>;
struct CacheElement {
  SearchStates search_states;
  int_Base base = 0;
};
using KmerIndexCache = std::list<CacheElement>;

static SearchStates process_read_char_search_states(int_Base pattern_char, SearchStates &states,
                                                    PRG_Info const &prg_info);  // fwd

static CacheElement get_next_cache_element(int_Base base, bool kmer_base_is_first_processed,
                                           CacheElement const &last, PRG_Info const &prg_info) {  // build.cpp:18-28
  SearchStates ns = last.search_states;
  if (!kmer_base_is_first_processed) process_markers_search_states(ns, prg_info);
  ns = search_base_backwards(base, ns, prg_info);
  return CacheElement{ns, base};
}
static CacheElement get_initial_cache_element(int_Base base, PRG_Info const &prg_info) {  // build.cpp:35-46
  SearchState ss;
  ss.sa_interval = SA_Interval{0, (SA_Index)(prg_info.fm_index.size() - 1)};
  CacheElement full{SearchStates{ss}, 0};
  return get_next_cache_element(base, true, full, prg_info);
}
static void build_kmer_cache(KmerIndexCache &cache, Sequence const &kmer_prefix_diff, int kmer_size,
                             PRG_Info const &prg_info) {  // build.cpp:55-86
  auto it = kmer_prefix_diff.rbegin();
  if ((int)kmer_prefix_diff.size() == kmer_size) {
    cache.resize(0);
    cache.emplace_back(get_initial_cache_element(*it, prg_info));
    ++it;
  } else {
    cache.resize(kmer_size - kmer_prefix_diff.size());
  }
  for (; it != kmer_prefix_diff.rend(); ++it) {
    auto &last = cache.back();
    cache.emplace_back(get_next_cache_element(*it, false, last, prg_info));
  }
}
static void update_full_kmer(Sequence &full_kmer, Sequence const &diff, int kmer_size) {  // build.cpp:91-99
  if ((int)diff.size() == kmer_size) {
    full_kmer = diff;
    return;
  }
  std::size_t i = 0;
  for (auto b : diff) full_kmer[i++] = b;
}
static KmerIndex index_kmers(std::vector<Sequence> const &kmer_prefix_diffs, int kmer_size, PRG_Info const &prg_info) {  // build.cpp:101-131
  KmerIndex kmer_index;
  KmerIndexCache cache;
  Sequence full_kmer;
  for (auto const &diff : kmer_prefix_diffs) {
    update_full_kmer(full_kmer, diff, kmer_size);
    build_kmer_cache(cache, diff, kmer_size, prg_info);
    auto const &last = cache.back();
    if (!last.search_states.empty()) kmer_index[full_kmer] = last.search_states;
  }
  return kmer_index;
}
// kmers.cpp:23-105. generate_all_kmers enumerates 1..4^k in lexicographic order of the
// *reversed* k-mer (ordered set), then reverses each, then takes prefix diffs.
static std::vector<Sequence> get_all_kmers(uint64_t k) {
  std::vector<Sequence> kmers;
  Sequence cur(k, 1);
  while (true) {
    Sequence rev(cur.rbegin(), cur.rend());
    kmers.push_back(rev);
    int64_t idx = (int64_t)k - 1;
    while (idx >= 0 && cur[idx] == 4) idx--;
    if (idx < 0) break;
    cur[idx]++;
    for (uint64_t i = idx + 1; i < k; ++i) cur[i] = 1;
  }
  return kmers;
}
static std::vector<Sequence> get_prefix_diffs(std::vector<Sequence> const &kmers) {  // kmers.cpp:42-73
  std::vector<Sequence> diffs;
  Sequence last;
  for (auto const &kmer : kmers) {
    if (last.empty()) {
      last = kmer;
      diffs.push_back(last);
      continue;
    }
    bool found = false;
    std::list<int_Base> d;
    for (int64_t i = (int64_t)last.size() - 1; i >= 0; --i) {
      if (kmer[i] != last[i]) found = true;
      if (found) d.push_front(kmer[i]);
    }
    last = kmer;
    diffs.emplace_back(d.begin(), d.end());
  }
  return diffs;
}

// ---------------------------------------------------------------------------
// RNG: src/common/random.cpp:4-19, include/common/random.hpp:14-38
// mt19937 restated; uniform_int_distribution<uint32_t> in its two libstdc++ forms.
// ---------------------------------------------------------------------------
struct MT19937 {
  uint32_t mt[624];
  int idx = 624;
  explicit MT19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
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
  }
};
enum RngMode { RNG_LEMIRE = 0 /* libstdc++ >= 11 */, RNG_DIVISION = 1 /* libstdc++ <= 10 */ };
// std::uniform_int_distribution<uint32_t>(min,max)(mt19937): urng range is 2^32.
static uint32_t uniform_u32(MT19937 &g, uint32_t min, uint32_t max, int mode) {
  uint32_t urange = max - min;
  if (urange == 0xffffffffu) return g.next() + min;
  uint32_t uerange = urange + 1;
  if (mode == RNG_LEMIRE) {  // /usr/include/c++/11/bits/uniform_int_dist.h:240-268 (_S_nd), :300-307
    uint64_t product = (uint64_t)g.next() * (uint64_t)uerange;
    uint32_t low = (uint32_t)product;
    if (low < uerange) {
      uint32_t threshold = (uint32_t)(-uerange) % uerange;
      while (low < threshold) {
        product = (uint64_t)g.next() * (uint64_t)uerange;
        low = (uint32_t)product;
      }
    }
    return (uint32_t)(product >> 32) + min;
  } else {  // GCC <= 10: scaling by division with rejection
    const uint64_t urngrange = 0xffffffffull;
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = (uint64_t)uerange * scaling;
    uint64_t ret;
    do ret = g.next();
    while (ret >= past);
    return (uint32_t)(ret / scaling) + min;
  }
}

// ---------------------------------------------------------------------------
// Coverage: types.hpp:15-46
// ---------------------------------------------------------------------------
using AlleleIds = std::vector<AlleleId>;
using GroupedAlleleCounts = std::map<AlleleIds, CovCount>;  // reference: unordered_map; order is not semantic
struct Coverage {
  std::vector<std::vector<CovCount>> allele_sum_coverage;
  std::vector<GroupedAlleleCounts> grouped_allele_counts;
};

static Coverage empty_structure(PRG_Info const &prg_info) {  // coverage_common.cpp:206-213, allele_sum.cpp:10-29
  Coverage c;
  c.grouped_allele_counts.assign(prg_info.num_variant_sites, {});
  c.allele_sum_coverage.assign(prg_info.num_variant_sites, {});
  for (auto const &b : prg_info.coverage_graph.bubble_map) {
    auto const &n = prg_info.coverage_graph.nodes[b.first];
    auto site_index = siteID_to_index(n.site_ID);
    // The reference indexes a vector of num_variant_sites entries (allele_sum.cpp:21-27): site markers must be
    // the contiguous odd numbers 5,7,9,... Graph-only test vectors with sparse numbering get no coverage arrays.
    if (site_index >= c.allele_sum_coverage.size())
      throw std::out_of_range("site markers are not numbered contiguously from 5; coverage structures undefined");
    for (std::size_t i = 0; i < n.next.size(); ++i) c.allele_sum_coverage[site_index].push_back(0);
  }
  return c;
}

using level0_Sites = std::set<Marker>;
using uniqueLoci = std::set<VariantLocus>;

// LocusFinder: coverage_common.cpp:10-83
struct LocusFinder {
  level0_Sites base_sites;
  std::set<Marker> used_sites;
  uniqueLoci unique_loci;

  LocusFinder() = default;
  LocusFinder(SearchState const &ss, PRG_Info const *info) {
    check_site_uniqueness(ss);
    assign_traversing_loci(ss, info);
    assign_traversed_loci(ss, info);
  }
  static void check_site_uniqueness(SearchState const &ss) {  // :17-32
    auto all = ss.traversed_path;
    all.insert(all.end(), ss.traversing_path.begin(), ss.traversing_path.end());
    std::set<Marker> uniq;
    for (auto const &e : all) {
      if (uniq.count(e.first))
        throw std::logic_error("ERROR: A site cannot have been traversed more than once by a read");
      uniq.insert(e.first);
    }
  }
  void assign_nested_locus(VariantLocus const &var_loc, PRG_Info const *info) {  // :34-51
    auto &par_map = info->coverage_graph.par_map;
    VariantLocus cur = var_loc;
    while (true) {
      if (used_sites.count(cur.first)) break;
      used_sites.insert(cur.first);
      unique_loci.insert(cur);
      if (par_map.find(cur.first) == par_map.end()) {
        base_sites.insert(cur.first);
        break;
      }
      cur = par_map.at(cur.first);
    }
  }
  void assign_traversing_loci(SearchState const &ss, PRG_Info const *info) {  // :53-76
    if (ss.traversing_path.empty()) return;
    Marker parent_seed = ss.traversing_path.back().first;
    VariantLocus new_locus;
    for (int64_t i = ss.sa_interval.first; i <= (int64_t)ss.sa_interval.second; ++i) {
      auto prg_pos = info->fm_index[(std::size_t)i];
      auto &na = info->coverage_graph.random_access[prg_pos];
      auto allele_id = info->coverage_graph.nodes[na.node].allele_ID;
      new_locus = VariantLocus{parent_seed, allele_id};
      unique_loci.insert(new_locus);
    }
    assign_nested_locus(new_locus, info);
  }
  void assign_traversed_loci(SearchState const &ss, PRG_Info const *info) {  // :78-83
    for (auto const &l : ss.traversed_path) assign_nested_locus(l, info);
  }
};

using traversal_info = std::pair<SearchStates, uniqueLoci>;
using uniqueSitePaths = std::map<level0_Sites, traversal_info>;
struct SelectedMapping {
  SearchStates navigational_search_states;
  uniqueLoci equivalence_class_loci;
};

// MappingInstanceSelector: coverage_common.cpp:85-146. `draw(min,max)` replaces the RandomGenerator*.
template <class Draw>
static SelectedMapping select_mapping(SearchStates const &search_states, PRG_Info const *info, Draw &&draw,
                                      uniqueSitePaths *usps_out = nullptr) {
  uniqueSitePaths usps;
  for (auto const &ss : search_states) {  // process_searchstates :124-128, add_searchstate :110-122
    if (!ss.has_path()) continue;
    LocusFinder l{ss, info};
    auto &cov_info = usps[l.base_sites];
    for (auto &locus : l.unique_loci) cov_info.second.insert(locus);
    cov_info.first.push_back(ss);
  }
  if (usps_out) *usps_out = usps;
  SelectedMapping selected;
  if (usps.size() == 0) return selected;  // random_select_entry :96-97
  uint32_t nonvariant_count = 0;          // count_nonvar_search_states :130-141
  for (auto const &ss : search_states)
    if (!ss.has_path()) nonvariant_count += (ss.sa_interval.second - ss.sa_interval.first + 1);
  uint32_t count_total_options = nonvariant_count + (uint32_t)usps.size();
  uint32_t selected_option = draw(1u, count_total_options);
  if (selected_option <= nonvariant_count) return selected;
  int32_t idx = (int32_t)(selected_option - nonvariant_count - 1);
  auto it = usps.begin();
  std::advance(it, idx);
  selected.navigational_search_states = it->second.first;
  selected.equivalence_class_loci = it->second.second;
  return selected;
}

// Per-base coverage: src/genotype/quasimap/coverage/allele_base.cpp:109-296
using node_coordinate = uint32_t;
using node_coordinates = std::pair<node_coordinate, node_coordinate>;
struct DummyCovNode {  // :109-135
  bool full = false;
  node_coordinate start_pos = 0, end_pos = 0;
  std::size_t node_size = 0;
  DummyCovNode() = default;
  DummyCovNode(node_coordinate s, node_coordinate e, std::size_t sz) : start_pos(s), end_pos(e), node_size(sz) {
    if (s > e) throw std::logic_error("start_pos must not be greater than end_pos");
    if (s >= sz || e >= sz) throw std::logic_error("node_size must be greater than start_pos and end_pos");
    if (e - s == sz - 1) full = true;
  }
  void extend_coordinates(node_coordinates c) {
    if (c.second >= node_size) throw std::logic_error("end coordinate must be less than node_size");
    if (full) return;
    if (c.first < start_pos) start_pos = c.first;
    if (c.second > end_pos) end_pos = c.second;
    if (end_pos - start_pos == node_size - 1) full = true;
  }
};

struct Traverser {  // :137-219
  coverage_Graph const *g = nullptr;
  int cur_Node = -1;
  std::size_t bases_remaining = 0;
  VariantSitePath traversed_loci;
  uint32_t traversed_index = 0;
  bool first_node = true;
  node_coordinate start_pos = 0, end_pos = 0;

  Traverser() = default;
  Traverser(coverage_Graph const *graph, node_access start_point, VariantSitePath loci, std::size_t read_size)
      : g(graph), cur_Node(start_point.node), bases_remaining(read_size), traversed_loci(std::move(loci)) {
    traversed_index = (uint32_t)traversed_loci.size();
    start_pos = (node_coordinate)start_point.offset;
  }
  coverage_Node const &node() const { return g->nodes[cur_Node]; }
  std::optional<int> next_Node() {  // :149-161
    if (first_node) {
      process_first_node();
      first_node = false;
      return cur_Node;
    } else if (bases_remaining == 0) {
      return {};
    } else {
      go_to_next_site();
      if (cur_Node < 0) return {};
      return cur_Node;
    }
  }
  void process_first_node() {  // :163-166
    update_coordinates();
    if (!node().is_in_bubble()) go_to_next_site();
  }
  void go_to_next_site() {  // :168-187
    start_pos = 0;
    while (node().next.size() == 1) {
      if (bases_remaining <= 0) {
        cur_Node = -1;
        return;
      }
      cur_Node = node().next[0];
      update_coordinates();
      if (node().is_in_bubble()) return;
    }
    --traversed_index;
    choose_allele();
    update_coordinates();
  }
  void update_coordinates() {  // :189-192
    assign_end_position();
    if (node().has_sequence()) bases_remaining -= (end_pos - start_pos + 1);
  }
  void assign_end_position() {  // :199-204
    end_pos = 0;
    std::size_t seq_size = node().sequence.size();
    if (seq_size > 0) end_pos = (node_coordinate)std::min(seq_size - 1, start_pos + bases_remaining - 1);
  }
  void choose_allele() {  // :206-219
    if (traversed_index >= traversed_loci.size()) throw std::logic_error("Traverser ran out of traversed loci");
    auto locus = traversed_loci[traversed_index];
    auto const &edges = node().next;
    if (locus.second < 0 || (std::size_t)locus.second >= edges.size()) throw std::logic_error("allele out of range");
    cur_Node = edges[locus.second];
  }
  node_coordinates get_node_coordinates() const { return {start_pos, end_pos}; }
};

struct PbCovRecorder {  // :221-296
  std::map<int, DummyCovNode> cov_mapping;
  PRG_Info const *prg_info = nullptr;
  std::size_t read_size = 0;

  PbCovRecorder(PRG_Info const &info, std::size_t rs) : prg_info(&info), read_size(rs) {}
  PbCovRecorder(PRG_Info const &info, SearchStates const &states, std::size_t rs) : prg_info(&info), read_size(rs) {
    for (auto const &ss : states) process_SearchState(ss);
    write_coverage_from_dummy_nodes();
  }
  void write_coverage_from_dummy_nodes() {  // :230-244
    for (auto const &el : cov_mapping) {
      auto &cov = prg_info->coverage_graph.nodes[el.first].coverage;
      for (auto i = el.second.start_pos; i <= el.second.end_pos; i++) {
        if (cov[i] == UINT16_MAX) continue;
#pragma omp atomic
        cov[i]++;
      }
    }
  }
  void process_SearchState(SearchState const &ss) {  // :246-270
    bool first = true;
    for (uint64_t occ = ss.sa_interval.first; occ <= ss.sa_interval.second; occ++) {
      auto coordinate = prg_info->fm_index[occ];
      auto access_point = prg_info->coverage_graph.random_access[coordinate];
      Traverser t{&prg_info->coverage_graph, access_point, ss.traversed_path, read_size};
      if (first) {
        first = false;
        record_full_traversal(t);
      } else {
        auto cur = t.next_Node().value();
        auto c = t.get_node_coordinates();
        process_Node(cur, c.first, c.second);
      }
    }
  }
  void record_full_traversal(Traverser &t) {  // :272-280
    auto cur = t.next_Node();
    auto c = t.get_node_coordinates();
    while (bool(cur)) {
      process_Node(cur.value(), c.first, c.second);
      cur = t.next_Node();
      c = t.get_node_coordinates();
    }
  }
  void process_Node(int cov_node, node_coordinate s, node_coordinate e) {  // :282-296
    auto const &n = prg_info->coverage_graph.nodes[cov_node];
    if (!n.has_sequence()) return;
    auto it = cov_mapping.find(cov_node);
    if (it == cov_mapping.end())
      cov_mapping.insert({cov_node, DummyCovNode{s, e, n.sequence.size()}});
    else
      it->second.extend_coordinates({s, e});
  }
};

static void record_allele_sum(Coverage &coverage, uniqueLoci const &loci) {  // allele_sum.cpp:31-43
  for (auto const &locus : loci) {
    auto site_index = siteID_to_index(locus.first);
#pragma omp atomic
    coverage.allele_sum_coverage[site_index][locus.second] += 1;
  }
}
static void record_grouped_allele_counts(Coverage &coverage, uniqueLoci const &loci) {  // grouped_allele_counts.cpp:17-49
  std::map<Marker, std::set<AlleleId>> site_allele_group;
  for (auto const &l : loci) site_allele_group[l.first].insert(l.second);
  for (auto const &e : site_allele_group) {
    AlleleIds ids(e.second.begin(), e.second.end());
    auto site_index = siteID_to_index(e.first);
    auto &site_coverage = coverage.grouped_allele_counts[site_index];
#pragma omp critical(gmo_grouped)
    site_coverage[ids] += 1;
  }
}

// allele_base_non_nested: allele_base.cpp:10-38
static std::vector<std::vector<std::vector<CovCount>>> allele_base_non_nested(PRG_Info const &info) {
  std::vector<std::vector<std::vector<CovCount>>> res;
  if (info.coverage_graph.is_nested) return res;
  res.assign(info.num_variant_sites, {});
  for (auto const &b : info.coverage_graph.bubble_map) {
    auto const &entry = info.coverage_graph.nodes[b.first];
    auto &referent = res.at(siteID_to_index(entry.site_ID));
    for (int a : entry.next) {
      auto const &an = info.coverage_graph.nodes[a];
      if (an.is_bubble_end())
        referent.emplace_back();
      else
        referent.emplace_back(an.coverage);
    }
  }
  return res;
}

// ---------------------------------------------------------------------------
// quasimap: src/genotype/quasimap/quasimap.cpp
// ---------------------------------------------------------------------------
static SearchStates process_read_char_search_states(int_Base pattern_char, SearchStates &states,
                                                    PRG_Info const &prg_info) {  // :258-268
  process_markers_search_states(states, prg_info);
  return search_base_backwards(pattern_char, states, prg_info);
}

struct QuasimapReadsStats {  // quasimap.hpp:17-24
  uint64_t all_reads_count = 0, skipped_reads_count = 0, missing_kmer_reads_count = 0, no_extension_reads_count = 0,
           exact_mapped_reads_count = 0;
};

struct Oracle {
  PRG_Info prg_info;
  KmerIndex kmer_index;
  uint32_t kmers_size = 0;
  Coverage coverage;
  QuasimapReadsStats stats;
  int rng_mode = RNG_LEMIRE;
  std::string last_error;
};

static bool all_read_kmers_occur_in_index(uint32_t k, Sequence const &read, KmerIndex const &idx) {  // :212-225
  for (std::size_t offset = 0; offset + k <= read.size(); ++offset) {
    Sequence kmer(read.begin() + offset, read.begin() + offset + k);
    if (idx.find(kmer) == idx.end()) return false;
  }
  return true;
}

static SearchStates search_read_backwards(Sequence const &read, Sequence const &kmer, KmerIndex const &kmer_index,
                                          PRG_Info const &prg_info) {  // :227-256
  auto f = kmer_index.find(kmer);
  if (f == kmer_index.end()) return SearchStates{};
  SearchStates ns = f->second;
  auto it = read.rbegin();
  std::advance(it, kmer.size());
  for (; it != read.rend(); ++it) {
    ns = process_read_char_search_states(*it, ns, prg_info);
    if (ns.empty()) break;
  }
  ns = handle_allele_encapsulated_states(ns, prg_info);
  return ns;
}

static void record_search_states(Oracle &o, SearchStates const &states, uint64_t read_length, uint32_t seed) {  // coverage_common.cpp:166-197
  MT19937 gen(seed);
  auto draw = [&](uint32_t lo, uint32_t hi) { return uniform_u32(gen, lo, hi, o.rng_mode); };
  SelectedMapping sel = select_mapping(states, &o.prg_info, draw);
  if (sel.navigational_search_states.empty()) return;
  PbCovRecorder rec{o.prg_info, sel.navigational_search_states, (std::size_t)read_length};
  record_allele_sum(o.coverage, sel.equivalence_class_loci);
  record_grouped_allele_counts(o.coverage, sel.equivalence_class_loci);
}

static void quasimap_read(Oracle &o, Sequence const &read, uint32_t seed) {  // :159-194
  if (read.size() < o.kmers_size) throw std::invalid_argument("read shorter than kmer size");
  if (!all_read_kmers_occur_in_index(o.kmers_size, read, o.kmer_index)) {
#pragma omp atomic
    o.stats.missing_kmer_reads_count += 1;
    return;
  }
  Sequence kmer(read.end() - o.kmers_size, read.end());
  auto states = search_read_backwards(read, kmer, o.kmer_index, o.prg_info);
  if (states.empty()) {
#pragma omp atomic
    o.stats.no_extension_reads_count += 1;
    return;
  }
  record_search_states(o, states, read.size(), seed);
#pragma omp atomic
  o.stats.exact_mapped_reads_count += 1;
}

static Sequence reverse_complement_read(Sequence const &read) {  // :273-298
  Sequence r;
  r.reserve(read.size());
  for (auto it = read.rbegin(); it != read.rend(); ++it) {
    int_Base b = *it;
    r.push_back(b >= 1 && b <= 4 ? (int_Base)(5 - b) : 0);
  }
  return r;
}

static void quasimap_forward_reverse(Oracle &o, Sequence const &read, uint32_t seed) {  // :143-157
  quasimap_read(o, read, seed);
  quasimap_read(o, reverse_complement_read(read), seed);
}

// ---------------------------------------------------------------------------
// read_stats depth: src/genotype/read_stats.cpp:72-160
// ---------------------------------------------------------------------------
static std::pair<AlleleId, CovCount> get_max_cov_haplogroup(GroupedAlleleCounts const &gc) {  // :72-92
  std::map<AlleleId, CovCount> counts;
  for (auto const &e : gc)
    for (auto a : e.first) counts[a] += e.second;  // uint16 arithmetic
  auto mx = std::max_element(counts.begin(), counts.end(),
                             [](auto const &a, auto const &b) { return a.second < b.second; });
  if (mx == counts.end()) return {0, 0};
  return *mx;
}
struct DepthStats {
  double mean = -1, variance = -1;
  uint64_t num_sites_noCov = 0, num_sites_total = 0;
};
static DepthStats compute_coverage_depth(Coverage const &coverage, coverage_Graph const &g) {  // :119-160
  DepthStats d;
  double total = 0;
  std::vector<double> coverages;
  for (auto const &np : g.bubble_map) {
    auto site_ID = g.nodes[np.first].site_ID;
    if (g.par_map.find(site_ID) != g.par_map.end()) continue;
    // extract_max_coverage_allele :94-117
    std::vector<CovCount> pb;
    int cur = np.first;
    auto mx = get_max_cov_haplogroup(coverage.grouped_allele_counts.at(siteID_to_index(site_ID)));
    CovCount allele_cov = mx.second;
    while (cur != np.second) {
      auto const &n = g.nodes[cur];
      if (n.is_bubble_start()) {
        mx = get_max_cov_haplogroup(coverage.grouped_allele_counts.at(siteID_to_index(n.site_ID)));
        cur = n.next.at(mx.first);
        continue;
      }
      if (n.has_sequence()) pb.insert(pb.end(), n.coverage.begin(), n.coverage.end());
      cur = n.next.at(0);
    }
    double site_cov;
    if (!pb.empty()) {
      double s = 0;  // Allele::get_average_cov: sum / size
      for (auto c : pb) s += c;
      site_cov = s / pb.size();
    } else
      site_cov = (double)allele_cov;
    total += site_cov;
    coverages.push_back(site_cov);
    if (allele_cov == 0) d.num_sites_noCov++;
  }
  d.mean = total / coverages.size();
  double tv = 0;
  for (auto c : coverages) tv += std::pow(c - d.mean, 2);
  d.variance = tv / coverages.size();
  d.num_sites_total = coverages.size();
  return d;
}

// ---------------------------------------------------------------------------
// Serialisation helpers for the C interface. A SearchStates list is flattened to
// int64: [n_states, {lo, hi, n_traversed, (site, allele)*, n_traversing, (site, allele)*}*]
// ---------------------------------------------------------------------------
static std::vector<int64_t> pack_states(SearchStates const &ss) {
  std::vector<int64_t> v;
  v.push_back((int64_t)ss.size());
  for (auto const &s : ss) {
    v.push_back(s.sa_interval.first);
    v.push_back(s.sa_interval.second);
    v.push_back((int64_t)s.traversed_path.size());
    for (auto const &l : s.traversed_path) {
      v.push_back(l.first);
      v.push_back(l.second);
    }
    v.push_back((int64_t)s.traversing_path.size());
    for (auto const &l : s.traversing_path) {
      v.push_back(l.first);
      v.push_back(l.second);
    }
  }
  return v;
}
static SearchStates unpack_states(const int64_t *p) {
  SearchStates ss;
  int64_t n = *p++;
  for (int64_t i = 0; i < n; ++i) {
    SearchState s;
    s.sa_interval.first = (SA_Index)*p++;
    s.sa_interval.second = (SA_Index)*p++;
    int64_t nt = *p++;
    for (int64_t j = 0; j < nt; ++j) {
      Marker m = (Marker)*p++;
      AlleleId a = (AlleleId)*p++;
      s.traversed_path.push_back({m, a});
    }
    int64_t ng = *p++;
    for (int64_t j = 0; j < ng; ++j) {
      Marker m = (Marker)*p++;
      AlleleId a = (AlleleId)*p++;
      s.traversing_path.push_back({m, a});
    }
    ss.push_back(s);
  }
  return ss;
}
static long emit(std::vector<int64_t> const &v, int64_t *out, long cap) {
  if ((long)v.size() > cap) return -(long)v.size();
  std::copy(v.begin(), v.end(), out);
  return (long)v.size();
}

}  // namespace gmo

// ===========================================================================
// C interface (ctypes). All functions return < 0 on error; gmo_last_error()
// gives the message of the last failure on this handle.
// ===========================================================================
using namespace gmo;

#define GMO_TRY(o) try {
#define GMO_CATCH(o, ret)                        \
  }                                              \
  catch (std::exception const &e) {              \
    if (o) ((Oracle *)(o))->last_error = e.what(); \
    return ret;                                  \
  }

extern "C" {

static std::string g_create_error;
const char *gmo_create_error() { return g_create_error.c_str(); }

// kmer_size == 0: no k-mer index. all_kmers != 0: index all 4^k k-mers (reference `gram build` behaviour,
// build.cpp:138-150); otherwise the index starts empty and gmo_index_kmers() adds listed k-mers.
void *gmo_create(const uint32_t *prg, uint64_t n, uint32_t kmer_size, int all_kmers, int rng_mode) {
  try {
    auto *o = new Oracle();
    std::vector<Marker> v(prg, prg + n);
    o->prg_info = generate_prg_info(v);
    o->kmers_size = kmer_size;
    o->rng_mode = rng_mode;
    try {
      o->coverage = empty_structure(o->prg_info);
    } catch (std::out_of_range const &) {
      o->coverage = Coverage{};  // graph introspection only
    }
    if (kmer_size > 0 && all_kmers) {
      auto diffs = get_prefix_diffs(get_all_kmers(kmer_size));
      o->kmer_index = index_kmers(diffs, (int)kmer_size, o->prg_info);
    }
    return o;
  } catch (std::exception const &e) {
    g_create_error = e.what();
    return nullptr;
  }
}
void gmo_destroy(void *h) { delete (Oracle *)h; }
const char *gmo_last_error(void *h) { return ((Oracle *)h)->last_error.c_str(); }

// index_kmers over an explicit k-mer list given as full k-mers (tests/.../test_BWT_search.cpp:218-229 style):
// each k-mer is its own "prefix diff" of full size.
int gmo_index_kmers(void *h, const uint8_t *kmers, uint64_t n_kmers) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  std::vector<Sequence> diffs;
  for (uint64_t i = 0; i < n_kmers; ++i)
    diffs.emplace_back(kmers + i * o->kmers_size, kmers + (i + 1) * o->kmers_size);
  auto idx = index_kmers(diffs, (int)o->kmers_size, o->prg_info);
  for (auto &e : idx) o->kmer_index[e.first] = e.second;
  return 0;
  GMO_CATCH(o, -1)
}


// index_kmers over a list of prefix diffs exactly as the reference takes them (build.cpp:101-131).
int gmo_index_kmer_diffs(void *h, const uint8_t *flat, const uint64_t *lengths, uint64_t n) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  std::vector<Sequence> diffs;
  uint64_t off = 0;
  for (uint64_t i = 0; i < n; ++i) {
    diffs.emplace_back(flat + off, flat + off + lengths[i]);
    off += lengths[i];
  }
  o->kmer_index = index_kmers(diffs, (int)o->kmers_size, o->prg_info);
  return 0;
  GMO_CATCH(o, -1)
}
// get_all_kmers order (kmers.cpp:88-96): out has 4^k * k bytes.
void gmo_all_kmers(uint32_t k, uint8_t *out) {
  auto kmers = get_all_kmers(k);
  uint64_t o = 0;
  for (auto &km : kmers)
    for (auto b : km) out[o++] = b;
}
// get_prefix_diffs (kmers.cpp:42-73): in = n kmers of size k; out_flat/out_len receive the diffs.
uint64_t gmo_prefix_diffs(const uint8_t *kmers, uint64_t n, uint32_t k, uint8_t *out_flat, uint64_t *out_len) {
  std::vector<Sequence> v;
  for (uint64_t i = 0; i < n; ++i) v.emplace_back(kmers + i * k, kmers + (i + 1) * k);
  auto d = get_prefix_diffs(v);
  uint64_t o = 0;
  for (uint64_t i = 0; i < d.size(); ++i) {
    out_len[i] = d[i].size();
    for (auto b : d[i]) out_flat[o++] = b;
  }
  return o;
}
void gmo_reverse_complement(const uint8_t *read, uint64_t len, uint8_t *out) {
  auto r = reverse_complement_read(Sequence(read, read + len));
  std::copy(r.begin(), r.end(), out);
}
// all_read_kmers_occur_in_index (quasimap.cpp:212-225)
int gmo_all_kmers_in_index(void *h, const uint8_t *read, uint64_t len) {
  auto *o = (Oracle *)h;
  return all_read_kmers_occur_in_index(o->kmers_size, Sequence(read, read + len), o->kmer_index) ? 1 : 0;
}
// get_max_cov_haplogroup on a site's recorded grouped counts (read_stats.cpp:72-92)
void gmo_max_cov_haplogroup(void *h, uint64_t site_index, int64_t *allele, int64_t *cov) {
  auto *o = (Oracle *)h;
  auto r = get_max_cov_haplogroup(o->coverage.grouped_allele_counts.at(site_index));
  *allele = r.first;
  *cov = r.second;
}
// directly set a grouped count (tests of read_stats with prepared coverage, test_read_stats.cpp:62-111)
void gmo_set_grouped(void *h, uint64_t site_index, const int64_t *ids, uint64_t n_ids, uint32_t count) {
  auto *o = (Oracle *)h;
  AlleleIds v;
  for (uint64_t i = 0; i < n_ids; ++i) v.push_back((AlleleId)ids[i]);
  o->coverage.grouped_allele_counts.at(site_index)[v] = (CovCount)count;
}

uint64_t gmo_text_size(void *h) { return ((Oracle *)h)->prg_info.fm_index.size(); }
void gmo_sa(void *h, uint32_t *out) {
  auto &fm = ((Oracle *)h)->prg_info.fm_index;
  std::copy(fm.sa.begin(), fm.sa.end(), out);
}
void gmo_bwt(void *h, uint32_t *out) {
  auto &fm = ((Oracle *)h)->prg_info.fm_index;
  std::copy(fm.bwt.begin(), fm.bwt.end(), out);
}
uint64_t gmo_rank(void *h, uint64_t upper, uint32_t base) { return dna_bwt_rank(upper, base, ((Oracle *)h)->prg_info); }
uint64_t gmo_C_of(void *h, uint32_t symbol) {
  auto &fm = ((Oracle *)h)->prg_info.fm_index;
  return fm.C[fm.char2comp(symbol)];
}
int gmo_marker_sa_interval(void *h, uint32_t marker, uint32_t *lo, uint32_t *hi) {
  auto iv = get_allele_marker_sa_interval(marker, ((Oracle *)h)->prg_info);
  *lo = iv.first;
  *hi = iv.second;
  return 0;
}
int gmo_base_next_sa_interval(void *h, uint32_t next_char, uint32_t first_sa, uint32_t lo, uint32_t hi, uint32_t *nlo,
                              uint32_t *nhi) {
  auto iv = base_next_sa_interval(next_char, first_sa, {lo, hi}, ((Oracle *)h)->prg_info);
  *nlo = iv.first;
  *nhi = iv.second;
  return 0;
}

long gmo_left_markers_search(void *h, uint32_t lo, uint32_t hi, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  SearchState s;
  s.sa_interval = {lo, hi};
  auto r = left_markers_search(s, o->prg_info);
  std::vector<int64_t> v;
  for (auto &l : r) {
    v.push_back(l.first);
    v.push_back(l.second);
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}
long gmo_vbwt_jumps(void *h, const int64_t *state_in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(state_in);
  return emit(pack_states(search_state_vBWT_jumps(ss.front(), o->prg_info)), out, cap);
  GMO_CATCH(o, -1)
}
long gmo_search_base_backwards(void *h, uint32_t base, const int64_t *in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  return emit(pack_states(search_base_backwards((int_Base)base, unpack_states(in), o->prg_info)), out, cap);
  GMO_CATCH(o, -1)
}
long gmo_process_read_char(void *h, uint32_t base, const int64_t *in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(in);
  return emit(pack_states(process_read_char_search_states((int_Base)base, ss, o->prg_info)), out, cap);
  GMO_CATCH(o, -1)
}
long gmo_kmer_states(void *h, const uint8_t *kmer, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  Sequence k(kmer, kmer + o->kmers_size);
  auto f = o->kmer_index.find(k);
  if (f == o->kmer_index.end()) return emit({-1}, out, cap);
  return emit(pack_states(f->second), out, cap);
  GMO_CATCH(o, -1)
}
uint64_t gmo_kmer_index_size(void *h) { return ((Oracle *)h)->kmer_index.size(); }
long gmo_search_read_backwards(void *h, const uint8_t *read, uint64_t len, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  Sequence r(read, read + len);
  Sequence kmer(r.end() - o->kmers_size, r.end());
  return emit(pack_states(search_read_backwards(r, kmer, o->kmer_index, o->prg_info)), out, cap);
  GMO_CATCH(o, -1)
}
long gmo_encapsulated(void *h, const int64_t *in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  return emit(pack_states(handle_allele_encapsulated_states(unpack_states(in), o->prg_info)), out, cap);
  GMO_CATCH(o, -1)
}

// LocusFinder on one state: out = [n_base, base..., n_loci, (site, allele)...]
long gmo_locus_finder(void *h, const int64_t *state_in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(state_in);
  LocusFinder l{ss.front(), &o->prg_info};
  std::vector<int64_t> v;
  v.push_back((int64_t)l.base_sites.size());
  for (auto s : l.base_sites) v.push_back(s);
  v.push_back((int64_t)l.unique_loci.size());
  for (auto &x : l.unique_loci) {
    v.push_back(x.first);
    v.push_back(x.second);
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}
// Selection with a forced draw value (MockRandomGenerator analogue, test_coverage_common.cpp:352-423).
// out = [draw_called, min, max, n_nav_states, n_loci, (site, allele)...]
long gmo_select_forced(void *h, const int64_t *states_in, uint32_t forced, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(states_in);
  int64_t called = 0, mn = 0, mx = 0;
  auto draw = [&](uint32_t a, uint32_t b) {
    called++;
    mn = a;
    mx = b;
    return forced;
  };
  auto sel = select_mapping(ss, &o->prg_info, draw);
  std::vector<int64_t> v{called, mn, mx, (int64_t)sel.navigational_search_states.size(),
                         (int64_t)sel.equivalence_class_loci.size()};
  for (auto &x : sel.equivalence_class_loci) {
    v.push_back(x.first);
    v.push_back(x.second);
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}

// Mock parental map (test_coverage_common.cpp:100-112, 300-322 inject one into an otherwise empty coverage_Graph):
// pairs = (site, parent_site, parent_allele) x n replaces coverage_graph.par_map.
int gmo_set_par_map(void *h, const int64_t *pairs, uint64_t n) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  o->prg_info.coverage_graph.par_map.clear();
  for (uint64_t i = 0; i < n; ++i)
    o->prg_info.coverage_graph.par_map.insert({(Marker)pairs[3 * i], VariantLocus{(Marker)pairs[3 * i + 1], (AlleleId)pairs[3 * i + 2]}});
  return 0;
  GMO_CATCH(o, -1)
}
// LocusFinder::check_site_uniqueness (coverage_common.cpp:17-32): 1 if it throws std::logic_error, 0 if not.
int gmo_check_site_uniqueness(void *h, const int64_t *state_in) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(state_in);
  try {
    LocusFinder::check_site_uniqueness(ss.front());
  } catch (std::logic_error const &) {
    return 1;
  }
  return 0;
  GMO_CATCH(o, -1)
}
// LocusFinder::assign_nested_locus for every given locus in turn, then (traversed_of >= 0) assign_traversed_loci of state
// `traversed_of` (coverage_common.cpp:34-51,78-83): out = [n_base, base..., n_used, used..., n_loci, (site, allele)...]
long gmo_assign_loci(void *h, const int64_t *loci, uint64_t n_loci, const int64_t *states_in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  LocusFinder l;
  for (uint64_t i = 0; i < n_loci; ++i) l.assign_nested_locus(VariantLocus{(Marker)loci[2 * i], (AlleleId)loci[2 * i + 1]}, &o->prg_info);
  if (states_in)
    for (auto const &ss : unpack_states(states_in)) l.assign_traversed_loci(ss, &o->prg_info);
  std::vector<int64_t> v;
  v.push_back((int64_t)l.base_sites.size());
  for (auto s : l.base_sites) v.push_back(s);
  v.push_back((int64_t)l.used_sites.size());
  for (auto s : l.used_sites) v.push_back(s);
  v.push_back((int64_t)l.unique_loci.size());
  for (auto &x : l.unique_loci) {
    v.push_back(x.first);
    v.push_back(x.second);
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}
// MappingInstanceSelector::process_searchstates + count_nonvar_search_states (coverage_common.cpp:110-141):
// out = [nonvariant_count, n_entries, {n_sites, sites..., n_states, (lo, hi)..., n_loci, (site, allele)...}*] in map order
long gmo_unique_site_paths(void *h, const int64_t *states_in, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto ss = unpack_states(states_in);
  uniqueSitePaths usps;
  select_mapping(ss, &o->prg_info, [](uint32_t a, uint32_t) { return a; }, &usps);
  int64_t nonvar = 0;
  for (auto const &s : ss)
    if (!s.has_path()) nonvar += (int64_t)s.sa_interval.second - (int64_t)s.sa_interval.first + 1;
  std::vector<int64_t> v{nonvar, (int64_t)usps.size()};
  for (auto const &e : usps) {
    v.push_back((int64_t)e.first.size());
    for (auto m : e.first) v.push_back(m);
    v.push_back((int64_t)e.second.first.size());
    for (auto const &st : e.second.first) {
      v.push_back(st.sa_interval.first);
      v.push_back(st.sa_interval.second);
    }
    v.push_back((int64_t)e.second.second.size());
    for (auto const &x : e.second.second) {
      v.push_back(x.first);
      v.push_back(x.second);
    }
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}

// RNG known answers (test_coverage_common.cpp:257-298)
void gmo_rng_raw(uint32_t seed, uint32_t n, uint32_t *out) {
  MT19937 g(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = g.next();
}
void gmo_rng_generate(uint32_t seed, uint32_t min, uint32_t max, uint32_t n, int mode, uint32_t *out) {
  MT19937 g(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = uniform_u32(g, min, max, mode);
}
// The toolchain's own libstdc++ (what a reference binary built here would do).
void gmo_rng_generate_std(uint32_t seed, uint32_t min, uint32_t max, uint32_t n, uint32_t *out) {
  std::mt19937 g(seed);
  for (uint32_t i = 0; i < n; ++i) {
    std::uniform_int_distribution<uint32_t> range(min, max);
    out[i] = range(g);
  }
}
// Per-read selection seeds: quasimap.cpp:120-141 — each file draws 5000 seeds per batch of <= 5000 reads
// from one master mt19937(seed) shared by all files.
void gmo_master_seeds(uint32_t master_seed, const uint64_t *reads_per_file, uint64_t n_files, uint32_t *out) {
  MT19937 g(master_seed);
  uint64_t o = 0;
  for (uint64_t f = 0; f < n_files; ++f) {
    uint64_t n = reads_per_file[f];
    for (uint64_t start = 0; start < n; start += 5000) {
      for (uint64_t i = 0; i < 5000; ++i) {
        uint32_t s = g.next();
        if (start + i < n) out[o++] = s;
      }
    }
  }
}

// Mapping ---------------------------------------------------------------
int gmo_quasimap_read(void *h, const uint8_t *read, uint64_t len, uint32_t seed) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  quasimap_read(*o, Sequence(read, read + len), seed);
  return 0;
  GMO_CATCH(o, -1)
}
// handle_reads_buffer semantics (quasimap.cpp:82-118) over pre-encoded reads: bases 1..4, any 0 => read skipped
// (encode_dna_bases returns an empty read, utils.cpp:73-92).
int gmo_map_reads(void *h, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds, uint64_t n,
                  int threads) {
  auto *o = (Oracle *)h;
  int err = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
#pragma omp atomic
    o->stats.all_reads_count += 2;
    Sequence read(reads + offsets[i], reads + offsets[i + 1]);
    bool bad = read.empty();
    for (auto b : read)
      if (b < 1 || b > 4) bad = true;
    if (bad) {
#pragma omp atomic
      o->stats.skipped_reads_count += 2;
      continue;
    }
    try {
      quasimap_forward_reverse(*o, read, seeds[i]);
    } catch (std::exception const &e) {
#pragma omp critical(gmo_err)
      {
        o->last_error = e.what();
        err = -1;
      }
    }
  }
  return err;
}
void gmo_stats(void *h, uint64_t *out5) {
  auto &s = ((Oracle *)h)->stats;
  out5[0] = s.all_reads_count;
  out5[1] = s.skipped_reads_count;
  out5[2] = s.missing_kmer_reads_count;
  out5[3] = s.no_extension_reads_count;
  out5[4] = s.exact_mapped_reads_count;
}
void gmo_reset_coverage(void *h) {
  auto *o = (Oracle *)h;
  o->coverage = empty_structure(o->prg_info);
  for (auto &n : o->prg_info.coverage_graph.nodes) std::fill(n.coverage.begin(), n.coverage.end(), 0);
  o->stats = QuasimapReadsStats{};
}

// PbCovRecorder on explicit states (test_allele_base.cpp:236-602).
int gmo_record_per_base(void *h, const int64_t *states_in, uint64_t read_size) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  PbCovRecorder rec{o->prg_info, unpack_states(states_in), (std::size_t)read_size};
  return 0;
  GMO_CATCH(o, -1)
}
// Dummy cov nodes of process_SearchState on one state: out = [n, (node, start, end, size)*]
long gmo_dummy_cov_nodes(void *h, const int64_t *states_in, uint64_t read_size, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  PbCovRecorder rec{o->prg_info, (std::size_t)read_size};
  for (auto const &s : unpack_states(states_in)) rec.process_SearchState(s);
  std::vector<int64_t> v{(int64_t)rec.cov_mapping.size()};
  for (auto &e : rec.cov_mapping) {
    v.push_back(e.first);
    v.push_back(e.second.start_pos);
    v.push_back(e.second.end_pos);
    v.push_back((int64_t)e.second.node_size);
  }
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}
// Traverser walk from a PRG position: out = [n, (node, site, allele, start, end)*, bases_remaining, final_start, final_end]
long gmo_traverse(void *h, uint64_t prg_pos, const int64_t *path_pairs, uint64_t n_pairs, uint64_t read_size,
                  int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  VariantSitePath p;
  for (uint64_t i = 0; i < n_pairs; ++i) p.push_back({(Marker)path_pairs[2 * i], (AlleleId)path_pairs[2 * i + 1]});
  Traverser t{&o->prg_info.coverage_graph, o->prg_info.coverage_graph.random_access[prg_pos], p, read_size};
  std::vector<int64_t> v{0};
  auto cur = t.next_Node();
  while (bool(cur)) {
    auto const &n = o->prg_info.coverage_graph.nodes[cur.value()];
    auto c = t.get_node_coordinates();
    v.push_back(cur.value());
    v.push_back(n.site_ID);
    v.push_back(n.allele_ID);
    v.push_back(c.first);
    v.push_back(c.second);
    v[0]++;
    cur = t.next_Node();
  }
  v.push_back((int64_t)t.bases_remaining);
  v.push_back(t.start_pos);  // the Traverser's own final coordinates (get_node_coordinates after exhaustion)
  v.push_back(t.end_pos);
  return emit(v, out, cap);
  GMO_CATCH(o, -1)
}
int gmo_record_loci(void *h, const int64_t *pairs, uint64_t n_pairs) {  // allele_sum + grouped on explicit loci
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  uniqueLoci l;
  for (uint64_t i = 0; i < n_pairs; ++i) l.insert({(Marker)pairs[2 * i], (AlleleId)pairs[2 * i + 1]});
  record_allele_sum(o->coverage, l);
  record_grouped_allele_counts(o->coverage, l);
  return 0;
  GMO_CATCH(o, -1)
}

// AbstractReadStats::extract_max_coverage_allele (read_stats.cpp:94-117) for the site with marker `site_marker`: the allele
// with the most coverage (nested sites resolved the same way) as letters, and that haplogroup's coverage.
long gmo_extract_max_cov_allele(void *h, uint64_t site_marker, char *seq_out, long cap, int64_t *cov) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto const &g = o->prg_info.coverage_graph;
  for (auto const &np : g.bubble_map) {
    if (g.nodes[np.first].site_ID != (Marker)site_marker) continue;
    std::string seq;
    int cur = np.first;
    auto mx = get_max_cov_haplogroup(o->coverage.grouped_allele_counts.at(siteID_to_index((Marker)site_marker)));
    *cov = (int64_t)mx.second;
    while (cur != np.second) {
      auto const &n = g.nodes[cur];
      if (n.is_bubble_start()) {
        mx = get_max_cov_haplogroup(o->coverage.grouped_allele_counts.at(siteID_to_index(n.site_ID)));
        cur = n.next.at(mx.first);
        continue;
      }
      if (n.has_sequence()) seq += n.sequence;
      cur = n.next.at(0);
    }
    if ((long)seq.size() + 1 > cap) return -(long)seq.size() - 1;
    memcpy(seq_out, seq.c_str(), seq.size() + 1);
    return (long)seq.size();
  }
  throw std::runtime_error("no such site");
  GMO_CATCH(o, -1000000)
}

// Coverage read-back ---------------------------------------------------
uint64_t gmo_num_sites(void *h) { return ((Oracle *)h)->prg_info.num_variant_sites; }
int gmo_is_nested(void *h) { return ((Oracle *)h)->prg_info.coverage_graph.is_nested ? 1 : 0; }
// allele sum: out = [n_sites, {n_alleles, counts...}*]
long gmo_allele_sum(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  std::vector<int64_t> v{(int64_t)o->coverage.allele_sum_coverage.size()};
  for (auto &s : o->coverage.allele_sum_coverage) {
    v.push_back((int64_t)s.size());
    for (auto c : s) v.push_back(c);
  }
  return emit(v, out, cap);
}
// grouped: out = [n_sites, {n_groups, {n_ids, ids..., count}*}*] (groups in lexicographic id order)
long gmo_grouped(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  std::vector<int64_t> v{(int64_t)o->coverage.grouped_allele_counts.size()};
  for (auto &s : o->coverage.grouped_allele_counts) {
    v.push_back((int64_t)s.size());
    for (auto &g : s) {
      v.push_back((int64_t)g.first.size());
      for (auto a : g.first) v.push_back(a);
      v.push_back(g.second);
    }
  }
  return emit(v, out, cap);
}
// per-base over ALL graph nodes that own coverage, in node creation order:
// out = [n_nodes_with_cov, {node_id, site, allele, first_prg_pos(-1 if unknown), len, cov...}*]
long gmo_per_base_nodes(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  auto &g = o->prg_info.coverage_graph;
  std::vector<int64_t> first_pos(g.nodes.size(), -1);
  for (std::size_t p = 0; p < g.random_access.size(); ++p) {
    int n = g.random_access[p].node;
    if (o->prg_info.encoded_prg[p] <= 4 && first_pos[n] < 0) first_pos[n] = (int64_t)p;
  }
  std::vector<int64_t> v{0};
  for (std::size_t i = 0; i < g.nodes.size(); ++i) {
    auto &n = g.nodes[i];
    if (!(n.is_in_bubble() && n.has_sequence())) continue;
    v[0]++;
    v.push_back((int64_t)i);
    v.push_back(n.site_ID);
    v.push_back(n.allele_ID);
    v.push_back(first_pos[i]);
    v.push_back((int64_t)n.coverage.size());
    for (auto c : n.coverage) v.push_back(c);
  }
  return emit(v, out, cap);
}
// coverage of the node owning PRG position pos (collect_coverage, tests/test_resources/test_resources.cpp:9-21)
long gmo_node_coverage_at(void *h, uint64_t pos, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  auto &g = o->prg_info.coverage_graph;
  auto &n = g.nodes[g.random_access[pos].node];
  std::vector<int64_t> v(n.coverage.begin(), n.coverage.end());
  return emit(v, out, cap);
}
// allele_base_non_nested: out = [n_sites, {n_alleles, {len, cov...}*}*]; n_sites = 0 when nested
long gmo_allele_base_non_nested(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  auto r = allele_base_non_nested(o->prg_info);
  std::vector<int64_t> v{(int64_t)r.size()};
  for (auto &s : r) {
    v.push_back((int64_t)s.size());
    for (auto &a : s) {
      v.push_back((int64_t)a.size());
      for (auto c : a) v.push_back(c);
    }
  }
  return emit(v, out, cap);
}
int gmo_depth_stats(void *h, double *mean, double *variance, uint64_t *no_cov, uint64_t *total) {
  auto *o = (Oracle *)h;
  GMO_TRY(o)
  auto d = compute_coverage_depth(o->coverage, o->prg_info.coverage_graph);
  *mean = d.mean;
  *variance = d.variance;
  *no_cov = d.num_sites_noCov;
  *total = d.num_sites_total;
  return 0;
  GMO_CATCH(o, -1)
}

// Graph introspection ---------------------------------------------------
// random_access: out[pos*5 + {0..4}] = node, offset, target.first, target.second, node.site, (5 values per pos)
void gmo_random_access(void *h, int64_t *out) {
  auto *o = (Oracle *)h;
  auto &g = o->prg_info.coverage_graph;
  for (std::size_t p = 0; p < g.random_access.size(); ++p) {
    auto &ra = g.random_access[p];
    out[p * 6 + 0] = ra.node;
    out[p * 6 + 1] = (int64_t)ra.offset;
    out[p * 6 + 2] = ra.target.first;
    out[p * 6 + 3] = ra.target.second;
    out[p * 6 + 4] = g.nodes[ra.node].site_ID;
    out[p * 6 + 5] = g.nodes[ra.node].allele_ID;
  }
}
// target_map: out = [n_keys, {key, n_targets, (ID, deletion_allele)*}*] keys ascending
long gmo_target_map(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  std::map<Marker, std::vector<targeted_marker>> sorted(o->prg_info.coverage_graph.target_map.begin(),
                                                         o->prg_info.coverage_graph.target_map.end());
  std::vector<int64_t> v{(int64_t)sorted.size()};
  for (auto &e : sorted) {
    v.push_back(e.first);
    v.push_back((int64_t)e.second.size());
    for (auto &t : e.second) {
      v.push_back(t.ID);
      v.push_back(t.direct_deletion_allele);
    }
  }
  return emit(v, out, cap);
}
// par_map: out = [n, (site, parent_site, parent_allele)*] ascending site
long gmo_par_map(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  std::map<Marker, VariantLocus> sorted(o->prg_info.coverage_graph.par_map.begin(),
                                        o->prg_info.coverage_graph.par_map.end());
  std::vector<int64_t> v{(int64_t)sorted.size()};
  for (auto &e : sorted) {
    v.push_back(e.first);
    v.push_back(e.second.first);
    v.push_back(e.second.second);
  }
  return emit(v, out, cap);
}
// bubble_map order: out = [n, (site_ID, entry_pos, n_edges)*] in std::map iteration order
long gmo_bubble_order(void *h, int64_t *out, long cap) {
  auto *o = (Oracle *)h;
  auto &g = o->prg_info.coverage_graph;
  std::vector<int64_t> v{(int64_t)g.bubble_map.size()};
  for (auto &b : g.bubble_map) {
    v.push_back(g.nodes[b.first].site_ID);
    v.push_back((int64_t)g.nodes[b.first].pos);
    v.push_back((int64_t)g.nodes[b.first].next.size());
  }
  return emit(v, out, cap);
}

}  // extern "C"
