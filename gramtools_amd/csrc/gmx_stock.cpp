// gmx_stock.cpp — readers (and, for round trips, writers) of the files a STOCK gramtools `build` leaves in gram_dir that
// are plain SDSL-lite 2.1.1 vectors:
//   kmers            sdsl::int_vector<3>   the indexed k-mers' bases, k per k-mer          (build/kmer_index/dump.cpp:27-43)
//   kmers_stats      sdsl::int_vector<>    per k-mer: #states, then each state's path length (:45-74)
//   sa_intervals     sdsl::int_vector<>    per state: first, last SA index                   (:76-100)
//   paths            sdsl::int_vector<>    per path element: site marker, allele id + 1      (:102-137)
//   {a,c,g,t}_base_bwt_mask  sdsl::bit_vector  BWT[i] == base                                (prg/make_data_structures.cpp:78-138)
// read the way load.cpp:71-173 / make_data_structures.cpp:140-156 read them. On-disk form (sdsl/int_vector.hpp of v2.1.1,
// int_vector_trait::write_header + int_vector::serialize): a 64-bit little-endian length IN BITS, then — variable-width
// vectors only — one byte holding the width, then the values packed least-significant-bit first into 64-bit little-endian
// words, (bits + 63) / 64 of them. `fm_index` (a serialised csa_wt) is not decoded and of `cov_graph` (a Boost binary archive)
// only the head is looked at (signature, library version, bubble_map's element count = the number of sites): both are pure
// functions of gram_dir/prg, which the native index is built from. `gram build --check_stock` runs the comparison
// (gmx_index_check_stock_files) on a gram_dir a stock build filled; `gram build --write_stock` writes the vectors and masks.
//
// PARITY UNPINNED: no file written by the reference exists in this repository (its tests hold none, and it cannot be built
// here: SDSL, Boost and htslib are absent), so these readers are checked against the format as specified above, against
// hand-assembled byte strings, and by a round trip through the writers below — not against a reference-written fixture.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_core.h"
#include "gmx_index.h"
#include "gmx_internal.h"

namespace {

struct IntVector {
  uint32_t width = 0;
  std::vector<uint64_t> v;
};

IntVector read_int_vector(const std::string &path, uint32_t fixed_width) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<unsigned char> buf;
  unsigned char chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
  fclose(f);
  size_t at = 0;
  if (buf.size() < 8) throw std::runtime_error(path + ": shorter than an SDSL vector header");
  uint64_t bits = 0;
  memcpy(&bits, buf.data(), 8);
  at = 8;
  IntVector out;
  out.width = fixed_width;
  if (fixed_width == 0) {
    if (buf.size() < 9) throw std::runtime_error(path + ": no width byte");
    out.width = buf[at++];
  }
  if (out.width == 0 || out.width > 64) throw std::runtime_error(path + ": element width " + std::to_string(out.width) + " is not 1..64");
  const uint64_t words = (bits + 63) / 64;
  if (bits / 8 > buf.size() || at + words * 8 > buf.size()) throw std::runtime_error(path + ": " + std::to_string(bits) + " bits announced, file too short");
  const uint64_t n = bits / out.width;
  out.v.resize(n);
  const unsigned char *d = buf.data() + at;
  auto word = [&](uint64_t i) {
    uint64_t w = 0;
    if (i < words) memcpy(&w, d + i * 8, 8);
    return w;
  };
  const uint64_t mask = out.width == 64 ? ~0ull : ((1ull << out.width) - 1ull);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t bit = i * out.width, w = bit >> 6, off = bit & 63;
    uint64_t x = word(w) >> off;
    if (off + out.width > 64) x |= word(w + 1) << (64 - off);
    out.v[i] = x & mask;
  }
  return out;
}

void write_int_vector(const std::string &path, const uint64_t *values, uint64_t n, uint32_t width, bool fixed) {
  if (width == 0 || width > 64) throw std::runtime_error("element width must be 1..64");
  const uint64_t bits = n * width, words = (bits + 63) / 64;
  std::vector<uint64_t> data(words, 0);
  const uint64_t mask = width == 64 ? ~0ull : ((1ull << width) - 1ull);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t x = values[i] & mask, bit = i * width, w = bit >> 6, off = bit & 63;
    data[w] |= x << off;
    if (off + width > 64) data[w + 1] |= x >> (64 - off);
  }
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  bool ok = fwrite(&bits, 8, 1, f) == 1;
  if (!fixed) {
    const unsigned char wb = (unsigned char)width;
    ok = ok && fwrite(&wb, 1, 1, f) == 1;
  }
  if (words) ok = ok && fwrite(data.data(), 8, words, f) == words;
  ok = (fclose(f) == 0) && ok;
  if (!ok) throw std::runtime_error("error writing " + path);
}

uint32_t bits_for(const std::vector<uint64_t> &v) {  // sdsl::util::bit_compress: hi(max) + 1, at least 1
  uint64_t m = 0;
  for (uint64_t x : v) m = std::max(m, x);
  uint32_t w = 1;
  while (w < 64 && (m >> w)) ++w;
  return w;
}

std::string join_path(const std::string &dir, const std::string &name) { return dir.empty() || dir.back() == '/' ? dir + name : dir + "/" + name; }

// BWT[i] == base (1..4) from the rank blocks (the sentinel is stored as code 00 without a marker bit: not an A)
bool bwt_is(const gmx::HostIndex &h, uint64_t i, uint32_t base) {
  if (i == h.sentinel_pos) return false;
  const GmxRankBlock &b = h.blocks[i >> GMX_BLK_SHIFT];
  uint64_t w0, w1;
  gmx_match_words(b, base, w0, w1);
  const uint32_t r = (uint32_t)(i & GMX_BLK_MASK);
  return ((r < 64 ? w0 >> r : w1 >> (r - 64)) & 1ull) != 0;
}

int fail(const std::string &m, int code = GMX_EINVAL) {
  gmx_set_error(m);
  return code;
}

}  // namespace

extern "C" {

int64_t gmx_stock_read_int_vector(const char *path, uint32_t fixed_width, uint64_t *out, uint64_t cap, uint32_t *width_out) try {
  if (!path) return fail("gmx_stock_read_int_vector: null path");
  try {
    IntVector v = read_int_vector(path, fixed_width);
    if (width_out) *width_out = v.width;
    if (out) {
      if (v.v.size() > cap) return fail("gmx_stock_read_int_vector: buffer too small", GMX_ECAP);
      std::copy(v.v.begin(), v.v.end(), out);
    }
    return (int64_t)v.v.size();
  } catch (std::exception const &e) {
    return fail(e.what());
  }
} GMX_GUARD_INT("gmx_stock_read_int_vector")

int gmx_stock_write_int_vector(const char *path, const uint64_t *values, uint64_t n, uint32_t width, int fixed) try {
  if (!path || (!values && n)) return fail("gmx_stock_write_int_vector: null argument");
  try {
    write_int_vector(path, values, n, width, fixed != 0);
    return GMX_OK;
  } catch (std::exception const &e) {
    return fail(e.what());
  }
} GMX_GUARD_INT("gmx_stock_write_int_vector")

// The k-mer index and the four base masks of `ix` in the stock files' formats (k-mers in ascending table order; the
// reference writes them in its hash map's order, and reads any order).
int gmx_index_write_stock_files(const gmx_index *ix, const char *gram_dir) try {
  if (!ix || !gram_dir) return fail("gmx_index_write_stock_files: null argument");
  const gmx::HostIndex &h = gmx_index_host(ix);
  const uint32_t k = h.kmer_size;
  if (k == 0 || k > 15) return fail("gmx_index_write_stock_files: the index has no k-mer table");
  try {
    std::vector<uint64_t> kmers, stats, sa, paths;
    for (uint64_t code = 0; code < (1ull << (2 * k)); ++code) {
      const std::vector<int64_t> st = gmx::seed_states_of(h, (uint32_t)code, false);
      if (st.empty() || st[0] < 0) continue;  // not indexed
      for (uint32_t j = 0; j < k; ++j) kmers.push_back(((code >> (2 * j)) & 3u) + 1u);
      stats.push_back((uint64_t)st[0]);
      size_t at = 1;
      for (int64_t s = 0; s < st[0]; ++s) {
        sa.push_back((uint64_t)st[at]);
        sa.push_back((uint64_t)st[at + 1]);
        const int64_t nt = st[at + 2];
        at += 3;
        for (int64_t j = 0; j < nt; ++j, at += 2) {
          paths.push_back((uint64_t)st[at]);
          paths.push_back((uint64_t)(st[at + 1] + 1));  // ALLELE_UNKNOWN = -1 is stored as 0 (dump.cpp:106-108)
        }
        const int64_t ng = st[at++];
        for (int64_t j = 0; j < ng; ++j, at += 2) {
          paths.push_back((uint64_t)st[at]);
          paths.push_back(0);
        }
        stats.push_back((uint64_t)(nt + ng));
      }
    }
    const std::string d = gram_dir;
    write_int_vector(join_path(d, "kmers"), kmers.data(), kmers.size(), 3, true);
    write_int_vector(join_path(d, "kmers_stats"), stats.data(), stats.size(), bits_for(stats), false);
    write_int_vector(join_path(d, "sa_intervals"), sa.data(), sa.size(), bits_for(sa), false);
    write_int_vector(join_path(d, "paths"), paths.data(), paths.size(), bits_for(paths), false);
    const uint64_t n = h.sa.size();
    std::vector<uint64_t> bits(n);
    const char *names[4] = {"a_base_bwt_mask", "c_base_bwt_mask", "g_base_bwt_mask", "t_base_bwt_mask"};
    for (uint32_t base = 1; base <= 4; ++base) {
      for (uint64_t i = 0; i < n; ++i) bits[i] = bwt_is(h, i, base) ? 1 : 0;
      write_int_vector(join_path(d, names[base - 1]), bits.data(), n, 1, true);
    }
    return GMX_OK;
  } catch (std::exception const &e) {
    return fail(e.what());
  }
} GMX_GUARD_INT("gmx_index_write_stock_files")

// Reads gram_dir's kmers / kmers_stats / sa_intervals / paths as kmer_index::load does (load.cpp:161-173) and the four
// masks (make_data_structures.cpp:140-156), and compares them with the native index of the same PRG and k.
int gmx_index_check_stock_files(const gmx_index *ix, const char *gram_dir, gmx_stock_report *out) try {
  if (!ix || !gram_dir || !out) return fail("gmx_index_check_stock_files: null argument");
  memset(out, 0, sizeof(*out));
  const gmx::HostIndex &h = gmx_index_host(ix);
  const uint32_t k = h.kmer_size;
  if (k == 0 || k > 15) return fail("gmx_index_check_stock_files: the index has no k-mer table");
  try {
    const std::string d = gram_dir;
    const IntVector kmers = read_int_vector(join_path(d, "kmers"), 3);
    const IntVector stats = read_int_vector(join_path(d, "kmers_stats"), 0);
    const IntVector sa = read_int_vector(join_path(d, "sa_intervals"), 0);
    const IntVector paths = read_int_vector(join_path(d, "paths"), 0);
    if (kmers.v.size() % k) return fail("kmers: length is not a multiple of k");
    uint64_t si = 0, sai = 0, pi = 0;
    std::vector<bool> seen(1ull << (2 * k), false);
    for (uint64_t at = 0; at + k <= kmers.v.size(); at += k) {
      uint32_t code = 0;
      for (uint32_t j = 0; j < k; ++j) {
        const uint64_t b = kmers.v[at + j];
        if (b < 1 || b > 4) return fail("kmers: a symbol outside 1..4");
        code |= (uint32_t)(b - 1) << (2 * j);
      }
      if (si >= stats.v.size()) return fail("kmers_stats: shorter than the k-mer list");
      const uint64_t n_states = stats.v[si];
      if (si + 1 + n_states > stats.v.size() || sai + 2 * n_states > sa.v.size()) return fail("kmers_stats / sa_intervals: truncated");
      // the k-mer's states as load.cpp rebuilds them: [n, {lo, hi, n_traversed, (site, allele)*, n_traversing, (site, -1)*}*]
      std::vector<int64_t> stock{(int64_t)n_states};
      for (uint64_t s = 0; s < n_states; ++s) {
        const uint64_t plen = stats.v[si + 1 + s];
        if (pi + 2 * plen > paths.v.size()) return fail("paths: truncated");
        std::vector<std::pair<int64_t, int64_t>> tvd, tvg;
        for (uint64_t j = 0; j < plen; ++j, pi += 2) {
          const int64_t allele = (int64_t)paths.v[pi + 1] - 1;
          (allele != -1 ? tvd : tvg).emplace_back((int64_t)paths.v[pi], allele);
        }
        stock.push_back((int64_t)sa.v[sai]);
        stock.push_back((int64_t)sa.v[sai + 1]);
        sai += 2;
        stock.push_back((int64_t)tvd.size());
        for (auto const &e : tvd) {
          stock.push_back(e.first);
          stock.push_back(e.second);
        }
        stock.push_back((int64_t)tvg.size());
        for (auto const &e : tvg) {
          stock.push_back(e.first);
          stock.push_back(e.second);
        }
      }
      si += 1 + n_states;
      out->kmers += 1;
      out->states += n_states;
      if (seen[code]) out->duplicate_kmers += 1;
      seen[code] = true;
      // (as sets of states: the order of a k-mer's states is the reference's list order, which nothing here pins)
      auto split = [](const std::vector<int64_t> &flat) {
        std::vector<std::vector<int64_t>> states;
        if (flat.empty() || flat[0] < 0) return states;
        size_t p = 1;
        for (int64_t s = 0; s < flat[0]; ++s) {
          const size_t b = p;
          p += 2;
          p += 1 + 2 * (size_t)flat[p];
          p += 1 + 2 * (size_t)flat[p];
          states.emplace_back(flat.begin() + b, flat.begin() + p);
        }
        std::sort(states.begin(), states.end());
        return states;
      };
      if (split(gmx::seed_states_of(h, code, false)) != split(stock)) out->kmer_mismatches += 1;
    }
    // k-mers the native index has and the files do not
    for (uint64_t code = 0; code < (1ull << (2 * k)); ++code)
      if (!seen[code]) {
        const std::vector<int64_t> mine = gmx::seed_states_of(h, (uint32_t)code, false);
        if (!mine.empty() && mine[0] >= 0) out->kmers_missing_in_files += 1;
      }
    const char *names[4] = {"a_base_bwt_mask", "c_base_bwt_mask", "g_base_bwt_mask", "t_base_bwt_mask"};
    for (uint32_t base = 1; base <= 4; ++base) {
      const IntVector m = read_int_vector(join_path(d, names[base - 1]), 1);
      out->mask_bits += m.v.size();
      if (m.v.size() != h.sa.size()) {
        out->mask_mismatches += std::max<uint64_t>(m.v.size(), h.sa.size());
        continue;
      }
      for (uint64_t i = 0; i < m.v.size(); ++i)
        if ((m.v[i] != 0) != bwt_is(h, i, base)) out->mask_mismatches += 1;
    }
    // cov_graph: the head of the Boost binary archive (basic_binary_oarchive: the signature as a length-prefixed string, the
    // library version, then the object — whose first member, bubble_map, starts with its element count). Where exactly the
    // count lies depends on the Boost version's widths for class information (tracking flag, class version) and collection
    // sizes: the first 8-byte value in the 64 bytes behind the version that equals the native site count is taken as found,
    // else the first plausible one (non-zero, below 2^32) is reported. PARITY UNPINNED like the rest of this file.
    {
      FILE *f = fopen(join_path(d, "cov_graph").c_str(), "rb");
      if (f) {
        unsigned char head[160];
        const size_t got = fread(head, 1, sizeof(head), f);
        fclose(f);
        static const char sig[] = "serialization::archive";
        const size_t sl = sizeof(sig) - 1;
        uint64_t len = 0;
        if (got >= 8 + sl + 2) memcpy(&len, head, 8);
        if (len == sl && memcmp(head + 8, sig, sl) == 0) {
          out->cov_graph_state = 1;
          out->cov_graph_library_version = (uint64_t)head[8 + sl] | ((uint64_t)head[8 + sl + 1] << 8);
          const uint64_t want = h.sites.size();
          uint64_t first_plausible = 0;
          for (size_t at = 8 + sl + 2; at + 8 <= got && at < 8 + sl + 2 + 64; ++at) {
            uint64_t v = 0;
            memcpy(&v, head + at, 8);
            if (v == want && want) {
              out->cov_graph_state = 2;
              out->cov_graph_sites = v;
              break;
            }
            if (!first_plausible && v && v < (1ull << 32) && at >= 8 + sl + 2 + 2) first_plausible = v;
          }
          if (out->cov_graph_state == 1 && want) {
            out->cov_graph_state = 3;
            out->cov_graph_sites = first_plausible;
          }
        }
      }
      FILE *g = fopen(join_path(d, "fm_index").c_str(), "rb");
      if (g) {
        if (fseek(g, 0, SEEK_END) == 0) out->fm_index_bytes = (uint64_t)std::max<long>(ftell(g), 0);
        fclose(g);
      }
    }
    return GMX_OK;
  } catch (std::exception const &e) {
    return fail(e.what());
  }
} GMX_GUARD_INT("gmx_index_check_stock_files")

}  // extern "C"
