// gmx_infer.cpp — the infer stage of `gram genotype` on the host: level genotyping of every
// site from the coverage the GPU recorded, jVCF JSON, VCF (BGZF-compressed) and the personalised reference.
//
// Follows libgramtools/src/genotype/infer/** on the flat graph of gmx_index.h:
//   allele extraction            allele_extracter.cpp:17-124 (nested sites contribute their called alleles)
//   likelihood model             level_genotyping/model.cpp:18-462, probabilities.cpp:8-39
//   per-PRG driver, invalidation level_genotyping/runner.cpp:29-337 (most nested sites first: bubble_map order)
//   genotype confidence percentile  lib/GCP/GCP.h (simulation seeded 42 + std::default_random_engine, as there)
//   outputs                      output_specs/make_json.cpp, make_vcf.cpp (htslib's text form written directly, bgzip
//                                blocks through zlib: htslib is not available), personalised_reference.cpp
// Float arithmetic is done in the reference's order and types (uint16 coverage counts, doubles), so the calls and
// confidences are the same numbers. This is per-site host work, a few microseconds per site: not a GPU path.
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <exception>
#include <unordered_map>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_internal.h"

namespace {
// contiguous ranges of [0, n) on up to 16 threads (fn(begin, end, part)); one range when n is small. Used where the
// reference's loops are independent per site: genotyping the sites of a non-nested PRG, formatting the output records.
template <class F>
void par_ranges(size_t n, size_t min_per_part, F fn) {
  size_t parts = std::min<size_t>(std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16), n / std::max<size_t>(min_per_part, 1) + 1);
  if (const char *e = getenv("GMX_INFER_THREADS")) parts = std::max(1, atoi(e));
  parts = std::max<size_t>(1, std::min(parts, std::max<size_t>(n, 1)));
  if (parts == 1) {
    fn((size_t)0, n, (size_t)0);
    return;
  }
  std::vector<std::exception_ptr> err(parts);
  {
    GmxThreads th;  // (joined also when a thread cannot be started)
    for (size_t p = 0; p < parts; ++p)
      th.run([&, p]() {
        try {
          fn(n * p / parts, n * (p + 1) / parts, p);
        } catch (...) {
          err[p] = std::current_exception();
        }
      });
  }
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
}
// GMX_PHASE_TRACE=1: wall time of the stage's parts on stderr (milliseconds since the first call)
void infer_phase(const char *what) {
  static const bool on = getenv("GMX_PHASE_TRACE") != nullptr;
  if (!on) return;
  static const auto t0 = std::chrono::steady_clock::now();
  fprintf(stderr, "[infer %8.2f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}
size_t par_parts(size_t n, size_t min_per_part) {  // how many ranges par_ranges will make (callers size their per-part buffers)
  size_t parts = std::min<size_t>(std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16), n / std::max<size_t>(min_per_part, 1) + 1);
  if (const char *e = getenv("GMX_INFER_THREADS")) parts = std::max(1, atoi(e));
  return std::max<size_t>(1, std::min(parts, std::max<size_t>(n, 1)));
}

using CovCount = uint16_t;  // common/data_types.hpp:52
using AlleleIds = std::vector<int32_t>;
using GroupedCounts = std::map<AlleleIds, CovCount>;  // the reference iterates an unordered_map: only integer sums depend on it
using Gt = std::vector<int32_t>;

struct Allele {  // infer/types.hpp:10-63
  std::string seq;
  std::vector<CovCount> pb;
  int32_t hapg = 0;
  bool callable = true;
  Allele operator+(const Allele &o) const {
    Allele r;
    r.seq = seq + o.seq;
    r.pb = pb;
    r.pb.insert(r.pb.end(), o.pb.begin(), o.pb.end());
    r.hapg = hapg;
    r.callable = callable & o.callable;
    return r;
  }
  bool operator<(const Allele &o) const { return seq < o.seq; }
  bool operator==(const Allele &o) const { return seq == o.seq && pb == o.pb && hapg == o.hapg; }
  double average_cov() const {
    double r = std::accumulate(pb.begin(), pb.end(), 0.0);
    return r / pb.size();
  }
};
using Alleles = std::vector<Allele>;

struct Site {  // GenotypedSite + LevelGenotypedSite (interfaces.hpp:49-132, level_genotyping/site.hpp)
  Alleles alleles;
  Gt genotype;
  std::vector<double> covs;
  size_t total_cov = 0;
  AlleleIds haplogroups;
  std::vector<std::string> filters;
  size_t pos = 0;
  uint32_t end_node = 0;
  size_t num_haplogroups = 0;
  std::optional<Alleles> extra;
  double gt_conf = 0., gt_conf_percentile = 0.;
  std::string debug_info;  // --debug: next best genotype's sequences and coverages (model.cpp:452-465)
  bool is_null() const { return !genotype.empty() && genotype[0] == -1; }
  void make_null() {
    genotype = Gt{-1};
    total_cov = 0;
    gt_conf = gt_conf_percentile = 0.;
  }
  bool has_filter(const std::string &n) const { return std::find(filters.begin(), filters.end(), n) != filters.end(); }
  Alleles unique_genotyped(const Alleles &all, const Gt &gt) const {  // interfaces.cpp:13-30
    std::set<int32_t> distinct;
    if (!is_null()) distinct.insert(gt.begin(), gt.end());
    Alleles r;
    for (int32_t g : distinct) r.push_back(all.at(g));
    return r;
  }
  Alleles unique_genotyped() const { return unique_genotyped(alleles, genotype); }
};

// ---- probabilities.cpp ----------------------------------------------------------------------------------------------
struct Pmf {
  bool negbinom = false;
  double lambda = 0, k = 0, p = 0;
  double operator()(double cov) const {
    // (lgamma_r: the same values as lgamma — glibc's lgamma is lgamma_r plus a store to the global `signgam`, a formal data race
    //  when the sites of a non-nested PRG are genotyped side by side)
    int sg = 0;
    if (!negbinom) return (-1 * lambda + cov * log(lambda) - lgamma_r(cov + 1, &sg));
    return (lgamma_r(k + cov, &sg) - lgamma_r(cov + 1, &sg) - lgamma_r(k, &sg) + k * log(p) + cov * log(1 - p));
  }
};
struct LStats {  // likelihood_related_stats
  double mean_cov = -1, mean_pb_error = -1, num_successes = -1, success_prob = -1;
  double log_mean_pb_error = 0, log_zero = 0, log_zero_half_depth = 0, log_no_zero = 0, log_no_zero_half_depth = 0;
  CovCount credible_cov_t = 1;
  Pmf pmf_full, pmf_half;
};

CovCount find_minimum_non_error_cov(double mean_pb_error, const Pmf &pmf) {  // runner.cpp:241-252
  double min_count{1};
  if (std::isinf(pmf(min_count))) return (CovCount)min_count;
  while (pmf(min_count) <= min_count * log(mean_pb_error)) ++min_count;
  return (CovCount)min_count;
}

LStats make_l_stats(double mean_cov, double var_cov, double mean_pb_error) {  // runner.cpp:199-239
  LStats l;
  l.mean_cov = mean_cov;
  l.mean_pb_error = mean_pb_error;
  if (var_cov > mean_cov) {
    double num_successes = pow(mean_cov, 2) / (var_cov - mean_cov);
    double success_prob = num_successes / (mean_cov + num_successes);
    l.pmf_full = Pmf{true, 0, num_successes, success_prob};
    l.log_no_zero = log(1 - pow(success_prob, num_successes));
    l.num_successes = num_successes;
    l.success_prob = success_prob;
    num_successes = pow(var_cov, 2) / (var_cov - mean_cov / 2);
    success_prob = num_successes / (mean_cov / 2 + num_successes);
    l.pmf_half = Pmf{true, 0, num_successes, success_prob};
    l.log_no_zero_half_depth = log(1 - pow(success_prob, num_successes));
  } else {
    l.pmf_full = Pmf{false, mean_cov, 0, 0};
    l.log_no_zero = log(1 - exp(mean_cov * -1));
    l.pmf_half = Pmf{false, mean_cov / 2, 0, 0};
    l.log_no_zero_half_depth = log(1 - exp(mean_cov * -0.5));
  }
  l.log_mean_pb_error = log(mean_pb_error);
  l.log_zero = l.pmf_full(0);
  l.log_zero_half_depth = l.pmf_half(0);
  l.credible_cov_t = find_minimum_non_error_cov(mean_pb_error, l.pmf_full);
  return l;
}

// ---- model.cpp ------------------------------------------------------------------------------------------------------
struct Model {
  const Alleles &input;
  const GroupedCounts &gp;
  int ploidy;  // 1 or 2
  const LStats &ls;
  std::vector<CovCount> haploid, singleton;
  std::map<AlleleIds, std::vector<double>> computed;
  size_t total_cov = 0;
  std::multimap<double, Gt, std::greater<>> likelihoods;
  Site site;

  bool ignore_ref() const { return !input.at(0).callable; }

  static std::vector<bool> multiplicities(const Alleles &in) {  // model.cpp:212-226
    std::map<int32_t, size_t> counts;
    for (auto const &a : in) counts[a.hapg] += 1;
    std::vector<bool> m(counts.size(), false);
    for (auto const &e : counts)
      if (e.second > 1) m.at(e.first) = true;
    return m;
  }
  void set_haploid(size_t num_haplogroups) {  // :62-77
    haploid.assign(num_haplogroups, 0);
    singleton.assign(num_haplogroups, 0);
    for (auto const &e : gp) {
      for (int32_t id : e.first) haploid.at(id) += e.second;
      if (e.first.size() == 1) singleton.at(e.first[0]) = e.second;
    }
  }
  std::pair<double, double> diploid_cov(AlleleIds h, const std::vector<bool> &mults) {  // :147-166, :89-145
    std::sort(h.begin(), h.end());
    auto f = computed.find(h);
    if (f != computed.end()) return {f->second.at(0), f->second.at(1)};
    if (h.at(0) == h.at(1)) {
      double c = (double)(haploid.at(h.at(0)));
      c /= 2;
      computed.insert({h, {c, c}});
      return {c, c};
    }
    const int32_t id1 = h.at(0), id2 = h.at(1);
    double c1 = (double)(haploid.at(id1)), c2 = (double)(haploid.at(id2));
    CovCount shared{0};
    for (auto const &e : gp) {
      bool has1 = std::find(e.first.begin(), e.first.end(), id1) != e.first.end();
      bool has2 = std::find(e.first.begin(), e.first.end(), id2) != e.first.end();
      if (has1 && has2) shared += e.second;
    }
    double s1 = c1 - shared, s2 = c2 - shared;
    double belonging;
    if (s1 == 0 && s2 == 0)
      belonging = 0.5;
    else
      belonging = s1 / (s1 + s2);
    c1 -= (1 - belonging) * shared;
    c2 -= belonging * shared;
    if (mults.at(id1)) c1 /= 2;
    if (mults.at(id2)) c2 /= 2;
    computed.insert({h, {c1, c2}});
    return {c1, c2};
  }
  double fraction_noncredible(const Allele &a) const {  // :168-178
    double n{0.};
    for (auto c : a.pb)
      if (c < ls.credible_cov_t) ++n;
    return n / a.pb.size();
  }
  void add_likelihood(const Alleles &als, double incompatible, const Gt &idx) {  // :262-294
    double ll = incompatible * ls.log_mean_pb_error;
    for (int i = 0; i < ploidy; ++i) {
      const Allele &a = als.at(i);
      double compatible = a.average_cov();
      double gap = fraction_noncredible(a);
      ll += ls.pmf_full(compatible);
      ll += gap * ls.log_zero;
    }
    likelihoods.insert({ll, idx});
  }
  AlleleIds haplogroups_of(const Alleles &als, const Gt &gt) const {  // :201-210
    AlleleIds r;
    for (int32_t i : gt) r.push_back(als.at(i).hapg);
    std::sort(r.begin(), r.end());
    return r;
  }
  static Gt rescale(const Gt &g) {  // :228-245
    std::unordered_map<int32_t, int32_t> m{{0, 0}};
    Gt r;
    int32_t next{1};
    for (int32_t x : g) {
      if (m.find(x) == m.end()) m.insert({x, next++});
      r.push_back(m.at(x));
    }
    return r;
  }

  // get_permutations, :247-260: the `size`-subsets of `idx` in the order std::prev_permutation of a selector gives them
  static std::vector<Gt> permutations(const Gt &idx, size_t size) {
    std::vector<Gt> r;
    const size_t n = idx.size();
    if (n < size) return r;
    std::vector<bool> v(n);
    std::fill(v.begin(), v.begin() + (long)size, true);
    do {
      Gt combo;
      for (size_t i = 0; i < n; ++i)
        if (v[i]) combo.push_back(idx.at(i));
      std::sort(combo.begin(), combo.end());
      r.push_back(combo);
    } while (std::prev_permutation(v.begin(), v.end()));
    return r;
  }
  // the pieces alone (test hook gmx_infer_debug: the reference's unit tests call them one by one)
  struct Bare {};
  Model(const Alleles &in, const GroupedCounts &g, int pl, const LStats &l, Bare) : input(in), gp(g), ploidy(pl), ls(l) {
    for (auto const &e : gp) total_cov += e.second;
  }

  Model(const Alleles &in, const GroupedCounts &g, int pl, const LStats &l) : input(in), gp(g), ploidy(pl), ls(l) {  // :18-60
    const Allele &ref = input.at(0);
    auto mults = multiplicities(input);
    site.num_haplogroups = mults.size();
    {  // check_for_duplicates, :8-16
      std::set<Allele> nodups;
      for (auto const &a : input)
        if (!nodups.insert(a).second) {
          site.filters.emplace_back("AMBIG");
          break;
        }
    }
    for (auto const &e : gp) total_cov += e.second;
    if (total_cov == 0 || ls.mean_cov == 0) {
      site.alleles = Alleles{ref};
      site.make_null();
      return;
    }
    set_haploid(mults.size());
    Alleles used(input);
    for (auto &a : used)
      if (a.seq.empty()) a.pb = {haploid.at(a.hapg)};  // assign_coverage_to_empty_alleles, :79-87
    int32_t idx = -1;
    if (ploidy == 1) {  // compute_haploid_log_likelihoods, :296-308
      for (auto const &a : used) {
        ++idx;
        if (idx == 0 && ignore_ref()) continue;
        double incompatible = (double)(total_cov - haploid.at(a.hapg));
        add_likelihood(Alleles{a}, incompatible, Gt{idx});
      }
    } else {
      for (auto const &a : used) {  // compute_homozygous_log_likelihoods, :310-327
        ++idx;
        if (idx == 0 && ignore_ref()) continue;
        auto c = diploid_cov(AlleleIds{a.hapg, a.hapg}, mults);
        double incompatible = total_cov - c.first - c.second;
        add_likelihood(Alleles{a, a}, incompatible, Gt{idx, idx});
      }
      Gt selected;  // compute_heterozygous_log_likelihoods, :329-359
      idx = -1;
      for (auto const &a : used) {
        ++idx;
        if (idx == 0 && ignore_ref()) continue;
        if (singleton.at(a.hapg) != 0) selected.push_back(idx);
      }
      if (selected.size() >= 2)
        for (auto const &combo : permutations(selected, 2)) {
          Allele a1 = used.at(combo.at(0)), a2 = used.at(combo.at(1));
          auto c = diploid_cov(AlleleIds{a1.hapg, a2.hapg}, mults);
          double incompatible = total_cov - c.first - c.second;
          add_likelihood(Alleles{a1, a2}, incompatible, combo);
        }
    }
    call(mults);
  }

  void call(const std::vector<bool> &mults) {  // CallGenotype :405-462, ChooseMaxLikelihood :381-403
    const Allele &ref = input.at(0);
    if (likelihoods.size() < 2) throw std::runtime_error("Less than 2 alleles have a likelihood.\nAllele extraction bug?");
    auto it = likelihoods.begin();
    while (it != likelihoods.end()) {
      bool callable = true;
      for (int32_t g : it->second)
        if (!input.at(g).callable) {
          callable = false;
          break;
        }
      if (callable) break;
      ++it;
    }
    if (std::distance(it, likelihoods.end()) < 2) throw std::runtime_error("Fewer than 2 alleles are callable.\nAllele extraction bug?");
    const double best = it->first;
    const Gt chosen = it->second;
    ++it;
    const double conf = best - it->first;
    const Gt next_best = it->second;
    if (conf == 0.) {
      site.alleles = Alleles{ref};
      site.make_null();
      std::set<int32_t> all(next_best.begin(), next_best.end());  // add_all_best_alleles, :391-399 of model.cpp
      all.insert(chosen.begin(), chosen.end());
      Alleles r;
      for (int32_t g : all) r.push_back(input.at(g));
      site.extra = r;
      return;
    }
    {  // add_next_best_alleles, :361-389
      const Allele &ca = input.at(chosen.at(0)), &na = input.at(next_best.at(0));
      bool low_total = total_cov < ls.mean_cov / 4;
      bool low_relative = haploid.at(ca.hapg) < haploid.at(na.hapg) * 2;
      if (low_total || low_relative) {
        std::set<int32_t> nb(next_best.begin(), next_best.end());
        for (int32_t g : chosen) nb.erase(g);
        Alleles r;
        for (int32_t g : nb) {
          Allele a = input.at(g);
          a.callable = false;
          r.push_back(a);
        }
        site.extra = r;
      }
    }
    site.genotype = chosen;  // not null: unique_genotyped uses it
    Alleles chosen_alleles = site.unique_genotyped(input, chosen);
    AlleleIds chosen_h = haplogroups_of(input, chosen);
    std::vector<double> covs;
    if (ploidy == 1)
      covs = {(double)haploid.at(chosen_h.at(0))};
    else {
      covs = computed.at(chosen_h);
      if (chosen.at(0) == chosen.at(1)) covs = {covs.at(0) + covs.at(1)};
    }
    Gt rescaled = rescale(chosen);
    if (rescaled.at(0) != 0) {
      chosen_alleles.insert(chosen_alleles.begin(), ref);
      double ref_cov = (double)singleton.at(0);
      if (mults.at(0)) ref_cov /= 2;
      covs.insert(covs.begin(), ref_cov);
    }
    site.alleles = chosen_alleles;
    site.genotype = rescaled;
    site.covs = covs;
    site.total_cov = total_cov;
    site.haplogroups.clear();
    for (int32_t g : rescaled) site.haplogroups.push_back(chosen_alleles.at(g).hapg);
    site.gt_conf = conf;
    {  // the --debug line of this site (model.cpp:452-465)
      std::string d = "\tnext_best_seq: ";
      for (int32_t g : next_best) d += input.at(g).seq + ",";
      d += "\tnext_best_cov: ";
      for (int32_t hg : haplogroups_of(input, next_best)) d += std::to_string(haploid.at(hg)) + ",";
      site.debug_info = d;
    }
  }
};

// ---- the genotyper over one PRG (runner.cpp) --------------------------------------------------------------------------
struct Genotyper {
  const gmx::HostIndex &h;
  std::vector<std::shared_ptr<Site>> recs;
  std::map<uint32_t, std::map<int32_t, std::vector<uint32_t>>> child_m;  // build_child_map, make_data_structures.cpp:53-68
  LStats ls;
  int ploidy = 1;
  std::string debug_text;  // site_gtyping_debug_info.txt (runner.cpp:66-75): one line per site, in genotyping order
  std::vector<uint32_t> per_base;  // final uint16 values, logical layout

  explicit Genotyper(const gmx::HostIndex &hi) : h(hi) {}

  std::string node_seq(uint32_t node) const {
    const GmxNode &n = h.nodes[node];
    std::string s(n.seq_len, 'A');
    for (uint32_t i = 0; i < n.seq_len; ++i) s[i] = "ACGT"[h.prg[n.first_pos + i] - 1];
    return s;
  }
  std::vector<CovCount> node_cov(uint32_t node) const {
    const GmxNode &n = h.nodes[node];
    std::vector<CovCount> c(n.seq_len, 0);  // sequence outside bubbles owns no counters: zero coverage
    const uint32_t off = h.l_cov_off[node];
    if (off != GMX_NO_COV)
      for (uint32_t i = 0; i < n.seq_len; ++i) c[i] = (CovCount)per_base[off + i];
    return c;
  }
  bool bubble_start(uint32_t node) const { return h.nodes[node].n_edges > 1 && h.nodes[node].seq_len == 0; }
  uint32_t first_edge(uint32_t node) const { return h.nodes[node].edge0; }

  Alleles combine(const Alleles &existing, size_t site_index) const {  // allele_combine, allele_extracter.cpp:33-66
    const Site &ref_site = *recs.at(site_index);
    Alleles relevant = ref_site.unique_genotyped();
    if (ref_site.extra) relevant.insert(relevant.end(), ref_site.extra->begin(), ref_site.extra->end());
    if (relevant.empty()) relevant.push_back(ref_site.alleles.at(0));
    while (existing.size() * relevant.size() > 10000) relevant.resize(relevant.size() - 1);
    Alleles out;
    out.reserve(existing.size() * relevant.size());
    for (auto const &a : existing)
      for (auto const &b : relevant) out.push_back(a + b);
    return out;
  }
  Allele ref_allele(uint32_t start, uint32_t end) const {  // extract_ref_allele, :88-101
    Allele r;
    uint32_t cur = start;
    while (cur != end) {
      if (h.nodes[cur].seq_len) {
        Allele piece;
        piece.seq = node_seq(cur);
        piece.pb = node_cov(cur);
        r = r + piece;
      }
      cur = first_edge(cur);
    }
    return r;
  }
  Alleles extract_haplogroup(int32_t hapg, uint32_t start, uint32_t site_end) const {  // extract_alleles, :103-139
    Alleles out(1);
    out[0].hapg = hapg;
    uint32_t cur = start;
    while (cur != site_end) {
      if (bubble_start(cur)) {
        const size_t si = (h.nodes[cur].site - 5) / 2;
        out = combine(out, si);
        cur = recs.at(si)->end_node;
      } else {
        Allele piece;
        piece.seq = node_seq(cur);
        piece.pb = node_cov(cur);
        for (auto &a : out) a = a + piece;
      }
      cur = first_edge(cur);
    }
    if (hapg == 0) {  // place_ref_as_first_allele, :77-86
      Allele ref = ref_allele(start, site_end);
      auto found = std::find(out.begin(), out.end(), ref);
      if (found == out.end()) {
        ref.callable = false;
        out.insert(out.begin(), ref);
      } else if (found != out.begin())
        std::swap(*found, out.at(0));
    }
    return out;
  }
  Alleles extract(uint32_t site_index) const {  // AlleleExtracter ctor, :17-31
    const GmxSite &s = h.sites[site_index];
    const GmxNode &entry = h.nodes[s.entry_node];
    Alleles all;
    for (uint32_t e = 0; e < entry.n_edges; ++e) {
      const uint32_t start = h.edges[entry.edge_begin + e];
      Alleles part = extract_haplogroup((int32_t)e, start, s.exit_node);
      all.insert(all.end(), part.begin(), part.end());
    }
    return all;
  }

  AlleleIds haplogroups_with_sites(uint32_t site_id, const AlleleIds &cands) const {  // runner.cpp:157-168
    AlleleIds r;
    auto f = child_m.find(site_id);
    if (f == child_m.end()) return r;
    for (int32_t c : cands)
      if (f->second.find(c) != f->second.end()) r.push_back(c);
    return r;
  }
  void invalidate(uint32_t parent, const AlleleIds &hapgs) {  // invalidate_if_needed, :170-197
    if (hapgs.empty()) return;
    std::vector<std::pair<uint32_t, int32_t>> todo;
    for (int32_t hgp : hapgs) todo.emplace_back(parent, hgp);
    while (!todo.empty()) {
      auto cur = todo.back();
      todo.pop_back();
      auto sites_on = child_m.at(cur.first).at(cur.second);
      for (uint32_t child : sites_on) {
        Site &rs = *recs.at((child - 5) / 2);
        if (rs.is_null()) continue;
        rs.make_null();
        AlleleIds all;
        for (size_t i = 0; i < rs.num_haplogroups; ++i) all.push_back((int32_t)i);
        for (int32_t hg : haplogroups_with_sites(child, all)) todo.emplace_back(child, hg);
      }
    }
  }

  void run(const std::vector<GroupedCounts> &gped, double mean_cov, double var_cov, double err, int pl) {  // LevelGenotyper ctor, :29-103
    ploidy = pl;
    const size_t n_sites = h.sites.size();
    for (size_t s = 0; s < n_sites; ++s)
      if (h.sites[s].parent_site != 0) child_m[h.sites[s].parent_site][h.sites[s].parent_allele].push_back(5 + 2 * (uint32_t)s);
    recs.assign(n_sites, nullptr);
    ls = make_l_stats(mean_cov, var_cov, err);
    // bubble_map order: descending position of the bubble start, then descending site id (prg/types.hpp:26,
    // coverage_graph.cpp:381-389): every site comes after the sites nested in it
    std::vector<uint32_t> order(n_sites);
    for (size_t s = 0; s < n_sites; ++s) order[s] = (uint32_t)s;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      if (h.site_ref_pos[a] != h.site_ref_pos[b]) return h.site_ref_pos[a] > h.site_ref_pos[b];
      return a > b;
    });
    infer_phase("  sites ordered");
    if (child_m.empty()) {
      // A non-nested PRG: no site reads another site's result (no invalidation, no filter propagation), so the sites are
      // genotyped side by side — the same arithmetic per site, in the same order within it; the debug text is put together
      // in genotyping order afterwards. (60 k sites: 72 ms on one thread of the GPU box.)
      std::vector<std::string> dbg(n_sites);
      par_ranges(n_sites, 512, [&](size_t b, size_t e, size_t) {
        for (size_t k = b; k < e; ++k) {
          const uint32_t si = order[k];
          Alleles extracted = extract(si);
          Model m(extracted, gped.at(si), ploidy, ls);
          auto site = std::make_shared<Site>(m.site);
          dbg[k] = "site index: \t" + std::to_string(si) + (site->is_null() ? std::string("\tnull gt \n") : site->debug_info + "\n");
          site->pos = h.site_ref_pos[si];
          site->end_node = h.sites[si].exit_node;
          recs[si] = site;
        }
      });
      infer_phase("  sites genotyped (side by side)");
      for (auto const &d : dbg) debug_text += d;
      infer_phase("  debug text joined");
      add_percentiles();
      return;
    }
    for (uint32_t si : order) {
      const uint32_t site_id = 5 + 2 * si;
      Alleles extracted = extract(si);
      Model m(extracted, gped.at(si), ploidy, ls);
      auto site = std::make_shared<Site>(m.site);
      debug_text += "site index: \t" + std::to_string(si) + (site->is_null() ? std::string("\tnull gt \n") : site->debug_info + "\n");
      site->pos = h.site_ref_pos[si];
      site->end_node = h.sites[si].exit_node;
      recs.at(si) = site;
      if (child_m.find(site_id) != child_m.end()) {  // run_invalidation_process, :146-155
        AlleleIds non_genotyped;  // get_nonGenotyped_haplogroups, site.cpp:6-22
        std::set<int32_t> genotyped;
        if (!site->is_null())
          for (int32_t g : site->genotype) genotyped.insert(site->alleles.at(g).hapg);
        for (size_t i = 0; i < site->num_haplogroups; ++i)
          if (!genotyped.count((int32_t)i)) non_genotyped.push_back((int32_t)i);
        invalidate(site_id, haplogroups_with_sites(site_id, non_genotyped));
      }
      if (site->has_filter("AMBIG")) {  // downpropagate_filter, :123-144
        std::vector<uint32_t> todo{site_id};
        while (!todo.empty()) {
          uint32_t cur = todo.back();
          todo.pop_back();
          auto f = child_m.find(cur);
          if (f == child_m.end()) continue;
          for (auto const &pair : f->second)
            for (uint32_t child : pair.second) {
              Site &rs = *recs.at((child - 5) / 2);
              if (!rs.has_filter("AMBIG")) {
                rs.filters.emplace_back("AMBIG");
                todo.push_back(child);
              }
            }
        }
      } else {  // uppropagate_filter, :107-121
        auto f = child_m.find(site_id);
        if (f != child_m.end()) {
          bool done = false;
          for (auto const &pair : f->second) {
            for (uint32_t child : pair.second)
              if (recs.at((child - 5) / 2)->has_filter("AMBIG")) {
                site->filters.emplace_back("AMBIG");
                done = true;
                break;
              }
            if (done) break;
          }
        }
      }
    }
    add_percentiles();
  }

  // get_gtconf_distrib (runner.cpp:299-337) + GCP::Percentiler (lib/GCP/GCP.h:101-175)
  void add_percentiles() {
    constexpr size_t N = 10000;
    std::vector<double> conf(N);
    size_t at = 0;
    if (recs.size() > N) {
      std::mt19937 generator(42);  // the reference seeds from std::random_device here: not reproducible there either
      std::uniform_int_distribution<> distrib(0, (int)recs.size() - 1);
      while (at < N) conf[at++] = recs.at(distrib(generator))->gt_conf;
    } else {
      for (auto const &s : recs) conf[at++] = s->gt_conf;
      std::default_random_engine rng(42);  // GCP::Model's generator
      std::vector<double> sim;
      for (size_t i = at; i < N; ++i) {  // ModelDataProducer::produce_data, runner.cpp:254-293
        CovCount correct;
        if (!ls.pmf_full.negbinom) {
          std::poisson_distribution<CovCount> d(ls.mean_cov);
          correct = d(rng);
        } else {
          std::negative_binomial_distribution<CovCount> d(ls.num_successes, ls.success_prob);
          correct = d(rng);
        }
        std::binomial_distribution<CovCount> b(ls.mean_cov, ls.mean_pb_error);
        const CovCount incorrect = b(rng);
        Alleles als(2);
        als[0].seq = "C";
        als[0].pb = {correct};
        als[0].hapg = 0;
        als[1].seq = "A";
        als[1].pb = {incorrect};
        als[1].hapg = 1;
        GroupedCounts g{{{0}, correct}, {{1}, incorrect}};
        Model m(als, g, ploidy, ls);
        sim.push_back(m.site.gt_conf);
      }
      std::sort(sim.begin(), sim.end());  // GCP::Simulator::simulate sorts its output
      std::copy(sim.begin(), sim.end(), conf.begin() + at);
    }
    std::sort(conf.begin(), conf.end());
    std::map<double, double> entries;
    auto pct = [&](size_t i) { return 100. * (double)(i + 1) / (double)conf.size(); };
    for (size_t i = 0; i < conf.size();) {
      size_t hi = std::upper_bound(conf.begin(), conf.end(), conf[i]) - conf.begin();
      double cur = pct(i);
      if (i == hi - 1)
        entries[conf[i]] = cur;
      else
        entries[conf[i]] = cur + (pct(hi - 1) - cur) / 2;
      i = hi;
    }
    for (auto &s : recs) {
      const double q = s->gt_conf;
      auto lb = entries.upper_bound(q);
      double r;
      if (lb == entries.end())
        r = 100.0;
      else if (lb->first == q)
        r = lb->second;
      else if (lb == entries.begin())
        r = 0.0;
      else {
        auto hi = lb;
        --lb;
        r = lb->second + (hi->second - lb->second) / (hi->first - lb->first) * (q - lb->first);
      }
      s->gt_conf_percentile = r;
    }
  }
};

// ---- outputs ----------------------------------------------------------------------------------------------------------
struct Segment {
  std::string id;
  size_t size;
};
struct Tracker {  // SegmentTracker, output_specs/segment_tracker.hpp
  std::vector<Segment> segs;
  size_t min = 0, max = 0, global_max = 0, cur = 0;
  explicit Tracker(std::istream *in) {
    Segment next{"gramtools_prg", std::numeric_limits<size_t>::max()};
    if (in)
      while (*in >> next.id >> next.size) {
        global_max += next.size;
        segs.push_back(next);
      }
    if (segs.empty()) {
      segs.push_back(Segment{"gramtools_prg", std::numeric_limits<size_t>::max()});
      global_max = std::numeric_limits<size_t>::max();
    }
    max = segs[0].size - 1;
  }
  const std::string &id_of(size_t pos) {
    // (the reference asserts: positions are queried in ascending order within the segments' total size)
    if (pos < min || pos >= global_max) throw std::runtime_error("segment tracker: position queried backwards or beyond the last segment");
    while (pos > max) {
      cur++;
      min = max + 1;
      max += segs.at(cur).size;
    }
    return segs.at(cur).id;
  }
  size_t relative(size_t pos) const {
    if (pos < min || pos >= global_max) throw std::runtime_error("segment tracker: position outside the current segment's reach");
    return pos - min;
  }
  size_t edge() const { return max; }
  size_t global_edge() const { return global_max - 1; }
  void reset() {
    min = 0;
    cur = 0;
    max = segs.at(0).size - 1;
  }
};

std::string jstr(const std::string &s) {
  std::string o = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') {
      o += '\\';
      o += c;
    } else if (c == '\n')
      o += "\\n";
    else if (c == '\t')
      o += "\\t";
    else
      o += c;
  }
  return o + "\"";
}
std::string jdouble(double v) {  // nlohmann::json's float form: shortest round trip, ".0" on integral values
  if (!std::isfinite(v)) return "null";
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), v);
  std::string s(buf, r.ptr);
  if (s.find_first_of(".e") == std::string::npos) s += ".0";
  return s;
}

const char *kGtConfDesc = "Genotype confidence as likelihood ratio of called and next most likely genotype.";
const char *kGcpDesc = "Percent of calls expected to have lower GT_CONF";
const char *kAmbigDesc = "Ambiguous site. Different variant paths can produce the same sequence.";

std::string site_json(const Site &s, const std::string *seg, size_t pos1) {  // make_json_site, make_json.cpp:60-82
  std::ostringstream o;
  o << "{\"ALS\":[";
  for (size_t i = 0; i < s.alleles.size(); ++i) o << (i ? "," : "") << jstr(s.alleles[i].seq);
  o << "],\"COV\":[[";
  for (size_t i = 0; i < s.covs.size(); ++i) o << (i ? "," : "") << jdouble(s.covs[i]);
  o << "]],\"DP\":[" << s.total_cov << "],\"FT\":[[";
  for (size_t i = 0; i < s.filters.size(); ++i) o << (i ? "," : "") << jstr(s.filters[i]);
  o << "]],\"GT\":[[";
  if (s.is_null())
    o << "null";
  else
    for (size_t i = 0; i < s.genotype.size(); ++i) o << (i ? "," : "") << s.genotype[i];
  o << "]],\"GT_CONF\":[" << jdouble(s.gt_conf) << "],\"GT_CONF_PERCENTILE\":[" << jdouble(s.gt_conf_percentile) << "],\"HAPG\":[[";
  for (size_t i = 0; i < s.haplogroups.size(); ++i) o << (i ? "," : "") << s.haplogroups[i];
  o << "]],\"POS\":" << pos1 << ",\"SEG\":" << jstr(seg ? *seg : std::string("")) << "}";
  return o.str();
}

// BGZF (the bgzip container htslib writes for "wz"): gzip members of <= 64 KiB with a BC extra field, then the EOF block
struct Bgzf {
  std::ofstream out;
  std::string buf;
  explicit Bgzf(const std::string &path) : out(path, std::ios::binary) {}
  bool ok() const { return out.good(); }
  void block(const char *data, size_t n) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    std::vector<unsigned char> comp(n + n / 8 + 1024);
    zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(data));
    zs.avail_in = (uInt)n;
    zs.next_out = comp.data();
    zs.avail_out = (uInt)comp.size();
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint16_t bsize = (uint16_t)(clen + 25);
    const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 0xFF), (unsigned char)(bsize >> 8)};
    out.write(reinterpret_cast<const char *>(head), 18);
    out.write(reinterpret_cast<const char *>(comp.data()), (std::streamsize)clen);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(data), (uInt)n), isize = (uint32_t)n;
    unsigned char tail[8];
    for (int i = 0; i < 4; ++i) {
      tail[i] = (unsigned char)(crc >> (8 * i));
      tail[4 + i] = (unsigned char)(isize >> (8 * i));
    }
    out.write(reinterpret_cast<const char *>(tail), 8);
  }
  // one BGZF member (header, raw deflate of up to 0xff00 bytes, CRC-32 and length) as bytes
  static std::string member(const char *data, size_t n) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    std::vector<unsigned char> comp(n + n / 8 + 1024);
    zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(data));
    zs.avail_in = (uInt)n;
    zs.next_out = comp.data();
    zs.avail_out = (uInt)comp.size();
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint16_t bsize = (uint16_t)(clen + 25);
    const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 0xFF), (unsigned char)(bsize >> 8)};
    std::string out(reinterpret_cast<const char *>(head), 18);
    out.append(reinterpret_cast<const char *>(comp.data()), clen);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(data), (uInt)n), isize = (uint32_t)n;
    for (int i = 0; i < 4; ++i) out.push_back((char)(unsigned char)(crc >> (8 * i)));
    for (int i = 0; i < 4; ++i) out.push_back((char)(unsigned char)(isize >> (8 * i)));
    return out;
  }
  void write(const std::string &s) {
    buf += s;
    const size_t n_blocks = buf.size() / 0xff00;
    if (n_blocks == 0) return;
    // the same members, in the same order, as one block after the other would give: compressed side by side
    std::vector<std::string> members(n_blocks);
    par_ranges(n_blocks, 4, [&](size_t b, size_t e, size_t) {
      for (size_t i = b; i < e; ++i) members[i] = member(buf.data() + i * 0xff00, 0xff00);
    });
    for (auto const &m : members) out.write(m.data(), (std::streamsize)m.size());
    buf.erase(0, n_blocks * 0xff00);
  }
  void close() {
    if (!buf.empty()) block(buf.data(), buf.size());
    buf.clear();
    block("", 0);  // EOF marker block
    out.close();
  }
};

std::string fmt_g(double v) {  // htslib writes FORMAT floats with %g of the float value
  char b[64];
  snprintf(b, sizeof(b), "%g", (double)(float)v);
  return b;
}

}  // namespace

struct gmx_infer {
  const gmx_index *ix = nullptr;
  std::unique_ptr<Genotyper> g;
};

static int fail(const std::string &m, int code = GMX_EINVAL) {
  gmx_set_error(m);
  return code;
}

// grouped counts per site from the raw totals (dense slots + log), uint16 wrap as the reference's counters
static int grouped_of_sites(const gmx::HostIndex &h, const uint32_t *dense, const uint32_t *glog, uint64_t n_log,
                            std::vector<GroupedCounts> &out) {
  out.assign(h.sites.size(), {});
  for (size_t s = 0; s < h.sites.size(); ++s) {
    if (h.l_grouped_off[s] == GMX_GROUPED_LOG) continue;
    const uint32_t n = h.sites[s].n_alleles, nm = (1u << n) - 1u;
    for (uint32_t m = 0; m < nm; ++m) {
      const uint32_t tot = dense[h.l_grouped_off[s] + m];
      if (!tot) continue;
      AlleleIds ids;
      for (uint32_t a = 0; a < n; ++a)
        if (((m + 1) >> a) & 1u) ids.push_back((int32_t)a);
      out[s][ids] = (CovCount)(tot & 0xFFFFu);
    }
  }
  for (uint64_t i = 0; i < n_log;) {
    if (glog[i] == 0xFFFFFFFFu) {
      ++i;
      continue;
    }
    if (i + 2 > n_log) break;
    const uint32_t s = glog[i], n = glog[i + 1] & ~GMX_LOG_COUNTED;
    const uint64_t head = (glog[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
    if (s >= h.sites.size() || i + head + n > n_log) return fail("corrupt grouped log");
    const uint64_t count = head == 4 ? ((uint64_t)glog[i + 2] | ((uint64_t)glog[i + 3] << 32)) : 1;
    AlleleIds ids(glog + i + head, glog + i + head + n);
    out[s][ids] = (CovCount)(out[s][ids] + count);
    i += head + n;
  }
  return GMX_OK;
}

extern "C" {

int gmx_infer_run(const gmx_index *ix, const uint32_t *per_base_raw, const uint32_t *grouped_dense_raw, const uint32_t *grouped_log,
                  uint64_t n_log_words, double mean_cov_depth, double variance_cov_depth, double mean_pb_error, int ploidy,
                  gmx_infer **out) try {
  if (!ix || !out || (ploidy != 1 && ploidy != 2)) return fail("gmx_infer_run: bad argument (ploidy is 1 or 2)");
  const gmx::HostIndex &h = gmx_index_host(ix);
  try {
    infer_phase("start");
    auto inf = std::make_unique<gmx_infer>();
    inf->ix = ix;
    inf->g = std::make_unique<Genotyper>(h);
    inf->g->per_base.assign(h.n_pb_slots, 0);
    for (uint32_t i = 0; i < h.n_pb_slots; ++i) inf->g->per_base[i] = std::min<uint32_t>(per_base_raw ? per_base_raw[i] : 0, 65535u);
    std::vector<GroupedCounts> gped;
    infer_phase("per-base values copied");
    int rc = grouped_of_sites(h, grouped_dense_raw, grouped_log, n_log_words, gped);
    if (rc) return rc;
    infer_phase("grouped counts per site");
    inf->g->run(gped, mean_cov_depth, variance_cov_depth, mean_pb_error, ploidy);
    infer_phase("sites genotyped, percentiles added");
    *out = inf.release();
    return GMX_OK;
  } catch (std::exception const &ex) {
    return fail(std::string("genotyping: ") + ex.what(), GMX_EREF);
  }
} GMX_GUARD_INT("gmx_infer_run")

void gmx_infer_destroy(gmx_infer *inf) { delete inf; }

// One site as the jVCF site object (tests, and the unit the JSON writer is made of). Returns the length needed.
// site_gtyping_debug_info.txt (`--debug`; genotype/parameters.cpp:98, runner.cpp:66-75): returns the length, copies when it fits.
int64_t gmx_infer_debug_text(const gmx_infer *inf, char *out, uint64_t cap) try {
  if (!inf || !inf->g) return fail("null genotyper");
  const std::string &t = inf->g->debug_text;
  if (out && cap > t.size()) memcpy(out, t.c_str(), t.size() + 1);
  return (int64_t)t.size();
} GMX_GUARD_INT("gmx_infer_debug_text")

int64_t gmx_infer_site_json(const gmx_infer *inf, uint32_t site_index, char *out, uint64_t cap) try {
  if (!inf || site_index >= inf->g->recs.size()) return fail("gmx_infer_site_json: bad argument");
  const Site &s = *inf->g->recs[site_index];
  const std::string js = site_json(s, nullptr, s.pos + 1);
  if (out && cap > js.size()) memcpy(out, js.c_str(), js.size() + 1);
  return (int64_t)js.size();
} GMX_GUARD_INT("gmx_infer_site_json")

// The likelihood model on explicit data (known-answer tests of level_genotyping/test_model.cpp): alleles as
// (sequence, per-base coverage, haplogroup, callable), grouped counts as (ids, count). Writes the site as JSON plus
// "EXTRA":[sequences of extra_alleles_to_consider] and "NUM_HAPG".
int64_t gmx_infer_model(uint32_t n_alleles, const char *const *seqs, const uint32_t *pb_off, const uint32_t *pb_cov,
                        const int32_t *haplogroups, const uint8_t *callable, uint32_t n_groups, const uint32_t *group_off,
                        const int32_t *group_ids, const uint32_t *group_counts, int ploidy, double mean_cov, double var_cov,
                        double mean_pb_error, char *out, uint64_t cap) try {
  try {
    Alleles als(n_alleles);
    for (uint32_t i = 0; i < n_alleles; ++i) {
      als[i].seq = seqs[i];
      for (uint32_t j = pb_off[i]; j < pb_off[i + 1]; ++j) als[i].pb.push_back((CovCount)pb_cov[j]);
      als[i].hapg = haplogroups[i];
      als[i].callable = callable ? callable[i] != 0 : true;
    }
    GroupedCounts gp;
    for (uint32_t g = 0; g < n_groups; ++g) gp[AlleleIds(group_ids + group_off[g], group_ids + group_off[g + 1])] = (CovCount)group_counts[g];
    LStats ls = make_l_stats(mean_cov, var_cov, mean_pb_error);
    Model m(als, gp, ploidy, ls);
    std::string js = site_json(m.site, nullptr, 0);
    js.pop_back();
    js += ",\"EXTRA\":[";
    if (m.site.extra)
      for (size_t i = 0; i < m.site.extra->size(); ++i) js += (i ? "," : "") + jstr((*m.site.extra)[i].seq);
    js += "],\"EXTRA_CALLABLE\":[";
    if (m.site.extra)
      for (size_t i = 0; i < m.site.extra->size(); ++i) js += std::string(i ? "," : "") + ((*m.site.extra)[i].callable ? "true" : "false");
    js += "],\"NUM_HAPG\":" + std::to_string(m.site.num_haplogroups) + ",\"CREDIBLE_COV_T\":" + std::to_string(ls.credible_cov_t) +
          ",\"LOG_ZERO\":" + jdouble(ls.log_zero) + ",\"LOG_NO_ZERO\":" + jdouble(ls.log_no_zero) + "}";
    if (out && cap > js.size()) memcpy(out, js.c_str(), js.size() + 1);
    return (int64_t)js.size();
  } catch (std::exception const &ex) {
    return fail(std::string("genotyping model: ") + ex.what(), GMX_EREF);
  }
} GMX_GUARD_INT("gmx_infer_model")

// Test hook: the model's pieces one by one, as the reference's unit tests call them (tests/genotype/infer/level_genotyping/
// test_model.cpp). Inputs as gmx_infer_model, plus `ids` (meaning per op) and, for GMX_DBG_CALL, a likelihood map.
//   0 INTERNALS   set_haploid_coverages + count_total_coverage + get_haplogroup_multiplicities + assign_coverage_to_empty_alleles
//                 (+ the number of likelihoods of the full model when it can be run)
//   1 DIPLOID     compute_diploid_coverage of haplogroups ids[0], ids[1] after set_haploid_coverages(gp, ids[2]); multiplicities = ids[3..]
//   2 NONCREDIBLE fraction_noncredible_positions of allele ids[0] with credible_cov_t = ids[1]
//   3 PERMUTATIONS get_permutations(ids[1..], ids[0])
//   4 RESCALE     rescale_genotypes(ids)
//   5 CALL        the testing constructor (haploid = singleton = group_counts as per-haplogroup coverages) + CallGenotype with
//                 the likelihood map (lik[i], genotype lik_gt[lik_off[i] .. lik_off[i + 1])) and multiplicities ids
int64_t gmx_infer_debug(int op, uint32_t n_alleles, const char *const *seqs, const uint32_t *pb_off, const uint32_t *pb_cov,
                        const int32_t *haplogroups, const uint8_t *callable, uint32_t n_groups, const uint32_t *group_off,
                        const int32_t *group_ids, const uint32_t *group_counts, int ploidy, double mean_cov, double var_cov,
                        double mean_pb_error, const int32_t *ids, uint32_t n_ids, const double *lik, const uint32_t *lik_off,
                        const int32_t *lik_gt, uint32_t n_lik, char *out, uint64_t cap) try {
  try {
    Alleles als(n_alleles);
    for (uint32_t i = 0; i < n_alleles; ++i) {
      als[i].seq = seqs[i];
      for (uint32_t j = pb_off[i]; j < pb_off[i + 1]; ++j) als[i].pb.push_back((CovCount)pb_cov[j]);
      als[i].hapg = haplogroups[i];
      als[i].callable = callable ? callable[i] != 0 : true;
    }
    GroupedCounts gp;
    if (op != 5)
      for (uint32_t g = 0; g < n_groups; ++g) gp[AlleleIds(group_ids + group_off[g], group_ids + group_off[g + 1])] = (CovCount)group_counts[g];
    LStats ls = make_l_stats(mean_cov, var_cov, mean_pb_error);
    std::string js = "{";
    auto list = [&](const char *name, auto const &v, auto fmt) {
      js += std::string("\"") + name + "\":[";
      bool first = true;
      for (auto const &x : v) {
        js += (first ? "" : ",") + fmt(x);
        first = false;
      }
      js += "]";
    };
    auto num = [](auto x) { return std::to_string((long long)x); };
    if (op == 0) {
      Model m(als, gp, ploidy, ls, Model::Bare{});
      const size_t n_h = n_ids ? (size_t)ids[0] : Model::multiplicities(als).size();
      m.set_haploid(n_h);
      list("HAPLOID", m.haploid, num);
      js += ",";
      list("SINGLETON", m.singleton, num);
      js += ",\"TOTAL_COV\":" + std::to_string(m.total_cov) + ",";
      std::vector<int> mu;
      for (bool b : Model::multiplicities(als)) mu.push_back(b ? 1 : 0);
      list("MULT", mu, num);
      js += ",\"EMPTY_PB\":[";
      for (uint32_t i = 0; i < n_alleles; ++i) {
        std::vector<CovCount> pb = als[i].pb;
        if (als[i].seq.empty() && (size_t)als[i].hapg < m.haploid.size()) pb = {m.haploid.at(als[i].hapg)};
        js += i ? ",[" : "[";
        for (size_t j = 0; j < pb.size(); ++j) js += (j ? "," : "") + std::to_string(pb[j]);
        js += "]";
      }
      js += "],\"N_LIKELIHOODS\":";
      long long n_l = -1;
      try {
        Model full(als, gp, ploidy, ls);
        n_l = (long long)full.likelihoods.size();
      } catch (std::exception const &) {
      }
      js += std::to_string(n_l);
    } else if (op == 1) {
      Model m(als, gp, ploidy, ls, Model::Bare{});
      std::vector<bool> mults;
      for (uint32_t i = 3; i < n_ids; ++i) mults.push_back(ids[i] != 0);
      m.set_haploid((size_t)ids[2]);
      auto c = m.diploid_cov(AlleleIds{ids[0], ids[1]}, mults);
      js += "\"C\":[" + jdouble(c.first) + "," + jdouble(c.second) + "]";
    } else if (op == 2) {
      LStats l2 = ls;
      l2.credible_cov_t = (CovCount)ids[1];
      Model m(als, gp, ploidy, l2, Model::Bare{});
      js += "\"F\":" + jdouble(m.fraction_noncredible(als.at(ids[0])));
    } else if (op == 3) {
      Gt idx(ids + 1, ids + n_ids);
      js += "\"P\":[";
      bool first = true;
      for (auto const &c : Model::permutations(idx, (size_t)ids[0])) {
        js += first ? "[" : ",[";
        first = false;
        for (size_t j = 0; j < c.size(); ++j) js += (j ? "," : "") + std::to_string(c[j]);
        js += "]";
      }
      js += "]";
    } else if (op == 4) {
      list("G", Model::rescale(Gt(ids, ids + n_ids)), num);
    } else if (op == 5) {
      Model m(als, gp, ploidy, ls, Model::Bare{});
      m.haploid.assign(group_counts, group_counts + n_groups);
      m.singleton = m.haploid;
      m.total_cov = 0;
      for (auto c : m.haploid) m.total_cov += c;
      for (uint32_t i = 0; i < n_lik; ++i) m.likelihoods.insert({lik[i], Gt(lik_gt + lik_off[i], lik_gt + lik_off[i + 1])});
      std::vector<bool> mults;
      for (uint32_t i = 0; i < n_ids; ++i) mults.push_back(ids[i] != 0);
      m.site.num_haplogroups = mults.size();
      m.call(mults);
      js = site_json(m.site, nullptr, 0);
      js.pop_back();
      js += ",\"EXTRA\":[";
      if (m.site.extra)
        for (size_t i = 0; i < m.site.extra->size(); ++i) js += (i ? "," : "") + jstr((*m.site.extra)[i].seq);
      js += "],\"EXTRA_CALLABLE\":[";
      if (m.site.extra)
        for (size_t i = 0; i < m.site.extra->size(); ++i) js += std::string(i ? "," : "") + ((*m.site.extra)[i].callable ? "true" : "false");
      js += "],\"HAS_EXTRA\":" + std::string(m.site.extra ? "true" : "false");
    } else {
      return fail("gmx_infer_debug: unknown op");
    }
    js += "}";
    if (out && cap > js.size()) memcpy(out, js.c_str(), js.size() + 1);
    return (int64_t)js.size();
  } catch (std::exception const &ex) {
    return fail(std::string("genotyping model: ") + ex.what(), GMX_EREF);
  }
} GMX_GUARD_INT("gmx_infer_debug")

// Test hook for the segment tracker (output_specs/segment_tracker.hpp; tests/genotype/infer/test_segment_tracker.cpp): `coords` =
// the text of prg_coords.tsv, `script` = commands separated by ';': "id N", "rel N", "edge", "global_edge", "reset". JSON list
// of the answers; a command on which the reference asserts ends the script with GMX_EREF.
int64_t gmx_infer_segments_debug(const char *coords, const char *script, char *out, uint64_t cap) try {
  try {
    std::istringstream in(coords ? coords : "");
    Tracker tr(&in);
    std::string js = "[", cmds = script ? script : "";
    bool first = true;
    for (size_t at = 0; at < cmds.size();) {
      size_t end = cmds.find(';', at);
      if (end == std::string::npos) end = cmds.size();
      const std::string c = cmds.substr(at, end - at);
      at = end + 1;
      if (c.empty()) continue;
      std::string ans;
      if (c.rfind("id ", 0) == 0)
        ans = jstr(tr.id_of(std::stoull(c.substr(3))));
      else if (c.rfind("rel ", 0) == 0)
        ans = std::to_string(tr.relative(std::stoull(c.substr(4))));
      else if (c == "edge")
        ans = std::to_string(tr.edge());
      else if (c == "global_edge")
        ans = std::to_string(tr.global_edge());
      else if (c == "reset") {
        tr.reset();
        ans = "null";
      } else
        return fail("gmx_infer_segments_debug: unknown command");
      js += (first ? "" : ",") + ans;
      first = false;
    }
    js += "]";
    if (out && cap > js.size()) memcpy(out, js.c_str(), js.size() + 1);
    return (int64_t)js.size();
  } catch (std::exception const &ex) {
    return fail(std::string("segment tracker: ") + ex.what(), GMX_EREF);
  }
} GMX_GUARD_INT("gmx_infer_segments_debug")

// Test hook for the allele extracter (infer/allele_extracter.cpp; known answers of tests/genotype/infer/test_allele_extracter.cpp).
// Alleles travel as text "SEQ/c,c,c/haplogroup/callable;...": `existing` for op 2; `mocks` = one line per already-genotyped
// child site: "site_index|g,g (-1 = null)|alleles|extra alleles ('-' = none)". per_base NULL = zero coverage everywhere.
//   0  AlleleExtracter(site).get_alleles()        1  extract_ref_allele(site)        2  allele_combine(existing, site)
int64_t gmx_infer_extract_debug(const gmx_index *ix, int op, uint32_t site_index, const uint32_t *per_base_raw, const char *existing,
                                const char *mocks, char *out, uint64_t cap) try {
  try {
    const gmx::HostIndex &h = gmx_index_host(ix);
    if (site_index >= h.sites.size()) return fail("gmx_infer_extract_debug: no such site");
    auto parse_alleles = [](const std::string &t) {
      Alleles r;
      if (t.empty() || t == "-") return r;
      size_t at = 0;
      while (at <= t.size()) {
        size_t end = t.find(';', at);
        if (end == std::string::npos) end = t.size();
        const std::string one = t.substr(at, end - at);
        std::vector<std::string> f;
        size_t p = 0;
        for (;;) {
          size_t q = one.find('/', p);
          f.push_back(one.substr(p, q == std::string::npos ? std::string::npos : q - p));
          if (q == std::string::npos) break;
          p = q + 1;
        }
        if (f.size() != 4) throw std::runtime_error("allele text: SEQ/cov,cov/haplogroup/callable");
        Allele a;
        a.seq = f[0];
        for (size_t i = 0; i < f[1].size();) {
          size_t j = f[1].find(',', i);
          if (j == std::string::npos) j = f[1].size();
          if (j > i) a.pb.push_back((CovCount)std::stoul(f[1].substr(i, j - i)));
          i = j + 1;
        }
        a.hapg = std::stoi(f[2]);
        a.callable = f[3] != "0";
        r.push_back(a);
        at = end + 1;
      }
      return r;
    };
    Genotyper g(h);
    g.per_base.assign(std::max<size_t>(h.n_pb_slots, 1), 0);
    if (per_base_raw)
      for (size_t i = 0; i < h.n_pb_slots; ++i) g.per_base[i] = std::min<uint32_t>(per_base_raw[i], 65535u);
    g.recs.assign(h.sites.size(), nullptr);
    for (size_t si = 0; si < h.sites.size(); ++si) {  // (as the runner leaves a site before it is genotyped)
      g.recs[si] = std::make_shared<Site>();
      g.recs[si]->end_node = h.sites[si].exit_node;
    }
    const std::string all_mocks = mocks ? mocks : "";
    for (size_t at = 0; at < all_mocks.size();) {
      size_t end = all_mocks.find('\n', at);
      if (end == std::string::npos) end = all_mocks.size();
      const std::string line = all_mocks.substr(at, end - at);
      at = end + 1;
      if (line.empty()) continue;
      std::vector<std::string> f;
      size_t p = 0;
      for (;;) {
        size_t q = line.find('|', p);
        f.push_back(line.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
      }
      if (f.size() != 4) throw std::runtime_error("mock site text: site|genotype|alleles|extra");
      const size_t si = std::stoul(f[0]);
      Site &st = *g.recs.at(si);
      st.genotype.clear();
      for (size_t i = 0; i < f[1].size();) {
        size_t j = f[1].find(',', i);
        if (j == std::string::npos) j = f[1].size();
        if (j > i) st.genotype.push_back(std::stoi(f[1].substr(i, j - i)));
        i = j + 1;
      }
      st.alleles = parse_alleles(f[2]);
      if (f[3] != "-") st.extra = parse_alleles(f[3]);
    }
    Alleles result;
    if (op == 0)
      result = g.extract(site_index);
    else if (op == 1)
      result = Alleles{g.ref_allele(h.sites[site_index].entry_node == 0xFFFFFFFFu ? 0u : h.nodes[h.sites[site_index].entry_node].edge0,
                                    h.sites[site_index].exit_node)};
    else if (op == 2)
      result = g.combine(parse_alleles(existing ? existing : ""), site_index);
    else
      return fail("gmx_infer_extract_debug: unknown op");
    std::string js = "[";
    for (size_t i = 0; i < result.size(); ++i) {
      js += i ? ",[" : "[";
      js += jstr(result[i].seq) + ",[";
      for (size_t j = 0; j < result[i].pb.size(); ++j) js += (j ? "," : "") + std::to_string(result[i].pb[j]);
      js += "]," + std::to_string(result[i].hapg) + "," + (result[i].callable ? "true" : "false") + "]";
    }
    js += "]";
    if (out && cap > js.size()) memcpy(out, js.c_str(), js.size() + 1);
    return (int64_t)js.size();
  } catch (std::exception const &ex) {
    return fail(std::string("allele extraction: ") + ex.what(), GMX_EREF);
  }
} GMX_GUARD_INT("gmx_infer_extract_debug")

// genotype/genotyped.json (genotype.cpp:97-105; make_json.cpp, json_prg_spec.cpp). coords_path: gram_dir/prg_coords.tsv or NULL.
int gmx_infer_write_json(const gmx_infer *inf, const char *coords_path, const char *sample_id, const char *out_path) try {
  if (!inf || !out_path) return fail("gmx_infer_write_json: bad argument");
  const Genotyper &g = *inf->g;
  std::ifstream coords;
  if (coords_path) coords.open(coords_path);
  Tracker tr(coords_path && coords.good() ? &coords : nullptr);
  std::ofstream o(out_path);
  if (!o) return fail(std::string("cannot write ") + out_path);
  o << "{\"Child_Map\":{";
  if (g.h.is_nested) {
    // keys are strings in a std::map<std::string, ...>: lexicographic order of the decimal site index
    std::map<std::string, std::map<std::string, std::vector<uint32_t>>> cm;
    for (auto const &e : g.child_m)
      for (auto const &hgp : e.second) {
        auto &v = cm[std::to_string((e.first - 5) / 2)][std::to_string(hgp.first)];
        for (uint32_t c : hgp.second) v.push_back((c - 5) / 2);
      }
    bool first = true;
    for (auto const &e : cm) {
      o << (first ? "" : ",") << jstr(e.first) << ":{";
      first = false;
      bool f2 = true;
      for (auto const &hgp : e.second) {
        o << (f2 ? "" : ",") << jstr(hgp.first) << ":[";
        f2 = false;
        for (size_t i = 0; i < hgp.second.size(); ++i) o << (i ? "," : "") << hgp.second[i];
        o << "]";
      }
      o << "}";
    }
  }
  o << "},\"Filters\":{\"AMBIG\":{\"Desc\":" << jstr(kAmbigDesc) << "}},\"Lvl1_Sites\":[";
  if (!g.h.is_nested)
    o << "\"all\"";
  else {
    bool first = true;
    for (size_t i = 0; i < g.recs.size(); ++i)
      if (g.h.sites[i].parent_site == 0) {
        o << (first ? "" : ",") << i;
        first = false;
      }
  }
  o << "],\"Model\":\"LevelGenotyping\",\"Samples\":[{\"Desc\":\"made by gramtools genotype\",\"Name\":" << jstr(sample_id ? sample_id : "")
    << "}],\"Site_Fields\":{\"ALS\":{\"Desc\":\"Alleles at this site\"},\"COV\":{\"Desc\":\"Read coverage on each allele\"},"
       "\"DP\":{\"Desc\":\"Total read depth on variant site\"},\"FT\":{\"Desc\":\"Filters failed in a sample\"},"
       "\"GT\":{\"Desc\":\"Genotype\"},\"GT_CONF\":{\"Desc\":"
    << jstr(kGtConfDesc) << "},\"GT_CONF_PERCENTILE\":{\"Desc\":" << jstr(kGcpDesc)
    << "},\"HAPG\":{\"Desc\":\"Sample haplogroups of genotyped alleles\"},\"POS\":{\"Desc\":\"Position on reference or pseudo-reference\"},"
       "\"SEG\":{\"Desc\":\"Segment ID\"}},\"Sites\":[";
  {  // the records formatted side by side (each range with a tracker of its own: positions ascend within a range), written in order
    const size_t n = g.recs.size();
    std::vector<std::string> part(par_parts(n, 2048));
    {  // ONE tracker over all sites first, as the reference's single tracker sees them: a position queried backwards is an error
       // wherever it stands — a range's own tracker below would not notice it at the range's first site
      Tracker serial = tr;
      for (size_t i = 0; i < n; ++i) (void)serial.id_of(g.recs[i]->pos);
    }
    par_ranges(n, 2048, [&](size_t b, size_t e, size_t p) {
      Tracker t2 = tr;
      std::string &out = part[p];
      for (size_t i = b; i < e; ++i) {
        const Site &s = *g.recs[i];
        const std::string seg = t2.id_of(s.pos);
        if (i) out += ',';
        out += site_json(s, &seg, t2.relative(s.pos) + 1);
      }
    });
    for (auto const &ps : part) o << ps;
  }
  o << "]}" << std::endl;
  o.close();
  return o.good() || !o.fail() ? GMX_OK : fail(std::string("error writing ") + out_path);
} GMX_GUARD_INT("gmx_infer_write_json")

// genotype/genotyped.vcf.gz (make_vcf.cpp:8-149): level-1 sites only, one sample
int gmx_infer_write_vcf(const gmx_infer *inf, const char *coords_path, const char *sample_id, const char *out_path) try {
  if (!inf || !out_path) return fail("gmx_infer_write_vcf: bad argument");
  const Genotyper &g = *inf->g;
  std::ifstream coords;
  if (coords_path) coords.open(coords_path);
  Tracker tr(coords_path && coords.good() ? &coords : nullptr);
  Bgzf z(out_path);
  if (!z.ok()) return fail(std::string("cannot write ") + out_path);
  std::ostringstream hd;
  hd << "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
  for (auto const &sg : tr.segs) hd << "##contig=<ID=" << sg.id << ",length=" << sg.size << ",Source=\"gramtools\">\n";
  hd << "##source=gramtools\n##Model=LevelGenotyping\n"
     << "##FORMAT=<ID=GT_CONF,Number=1,Type=Float,Description=\"" << kGtConfDesc << "\",Source=\"gramtools\">\n"
     << "##FORMAT=<ID=GT_CONF_PERCENTILE,Number=1,Type=Float,Description=\"" << kGcpDesc << "\",Source=\"gramtools\">\n"
     << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\",Source=\"gramtools\">\n"
     << "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Total read depth on variant site\",Source=\"gramtools\">\n"
     << "##FORMAT=<ID=COV,Number=R,Type=Float,Description=\"Read coverage on each allele\",Source=\"gramtools\">\n"
     << "##FORMAT=<ID=FT,Number=1,Type=String,Description=\"Filters failed in a sample\",Source=\"gramtools\">\n"
     << "##FILTER=<ID=AMBIG,Description=\"" << kAmbigDesc << "\",Source=\"gramtools\">\n"
     << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" << (sample_id ? sample_id : "sample") << "\n";
  z.write(hd.str());
  const size_t n_recs = g.recs.size();
  std::vector<std::string> part(par_parts(n_recs, 2048));
  {  // (the serial check of the order of positions: as in the jVCF writer above)
    Tracker serial = tr;
    for (size_t i = 0; i < n_recs; ++i)
      if (g.h.sites[i].parent_site == 0) (void)serial.id_of(g.recs[i]->pos);
  }
  par_ranges(n_recs, 2048, [&](size_t rb, size_t re, size_t pp) {
  Tracker t2 = tr;  // (a tracker per range: positions ascend within it)
  std::string &acc = part[pp];
  for (size_t i = rb; i < re; ++i) {
    if (g.h.sites[i].parent_site != 0) continue;  // next_valid_idx: sites nested in no other
    const Site &s = *g.recs[i];
    std::ostringstream r;
    const std::string chrom = t2.id_of(s.pos);
    r << chrom << "\t" << t2.relative(s.pos) + 1 << "\t.\t";
    // an empty allele would be an invalid VCF field; htslib writes what it is given, so do we
    r << (s.alleles.empty() ? std::string(".") : s.alleles[0].seq) << "\t";
    if (s.alleles.size() < 2)
      r << ".";
    else
      for (size_t a = 1; a < s.alleles.size(); ++a) r << (a > 1 ? "," : "") << s.alleles[a].seq;
    const bool has_cov = !s.covs.empty();
    r << "\t.\t.\t.\tGT:DP" << (has_cov ? ":COV" : "") << ":FT:GT_CONF:GT_CONF_PERCENTILE\t";
    if (s.is_null())
      r << ".";
    else
      for (size_t k = 0; k < s.genotype.size(); ++k) r << (k ? "/" : "") << s.genotype[k];
    r << ":" << s.total_cov;
    if (has_cov) {
      r << ":";
      for (size_t k = 0; k < s.covs.size(); ++k) r << (k ? "," : "") << fmt_g(s.covs[k]);
    }
    r << ":";
    if (s.filters.empty())
      r << "PASS";
    else  // the reference hands htslib only the first string of the vector (make_vcf.cpp:124-134)
      r << s.filters[0] << (s.filters.size() > 1 ? "," : "");
    r << ":" << fmt_g(s.gt_conf) << ":" << fmt_g(s.gt_conf_percentile) << "\n";
    acc += r.str();
  }
  });
  for (auto const &ps : part) z.write(ps);
  z.close();
  return GMX_OK;
} GMX_GUARD_INT("gmx_infer_write_vcf")

// genotype/personalised_reference.fasta (personalised_reference.cpp:8-151; genotype.cpp:16-21 dedups by sequence)
int gmx_infer_write_fasta(const gmx_infer *inf, const char *coords_path, const char *description, const char *out_path) try {
  if (!inf || !out_path) return fail("gmx_infer_write_fasta: bad argument");
  const Genotyper &g = *inf->g;
  const gmx::HostIndex &h = g.h;
  std::ifstream coords;
  if (coords_path) coords.open(coords_path);
  Tracker tr(coords_path && coords.good() ? &coords : nullptr);
  size_t ploidy = 1;  // get_ploidy, :27-38
  for (auto const &s : g.recs)
    if (!s->is_null()) {
      ploidy = s->genotype.size();
      break;
    }
  struct Fa {
    std::string id, seq;
  };
  std::vector<Fa> refs(tr.segs.size() * ploidy);
  size_t offset = 0;
  auto add_ids = [&](const std::string &id) {
    if (ploidy == 1)
      refs.at(offset).id = id;
    else
      for (size_t i = 0; i < ploidy; ++i) refs.at(i + offset).id = id + "_" + std::to_string(i + 1);
  };
  auto switch_segment = [&]() {  // :54-61
    if (tr.edge() != tr.global_edge()) {
      const std::string nid = tr.id_of(tr.edge() + 1);
      offset += ploidy;
      add_ids(nid);
    }
    return tr.edge();
  };
  try {
    size_t cur_edge = tr.edge();
    add_ids(tr.id_of(cur_edge));
    // the graph's root is the node without predecessor: node 0 of the flat graph (empty, then the first sequence / site)
    uint32_t cur = 0;
    size_t ref_pos = 0;  // coverage_Node::pos of `cur`: first-allele coordinates (coverage_graph.cpp:97-254)
    bool at_root = true;
    while (h.nodes[cur].n_edges > 0) {
      if (g.bubble_start(cur)) {
        const size_t si = (h.nodes[cur].site - 5) / 2;
        const Site &site = *g.recs.at(si);
        Gt gts = site.is_null() ? Gt(ploidy, 0) : site.genotype;
        if (gts.size() != ploidy) throw std::runtime_error("The sites do not all have the same GT cardinality (ploidy)");
        for (size_t i = 0; i < ploidy; ++i) refs.at(i + offset).seq += site.alleles.at(gts.at(i)).seq;
        // the site's end node sits at start + length of its first allele
        {  // (the LENGTH of extract_ref_allele's sequence: the first-edge path from the first allele to the site's end)
          size_t ref_len = 0;
          for (uint32_t n = h.edges[h.nodes[cur].edge_begin]; n != site.end_node; n = g.first_edge(n)) ref_len += h.nodes[n].seq_len;
          ref_pos = site.pos + ref_len;
        }
        cur = site.end_node;
        if (cur_edge == ref_pos - 1) cur_edge = switch_segment();
      }
      if (h.nodes[cur].seq_len) {
        const std::string seq = g.node_seq(cur);
        size_t cur_pos = ref_pos, end_pos = ref_pos + seq.size() - 1;
        const size_t node_pos = ref_pos;
        while (cur_pos <= end_pos) {
          if (cur_edge <= end_pos) {
            const std::string part = seq.substr(cur_pos - node_pos, cur_edge - cur_pos + 1);
            for (size_t i = 0; i < ploidy; ++i) refs.at(i + offset).seq += part;
            cur_pos = cur_edge + 1;
            cur_edge = switch_segment();
          } else {
            const std::string part = seq.substr(cur_pos - node_pos);
            for (size_t i = 0; i < ploidy; ++i) refs.at(i + offset).seq += part;
            cur_pos = end_pos + 1;
          }
        }
        ref_pos += seq.size();
      }
      (void)at_root;
      at_root = false;
      cur = h.nodes[cur].edge0;
    }
  } catch (std::exception const &ex) {
    return fail(std::string("personalised reference: ") + ex.what(), GMX_EREF);
  }
  std::map<std::string, std::string> dedup;  // std::set<Fasta> ordered by sequence: first of equal sequences stays
  for (auto const &f : refs) dedup.insert({f.seq, f.id});
  std::ofstream o(out_path);
  if (!o) return fail(std::string("cannot write ") + out_path);
  const std::string desc = description ? description : "";
  for (auto const &e : dedup) {
    o << '>' << e.second << " " << desc;
    if (desc.empty() || desc.back() != '\n') o << std::endl;
    const char *p = e.first.c_str();
    size_t remaining = e.first.size();
    while (remaining > 60) {
      o.write(p, 60);
      p += 60;
      remaining -= 60;
      o << std::endl;
    }
    o.write(p, (std::streamsize)remaining);
    o << std::endl;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_infer_write_fasta")

}  // extern "C"
