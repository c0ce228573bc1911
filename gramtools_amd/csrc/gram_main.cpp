// gram_main.cpp — the `gram` executable: drop-in for the quasimap phase of the reference's backend CLI.
//
// Process boundary reproduced: the Python front-end runs
//   gram genotype --gram_dir D --reads F1 [F2 ...] --sample_id S --ploidy {haploid,diploid} --kmer_size K
//                 --genotype_dir G --max_threads T [--seed U32] [--debug]
// (gramtools/commands/genotype/genotype.py:71-93; flags of libgramtools/src/genotype/parameters.cpp:54-72;
// two-stage parse and exit codes of libgramtools/src/main.cpp:28-100), reads D/prg and must find
//   G/coverage/allele_sum_coverage, G/coverage/allele_base_coverage.json,
//   G/coverage/grouped_allele_counts_coverage.json and G/read_stats.json (parameters.cpp:94-105).
// Host code here only parses, feeds and writes; mapping is done by the HIP engine through the C ABI (gmx.h).
// The infer stage (genotyped.json / .vcf.gz / personalised reference) is outside this engine's scope.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <condition_variable>
#include <exception>
#include <functional>
#include <new>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_gzsource.h"

namespace {

const char *kGlobalHelp =
    "Gramtools! Global options:\n"
    "  --command arg         command to execute: {build, genotype, simulate}\n"
    "  --subargs arg         arguments to command\n"
    "  --help                Produce this help message\n"
    "  --debug               Turn on debug output\n";

const char *kGenotypeHelp =
    "genotype options:\n"
    "  --gram_dir arg              gramtools directory\n"
    "  --reads arg                 file containing reads (FASTA or FASTQ)\n"
    "  --sample_id arg\n"
    "  --ploidy arg                expected ploidy of the sample. Choices: {haploid, diploid}\n"
    "  --kmer_size arg             kmer size that got used in build step\n"
    "  --genotype_dir arg          output directory\n"
    "  --max_threads arg (=1)      maximum number of threads used\n"
    "  --seed arg                  seed for pseudo-random selection of multi-mapping reads. a random seed is\n"
    "                              generated if this option is not used.\n"
    "  --device arg                HIP device ordinal (engine extension; neither --device nor --devices: every visible GPU\n"
    "                              when the reads files are large, else device 0)\n"
    "  --devices arg               several GPUs, e.g. 0-7 or 0,2,5: reads sharded, coverage summed (engine extension)\n"
    "  --rng_compat arg (=gcc11)   uniform_int_distribution flavour of the reference build to reproduce:\n"
    "                              gcc11 (libstdc++ >= 11) or gcc10 (libstdc++ <= 10) (engine extension)\n"
    "  --samples_list arg          many samples in one call (engine extension): a file with one line per sample,\n"
    "                              tab-separated: sample_id, genotype_dir, reads file(s). Replaces --reads / --sample_id /\n"
    "                              --genotype_dir. The index is loaded and uploaded once; every sample's files are those of\n"
    "                              a call of its own with the same --seed\n";

[[noreturn]] void die(const std::string &msg, int code = 1) {
  std::cout << msg << std::endl;
  exit(code);
}

void mkdirs(const std::string &path) {
  std::string cur;
  for (size_t i = 0; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty()) mkdir(cur.c_str(), 0777);
    }
    if (i < path.size()) cur += path[i];
  }
  struct stat st;
  if (stat(path.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) die("gram: cannot create directory " + path);
}

// every output file is checked after its last write: a full disk must not leave a truncated file behind an exit code 0
void close_checked(std::ofstream &o, const std::string &path) {
  o.flush();
  if (!o.good()) die("gram: error writing " + path);
  o.close();
}

std::string join(const std::string &dir, const std::string &name) {
  if (dir.empty()) return name;
  return dir.back() == '/' ? dir + name : dir + "/" + name;
}

// ---- sequence file reader: FASTQ / FASTA / one-sequence-per-line, optionally gzipped -------------------
// (the reference uses the third-party seq_file reader, include/sequence_read/seq_file.h; SAM/BAM/CRAM need
// htslib and are not supported here)
struct SeqRecord {
  std::string seq, qual;
};
class SeqReader {
 public:
  explicit SeqReader(const std::string &path) : gz_(gzopen(path.c_str(), "rb")) {
    if (!gz_) die("Cannot open reads file: " + path);
    gzbuffer(gz_, 1 << 20);
    have_line_ = next_line(line_);
  }
  ~SeqReader() {
    if (gz_) gzclose(gz_);
  }
  bool next(SeqRecord &r) {
    r.seq.clear();
    r.qual.clear();
    while (have_line_ && line_.empty()) have_line_ = next_line(line_);
    if (!have_line_) return false;
    if (line_[0] == '@') {  // FASTQ (multi-line tolerant)
      std::string l;
      while ((have_line_ = next_line(l)) && (l.empty() || l[0] != '+')) r.seq += l;
      if (!have_line_) return !r.seq.empty();
      while ((have_line_ = next_line(l))) {
        r.qual += l;
        if (r.qual.size() >= r.seq.size()) break;
      }
      have_line_ = next_line(line_);
      return true;
    }
    if (line_[0] == '>') {  // FASTA
      std::string l;
      while ((have_line_ = next_line(l)) && (l.empty() || l[0] != '>')) r.seq += l;
      line_ = l;
      return true;
    }
    r.seq = line_;  // plain
    have_line_ = next_line(line_);
    return true;
  }

 private:
  bool next_line(std::string &out) {
    out.clear();
    char buf[1 << 16];
    bool any = false;
    while (gzgets(gz_, buf, sizeof(buf))) {
      any = true;
      size_t n = strlen(buf);
      bool eol = n && buf[n - 1] == '\n';
      if (eol) --n;
      if (n && buf[n - 1] == '\r') --n;
      out.append(buf, n);
      if (eol) return true;
    }
    if (!gzeof(gz_)) {  // gzgets stopped before the end of the file: a damaged or truncated gzip stream
      int err = 0;
      const char *msg = gzerror(gz_, &err);
      if (err != Z_OK && err != Z_STREAM_END) die(std::string("gram: error reading the reads file: ") + (msg ? msg : "zlib error"));
    }
    return any;
  }
  gzFile gz_;
  std::string line_;
  bool have_line_ = false;
};

// encode_dna_bases (common/utils.cpp:73-92): A,C,G,T -> 1..4; anything else => the whole read is dropped
bool encode_read(const std::string &s, std::vector<uint8_t> &out) {
  size_t at = out.size();
  for (char c : s) {
    uint8_t v;
    switch (c) {
      case 'A': case 'a': v = 1; break;
      case 'C': case 'c': v = 2; break;
      case 'G': case 'g': v = 3; break;
      case 'T': case 't': v = 4; break;
      default: out.resize(at); return false;
    }
    out.push_back(v);
  }
  return true;
}

// ---- fast path: uncompressed four-line FASTQ, parsed by all host threads -------------------------------------
// The kernels map ~800 M reads/s; one thread parsing FASTQ text feeds ~1 M reads/s. A plain FASTQ file whose
// records are exactly four lines is memory-mapped and split into byte ranges; every thread finds the first record
// start in its range (a line starting with '@' whose second next line starts with '+': a quality line starting
// with '@' is followed by a header and a sequence line, never by '+') and encodes the records that start in it.
// Anything else (gzip, FASTA, multi-line records, blank lines) returns false and takes the sequential reader.
// Output = exactly what the sequential loop produces: bases 1..4 back to back, offsets, unencodable reads empty.
// Host buffers the engine can DMA from directly (gmx_host_alloc: page-locked memory): an upload from pageable memory is
// staged by the runtime through small pinned chunks at a fraction of the PCIe rate. Contents are not kept on growth.
template <class T>
struct HostBuf {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  HostBuf() = default;
  HostBuf(const HostBuf &) = delete;
  HostBuf &operator=(const HostBuf &) = delete;
  ~HostBuf() {
    if (p) gmx_host_free(p);
  }
  void resize(size_t count) {
    if (count > cap) {
      if (p) gmx_host_free(p);
      cap = count + count / 8 + 64;
      p = static_cast<T *>(gmx_host_alloc(cap * sizeof(T)));
      if (!p) die("gram: out of memory");
    }
    n = count;
  }
  void clear() { n = 0; }
  bool empty() const { return n == 0; }
  size_t size() const { return n; }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

// One parsed block of a reads file in the form the engine uploads as it is (gmx.h: gmx_map_reads_packed_host): bit
// planes of the base codes, one uint64 per 32 bases; reads of one length back to back (uniform_len != 0, no offsets),
// else read r at pair (offsets[r] >> 5) + r. A read with a non-ACGT symbol keeps its place and its seed and is flagged
// in `skip` (encode_dna_bases, common/utils.cpp:73-92: the whole read is dropped; quasimap.cpp:109-113 counts it).
struct ParsedReads {
  HostBuf<uint64_t> planes;
  HostBuf<uint64_t> offsets;  // n + 1 base offsets (valid when uniform_len == 0)
  HostBuf<uint8_t> skip;      // n
  HostBuf<uint32_t> seeds;    // filled by the consumer
  uint32_t uniform_len = 0;
  size_t n_reads = 0;
  uint64_t n_bases = 0;
  bool any_skip = false;
  void reset() {
    n_reads = 0;
    n_bases = 0;
    uniform_len = 0;
    any_skip = false;
  }
  uint64_t pair_of(size_t r) const {
    return uniform_len ? (uint64_t)r * ((uniform_len + 31u) / 32u) : ((offsets[r] >> 5) - (offsets[0] >> 5)) + r;
  }
  uint32_t len_of(size_t r) const { return uniform_len ? uniform_len : (uint32_t)(offsets[r + 1] - offsets[r]); }
};

// encode_dna_bases (common/utils.cpp:73-92) as a table: A,C,G,T (either case) -> 1..4, anything else 0
struct BaseTable {
  uint8_t v[256];
  BaseTable() {
    memset(v, 0, sizeof(v));
    v[(unsigned char)'A'] = v[(unsigned char)'a'] = 1;
    v[(unsigned char)'C'] = v[(unsigned char)'c'] = 2;
    v[(unsigned char)'G'] = v[(unsigned char)'g'] = 3;
    v[(unsigned char)'T'] = v[(unsigned char)'t'] = 4;
  }
};
static const BaseTable kBaseTable;

// n ASCII bases -> ceil(n / 32) plane pairs (low word = bit 0 of the codes A,C,G,T = 0..3, high word = bit 1; base j of a
// pair at bit j). Returns false when a symbol is not one of ACGTacgt. From the letters' own bits: bit 2 of the ASCII
// code is the high bit of the base code, bit 1 XOR bit 2 the low one ('A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54).
static bool pack_ascii_scalar(const unsigned char *src, size_t n, uint64_t *out) {
  bool ok = true;
  for (size_t i = 0; i < n; i += 32) {
    uint32_t lo = 0, hi = 0;
    const size_t m = std::min<size_t>(32, n - i);
    for (size_t j = 0; j < m; ++j) {
      const uint8_t v = kBaseTable.v[src[i + j]];
      ok = ok && v != 0;
      const uint32_t c = (uint32_t)(v - 1u) & 3u;
      lo |= (c & 1u) << j;
      hi |= (c >> 1) << j;
    }
    out[i >> 5] = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  return ok;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool pack_ascii_avx2(const unsigned char *src, size_t n, uint64_t *out) {
  const __m256i upper = _mm256_set1_epi8((char)0xDF);
  const __m256i cA = _mm256_set1_epi8(0x41), cC = _mm256_set1_epi8(0x43), cG = _mm256_set1_epi8(0x47), cT = _mm256_set1_epi8(0x54);
  uint32_t all_ok = 0xFFFFFFFFu;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
    const __m256i u = _mm256_and_si256(v, upper);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    all_ok &= (uint32_t)_mm256_movemask_epi8(ok);
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 5));
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 6)) ^ hi;
    out[i >> 5] = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  bool ok = all_ok == 0xFFFFFFFFu;
  if (i < n) {
    alignas(32) unsigned char tail[32];
    memset(tail, 'A', 32);
    memcpy(tail, src + i, n - i);
    uint64_t w;
    ok = pack_ascii_avx2(tail, 32, &w) && ok;
    const uint32_t keep = (uint32_t)((1ull << (n - i)) - 1ull);
    out[i >> 5] = (uint64_t)((uint32_t)w & keep) | ((uint64_t)((uint32_t)(w >> 32) & keep) << 32);
  }
  return ok;
}
static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
#else
static const bool kHaveAvx2 = false;
static bool pack_ascii_avx2(const unsigned char *src, size_t n, uint64_t *out) { return pack_ascii_scalar(src, n, out); }
#endif
static const bool kUseAvx2 = kHaveAvx2 && !getenv("GMX_NO_AVX2");  // (GMX_NO_AVX2=1: the table path, for the tests)
static inline bool pack_ascii(const unsigned char *src, size_t n, uint64_t *out) {
  return kUseAvx2 ? pack_ascii_avx2(src, n, out) : pack_ascii_scalar(src, n, out);
}

// fn(t) for t = 0 .. T-1 on a pool of waiting threads: a 96 MB block goes through three such phases, and starting 64
// threads for each of them cost more than the parsing (measured: 88 ms of 160 for 4 M reads)
class WorkerPool {
 public:
  static WorkerPool &get() {
    static WorkerPool p;
    return p;
  }
  void run(unsigned T, const std::function<void(unsigned)> &fn) {
    if (T <= 1) {
      fn(0u);
      return;
    }
    // one caller at a time (round 6: the plain-text feed's reader thread uses the pool beside the main thread)
    std::lock_guard<std::mutex> one_caller(callers_);
    std::unique_lock<std::mutex> lk(m_);
    while (threads_.size() < T - 1) {
      const unsigned id = (unsigned)threads_.size() + 1;
      threads_.emplace_back([this, id]() { loop(id); });
    }
    fn_ = &fn;
    width_ = T;
    pending_ = T - 1;
    ++generation_;
    lk.unlock();
    cv_.notify_all();
    fn(0u);
    lk.lock();
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }

 private:
  void loop(unsigned id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)> *fn;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        if (id >= width_) continue;
        fn = fn_;
      }
      (*fn)(id);
      {
        std::lock_guard<std::mutex> lk(m_);
        --pending_;
      }
      done_.notify_one();
    }
  }
  std::mutex m_, callers_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  const std::function<void(unsigned)> *fn_ = nullptr;
  unsigned width_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};
template <class F>
void parallel_for(unsigned T, F fn) {
  const std::function<void(unsigned)> f = fn;
  WorkerPool::get().run(T, f);
}
// where the feed's wall time goes (printed with the timer report): reading, parsing, and the consumer's engine calls
struct FeedTimes {
  double read_s = 0, parse_s = 0, map_s = 0, wait_slot_s = 0, scan_s = 0, pack_s = 0, seeds_s = 0;
};
static FeedTimes g_feed;
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// GMX_PHASE_TRACE=1 in the environment: where a whole `gram genotype` call spends its wall time, on stderr — milliseconds
// since main() was entered (the dynamic loader's part of the call comes before that; tools/cli_phases.sh adds it).
static const double g_phase_t0 = now_s();
static inline void phase(const char *what) {
  static const bool on = getenv("GMX_PHASE_TRACE") != nullptr;
  if (on) fprintf(stderr, "[phase %9.2f ms] %s\n", (now_s() - g_phase_t0) * 1e3, what);
}
// Parses the complete four-line records of d[0, size). When `final` is false a record that is not complete within
// the buffer ends the parse (`consumed` = its start), so that a stream can be parsed block by block.
// Returns false on anything that is not plain four-line FASTQ.
// Two parallel phases: every thread finds the records that start in its byte range and packs their letters at once
// (the text is read once, while it is in cache) into planes of its own; then, with the reads' places known (prefix sums;
// one length for all or not), every thread copies its planes (40 bytes per 150-base read) into the block's page-locked output.
// `fill` (plain files): the bytes d[fill->have, size) are not there yet — thread t of the scan first reads its slice of them
// from the file (pread, straight from the page cache) and then parses its range of the buffer, waiting for a neighbour's
// slice only where a record of its range reaches into it: the file is read and parsed in ONE pass over memory, by the same
// core, instead of a read phase and a parse phase with a barrier between them.
struct BlockFill {
  int fd = -1;
  size_t have = 0;      // bytes of d already valid (the tail of the block before)
  size_t file_at = 0;   // file offset of d[have]
  std::atomic<int> *loaded = nullptr;  // per slice: 0 = being read, 1 = there, -1 = read error
  bool io_error = false;
  int populate = 0;     // memory-mapped file (fd < 0): 1 = every thread maps its range's pages in one call first (MADV_POPULATE_READ)
};
bool parse_fastq_buffer(const char *d, size_t size, bool final, int threads, ParsedReads &out, size_t &consumed,
                        BlockFill *fill = nullptr) {
  out.reset();
  consumed = 0;
  if (size == 0) return true;
  if (fill && fill->fd >= 0 && fill->have == 0 && pread(fill->fd, const_cast<char *>(d), 1, (off_t)fill->file_at) != 1) return false;  // (d[0] ahead of the slices)
  if (d[0] != '@') return false;
  const unsigned T = (unsigned)std::max(1, std::min(threads, 128));
  struct Part {
    std::vector<uint64_t> planes;  // ceil(len / 32) pairs per read, back to back
    std::vector<uint32_t> lens;
    std::vector<uint8_t> skip;
    bool bad = false;
    size_t first = 0, stop = 0;  // start of the first record this range parsed, end of its last one
    uint64_t bases = 0;
    uint32_t len0 = 0;
    bool one_len = true;
    bool any_skip = false;
  };
  static std::vector<Part> parts;  // (kept between blocks: the vectors keep their capacity and their touched pages)
  parts.resize(T);
  const bool filling = fill && fill->fd >= 0;  // (a memory-mapped block is all there)
  const size_t fill_have = filling ? fill->have : size, fill_want = size - fill_have;
  auto slice_at = [&](unsigned j) { return fill_have + fill_want * j / T; };  // slice j = d[slice_at(j), slice_at(j + 1))
  std::atomic<bool> io_error{false};
  auto scan = [&](unsigned t) {
    Part &p = parts[t];
    // --- this thread's slice of the file ---
    if (fill && !filling && fill->populate) {
#ifdef MADV_POPULATE_READ
      const uintptr_t lo = (reinterpret_cast<uintptr_t>(d) + size * t / T) & ~(uintptr_t)4095,
                      hi = (reinterpret_cast<uintptr_t>(d) + size * (t + 1) / T + 4095) & ~(uintptr_t)4095;
      if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_READ);
#endif
    }
    if (filling) {
      size_t lo = slice_at(t), hi = slice_at(t + 1);
      bool ok = true;
      while (lo < hi && ok) {
        const ssize_t got = pread(fill->fd, const_cast<char *>(d) + lo, hi - lo, (off_t)(fill->file_at + (lo - fill_have)));
        if (got <= 0) ok = false; else lo += (size_t)got;
      }
      if (!ok) io_error.store(true);
      fill->loaded[t].store(ok ? 1 : -1, std::memory_order_release);
    }
    // --- bytes of the buffer this thread may look at: [.., avail), grown slice by slice as its records need them ---
    size_t avail = filling ? fill_have : size;
    unsigned next_slice = 0;
    bool aborted = false;
    auto grow = [&]() {  // waits for the next slice; false when there is none (or its read failed)
      if (next_slice >= T) return false;
      int v;
      while ((v = fill->loaded[next_slice].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (v < 0) {
        aborted = true;
        return false;
      }
      ++next_slice;
      avail = slice_at(next_slice);
      return true;
    };
    auto ensure = [&](size_t end) {  // d[0, min(end, size)) readable
      while (avail < end && avail < size)
        if (!grow()) break;
    };
    auto line_end = [&](size_t at) {  // index of the '\n' ending the line at `at`, or size
      for (size_t from = at;;) {
        if (from >= size) return size;
        ensure(from + 1);
        if (aborted) return size;
        const void *nl = memchr(d + from, '\n', avail - from);
        if (nl) return (size_t)((const char *)nl - d);
        if (avail >= size) return size;
        from = avail;
      }
    };
    p.planes.clear();
    p.lens.clear();
    p.skip.clear();
    p.bad = false;
    p.bases = 0;
    p.len0 = 0;
    p.one_len = true;
    p.any_skip = false;
    size_t lo = size * t / T, hi = size * (t + 1) / T;
    size_t at = lo;
    if (t > 0) {  // first record start at or after lo (a record starting exactly at lo belongs to this range)
      at = line_end(lo - 1) + 1;
      for (int tries = 0; at < size; ++tries) {
        size_t e1 = line_end(at), e2 = line_end(e1 + 1);
        if (e2 >= size) {  // fewer than three lines left: nothing starts here that an earlier range does not own
          at = size;
          break;
        }
        ensure(e2 + 2);
        if (d[at] == '@' && d[e2 + 1 < size ? e2 + 1 : e2] == '+' && e2 + 1 < size) break;
        if (tries == 5) {
          p.bad = true;
          break;
        }
        at = e1 + 1;
      }
    }
    p.first = at;
    if (p.planes.capacity() < (hi - lo) / 6 + 64) p.planes.reserve((hi - lo) / 6 + 64);
    while (at < hi && at < size && !p.bad) {
      const size_t e1 = line_end(at), s2 = e1 + 1, e2 = line_end(s2), s3 = e2 + 1, e3 = line_end(s3), s4 = e3 + 1,
                   e4 = line_end(s4);
      const bool complete = e3 < size && (e4 < size || (final && s4 <= size));
      if (!complete) {
        if (final) p.bad = true;  // truncated record
        break;
      }
      size_t n = e2 - s2, nq = e4 - s4;
      if (n && d[s2 + n - 1] == '\r') --n;
      if (nq && d[s4 + nq - 1] == '\r') --nq;
      if (d[at] != '@' || d[s3] != '+' || nq != n || n == 0 || n > 0x7FFFFFFFu) {  // blank / multi-line record: the sequential reader's job
        p.bad = true;
        break;
      }
      if (p.lens.empty()) p.len0 = (uint32_t)n;
      p.one_len = p.one_len && (uint32_t)n == p.len0;
      const size_t pairs = (n + 31) / 32, had = p.planes.size();
      p.planes.resize(had + pairs);
      const bool ok = pack_ascii(reinterpret_cast<const unsigned char *>(d) + s2, n, p.planes.data() + had);
      p.lens.push_back((uint32_t)n);
      p.skip.push_back(ok ? 0 : 1);
      p.any_skip = p.any_skip || !ok;
      p.bases += n;
      at = e4 < size ? e4 + 1 : size;
    }
    p.stop = at;
    if (aborted) p.bad = true;
  };
  const double t_scan = now_s();
  parallel_for(T, scan);
  g_feed.scan_s += now_s() - t_scan;
  if (fill) fill->io_error = io_error.load();
  for (unsigned t = 0; t < T; ++t)
    if (parts[t].bad) return false;
  // the ranges' records must chain into one gap-free prefix of the buffer; what follows it (an incomplete record, or
  // records no range could recognise from inside) is left for the next block, where it sits at the start
  size_t cursor = 0;
  for (unsigned t = 0; t < T; ++t) {
    if (parts[t].lens.empty()) continue;
    if (parts[t].first != cursor) return false;
    cursor = parts[t].stop;
  }
  if (final && cursor != size) return false;
  consumed = cursor;
  for (unsigned t = 0; t < T; ++t)  // drop what lies behind the prefix (cannot happen when the chain is intact)
    if (!parts[t].lens.empty() && parts[t].first >= consumed) return false;
  size_t n_reads = 0;
  uint64_t n_bases = 0;
  std::vector<size_t> r0(T);
  std::vector<uint64_t> b0(T);
  bool uniform = true;
  uint32_t len0 = 0;
  for (unsigned t = 0; t < T; ++t) {
    r0[t] = n_reads;
    b0[t] = n_bases;
    if (!parts[t].lens.empty()) {
      if (n_reads == 0) len0 = parts[t].len0;
      uniform = uniform && parts[t].one_len && parts[t].len0 == len0;
      out.any_skip = out.any_skip || parts[t].any_skip;
    }
    n_reads += parts[t].lens.size();
    n_bases += parts[t].bases;
  }
  out.n_reads = n_reads;
  out.n_bases = n_bases;
  out.uniform_len = uniform && n_reads ? len0 : 0u;
  const uint64_t ppr = (out.uniform_len + 31u) / 32u;
  const uint64_t n_pairs = out.uniform_len ? n_reads * ppr : (n_bases >> 5) + n_reads;
  out.planes.resize(n_pairs + 8);  // (+ slack: the device fetches whole 16-byte pieces)
  out.skip.resize(std::max<size_t>(n_reads, 1));
  out.offsets.resize(n_reads + 1);
  const double t_pack = now_s();
  parallel_for(T, [&](unsigned t) {
    const Part &p = parts[t];
    if (p.lens.empty()) return;
    memcpy(out.skip.data() + r0[t], p.skip.data(), p.skip.size());
    if (out.uniform_len) {
      memcpy(out.planes.data() + r0[t] * ppr, p.planes.data(), p.planes.size() * sizeof(uint64_t));
      return;
    }
    uint64_t off = b0[t];
    size_t from = 0;
    for (size_t i = 0; i < p.lens.size(); ++i) {
      const size_t r = r0[t] + i;
      const uint32_t len = p.lens[i];
      const uint64_t at = (off >> 5) + r, pairs = (len + 31u) / 32u, next = ((off + len) >> 5) + r + 1;
      memcpy(out.planes.data() + at, p.planes.data() + from, pairs * sizeof(uint64_t));
      for (uint64_t q = at + pairs; q < next; ++q) out.planes[q] = 0;  // gap pair of the offsets form
      out.offsets[r] = off;
      from += pairs;
      off += len;
    }
  });
  g_feed.pack_s += now_s() - t_pack;
  out.offsets[n_reads] = n_bases;
  for (uint64_t q = n_pairs; q < n_pairs + 8; ++q) out.planes[q] = 0;
  return true;
}

// GMX_FEED_TRACE=1 in the environment: the feed's events with their times on stderr
static double g_trace_t0 = 0;
static const bool g_trace = getenv("GMX_FEED_TRACE") != nullptr;
static inline void feed_trace(const char *what) {
  if (!g_trace) return;
  if (g_trace_t0 == 0) g_trace_t0 = now_s();
  fprintf(stderr, "[feed %8.2f ms] %s\n", (now_s() - g_trace_t0) * 1e3, what);
}

// Two parsed blocks in flight: while the consumer thread hands block i to the engine (seeds, upload, kernels), the
// caller's thread and the parser threads work on block i + 1. Blocks are consumed in file order.
struct BlockPipe {
  ParsedReads slot[2];
  int state[2] = {0, 0};  // 0 free, 1 filled
  std::mutex m;
  std::condition_variable cv;
  bool closing = false;
  std::thread consumer;
  uint64_t produced = 0;
  explicit BlockPipe(std::function<void(ParsedReads &)> sink) {
    consumer = std::thread([this, sink]() {
      for (uint64_t k = 0;; ++k) {
        ParsedReads *blk;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return state[k & 1] == 1 || (closing && k >= produced); });
          if (state[k & 1] != 1) return;
          blk = &slot[k & 1];
        }
        const double t0 = now_s();
        feed_trace("consumer: block taken");
        sink(*blk);
        feed_trace("consumer: block mapped");
        g_feed.map_s += now_s() - t0;
        {
          std::lock_guard<std::mutex> lk(m);
          state[k & 1] = 0;
        }
        cv.notify_all();
      }
    });
  }
  ParsedReads &acquire() {  // the slot the next block is parsed into (waits until the consumer is done with it)
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return state[produced & 1] == 0; });
    g_feed.wait_slot_s += now_s() - t0;
    return slot[produced & 1];
  }
  void submit() {
    {
      std::lock_guard<std::mutex> lk(m);
      state[produced & 1] = 1;
      ++produced;
    }
    cv.notify_all();
  }
  void finish() {
    {
      std::lock_guard<std::mutex> lk(m);
      closing = true;
    }
    cv.notify_all();
    if (consumer.joinable()) consumer.join();
  }
  ~BlockPipe() { finish(); }
};

// Reads per engine call. A nested PRG's batch ends with ~2 ms of a few straggler tasks whatever its size (gmx_engine.hip:
// gmx_feed_chunk), so there the feed hands over blocks of ~4 M reads instead of ~0.3 M: file blocks 12 times as large,
// three times the BGZF members per device chunk (set by run_genotype once the index is loaded).
static size_t g_block_scale = 1;
static uint64_t g_ingest_member_scale = 1;

// the file block being parsed (kept between files; `gram genotype` allocates and touches it beside the index load)
static std::unique_ptr<char[]> g_block_mem;
static size_t g_block_cap = 0;

// A whole reads file through the fast path, block by block (GMX_FASTQ_BLOCK bytes, 96 MB by default): a plain file is
// read with parallel pread calls, a gzip file is inflated by this thread; every block is parsed by all threads while the
// engine works on the block before it (BlockPipe). `sink` receives every block's reads in file order, on another thread.
// Returns false — before anything was delivered — if the file is not plain four-line FASTQ.
template <class Sink>
bool parse_fastq_file(const std::string &path, int threads, Sink sink) {
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  unsigned char magic[2] = {0, 0};
  const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  struct stat sb;
  const bool stat_ok = fstat(fd, &sb) == 0;
  // (gzip input: larger blocks — the inflated text is parsed from a buffer, block after block, and a block's fixed costs,
  //  64 thread wake-ups and the leftover's move, were two thirds of the BGZF leg's parse time at 96 MB)
  size_t kBlock = (gz ? (size_t)384 << 20 : (size_t)96 << 20) * g_block_scale;
  if (const char *eb = getenv("GMX_FASTQ_BLOCK")) kBlock = std::max<size_t>(64, (size_t)atoll(eb));  // tests: tiny blocks
  const unsigned T = (unsigned)std::max(1, std::min(threads, 128));
  feed_trace("file opened");
  struct {
    char *p;
    char *data() const { return p; }
  } buf{nullptr};
  size_t have = 0, consumed = 0;
  bool first = true;
  std::unique_ptr<gmx::GzSource> g;  // gzip input: BGZF members inflated side by side, other members through one zlib stream
  size_t file_at = 0;
  const size_t file_size = stat_ok ? (size_t)sb.st_size : 0;
  if (gz) {
    close(fd);
    fd = -1;
    try {
      g.reset(new gmx::GzSource(path, T));
    } catch (std::exception const &) {
      return false;
    }
  } else if (!stat_ok || file_size == 0) {
    close(fd);
    return false;
  }
  auto shut = [&]() {
    g.reset();
    if (fd >= 0) close(fd);
  };
  // Plain files are memory-mapped and parsed in place: one pass over the page cache's own pages (measured on the GPU
  // box, 4 M reads, 64 threads: 16 ms against 25-33 ms for pread into a block buffer + parse, whose copy triples the
  // memory traffic). GMX_FASTQ_MMAP=0: the pread path (also taken when the file cannot be mapped); =2: every thread maps
  // its range's pages in one call first (no gain measured).
  const char *mm_env = getenv("GMX_FASTQ_MMAP");
  const int mm_mode = gz ? 0 : (mm_env ? atoi(mm_env) : 1);
  const char *map = nullptr;
  size_t map_pos = 0;
  if (mm_mode) {
    void *m = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
    if (m != MAP_FAILED) map = static_cast<const char *>(m);
  }
  if (!map) {  // (not value-initialised: the pages are first touched by the parallel reads below; kept for the next file)
    if (g_block_cap < kBlock + (1u << 20)) {
      g_block_cap = kBlock + (1u << 20);
      g_block_mem.reset(new char[g_block_cap]);
    }
    buf.p = g_block_mem.get();
  }
  feed_trace("block source ready");
  std::unique_ptr<BlockPipe> pipe;  // started with the first good block
  std::thread unmapper;
  size_t unmapped = 0;
  BlockFill fill;
  bool use_fill = false;
  std::unique_ptr<std::atomic<int>[]> loaded(new std::atomic<int>[T]);
  for (;;) {
    bool final;
    const double t_read = now_s();
    if (g) {
      try {  // a damaged or truncated gzip stream must not pass for the end of the reads: GzSource throws
        have += g->read(buf.data() + have, kBlock - std::min(kBlock, have));
      } catch (std::exception const &e) {
        die(std::string("gram: ") + e.what());
      }
      final = g->at_end() || have < kBlock;
    } else if (map) {
      have = std::min(kBlock, file_size - map_pos);
      final = map_pos + have >= file_size;
      fill.fd = -1;
      fill.populate = mm_mode >= 2;
      use_fill = true;
    } else {
      // (plain file: the parser's threads read the block themselves, each its slice, and parse it at once: BlockFill)
      const size_t want = std::min(kBlock - std::min(kBlock, have), file_size - file_at);
      fill.fd = fd;
      fill.have = have;
      fill.file_at = file_at;
      for (unsigned t = 0; t < T; ++t) loaded[t].store(0, std::memory_order_relaxed);
      fill.loaded = loaded.get();
      use_fill = true;
      have += want;
      file_at += want;
      final = file_at >= file_size;
    }
    g_feed.read_s += now_s() - t_read;
    feed_trace("block read");
    ParsedReads scratch;
    ParsedReads &block = pipe ? pipe->acquire() : scratch;
    const double t_parse = now_s();
    const bool parsed = parse_fastq_buffer(map ? map + map_pos : buf.data(), have, final, threads, block, consumed, use_fill ? &fill : nullptr);
    if (use_fill && fill.io_error) die("gram: " + path + ": read error");
    g_feed.parse_s += now_s() - t_parse;
    feed_trace("block parsed");
    if (!parsed || (!final && consumed == 0)) {
      if (first) {
        if (map) munmap(const_cast<char *>(map), file_size);
        shut();
        return false;
      }
      die("gram: " + path + ": irregular FASTQ record after the first " + std::to_string(file_at >> 20) +
          " MB (multi-line or blank lines); decompress and reformat, or use a four-line FASTQ");
    }
    if (first) {  // the file is what this path covers: from here on blocks go through the pipe
      first = false;
      pipe.reset(new BlockPipe([&](ParsedReads &b) { sink(b); }));
      ParsedReads &slot0 = pipe->acquire();
      auto swap_buf = [](auto &x, auto &y) {
        std::swap(x.p, y.p);
        std::swap(x.n, y.n);
        std::swap(x.cap, y.cap);
      };
      swap_buf(slot0.planes, scratch.planes);
      swap_buf(slot0.offsets, scratch.offsets);
      swap_buf(slot0.skip, scratch.skip);
      std::swap(slot0.uniform_len, scratch.uniform_len);
      std::swap(slot0.n_reads, scratch.n_reads);
      std::swap(slot0.n_bases, scratch.n_bases);
      std::swap(slot0.any_skip, scratch.any_skip);
    }
    pipe->submit();
    if (map) {
      map_pos += consumed;
      have = 0;
      // the text behind map_pos is done with (the reads left as bit planes): its pages are unmapped beside the parse of
      // the next block, whole pages only, instead of 4 ms per GB at the end of the file
      const size_t upto = final ? file_size : (map_pos & ~(size_t)4095);
      if (upto > unmapped) {
        if (unmapper.joinable()) unmapper.join();
        const char *from = map + unmapped;
        const size_t len = upto - unmapped;
        unmapper = std::thread([from, len]() { munmap(const_cast<char *>(from), len); });
        unmapped = upto;
      }
    } else {
      memmove(buf.data(), buf.data() + consumed, have - consumed);
      have -= consumed;
    }
    if (final) break;
  }
  pipe->finish();
  feed_trace("pipe drained");
  if (unmapper.joinable()) unmapper.join();
  shut();
  return true;
}

#define GMX_CHECK(expr)                                                        \
  do {                                                                         \
    int _rc = (expr);                                                          \
    if (_rc < 0) die(std::string("gram: ") + #expr + " failed: " + gmx_last_error()); \
  } while (0)

// ---- BGZF files decoded on the GPU (include/gmx.h gmx_ingest_*, gmx_ingest.hip) ------------------------------------------
// A file that is BGZF members from its first byte to its last (bgzip, htslib, BCL Convert) is handed to the device as it
// lies in the page cache: the member table is walked here (18 bytes per member), the deflate data of a few thousand members
// at a time is copied into page-locked memory by all threads and uploaded, and HIP kernels inflate it, check every member's
// CRC-32, find the records and pack the bases — the host inflates nothing (sixteen cores manage 32-48 M reads/s of BGZF; the
// mapping kernels take 2 400 M). `on_chunk(result, slot)` is called for every chunk in file order, with the chunk's reads in
// HBM; it must enqueue what reads them and call gmx_ingest_release_after.
// Returns  0  the whole file was delivered
//          1  declined before anything was delivered (not pure BGZF, or not four-line FASTQ: the caller's other readers decide)
//          2  a chunk could not be decoded after `delivered` reads had been: the caller re-reads the file on the host, which
//             either reports the damage or — a defect of the device decoder — delivers the rest
static bool bgzf_member_at(const unsigned char *in, size_t size, size_t at, gmx_bgzf_member *m, size_t *next) {
  auto le16 = [](const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; };
  auto le32 = [](const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; };
  if (at + 18 > size) return false;
  const unsigned char *p = in + at;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return false;
  const uint32_t xlen = le16(p + 10);
  if (at + 12 + xlen > size) return false;
  uint32_t bsize = 0;
  bool found = false;
  for (uint32_t x = 0; x + 4 <= xlen;) {
    const unsigned char *f = p + 12 + x;
    const uint32_t slen = le16(f + 2);
    if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) {
      bsize = le16(f + 4) + 1;
      found = true;
    }
    x += 4 + slen;
  }
  if (!found || bsize < 12 + xlen + 8 || at + bsize > size) return false;
  m->offset = at + 12 + xlen;
  m->size = bsize - 12 - xlen - 8;
  m->crc32 = le32(p + bsize - 8);
  m->isize = le32(p + bsize - 4);
  m->reserved = 0;
  *next = at + bsize;
  return m->isize <= 65536u;
}

struct DeviceFeed {  // one per process: the ingest object and its page-locked staging, sized by the largest file seen
  gmx_ingest *ing = nullptr;
  uint64_t max_text = 0;
  int device = 0;
  HostBuf<uint8_t> stage[4];  // (BGZF uses three: chunk i + 2's bytes are staged while chunks i and i + 1 are on the device; plain text four: ingest_text_file)
  ~DeviceFeed() {
    if (ing) gmx_ingest_destroy(ing);
  }
};
static DeviceFeed g_device_feed;                             // the first (or only) engine's
static std::vector<std::unique_ptr<DeviceFeed>> g_more_feeds;  // the other engines' (several GPUs: chunks dealt round)

// the ingest object and its staging buffers ahead of the first BGZF file (called beside the index load: device memory for a
// chunk and two page-locked buffers take 30 ms to come by)
static bool hipSetDeviceForPrewarm(int) { return true; }  // (gmx_ingest_create selects the device itself)
static void device_feed_prepare(int device, uint64_t want_text, uint64_t stage_bytes, DeviceFeed *which = nullptr) {
  DeviceFeed &df = which ? *which : g_device_feed;
  if (!df.ing || df.max_text < want_text || df.device != device) {
    if (df.ing) gmx_ingest_destroy(df.ing);
    df.ing = nullptr;
    df.max_text = 0;
    if (gmx_ingest_create(device, want_text, &df.ing) != GMX_OK) return;
    df.max_text = want_text;
    df.device = device;
  }
  for (auto &st : df.stage)
    if (st.size() < stage_bytes) st.resize(stage_bytes);
}
static uint64_t device_feed_members() {
  uint64_t k = 7168 * g_ingest_member_scale;  // one round of the wavefronts an MI355X holds of gmx_inflate_kernel (28 per CU)
  if (const char *e = getenv("GMX_INGEST_MEMBERS")) k = std::max<uint64_t>(1, (uint64_t)atoll(e));
  return k;
}
static uint64_t device_feed_text_for(uint64_t file_text) {
  return std::min<uint64_t>(std::max<uint64_t>(std::min<uint64_t>(file_text, device_feed_members() * 65536ull), 1u << 16) + (1u << 16), 3ull << 30);
}

template <class OnChunk>
int ingest_bgzf_file(const std::string &path, int threads, int device, OnChunk on_chunk, uint64_t *delivered) {
  *delivered = 0;
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return 1;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size < 28) {
    close(fd);
    return 1;
  }
  const size_t size = (size_t)sb.st_size;
  void *mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
  close(fd);
  if (mp == MAP_FAILED) return 1;
  const unsigned char *in = static_cast<const unsigned char *>(mp);
  struct Unmap {
    void *p;
    size_t n;
    ~Unmap() { munmap(p, n); }
  } unmap{mp, size};
  // The member table is walked a chunk at a time, beside the device (round 5: the whole table first cost 7 ms of a 22 000-member file
  // before the first byte went up). Every byte of the file must belong to a BGZF member: a file that stops being BGZF behind chunks
  // already delivered is handed to the host reader, which drops what was mapped (return value 2).
  {  // not BGZF at its first byte: declined BEFORE the ingest is sized for it (round 6: a plain FASTQ of 1.26 GB had the ingest the
     // prewarm thread made for its text chunks thrown away and one for 470 MB chunks made here — 40 ms — just to be declined)
    gmx_bgzf_member m0;
    size_t next0;
    if (!bgzf_member_at(in, size, 0, &m0, &next0)) return 1;
  }
  DeviceFeed &df = g_device_feed;
  device_feed_prepare(device, device_feed_text_for((uint64_t)size * 6), 0);  // (as the call beside the index load sized it)
  if (!df.ing) return 1;
  gmx_ingest *ing = df.ing;
  GMX_CHECK(gmx_ingest_reset(ing));
  // (a chunk holds at most what the ingest's member table holds: a file of many tiny members must not be refused by the submit)
  const uint64_t kMembers = std::min<uint64_t>(device_feed_members(), gmx_ingest_max_members(ing));
  const uint64_t max_text = gmx_ingest_max_text(ing), max_comp = gmx_ingest_max_compressed(ing);
  struct Chunk {
    std::vector<gmx_bgzf_member> rel;  // its members, offsets from lo
    size_t lo = 0, hi = 0;             // file bytes their deflate data spans
  };
  size_t walk_at = 0;
  auto skip_empty = [&]() -> bool {  // to the next member that holds text; false: the bytes there are no BGZF member
    while (walk_at < size) {
      gmx_bgzf_member m;
      size_t next;
      if (!bgzf_member_at(in, size, walk_at, &m, &next)) return false;
      if (m.isize) break;
      walk_at = next;  // (empty members — the EOF marker — hold nothing)
    }
    return true;
  };
  // the next chunk of members: at most kMembers, and what the ingest has room for. 1: a chunk, 0: the file's end, -1: not BGZF
  auto next_chunk = [&](Chunk &c) -> int {
    c.rel.clear();
    c.lo = c.hi = 0;
    uint64_t text = 0;
    while (walk_at < size) {
      gmx_bgzf_member m;
      size_t next;
      if (!bgzf_member_at(in, size, walk_at, &m, &next)) return -1;
      if (m.isize) {
        if (c.rel.empty()) c.lo = (size_t)m.offset;
        if (c.rel.size() >= kMembers || text + m.isize > max_text || m.offset + m.size - c.lo > max_comp) break;
        text += m.isize;
        c.hi = (size_t)(m.offset + m.size);
        m.offset -= c.lo;
        c.rel.push_back(m);
      }
      walk_at = next;
    }
    if (c.rel.empty()) return walk_at < size ? -1 : 0;  // (a member the ingest has no room for: cannot happen with <= 64 KB members)
    return skip_empty() ? 1 : -1;                        // (so that walk_at == size tells a chunk it is the file's last)
  };
  const unsigned T = (unsigned)std::max(1, std::min(threads, 64));
  // Chunk i + 2's bytes are staged (page cache -> page-locked memory, 5 ms a chunk) and submitted while chunks i and i + 1 are on the
  // device (the ingest's three slots): the device never waits for the host's memcpy
  // (round 5: staged inside the submit, the copy sat between a chunk's result and the next chunk's upload — a GPU idle for 6 ms of
  // every 13). Three staging buffers: the one chunk i + 2 takes was chunk i - 1's, whose upload is long done (it has been waited for).
  Chunk ring[3];
  bool last_of_file[3] = {false, false, false};
  size_t n_staged = 0;  // chunks walked and staged so far
  bool at_end = false, bad_walk = false;
  auto stage = [&]() -> bool {  // the next chunk, into slot n_staged % 3; false: there is none
    if (at_end || bad_walk) return false;
    Chunk &c = ring[n_staged % 3];
    const int rc = next_chunk(c);
    if (rc <= 0) {
      at_end = rc == 0;
      bad_walk = rc < 0;
      return false;
    }
    last_of_file[n_staged % 3] = walk_at >= size;
    HostBuf<uint8_t> &st = df.stage[n_staged % 3];
    const size_t n = c.hi - c.lo;
    st.resize(n + 64);
    parallel_for(T, [&](unsigned t) {  // the compressed bytes, from the page cache into page-locked memory
      const size_t a = n * t / T, b = n * (t + 1) / T;
      if (b > a) memcpy(st.data() + a, in + c.lo + a, b - a);
    });
    ++n_staged;
    return true;
  };
  size_t n_submitted = 0;
  auto submit = [&]() {  // chunk n_submitted (staged)
    const size_t ci = n_submitted;
    const Chunk &c = ring[ci % 3];
    GMX_CHECK(gmx_ingest_submit_bgzf(ing, (int)(ci % 3), df.stage[ci % 3].data(), c.hi - c.lo, c.rel.data(), c.rel.size(), last_of_file[ci % 3] ? 1 : 0));
    ++n_submitted;
    feed_trace("chunk submitted to the device");
  };
  if (!skip_empty()) return 1;
  if (!stage()) return bad_walk ? 1 : 0;  // (0: no reads at all)
  submit();
  feed_trace("first chunk on its way; the member table is walked beside the device");
  if (stage()) submit();
  for (size_t ci = 0; ci < n_submitted; ++ci) {
    // chunk ci + 2 goes to the device BEFORE chunk ci is waited for (three slots: its slot is chunk ci - 1's, waited for and handed to
    // the engine): two inflate kernels are queued behind the one in flight, and a kernel's last wavefronts never have the GPU alone
    if (n_staged == n_submitted) stage();
    if (n_staged > n_submitted) submit();
    gmx_ingest_result res;
    GMX_CHECK(gmx_ingest_wait(ing, (int)(ci % 3), &res));
    feed_trace("chunk decoded");
    if (const char *tf = getenv("GMX_INGEST_TEST_FAIL_CHUNK"))  // test hook: the device decoder "gives up" on this chunk (tests/test_ingest.py)
      if ((size_t)atoll(tf) == ci) res.status |= GMX_INGEST_BAD_MEMBER;
    if (res.status) {
      for (size_t cj = ci + 1; cj < n_submitted; ++cj) {  // (the chunks behind are in flight: let them finish before the slots are reused)
        gmx_ingest_result drop;
        GMX_CHECK(gmx_ingest_wait(ing, (int)(cj % 3), &drop));
      }
      const bool decoder = (res.status & (GMX_INGEST_BAD_MEMBER | GMX_INGEST_BAD_CRC | GMX_INGEST_TOO_MANY_LINES)) != 0;
      if (!decoder) {  // the text itself is not four-line FASTQ: the host's fast path would say the same
        if (*delivered == 0) return 1;
        die("gram: " + path + ": irregular FASTQ record after the first " + std::to_string(*delivered) +
            " reads (multi-line or blank lines); decompress and reformat, or use a four-line FASTQ");
      }
      // a member the kernels would not decode, or lines of a few bytes (more records than the ingest has room for): the host reader's
      return *delivered == 0 && !(res.status & (GMX_INGEST_BAD_MEMBER | GMX_INGEST_BAD_CRC)) ? 1 : 2;
    }
    on_chunk(res, (int)(ci % 3));
    *delivered += res.n_reads;
  }
  if (bad_walk) return *delivered == 0 ? 1 : 2;  // bytes that are no BGZF member behind the chunks delivered: the host reader's
  return 0;
}

// The same over SEVERAL engines (`--devices`): one ingest per engine, the file's chunks dealt round. Every chunk
// is uploaded and inflated ahead (two per device in flight); the chunks are scanned in file order, each with the cut record of
// the chunk before — which lies on another device — handed over through the host (a few hundred bytes). on_chunk(result, k, slot):
// engine k's ingest holds the chunk's reads. Return values as ingest_bgzf_file.
template <class OnChunk>
int ingest_bgzf_file_dealt(const std::string &path, int threads, const std::vector<int> &devs, OnChunk on_chunk, uint64_t *delivered) {
  *delivered = 0;
  const size_t N = devs.size();
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return 1;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size < 28) {
    close(fd);
    return 1;
  }
  const size_t size = (size_t)sb.st_size;
  void *mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
  close(fd);
  if (mp == MAP_FAILED) return 1;
  const unsigned char *in = static_cast<const unsigned char *>(mp);
  struct Unmap {
    void *p;
    size_t n;
    ~Unmap() { munmap(p, n); }
  } unmap{mp, size};
  std::vector<gmx_bgzf_member> members;
  members.reserve(size / 16000 + 16);
  for (size_t at = 0; at < size;) {
    gmx_bgzf_member m;
    size_t next;
    if (!bgzf_member_at(in, size, at, &m, &next)) return 1;
    if (m.isize) members.push_back(m);
    at = next;
  }
  uint64_t kMembers = device_feed_members();
  uint64_t file_text = 0;
  for (const auto &m : members) file_text += m.isize;
  while (g_more_feeds.size() + 1 < N) g_more_feeds.emplace_back(new DeviceFeed());
  auto feed = [&](size_t k) -> DeviceFeed & { return k == 0 ? g_device_feed : *g_more_feeds[k - 1]; };
  for (size_t k = 0; k < N; ++k) {
    device_feed_prepare(devs[k], device_feed_text_for(file_text), 0, &feed(k));
    if (!feed(k).ing) return 1;
    GMX_CHECK(gmx_ingest_reset(feed(k).ing));
  }
  feed_trace("ingests of all engines ready");
  const uint64_t max_text = gmx_ingest_max_text(feed(0).ing), max_comp = gmx_ingest_max_compressed(feed(0).ing);
  kMembers = std::min<uint64_t>(kMembers, gmx_ingest_max_members(feed(0).ing));
  struct Chunk {
    size_t first, count, lo, hi;
  };
  std::vector<Chunk> chunks;
  for (size_t i = 0; i < members.size();) {
    Chunk c{i, 0, (size_t)members[i].offset, 0};
    uint64_t text = 0;
    while (i < members.size() && c.count < kMembers && text + members[i].isize <= max_text && members[i].offset + members[i].size - c.lo <= max_comp) {
      text += members[i].isize;
      c.hi = (size_t)(members[i].offset + members[i].size);
      ++c.count;
      ++i;
    }
    if (c.count == 0) return 1;
    chunks.push_back(c);
  }
  if (chunks.empty()) return 0;
  const unsigned T = (unsigned)std::max(1, std::min(threads, 64));
  std::vector<gmx_bgzf_member> rel;
  auto dev_of = [&](size_t ci) { return ci % N; };
  auto slot_of = [&](size_t ci) { return (int)((ci / N) & 1); };
  auto submit = [&](size_t ci) {  // upload + inflate; the scan follows when the chunk before has been scanned
    const Chunk &c = chunks[ci];
    HostBuf<uint8_t> &st = feed(dev_of(ci)).stage[slot_of(ci)];
    const size_t n = c.hi - c.lo;
    st.resize(n + 64);
    parallel_for(T, [&](unsigned t) {
      const size_t a = n * t / T, b = n * (t + 1) / T;
      if (b > a) memcpy(st.data() + a, in + c.lo + a, b - a);
    });
    rel.assign(members.begin() + (long)c.first, members.begin() + (long)(c.first + c.count));
    for (auto &m : rel) m.offset -= c.lo;
    GMX_CHECK(gmx_ingest_submit_bgzf_deferred(feed(dev_of(ci)).ing, slot_of(ci), st.data(), n, rel.data(), rel.size()));
    feed_trace("chunk submitted to its device (inflate)");
  };
  size_t submitted = 0;
  for (; submitted < std::min(chunks.size(), 2 * N); ++submitted) submit(submitted);
  std::vector<uint8_t> tail;
  for (size_t ci = 0; ci < chunks.size(); ++ci) {
    gmx_ingest *ing = feed(dev_of(ci)).ing;
    GMX_CHECK(gmx_ingest_scan(ing, slot_of(ci), tail.data(), tail.size(), ci + 1 == chunks.size() ? 1 : 0));
    gmx_ingest_result res;
    GMX_CHECK(gmx_ingest_wait(ing, slot_of(ci), &res));
    feed_trace("chunk scanned");
    if (const char *tf = getenv("GMX_INGEST_TEST_FAIL_CHUNK"))
      if ((size_t)atoll(tf) == ci) res.status |= GMX_INGEST_BAD_MEMBER;
    if (res.status) {
      for (size_t cj = ci + 1; cj < submitted; ++cj) {  // (chunks inflating ahead: scanned with nothing and dropped, so that their slots are free again)
        gmx_ingest_result drop;
        GMX_CHECK(gmx_ingest_scan(feed(dev_of(cj)).ing, slot_of(cj), nullptr, 0, 0));
        GMX_CHECK(gmx_ingest_wait(feed(dev_of(cj)).ing, slot_of(cj), &drop));
      }
      const bool decoder = (res.status & (GMX_INGEST_BAD_MEMBER | GMX_INGEST_BAD_CRC | GMX_INGEST_TOO_MANY_LINES)) != 0;
      if (!decoder) {
        if (*delivered == 0) return 1;
        die("gram: " + path + ": irregular FASTQ record after the first " + std::to_string(*delivered) +
            " reads (multi-line or blank lines); decompress and reformat, or use a four-line FASTQ");
      }
      return *delivered == 0 && !(res.status & (GMX_INGEST_BAD_MEMBER | GMX_INGEST_BAD_CRC)) ? 1 : 2;
    }
    tail.resize(res.tail_bytes);
    if (res.tail_bytes && gmx_ingest_fetch_tail(ing, slot_of(ci), tail.data(), tail.size()) < 0) die(std::string("gram: ") + gmx_last_error());
    on_chunk(res, dev_of(ci), slot_of(ci));
    *delivered += res.n_reads;
    if (submitted < chunks.size()) submit(submitted++);  // (its slot: chunk submitted - 2 N, waited for and handed on two rounds ago)
  }
  return 0;
}

// ---- plain (uncompressed) four-line FASTQ through the same device chain (round 6) -----------------------------------------
// The reference reads every text format through one host reader (include/sequence_read/seqread.hpp:94-180, seq_file.h:626-633,
// quasimap.cpp:65-76). Until round 6 `gram` took the device route for BGZF only and parsed a plain FASTQ on the host: 42 M
// reads/s parse + map at configs[1], 3.5 x slower than the same reads bgzipped. Now the file's bytes go up as they are: all
// threads pread their slices of a chunk from the page cache into page-locked memory (ONE pass over the text on the host, no
// parsing), the chunk is uploaded (gmx_ingest_submit_text) and gmx_nl_* / gmx_records / gmx_fq_pack find the records and pack the
// bases in HBM; a record cut by a chunk's end is carried into the next chunk on the device. Bound: the host link at ~316 B per
// 150-base read (measured 120-140 M reads/s; the link's 55 GB/s would be 175 M). Whether a file takes this route or the host parser:
// plain_fastq_on_device() below. Return values as ingest_bgzf_file.
// Which reader takes a plain FASTQ (measured on the GPU boxes, configs[1], 4 M x 150 bp = 1.26 GB in the page cache, parse + map;
// profiles/round6/cli_text_feed_by_threads.txt): the device route moves 316 B per read over the host link and is bound by it —
// 28-33 ms = 120-140 M reads/s from 2 host threads on, 67 M with ONE (one core copies 21 GB/s out of the page cache) — while the
// host parser sends 40 B per read and scales with the cores: 24 M reads/s on 1 thread, 35-45 M on 2-4, 52-82 M on 8, 75-102 M on 16,
// 140-165 M on 32, 140-215 M on 64. So: the device route below 32 host threads (`--max_threads` defaults to 1 in the reference's
// front-end: 2.8 x), the host parser from 32 threads on. GMX_DEVICE_FASTQ=1 / GMX_HOST_FASTQ=1 force either.
static bool plain_fastq_on_device(int max_threads, size_t n_engines) {
  if (getenv("GMX_HOST_FASTQ")) return false;
  if (const char *e = getenv("GMX_DEVICE_FASTQ")) return atoi(e) != 0;
  (void)n_engines;  // (several engines: the same rule — the chunks are then dealt over the engines' ingests, ingest_text_file_dealt; measured
                    //  only with several engines on ONE GPU, tools/feed_x8.py, where it has nothing to gain)
  return max_threads < 32;
}
static uint64_t device_feed_text_chunk(size_t scale = 0) {
  uint64_t c = (128ull << 20) * (scale ? scale : std::min<size_t>(g_block_scale, 4));  // (nested PRGs: larger launches, see g_block_scale; 128 MB: 29-33 ms per 1.26 GB where 64 MB chunks take 34-40)
  if (const char *e = getenv("GMX_TEXT_CHUNK")) c = std::max<uint64_t>(64, (uint64_t)atoll(e));
  return std::min<uint64_t>(c, (3ull << 30) - (1u << 20));
}
static uint64_t fstat_ok_size(const std::string &path) {
  struct stat sb;
  return stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) ? (uint64_t)sb.st_size : 0;
}
static bool looks_like_plain_fastq(const std::string &path, uint64_t *size_out) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat sb;
  unsigned char h[2] = {0, 0};
  const bool ok = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 4 && pread(fd, h, 2, 0) == 2 && h[0] == '@';
  close(fd);
  if (ok && size_out) *size_out = (uint64_t)sb.st_size;
  return ok;
}
// bytes [at, at + n) of the file into dst, every thread its slice (page cache -> page-locked memory); false: a read failed
static bool pread_parallel(int fd, uint64_t at, uint8_t *dst, size_t n, unsigned T) {
  std::atomic<bool> bad{false};
  parallel_for(T, [&](unsigned t) {
    size_t lo = n * t / T;
    const size_t hi = n * (t + 1) / T;
    while (lo < hi) {
      const ssize_t got = pread(fd, dst + lo, hi - lo, (off_t)(at + lo));
      if (got <= 0) {
        bad.store(true);
        return;
      }
      lo += (size_t)got;
    }
  });
  return !bad.load();
}
struct FdCloser {
  int fd;
  ~FdCloser() {
    if (fd >= 0) close(fd);
  }
};
// what a chunk's status means for the caller (both text feeds): 0 go on, else the function's return value
static int text_chunk_verdict(const gmx_ingest_result &res, const std::string &path, uint64_t delivered) {
  if (!res.status) return 0;
  if (res.status & GMX_INGEST_BAD_RECORD) {  // not four-line FASTQ: the host's fast path would say the same
    if (delivered == 0) return 1;
    die("gram: " + path + ": irregular FASTQ record after the first " + std::to_string(delivered) +
        " reads (multi-line or blank lines); reformat, or use a four-line FASTQ");
  }
  return delivered == 0 ? 1 : 2;  // lines of a few bytes (more records than the ingest has room for): the host reader's
}

template <class OnChunk>
int ingest_text_file(const std::string &path, int threads, int device, OnChunk on_chunk, uint64_t *delivered) {
  *delivered = 0;
  uint64_t size = 0;
  if (!looks_like_plain_fastq(path, &size)) return 1;
  FdCloser f{open(path.c_str(), O_RDONLY)};
  if (f.fd < 0) return 1;
  const uint64_t chunk = std::min<uint64_t>(device_feed_text_chunk(), size);
  DeviceFeed &df = g_device_feed;
  device_feed_prepare(device, std::max<uint64_t>(chunk, 1u << 16) + (1u << 16), 0);
  if (!df.ing) return 1;
  gmx_ingest *ing = df.ing;
  GMX_CHECK(gmx_ingest_reset(ing));
  const unsigned T = (unsigned)std::max(1, std::min(threads, 64));
  const size_t n_chunks = (size_t)((size + chunk - 1) / chunk);
  auto bytes_of = [&](size_t ci) { return (size_t)std::min<uint64_t>(chunk, size - (uint64_t)ci * chunk); };
  // A thread of its own reads the chunks ahead into four page-locked buffers (round 6: read, submit, wait and map in turn on one
  // thread cost 1.55 ms per 64 MB chunk of which 1.25 were the read — the reads now run back to back). Buffer k % 4 is free for
  // chunk k once chunk k - 4 has been waited for (its upload is over).
  constexpr size_t NB = 4;
  HostBuf<uint8_t> *const ring = df.stage;  // (sized beside the index load: device_feed_prepare; page-locking 4 x 64 MB takes 30 ms)
  struct Reader {
    std::mutex m;
    std::condition_variable cv;
    size_t staged = 0, waited = 0;  // chunks read so far / chunks the consumer is done with
    bool failed = false, stop = false;
    std::thread th;
    ~Reader() {
      {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
      }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
  } rd;
  for (size_t b = 0; b < std::min(NB, n_chunks); ++b) ring[b].resize(bytes_of(0) + 64);
  rd.th = std::thread([&]() {
    for (size_t k = 0; k < n_chunks; ++k) {
      {
        std::unique_lock<std::mutex> lk(rd.m);
        rd.cv.wait(lk, [&] { return rd.stop || k < rd.waited + NB; });
        if (rd.stop) return;
      }
      const double t_read = now_s();
      const bool ok = pread_parallel(f.fd, (uint64_t)k * chunk, ring[k % NB].data(), bytes_of(k), T);
      g_feed.read_s += now_s() - t_read;
      feed_trace("  text chunk read into page-locked memory");
      {
        std::lock_guard<std::mutex> lk(rd.m);
        if (!ok) rd.failed = true;
        rd.staged = k + 1;
      }
      rd.cv.notify_all();
      if (!ok) return;
    }
  });
  auto staged = [&](size_t k, bool block) -> int {  // 1: chunk k is read, 0: not yet (block = false), -1: its read failed
    std::unique_lock<std::mutex> lk(rd.m);
    if (block) rd.cv.wait(lk, [&] { return rd.staged > k || rd.failed; });
    if (rd.staged > k) return 1;
    return rd.failed ? -1 : 0;
  };
  size_t n_submitted = 0;
  auto submit = [&]() {
    const size_t ci = n_submitted;
    GMX_CHECK(gmx_ingest_submit_text(ing, (int)(ci % 3), ring[ci % NB].data(), bytes_of(ci), ci + 1 == n_chunks ? 1 : 0));
    ++n_submitted;
    feed_trace("text chunk submitted to the device");
  };
  bool io_error = false;
  for (size_t ci = 0; ci < n_chunks && !io_error; ++ci) {
    if (n_submitted <= ci) {  // this chunk itself: wait for its bytes
      if (staged(ci, true) < 0) {
        io_error = true;
        break;
      }
      submit();
    }
    while (n_submitted < std::min(n_chunks, ci + 3) && staged(n_submitted, false) == 1) submit();  // the two behind it, when they are there
    gmx_ingest_result res;
    GMX_CHECK(gmx_ingest_wait(ing, (int)(ci % 3), &res));
    feed_trace("text chunk scanned and packed");
    if (const char *tf = getenv("GMX_INGEST_TEST_FAIL_CHUNK"))  // test hook: "more records than the ingest has room for" in this chunk
      if ((size_t)atoll(tf) == ci) res.status |= GMX_INGEST_TOO_MANY_LINES;
    if (res.status) {
      for (size_t cj = ci + 1; cj < n_submitted; ++cj) {  // (the chunks behind are in flight: let them finish before the slots are reused)
        gmx_ingest_result drop;
        GMX_CHECK(gmx_ingest_wait(ing, (int)(cj % 3), &drop));
      }
      return text_chunk_verdict(res, path, *delivered);
    }
    on_chunk(res, (int)(ci % 3));
    *delivered += res.n_reads;
    {
      std::lock_guard<std::mutex> lk(rd.m);
      rd.waited = ci + 1;
    }
    rd.cv.notify_all();
  }
  if (io_error) {  // (chunks in flight first; then the host reader reports the file)
    for (size_t cj = 0; cj < n_submitted; ++cj) {
      gmx_ingest_result drop;
      (void)gmx_ingest_wait(ing, (int)(cj % 3), &drop);
    }
    return *delivered == 0 ? 1 : 2;
  }
  return 0;
}

// Several engines: the chunks dealt round, each uploaded to its device at once (gmx_ingest_submit_text_deferred, two per device
// ahead), scanned in file order with the cut record of the chunk before handed over through the host — as ingest_bgzf_file_dealt.
template <class OnChunk>
int ingest_text_file_dealt(const std::string &path, int threads, const std::vector<int> &devs, OnChunk on_chunk, uint64_t *delivered) {
  *delivered = 0;
  const size_t N = devs.size();
  uint64_t size = 0;
  if (!looks_like_plain_fastq(path, &size)) return 1;
  FdCloser f{open(path.c_str(), O_RDONLY)};
  if (f.fd < 0) return 1;
  // (chunks of 32 MB here — scaled for nested PRGs —: every engine's ingest and its two page-locked buffers are made for the chunk size
  //  the moment the first plain file arrives — N x (1.1 GB of device memory + 256 MB page-locked) at the single-engine size of 128 MB
  //  cost a four-engine run 0.25 s before its first read was mapped, tools/cli_dealt_text.sh)
  const uint64_t chunk = std::min<uint64_t>(getenv("GMX_TEXT_CHUNK") ? device_feed_text_chunk() : device_feed_text_chunk() / 4, size);
  while (g_more_feeds.size() + 1 < N) g_more_feeds.emplace_back(new DeviceFeed());
  auto feed = [&](size_t k) -> DeviceFeed & { return k == 0 ? g_device_feed : *g_more_feeds[k - 1]; };
  for (size_t k = 0; k < N; ++k) {
    device_feed_prepare(devs[k], std::max<uint64_t>(chunk, 1u << 16) + (1u << 16), 0, &feed(k));
    if (!feed(k).ing) return 1;
    GMX_CHECK(gmx_ingest_reset(feed(k).ing));
  }
  const size_t n_chunks = (size_t)((size + chunk - 1) / chunk);
  const unsigned T = (unsigned)std::max(1, std::min(threads, 64));
  auto dev_of = [&](size_t ci) { return ci % N; };
  auto slot_of = [&](size_t ci) { return (int)((ci / N) & 1); };
  bool io_error = false;
  auto submit = [&](size_t ci) {
    const uint64_t lo = (uint64_t)ci * chunk;
    const size_t n = (size_t)std::min<uint64_t>(chunk, size - lo);
    HostBuf<uint8_t> &st = feed(dev_of(ci)).stage[slot_of(ci)];
    st.resize(n + 64);
    if (!pread_parallel(f.fd, lo, st.data(), n, T)) io_error = true;
    GMX_CHECK(gmx_ingest_submit_text_deferred(feed(dev_of(ci)).ing, slot_of(ci), st.data(), n));
    feed_trace("text chunk submitted to its device");
  };
  size_t submitted = 0;
  for (; submitted < std::min(n_chunks, 2 * N); ++submitted) submit(submitted);
  std::vector<uint8_t> tail;
  for (size_t ci = 0; ci < n_chunks; ++ci) {
    gmx_ingest *ing = feed(dev_of(ci)).ing;
    GMX_CHECK(gmx_ingest_scan(ing, slot_of(ci), tail.data(), tail.size(), ci + 1 == n_chunks ? 1 : 0));
    gmx_ingest_result res;
    GMX_CHECK(gmx_ingest_wait(ing, slot_of(ci), &res));
    feed_trace("text chunk scanned");
    if (const char *tf = getenv("GMX_INGEST_TEST_FAIL_CHUNK"))
      if ((size_t)atoll(tf) == ci) res.status |= GMX_INGEST_TOO_MANY_LINES;
    if (io_error) res.status |= GMX_INGEST_TOO_MANY_LINES;  // (a failed read: the host reader takes the file and reports it)
    if (res.status) {
      for (size_t cj = ci + 1; cj < submitted; ++cj) {  // (chunks uploaded ahead: scanned with nothing and dropped, so that their slots are free again)
        gmx_ingest_result drop;
        GMX_CHECK(gmx_ingest_scan(feed(dev_of(cj)).ing, slot_of(cj), nullptr, 0, 0));
        GMX_CHECK(gmx_ingest_wait(feed(dev_of(cj)).ing, slot_of(cj), &drop));
      }
      return text_chunk_verdict(res, path, *delivered);
    }
    tail.resize(res.tail_bytes);
    if (res.tail_bytes && gmx_ingest_fetch_tail(ing, slot_of(ci), tail.data(), tail.size()) < 0) die(std::string("gram: ") + gmx_last_error());
    on_chunk(res, dev_of(ci), slot_of(ci));
    *delivered += res.n_reads;
    if (submitted < n_chunks) submit(submitted++);
  }
  return 0;
}

struct ReadStats {  // include/genotype/read_stats.hpp
  double mean_cov_depth = -1, variance_cov_depth = -1;
  uint64_t num_sites_noCov = 0;
  int64_t num_sites_total = -1;
  double mean_pb_error = -1;
  int64_t no_qual_reads = -1, num_bases_processed = -1;
  uint64_t max_read_length = 0;
};

// ReadStats::process_read_perbase_error_rates (src/genotype/read_stats.cpp:21-70): first <= 10000 reads with qualities
void compute_base_error_rate(const std::string &path, ReadStats &rs) {
  SeqReader reader(path);
  SeqRecord rec;
  uint64_t informative = 0;
  int64_t no_qual = 0, n_bases = 0;
  float running = 0.0f;
  while (informative < 10000 && reader.next(rec)) {
    if (rec.seq.size() > rs.max_read_length) rs.max_read_length = rec.seq.size();
    if (rec.qual.empty()) {
      no_qual++;
      continue;
    }
    for (char q : rec.qual) {
      running += (float)(q - 33);
      n_bases++;
    }
    informative++;
  }
  double mean_error = 0;
  if (n_bases > 0) {
    double mean_qual = running / n_bases;
    mean_error = std::pow(10, -mean_qual / 10);
  }
  rs.num_bases_processed = n_bases;
  rs.no_qual_reads = no_qual;
  rs.mean_pb_error = mean_error;
}

void write_read_stats(const std::string &path, const ReadStats &rs) {  // ReadStats::serialise, read_stats.cpp:162-209
  std::ofstream o(path);
  o << "\n{\n\"Read_depth\":\n    {\"Mean\": " << rs.mean_cov_depth << ",";
  o << "\n    \"Variance\": " << rs.variance_cov_depth << ",";
  o << "\n    \"num_sites_noCov\": " << rs.num_sites_noCov << ",";
  o << "\n    \"num_sites_total\": " << rs.num_sites_total;
  o << "\n    },";
  o << "\n\"Max_read_length\": " << rs.max_read_length << ",";
  o << "\n\"Quality\":\n    {\"Error_rate_mean\": " << rs.mean_pb_error << ",";
  o << "\n    \"Num_bases\": " << rs.num_bases_processed << ",";
  o << "\n    \"No_qual_reads\": " << rs.no_qual_reads;
  o << "\n    }}\n";
  close_checked(o, path);
}


struct Args {
  std::map<std::string, std::vector<std::string>> opt;
  bool has(const std::string &k) const { return opt.count(k) != 0; }
  std::string one(const std::string &k) const {
    auto it = opt.find(k);
    if (it == opt.end() || it->second.size() != 1) die("the option '--" + k + "' is required and takes one value\n" + kGenotypeHelp);
    return it->second[0];
  }
};

Args parse_sub(int argc, const char *const *argv, int from) {
  Args a;
  for (int i = 1; i < from && i < argc; ++i)  // `--debug` is a global switch (main.cpp:58): also accepted ahead of the command
    if (std::string(argv[i]) == "--debug") a.opt["debug"];
  std::string cur;
  for (int i = from; i < argc; ++i) {
    std::string t = argv[i];
    if (t.rfind("--", 0) == 0) {
      cur = t.substr(2);
      size_t eq = cur.find('=');
      if (eq != std::string::npos) {
        a.opt[cur.substr(0, eq)].push_back(cur.substr(eq + 1));
        cur = cur.substr(0, eq);
      } else
        a.opt[cur];
    } else if (!cur.empty())
      a.opt[cur].push_back(t);
  }
  return a;
}

// `gram _parse_check FILE THREADS`: prints what each parser makes of FILE as "<name> <n_reads> <n_bases> <fnv1a>" (or
// "fast declined"); the two lines must agree whenever the fast path accepts the file.
int run_parse_check(const std::string &path, int threads) {
  struct Flat {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets{0};
  };
  auto fnv = [](const Flat &p) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
      for (int i = 0; i < 8; ++i) {
        h ^= (v >> (8 * i)) & 0xFF;
        h *= 1099511628211ull;
      }
    };
    for (auto o : p.offsets) mix(o);
    for (auto b : p.bases) mix(b);
    return h;
  };
  Flat fast, slow;
  auto collect = [&](const ParsedReads &block) {  // planes back to one byte per base; an unencodable read as the empty read
    for (size_t r = 0; r < block.n_reads; ++r) {
      if (!block.skip[r]) {
        const uint64_t at = block.pair_of(r);
        const uint32_t len = block.len_of(r);
        for (uint32_t i = 0; i < len; ++i) {
          const uint64_t w = block.planes[at + (i >> 5)];
          fast.bases.push_back((uint8_t)(1u + ((w >> (i & 31u)) & 1u) + 2u * ((w >> (32u + (i & 31u))) & 1u)));
        }
      }
      fast.offsets.push_back(fast.bases.size());
    }
  };
  if (parse_fastq_file(path, threads, collect))
    std::cout << "fast " << fast.offsets.size() - 1 << " " << fast.bases.size() << " " << fnv(fast) << std::endl;
  else
    std::cout << "fast declined" << std::endl;
  SeqReader reader(path);
  SeqRecord rec;
  while (reader.next(rec)) {
    encode_read(rec.seq, slow.bases);
    slow.offsets.push_back(slow.bases.size());
  }
  std::cout << "slow " << slow.offsets.size() - 1 << " " << slow.bases.size() << " " << fnv(slow) << std::endl;
  if (getenv("GMX_PARSE_CHECK_DEVICE")) {  // the same file through the device-side decoder (needs a GPU): a third line
    fast = Flat{};
    ParsedReads blk;
    uint64_t delivered = 0;
    auto take = [&](const gmx_ingest_result &res, int slot) {
      blk.reset();
      blk.n_reads = res.n_reads;
      blk.n_bases = res.n_bases;
      blk.uniform_len = res.uniform_len;
      blk.planes.resize(res.n_pairs + 8);
      blk.offsets.resize(res.n_reads + 1);
      blk.skip.resize(std::max<uint64_t>(res.n_reads, 1));
      GMX_CHECK(gmx_ingest_fetch_reads(g_device_feed.ing, slot, blk.planes.data(), blk.offsets.data(), blk.skip.data()));
      collect(blk);
    };
    int rc = ingest_bgzf_file(path, threads, 0, take, &delivered);
    if (rc == 1 && delivered == 0) rc = ingest_text_file(path, threads, 0, take, &delivered);  // (not BGZF: plain text through the same kernels)
    if (rc == 0)
      std::cout << "device " << fast.offsets.size() - 1 << " " << fast.bases.size() << " " << fnv(fast) << std::endl;
    else
      std::cout << "device " << (rc == 1 ? "declined" : "failed") << " after " << delivered << " reads" << std::endl;
  }
  return 0;
}

// `gram _parse_bench FILE THREADS [REPEATS]`: the parallel FASTQ parser alone (no engine): wall time per pass and where it went
int run_parse_bench(const std::string &path, int threads, int repeats) {
  for (int rep = 0; rep < repeats; ++rep) {
    g_feed = FeedTimes{};
    uint64_t n = 0, bases = 0;
    const double t0 = now_s();
    const bool ok = parse_fastq_file(path, threads, [&](ParsedReads &b) {
      n += b.n_reads;
      bases += b.n_bases;
    });
    const double dt = now_s() - t0;
    std::cout << (ok ? "parsed " : "declined ") << n << " reads, " << bases << " bases in " << dt << " s = " << n / dt / 1e6
              << " M reads/s: read " << g_feed.read_s << ", parse " << g_feed.parse_s << " (scan " << g_feed.scan_s << ", pack "
              << g_feed.pack_s << "), slot wait " << g_feed.wait_slot_s << std::endl;
  }
  return 0;
}

int run_build(const Args &a) {
  // The reference's `gram build` writes SDSL/Boost artefacts derived from gram_dir/prg (src/build/build.cpp:8-72).
  // This engine derives its own tables from `prg`; build validates the PRG and leaves them in gram_dir as one cache
  // file that `gram genotype` loads instead of rebuilding (it rebuilds in memory when the file is absent or stale).
  std::string gram_dir = a.one("gram_dir");
  uint32_t k = a.has("kmer_size") ? (uint32_t)std::stoul(a.one("kmer_size")) : 0;
  int threads = a.has("max_threads") ? std::stoi(a.one("max_threads")) : 0;
  gmx_index *ix = nullptr;
  GMX_CHECK(gmx_index_build_from_file(join(gram_dir, "prg").c_str(), k, threads, &ix));
  gmx_index_info info;
  gmx_index_get_info(ix, &info);
  std::cout << "PRG ok: " << info.n_text - 1 << " symbols, " << info.n_sites << " variant sites, " << info.n_kmers_present
            << " indexed kmers" << std::endl;
  if (k > 0) {
    const std::string cache = join(gram_dir, "gmx_index.k" + std::to_string(k) + ".bin");
    GMX_CHECK(gmx_index_save(ix, cache.c_str()));
    std::cout << "Wrote index cache " << cache << std::endl;
  }
  int rc = 0;
  // a gram_dir that a STOCK `gramtools build` filled (kmers, kmers_stats, sa_intervals, paths, the four base
  // masks: build/kmer_index/dump.cpp:27-137, load.cpp:71-173; cov_graph: prg_info.cpp:13-15) checked against the index built
  // here from the same prg; --write_stock writes the SDSL vectors and masks of this index in those formats.
  if (a.has("write_stock") && k > 0) {
    GMX_CHECK(gmx_index_write_stock_files(ix, gram_dir.c_str()));
    std::cout << "Wrote kmers, kmers_stats, sa_intervals, paths and the four base masks in the stock (SDSL) formats" << std::endl;
  }
  if (a.has("check_stock") && k > 0) {
    gmx_stock_report rep;
    GMX_CHECK(gmx_index_check_stock_files(ix, gram_dir.c_str(), &rep));
    const bool ok = rep.kmer_mismatches == 0 && rep.duplicate_kmers == 0 && rep.mask_mismatches == 0 && rep.cov_graph_state != 3;
    std::cout << "Stock files: " << rep.kmers << " kmers, " << rep.states << " states, " << rep.kmer_mismatches << " kmers with other states, "
              << rep.kmers_missing_in_files << " indexed here and not in the files, " << rep.duplicate_kmers << " duplicates; masks: " << rep.mask_bits
              << " bits, " << rep.mask_mismatches << " differ; cov_graph: "
              << (rep.cov_graph_state == 0 ? "absent" : rep.cov_graph_state == 1 ? "archive signature found" : rep.cov_graph_state == 2 ? "archive signature and site count agree" : "site count differs")
              << (rep.cov_graph_state ? " (Boost archive version " + std::to_string(rep.cov_graph_library_version) + ")" : std::string())
              << "; fm_index: " << (rep.fm_index_bytes ? std::to_string(rep.fm_index_bytes) + " bytes (not decoded)" : std::string("absent")) << std::endl;
    std::cout << (ok ? "Stock files agree with the native index" : "Stock files DIFFER from the native index") << std::endl;
    if (!ok) rc = 1;
  }
  gmx_index_destroy(ix);
  return rc;
}

// The master generator's raw draws (RandomInclusiveInt's mt19937, random.cpp:4-19: operator() = the raw 32-bit output),
// produced ahead of the reads on a thread of its own in chunks of 1 M draws: 2.5 ns a draw on the feed's consumer
// thread was half of that thread's time. copy() waits for the chunks it needs; at most kAhead chunks are drawn beyond
// the last one asked for.
class SeedStream {
 public:
  uint64_t base = 0;  // draw index of the current file's first read
  explicit SeedStream(uint32_t seed) : gen_(seed) {
    worker_ = std::thread([this]() { run(); });
  }
  ~SeedStream() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    worker_.join();
  }
  void copy(uint64_t first, uint64_t n, uint32_t *out) {
    for (uint64_t done = 0; done < n;) {
      const uint64_t at = first + done, c = at / kChunk, in = at % kChunk, take = std::min<uint64_t>(n - done, kChunk - in);
      const uint32_t *src;
      {
        std::unique_lock<std::mutex> lk(m_);
        want_ = std::max(want_, c);
        cv_.notify_all();
        cv_.wait(lk, [&] { return chunks_.size() > c; });
        src = chunks_[c].get();
        for (; freed_ < c; ++freed_) chunks_[freed_].reset();  // (requests only move forward: the chunks behind are done)
      }
      memcpy(out + done, src + in, take * sizeof(uint32_t));
      done += take;
    }
  }

 private:
  static constexpr uint64_t kChunk = 1u << 20, kAhead = 16;
  void run() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || chunks_.size() <= want_ + kAhead; });
        if (stop_) return;
      }
      std::unique_ptr<uint32_t[]> c(new uint32_t[kChunk]);
      for (uint64_t i = 0; i < kChunk; ++i) c[i] = (uint32_t)gen_();
      {
        std::lock_guard<std::mutex> lk(m_);
        chunks_.push_back(std::move(c));
      }
      cv_.notify_all();
    }
  }
  std::mt19937 gen_;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::unique_ptr<uint32_t[]>> chunks_;
  uint64_t want_ = 0, freed_ = 0;
  bool stop_ = false;
  std::thread worker_;
};

int run_genotype(const Args &a) {
  using clk = std::chrono::steady_clock;
  // The HIP runtime, device 0's context and the library's kernels come up on a thread of their own (150-250 ms) while this
  // one reads the arguments, the first reads and the index cache; joined before the first call that needs a device.
  std::thread hip_warm([]() { (void)gmx_device_warmup(0); });
  struct Joiner {
    std::thread &t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } hip_warm_join{hip_warm};
  std::string gram_dir = a.one("gram_dir");
  // One sample (the reference's interface: genotype/parameters.cpp:54-72), or --samples_list: many samples on one index
  // upload — "(tens of!) thousands of samples" of one species (the reference's README) pay the start-up once.
  struct Sample {
    std::string id, run_dir;
    std::vector<std::string> reads;
  };
  std::vector<Sample> samples;
  if (a.has("samples_list")) {
    if (a.has("reads") || a.has("sample_id") || a.has("genotype_dir")) die(std::string("--samples_list replaces --reads, --sample_id and --genotype_dir\n") + kGenotypeHelp);
    std::ifstream in(a.one("samples_list"));
    if (!in) die("cannot read the samples list " + a.one("samples_list"));
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty() || line[0] == '#') continue;
      std::vector<std::string> f;
      size_t i = 0;
      while (i <= line.size()) {
        size_t j = line.find('\t', i);
        if (j == std::string::npos) j = line.size();
        f.push_back(line.substr(i, j - i));
        i = j + 1;
      }
      if (f.size() < 3 || f[0].empty() || f[1].empty() || f[2].empty()) die("samples list: a line needs sample_id, genotype_dir and at least one reads file, tab-separated: " + line);
      samples.push_back(Sample{f[0], f[1], std::vector<std::string>(f.begin() + 2, f.end())});
    }
    if (samples.empty()) die("samples list: no sample in " + a.one("samples_list"));
  } else {
    if (!a.has("reads") || a.opt.at("reads").empty()) die(std::string("the option '--reads' is required but missing\n") + kGenotypeHelp);
    samples.push_back(Sample{a.one("sample_id"), a.one("genotype_dir"), a.opt.at("reads")});
  }
  std::vector<std::string> reads_paths;  // of all samples (the automatic choice of devices looks at their total size)
  for (auto const &smp : samples) reads_paths.insert(reads_paths.end(), smp.reads.begin(), smp.reads.end());
  std::string ploidy = a.one("ploidy");
  if (ploidy != "haploid" && ploidy != "diploid") die(std::string("Invalid/unsupported ploidy\n") + kGenotypeHelp);
  uint32_t kmer_size = (uint32_t)std::stoul(a.one("kmer_size"));
  int max_threads = a.has("max_threads") ? std::stoi(a.one("max_threads")) : 1;
  // --device N, or --devices 0-7 / 0,2,5: one engine and one host thread per listed GPU, reads dealt by read index,
  // coverage summed at the end (gmx.h: gmx_group_*). The result does not depend on the number of GPUs.
  std::vector<int> devices;
  bool devices_auto = false;  // the list below was chosen here, not named by the caller
  if (a.has("devices")) {
    std::string spec = a.one("devices");
    size_t i = 0;
    while (i < spec.size()) {
      size_t j = spec.find(',', i);
      std::string part = spec.substr(i, j == std::string::npos ? std::string::npos : j - i);
      size_t dash = part.find('-');
      try {
        if (dash == std::string::npos)
          devices.push_back(std::stoi(part));
        else
          for (int d = std::stoi(part.substr(0, dash)); d <= std::stoi(part.substr(dash + 1)); ++d) devices.push_back(d);
      } catch (std::exception const &) {
        die("--devices takes a list like 0-7 or 0,2,5");
      }
      if (j == std::string::npos) break;
      i = j + 1;
    }
    if (devices.empty()) die("--devices takes a list like 0-7 or 0,2,5");
  } else if (a.has("device")) {
    devices.push_back(std::stoi(a.one("device")));
  } else {
    // Neither given — what the unmodified front-end does (genotype.py:71-93 passes a fixed argument list): every visible
    // GPU when the reads are worth sharding, else device 0. "Worth it": the files hold more than GMX_AUTO_DEVICES_MIN_MB of
    // reads (default 2048 MB of FASTQ, a quarter of that gzipped: ~6 M reads of 150 bp; below it a second index upload costs
    // more than it saves). GMX_AUTO_DEVICES=0 keeps device 0, =N takes at most N.
    devices.push_back(0);
    devices_auto = true;
    uint64_t mb = 0;
    for (auto const &p : reads_paths) {
      struct stat st;
      if (stat(p.c_str(), &st) != 0) continue;
      const bool gz = p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0;
      mb += ((uint64_t)st.st_size >> 20) * (gz ? 4 : 1);
    }
    uint64_t min_mb = 2048;
    if (const char *mm = getenv("GMX_AUTO_DEVICES_MIN_MB")) min_mb = (uint64_t)atoll(mm);
    if (mb >= min_mb) {  // (only then is the device count needed this early: it waits for the runtime to come up)
      int n_vis = gmx_device_count();
      if (const char *ad = getenv("GMX_AUTO_DEVICES")) n_vis = std::min(n_vis, std::max(1, atoi(ad)));
      for (int d = 1; d < n_vis; ++d) devices.push_back(d);
    }
  }
  int rng_mode = 0;
  if (a.has("rng_compat")) {
    std::string m = a.one("rng_compat");
    if (m == "gcc10")
      rng_mode = 1;
    else if (m != "gcc11")
      die("--rng_compat must be gcc11 or gcc10");
  }
  for (auto const &smp : samples) {
    mkdirs(join(smp.run_dir, "coverage"));
    mkdirs(join(smp.run_dir, "genotype"));
  }

  std::cout << "Executing genotype command" << std::endl;
  phase("arguments parsed, output directories made");

  auto t0 = clk::now();
  // Beside the index load: the page-locked buffers the reads feed will ask for (two parsed blocks in flight: bases,
  // offsets, seeds, sized for 150 bp reads in blocks of GMX_FASTQ_BLOCK bytes; other sizes are allocated when needed).
  // Freed right away, they wait in the library's cache of page-locked blocks.
  std::thread prewarm([&]() {
    if (!getenv("GMX_HOST_GZ")) {  // a BGZF reads file: the device-side decoder's memory, now (on the first device)
      for (const auto &smp : samples) {
        if (smp.reads.empty()) continue;
        unsigned char h[16] = {0};
        struct stat sb;
        const int fd = open(smp.reads[0].c_str(), O_RDONLY);
        if (fd < 0) break;
        const bool bg = pread(fd, h, 16, 0) == 16 && fstat(fd, &sb) == 0 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && h[3] == 4 && h[12] == 'B' && h[13] == 'C';
        close(fd);
        if (bg && hipSetDeviceForPrewarm(devices[0])) {
          const uint64_t text = device_feed_text_for((uint64_t)sb.st_size * 6);
          device_feed_prepare(devices[0], text, std::min<uint64_t>((uint64_t)sb.st_size, text / 2) + 64);
        } else if (h[0] == '@' && plain_fastq_on_device(max_threads, devices.size()) && fstat_ok_size(smp.reads[0]) >= 4) {  // plain FASTQ: text chunks (ingest_text_file)
          const uint64_t chunk = std::min<uint64_t>(device_feed_text_chunk(1), fstat_ok_size(smp.reads[0]));  // (a nested PRG's larger chunks: sized when the file is opened)
          device_feed_prepare(devices[0], std::max<uint64_t>(chunk, 1u << 16) + (1u << 16), chunk + 64);
        }
        break;
      }
    }
    size_t block = 96u << 20;
    if (const char *eb = getenv("GMX_FASTQ_BLOCK")) block = std::max<size_t>(64, (size_t)atoll(eb));
    if (block < (8u << 20)) return;
    void *p[8];
    const size_t sizes[4] = {block / 6, block / 24, block / 48, block / 192};  // planes, offsets, seeds, skip flags
    for (int i = 0; i < 8; ++i) p[i] = gmx_host_alloc(sizes[i % 4]);
    for (int i = 0; i < 8; ++i) gmx_host_free(p[i]);
    // (the file block buffer is needed by gzip and unmappable files only: parse_fastq_file allocates it then)
  });
  {  // ... and the parser's threads, started and parked (63 thread starts cost the first block 3 ms)
    const unsigned T = (unsigned)std::max(1, std::min(max_threads, 128));
    parallel_for(T, [](unsigned) {});
  }
  std::cout << "Loading PRG data" << std::endl;
  gmx_index *ix = nullptr;
  {  // the index cache `gram build` leaves in gram_dir (gmx_index.k<K>.bin); rebuilt in memory when absent or stale
    const std::string cache = join(gram_dir, "gmx_index.k" + std::to_string(kmer_size) + ".bin");
    if (gmx_index_load(cache.c_str(), join(gram_dir, "prg").c_str(), kmer_size, &ix) == GMX_OK)
      std::cout << "Loaded index cache " << cache << std::endl;
    else
      GMX_CHECK(gmx_index_build_from_file(join(gram_dir, "prg").c_str(), kmer_size, max_threads, &ix));
  }
  phase("index loaded (host)");
  gmx_index_info info;
  GMX_CHECK(gmx_index_get_info(ix, &info));
  if (info.is_nested) {  // (a batch's straggler tail: larger batches)
    g_block_scale = 12;
    g_ingest_member_scale = 3;
  }
  std::cout << "Loading kmer index data" << std::endl;
  gmx_engine_opts opts;
  gmx_engine_default_opts(&opts);
  opts.rng_mode = rng_mode;
  if (hip_warm.joinable()) hip_warm.join();
  phase("HIP runtime up (warm-up thread joined)");
  gmx_group *grp = nullptr;
  {
    int grc = gmx_group_create(ix, &opts, devices.data(), (int)devices.size(), &grp);
    if (grc != GMX_OK && devices_auto && devices.size() > 1) {
      // The list was OUR choice (neither --device nor --devices: every visible GPU for a large reads file). A busy or full
      // secondary GPU on a shared node must not fail a run that device 0 alone — the default before the auto-choice
      // existed — would have completed: the devices that do come up, at least device 0. Named devices stay a hard failure.
      std::cerr << "warning: not every visible GPU could be used (" << gmx_last_error() << "); ";
      std::vector<int> usable;
      for (int d : devices) {
        gmx_group *one = nullptr;
        if (d == 0 || gmx_group_create(ix, &opts, &d, 1, &one) == GMX_OK) usable.push_back(d);
        if (one) gmx_group_destroy(one);
      }
      std::cerr << "continuing on " << usable.size() << " of " << devices.size() << " devices" << std::endl;
      devices = usable;
      grc = gmx_group_create(ix, &opts, devices.data(), (int)devices.size(), &grp);
      if (grc != GMX_OK && devices.size() > 1) {
        devices.assign(1, 0);
        grc = gmx_group_create(ix, &opts, devices.data(), 1, &grp);
      }
    }
    GMX_CHECK(grc);
  }
  phase("engines created (HIP start-up, index upload)");
  gmx_engine *eng = gmx_group_engine(grp, 0);  // after the exchange every engine holds the totals: engine 0 is read back
  // workspace for the calls the feed will make (a block of a reads file per call, at most 1 M reads per engine)
  for (int d = 0; d < gmx_group_size(grp); ++d) GMX_CHECK(gmx_engine_reserve_packed(gmx_group_engine(grp, d), info.is_nested ? 4u << 20 : 3u << 19 /* (a decoded BGZF chunk of 150 bp reads: 1.46 M) */, ((info.is_nested ? 24ull : 6ull) << 20) + 64));
  if (prewarm.joinable()) prewarm.join();
  phase("workspace reserved, page-locked buffers warmed");
  double t_load = std::chrono::duration<double>(clk::now() - t0).count();

  for (size_t sample_i = 0; sample_i < samples.size(); ++sample_i) {  // ---- one sample: what a call of its own does from here on ----
  const std::vector<std::string> &reads_paths = samples[sample_i].reads;
  const std::string &sample_id = samples[sample_i].id, &run_dir = samples[sample_i].run_dir;
  const std::string cov_dir = join(run_dir, "coverage"), geno_dir = join(run_dir, "genotype");
  if (samples.size() > 1) std::cout << "==== sample " << sample_id << " (" << sample_i + 1 << " of " << samples.size() << ")" << std::endl;
  uint32_t seed;
  if (a.has("seed"))
    seed = (uint32_t)std::stoul(a.one("seed"));
  else {
    std::random_device rd;  // random.cpp:8-13
    seed = rd();
  }
  ReadStats rs;
  compute_base_error_rate(reads_paths[0], rs);  // genotype.cpp:32-34
  phase("base error rate of the first 10 000 reads");
  if (sample_i > 0) {  // the accumulators and read counters of the sample before
    for (int d = 0; d < gmx_group_size(grp); ++d) GMX_CHECK(gmx_engine_reset(gmx_group_engine(grp, d)));
    g_feed = FeedTimes{};
  }
  std::cout << "Running quasimap" << std::endl;
  std::cout << "Generating allele quasimap data structure" << std::endl;
  std::cout << "Done generating allele quasimap data structure" << std::endl;
  std::cout << "Master random seed for read selection: " << seed << std::endl;
  std::cout << "Maximum thread count: " << max_threads << std::endl;
  std::cout << "Processing reads:" << std::endl;
  t0 = clk::now();
  feed_trace("quasimap stage starts");
  // One master mt19937(seed) for all files; 5000 draws per batch of <= 5000 reads (quasimap.cpp:120-141).
  std::mt19937 master(seed);
  SeedStream seed_stream(seed);  // the same stream, drawn ahead on a thread of its own (fast path)
  const uint64_t kBatch = 5000;
  const uint64_t kChunkReads = (uint64_t)(1u << 20) * devices.size();  // reads staged per call: 1 M per engine (multiple of 5000 not required: seeds are per read)
  uint64_t total_reads = 0;
  for (auto const &path : reads_paths) {
    // fast path: seeds as the batch loop below draws them (5000 master draws per batch of <= 5000 reads, per file): read i
    // of the file takes draw file_base + i of the master stream, and the file uses up ceil(n / 5000) * 5000 draws
    uint64_t in_file = 0;
    const uint64_t file_base = seed_stream.base;
    // A BGZF file is decoded on the GPU (GMX_HOST_GZ=1: on the host as before): compressed members up, reads found and packed in
    // HBM, mapped where they lie. Should the device decoder give up on a chunk, the host reader below takes the file from its
    // start and drops the reads already mapped. With several engines the file's chunks are dealt round: every GPU decodes and maps
    // its share (ingest_bgzf_file_dealt), the record a chunk's end cuts travels to the next GPU through the host.
    uint64_t skip_reads = 0;
    if (!getenv("GMX_HOST_GZ")) {
      static std::vector<std::unique_ptr<HostBuf<uint32_t>>> dev_seeds;  // per engine and slot
      while (dev_seeds.size() < 3 * devices.size()) dev_seeds.emplace_back(new HostBuf<uint32_t>());
      uint64_t delivered = 0;
      // several engines: every one decodes and maps its share of the file's chunks (GMX_INGEST_ONE_DEVICE=1: the first one all of them)
      const bool dealt = devices.size() > 1 && !getenv("GMX_INGEST_ONE_DEVICE");
      auto map_chunk = [&](const gmx_ingest_result &res, size_t k, int slot) {  // engine k's ingest holds the chunk's reads in its HBM
        const uint64_t n = res.n_reads;
        if (n == 0) return;
        gmx_engine *ek = gmx_group_engine(grp, (int)k);
        HostBuf<uint32_t> &sd = *dev_seeds[3 * k + (size_t)slot];
        // The kernels read the few seeds they need in place: the buffer a slot used three chunks ago must be done with. One ingest,
        // chunks taking its slots in turn: it is — this chunk's result was waited for, its scan waited (on the device) for the
        // slot's release, and the slot was released behind the mapping kernels of the chunk three before (gmx_ingest_release_after
        // below; ing_begin). Until round 5 the engine was synchronised here: the host sat out the mapping of the chunk before,
        // queued behind the inflate kernels, 6-13 ms of every chunk in which nothing new went to the device. Chunks dealt over
        // several ingests are submitted in another order: there the engine is synchronised as before.
        if (dealt) GMX_CHECK(gmx_engine_sync(ek));
        sd.resize(n);
        feed_trace("  chunk's seed buffer ready");
        seed_stream.copy(file_base + in_file, n, sd.data());
        feed_trace("  chunk's seeds copied");
        if (res.uniform_len) {
          GMX_CHECK(gmx_map_reads_packed_device(ek, res.d_planes, nullptr, res.uniform_len, sd.data(), res.any_skip ? res.d_skip : nullptr, n));
        } else {
          for (uint64_t r0 = 0, i = 0; r0 < n; r0 += 1u << 20, ++i) {
            const uint64_t m = std::min<uint64_t>(1u << 20, n - r0);
            GMX_CHECK(gmx_map_reads_packed_device(ek, res.d_planes + res.sub_pairs[i], res.d_offsets + r0, 0, sd.data() + r0,
                                                  res.any_skip ? res.d_skip + r0 : nullptr, m));
          }
        }
        feed_trace("  chunk handed to the engine");
        // (the slot's planes are read by the launch just enqueued: on the engine's NULL stream or on its second workspace's stream)
        GMX_CHECK(gmx_ingest_release_after(k == 0 ? g_device_feed.ing : g_more_feeds[k - 1]->ing, slot, nullptr));
        if (void *second = gmx_engine_second_stream(ek)) GMX_CHECK(gmx_ingest_release_after(k == 0 ? g_device_feed.ing : g_more_feeds[k - 1]->ing, slot, second));
        in_file += n;
        total_reads += n;
      };
      int rc = dealt ? ingest_bgzf_file_dealt(path, max_threads, devices, map_chunk, &delivered)
                     : ingest_bgzf_file(path, max_threads, devices[0], [&](const gmx_ingest_result &res, int slot) { map_chunk(res, 0, slot); }, &delivered);
      // not BGZF: plain four-line FASTQ takes the same route minus the inflate kernel (round 6; GMX_HOST_FASTQ=1: the host parser)
      const bool text_route = rc == 1 && delivered == 0 && plain_fastq_on_device(max_threads, devices.size());
      if (text_route)
        rc = dealt ? ingest_text_file_dealt(path, max_threads, devices, map_chunk, &delivered)
                   : ingest_text_file(path, max_threads, devices[0], [&](const gmx_ingest_result &res, int slot) { map_chunk(res, 0, slot); }, &delivered);
      auto sync_all = [&]() {
        for (int d = 0; d < (dealt ? gmx_group_size(grp) : 1); ++d) GMX_CHECK(gmx_engine_sync(gmx_group_engine(grp, d)));
      };
      if (rc == 0) {
        sync_all();
        seed_stream.base = file_base + (in_file + kBatch - 1) / kBatch * kBatch;
        continue;
      }
      if (rc == 2) {
        sync_all();
        std::cerr << "warning: " << path << ": the device-side " << (text_route ? "FASTQ scanner" : "BGZF decoder") << " gave up after " << delivered << " reads; the host reader takes over" << std::endl;
        skip_reads = delivered;
        in_file = 0;               // (the host reader counts the file's reads from its start again)
        total_reads -= delivered;
      }
    }
    auto sink = [&](ParsedReads &block) {  // runs on the pipe's consumer thread, block after block in file order
      uint64_t n = block.n_reads, first = 0;
      if (skip_reads) {  // reads the device feed already mapped (a sub-range of a packed batch is a packed batch, gmx.h)
        first = std::min<uint64_t>(skip_reads, n);
        skip_reads -= first;
        in_file += first;
        total_reads += first;
        n -= first;
      }
      block.seeds.resize(std::max<uint64_t>(n, 1));
      const double t_seeds = now_s();
      seed_stream.copy(file_base + in_file, n, block.seeds.data());
      g_feed.seeds_s += now_s() - t_seeds;
      in_file += n;
      // the block goes up as it is — bit planes from page-locked memory, chunk by chunk beside the kernels (the call returns
      // once everything is enqueued) — and may be overwritten by the parser as soon as its uploads are done
      if (n) {
        GMX_CHECK(gmx_group_map_reads_packed_host(grp, block.planes.data() + block.pair_of(first), block.uniform_len ? nullptr : block.offsets.data() + first,
                                                  block.uniform_len, block.seeds.data(), block.any_skip ? block.skip.data() + first : nullptr, n));
        GMX_CHECK(gmx_group_sync_uploads(grp));
      }
      total_reads += n;
    };
    if (parse_fastq_file(path, max_threads, sink)) {
      seed_stream.base = file_base + (in_file + kBatch - 1) / kBatch * kBatch;
      continue;
    }
    if (skip_reads) die("gram: " + path + ": the device-side decoder delivered reads of a file the host readers cannot parse");
    // (the general reader below draws from `master`: bring it to where the stream stands)
    master.seed(seed);
    master.discard(seed_stream.base);
    const uint64_t reads_before = total_reads;
    SeqReader reader(path);
    SeqRecord rec;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets{0};
    std::vector<uint32_t> seeds;
    uint64_t in_batch = 0;
    auto flush = [&]() {
      if (offsets.size() > 1) {
        if (bases.empty()) bases.push_back(0);
        GMX_CHECK(gmx_group_map_reads_host(grp, bases.data(), offsets.data(), seeds.data(), offsets.size() - 1));
      }
      bases.clear();
      offsets.assign(1, 0);
      seeds.clear();
    };
    std::vector<uint32_t> batch_seeds(kBatch);
    while (reader.next(rec)) {
      if (in_batch == 0)
        for (auto &s : batch_seeds) s = (uint32_t)master();  // always 5000 draws per batch
      encode_read(rec.seq, bases);  // an unencodable read stays as an empty read: counted as skipped, keeps its seed
      offsets.push_back(bases.size());
      seeds.push_back(batch_seeds[in_batch]);
      if (++in_batch == kBatch) in_batch = 0;
      total_reads++;
      if (offsets.size() - 1 >= kChunkReads && in_batch == 0) flush();
    }
    flush();
    seed_stream.base += (total_reads - reads_before + kBatch - 1) / kBatch * kBatch;
  }
  feed_trace("files done");
  GMX_CHECK(gmx_group_allreduce(grp));  // the one exchange (a single engine: nothing to do)
  feed_trace("exchange done");
  GMX_CHECK(gmx_engine_sync(eng));
  feed_trace("engine synchronised");
  double t_map = std::chrono::duration<double>(clk::now() - t0).count();
  phase("quasimap done (reads parsed, mapped, coverage exchanged)");

  // ---- coverage read-back + uint16 semantics --------------------------------------------------------------
  std::vector<uint32_t> allele_sum(std::max<uint32_t>(info.n_allele_slots, 1)), per_base(std::max<uint32_t>(info.n_per_base_slots, 1)),
      grouped(std::max<uint32_t>(info.n_grouped_slots, 1));
  gmx_stats st;
  GMX_CHECK(gmx_coverage_fetch(eng, allele_sum.data(), per_base.data(), grouped.data(), &st));
  int64_t n_log = gmx_coverage_fetch_grouped_log(eng, nullptr, 0);
  if (n_log < 0) die(std::string("gram: grouped log: ") + gmx_last_error());
  std::vector<uint32_t> glog(std::max<int64_t>(n_log, 1));
  if (n_log) gmx_coverage_fetch_grouped_log(eng, glog.data(), (uint64_t)n_log);

  gmx_depth_stats ds;  // readstats.compute_coverage_depth, quasimap.cpp:48
  GMX_CHECK(gmx_compute_coverage_depth(ix, per_base.data(), grouped.data(), glog.data(), (uint64_t)n_log, &ds));
  rs.mean_cov_depth = ds.mean_cov_depth;
  rs.variance_cov_depth = ds.variance_cov_depth;
  rs.num_sites_noCov = ds.num_sites_noCov;
  rs.num_sites_total = (int64_t)ds.num_sites_total;

  std::vector<uint32_t> n_alleles(info.n_sites), as_off(info.n_sites), g_off(info.n_sites);
  GMX_CHECK(gmx_index_site_layout(ix, n_alleles.data(), as_off.data(), g_off.data(), nullptr, nullptr));

  // The three coverage files and read_stats.json are written on a thread of their own, beside the genotyping model below
  // (round 5: at configs[1] they were 42 ms of a call whose mapping takes 24; nothing below reads what they write).
  std::string rs_path = join(run_dir, "read_stats.json");
  std::cout << "Writing read stats to " << rs_path << std::endl;
  std::thread cov_writer([&]() {
  {  // coverage::dump::allele_sum (allele_sum.cpp:45-57): uint16 wrap
    std::ofstream o(join(cov_dir, "allele_sum_coverage"));
    for (uint32_t s = 0; s < info.n_sites; ++s) {
      for (uint32_t al = 0; al < n_alleles[s]; ++al) {
        o << (allele_sum[as_off[s] + al] & 0xFFFFu);
        if (al + 1 < n_alleles[s]) o << " ";
      }
      o << "\n";
    }
    close_checked(o, join(cov_dir, "allele_sum_coverage"));
  }
  {  // coverage::dump::allele_base (allele_base.cpp:49-107): saturating uint16; [] for nested PRGs
    std::ofstream o(join(cov_dir, "allele_base_coverage.json"));
    o << "{\"allele_base_counts\":[";
    if (!info.is_nested) {
      std::vector<uint32_t> pb_off(std::max<uint32_t>(info.n_allele_slots, 1)), pb_len(std::max<uint32_t>(info.n_allele_slots, 1));
      GMX_CHECK(gmx_index_allele_base_layout(ix, pb_off.data(), pb_len.data()));
      for (uint32_t s = 0; s < info.n_sites; ++s) {
        o << "[";
        for (uint32_t al = 0; al < n_alleles[s]; ++al) {
          o << "[";
          uint32_t slot = as_off[s] + al;
          for (uint32_t i = 0; i < pb_len[slot]; ++i) {
            o << std::min<uint32_t>(per_base[pb_off[slot] + i], 65535u);
            if (i + 1 < pb_len[slot]) o << ",";
          }
          o << "]";
          if (al + 1 < n_alleles[s]) o << ",";
        }
        o << "]";
        if (s + 1 < info.n_sites) o << ",";
      }
    }
    o << "]}\n";
    close_checked(o, join(cov_dir, "allele_base_coverage.json"));
  }
  {  // coverage::dump::grouped_allele_counts (grouped_allele_counts.cpp:51-110): uint16 wrap, arbitrary group ids
    std::vector<std::map<std::vector<int32_t>, uint32_t>> sites(info.n_sites);
    for (uint32_t s = 0; s < info.n_sites; ++s) {
      if (g_off[s] == 0xFFFFFFFFu) continue;
      uint32_t nm = (1u << n_alleles[s]) - 1u;
      for (uint32_t m = 0; m < nm; ++m) {
        uint32_t tot = grouped[g_off[s] + m];
        if (!tot) continue;
        std::vector<int32_t> ids;
        for (uint32_t al = 0; al < n_alleles[s]; ++al)
          if (((m + 1) >> al) & 1u) ids.push_back((int32_t)al);
        sites[s][ids] = tot & 0xFFFFu;
      }
    }
    for (int64_t i = 0; i + 1 < n_log;) {  // records worth +1 or +count (gmx.h: gmx_coverage_fetch_grouped_log)
      if (glog[i] == 0xFFFFFFFFu) {
        ++i;
        continue;
      }
      const uint32_t s = glog[i], n = glog[i + 1] & ~GMX_LOG_COUNTED;
      const int64_t head = (glog[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
      const uint64_t count = head == 4 ? ((uint64_t)glog[i + 2] | ((uint64_t)glog[i + 3] << 32)) : 1;
      std::vector<int32_t> ids(glog.begin() + i + head, glog.begin() + i + head + n);
      sites[s][ids] = (sites[s][ids] + count) & 0xFFFFu;
      i += head + n;
    }
    std::map<std::vector<int32_t>, uint64_t> group_id;
    std::vector<std::vector<int32_t>> by_id;
    for (auto &site : sites)
      for (auto &e : site)
        if (!group_id.count(e.first)) {
          group_id[e.first] = by_id.size();
          by_id.push_back(e.first);
        }
    std::ofstream o(join(cov_dir, "grouped_allele_counts_coverage.json"));
    o << "{\"grouped_allele_counts\":{\"allele_groups\":{";
    for (size_t g = 0; g < by_id.size(); ++g) {
      o << "\"" << g << "\":[";
      for (size_t j = 0; j < by_id[g].size(); ++j) o << by_id[g][j] << (j + 1 < by_id[g].size() ? "," : "");
      o << "]" << (g + 1 < by_id.size() ? "," : "");
    }
    o << "},\"site_counts\":[";
    for (uint32_t s = 0; s < info.n_sites; ++s) {
      std::map<uint64_t, uint32_t> by_gid;
      for (auto &e : sites[s]) by_gid[group_id[e.first]] = e.second;
      o << "{";
      size_t j = 0;
      for (auto &e : by_gid) o << "\"" << e.first << "\":" << e.second << (++j < by_gid.size() ? "," : "");
      o << "}" << (s + 1 < info.n_sites ? "," : "");
    }
    o << "]}}\n";
    close_checked(o, join(cov_dir, "grouped_allele_counts_coverage.json"));
  }
  write_read_stats(rs_path, rs);
  phase("coverage files written (writer thread)");
  });

  std::cout << std::endl;
  std::cout << "The following counts include generated reverse complement reads." << std::endl;
  std::cout << "Count all reads: " << st.all_reads_count << std::endl;
  std::cout << "Count skipped reads with no sequence: " << st.skipped_reads_count << std::endl;
  std::cout << "Count reads with >0 kmers not in kmer index: " << st.missing_kmer_reads_count << std::endl;
  std::cout << "Count reads with no exact mapping: " << st.no_extension_reads_count << std::endl;
  std::cout << "Count exact mapped reads: " << st.exact_mapped_reads_count << std::endl;
  std::cout << std::endl
            << "Timer report (wall seconds)" << std::endl
            << "  Load data (index build + upload): " << t_load << std::endl
            << "  Quasimap (parse + map " << total_reads << " reads): " << t_map << std::endl
            << "    feed: read " << g_feed.read_s << ", parse " << g_feed.parse_s << " (scan " << g_feed.scan_s << ", pack " << g_feed.pack_s << "), waiting for a free block " << g_feed.wait_slot_s
            << "; engine calls (beside the parser) " << g_feed.map_s << " (seeds " << g_feed.seeds_s << ")" << std::endl;
  // ---- infer (genotype.cpp:72-118): level genotyping on the host from the coverage just recorded -----------------
  std::cout << "====================" << std::endl << "Running genotyping" << std::endl;
  auto t_inf = clk::now();
  std::cout << "Running genotyping model" << std::endl;
  gmx_infer *inf = nullptr;
  GMX_CHECK(gmx_infer_run(ix, per_base.data(), grouped.data(), glog.data(), (uint64_t)n_log, rs.mean_cov_depth, rs.variance_cov_depth,
                          rs.mean_pb_error, ploidy == "haploid" ? 1 : 2, &inf));
  const std::string coords = join(gram_dir, "prg_coords.tsv");
  if (a.has("debug")) {  // site_gtyping_debug_info.txt (parameters.cpp:98; genotype.cpp:76-82)
    const std::string dbg_path = join(run_dir, "site_gtyping_debug_info.txt");
    std::cout << "Logging debug genotyping stats to " << dbg_path << std::endl;
    const int64_t n = gmx_infer_debug_text(inf, nullptr, 0);
    std::string text((size_t)std::max<int64_t>(n, 0) + 1, '\0');
    if (n > 0) gmx_infer_debug_text(inf, &text[0], (uint64_t)n + 1);
    std::ofstream o(dbg_path);
    o.write(text.data(), std::max<int64_t>(n, 0));
    close_checked(o, dbg_path);
  }
  phase("genotyping model run");
  // the three writers read the same results and write three different files: side by side
  std::cout << "Producing json vcf" << std::endl;
  std::cout << "Producing personalised reference" << std::endl;
  std::cout << "Producing vcf" << std::endl;
  const std::string desc = sample_id + " personalised reference made by gramtools genotype";
  int w_rc[3] = {GMX_OK, GMX_OK, GMX_OK};
  std::string w_err[3];
  std::thread w_json([&]() {
    if ((w_rc[0] = gmx_infer_write_json(inf, coords.c_str(), sample_id.c_str(), join(geno_dir, "genotyped.json").c_str()))) w_err[0] = gmx_last_error();
    phase("  jVCF written (writer thread)");
  });
  std::thread w_fasta([&]() {
    if ((w_rc[1] = gmx_infer_write_fasta(inf, coords.c_str(), desc.c_str(), join(geno_dir, "personalised_reference.fasta").c_str()))) w_err[1] = gmx_last_error();
    phase("  personalised reference written (writer thread)");
  });
  if ((w_rc[2] = gmx_infer_write_vcf(inf, coords.c_str(), sample_id.c_str(), join(geno_dir, "genotyped.vcf.gz").c_str()))) w_err[2] = gmx_last_error();
  phase("  VCF written");
  w_json.join();
  w_fasta.join();
  cov_writer.join();
  for (int i = 0; i < 3; ++i)
    if (w_rc[i] != GMX_OK) die("gram: " + w_err[i]);
  phase("jVCF, personalised reference and VCF written");
  // (60 k site records freed one by one: 57 ms — left to the process exit for the last sample)
  if (getenv("GMX_FULL_TEARDOWN") || sample_i + 1 < samples.size()) gmx_infer_destroy(inf);
  std::cout << "  Genotyping: " << std::chrono::duration<double>(clk::now() - t_inf).count() << std::endl;
  }  // ---- next sample ----
  // Every output file is closed. The engines' and the index's memory goes back with the process: freeing 0.7 GB of device
  // and host tables one by one and unloading the runtime took 65 + ~100 ms of a 1 s call (round 5, tools/cli_phases.sh).
  // GMX_FULL_TEARDOWN=1 destroys everything in order (leak checks).
  if (getenv("GMX_FULL_TEARDOWN")) {
    gmx_group_destroy(grp);
    gmx_index_destroy(ix);
    phase("engines and index destroyed");
    return 0;
  }
  phase("done (teardown left to the process exit)");
  std::cout.flush();
  std::cerr.flush();
  fflush(nullptr);
  _exit(0);
}

}  // namespace

// A C++ exception nobody caught — std::bad_alloc in a parser or writer thread, in a destructor's path — ends the run the way the
// reference's front-end expects a failed `gram` to end (genotype.py:106-107: non-zero exit code, message on the output), not
// with SIGABRT and a core file. Called by the runtime on whatever thread the exception died on; does not return.
static void gram_terminate() {
  const char *what = "unknown error";
  char buf[512];
  if (std::exception_ptr ep = std::current_exception()) {
    try {
      std::rethrow_exception(ep);
    } catch (std::bad_alloc const &) {
      what = "out of host memory";
    } catch (std::exception const &e) {
      snprintf(buf, sizeof(buf), "%s", e.what());
      what = buf;
    } catch (...) {
    }
  }
  char line[640];
  const int n = snprintf(line, sizeof(line), "gram: fatal: %s\n", what);
  if (n > 0) {
    (void)!write(1, line, (size_t)std::min<int>(n, (int)sizeof(line) - 1));
    (void)!write(2, line, (size_t)std::min<int>(n, (int)sizeof(line) - 1));
  }
  _exit(1);
}

static int gram_run(int argc, const char *const *argv);
int main(int argc, const char *const *argv) {
  std::set_terminate(gram_terminate);
  try {
    return gram_run(argc, argv);
  } catch (std::bad_alloc const &) {
    std::cout << "gram: out of host memory" << std::endl;
  } catch (std::exception const &e) {
    std::cout << "gram: " << e.what() << std::endl;
  }
  return 1;
}

static int gram_run(int argc, const char *const *argv) {
  // main.cpp:51-100: first positional token is the command; --help or no command prints help and exits 0
  std::string command;
  int cmd_at = -1;
  bool help = false;
  for (int i = 1; i < argc; ++i) {
    std::string t = argv[i];
    if (t == "--help") help = true;
    if (cmd_at < 0 && t.rfind("--", 0) != 0) {
      command = t;
      cmd_at = i;
      break;
    }
  }
  if (help || command.empty()) {
    std::cout << kGlobalHelp << std::endl;
    return 0;
  }
  if (command == "_parse_bench") {  // the parallel reads parser alone, timed (tools/parse_bench.sh)
    if (argc < cmd_at + 3) die("usage: gram _parse_bench FILE THREADS [REPEATS]");
    return run_parse_bench(argv[cmd_at + 1], atoi(argv[cmd_at + 2]), argc > cmd_at + 3 ? atoi(argv[cmd_at + 3]) : 3);
  }
  if (command == "_read_stats") {  // test hook: ReadStats::compute_base_error_rate on a reads file (test_read_stats.cpp:14-48)
    if (cmd_at + 1 >= argc) return 1;
    ReadStats rs;
    compute_base_error_rate(argv[cmd_at + 1], rs);
    std::cout.precision(9);
    std::cout << "num_bases=" << rs.num_bases_processed << " max_read_len=" << rs.max_read_length << " no_qual_reads=" << rs.no_qual_reads
              << " mean_pb_error=" << rs.mean_pb_error << std::endl;
    return 0;
  }
  if (command == "_gz_info") {  // test / bench hook: a gzip file through GzSource alone: bytes, CRC-32, how it was decompressed, rate
    if (cmd_at + 2 >= argc) return 1;
    try {
      const double t0 = now_s();
      gmx::GzSource src(argv[cmd_at + 1], (unsigned)std::max(1, atoi(argv[cmd_at + 2])));
      std::vector<char> buf((size_t)96 << 20);
      uint64_t total = 0;
      uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
      const bool with_crc = !getenv("GMX_GZ_INFO_NO_CRC");
      for (;;) {
        const size_t got = src.read(buf.data(), buf.size());
        if (got == 0) break;
        if (with_crc) crc = (uint32_t)crc32(crc, reinterpret_cast<const unsigned char *>(buf.data()), (uInt)got);
        total += got;
      }
      const double dt = now_s() - t0;
      std::cout << "bytes=" << total << " crc=" << crc << " pieces=" << src.parallel_pieces() << " bgzf_members=" << src.bgzf_members()
                << " stream_bytes=" << src.stream_bytes() << " seconds=" << dt << " MBps=" << (dt > 0 ? total / dt / 1e6 : 0) << std::endl;
      return 0;
    } catch (std::exception const &e) {
      std::cout << "gram: " << e.what() << std::endl;
      return 1;
    }
  }
  if (command == "_parse_check") {  // test hook: both read parsers on one file, no GPU (tests/test_gram_cli.py)
    if (cmd_at + 2 >= argc) return 1;
    return run_parse_check(argv[cmd_at + 1], atoi(argv[cmd_at + 2]));
  }
  if (command != "build" && command != "genotype" && command != "simulate") {
    std::cout << "Unrecognised command: " << command << std::endl;
    std::cout << kGlobalHelp << std::endl;
    return 1;
  }
  Args a = parse_sub(argc, argv, cmd_at + 1);
  if (command == "genotype") return run_genotype(a);
  if (command == "build") return run_build(a);
  std::cout << "The simulate command is not part of the MI355X quasimap engine." << std::endl;
  return 1;
}
