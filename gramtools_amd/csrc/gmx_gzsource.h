// gmx_gzsource.h — gzip input for the reads feed of `gram genotype` (host only; gram_main.cpp).
//
// The reference reads its reads files through a zlib-backed reader that takes gzip transparently
// (libgramtools/include/sequence_read/seqread.hpp:94-180, seq_file.h); real FASTQ is almost always gzipped. One zlib stream
// inflates at ~0.4 GB/s of text — 1.4 M reads/s, two orders below what the parser and the GPU take — so this source
// decompresses on all host threads where the container format allows it:
//   * BGZF (bgzip, htslib, Illumina's BCL Convert: gzip members of <= 64 KB that carry their compressed size in a `BC`
//     extra field): the member table is walked without inflating anything, members are inflated side by side, each
//     straight to its place in the caller's buffer (the trailer's ISIZE says where), CRC-32 checked per member;
//   * a plain gzip stream: deflate blocks found by speculation inside the stream and decoded on all threads with the
//     32 KB of history each piece cannot know yet carried as placeholders (gmx_pargz.h), or — small files, streams the
//     speculation cannot handle — one zlib stream over the mapped file; members may follow each other in any mix.
// Every failure (truncated member, CRC or length mismatch, damaged stream) throws: a damaged file must not pass for the
// end of the reads.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gmx_crc32.h"
#include "gmx_pargz.h"

namespace gmx {

class GzSource {
 public:
  GzSource(const std::string &path, unsigned threads) : path_(path), threads_(threads ? threads : 1) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open " + path);
    struct stat sb;
    if (fstat(fd_, &sb) != 0) fail("cannot stat");
    size_ = (size_t)sb.st_size;
    if (size_) {
      void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd_, 0);
      if (m == MAP_FAILED) fail("cannot map");
      in_ = static_cast<const unsigned char *>(m);
      madvise(m, size_, MADV_SEQUENTIAL);
    }
    if (const char *e = getenv("GMX_PARGZ_MIN")) par_min_ = (size_t)atoll(e);      // compressed bytes left for the parallel decoder to engage
    if (const char *e = getenv("GMX_PARGZ_CHUNK")) par_chunk_ = std::max<size_t>(1024, (size_t)atoll(e));  // compressed bytes per piece
    if (const char *e = getenv("GMX_PARGZ")) par_on_ = atoi(e) != 0;
  }
  GzSource(const GzSource &) = delete;
  GzSource &operator=(const GzSource &) = delete;
  ~GzSource() {
    end_stream();
    if (in_) munmap(const_cast<unsigned char *>(in_), size_);
    if (fd_ >= 0) close(fd_);
  }

  // Up to `want` decompressed bytes to dst; fewer only at the end of the data (0: nothing left).
  size_t read(char *dst, size_t want) {
    size_t got = 0;
    while (got < want) {
      if (carry_at_ < carry_.size()) {  // the rest of a member that did not fit the caller's last request
        const size_t n = std::min(want - got, carry_.size() - carry_at_);
        memcpy(dst + got, carry_.data() + carry_at_, n);
        carry_at_ += n;
        got += n;
        continue;
      }
      if (par_active_) {  // a plain member being decoded on all threads: its next round of pieces, straight to the caller
        got += par_round(dst + got, want - got);  // (what the caller did not ask for waits in the carry buffer)
        continue;
      }
      if (!streaming_ && pos_ >= size_) break;
      if (!streaming_ && bgzf_at(pos_)) {
        got += read_bgzf(dst + got, want - got);
        continue;
      }
      const size_t n = read_stream(dst + got, want - got);
      got += n;
      if (n == 0 && !streaming_ && pos_ >= size_) break;
    }
    return got;
  }
  bool at_end() const { return !streaming_ && !par_active_ && pos_ >= size_ && carry_at_ >= carry_.size(); }
  uint64_t parallel_pieces() const { return n_par_pieces_; }
  // what the file was, for the feed trace: members inflated side by side / bytes through the single zlib stream
  uint64_t bgzf_members() const { return n_bgzf_; }
  uint64_t stream_bytes() const { return n_stream_bytes_; }

 private:
  struct Member {
    size_t data, clen;  // deflate data in the file
    uint32_t crc, isize;
    size_t out;         // offset in this batch's output
  };

  [[noreturn]] void fail(const std::string &what) const { throw std::runtime_error(path_ + ": " + what); }
  static uint32_t le16(const unsigned char *p) { return p[0] | (uint32_t)p[1] << 8; }
  static uint32_t le32(const unsigned char *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

  // a BGZF member at `at`? (SAM spec §4.1: FLG.FEXTRA, an extra subfield 'B' 'C' of two bytes = total member size - 1)
  bool bgzf_at(size_t at, Member *m = nullptr, size_t *next = nullptr) const {
    if (at + 18 > size_) return false;
    const unsigned char *p = in_ + at;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return false;  // (exactly FEXTRA, as bgzip writes it)
    const uint32_t xlen = le16(p + 10);
    if (at + 12 + xlen > size_) return false;
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
    if (!found || bsize < 12 + xlen + 8 || at + bsize > size_) return false;
    if (m) {
      m->data = at + 12 + xlen;
      m->clen = bsize - 12 - xlen - 8;
      m->crc = le32(p + bsize - 8);
      m->isize = le32(p + bsize - 4);
    }
    if (next) *next = at + bsize;
    return true;
  }

  void inflate_member(const Member &m, char *dst) const {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) fail("zlib: inflateInit2 failed");
    zs.next_in = const_cast<unsigned char *>(in_ + m.data);
    zs.avail_in = (uInt)m.clen;
    zs.next_out = reinterpret_cast<unsigned char *>(dst);
    zs.avail_out = m.isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 && zs.avail_in == 0;
    inflateEnd(&zs);
    if (!ok) fail("damaged BGZF member at byte " + std::to_string(m.data));
    if (m.isize && crc32_fast((uint32_t)crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(dst), m.isize) != m.crc)
      fail("CRC mismatch in the BGZF member at byte " + std::to_string(m.data));
  }

  size_t read_bgzf(char *dst, size_t want) {
    std::vector<Member> ms;
    size_t out = 0, at = pos_;
    Member m;
    size_t next;
    while (at < size_ && bgzf_at(at, &m, &next)) {
      if (out + m.isize > want) break;
      m.out = out;
      out += m.isize;
      ms.push_back(m);
      at = next;
      if (ms.size() >= (1u << 20)) break;
    }
    if (ms.empty()) {  // the next member alone is more than the caller asked for: through the carry buffer
      if (!bgzf_at(pos_, &m, &next)) fail("damaged BGZF member at byte " + std::to_string(pos_));
      carry_.resize(m.isize);
      carry_at_ = 0;
      inflate_member(m, carry_.data());
      pos_ = next;
      ++n_bgzf_;
      return 0;
    }
    std::atomic<size_t> next_i{0};
    std::string error;
    std::mutex mu;
    auto work = [&]() {
      try {
        for (;;) {
          const size_t i0 = next_i.fetch_add(16);  // a few members per grab: they are 64 KB at most
          if (i0 >= ms.size()) break;
          for (size_t i = i0; i < std::min(ms.size(), i0 + 16); ++i) inflate_member(ms[i], dst + ms[i].out);
        }
      } catch (std::exception const &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (error.empty()) error = e.what();
      }
    };
    const unsigned T = (unsigned)std::min<size_t>(threads_, (ms.size() + 15) / 16);
    if (T <= 1) {
      work();
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < T; ++t) pool.emplace_back(work);
      for (auto &t : pool) t.join();
    }
    if (!error.empty()) throw std::runtime_error(error);
    pos_ = at;
    n_bgzf_ += ms.size();
    return out;
  }

  void end_stream() {
    if (streaming_) inflateEnd(&zs_);
    streaming_ = false;
  }

  // One zlib stream over the mapped file from pos_: gzip members one after the other (as gzread treats them), until `want`
  // bytes are out, the data ends, or a BGZF member comes up at a member boundary.
  size_t read_stream(char *dst, size_t want) {
    if (!streaming_) {
      if (pos_ + 2 > size_ || in_[pos_] != 0x1f || in_[pos_ + 1] != 0x8b) {
        if (pos_ == 0) fail("not a gzip file");
        pos_ = size_;  // trailing bytes that are no gzip member: ignored, as zlib's gzread does
        return 0;
      }
      if (par_on_ && threads_ >= 2 && size_ - pos_ >= par_min_ && par_begin()) return 0;
      memset(&zs_, 0, sizeof(zs_));
      if (inflateInit2(&zs_, 15 + 16) != Z_OK) fail("zlib: inflateInit2 failed");
      streaming_ = true;
      raw_mode_ = false;
    }
    size_t got = 0;
    while (got < want) {
      const size_t in_chunk = std::min<size_t>(size_ - pos_, 1u << 30), out_chunk = std::min<size_t>(want - got, 1u << 30);
      zs_.next_in = const_cast<unsigned char *>(in_ + pos_);
      zs_.avail_in = (uInt)in_chunk;
      zs_.next_out = reinterpret_cast<unsigned char *>(dst + got);
      zs_.avail_out = (uInt)out_chunk;
      const int rc = inflate(&zs_, Z_NO_FLUSH);
      pos_ += in_chunk - zs_.avail_in;
      const size_t n = out_chunk - zs_.avail_out;
      got += n;
      n_stream_bytes_ += n;
      if (raw_mode_ && n) {
        par_crc_ = (uint32_t)crc32(par_crc_, reinterpret_cast<const unsigned char *>(dst + got - n), (uInt)n);
        par_len_ += n;
      }
      if (rc == Z_STREAM_END && raw_mode_) {  // the deflate data zlib took over from the parallel decoder ends: the trailer is ours
        end_stream();
        raw_mode_ = false;
        check_trailer();
        break;
      }
      if (rc == Z_STREAM_END) {  // end of a member: another one, a BGZF run, trailing bytes, or the end of the file
        if (pos_ + 2 <= size_ && in_[pos_] == 0x1f && in_[pos_ + 1] == 0x8b && !bgzf_at(pos_)) {
          if (inflateReset(&zs_) != Z_OK) fail("zlib: inflateReset failed");
          continue;
        }
        end_stream();
        if (!(pos_ + 2 <= size_ && in_[pos_] == 0x1f && in_[pos_ + 1] == 0x8b)) pos_ = size_;
        break;
      }
      if (rc != Z_OK && rc != Z_BUF_ERROR) fail(std::string("damaged gzip stream: ") + (zs_.msg ? zs_.msg : "zlib error"));
      if (pos_ >= size_ && zs_.avail_out != 0) fail("truncated gzip stream (unexpected end of file)");
      if (rc == Z_BUF_ERROR && n == 0 && in_chunk - zs_.avail_in == 0) fail("damaged gzip stream (no progress)");
    }
    return got;
  }

  std::string path_;
  unsigned threads_;
  int fd_ = -1;
  const unsigned char *in_ = nullptr;
  size_t size_ = 0, pos_ = 0;
  bool streaming_ = false;
  z_stream zs_;
  struct RawBuf {  // bytes that are always written before they are read: no zero-fill, no copy when it grows
    char *p = nullptr;
    size_t n = 0, cap = 0;
    ~RawBuf() { free(p); }
    char *data() { return p; }
    size_t size() const { return n; }
    void resize(size_t want) {
      if (want > cap) {
        free(p);
        p = static_cast<char *>(malloc(want + want / 8 + 64));
        if (!p) throw std::bad_alloc();
        cap = want + want / 8 + 64;
      }
      n = want;
    }
  } carry_;
  size_t carry_at_ = 0;
  uint64_t n_bgzf_ = 0, n_stream_bytes_ = 0, n_par_pieces_ = 0;
  // ---- a plain member on all threads (gmx_pargz.h) ----
  bool par_on_ = true, par_active_ = false, raw_mode_ = false;
  size_t par_min_ = (size_t)8 << 20, par_chunk_ = (size_t)2 << 20;
  uint64_t par_bit_ = 0;              // verified position in the deflate data (bit offset in the file)
  std::vector<uint8_t> par_window_;   // the last <= 32 KB of the member's output so far
  uint32_t par_crc_ = 0;
  uint64_t par_len_ = 0;
  int par_lone_rounds_ = 0;           // rounds in a row in which only the first piece counted

  // the member at pos_: gzip header (RFC 1952) -> where its deflate data starts. False: not a header this decoder takes.
  bool par_begin() {
    const unsigned char *p = in_ + pos_;
    const size_t left = size_ - pos_;
    if (left < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return false;
    const unsigned flg = p[3];
    size_t at = 10;
    if (flg & 4) {
      if (at + 2 > left) return false;
      at += 2 + le16(p + at);
    }
    for (int f = 0; f < 2; ++f)
      if (flg & (f ? 16 : 8)) {
        while (at < left && p[at]) ++at;
        ++at;
      }
    if (flg & 2) at += 2;
    if (at + 8 >= left) return false;
    par_bit_ = (uint64_t)(pos_ + at) * 8;
    par_window_.clear();
    par_crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
    par_len_ = 0;
    par_lone_rounds_ = 0;
    par_active_ = true;
    return true;
  }

  void check_trailer() {  // pos_ at the member's 8-byte trailer: CRC-32 and length (mod 2^32) of what was decoded
    if (pos_ + 8 > size_) fail("truncated gzip stream (no trailer)");
    if (le32(in_ + pos_) != par_crc_) fail("gzip CRC mismatch");
    if (le32(in_ + pos_ + 4) != (uint32_t)par_len_) fail("gzip length mismatch");
    pos_ += 8;
    if (!(pos_ + 2 <= size_ && in_[pos_] == 0x1f && in_[pos_ + 1] == 0x8b)) pos_ = size_;  // trailing bytes: ignored, as gzread does
  }

  // hands the rest of the member to zlib: a raw deflate stream from bit par_bit_ with the known window
  void par_to_zlib() {
    par_active_ = false;
    memset(&zs_, 0, sizeof(zs_));
    if (inflateInit2(&zs_, -15) != Z_OK) fail("zlib: inflateInit2 failed");
    streaming_ = true;
    raw_mode_ = true;
    pos_ = (size_t)(par_bit_ >> 3);
    const unsigned k = (unsigned)(par_bit_ & 7);
    if (k) {
      if (inflatePrime(&zs_, 8 - (int)k, in_[pos_] >> k) != Z_OK) fail("zlib: inflatePrime failed");
      ++pos_;
    }
    if (!par_window_.empty() && inflateSetDictionary(&zs_, par_window_.data(), (uInt)par_window_.size()) != Z_OK)
      fail("zlib: inflateSetDictionary failed");
  }

  template <class F>
  void on_threads(size_t n_items, F fn) {
    std::atomic<size_t> next{0};
    std::string error;
    std::mutex mu;
    auto work = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n_items) break;
        try {
          fn(i);
        } catch (std::exception const &e) {
          std::lock_guard<std::mutex> lk(mu);
          if (error.empty()) error = e.what();
        }
      }
    };
    const unsigned T = (unsigned)std::min<size_t>(threads_, n_items);
    if (T <= 1) {
      work();
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < T; ++t) pool.emplace_back(work);
      for (auto &t : pool) t.join();
    }
    if (!error.empty()) throw std::runtime_error(error);
  }

  // One round: up to `threads_` pieces of par_chunk_ compressed bytes from par_bit_ on, decoded side by side, verified in
  // stream order, written to the carry buffer as bytes.
  size_t par_round(char *dst, size_t want) {
    using namespace pargz;
    static const bool trace = getenv("GMX_PARGZ_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    const uint8_t *base = in_, *end = in_ + size_;
    const uint64_t byte0 = par_bit_ >> 3;
    const size_t T = std::max<size_t>(2, threads_);
    // piece starts: the verified position, then a block found in each following chunk
    std::vector<uint64_t> starts(T, ~0ull);
    starts[0] = par_bit_;
    on_threads(T - 1, [&](size_t i) {
      const uint64_t from = (byte0 + (i + 1) * par_chunk_) * 8;
      if (from + 64 >= (uint64_t)size_ * 8) return;
      starts[i + 1] = find_block(base, end, from, std::min<uint64_t>(from + par_chunk_ * 8, (uint64_t)size_ * 8));
    });
    const double t1 = now();
    std::vector<Piece> pcs;
    for (size_t i = 0; i < T; ++i) {
      if (starts[i] == ~0ull) continue;
      pcs.emplace_back();
      pcs.back().start_bit = starts[i];
      pcs.back().stop_bit = ~0ull;
      if (pcs.size() > 1) pcs[pcs.size() - 2].stop_bit = starts[i];
    }
    pcs[0].known_window = true;
    const uint64_t soft = (byte0 + T * par_chunk_) * 8;
    on_threads(pcs.size(), [&](size_t j) {
      Piece &pc = pcs[j];
      const size_t expect = par_chunk_ * 6;  // (FASTQ inflates 3-5x; the buffer grows if that is not enough)
      if (pc.known_window) pc.sym.init_known(par_window_.data(), par_window_.size(), expect);
      else pc.sym.init_unknown(expect);
      decode_piece(base, end, pc, soft, par_chunk_ * 32);  // (FASTQ inflates 3-5x; 32x covers any text worth the name)
    });
    const double t2 = now();
    if (!pcs[0].ok && pcs[0].capped) {  // not damage: more output per compressed byte than the pieces may hold: zlib from here
      par_to_zlib();
      return 0;
    }
    if (!pcs[0].ok) fail("damaged gzip stream (deflate data at byte " + std::to_string(byte0) + ")");
    size_t n_valid = 1;
    while (n_valid < pcs.size() && !pcs[n_valid - 1].final && pcs[n_valid].ok && pcs[n_valid - 1].end_bit == pcs[n_valid].start_bit) ++n_valid;
    // windows in stream order: the resolved tail of every piece is the window of the next (32 K symbols each: the serial part)
    std::vector<std::vector<uint8_t>> win(n_valid + 1);
    win[0].assign(kWindow, 0);
    std::copy(par_window_.begin(), par_window_.end(), win[0].end() - par_window_.size());
    size_t total = 0;
    for (size_t j = 0; j < n_valid; ++j) {
      Piece &pc = pcs[j];
      pc.out_at = total;
      const size_t n = pc.sym.out_size();
      total += n;
      win[j + 1].resize(kWindow);
      const size_t take = std::min<size_t>(n, kWindow);
      if (take < kWindow) std::copy(win[j].begin() + take, win[j].end(), win[j + 1].begin());
      const uint16_t *tail = pc.sym.v.data() + pc.sym.n - take;
      for (size_t i = 0; i < take; ++i) {
        const uint16_t s = tail[i];
        win[j + 1][kWindow - take + i] = s < kUnknown ? (uint8_t)s : win[j][s - kUnknown];
      }
    }
    const double t3 = now();
    // bytes: the first `direct` of them to the caller's buffer, the rest to the carry buffer
    const size_t direct = std::min(total, want);
    carry_.resize(total - direct);
    carry_at_ = 0;
    const double t4 = now();
    on_threads(n_valid, [&](size_t j) {
      Piece &pc = pcs[j];
      const uint16_t *src = pc.sym.v.data() + kWindow;
      const size_t n = pc.sym.out_size();
      const uint8_t *w = win[j].data();
      uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
      // the piece's bytes [out_at, out_at + n) of the round: below `direct` to dst, from there on to the carry buffer
      for (int part = 0; part < 2; ++part) {
        const size_t lo = part == 0 ? pc.out_at : std::max(pc.out_at, direct);
        const size_t hi = part == 0 ? std::min(pc.out_at + n, direct) : pc.out_at + n;
        if (lo >= hi) continue;
        uint8_t *out = reinterpret_cast<uint8_t *>(part == 0 ? dst + lo : carry_.data() + (lo - direct));
        const uint16_t *from = src + (lo - pc.out_at);
        const size_t m = hi - lo;
        for (size_t i = 0; i < m; ++i) {
          const uint16_t sy = from[i];
          out[i] = sy < kUnknown ? (uint8_t)sy : w[sy - kUnknown];
        }
        c = crc32_fast(c, out, m);
      }
      pc.crc = c;
      pc.sym.v.release();
    });
    for (size_t j = 0; j < n_valid; ++j) {
      const size_t n = (j + 1 < n_valid ? pcs[j + 1].out_at : total) - pcs[j].out_at;
      par_crc_ = (uint32_t)crc32_combine(par_crc_, pcs[j].crc, (z_off_t)n);
      par_len_ += n;
    }
    n_par_pieces_ += n_valid;
    if (trace)
      fprintf(stderr, "[pargz] round: %zu of %zu pieces, %zu bytes: find %.1f ms, decode %.1f ms, windows %.1f ms, resize %.1f ms, bytes+crc %.1f ms\n",
              n_valid, pcs.size(), total, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (now() - t4) * 1e3);
    const Piece &last = pcs[n_valid - 1];
    par_bit_ = last.end_bit;
    // the window as the known bytes before par_bit_: at most 32 KB, fewer near the member's start
    const uint64_t known = std::min<uint64_t>(par_len_, kWindow);
    par_window_.assign(win[n_valid].end() - known, win[n_valid].end());
    if (last.final) {
      par_active_ = false;
      pos_ = (size_t)((par_bit_ + 7) >> 3);
      check_trailer();
      return direct;
    }
    // speculation that keeps failing (no dynamic block to find, binary data): zlib takes the rest of the member
    par_lone_rounds_ = n_valid == 1 && pcs.size() == 1 && (uint64_t)size_ - (par_bit_ >> 3) > 2 * par_chunk_ ? par_lone_rounds_ + 1 : 0;
    if (par_lone_rounds_ >= 2) par_to_zlib();
    return direct;
  }
};

}  // namespace gmx
