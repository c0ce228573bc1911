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
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

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
  bool at_end() const { return !streaming_ && pos_ >= size_ && carry_at_ >= carry_.size(); }
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
    if (m.isize && (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(dst), m.isize) != m.crc)
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
      memset(&zs_, 0, sizeof(zs_));
      if (inflateInit2(&zs_, 15 + 16) != Z_OK) fail("zlib: inflateInit2 failed");
      streaming_ = true;
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
  std::vector<char> carry_;
  size_t carry_at_ = 0;
  uint64_t n_bgzf_ = 0, n_stream_bytes_ = 0;
};

}  // namespace gmx
