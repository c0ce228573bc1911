// gmx_pargz.h — a plain gzip (deflate) stream decompressed on all host threads (host only; used by gmx_gzsource.h).
//
// A deflate stream has no index: a block can refer to any of the 32 KB before it, so zlib inflates one stream on one
// thread (~0.4 GB/s of FASTQ text: 1.4 M reads/s, against a parser and a GPU that take a hundred times that). The way
// round it (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA sequences",
// 2019 — the published idea; no code of theirs is used or was available here):
//   1. cut the compressed stream into pieces; inside every piece but the first FIND a deflate block start by trying bit
//      positions: a dynamic-Huffman block header whose three code sets are complete, a whole block that decodes to text
//      (FASTQ is ASCII), and a plausible header behind it;
//   2. decode every piece from its block start to the next piece's on its own thread, into 16-bit symbols: a byte, or
//      "the byte j positions into the 32 KB window I cannot know yet" (back-references into the unknown window copy
//      such placeholders along);
//   3. in stream order, each piece's window is the resolved tail of the piece before it — 32 K symbols per piece, so the
//      serial part is tiny; then every piece resolves its placeholders and writes bytes, again on its own thread.
// Nothing is taken on trust: a piece counts only if the piece before it ends EXACTLY on the bit where it started; the
// member's CRC-32 and length are checked against the gzip trailer. Whatever cannot be handled this way (no dynamic block
// to find, speculation that does not line up) is decoded from the last verified bit position by zlib itself, primed with
// the known window (inflatePrime + inflateSetDictionary).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gmx {
namespace pargz {

constexpr uint32_t kWindow = 32768;
constexpr uint16_t kUnknown = 256;  // symbol kUnknown + j = byte j of the (yet unknown) window before the piece

// ---- bit reader over a byte range (LSB-first, as deflate packs its bits) ---------------------------------------------------
struct Bits {
  const uint8_t *base, *end;
  const uint8_t *p;
  uint64_t buf = 0;
  uint32_t cnt = 0;
  bool overrun = false;
  Bits(const uint8_t *b, const uint8_t *e, uint64_t bit_at) : base(b), end(e) { seek(bit_at); }
  void seek(uint64_t bit_at) {
    p = base + (bit_at >> 3);
    buf = 0;
    cnt = 0;
    overrun = false;
    refill();
    const uint32_t skip = (uint32_t)(bit_at & 7);
    buf >>= skip;
    cnt -= skip;
  }
  __attribute__((always_inline)) inline void refill() {
    if (end - p >= 8) {
      uint64_t w;
      memcpy(&w, p, 8);
      buf |= w << cnt;
      p += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56) {
        if (p < end) {
          buf |= (uint64_t)*p << cnt;
        } else if (p >= end + 8) {
          overrun = true;  // (zeros are fed in; the decoder notices through `overrun` or a position past the end)
          break;
        }
        ++p;
        cnt += 8;
      }
    }
  }
  __attribute__((always_inline)) inline uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  __attribute__((always_inline)) inline void skip(uint32_t n) {
    buf >>= n;
    cnt -= n;
  }
  __attribute__((always_inline)) inline uint32_t get(uint32_t n) {  // n <= 32; the caller has refilled
    const uint32_t v = peek(n);
    skip(n);
    return v;
  }
  uint64_t bit_pos() const { return (uint64_t)(p - base) * 8 - cnt; }
  bool past_end() const { return bit_pos() > (uint64_t)(end - base) * 8; }
};

// ---- canonical Huffman code: a fast table for codes of up to kFast bits, count/first arrays for the longer ones ------------
constexpr uint32_t kFast = 10;
struct Huff {
  uint16_t fast[1u << kFast];  // symbol << 4 | length; 0 = longer code (or none)
  uint16_t count[16], first_sym[16];
  uint16_t sorted[288];
  uint32_t max_len = 0;
  // lengths[n] -> tables. Returns false for an over-subscribed set, or an incomplete one unless `allow_single` and it is
  // one code of length 1 (what zlib's inflate_table accepts: RFC 1951 allows a lone distance code).
  bool build(const uint8_t *lengths, uint32_t n, bool allow_single) {
    memset(count, 0, sizeof(count));
    for (uint32_t i = 0; i < n; ++i) count[lengths[i]]++;
    if (count[0] == n) {  // no code at all
      max_len = 0;
      memset(fast, 0, sizeof(fast));
      return allow_single;
    }
    count[0] = 0;
    int32_t left = 1;
    max_len = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return false;
      if (count[l]) max_len = l;
    }
    if (left > 0 && !(allow_single && max_len == 1 && count[1] == 1)) return false;
    uint16_t offs[16];
    offs[1] = 0;
    for (uint32_t l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (uint32_t l = 1; l <= 15; ++l) first_sym[l] = offs[l];
    for (uint32_t i = 0; i < n; ++i)
      if (lengths[i]) sorted[offs[lengths[i]]++] = (uint16_t)i;
    memset(fast, 0, sizeof(fast));
    uint32_t code = 0, idx = 0;
    for (uint32_t l = 1; l <= std::min(max_len, kFast); ++l) {
      for (uint32_t k = 0; k < count[l]; ++k, ++idx, ++code) {
        uint32_t rev = 0;
        for (uint32_t b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
        const uint16_t e = (uint16_t)(sorted[idx] << 4 | l);
        for (uint32_t x = rev; x < (1u << kFast); x += 1u << l) fast[x] = e;
      }
      code <<= 1;
    }
    return true;
  }
  // decode one symbol; the caller has >= 15 bits in the buffer. Returns 0xFFFF on an invalid code.
  __attribute__((always_inline)) inline uint32_t decode(Bits &in) const {
    const uint16_t e = fast[in.peek(kFast)];
    if (e) {
      in.skip(e & 15u);
      return e >> 4;
    }
    // longer than kFast bits: canonical decode, one bit at a time (rare: long codes are the improbable symbols)
    uint32_t code = 0, first = 0, index = 0;
    uint64_t b = in.buf;
    for (uint32_t l = 1; l <= max_len; ++l) {
      code |= (uint32_t)(b & 1u);
      b >>= 1;
      const uint32_t c = count[l];
      if (code < first + c) {
        if (l <= kFast) break;  // (a code this short is in the fast table: an unused slot of an incomplete set)
        in.skip(l);
        return sorted[index + (code - first)];
      }
      index += c;
      first = (first + c) << 1;
      code <<= 1;
    }
    return 0xFFFFu;
  }
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct BlockCodes {
  Huff lit, dist;
};

// the header of a dynamic block (after its 3 header bits): HLIT, HDIST, HCLEN, the code-length code, the two code sets
inline bool read_dynamic_header(Bits &in, BlockCodes &bc) {
  in.refill();
  const uint32_t hlit = in.get(5) + 257, hdist = in.get(5) + 1, hclen = in.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t cl[19] = {0};
  in.refill();
  for (uint32_t i = 0; i < hclen; ++i) {
    if (in.cnt < 3) in.refill();
    cl[order[i]] = (uint8_t)in.get(3);
  }
  Huff clh;
  if (!clh.build(cl, 19, false)) return false;
  uint8_t lens[286 + 30];
  uint32_t i = 0;
  while (i < hlit + hdist) {
    in.refill();
    if (in.overrun) return false;
    const uint32_t s = clh.decode(in);
    if (s < 16) {
      lens[i++] = (uint8_t)s;
    } else if (s == 16) {
      if (i == 0) return false;
      uint32_t r = 3 + in.get(2);
      if (i + r > hlit + hdist) return false;
      const uint8_t v = lens[i - 1];
      while (r--) lens[i++] = v;
    } else if (s == 17 || s == 18) {
      uint32_t r = s == 17 ? 3 + in.get(3) : 11 + in.get(7);
      if (i + r > hlit + hdist) return false;
      while (r--) lens[i++] = 0;
    } else {
      return false;
    }
  }
  if (lens[256] == 0) return false;  // no end-of-block code
  if (!bc.lit.build(lens, hlit, false)) return false;
  return bc.dist.build(lens + hlit, hdist, true);
}

inline const BlockCodes &fixed_codes() {
  static const BlockCodes bc = [] {
    BlockCodes b;
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    b.lit.build(l, 288, false);
    uint8_t d[32];  // (RFC 1951 3.2.6: 32 five-bit distance codes; 30 and 31 never occur but are part of the code)
    for (int i = 0; i < 32; ++i) d[i] = 5;
    b.dist.build(d, 32, true);
    return b;
  }();
  return bc;
}

// Output of a piece: symbols behind a window of kWindow entries (placeholders, or the real bytes when they are known).
struct Symbols {
  // [0, kWindow): the window; the piece's output follows. A raw array: growing it must not zero-fill (a vector's resize
  // does, and that was a quarter of a round's time), only the used part is copied.
  struct Buf {
    uint16_t *p = nullptr;
    size_t cap = 0;
    ~Buf() { free(p); }
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    Buf(Buf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr, o.cap = 0; }
    uint16_t *data() { return p; }
    const uint16_t *data() const { return p; }
    uint16_t &operator[](size_t i) { return p[i]; }
    size_t size() const { return cap; }
    void grow(size_t want, size_t used) {
      uint16_t *q = static_cast<uint16_t *>(malloc(want * sizeof(uint16_t)));
      if (!q) throw std::bad_alloc();
      if (used) memcpy(q, p, used * sizeof(uint16_t));
      free(p);
      p = q;
      cap = want;
    }
    void release() {
      free(p);
      p = nullptr;
      cap = 0;
    }
  } v;
  size_t n = kWindow;
  // symbols of the window that stand for real history: all of it for a piece inside the stream (placeholders of the bytes
  // before it), only the bytes the member has produced so far for its first piece. A back-reference beyond them reaches
  // before the start of the member: zlib rejects that ("invalid distance too far back"), and so does decode_block.
  size_t hist = kWindow;
  void init_unknown(size_t expect = (size_t)1 << 20) {
    if (v.size() < kWindow + expect) v.grow(kWindow + expect, 0);
    for (uint32_t j = 0; j < kWindow; ++j) v[j] = (uint16_t)(kUnknown + j);
    n = kWindow;
    hist = kWindow;
  }
  void init_known(const uint8_t *win, size_t have, size_t expect = (size_t)1 << 20) {  // the last `have` (<= kWindow) bytes before the piece
    if (v.size() < kWindow + expect) v.grow(kWindow + expect, 0);
    for (uint32_t j = 0; j < kWindow; ++j) v[j] = 0;
    for (size_t j = 0; j < have; ++j) v[kWindow - have + j] = win[j];
    n = kWindow;
    hist = have;
  }
  inline void room(size_t more) {
    if (n + more > v.size()) v.grow(std::max(v.size() + v.size() / 2, n + more + (1u << 16)), n);
  }
  size_t out_size() const { return n - kWindow; }
};

enum class BlockEnd { Ok, Final, Bad };

// One block at the reader's position into `out`. `text_only`: literals must be text (the block finder's test).
// `max_out`: give up (Bad) beyond this many output symbols (finder: a wrong start may "decode" forever).
template <bool TEXT>
inline BlockEnd decode_block_t(Bits &in, Symbols &out, size_t max_out) {
  auto is_text = [](uint32_t c) { return (c >= 0x20 && c < 0x7F) || c == '\n' || c == '\r' || c == '\t'; };
  in.refill();
  if (in.overrun || in.past_end()) return BlockEnd::Bad;
  const uint32_t final = in.get(1), type = in.get(2);
  if (type == 3) return BlockEnd::Bad;
  if (type == 0) {  // stored: to the next byte boundary, LEN, ~LEN, bytes
    in.skip(in.cnt & 7u);
    in.refill();
    const uint32_t len = in.get(16), nlen = in.get(16);
    if ((len ^ nlen) != 0xFFFFu) return BlockEnd::Bad;
    out.room(len);
    for (uint32_t i = 0; i < len; ++i) {
      if (in.cnt < 8) in.refill();
      if (in.overrun) return BlockEnd::Bad;
      const uint32_t c = in.get(8);
      if (TEXT && !is_text(c)) return BlockEnd::Bad;
      out.v[out.n++] = (uint16_t)c;
    }
    if (in.past_end()) return BlockEnd::Bad;
    return final ? BlockEnd::Final : BlockEnd::Ok;
  }
  BlockCodes dyn;
  const BlockCodes *bc = &fixed_codes();
  if (type == 2) {
    if (!read_dynamic_header(in, dyn)) return BlockEnd::Bad;
    bc = &dyn;
  }
  const size_t limit = max_out ? kWindow + max_out : ~(size_t)0;
  const size_t hist = out.hist;
  const uint16_t *const lit_fast = bc->lit.fast;
  // (the bit reader as a local whose address never leaves this function: its words stay in registers; through the
  //  reference every step was a store and a reload)
  Bits lb = in;
  const BlockEnd result = [&]() __attribute__((always_inline)) -> BlockEnd {
#define in lb
  for (;;) {
    in.refill();
    if (in.overrun) return BlockEnd::Bad;
    if (out.n + 600 > out.v.size()) out.room(600);  // per refill: at most seven literals, or a match of 258 (+ 3 of the wide copy)
    uint16_t *const base = out.v.data();
    size_t n = out.n;
    // literals for as long as the bit buffer holds a whole code (FASTQ quality strings are runs of literals: one refill
    // serves four to seven of them)
    uint32_t s;
    for (;;) {
      const uint16_t e = lit_fast[in.peek(kFast)];
      if (e) {
        in.skip(e & 15u);
        s = e >> 4;
      } else {
        s = bc->lit.decode(in);
      }
      if (s >= 256) break;
      if (TEXT && !is_text(s)) return BlockEnd::Bad;
      base[n++] = (uint16_t)s;
      if (in.cnt < 15) break;
    }
    out.n = n;
    if (s < 256) {
      if (n > limit) return BlockEnd::Bad;
      continue;
    }
    if (s == 256) break;
    if (s > 285) return BlockEnd::Bad;  // (286, 287 and the invalid-code mark)
    s -= 257;
    if (in.cnt < 48) in.refill();  // length extra <= 5, distance code <= 15, distance extra <= 13
    const uint32_t len = kLenBase[s] + in.get(kLenExtra[s]);
    const uint32_t ds = bc->dist.decode(in);
    if (ds > 29) return BlockEnd::Bad;
    const uint32_t dist = kDistBase[ds] + in.get(kDistExtra[ds]);
    if (dist > n - (kWindow - hist)) return BlockEnd::Bad;  // too far back: before the member's first byte (Symbols::hist)
    uint16_t *dst = base + n;
    const uint16_t *src = dst - dist;
    if (dist >= 4) {  // four symbols per copy (the source chunk ends before the destination chunk starts); up to 3 symbols
      for (uint32_t i = 0; i < len; i += 4) memcpy(dst + i, src + i, 8);  // of overshoot land in the buffer's slack
    } else {
      for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];  // (overlapping on purpose: a run repeats its period)
    }
    out.n = n + len;
    if (out.n > limit) return BlockEnd::Bad;
  }
  if (in.past_end()) return BlockEnd::Bad;
  return final ? BlockEnd::Final : BlockEnd::Ok;
#undef in
  }();
  in = lb;
  return result;
}
inline BlockEnd decode_block(Bits &in, Symbols &out, bool text_only, size_t max_out) {
  return text_only ? decode_block_t<true>(in, out, max_out) : decode_block_t<false>(in, out, max_out);
}

// A deflate block start at or after bit `from` (below bit `upto`): a non-final dynamic block whose header is valid, that
// decodes to text, and behind which another valid block header stands. Returns the bit position, or ~0 if none.
inline uint64_t find_block(const uint8_t *base, const uint8_t *end, uint64_t from, uint64_t upto) {
  Symbols scratch;
  scratch.init_unknown();
  for (uint64_t at = from; at < upto; ++at) {
    // cheap rejections on the first 17 bits: BFINAL = 0, BTYPE = 10b, HLIT <= 29, HDIST <= 29
    const uint8_t *p = base + (at >> 3);
    if (end - p < 8) return ~0ull;
    uint64_t w;
    memcpy(&w, p, 8);
    w >>= at & 7;
    if ((w & 7u) != 4u) continue;  // bits: final (0), type low bit first: 2 = binary 10 -> bit1 = 0, bit2 = 1
    if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
    Bits in(base, end, at);
    scratch.n = kWindow;
    const BlockEnd r = decode_block(in, scratch, true, 1u << 22);
    if (r != BlockEnd::Ok) continue;
    if (scratch.out_size() < 1024) continue;  // (real blocks of a large text stream are tens of KB; tiny "blocks" are chance)
    // the block behind it must at least have a plausible header
    Bits nx(base, end, in.bit_pos());
    nx.refill();
    if (nx.overrun) continue;
    const uint32_t h = nx.get(3), type = h >> 1;
    if (type == 3) continue;
    if (type == 2) {
      BlockCodes bc;
      if (!read_dynamic_header(nx, bc)) continue;
    } else if (type == 0) {
      nx.skip(nx.cnt & 7u);
      nx.refill();
      const uint32_t len = nx.get(16), nlen = nx.get(16);
      if ((len ^ nlen) != 0xFFFFu) continue;
    }
    return at;
  }
  return ~0ull;
}

struct Piece {
  uint64_t start_bit = 0, stop_bit = 0, end_bit = 0;  // from, where the next piece starts (~0: open end), where decoding ended
  bool known_window = false, ok = false, final = false;
  bool capped = false;  // gave up because the output passed the per-piece bound (not damage: zlib takes the member over)
  Symbols sym;
  uint32_t crc = 0;
  size_t out_at = 0;
};

// Decodes from start_bit, block by block, until a block ends at or beyond stop_bit (or the final block ends, or, with an
// open end, once `soft_limit_bit` is passed).
// `max_out`: bound on the piece's output symbols (two bytes each in memory) — a crafted stream inflates 1000-fold, and every
// thread would grow its buffer without limit; a piece that passes the bound is marked `capped`.
inline void decode_piece(const uint8_t *base, const uint8_t *end, Piece &pc, uint64_t soft_limit_bit, size_t max_out) {
  Bits in(base, end, pc.start_bit);
  pc.ok = false;
  pc.capped = false;
  for (;;) {
    const BlockEnd r = decode_block(in, pc.sym, false, max_out);
    if (r == BlockEnd::Bad) {
      pc.capped = max_out != 0 && pc.sym.out_size() > max_out;
      return;
    }
    pc.end_bit = in.bit_pos();
    if (r == BlockEnd::Final) {
      pc.final = true;
      pc.ok = true;
      return;
    }
    if (pc.stop_bit != ~0ull ? pc.end_bit >= pc.stop_bit : pc.end_bit >= soft_limit_bit) {
      pc.ok = true;
      return;
    }
  }
}

}  // namespace pargz
}  // namespace gmx
