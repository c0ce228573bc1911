// gmx_ingest.hip — reads files decoded ON the device (round 5; SURVEY.md §8f-3, VERDICT round 4 "missing" #4).
//
// The reference reads `.fastq.gz` through zlib / htslib on one host thread (libgramtools/include/sequence_read/seqread.hpp:94-180,
// src/genotype/quasimap/quasimap.cpp:65-76). The mapping kernels take 1 M reads in 0.4 ms; sixteen host cores inflate BGZF at
// 32-48 M reads/s. Here the compressed members go over PCIe as they lie in the file (a quarter of the text) and the GPU does
// the rest:
//
//   gmx_inflate_kernel     one wavefront per BGZF member (<= 64 KB of text, independent deflate streams, SAM spec §4.1): the
//                          Huffman tables of the member's blocks and a 4 KB window of its output live in LDS; the bit stream
//                          is decoded by the wave as ONE scalar thread of control (every value wave-uniform: the compiler
//                          keeps the bit buffer in SGPRs), match copies and line flushes use the 64 lanes; CRC-32 of the
//                          member's text by the same wave (64 slices, GF(2) combination), compared with the trailer's.
//   gmx_nl_count/_mark     newline positions of the chunk's text (tile counts -> scan -> positions)
//   gmx_records_kernel     four-line records: checks '@' / '+' / equal lengths, read lengths, one length for all or not
//   gmx_fq_pack_kernel     32 letters -> one pair of bit planes, in the layout gmx_pack_reads / the host parser produce
//                          (include/gmx.h), unencodable reads flagged in skip[] (encode_dna_bases, common/utils.cpp:73-92)
//
// A chunk's incomplete last record is carried into the next chunk on the device. Anything irregular (not four-line FASTQ, a
// damaged member, a CRC mismatch) is reported in gmx_ingest_result::status and decided by the caller (`gram` re-inflates the
// chunk with zlib and hands the text to gmx_ingest_submit_text, so a defect of this decoder cannot lose or invent reads).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gmx_internal.h"

#define ING_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gmx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
      return GMX_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

namespace {

// ------------------------------------------------------------------------------------------------------------------
// device-side state of one chunk (one slot): written by the kernels, read by the next kernel and, at the end, by the host
// ------------------------------------------------------------------------------------------------------------------
struct IngestState {
  uint32_t text_start;     // first byte of the chunk's text in the slot's text buffer (carry of the chunk before included)
  uint32_t text_len;       // bytes from there
  uint32_t n_lines;        // complete lines (a last line without '\n' counts when the chunk is the file's last)
  uint32_t n_reads;
  uint32_t min_len, max_len;
  uint32_t flags;          // GMX_INGEST_* status bits
  uint32_t bad_member;     // first member that failed (index within the chunk)
  uint32_t consumed;       // bytes of text the records take
  uint32_t tail_len;       // text_len - consumed: carried into the next chunk
  uint32_t any_skip;
  uint32_t uniform_len;
  unsigned long long n_bases;
  unsigned long long n_pairs;
  unsigned long long sub_pairs[16];  // pair index of read i * 2^20 (offsets form): where a launch of <= 2^20 reads starts
  uint32_t final_chunk;
  uint32_t pad;
};

struct IngestInflateStatus {  // written by gmx_inflate_kernel (a stream of its own), merged into the chunk's state by gmx_layout_kernel
  uint32_t flags, bad_member;
};

struct IngestMember {
  uint32_t in_off, in_len;   // deflate data within the chunk's compressed bytes
  uint32_t out_off, isize;   // its text within the chunk's text (from the first member's first byte)
  uint32_t crc, pad;
};

#define ING_CARRY_MAX (1u << 20)  // bytes of an incomplete last record that can be carried (a record longer than this: irregular)
#ifndef ING_LIT_ROOT
#define ING_LIT_ROOT 10
#endif
#ifndef ING_DIST_ROOT
#define ING_DIST_ROOT 7
#endif
#ifndef ING_RING
#define ING_RING 2048u
#endif
#ifndef ING_WAVES
#define ING_WAVES 5
#endif
#define ING_RING_MASK (ING_RING - 1u)
#define ING_NEAR_MAX (ING_RING - 320u)  // distances up to this are served from the LDS window

// table entry (32 bits): bits 0-3 the bits its code takes (0: no code here), bit 31 ING_RARE, and
//   literals     (literal/length table) bits 6-7 how many (ing_fuse: up to three literals whose codes fit the root bits together
//                take ONE look-up; 0: the entry that does nothing), bits 8-31 the bytes, first one lowest; bits 4-5 clear
//   length       (literal/length table) bit 5 (ING_LENGTH), bits 6-14 base value, bits 16-18 extra bits, bits 23-27 code + extra
//                bits: the entry itself is the operand of s_bfe_u32 that takes the extra bits out of the stream (offset = bits
//                0-4, width = bits 16-22), and one shift drops code and extra bits together
//   distance     (distance table) bit 4, bits 8-11 extra bits, bits 12-27 base value
//   code length  bits 8-12 the symbol
//   ING_RARE     bits 4-5 say which: ING_T_EOB, ING_T_LONG (a code longer than the table's bits), else no code
#define ING_T_LIT 0u
#define ING_T_BASE 1u
#define ING_T_EOB 2u
#define ING_T_LONG 3u
#define ING_TYPE(e) (((e) >> 4) & 3u)  // of an ING_RARE entry
#define ING_LENGTH 32u
#define ING_IS_LIT(e) (((e) & (ING_RARE | 48u)) == 0)
#define ING_RARE 0x80000000u  // set in every entry that is not a literal or a length / distance base: no code, end of block, longer code
                               // (a third fused literal is below 0x80, so that its byte does not reach the bit)

struct WaveLds {  // 8 KB: twenty wavefronts per CU
  uint32_t lit[1u << ING_LIT_ROOT];    // literal/length codes of up to ING_LIT_ROOT bits, by the next bits of the stream (4 KB); the CRC table afterwards
  uint64_t dist[1u << ING_DIST_ROOT];  // distance codes (1 KB, entries of two words); as 32-bit entries the code-length code while a dynamic block's lengths are read
  alignas(16) uint8_t ring[ING_RING];  // the last 2 KB of the member's text
  uint16_t lit_sorted[288];            // symbols by (code length, symbol): codes longer than the root are decoded bit by bit
  uint16_t dist_sorted[32];
  uint16_t lit_count[16], dist_count[16];
  union {
    uint8_t lens[320];                 // code lengths of the block being set up
    uint8_t dummy[64];                 // in the symbol loop (lens[] is dead then): where a lane with nothing to write writes, no lane-divergent branch
  };
};
static_assert(sizeof(WaveLds) <= 8192, "twenty wavefronts per CU");

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the low N bits of a wave-uniform word, computed on the vector unit (the table index of a look-up: the scalar unit is what
// bounds the decoder, the address has to be in a vector register anyway)
template <uint32_t N>
__device__ __forceinline__ uint32_t ing_vpeek(uint32_t v) {
  uint32_t r;
  asm("v_bfe_u32 %0, %1, 0, %2" : "=v"(r) : "s"(v), "n"(N));
  return r;
}
// bits [offset, offset + width) of v, offset = op bits 0-4, width = op bits 16-22 (the other bits of op are ignored): one scalar
// instruction where shift-and-mask by run-time amounts takes three
__device__ __forceinline__ uint32_t ing_sbfe(uint32_t v, uint32_t op) {
  uint32_t r;
  asm("s_bfe_u32 %0, %1, %2" : "=s"(r) : "s"(v), "s"(op) : "scc");
  return r;
}


// The bit reader. Everything but `cur` is wave-uniform. The compressed words are fetched 64 at a time, one per lane (a
// coalesced load; a single word per refill left the wave waiting a memory round trip every 32 bits) and taken, lane by
// lane, with v_readlane. No load stays in flight between refills: a prefetched next set made every iteration of the
// symbol loop wait for vector memory (its register copies), i.e. for every line of text flushed so far.
struct Bits {
  const uint32_t *w;
  uint32_t base;   // index of the word lane 0 holds in `cur`
  uint32_t rel;    // lane of `cur` that holds the next word to take (64: the next set is due)
  uint32_t cur;
  uint64_t buf;
  uint32_t cnt;
  uint32_t limit;  // words at and beyond this index are not the member's: never loaded, read as zero (set before start())
  __device__ __forceinline__ uint32_t fetch(uint32_t idx) const { return idx < limit ? w[idx] : 0u; }
  __device__ __forceinline__ void start(const uint32_t *words, uint32_t byte_off) {
    w = words;
    const uint32_t idx = byte_off >> 2, skip = (byte_off & 3u) * 8u;
    base = idx;
    cur = fetch(idx + threadIdx.x);
    buf = (uint64_t)(word_of(0) >> skip);
    cnt = 32u - skip;
    rel = 1u;
  }
  __device__ __forceinline__ uint32_t word_of(uint32_t lane_of) const {  // the word that lane holds
    return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)lane_of);
  }
  __device__ __forceinline__ void refill() {  // afterwards at least 33 bits
    if (cnt <= 32u) {
      if (rel == 64u) {
        base += 64u;
        cur = fetch(base + threadIdx.x);  // (a truncated or damaged member decodes zeros from its end on: BAD_MEMBER below, nothing read beyond it)
        rel = 0;
      }
      buf |= (uint64_t)word_of(rel) << cnt;
      ++rel;
      cnt += 32u;
    }
  }
  __device__ __forceinline__ uint32_t next_word() const { return base + rel; }  // index of the next word to take
  __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
  __device__ __forceinline__ void drop(uint32_t n) {
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ uint32_t take(uint32_t n) {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  __device__ __forceinline__ uint32_t byte_pos() const { return next_word() * 4u - cnt / 8u; }  // of the next unread bit's byte (cnt a multiple of 8)
};

// what a symbol of one of the three codes stands for, as a table entry of `len` bits
__device__ __forceinline__ uint32_t ing_entry(int kind, uint32_t s, uint32_t len) {
  if (kind == 0) {  // literal / length (RFC 1951 §3.2.5)
    if (s < 256u) return len | (1u << 6) | (s << 8);
    if (s == 256u) return len | (ING_T_EOB << 4) | ING_RARE;
    if (s > 285u) return ING_RARE;
    const uint32_t t = s - 257u;
    uint32_t base, extra;
    if (t < 8u) {
      base = 3u + t;
      extra = 0;
    } else if (t == 28u) {
      base = 258u;
      extra = 0;
    } else {
      extra = (t - 4u) >> 2;
      base = 3u + ((4u + (t & 3u)) << extra);
    }
    return len | ING_LENGTH | (base << 6) | (extra << 16) | ((len + extra) << 23);
  }
  return len | (ING_T_LIT << 4) | (s << 8);  // code-length code: the symbol itself
}

// distance symbol s as a table entry of `len` bits: low word = bits 0-3 len, bits 16-19 extra bits, bits 23-27 len + extra bits
// (the word is the s_bfe_u32 operand that takes the extra bits out of the stream); high word = base value. s > 29: ING_RARE
__device__ __forceinline__ uint64_t ing_dist_entry(uint32_t s, uint32_t len) {
  if (s > 29u) return ING_RARE;
  uint32_t base, extra;
  if (s < 4u) {
    base = 1u + s;
    extra = 0;
  } else {
    extra = (s - 2u) >> 1;
    base = 1u + ((2u + (s & 1u)) << extra);
  }
  return ((uint64_t)base << 32) | len | (extra << 16) | ((len + extra) << 23);
}

// Canonical Huffman code of lens[0, n) (RFC 1951 §3.2.2) -> look-up table of `root` bits + the sorted symbols and the
// counts per length for the bit-by-bit path. The wave works on 64 symbols at a time; a symbol's rank among those of its
// length comes from ballots. Returns false on an over-subscribed set of lengths.
__device__ __attribute__((noinline)) bool ing_build(const uint8_t *lens, uint32_t n, uint32_t *tab, uint32_t root, uint16_t *sorted, uint16_t *count, int kind) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint64_t *const tab2 = reinterpret_cast<uint64_t *>(tab);  // kind 1: the distance table's entries have two words
  for (uint32_t i = lane; i < (1u << root); i += 64u) {
    if (kind == 1) tab2[i] = ING_RARE; else tab[i] = ING_RARE;
  }
  uint32_t cnt[16];
#pragma unroll
  for (int L = 0; L < 16; ++L) cnt[L] = 0;
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t s = base + lane, l = s < n ? lens[s] : 0u;
#pragma unroll
    for (uint32_t L = 1; L <= 15u; ++L) cnt[L] += (uint32_t)__popcll(__ballot(l == L));
  }
  int left = 1;
#pragma unroll
  for (int L = 1; L <= 15; ++L) {
    left <<= 1;
    left -= (int)cnt[L];
    if (left < 0) return false;
  }
  uint32_t first[16], offs[16], run[16];
  {
    uint32_t code = 0, o = 0;
    first[0] = offs[0] = run[0] = 0;
#pragma unroll
    for (int L = 1; L <= 15; ++L) {
      code = (code + cnt[L - 1]) << 1;
      first[L] = code;
      offs[L] = run[L] = o;
      o += cnt[L];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int L = 0; L < 16; ++L) count[L] = (uint16_t)cnt[L];
    count[0] = 0;
  }
  __syncthreads();  // (the zeroed table before the entries)
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t s = base + lane, l = s < n ? lens[s] : 0u;
    uint32_t my_at = 0, my_first = 0, my_offs = 0;
#pragma unroll
    for (uint32_t L = 1; L <= 15u; ++L) {
      const unsigned long long m = __ballot(l == L);
      if (l == L) {
        my_at = run[L] + (uint32_t)__popcll(m & lt);
        my_first = first[L];
        my_offs = offs[L];
      }
      run[L] += (uint32_t)__popcll(m);
    }
    if (l) {
      sorted[my_at] = (uint16_t)s;
      const uint32_t code = my_first + (my_at - my_offs);
      const uint32_t rev = __builtin_bitreverse32(code) >> (32u - l);
      if (l <= root) {
        if (kind == 1) {
          const uint64_t e = ing_dist_entry(s, l);
          for (uint32_t j = rev; j < (1u << root); j += 1u << l) tab2[j] = e;
        } else {
          const uint32_t e = ing_entry(kind, s, l);
          for (uint32_t j = rev; j < (1u << root); j += 1u << l) tab[j] = e;
        }
      } else if (kind == 1) {
        tab2[rev & ((1u << root) - 1u)] = 15u | (ING_T_LONG << 4) | ING_RARE;
      } else {
        tab[rev & ((1u << root) - 1u)] = 15u | (ING_T_LONG << 4) | ING_RARE;
      }
    }
  }
  __syncthreads();
  return true;
}

// Literal/length table: an entry whose literal leaves room in the root bits for the next code, when that is a literal too,
// becomes both (and a third): the bases of a FASTQ take two bits each, a run of one quality value one or two. The entry at j
// = the bits behind the first code does not depend on bits it does not cover, so T[i >> l1] IS the second look-up. In place,
// from the top down: an entry only looks at entries below itself, and a wave's LDS accesses keep their order.
__device__ __attribute__((noinline)) void ing_fuse(uint32_t *tab, uint32_t root) {
  const uint32_t lane = threadIdx.x & 63u;
  for (int c = (int)((1u << root) / 64u) - 1; c >= 0; --c) {
    const uint32_t i = (uint32_t)c * 64u + lane;
    uint32_t e = tab[i];
    uint32_t l = e & 15u;
    if (l != 0 && ING_IS_LIT(e) && l < root) {
      const uint32_t e2 = tab[i >> l], l2 = e2 & 15u;
      if (l2 != 0 && ING_IS_LIT(e2) && l + l2 <= root) {
        e = (l + l2) | (2u << 6) | (e & 0xFF00u) | ((e2 & 0xFF00u) << 8);
        l += l2;
        if (l < root) {
          const uint32_t e3 = tab[i >> l], l3 = e3 & 15u;
          if (l3 != 0 && ING_IS_LIT(e3) && l + l3 <= root && (e3 & 0x8000u) == 0) e = (l + l3) | (3u << 6) | (e & 0xFFFF00u) | ((e3 & 0xFF00u) << 16);
        }
      }
    }
    tab[i] = e;
  }
  __syncthreads();
}

// a code longer than the table's root: bit by bit against the canonical code's first code of every length. Out of line and
// by value (the symbol loop stays small): returns symbol | bits taken << 16, or ~0 when no code matches.
__device__ __attribute__((noinline)) uint32_t ing_decode_slow(uint64_t buf, const uint16_t *count, const uint16_t *sorted) {
  uint32_t code = 0, first = 0, index = 0;
  for (uint32_t len = 1; len <= 15u; ++len) {
    code |= (uint32_t)(buf >> (len - 1u)) & 1u;
    const uint32_t c = uni(count[len]);
    if (code - first < c) return uni(sorted[index + (code - first)]) | (len << 16);
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return 0xFFFFFFFFu;
}

// What the symbol loop does with a table entry that has ING_RARE set (e: its low word), out of line and with ONE result.
// Low word: bits 0-7 bits to drop now, bits 8-9 the loop's `stop` (1 damaged, 2 the block's end). High word, literal/length
// table (kind 0): the entry the loop goes on with — the symbol's, as an entry of ONE bit (the code's other bits are the ones
// to drop now), or when stopping the entry that does nothing; distance table (kind 1): the distance itself, extra bits
// included (all of its bits are dropped now), 1 when stopping.
__device__ __attribute__((noinline)) uint64_t ing_rare(int kind, uint32_t e, uint64_t buf, const uint16_t *count, const uint16_t *sorted) {
  const uint64_t nop = (uint64_t)(kind == 0 ? 0u : 1u) << 32;
  const uint64_t bad = nop | (1u << 8);
  if ((e & 15u) == 0) return bad;                                             // no code here
  if (ING_TYPE(e) == ING_T_EOB) return nop | (2u << 8) | (e & 15u);             // the end of the block, from the table
  const uint32_t r = ing_decode_slow(buf, count, sorted);                     // a code longer than the table's bits
  if (r == 0xFFFFFFFFu) return bad;
  const uint32_t sym = r & 0xFFFFu, bits = r >> 16;
  if (kind == 1) {
    const uint64_t de = ing_dist_entry(sym, bits);
    if ((uint32_t)de & ING_RARE) return bad;
    const uint32_t extra = ((uint32_t)de >> 16) & 15u;
    const uint32_t dist = (uint32_t)(de >> 32) + ((uint32_t)(buf >> bits) & ((1u << extra) - 1u));
    return ((uint64_t)dist << 32) | (bits + extra);
  }
  const uint32_t e1 = ing_entry(0, sym, 1);
  if (e1 & ING_RARE) return (e1 & 15u) != 0 && ING_TYPE(e1) == ING_T_EOB ? nop | (2u << 8) | bits : bad;
  return ((uint64_t)e1 << 32) | (bits - 1u);
}

// CRC-32 (IEEE, reflected) as polynomial arithmetic: a * b mod P, and x^n mod P (zlib's crc32_combine does the same)
#define ING_CRC_POLY 0xedb88320u
__device__ __forceinline__ uint32_t ing_mulmod(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & m) p ^= b;
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ ING_CRC_POLY : b >> 1;
  }
  return p;
}
__device__ __forceinline__ uint32_t ing_xpow8(uint32_t n_bytes) {  // x^(8 n)
  uint32_t base = 1u << 30;  // x^1
  base = ing_mulmod(base, base);
  base = ing_mulmod(base, base);
  base = ing_mulmod(base, base);  // x^8
  uint32_t r = 1u << 31;          // x^0
  while (n_bytes) {
    if (n_bytes & 1u) r = ing_mulmod(r, base);
    base = ing_mulmod(base, base);
    n_bytes >>= 1;
  }
  return r;
}

// The member's text is addressed as GLOBAL memory explicitly: through a generic pointer the stores were flat_store_byte,
// which count as LDS operations too — every table look-up behind a flushed line then waited for the line to reach memory.
typedef __attribute__((address_space(1))) uint8_t ing_g8;
typedef __attribute__((address_space(1))) uint32_t ing_g32;
typedef uint32_t ing_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) ing_v4 ing_g128;
__device__ __forceinline__ uint8_t ing_load_coherent(const ing_g8 *p) {  // text this wave stored earlier: past the vector L1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// [lo, hi) within one KB of the window -> memory; the member's first and last pieces: a neighbour owns the rest of a 16-byte piece
__device__ __attribute__((noinline)) void ing_flush_partial(const uint8_t *ring, ing_g8 *al, uint32_t lo, uint32_t hi) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t cs = (lo & ~1023u) + 16u * lane, ce = cs + 16u;
  if (cs >= lo && ce <= hi) {
    *(ing_g128 *)(al + cs) = *reinterpret_cast<const ing_v4 *>(&ring[cs & ING_RING_MASK]);
  } else if (ce > lo && cs < hi) {
    for (uint32_t i = max(cs, lo); i < min(ce, hi); ++i) al[i] = ring[i & ING_RING_MASK];
  }
}

// One wavefront per member: its text goes to text[out_off, out_off + isize).
template <bool STATS>
__device__ __forceinline__ void ing_inflate_member(WaveLds &L, const uint32_t *__restrict__ comp, const IngestMember *__restrict__ members, uint32_t n_members,
                                                   uint8_t *text, IngestInflateStatus *st, int check_crc, unsigned long long *dbg, uint32_t exp_mode) {
  const uint32_t xm = STATS ? uni(exp_mode) : 0u;  // timing experiments (GMX_INGEST_EXP, with GMX_INGEST_STATS): the text is then wrong
  // dbg (GMX_INGEST_STATS=1): [0] look-ups of the literal/length table [1] literal bytes [2] matches [3] far matches [4] bit-by-bit
  // decodes [5] blocks [6..9] clocks: whole member, table building, match copies, CRC [10] members [11] match bytes
  unsigned long long n_look = 0, n_lit_total = 0, n_match = 0, n_far = 0, n_slow = 0, n_blocks = 0, t_build = 0, t_copy = 0, t_crc = 0, n_mbytes = 0;
  const long long t_begin = STATS ? clock64() : 0;
  const uint32_t mi = blockIdx.x;
  if (mi >= n_members) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t in_off = uni(members[mi].in_off), in_len = uni(members[mi].in_len), isize = uni(members[mi].isize);
  // Positions are counted from the 16-byte boundary at or below the member's first byte: byte p of that line-up lives at
  // al[p] and at ring[p & mask], so that a KB of the window goes to memory as one 16-byte store per lane (64-byte lines
  // stored byte by byte were 1 024 store instructions per member — and, loads and stores returning in order, every wait
  // for a load was a wait for the last of them).
  ing_g8 *const out0 = (ing_g8 *)(text + uni(members[mi].out_off));
  const uint32_t mis = (uint32_t)((uintptr_t)out0 & 15u);
  ing_g8 *const al = out0 - mis;
  const uint32_t end_v = isize + mis;
  Bits bs;
  bs.limit = (in_off + in_len + 3u) / 4u + 2u;  // (the member's words and the zeroed ones right behind them: ADVICE round 5 — the reader used to run on into the next member, or past the chunk's bytes)
  bs.start(comp, in_off);
  uint32_t out_pos = mis, flushed = mis;  // bytes decoded so far end here; bytes already stored from the window end here
  uint32_t err = 0;
  // The symbol loop below has only wave-uniform branches: a lane with nothing to do writes its byte to L.dummy instead of
  // sitting out a branch. (With lane-divergent branches and loops inside it the compiler restructured the whole loop around
  // flag registers: some 170 instruction slots per table look-up — and one wave issues an instruction every four clocks.)
  // (out of line: its branches are lane-divergent, and one divergent branch inside the symbol loop makes the compiler restructure
  // the whole loop around flag registers; with none the loop's wave-uniform branches stay plain scalar branches)
  auto flush_partial = [&](uint32_t lo, uint32_t hi) { ing_flush_partial(L.ring, al, lo, hi); };
  auto flush_to = [&](uint32_t upto) {  // [flushed, upto), upto a multiple of 1024
    if (xm & 4u) {
      flushed = upto;
      return;
    }
    uint32_t blk = flushed;
    if (blk & 1023u) {  // (once: the member's first KB starts at `mis`)
      flush_partial(blk, (blk & ~1023u) + 1024u);
      blk = (blk & ~1023u) + 1024u;
    }
    for (; blk < upto; blk += 1024u) *(ing_g128 *)(al + blk + 16u * lane) = *reinterpret_cast<const ing_v4 *>(&L.ring[(blk + 16u * lane) & ING_RING_MASK]);
    flushed = upto;
  };
  // bytes [out_pos, out_pos + len) = the len bytes starting dist back (RFC 1951 §3.2.3: may overlap what it writes); len >= 3.
  // Three loops that run at least once, so that the choice between them is two plain branches.
  auto copy_match = [&](uint32_t len, uint32_t dist) {
    if (xm & 1u) {
      out_pos += len;
      if ((out_pos & ~1023u) > flushed) flush_to(out_pos & ~1023u);
      return;
    }
    uint32_t i0 = 0;
    if (dist > ING_NEAR_MAX && !(xm & 2u)) {  // beyond the window: from the KBs already flushed (dist > 1 728: every byte read lies below `flushed`, for the idle lanes too)
      do {
        const uint32_t i = i0 + lane;
        const uint8_t v = ing_load_coherent(al + (out_pos - dist + i));
        uint8_t *dst = i < len ? &L.ring[(out_pos + i) & ING_RING_MASK] : &L.dummy[lane];
        *dst = v;
        i0 += 64u;
      } while (i0 < len);
    } else if (dist >= len) {  // from the window
      do {
        const uint32_t i = i0 + lane;
        const uint8_t v = L.ring[(out_pos - dist + i) & ING_RING_MASK];
        uint8_t *dst = i < len ? &L.ring[(out_pos + i) & ING_RING_MASK] : &L.dummy[lane];
        *dst = v;
        i0 += 64u;
      } while (i0 < len);
    } else {  // the copy overlaps what it writes: byte i repeats byte i mod dist
      do {
        const uint32_t i = i0 + lane;
        const uint8_t v = L.ring[(out_pos - dist + i % dist) & ING_RING_MASK];
        uint8_t *dst = i < len ? &L.ring[(out_pos + i) & ING_RING_MASK] : &L.dummy[lane];
        *dst = v;
        i0 += 64u;
      } while (i0 < len);
    }
    out_pos += len;
    if (__builtin_expect((out_pos & ~1023u) > flushed, 0)) flush_to(out_pos & ~1023u);
  };
  for (bool last = false; !last && !err;) {
    bs.refill();
    last = bs.take(1) != 0;
    const uint32_t btype = bs.take(2);
    if (STATS) ++n_blocks;
    const long long t_b0 = STATS ? clock64() : 0;
    if (btype == 0) {  // stored (RFC 1951 §3.2.4)
      bs.drop(bs.cnt & 7u);
      bs.refill();
      const uint32_t len = bs.take(16), nlen = bs.take(16);
      if ((len ^ nlen) != 0xFFFFu) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
      const uint32_t from = bs.byte_pos();
      if (from + len > in_off + in_len || out_pos + len > end_v) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
      const uint8_t *src = reinterpret_cast<const uint8_t *>(comp) + from;
      for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
        const uint32_t n = min(64u, len - i0);
        *(lane < n ? &L.ring[(out_pos + lane) & ING_RING_MASK] : &L.dummy[lane]) = src[i0 + min(lane, n - 1u)];
        out_pos += n;
        if ((out_pos & ~1023u) > flushed) flush_to(out_pos & ~1023u);
      }
      bs.start(comp, from + len);
      continue;
    }
    if (btype == 3) {
      err = GMX_INGEST_BAD_MEMBER;
      break;
    }
    if (btype == 1) {  // fixed codes (§3.2.6)
      for (uint32_t s0 = 0; s0 < 320u; s0 += 64u) {  // (lens[] has 320 entries; no lane-divergent branch inside the block loop, see the symbol loop)
        const uint32_t s = s0 + lane;
        L.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8;
      }
      __syncthreads();
      bool ok = uni(ing_build(L.lens, 288, L.lit, ING_LIT_ROOT, L.lit_sorted, L.lit_count, 0)) != 0;
      L.lens[lane] = 5;  // (32 distance codes; the lanes above write lengths nobody reads)
      __syncthreads();
      ok = uni(ing_build(L.lens, 32, reinterpret_cast<uint32_t *>(L.dist), ING_DIST_ROOT, L.dist_sorted, L.dist_count, 1)) != 0 && ok;
      if (!ok) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
    } else {  // dynamic codes (§3.2.7)
      // (while the lengths are read a lane with nothing to write writes into lit_sorted[]: the block before is done with it, this one
      // fills it when its lengths are complete; dummy[] shares its bytes with lens[])
      uint8_t *const hdr_dummy = reinterpret_cast<uint8_t *>(L.lit_sorted) + lane;
      const uint32_t hlit = bs.take(5) + 257u, hdist = bs.take(5) + 1u, hclen = bs.take(4) + 4u;
      if (hlit > 286u || hdist > 30u) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
      *(lane < 19u ? &L.lens[lane] : hdr_dummy) = 0;
      __syncthreads();
      for (uint32_t i = 0; i < hclen; ++i) {
        bs.refill();
        const uint32_t v = bs.take(3);
        // order of the code-length code's lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const uint32_t sym = i < 3u ? 16u + i : i == 3u ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1) + 0u;
        *(lane == 0 ? &L.lens[sym] : hdr_dummy) = (uint8_t)v;
      }
      __syncthreads();
      // (the code-length code's table borrows the distance table: root 7, every code at most 7 bits)
      uint32_t *const cl_tab = reinterpret_cast<uint32_t *>(L.dist);
      if (!uni(ing_build(L.lens, 19, cl_tab, 7, L.dist_sorted, L.dist_count, 2))) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
      const uint32_t total = hlit + hdist;
      uint32_t at = 0, prev = 0;
      while (at < total && !err) {
        bs.refill();
        const uint32_t e = uni(cl_tab[bs.peek(7)]);
        if ((e & 15u) == 0) {
          err = GMX_INGEST_BAD_MEMBER;
          break;
        }
        bs.drop(e & 15u);
        const uint32_t sym = e >> 8;
        uint32_t rep = 1, val = sym;
        if (sym == 16u) {
          if (at == 0) {
            err = GMX_INGEST_BAD_MEMBER;
            break;
          }
          rep = 3u + bs.take(2);
          val = prev;
        } else if (sym == 17u) {
          rep = 3u + bs.take(3);
          val = 0;
        } else if (sym == 18u) {
          rep = 11u + bs.take(7);
          val = 0;
        }
        if (at + rep > total) {
          err = GMX_INGEST_BAD_MEMBER;
          break;
        }
        for (uint32_t i0 = 0; i0 < rep; i0 += 64u) *(i0 + lane < rep ? &L.lens[at + i0 + lane] : hdr_dummy) = (uint8_t)val;  // (lens[0, 19) held the code-length code's lengths: its table is built)
        at += rep;
        prev = val;
      }
      if (err) break;
      for (uint32_t i0 = total; i0 < 320u; i0 += 64u) *(i0 + lane < 320u ? &L.lens[i0 + lane] : hdr_dummy) = 0;
      __syncthreads();
      if (uni(L.lens[256]) == 0) {  // no end-of-block code
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
      bool ok = uni(ing_build(L.lens, hlit, L.lit, ING_LIT_ROOT, L.lit_sorted, L.lit_count, 0)) != 0;
      ok = uni(ing_build(L.lens + hlit, hdist, reinterpret_cast<uint32_t *>(L.dist), ING_DIST_ROOT, L.dist_sorted, L.dist_count, 1)) != 0 && ok;
      if (!ok) {
        err = GMX_INGEST_BAD_MEMBER;
        break;
      }
    }
    ing_fuse(L.lit, ING_LIT_ROOT);
    if (STATS) t_build += clock64() - t_b0;
    // ---- the block's symbols ----
    // The decoder is bound by the scalar instructions it issues (see below the function), so the loop is written the way the
    // compiler's control-flow passes leave alone: properly nested if / else, ONE way out at the bottom (`stop`, kept from the
    // optimizer so that it does not thread the rare cases to the code behind the loop), the rare cases out of line with one
    // result, nothing lane-divergent. As lambdas with early returns it carried its exits as a guard value and flag registers set
    // and tested on every path: 87 scalar instructions per match; this form takes 60-odd.
    {
      uint32_t stop = 0;  // 1: the member is damaged, 2: the block's end
      do {
        bs.refill();
        uint32_t e = uni(L.lit[ing_vpeek<ING_LIT_ROOT>((uint32_t)bs.buf)]);
        if (STATS) ++n_look;
        if (__builtin_expect((int32_t)e < 0, 0)) {  // ING_RARE: no code here, the end of the block, or a code longer than the table's bits
          if (STATS) ++n_slow;
          const uint64_t rr = ing_rare(0, e, bs.buf, L.lit_count, L.lit_sorted);
          const uint32_t r_lo = uni((uint32_t)rr);
          bs.drop(r_lo & 0xFFu);
          stop = r_lo >> 8;
          e = uni((uint32_t)(rr >> 32));  // (when stopping: the entry that does nothing)
        }
        if ((e & ING_LENGTH) == 0) {  // one to three literals (or none)
          const uint32_t n_lit = (e >> 6) & 3u;
          bs.drop(e & 15u);
          if (__builtin_expect(out_pos + n_lit > end_v, 0)) {
            stop = 1u;
          } else {
            uint8_t *dst = lane < n_lit ? &L.ring[(out_pos + lane) & ING_RING_MASK] : &L.dummy[lane];
            *dst = (uint8_t)(e >> (8u + 8u * (lane & 3u)));
            out_pos += n_lit;
            if (STATS) n_lit_total += n_lit;
            if (__builtin_expect((out_pos & ~1023u) > flushed, 0)) flush_to(out_pos & ~1023u);
          }
        } else {  // a length: its extra bits, then the distance code and its extra bits (RFC 1951 §3.2.5)
          const uint32_t len = ((e >> 6) & 0x1FFu) + ing_sbfe((uint32_t)bs.buf, e);
          bs.drop(e >> 23);
          bs.refill();
          const uint64_t de = L.dist[ing_vpeek<ING_DIST_ROOT>((uint32_t)bs.buf)];
          const uint32_t d = uni((uint32_t)de);
          uint32_t dist;
          if (__builtin_expect((int32_t)d < 0, 0)) {  // ING_RARE
            const uint64_t rr = ing_rare(1, d, bs.buf, L.dist_count, L.dist_sorted);
            const uint32_t r_lo = uni((uint32_t)rr);
            bs.drop(r_lo & 0xFFu);
            stop = r_lo >> 8;
            dist = uni((uint32_t)(rr >> 32));  // (when stopping: 1)
          } else {
            dist = uni((uint32_t)(de >> 32)) + ing_sbfe((uint32_t)bs.buf, d);
            bs.drop(d >> 23);
          }
          // (both in one test: every value is far below 2^31)
          if (__builtin_expect((int32_t)((out_pos - mis - dist) | (end_v - out_pos - len)) < 0, 0)) {
            stop = 1u;
          } else {
            if (STATS) {
              ++n_match;
              n_mbytes += len;
              if (dist > ING_NEAR_MAX) ++n_far;
            }
            const long long t_c0 = STATS ? clock64() : 0;
            copy_match(len, dist);
            if (STATS) t_copy += clock64() - t_c0;
          }
        }
        asm volatile("" : "+s"(stop));
      } while (stop == 0);
      if (stop == 1u) err = GMX_INGEST_BAD_MEMBER;
    }
  }
  if (!err) {
    if ((out_pos & ~1023u) > flushed) flush_to(out_pos & ~1023u);
    if (out_pos > flushed) {  // the member's last piece (and, for a member below one KB, its first)
      flush_partial(flushed, out_pos);
      flushed = out_pos;
    }
    // every byte of the member's deflate data used, and as much text as its trailer says
    const uint32_t used_bits = (bs.next_word() * 32u - bs.cnt) - in_off * 8u;
    if (out_pos != end_v || (used_bits + 7u) / 8u != in_len) err = GMX_INGEST_BAD_MEMBER;
  }
  const long long t_crc0 = STATS ? clock64() : 0;
  if (!err && check_crc && isize) {
    // CRC-32 of the member's text: 64 slices side by side, four table look-ups per word (the tables take the place of the
    // literal/length table), then the slices' registers combined pairwise. (One byte per step from memory — a memory round
    // trip each — was a third of the kernel's time.)
    static_assert(offsetof(WaveLds, ring) + ING_RING >= 4096u && offsetof(WaveLds, lit) == 0, "the four CRC tables take the place of the tables and the window");
    uint32_t *const crc_tab = reinterpret_cast<uint32_t *>(&L);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    for (uint32_t i = lane; i < 256u; i += 64u) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ ING_CRC_POLY : c >> 1;
      crc_tab[i] = c;
    }
    __syncthreads();
    for (uint32_t t = 1; t < 4u; ++t) {
      for (uint32_t i = lane; i < 256u; i += 64u) {
        const uint32_t v = crc_tab[(t - 1u) * 256u + i];
        crc_tab[t * 256u + i] = (v >> 8) ^ crc_tab[v & 0xFFu];
      }
      __syncthreads();
    }
    // slices of whole words, counted like everything else from the 16-byte boundary below the member's first byte
    const uint32_t total = end_v;
    const uint32_t per = (((total + 63u) / 64u) + 3u) & ~3u;
    // (EVERY lane's slice starts at the member's first byte at the earliest — `mis` bytes above the boundary: with only lane 0
    //  clamped, a member of a few hundred bytes that does not start on a 16-byte boundary had the bytes in front of it hashed
    //  by lanes 1.. and was reported as GMX_INGEST_BAD_CRC)
    uint32_t lo = min(total, max(mis, lane * per));
    const uint32_t hi = max(lo, min(total, (lane + 1u) * per));
    uint32_t c = lane == 0 ? 0xFFFFFFFFu : 0u;
    const uint32_t n = hi > lo ? hi - lo : 0u;
    auto byte_step = [&](uint32_t at) { c = crc_tab[(c ^ ing_load_coherent(al + at)) & 0xFFu] ^ (c >> 8); };
    while (lo < hi && (lo & 3u)) byte_step(lo++);
    const ing_g32 *words = (const ing_g32 *)al;
    for (; lo + 16u <= hi; lo += 16u) {  // four words in flight
      const uint32_t w0 = __hip_atomic_load(words + lo / 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     w1 = __hip_atomic_load(words + lo / 4u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     w2 = __hip_atomic_load(words + lo / 4u + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     w3 = __hip_atomic_load(words + lo / 4u + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t ws[4] = {w0, w1, w2, w3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c ^= ws[q];
        c = crc_tab[768u + (c & 0xFFu)] ^ crc_tab[512u + ((c >> 8) & 0xFFu)] ^ crc_tab[256u + ((c >> 16) & 0xFFu)] ^ crc_tab[c >> 24];
      }
    }
    for (; lo + 4u <= hi; lo += 4u) {
      c ^= __hip_atomic_load(words + lo / 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c = crc_tab[768u + (c & 0xFFu)] ^ crc_tab[512u + ((c >> 8) & 0xFFu)] ^ crc_tab[256u + ((c >> 16) & 0xFFu)] ^ crc_tab[c >> 24];
    }
    while (lo < hi) byte_step(lo++);
    uint32_t nn = n;
    for (uint32_t dlt = 1; dlt < 64u; dlt <<= 1) {  // register of (left part || right part) = left * x^(8 |right|) + right
      const uint32_t c_r = __shfl_down(c, dlt), n_r = __shfl_down(nn, dlt);
      if ((lane & (2u * dlt - 1u)) == 0) {
        c = (n_r ? ing_mulmod(c, ing_xpow8(n_r)) : c) ^ c_r;
        nn += n_r;
      }
    }
    if (uni(~c) != uni(members[mi].crc)) err = GMX_INGEST_BAD_CRC;
  }
  if (STATS && dbg && lane == 0) {
    t_crc = clock64() - t_crc0;
    const unsigned long long v[12] = {n_look, n_lit_total, n_match, n_far, n_slow, n_blocks, (unsigned long long)(clock64() - t_begin), t_build, t_copy, t_crc, 1ull, n_mbytes};
    for (int i = 0; i < 12; ++i) atomicAdd(&dbg[i], v[i]);
  }
  if (err && lane == 0) {
    atomicOr(&st->flags, err);
    atomicMin(&st->bad_member, mi);
  }
}

// The decoder is ONE thread of control per wavefront, compiled scalar: it runs on the CU's single scalar unit, which all of
// the CU's wavefronts share. That unit is what bounds the kernel (profiles/round5/ingest_inflate_sq_counters.txt: 603 k scalar
// instructions per member, the scalar unit busy 68 % of the kernel's cycles; 4 or 7 waves per SIMD decode the same
// 30-34 GB/s of text). The same arithmetic compiled for the vector units — every lane redundantly — came out no faster: its
// branches cost as many scalar instructions (exec masks) as the scalar form's arithmetic. DESIGN.md §11: what comes next.
template <bool STATS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ING_WAVES, 8)))
gmx_inflate_kernel(const uint32_t *__restrict__ comp, const IngestMember *__restrict__ members, uint32_t n_members, uint8_t *text, IngestInflateStatus *st,
                   int check_crc, unsigned long long *dbg, uint32_t exp_mode) {
  __shared__ WaveLds L;
  ing_inflate_member<STATS>(L, comp, members, n_members, text, st, check_crc, dbg, exp_mode);
}

// ------------------------------------------------------------------------------------------------------------------
// the chunk's text -> records -> bit planes
// ------------------------------------------------------------------------------------------------------------------
// starts a chunk: the incomplete last record of the chunk before (prev; null: a file's first chunk) in front of this one's text
__global__ void gmx_carry_kernel(const IngestState *prev, const uint8_t *prev_text, IngestState *cur, uint8_t *cur_text, uint32_t members_text,
                                 uint32_t final_chunk, uint32_t host_tail) {
  uint32_t tail = host_tail;  // (gmx_ingest_scan: the caller has put that many bytes in front of the chunk's text already)
  if (prev) {
    tail = prev->tail_len;
    if (tail > ING_CARRY_MAX) tail = 0;  // (reported below)
    const uint8_t *src = prev_text + prev->consumed;
    uint8_t *dst = cur_text + ING_CARRY_MAX - tail;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tail; i += gridDim.x * blockDim.x) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cur->text_start = ING_CARRY_MAX - tail;
    cur->text_len = tail + members_text;
    cur->n_lines = cur->n_reads = 0;
    cur->min_len = 0xFFFFFFFFu;
    cur->max_len = 0;
    cur->flags = (prev && prev->tail_len > ING_CARRY_MAX) ? GMX_INGEST_BAD_RECORD : 0u;
    cur->bad_member = 0xFFFFFFFFu;
    cur->consumed = ING_CARRY_MAX - tail;
    cur->tail_len = 0;
    cur->any_skip = 0;
    cur->uniform_len = 0;
    cur->n_bases = 0;
    cur->n_pairs = 0;
    for (int i = 0; i < 16; ++i) cur->sub_pairs[i] = 0;
    cur->final_chunk = final_chunk;
  }
}

#define ING_TILE 4096u  // bytes of text per 256-thread block of the newline kernels (16 per thread)
__device__ __forceinline__ uint32_t ing_nl_mask16(const uint8_t *text, uint32_t at, uint32_t lo, uint32_t hi) {  // bit j: text[at + j] == '\n', within [lo, hi)
  if (at + 16u <= lo || at >= hi) return 0u;
  const uint4 v = *reinterpret_cast<const uint4 *>(text + at);  // (at a multiple of 16; the buffer is padded)
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t t = w[q] ^ 0x0A0A0A0Au;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);  // 0x80 in every zero byte, exactly
    m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
  }
  if (at < lo) m &= ~((1u << (lo - at)) - 1u);
  if (at + 16u > hi) m &= (1u << (hi - at)) - 1u;
  return m;
}
__global__ void __launch_bounds__(256) gmx_nl_count_kernel(const uint8_t *text, const IngestState *st, uint32_t *tile_count) {
  const uint32_t lo = st->text_start, hi = lo + st->text_len;
  const uint32_t at = blockIdx.x * ING_TILE + threadIdx.x * 16u;
  uint32_t c = (uint32_t)__popc(ing_nl_mask16(text, at, lo, hi));
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// Exclusive scans in two levels, ONE WAVEFRONT per block and no LDS (round 5, second half). The chunk's result waits for these
// scans, and they run beside the inflate kernel of the next chunk, whose twenty wavefronts per CU hold all of the CU's LDS and five
// of a SIMD's eight wave slots: the single block of 1024 threads with 8 KB of LDS that did this until then found no CU to start on
// before that kernel's queue of workgroups ran dry (1.7-3.4 ms instead of 0.25). A wavefront without LDS fits beside them.
//   level 1 (a block per 1024 elements, 16 per lane): offsets within the block (out may alias in), the block's sum -> blk_tot
//   level 2 (one wavefront): blk_tot -> the blocks' offsets, in place; returns the total
// and whoever reads element i adds blk_tot[i >> 10].
#define ING_SCAN_BLOCK 1024u
template <class TIn, class TOut, class TTot>
__device__ __forceinline__ void ing_scan_level1(const TIn *in, TOut *out, uint32_t n, uint32_t blk, TTot *blk_tot) {
  const uint32_t lane = threadIdx.x & 63u, base = blk * ING_SCAN_BLOCK + lane * 16u;
  TTot v[16], sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < 16u; ++j) {
    v[j] = base + j < n ? (TTot)in[base + j] : (TTot)0;
    sum += v[j];
  }
  TTot incl = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const TTot up = __shfl_up(incl, d);
    if ((int)lane >= d) incl += up;
  }
  TTot run = incl - sum;
#pragma unroll
  for (uint32_t j = 0; j < 16u; ++j) {
    if (base + j < n) out[base + j] = (TOut)run;
    run += v[j];
  }
  if (lane == 63u) blk_tot[blk] = incl;
}
template <class TTot>
__device__ __forceinline__ TTot ing_scan_level2(TTot *blk_tot, uint32_t n_blk) {
  const uint32_t lane = threadIdx.x & 63u, per = (n_blk + 63u) / 64u;
  const uint32_t lo = min(n_blk, lane * per), hi = min(n_blk, lo + per);
  TTot sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += blk_tot[i];
  TTot incl = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const TTot up = __shfl_up(incl, d);
    if ((int)lane >= d) incl += up;
  }
  TTot run = incl - sum;
  for (uint32_t i = lo; i < hi; ++i) {
    const TTot v = blk_tot[i];
    blk_tot[i] = run;
    run += v;
  }
  return __shfl(incl, 63);
}
__global__ void __launch_bounds__(64) gmx_tile_scan1_kernel(uint32_t *tile_count, uint32_t n_tiles, uint32_t *tile_blk) {
  ing_scan_level1<uint32_t, uint32_t, uint32_t>(tile_count, tile_count, n_tiles, blockIdx.x, tile_blk);
}
__global__ void __launch_bounds__(64) gmx_tile_scan2_kernel(uint32_t *tile_blk, uint32_t n_blk, IngestState *st, uint32_t cap_lines) {
  // (a tile holds at most 4096 newlines and a chunk's text at most 3 GB: the total fits 32 bits)
  const uint32_t total = ing_scan_level2<uint32_t>(tile_blk, n_blk);
  if (threadIdx.x == 0) {
    st->n_lines = total;
    if (total > cap_lines) atomicOr(&st->flags, GMX_INGEST_TOO_MANY_LINES);
  }
}
__global__ void __launch_bounds__(256) gmx_nl_mark_kernel(const uint8_t *text, const IngestState *st, const uint32_t *tile_base, const uint32_t *tile_blk,
                                                          uint32_t *line_end, uint32_t cap_lines) {
  if (st->flags & GMX_INGEST_TOO_MANY_LINES) return;
  const uint32_t lo = st->text_start, hi = lo + st->text_len;
  const uint32_t at = blockIdx.x * ING_TILE + threadIdx.x * 16u;
  uint32_t m = ing_nl_mask16(text, at, lo, hi);
  const uint32_t c = (uint32_t)__popc(m);
  uint32_t incl = c;  // inclusive scan over the block: within the wave by shuffles, across the four waves through LDS
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if ((int)(threadIdx.x & 63u) >= d) incl += up;
  }
  __shared__ uint32_t wsum[4];
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t before = tile_blk[blockIdx.x / ING_SCAN_BLOCK] + tile_base[blockIdx.x] + incl - c;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
  while (m) {
    const uint32_t j = (uint32_t)__builtin_ctz(m);
    m &= m - 1u;
    if (before < cap_lines) line_end[before] = at + j;
    ++before;
  }
}

// One thread per record (grid-stride). Line i of the chunk ends at line_end[i]; a last line without '\n' ends at the text's
// end when the chunk is the file's last.
__global__ void __launch_bounds__(256) gmx_records_kernel(const uint8_t *text, IngestState *st, const uint32_t *line_end, uint32_t *rec_start,
                                                          uint32_t *rec_len, uint8_t *skip, uint32_t cap_reads) {
  if (st->flags & GMX_INGEST_TOO_MANY_LINES) return;
  const uint32_t lo = st->text_start, hi = lo + st->text_len, n_nl = st->n_lines;
  const uint32_t last_nl_end = n_nl ? line_end[n_nl - 1] + 1u : lo;
  const bool virt = st->final_chunk && last_nl_end < hi;  // text behind the last newline of the file's last chunk: a line
  const uint32_t lines = n_nl + (virt ? 1u : 0u);
  uint32_t n_reads = lines / 4u;
  bool too_many = false;
  if (n_reads > cap_reads) {
    n_reads = 0;
    too_many = true;
  }
  auto le = [&](uint32_t i) { return i < n_nl ? line_end[i] : hi; };
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  unsigned long long bases = 0;
  bool bad = false;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const uint32_t s1 = r ? line_end[4u * r - 1u] + 1u : lo;
    const uint32_t e1 = le(4u * r), e2 = le(4u * r + 1u), e3 = le(4u * r + 2u), e4 = le(4u * r + 3u);
    const uint32_t s2 = e1 + 1u, s3 = e2 + 1u, s4 = e3 + 1u;
    uint32_t n = e2 - s2, nq = e4 >= s4 ? e4 - s4 : 0u;
    if (n && text[s2 + n - 1u] == '\r') --n;
    if (nq && text[s4 + nq - 1u] == '\r') --nq;
    if (text[s1] != '@' || s3 >= hi || text[s3] != '+' || nq != n || n == 0) bad = true;
    rec_start[r] = s2;
    rec_len[r] = n;
    skip[r] = 0;
    mn = min(mn, n);
    mx = max(mx, n);
    bases += n;
    if (r == n_reads - 1u) {
      const uint32_t end = e4 < hi ? e4 + 1u : hi;
      st->consumed = end;
      st->tail_len = hi - end;
      if (st->final_chunk && end != hi) atomicOr(&st->flags, GMX_INGEST_BAD_RECORD);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->n_reads = n_reads;
    if (too_many) atomicOr(&st->flags, GMX_INGEST_TOO_MANY_LINES);
    if (n_reads == 0) {
      st->consumed = lo;
      st->tail_len = hi - lo;
      if (st->final_chunk && hi != lo) atomicOr(&st->flags, GMX_INGEST_BAD_RECORD);  // fewer than four lines at the end of the file
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_down(mn, off));
    mx = max(mx, (uint32_t)__shfl_down(mx, off));
    bases += __shfl_down(bases, off);
  }
  const unsigned long long any_bad = __ballot(bad);
  if ((threadIdx.x & 63u) == 0) {
    if (mn != 0xFFFFFFFFu) atomicMin(&st->min_len, mn);
    if (mx) atomicMax(&st->max_len, mx);
    if (bases) atomicAdd(&st->n_bases, bases);
    if (any_bad) atomicOr(&st->flags, GMX_INGEST_BAD_RECORD);
  }
}
// one length for all reads? -> layout of the planes; reads of different lengths: their base offsets (one block)
__global__ void __launch_bounds__(64) gmx_layout_kernel(IngestState *st, const IngestInflateStatus *inf) {
  if (threadIdx.x == 0 && inf && inf->flags) {  // what the inflate kernel reported
    st->flags |= inf->flags;
    st->bad_member = inf->bad_member;
  }
  const uint32_t n = st->n_reads;
  const bool uniform = n != 0 && st->min_len == st->max_len;
  if (threadIdx.x == 0) {
    st->uniform_len = uniform ? st->max_len : 0u;
    const unsigned long long ppr = (st->max_len + 31u) / 32u;
    if (uniform || n == 0) st->n_pairs = (unsigned long long)n * ppr;
    for (uint32_t i = 0; i < 16u; ++i) st->sub_pairs[i] = uniform || n == 0 ? ((unsigned long long)i << 20) * ppr : 0ull;
  }
}
// reads of different lengths: their base offsets, offsets[0, n] (two-level scan as above; nothing to do for one length)
__global__ void __launch_bounds__(64) gmx_offsets1_kernel(const IngestState *st, const uint32_t *rec_len, unsigned long long *offsets, unsigned long long *off_blk) {
  const uint32_t n = st->n_reads;
  if (st->uniform_len || blockIdx.x * ING_SCAN_BLOCK >= n) return;
  ing_scan_level1<uint32_t, unsigned long long, unsigned long long>(rec_len, offsets, n, blockIdx.x, off_blk);
}
__global__ void __launch_bounds__(64) gmx_offsets2_kernel(IngestState *st, unsigned long long *offsets, unsigned long long *off_blk) {
  const uint32_t n = st->n_reads;
  if (st->uniform_len || n == 0) return;
  const unsigned long long total = ing_scan_level2<unsigned long long>(off_blk, (n + ING_SCAN_BLOCK - 1u) / ING_SCAN_BLOCK);
  if (threadIdx.x == 0) {
    offsets[n] = total;
    st->n_pairs = (total >> 5) + n;
  }
}
__global__ void __launch_bounds__(64) gmx_offsets3_kernel(IngestState *st, unsigned long long *offsets, const unsigned long long *off_blk) {
  const uint32_t n = st->n_reads;
  if (st->uniform_len || blockIdx.x * ING_SCAN_BLOCK >= n) return;
  const unsigned long long base = off_blk[blockIdx.x];
  const uint32_t at = blockIdx.x * ING_SCAN_BLOCK + (threadIdx.x & 63u) * 16u;
  for (uint32_t j = 0; j < 16u; ++j) {
    const uint32_t r = at + j;
    if (r >= n) break;
    const unsigned long long off = offsets[r] + base;
    offsets[r] = off;
    if ((r & 0xFFFFFu) == 0 && (r >> 20) < 16u) st->sub_pairs[r >> 20] = (off >> 5) + r;  // where a launch of <= 2^20 reads starts
  }
}
// One thread per pair of planes: 32 letters -> (low bits, high bits) of the codes A,C,G,T = 0..3; from the letters' own bits
// (bit 2 of the ASCII code is the code's high bit, bit 1 XOR bit 2 the low one: 'A' 0x41 'C' 0x43 'G' 0x47 'T' 0x54, either case)
__global__ void __launch_bounds__(256) gmx_fq_pack_kernel(const uint8_t *text, IngestState *st, const uint32_t *rec_start, const uint32_t *rec_len,
                                                          const unsigned long long *offsets, unsigned long long *planes, uint8_t *skip) {
  if (st->flags & (GMX_INGEST_TOO_MANY_LINES | GMX_INGEST_BAD_RECORD)) return;
  const uint32_t n_reads = st->n_reads, uniform = st->uniform_len;
  const uint32_t wpr = (st->max_len + 31u) / 32u + (uniform ? 0u : 1u);
  const unsigned long long items = (unsigned long long)n_reads * wpr;
  for (unsigned long long it = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(it / wpr), w = (uint32_t)(it - (unsigned long long)r * wpr);
    const uint32_t len = rec_len[r];
    unsigned long long first, count;
    if (uniform) {
      first = (unsigned long long)r * wpr;
      count = wpr;
    } else {
      const unsigned long long off = offsets[r];
      first = (off >> 5) + r;
      count = ((off + len) >> 5) - (off >> 5) + 1ull;
    }
    if (w >= count) continue;
    uint32_t lo = 0, hi = 0;
    bool ok = true;
    if (w * 32u < len) {
      const uint8_t *src = text + rec_start[r] + w * 32u;
      const uint32_t m = min(32u, len - w * 32u);
      for (uint32_t j = 0; j < m; ++j) {
        const uint32_t c = src[j], u = c & 0xDFu;
        ok = ok && (u == 0x41u || u == 0x43u || u == 0x47u || u == 0x54u);
        const uint32_t h = (c >> 2) & 1u;
        hi |= h << j;
        lo |= (((c >> 1) & 1u) ^ h) << j;
      }
    }
    planes[first + w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
    if (!ok) {
      skip[r] = 1;
      st->any_skip = 1;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
#define GMX_INGEST_SLOTS 3  // chunks in flight per ingest (include/gmx.h: slots 0, 1, 2; a caller may alternate between two of them)
struct gmx_ingest {
  int device = 0;
  uint64_t max_text = 0, max_comp = 0;
  uint32_t cap_reads = 0, cap_lines = 0, cap_members = 0, n_tiles_max = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  struct Slot {
    uint32_t *d_comp = nullptr;
    IngestMember *d_members = nullptr;
    uint8_t *d_text = nullptr;
    uint32_t *d_line_end = nullptr, *d_rec_start = nullptr, *d_rec_len = nullptr, *d_tiles = nullptr, *d_tile_blk = nullptr;
    unsigned long long *d_off_blk = nullptr;  // (the scans' block sums: gmx_tile_scan1/2_kernel, gmx_offsets1/2/3_kernel)
    unsigned long long *d_planes = nullptr, *d_offsets = nullptr;
    uint8_t *d_skip = nullptr;
    IngestState *d_state = nullptr, *h_state = nullptr;
    IngestMember *h_members = nullptr;  // page-locked staging of the member table
    IngestInflateStatus *d_inflate_status = nullptr;
    hipStream_t inflate_stream = nullptr;  // the slot's inflate kernel: beside the scan of the chunk before and the tail of its inflate kernel
    hipEvent_t copied = nullptr, done = nullptr, released = nullptr, released2 = nullptr, inflated = nullptr, carried = nullptr;
    bool in_flight = false, has_release = false, has_release2 = false, has_carried = false;
    hipEvent_t follower_carried = nullptr;  // `carried` of the chunk that continued this slot's: its carry kernel read the end of this slot's text
    bool has_follower = false;
    bool deferred = false;       // gmx_ingest_submit_bgzf_deferred / _text_deferred: uploaded (and inflating), scan still to come (gmx_ingest_scan)
    bool deferred_inflate = true;  // ... with an inflate kernel (false: the chunk arrived as text)
    uint32_t deferred_text = 0;  // ... bytes of text of its members
  } slot[GMX_INGEST_SLOTS];
  int last_slot = -1;  // the slot whose chunk the next one continues (-1: a file's first chunk)
  std::vector<void *> allocs;
  int check_crc = 1;
  uint32_t exp_mode = 0;
  unsigned long long *d_dbg = nullptr;  // GMX_INGEST_STATS=1: counters of gmx_inflate_kernel, printed by gmx_ingest_destroy
};

namespace {
template <class T>
int ing_alloc(gmx_ingest *g, T **p, size_t count, bool zero) {
  void *q = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  ING_TRY(hipMalloc(&q, bytes));
  if (zero) ING_TRY(hipMemset(q, 0, bytes));
  g->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return GMX_OK;
}
}  // namespace

extern "C" {

// The slots' inflate streams are created at the LOWEST stream priority: the runtime keeps a pool of hardware queues per priority
// (4 each by default), so the three of them get queues of their own — two streams on one hardware queue run their kernels one after the
// other, and chunk i + 1's inflate kernel then waits for chunk i's instead of filling the CUs its tail leaves idle — without the
// process asking for more queues (GPU_MAX_HW_QUEUES), which changes how an engine's own streams are spread (configs[3]'s host feed
// -12 % at 16). And it is the right order of precedence: scans and mapping kernels go first. GMX_INGEST_STREAM_PRIORITY=0: as the others.
static hipError_t ing_inflate_stream_create(hipStream_t *st) {
  int least = 0, greatest = 0;
  if (getenv("GMX_INGEST_STREAM_PRIORITY") && atoi(getenv("GMX_INGEST_STREAM_PRIORITY")) == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  }
  return hipStreamCreateWithPriority(st, hipStreamNonBlocking, least);
}

int gmx_ingest_create(int device, uint64_t max_text_bytes, gmx_ingest **out) try {
  if (!out || max_text_bytes < (1u << 16) || max_text_bytes > (3ull << 30)) {
    gmx_set_error("gmx_ingest_create: max_text_bytes must lie between 64 KB and 3 GB");
    return GMX_EINVAL;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
    (void)hipGetLastError();
    gmx_set_error("gmx_ingest_create: no such HIP device (reads are decoded on the GPU: there is no CPU fallback here)");
    return GMX_ENODEV;
  }
  ING_TRY(hipSetDevice(device));
  gmx_ingest *g = new gmx_ingest();
  g->device = device;
  g->max_text = max_text_bytes;
  g->max_comp = max_text_bytes / 2 + (1u << 20);
  g->cap_reads = (uint32_t)(max_text_bytes / 32 + 1024);
  g->cap_lines = 4u * g->cap_reads + 8u;
  g->cap_members = (uint32_t)(max_text_bytes / 512 + 1024);  // (members of half a KB of text on average, or larger)
  g->n_tiles_max = (uint32_t)((ING_CARRY_MAX + max_text_bytes + ING_TILE - 1) / ING_TILE);
  g->check_crc = getenv("GMX_INGEST_NO_CRC") ? 0 : 1;
  if (getenv("GMX_INGEST_EXP")) g->exp_mode = (uint32_t)atoi(getenv("GMX_INGEST_EXP"));
  if (getenv("GMX_INGEST_STATS") && ing_alloc(g, &g->d_dbg, 16, true) != GMX_OK) g->d_dbg = nullptr;
  int rc = GMX_OK;
  auto fail = [&](int code) {
    gmx_ingest_destroy(g);
    return code;
  };
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    gmx_set_error("gmx_ingest_create: hipStreamCreate failed");
    return fail(GMX_EHIP);
  }
  for (auto &s : g->slot) {
    if ((rc = ing_alloc(g, &s.d_comp, g->max_comp / 4 + 256, false)) || (rc = ing_alloc(g, &s.d_members, g->cap_members, false)) ||
        (rc = ing_alloc(g, &s.d_text, ING_CARRY_MAX + max_text_bytes + 64, false)) || (rc = ing_alloc(g, &s.d_line_end, g->cap_lines, false)) ||
        (rc = ing_alloc(g, &s.d_rec_start, g->cap_reads, false)) || (rc = ing_alloc(g, &s.d_rec_len, g->cap_reads, false)) ||
        (rc = ing_alloc(g, &s.d_tiles, g->n_tiles_max + 1, false)) || (rc = ing_alloc(g, &s.d_tile_blk, g->n_tiles_max / ING_SCAN_BLOCK + 2, false)) ||
        (rc = ing_alloc(g, &s.d_off_blk, (size_t)g->cap_reads / ING_SCAN_BLOCK + 2, false)) || (rc = ing_alloc(g, &s.d_planes, max_text_bytes / 16 + 2ull * g->cap_reads + 64, false)) ||
        (rc = ing_alloc(g, &s.d_offsets, (size_t)g->cap_reads + 1, false)) || (rc = ing_alloc(g, &s.d_skip, g->cap_reads, false)) ||
        (rc = ing_alloc(g, &s.d_state, 1, true)) || (rc = ing_alloc(g, &s.d_inflate_status, 1, true)))
      return fail(rc);
    if (hipHostMalloc(reinterpret_cast<void **>(&s.h_state), sizeof(IngestState), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&s.h_members), (size_t)g->cap_members * sizeof(IngestMember), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess ||
        hipEventCreateWithFlags(&s.released, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.released2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.inflated, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.carried, hipEventDisableTiming) != hipSuccess || ing_inflate_stream_create(&s.inflate_stream) != hipSuccess) {
      gmx_set_error("gmx_ingest_create: page-locked memory / events");
      return fail(GMX_EHIP);
    }
  }
  *out = g;
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_create")

void gmx_ingest_destroy(gmx_ingest *g) try {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
  if (g->d_dbg) {
    unsigned long long v[16] = {0};
    if (hipMemcpy(v, g->d_dbg, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess && v[10])
      fprintf(stderr, "[gmx_ingest] per member: %.0f look-ups, %.0f literal bytes, %.0f matches (%.0f bytes, %.1f far), %.1f bit-by-bit, %.2f blocks; kclocks %.0f total, %.0f tables, %.0f copies, %.0f crc (%llu members)\n",
              (double)v[0] / v[10], (double)v[1] / v[10], (double)v[2] / v[10], (double)v[11] / v[10], (double)v[3] / v[10], (double)v[4] / v[10], (double)v[5] / v[10],
              v[6] / 1e3 / v[10], v[7] / 1e3 / v[10], v[8] / 1e3 / v[10], v[9] / 1e3 / v[10], v[10]);
  }
  for (auto &s : g->slot) {
    if (s.h_state) (void)hipHostFree(s.h_state);
    if (s.h_members) (void)hipHostFree(s.h_members);
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.released) (void)hipEventDestroy(s.released);
    if (s.released2) (void)hipEventDestroy(s.released2);
    if (s.inflated) (void)hipEventDestroy(s.inflated);
    if (s.carried) (void)hipEventDestroy(s.carried);
    if (s.inflate_stream) {
      (void)hipStreamSynchronize(s.inflate_stream);
      (void)hipStreamDestroy(s.inflate_stream);
    }
  }
  for (void *p : g->allocs) (void)hipFree(p);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  (void)hipGetLastError();
  delete g;
} GMX_GUARD_VOID("gmx_ingest_destroy")

uint64_t gmx_ingest_max_text(const gmx_ingest *g) { return g ? g->max_text : 0; }
uint64_t gmx_ingest_max_compressed(const gmx_ingest *g) { return g ? g->max_comp : 0; }
uint64_t gmx_ingest_max_members(const gmx_ingest *g) { return g ? g->cap_members : 0; }

int gmx_ingest_reset(gmx_ingest *g) try {  // the next chunk starts a file: nothing is carried into it
  if (!g) {
    gmx_set_error("null ingest");
    return GMX_EINVAL;
  }
  g->last_slot = -1;
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_reset")

// The kernels behind a chunk's bytes. Three streams: the slot's own for its inflate kernel — so that it runs beside the scan of
// the chunk before and fills the CUs the tail of that chunk's inflate kernel leaves idle —, one for the scans (in chunk order:
// a chunk's carry needs the state of the chunk before), the copy stream. Orderings that are not a stream's own:
//   inflate(i) after the upload of chunk i, and after carry(i - 1): that kernel reads the END of chunk i - 2's text from this slot
//   scan(i) after inflate(i); the slot's next upload / inflate / pack after whatever read its planes (gmx_ingest_release_after)
static int ing_enqueue_inflate(gmx_ingest *g, int si, uint32_t n_members) {
  gmx_ingest::Slot &s = g->slot[si];
  ING_TRY(hipStreamWaitEvent(s.inflate_stream, s.copied, 0));
  ING_TRY(hipMemsetAsync(&s.d_inflate_status->flags, 0, 4, s.inflate_stream));
  ING_TRY(hipMemsetAsync(&s.d_inflate_status->bad_member, 0xFF, 4, s.inflate_stream));
  if (n_members) {
    if (g->d_dbg)
      hipLaunchKernelGGL(gmx_inflate_kernel<true>, dim3(n_members), dim3(64), 0, s.inflate_stream, s.d_comp, s.d_members, n_members, s.d_text + ING_CARRY_MAX,
                         s.d_inflate_status, g->check_crc, g->d_dbg, g->exp_mode);
    else
      hipLaunchKernelGGL(gmx_inflate_kernel<false>, dim3(n_members), dim3(64), 0, s.inflate_stream, s.d_comp, s.d_members, n_members, s.d_text + ING_CARRY_MAX,
                         s.d_inflate_status, g->check_crc, g->d_dbg, 0u);
  }
  ING_TRY(hipGetLastError());
  ING_TRY(hipEventRecord(s.inflated, s.inflate_stream));
  return GMX_OK;
}
// host_tail = ~0: the cut record comes from the chunk before on this device (g->last_slot); else that many bytes lie in front of the text
static int ing_enqueue_scan(gmx_ingest *g, int si, uint32_t members_text, int final_chunk, bool inflate, uint32_t n_members, uint32_t host_tail = 0xFFFFFFFFu) {
  gmx_ingest::Slot &s = g->slot[si];
  const bool from_device = host_tail == 0xFFFFFFFFu;
  gmx_ingest::Slot *prev = from_device && g->last_slot >= 0 ? &g->slot[g->last_slot] : nullptr;
  if (inflate && from_device) {
    int rc = ing_enqueue_inflate(g, si, n_members);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(gmx_carry_kernel, dim3(64), dim3(256), 0, g->stream, prev ? prev->d_state : nullptr, prev ? prev->d_text : nullptr, s.d_state, s.d_text,
                     members_text, (uint32_t)(final_chunk ? 1 : 0), from_device ? 0u : host_tail);
  ING_TRY(hipEventRecord(s.carried, g->stream));
  s.has_carried = true;
  if (prev) {  // (the slot of the chunk before may be overwritten once this kernel has read its end: ing_begin)
    prev->follower_carried = s.carried;
    prev->has_follower = true;
  }
  ING_TRY(hipStreamWaitEvent(g->stream, inflate ? s.inflated : s.copied, 0));
  const uint32_t n_tiles = (uint32_t)((ING_CARRY_MAX + (uint64_t)members_text + ING_TILE - 1) / ING_TILE);
  hipLaunchKernelGGL(gmx_nl_count_kernel, dim3(n_tiles), dim3(256), 0, g->stream, s.d_text, s.d_state, s.d_tiles);
  const uint32_t n_tile_blk = (n_tiles + ING_SCAN_BLOCK - 1) / ING_SCAN_BLOCK;
  hipLaunchKernelGGL(gmx_tile_scan1_kernel, dim3(n_tile_blk), dim3(64), 0, g->stream, s.d_tiles, n_tiles, s.d_tile_blk);
  hipLaunchKernelGGL(gmx_tile_scan2_kernel, dim3(1), dim3(64), 0, g->stream, s.d_tile_blk, n_tile_blk, s.d_state, g->cap_lines);
  hipLaunchKernelGGL(gmx_nl_mark_kernel, dim3(n_tiles), dim3(256), 0, g->stream, s.d_text, s.d_state, s.d_tiles, s.d_tile_blk, s.d_line_end, g->cap_lines);
  hipLaunchKernelGGL(gmx_records_kernel, dim3(2048), dim3(256), 0, g->stream, s.d_text, s.d_state, s.d_line_end, s.d_rec_start, s.d_rec_len, s.d_skip, g->cap_reads);
  const uint32_t n_off_blk = (uint32_t)((g->cap_reads + ING_SCAN_BLOCK - 1) / ING_SCAN_BLOCK);  // (the chunk's read count is on the device: blocks beyond it leave at once)
  hipLaunchKernelGGL(gmx_layout_kernel, dim3(1), dim3(64), 0, g->stream, s.d_state, inflate ? s.d_inflate_status : nullptr);
  hipLaunchKernelGGL(gmx_offsets1_kernel, dim3(n_off_blk), dim3(64), 0, g->stream, s.d_state, s.d_rec_len, s.d_offsets, s.d_off_blk);
  hipLaunchKernelGGL(gmx_offsets2_kernel, dim3(1), dim3(64), 0, g->stream, s.d_state, s.d_offsets, s.d_off_blk);
  hipLaunchKernelGGL(gmx_offsets3_kernel, dim3(n_off_blk), dim3(64), 0, g->stream, s.d_state, s.d_offsets, s.d_off_blk);
  hipLaunchKernelGGL(gmx_fq_pack_kernel, dim3(4096), dim3(256), 0, g->stream, s.d_text, s.d_state, s.d_rec_start, s.d_rec_len, s.d_offsets, s.d_planes, s.d_skip);
  ING_TRY(hipGetLastError());
  ING_TRY(hipMemcpyAsync(s.h_state, s.d_state, sizeof(IngestState), hipMemcpyDeviceToHost, g->stream));
  ING_TRY(hipEventRecord(s.done, g->stream));
  s.in_flight = true;
  s.deferred = false;
  g->last_slot = from_device ? si : -1;  // (a chunk scanned with a host-provided start carries nothing on the device)
  return GMX_OK;
}

static int ing_begin(gmx_ingest *g, int si, const char *who) {
  if (!g || si < 0 || si >= GMX_INGEST_SLOTS) {
    gmx_set_error(std::string(who) + ": null ingest or slot not 0 / 1 / 2");
    return GMX_EINVAL;
  }
  if (g->slot[si].in_flight || g->slot[si].deferred) {
    gmx_set_error(std::string(who) + ": the slot's chunk before has not been waited for (gmx_ingest_wait), or still awaits its gmx_ingest_scan");
    return GMX_EINVAL;
  }
  if (g->last_slot == si) {
    gmx_set_error(std::string(who) + ": the chunk before went to the same slot (the slots alternate: its text holds the start of this chunk's first record; gmx_ingest_reset starts a new file)");
    return GMX_EINVAL;
  }
  ING_TRY(hipSetDevice(g->device));
  gmx_ingest::Slot &s = g->slot[si];
  if (s.has_release) {  // the mapping kernels that read the slot's planes (gmx_ingest_release_after)
    ING_TRY(hipStreamWaitEvent(g->copy_stream, s.released, 0));
    ING_TRY(hipStreamWaitEvent(g->stream, s.released, 0));
    ING_TRY(hipStreamWaitEvent(s.inflate_stream, s.released, 0));
    s.has_release = false;
  }
  if (s.has_release2) {  // (an engine with two workspaces: the launches on its second stream)
    ING_TRY(hipStreamWaitEvent(g->copy_stream, s.released2, 0));
    ING_TRY(hipStreamWaitEvent(g->stream, s.released2, 0));
    ING_TRY(hipStreamWaitEvent(s.inflate_stream, s.released2, 0));
    s.has_release2 = false;
  }
  // this slot's text is about to be overwritten (inflate kernel, or the upload of a text chunk): the carry kernel of the chunk that
  // continued this slot's old chunk reads the end of that text. (With two slots that is the chunk before the one being submitted;
  // with three it ran long ago — the new chunk's inflate kernel does not wait for the scan of the chunk two before it.)
  if (s.has_follower) {
    ING_TRY(hipStreamWaitEvent(s.inflate_stream, s.follower_carried, 0));
    ING_TRY(hipStreamWaitEvent(g->copy_stream, s.follower_carried, 0));
    s.has_follower = false;
  }
  return GMX_OK;
}

int gmx_ingest_submit_bgzf(gmx_ingest *g, int slot, const uint8_t *compressed, uint64_t n_bytes, const gmx_bgzf_member *members, uint64_t n_members,
                           int final_chunk) try {
  int rc = ing_begin(g, slot, "gmx_ingest_submit_bgzf");
  if (rc) return rc;
  if ((!compressed && n_bytes) || (!members && n_members) || n_bytes > g->max_comp || n_members > g->cap_members) {
    gmx_set_error("gmx_ingest_submit_bgzf: null argument, or more compressed bytes / members than the ingest was created for");
    return GMX_EINVAL;
  }
  gmx_ingest::Slot &s = g->slot[slot];
  uint64_t text = 0;
  for (uint64_t i = 0; i < n_members; ++i) {
    const gmx_bgzf_member &m = members[i];
    if (m.offset + m.size > n_bytes || m.isize > (1u << 16)) {
      gmx_set_error("gmx_ingest_submit_bgzf: member " + std::to_string(i) + " lies outside the bytes given, or holds more than 64 KB of text");
      return GMX_EINVAL;
    }
    s.h_members[i] = IngestMember{(uint32_t)m.offset, (uint32_t)m.size, (uint32_t)text, m.isize, m.crc32, 0u};
    text += m.isize;
  }
  if (text > g->max_text) {
    gmx_set_error("gmx_ingest_submit_bgzf: the members hold more text than the ingest was created for");
    return GMX_EINVAL;
  }
  // (the bit reader fetches up to 128 words ahead: the staging buffer has slack; the words right behind the data are zeroed)
  if (n_bytes) ING_TRY(hipMemcpyAsync(s.d_comp, compressed, n_bytes, hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(s.d_comp) + n_bytes, 0, 16, g->copy_stream));
  if (n_members) ING_TRY(hipMemcpyAsync(s.d_members, s.h_members, n_members * sizeof(IngestMember), hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipEventRecord(s.copied, g->copy_stream));
  return ing_enqueue_scan(g, slot, (uint32_t)text, final_chunk, true, (uint32_t)n_members);
} GMX_GUARD_INT("gmx_ingest_submit_bgzf")

int gmx_ingest_submit_text(gmx_ingest *g, int slot, const uint8_t *text, uint64_t n_bytes, int final_chunk) try {
  int rc = ing_begin(g, slot, "gmx_ingest_submit_text");
  if (rc) return rc;
  if ((!text && n_bytes) || n_bytes > g->max_text) {
    gmx_set_error("gmx_ingest_submit_text: null text, or more text than the ingest was created for");
    return GMX_EINVAL;
  }
  gmx_ingest::Slot &s = g->slot[slot];
  if (n_bytes) ING_TRY(hipMemcpyAsync(s.d_text + ING_CARRY_MAX, text, n_bytes, hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipEventRecord(s.copied, g->copy_stream));
  return ing_enqueue_scan(g, slot, (uint32_t)n_bytes, final_chunk, false, 0);
} GMX_GUARD_INT("gmx_ingest_submit_text")

// Chunks dealt over several devices (one ingest per device): the cut record at a chunk's start lies on ANOTHER device, so a chunk
// is uploaded and inflated at once (_deferred) and scanned (gmx_ingest_scan) when the caller has the end of the chunk before —
// gmx_ingest_fetch_tail of that chunk's slot, on its device — which it hands over as host bytes.
int gmx_ingest_submit_bgzf_deferred(gmx_ingest *g, int slot, const uint8_t *compressed, uint64_t n_bytes, const gmx_bgzf_member *members, uint64_t n_members) try {
  if (g) g->last_slot = -1;  // (nothing is carried on the device in this mode: the slots need not alternate)
  int rc = ing_begin(g, slot, "gmx_ingest_submit_bgzf_deferred");
  if (rc) return rc;
  if ((!compressed && n_bytes) || (!members && n_members) || n_bytes > g->max_comp || n_members > g->cap_members) {
    gmx_set_error("gmx_ingest_submit_bgzf_deferred: null argument, or more compressed bytes / members than the ingest was created for");
    return GMX_EINVAL;
  }
  gmx_ingest::Slot &s = g->slot[slot];
  uint64_t text = 0;
  for (uint64_t i = 0; i < n_members; ++i) {
    const gmx_bgzf_member &m = members[i];
    if (m.offset + m.size > n_bytes || m.isize > (1u << 16)) {
      gmx_set_error("gmx_ingest_submit_bgzf_deferred: member " + std::to_string(i) + " lies outside the bytes given, or holds more than 64 KB of text");
      return GMX_EINVAL;
    }
    s.h_members[i] = IngestMember{(uint32_t)m.offset, (uint32_t)m.size, (uint32_t)text, m.isize, m.crc32, 0u};
    text += m.isize;
  }
  if (text > g->max_text) {
    gmx_set_error("gmx_ingest_submit_bgzf_deferred: the members hold more text than the ingest was created for");
    return GMX_EINVAL;
  }
  if (n_bytes) ING_TRY(hipMemcpyAsync(s.d_comp, compressed, n_bytes, hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(s.d_comp) + n_bytes, 0, 16, g->copy_stream));
  if (n_members) ING_TRY(hipMemcpyAsync(s.d_members, s.h_members, n_members * sizeof(IngestMember), hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipEventRecord(s.copied, g->copy_stream));
  rc = ing_enqueue_inflate(g, slot, (uint32_t)n_members);
  if (rc) return rc;
  s.deferred = true;
  s.deferred_inflate = true;
  s.deferred_text = (uint32_t)text;
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_submit_bgzf_deferred")

// The same for a chunk of plain text (an uncompressed FASTQ dealt over several devices): uploaded at once, scanned by gmx_ingest_scan.
int gmx_ingest_submit_text_deferred(gmx_ingest *g, int slot, const uint8_t *text, uint64_t n_bytes) try {
  if (g) g->last_slot = -1;
  int rc = ing_begin(g, slot, "gmx_ingest_submit_text_deferred");
  if (rc) return rc;
  if ((!text && n_bytes) || n_bytes > g->max_text) {
    gmx_set_error("gmx_ingest_submit_text_deferred: null text, or more text than the ingest was created for");
    return GMX_EINVAL;
  }
  gmx_ingest::Slot &s = g->slot[slot];
  if (n_bytes) ING_TRY(hipMemcpyAsync(s.d_text + ING_CARRY_MAX, text, n_bytes, hipMemcpyHostToDevice, g->copy_stream));
  ING_TRY(hipEventRecord(s.copied, g->copy_stream));
  s.deferred = true;
  s.deferred_inflate = false;
  s.deferred_text = (uint32_t)n_bytes;
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_submit_text_deferred")

int gmx_ingest_scan(gmx_ingest *g, int slot, const uint8_t *carry, uint64_t n_carry, int final_chunk) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS || !g->slot[slot].deferred || (!carry && n_carry) || n_carry > ING_CARRY_MAX) {
    gmx_set_error("gmx_ingest_scan: null ingest, slot without a gmx_ingest_submit_bgzf_deferred chunk, or more than 1 MB of carried text (a record that long is not FASTQ)");
    return GMX_EINVAL;
  }
  ING_TRY(hipSetDevice(g->device));
  gmx_ingest::Slot &s = g->slot[slot];
  // the carried bytes right in front of the members' text (pageable memory: the copy is over when the call returns)
  if (n_carry) ING_TRY(hipMemcpyAsync(s.d_text + ING_CARRY_MAX - n_carry, carry, n_carry, hipMemcpyHostToDevice, g->stream));
  return ing_enqueue_scan(g, slot, s.deferred_text, final_chunk, s.deferred_inflate, 0, (uint32_t)n_carry);
} GMX_GUARD_INT("gmx_ingest_scan")

int64_t gmx_ingest_fetch_tail(gmx_ingest *g, int slot, uint8_t *out, uint64_t cap) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS || g->slot[slot].in_flight) {
    gmx_set_error("gmx_ingest_fetch_tail: null ingest, bad slot, or the slot's chunk is still in flight");
    return GMX_EINVAL;
  }
  const IngestState &st = *g->slot[slot].h_state;
  if (!out) return (int64_t)st.tail_len;
  if (cap < st.tail_len) {
    gmx_set_error("gmx_ingest_fetch_tail: buffer too small");
    return GMX_EINVAL;
  }
  if (st.tail_len && (hipSetDevice(g->device) != hipSuccess ||
                      hipMemcpy(out, g->slot[slot].d_text + st.consumed, st.tail_len, hipMemcpyDeviceToHost) != hipSuccess)) {
    gmx_set_error("gmx_ingest_fetch_tail: hipMemcpy failed");
    return GMX_EHIP;
  }
  return (int64_t)st.tail_len;
} GMX_GUARD_INT("gmx_ingest_fetch_tail")

int gmx_ingest_wait(gmx_ingest *g, int slot, gmx_ingest_result *out) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS || !out) {
    gmx_set_error("gmx_ingest_wait: null argument or slot not 0 / 1 / 2");
    return GMX_EINVAL;
  }
  gmx_ingest::Slot &s = g->slot[slot];
  if (!s.in_flight) {
    gmx_set_error("gmx_ingest_wait: nothing was submitted to the slot");
    return GMX_EINVAL;
  }
  ING_TRY(hipSetDevice(g->device));
  ING_TRY(hipEventSynchronize(s.done));
  s.in_flight = false;
  const IngestState &st = *s.h_state;
  memset(out, 0, sizeof(*out));
  out->status = st.flags;
  out->bad_member = st.bad_member;
  out->n_reads = st.n_reads;
  out->n_bases = st.n_bases;
  out->uniform_len = st.uniform_len;
  out->any_skip = st.any_skip;
  out->n_pairs = st.n_pairs;
  out->text_bytes = st.text_len;
  out->consumed_bytes = st.consumed - st.text_start;
  out->tail_bytes = st.tail_len;
  for (int i = 0; i < 16; ++i) out->sub_pairs[i] = st.sub_pairs[i];
  out->d_planes = reinterpret_cast<const uint64_t *>(s.d_planes);
  out->d_offsets = st.uniform_len ? nullptr : reinterpret_cast<const uint64_t *>(s.d_offsets);
  out->d_skip = s.d_skip;
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_wait")

int gmx_ingest_release_after(gmx_ingest *g, int slot, void *hip_stream) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS) {
    gmx_set_error("gmx_ingest_release_after: null ingest or slot not 0 / 1 / 2");
    return GMX_EINVAL;
  }
  ING_TRY(hipSetDevice(g->device));
  gmx_ingest::Slot &s = g->slot[slot];
  if (s.has_release) {  // a second stream behind the same chunk
    ING_TRY(hipEventRecord(s.released2, (hipStream_t)hip_stream));
    s.has_release2 = true;
  } else {
    ING_TRY(hipEventRecord(s.released, (hipStream_t)hip_stream));
    s.has_release = true;
  }
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_release_after")

int64_t gmx_ingest_fetch_text(gmx_ingest *g, int slot, uint8_t *out, uint64_t cap) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS || g->slot[slot].in_flight) {
    gmx_set_error("gmx_ingest_fetch_text: null ingest, bad slot, or the slot's chunk is still in flight");
    return GMX_EINVAL;
  }
  const IngestState &st = *g->slot[slot].h_state;
  if (!out) return (int64_t)st.text_len;
  if (cap < st.text_len) {
    gmx_set_error("gmx_ingest_fetch_text: buffer too small");
    return GMX_EINVAL;
  }
  if (hipSetDevice(g->device) != hipSuccess || hipMemcpy(out, g->slot[slot].d_text + st.text_start, st.text_len, hipMemcpyDeviceToHost) != hipSuccess) {
    gmx_set_error("gmx_ingest_fetch_text: hipMemcpy failed");
    return GMX_EHIP;
  }
  return (int64_t)st.text_len;
} GMX_GUARD_INT("gmx_ingest_fetch_text")

int gmx_ingest_fetch_reads(gmx_ingest *g, int slot, uint64_t *planes, uint64_t *offsets, uint8_t *skip) try {
  if (!g || slot < 0 || slot >= GMX_INGEST_SLOTS || g->slot[slot].in_flight) {
    gmx_set_error("gmx_ingest_fetch_reads: null ingest, bad slot, or the slot's chunk is still in flight");
    return GMX_EINVAL;
  }
  const gmx_ingest::Slot &s = g->slot[slot];
  const IngestState &st = *s.h_state;
  ING_TRY(hipSetDevice(g->device));
  if (planes && st.n_pairs) ING_TRY(hipMemcpy(planes, s.d_planes, st.n_pairs * 8, hipMemcpyDeviceToHost));
  if (offsets && !st.uniform_len && st.n_reads) ING_TRY(hipMemcpy(offsets, s.d_offsets, ((size_t)st.n_reads + 1) * 8, hipMemcpyDeviceToHost));
  if (skip && st.n_reads) ING_TRY(hipMemcpy(skip, s.d_skip, st.n_reads, hipMemcpyDeviceToHost));
  return GMX_OK;
} GMX_GUARD_INT("gmx_ingest_fetch_reads")

}  // extern "C"
