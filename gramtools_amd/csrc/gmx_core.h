// gmx_core.h — the per-state vBWT backward-extension step on the flat index.
//
// Plain functions marked GMX_HD so that (a) the HIP search kernel inlines them per
// lane and (b) the host index builder uses the very same step to enumerate the
// k-mer seed table (the reference does this in `gram build`,
// libgramtools/src/build/kmer_index/build.cpp:18-131). The mapping entry points of
// the C-ABI never run these on the host: mapping always launches HIP kernels.
//
// A "Ctx" supplies the state pool and the path arena:
//   uint32_t n_states();                      void set_n_states(uint32_t);
//   void get(uint32_t s, uint32_t &lo, uint32_t &hi, uint32_t &tvd, uint32_t &tvg);
//   void put(uint32_t s, uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg);
//   bool push(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg);     // false = state pool full
//   uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next);      // GMX_NIL = arena full
//   uint32_t arena_site(uint32_t node);  uint32_t arena_next(uint32_t node);
//   void fail(uint32_t status);                                           // GMX_TASK_OVERFLOW / GMX_TASK_ERROR
#pragma once
#include "gmx_types.h"

GMX_HD uint32_t gmx_popc64(uint64_t x) {
  return (uint32_t)__builtin_popcountll(x);
}

// mask of in-block positions [0, r), r in [0, 128], split in two 64-bit words
GMX_HD void gmx_prefix_mask(uint32_t r, uint64_t &m0, uint64_t &m1) {
  m0 = r >= 64 ? ~0ull : ((1ull << r) - 1ull);
  m1 = r <= 64 ? 0ull : (r >= 128 ? ~0ull : ((1ull << (r - 64)) - 1ull));
}

// 64-bit words of the positions of a block holding base c (1..4)
GMX_HD void gmx_match_words(const GmxRankBlock &b, uint32_t c, uint64_t &w0, uint64_t &w1) {
  uint64_t l0 = b.lo[0], l1 = b.lo[1], h0 = b.hi[0], h1 = b.hi[1];
  switch (c) {
    case 1: w0 = ~l0 & ~h0 & ~b.mk[0]; w1 = ~l1 & ~h1 & ~b.mk[1]; break;  // A (and the sentinel)
    case 2: w0 = l0 & ~h0; w1 = l1 & ~h1; break;
    case 3: w0 = ~l0 & h0; w1 = ~l1 & h1; break;
    default: w0 = l0 & h0; w1 = l1 & h1; break;
  }
}

// count of base c in BWT[0, 128*b): cnt[] holds A(raw), C, G, M; T is derived
GMX_HD uint32_t gmx_block_base_count(const GmxRankBlock &b, uint32_t blk, uint32_t c) {
  // selects, not b.cnt[c - 1]: a runtime index would force the register copy of the block into scratch memory
  uint32_t a = b.cnt[0], cc = b.cnt[1], g = b.cnt[2];
  uint32_t t = (blk << GMX_BLK_SHIFT) - a - cc - g - b.cnt[3];
  return c == 1 ? a : (c == 2 ? cc : (c == 3 ? g : t));
}

// rank_c(i) = number of base c in BWT[0, i)  (dna_bwt_rank, BWT_search.cpp:8-22). Host/index-build use.
GMX_HD uint32_t gmx_rank(const GmxIndexView &ix, uint32_t i, uint32_t c) {
  uint32_t blk = i >> GMX_BLK_SHIFT, r = i & GMX_BLK_MASK;
  const GmxRankBlock &b = ix.blocks[blk];
  uint64_t w0, w1, m0, m1;
  gmx_match_words(b, c, w0, w1);
  gmx_prefix_mask(r, m0, m1);
  uint32_t res = gmx_block_base_count(b, blk, c) + gmx_popc64(w0 & m0) + gmx_popc64(w1 & m1);
  if (c == 1 && ix.sentinel_pos < i) res -= 1;  // the sentinel is stored as code 00
  return res;
}

// ---------------------------------------------------------------------------
// Jump program interpreter: applies the pre-resolved closure of one marker hit
// to the state (tvd, tvg) and appends the committed states.
// Reference: search_state_vBWT_jumps + extend_targets_* (vBWT_jump.cpp:134-265).
// ---------------------------------------------------------------------------
template <class Ctx>
GMX_HD void gmx_run_program(const GmxIndexView &ix, uint32_t off, uint32_t tvd0, uint32_t tvg0, Ctx &ctx) {
  const uint32_t *p = ix.prog + off;
  uint32_t n_out = *p++;
  for (uint32_t o = 0; o < n_out; ++o) {
    uint32_t n_ops = *p++;
    uint32_t tvd = tvd0, tvg = tvg0;
    bool ok = true;
    for (uint32_t k = 0; k < n_ops; ++k) {
      uint32_t op = p[0], site = p[1];
      int32_t allele = (int32_t)p[2];
      p += 3;
      if (!ok) continue;
      if (op == GMX_OP_EXIT) {  // update_variant_site_path, vBWT_jump.cpp:51-69
        if (tvg != GMX_NIL) {
          if (ctx.arena_site(tvg) != site) {  // reference asserts existing_locus.first == site_ID
            ctx.fail(GMX_TASK_ERROR);
            ok = false;
            continue;
          }
          tvg = ctx.arena_next(tvg);
        }
        uint32_t nn = ctx.arena_new(site, allele, tvd);
        if (nn == GMX_NIL) {
          ctx.fail(GMX_TASK_OVERFLOW);
          ok = false;
          continue;
        }
        tvd = nn;
      } else {  // GMX_OP_ENTER, vBWT_jump.cpp:29-44
        uint32_t nn = ctx.arena_new(site, -1, tvg);
        if (nn == GMX_NIL) {
          ctx.fail(GMX_TASK_OVERFLOW);
          ok = false;
          continue;
        }
        tvg = nn;
      }
    }
    uint32_t lo = p[0], hi = p[1];
    p += 2;
    if (ok && !ctx.push(lo, hi, tvd, tvg)) ctx.fail(GMX_TASK_OVERFLOW);
  }
}

// Marker pass over one state's interval [lo, hi] (left_markers_search + jumps, vBWT_jump.cpp:94-132).
// New states are appended to the pool; the scanned state is untouched.
// `b_lo` is the already-loaded block of `lo` (the LF step needs it anyway).
template <class Ctx>
GMX_HD void gmx_marker_pass(const GmxIndexView &ix, uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg,
                            const GmxRankBlock &b_lo, Ctx &ctx) {
  uint32_t blk_lo = lo >> GMX_BLK_SHIFT, blk_hi = hi >> GMX_BLK_SHIFT;
  for (uint32_t blk = blk_lo; blk <= blk_hi; ++blk) {
    uint64_t k0, k1;
    uint32_t mbase;
    if (blk == blk_lo) {
      k0 = b_lo.mk[0];
      k1 = b_lo.mk[1];
      mbase = b_lo.cnt[3];
    } else {
      const GmxRankBlock &b = ix.blocks[blk];
      k0 = b.mk[0];
      k1 = b.mk[1];
      mbase = b.cnt[3];
    }
    if ((k0 | k1) == 0) continue;
    // restrict to [lo, hi]
    uint32_t r_lo = blk == blk_lo ? (lo & GMX_BLK_MASK) : 0;
    uint32_t r_hi = blk == blk_hi ? (hi & GMX_BLK_MASK) + 1 : 128;
    uint64_t a0, a1, z0, z1;
    gmx_prefix_mask(r_lo, a0, a1);
    gmx_prefix_mask(r_hi, z0, z1);
    uint64_t s0 = k0 & z0 & ~a0, s1 = k1 & z1 & ~a1;
    // ascending BWT index order (the reference pushes hits in ascending order, then pops LIFO; the
    // resulting list order is irrelevant to coverage)
    while (s0) {
      uint32_t bit = (uint32_t)__builtin_ctzll(s0);
      s0 &= s0 - 1;
      uint32_t h = mbase + gmx_popc64(k0 & ((1ull << bit) - 1ull));
      gmx_run_program(ix, ix.hit_prog[ix.hit_perm[h]], tvd, tvg, ctx);
    }
    uint32_t c0 = gmx_popc64(k0);
    while (s1) {
      uint32_t bit = (uint32_t)__builtin_ctzll(s1);
      s1 &= s1 - 1;
      uint32_t h = mbase + c0 + gmx_popc64(k1 & ((1ull << bit) - 1ull));
      gmx_run_program(ix, ix.hit_prog[ix.hit_perm[h]], tvd, tvg, ctx);
    }
  }
}

// LF step of one interval (base_next_sa_interval + validity test, BWT_search.cpp:28-76).
// Returns false when the extended interval is empty (the state is dropped).
// `b_lo` is the block of `lo`; the block of `hi` is fetched only when it differs.
GMX_HD bool gmx_lf(const GmxIndexView &ix, uint32_t c, uint32_t &lo, uint32_t &hi, const GmxRankBlock &b_lo) {
  uint32_t blk_lo = lo >> GMX_BLK_SHIFT, blk_hi = hi >> GMX_BLK_SHIFT;
  uint32_t r_lo = lo & GMX_BLK_MASK, r_hi = (hi & GMX_BLK_MASK) + 1;
  uint64_t w0, w1, a0, a1, z0, z1;
  gmx_match_words(b_lo, c, w0, w1);
  gmx_prefix_mask(r_lo, a0, a1);
  uint32_t rank_lo = gmx_block_base_count(b_lo, blk_lo, c) + gmx_popc64(w0 & a0) + gmx_popc64(w1 & a1);
  uint32_t rank_hi1;  // rank_c(hi + 1)
  gmx_prefix_mask(r_hi, z0, z1);
  if (blk_hi == blk_lo) {
    rank_hi1 = rank_lo + gmx_popc64(w0 & z0 & ~a0) + gmx_popc64(w1 & z1 & ~a1);
  } else {
    const GmxRankBlock &b_hi = ix.blocks[blk_hi];
    uint64_t v0, v1;
    gmx_match_words(b_hi, c, v0, v1);
    rank_hi1 = gmx_block_base_count(b_hi, blk_hi, c) + gmx_popc64(v0 & z0) + gmx_popc64(v1 & z1);
  }
  if (c == 1) {  // sentinel correction (stored as code 00, not a base)
    if (ix.sentinel_pos < lo) rank_lo -= 1;
    if (ix.sentinel_pos <= hi) rank_hi1 -= 1;
  }
  if (rank_lo == rank_hi1) return false;  // next.first - 1 == next.second in uint32 arithmetic
  lo = ix.C[c] + rank_lo;
  hi = ix.C[c] + rank_hi1 - 1;
  return true;
}

// One backward-extension step of every state in the pool by base c
// (process_read_char_search_states, quasimap.cpp:258-268): marker pass on the states present at entry,
// then the LF step on all states including the freshly created ones; dead states are compacted away.
// With `skip_marker_pass` the step is the k-mer index's first base (build.cpp:23-27).
template <class Ctx>
GMX_HD void gmx_extend(const GmxIndexView &ix, uint32_t c, Ctx &ctx, bool skip_marker_pass = false) {
  uint32_t n0 = ctx.n_states();
  uint32_t w = 0;  // write cursor for survivors among the first n0 states
  for (uint32_t s = 0; s < n0; ++s) {
    uint32_t lo, hi, tvd, tvg;
    ctx.get(s, lo, hi, tvd, tvg);
    const GmxRankBlock b = ix.blocks[lo >> GMX_BLK_SHIFT];
    if (!skip_marker_pass) gmx_marker_pass(ix, lo, hi, tvd, tvg, b, ctx);
    if (gmx_lf(ix, c, lo, hi, b)) {
      ctx.put(w, lo, hi, tvd, tvg);
      ++w;
    }
  }
  uint32_t n1 = ctx.n_states();
  for (uint32_t s = n0; s < n1; ++s) {
    uint32_t lo, hi, tvd, tvg;
    ctx.get(s, lo, hi, tvd, tvg);
    const GmxRankBlock b = ix.blocks[lo >> GMX_BLK_SHIFT];
    if (gmx_lf(ix, c, lo, hi, b)) {
      ctx.put(w, lo, hi, tvd, tvg);
      ++w;
    }
  }
  ctx.set_n_states(w);
}
