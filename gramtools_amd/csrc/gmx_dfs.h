// gmx_dfs.h — the per-lane search loop of the HIP kernels: a depth-first work queue.
//
// search_read_backwards (libgramtools/src/genotype/quasimap/quasimap.cpp:227-256) advances ALL states of a
// read one base at a time. States never interact, so a lane may instead carry ONE state in registers down the
// read and keep the others (siblings created at variant markers, extra seed states) on a small LIFO stack:
//
//   every iteration = one small fetch (a rank block, a marker-hit record, or 32 symbols of the PRG itself once
//   the state has narrowed to a single suffix-array position) + register arithmetic.
//
// No second dependent load inside an iteration, no per-wave serialisation of the rare marker path: a marker
// hit only pushes {hit rank, path handles, position}; it is resolved when popped — by then as the lane's
// one fetch of that iteration. Final coverage does not depend on the order in which states are explored
// (HISTORY.md §4), and every state is explored exactly as the reference would.
//
// Ctx interface (FastCtx in gmx_engine.hip on the device, EmuDfsCtx in tests/hostemu on the host):
//   bool pop(a, b, tvd, tvg, pos, mode)          next pending entry
//   bool push(a, b, tvd, tvg, pos, mode)         false = stack full
//   bool emit(lo, hi, tvd, tvg)                  a state reached the stop position; false = output full
//   arena_new / arena_site / arena_next / fail / status   as in gmx_core.h
#pragma once
#include "gmx_core.h"

#define GMX_MODE_STATE 0u  // marker pass, then LF with the base left of `pos`
#define GMX_MODE_LF 1u     // LF only (state fresh from a general jump program; reference: appended states, vBWT_jump.cpp:119-132)
#define GMX_MODE_HIT 2u    // unresolved marker hit: a = index of its record in hits[]

// adapter: states produced by the general jump-program interpreter go on the stack as LF-only entries
template <class Ctx>
struct GmxDfsProgSink {
  Ctx &ctx;
  uint32_t pos;
  GMX_HD bool push(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) { return ctx.push(lo, hi, tvd, tvg, pos, GMX_MODE_LF); }
  GMX_HD uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) { return ctx.arena_new(site, allele, next); }
  GMX_HD uint32_t arena_site(uint32_t h) { return ctx.arena_site(h); }
  GMX_HD uint32_t arena_next(uint32_t h) { return ctx.arena_next(h); }
  GMX_HD void fail(uint32_t s) { ctx.fail(s); }
};

// marker bits of [lo, hi] -> one pending HIT entry each (left_markers_search, vBWT_jump.cpp:94-117)
template <class Ctx>
GMX_HD void gmx_dfs_push_hits(const GmxIndexView &ix, uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg, uint32_t pos,
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
    uint32_t r_lo = blk == blk_lo ? (lo & GMX_BLK_MASK) : 0;
    uint32_t r_hi = blk == blk_hi ? (hi & GMX_BLK_MASK) + 1 : 128;
    uint64_t a0, a1, z0, z1;
    gmx_prefix_mask(r_lo, a0, a1);
    gmx_prefix_mask(r_hi, z0, z1);
    uint64_t s0 = k0 & z0 & ~a0, s1 = k1 & z1 & ~a1;
    while (s0) {
      uint32_t bit = (uint32_t)__builtin_ctzll(s0);
      s0 &= s0 - 1;
      uint32_t h = ix.hit_perm[mbase + gmx_popc64(k0 & ((1ull << bit) - 1ull))];
      if (!ctx.push(h, 0, tvd, tvg, pos, GMX_MODE_HIT)) ctx.fail(GMX_TASK_OVERFLOW);
    }
    uint32_t c0 = gmx_popc64(k0);
    while (s1) {
      uint32_t bit = (uint32_t)__builtin_ctzll(s1);
      s1 &= s1 - 1;
      uint32_t h = ix.hit_perm[mbase + c0 + gmx_popc64(k1 & ((1ull << bit) - 1ull))];
      if (!ctx.push(h, 0, tvd, tvg, pos, GMX_MODE_HIT)) ctx.fail(GMX_TASK_OVERFLOW);
    }
  }
}

// The lane's current entry.
struct GmxLane {
  uint32_t a, b, tvd, tvg, pos, mode;
  bool have;
};
#define GMX_MODE_DEAD 3u  // the fast path found the state dead: the general path only has to pop the next entry

// One GENERAL iteration: handles every case (emit at the stop position, marker hits, wide intervals, programs).
template <class Ctx, class Reader>
GMX_HD void gmx_dfs_slow_iter(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop, GmxLane &ln) {
  if (ln.mode == GMX_MODE_DEAD) {
    ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
    return;
  }
  if (ln.pos <= stop) {  // parked / seed already at the stop position
    if (ln.mode == GMX_MODE_HIT) {
      ctx.fail(GMX_TASK_ERROR);  // hits are only created for positions > stop
      ln.have = false;
      return;
    }
    if (!ctx.emit(ln.a, ln.b, ln.tvd, ln.tvg)) ctx.fail(GMX_TASK_OVERFLOW);
    ln.have = ctx.status == GMX_TASK_MAPPED && ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
    return;
  }
  if (ln.mode == GMX_MODE_STATE && ln.b == GMX_TEXT_MARK) {  // text-form states only ever take the fast iteration
    ctx.fail(GMX_TASK_ERROR);
    ln.have = false;
    return;
  }
  const uint32_t c = rd.at(ln.pos - 1);
  bool alive = false;
  if (ln.mode == GMX_MODE_HIT) {
    const GmxHitSub hs = ix.hits[ln.a].sub[c - 1];
    const uint32_t kind = hs.head & 3u;
    if (kind == GMX_HIT_EXIT) {  // update_variant_site_path + exiting_site_search_state, vBWT_jump.cpp:51-92
      bool ok = true;
      if (ln.tvg != GMX_NIL) {
        if (ctx.arena_site(ln.tvg) != hs.site) {
          ctx.fail(GMX_TASK_ERROR);
          ok = false;
        } else
          ln.tvg = ctx.arena_next(ln.tvg);
      }
      if (ok) {
        uint32_t nn = ctx.arena_new(hs.site, (int32_t)hs.y, ln.tvd);
        if (nn == GMX_NIL) {
          ctx.fail(GMX_TASK_OVERFLOW);
        } else {
          ln.tvd = nn;
          alive = (hs.head & GMX_HITF_ALIVE) != 0;  // c is the only base that can precede the site marker
          ln.a = hs.x;
          ln.b = GMX_TEXT_MARK;
        }
      }
    } else if (kind == GMX_HIT_FUSED && ln.pos - 1u > stop) {
      uint32_t nn = ctx.arena_new(hs.site, (int32_t)hs.y, ln.tvd);
      if (nn == GMX_NIL) {
        ctx.fail(GMX_TASK_OVERFLOW);
      } else {
        ln.tvd = nn;
        alive = true;
        ln.a = hs.x - (hs.head >> 4);
        ln.b = GMX_TEXT_MARK;
      }
    } else if (kind == GMX_HIT_ENTER || kind == GMX_HIT_FUSED) {  // entering_site_search_state, vBWT_jump.cpp:29-44
      uint32_t nn = ctx.arena_new(hs.site, -1, ln.tvg);
      if (nn == GMX_NIL) {
        ctx.fail(GMX_TASK_OVERFLOW);
      } else {
        ln.tvg = nn;
        alive = (hs.head & GMX_HITF_ALIVE) != 0;
        ln.a = hs.x;
        ln.b = (hs.head & GMX_HITF_TEXT) ? GMX_TEXT_MARK : hs.y;
      }
    } else {  // general jump program: its outputs still need their LF step -> pushed as LF-only entries
      GmxDfsProgSink<Ctx> sink{ctx, ln.pos};
      gmx_run_program(ix, hs.site, ln.tvd, ln.tvg, sink);
    }
  } else {
    const GmxRankBlock blk = ix.blocks[ln.a >> GMX_BLK_SHIFT];
    if (ln.mode == GMX_MODE_STATE) gmx_dfs_push_hits(ix, ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, blk, ctx);
    alive = gmx_lf(ix, c, ln.a, ln.b, blk);
  }
  if (ctx.status != GMX_TASK_MAPPED) {
    ln.have = false;
    return;
  }
  if (alive) {
    --ln.pos;
    ln.mode = GMX_MODE_STATE;
  } else {
    ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
  }
}

// The FAST iterations cover what makes up almost every step of a read on a flat PRG. A lane is in exactly one
// of three situations (gmx_dfs_fast_kind), each costing one small fetch and register arithmetic:
//   CONVERT  a width-one interval [i, i]: the state switches to TEXT FORM (a = SA[i], b = GMX_TEXT_MARK);
//   TEXT     a text-form state compares up to 64 read bases against one GmxTextRec (and resolves inline sites). Equivalent to that many
//            backward steps of the reference: for a single SA position i the LF step with base c succeeds iff
//            BWT[i] == c, BWT[i] = PRG[SA[i] - 1], and the next position is SA[i] - 1 (BWT_search.cpp:28-76);
//            a variant marker left of the position is the marker hit left_markers_search would report
//            (vBWT_jump.cpp:94-117), and the state's own LF step dies on it (a marker is not a base);
//   HIT      a pending hit whose record is pre-resolved (GMX_HIT_EXIT / GMX_HIT_ENTER) and whose traversing
//            path is empty or inline: path update + the precomputed LF result.
// No stack traffic, no calls. A fast HIT iteration returns false, leaving the lane untouched, when the general
// iteration is needed.
#define GMX_FAST_NONE 0u
#define GMX_FAST_HIT 1u
#define GMX_FAST_TEXT 2u
#define GMX_FAST_CONVERT 3u
#define GMX_FAST_WIDE 4u  // an interval inside one rank block: marker check + LF step from that one line
#define GMX_FAST_EMIT 5u  // the state reached the stop position: publish it, take the next pending entry
#define GMX_FAST_POP 6u   // the state died: take the next pending entry
GMX_HD uint32_t gmx_dfs_fast_kind(const GmxLane &ln, uint32_t stop) {
  if (!ln.have) return GMX_FAST_NONE;
  if (ln.mode == GMX_MODE_DEAD) return GMX_FAST_POP;
  if (ln.pos <= stop) return ln.mode == GMX_MODE_HIT ? GMX_FAST_NONE : GMX_FAST_EMIT;
  if (ln.mode == GMX_MODE_HIT) return GMX_FAST_HIT;
  if (ln.mode != GMX_MODE_STATE) return GMX_FAST_NONE;
  if (ln.b == GMX_TEXT_MARK) return GMX_FAST_TEXT;
  if (ln.a == ln.b) return GMX_FAST_CONVERT;
  return (ln.a >> GMX_BLK_SHIFT) == (ln.b >> GMX_BLK_SHIFT) ? GMX_FAST_WIDE : GMX_FAST_NONE;
}
template <class Ctx>
GMX_HD void gmx_dfs_emit(Ctx &ctx, GmxLane &ln) {
  if (!ctx.emit(ln.a, ln.b, ln.tvd, ln.tvg)) ctx.fail(GMX_TASK_OVERFLOW);
  ln.have = ctx.status == GMX_TASK_MAPPED && ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
}
template <class Ctx>
GMX_HD void gmx_dfs_pop(Ctx &ctx, GmxLane &ln) {
  ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
}
GMX_HD uint32_t gmx_bitrev32(uint32_t v) {
#if defined(__clang__)
  return __builtin_bitreverse32(v);  // v_bfrev_b32
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(v);
#endif
}
GMX_HD uint64_t gmx_bitrev64(uint64_t v) { return ((uint64_t)gmx_bitrev32((uint32_t)v) << 32) | gmx_bitrev32((uint32_t)(v >> 32)); }
GMX_HD uint64_t gmx_below64(uint32_t s) { return s >= 64u ? ~0ull : ((1ull << s) - 1ull); }  // bits of slots < s
// TEXT: which record and which 64 raw read bases the iteration needs.
//   q = a - 1 is the PRG position left of the state, t its slot in the record. Slot s (<= t) is compared with
//   oriented read base pos - 1 - (t - s).
//   forward read: that is raw base (pos - 1 - t) + s   -> window from max(pos - 1 - t, 0), shifted up by `shift`
//   reverse-complement read: raw base (len - pos) + (t - s), complemented -> window from len - pos, bit-reversed
GMX_HD uint32_t gmx_dfs_text_rec(const GmxLane &ln) { return (ln.a ? ln.a - 1u : 0u) >> GMX_TEXT_SHIFT; }
GMX_HD void gmx_dfs_text_window(const GmxLane &ln, uint32_t len, bool rc, uint32_t &start, uint32_t &shift) {
  const uint32_t t = (ln.a ? ln.a - 1u : 0u) & GMX_TEXT_MASK;
  if (rc) {
    start = len - ln.pos;
    shift = 0;
  } else if (ln.pos > t) {
    start = ln.pos - 1u - t;
    shift = 0;
  } else {
    start = 0;
    shift = t + 1u - ln.pos;
  }
}
// the read's planes aligned with the record: bit s = oriented read base pos - 1 - (t - s) (bits outside the read: anything)
template <class Reader>
GMX_HD void gmx_dfs_text_read_planes(const GmxLane &ln, Reader &rd, uint64_t &rlo, uint64_t &rhi) {
  uint32_t start, shift, l0, h0, l1, h1;
  gmx_dfs_text_window(ln, rd.len, rd.rc, start, shift);
  rd.planes(start, l0, h0);
  rd.planes(start + 32u, l1, h1);
  const uint64_t xlo = (uint64_t)l0 | ((uint64_t)l1 << 32), xhi = (uint64_t)h0 | ((uint64_t)h1 << 32);
  if (rd.rc) {
    const uint32_t t = (ln.a - 1u) & GMX_TEXT_MASK;
    rlo = ~(gmx_bitrev64(xlo) >> (63u - t));
    rhi = ~(gmx_bitrev64(xhi) >> (63u - t));
  } else {
    rlo = xlo << shift;
    rhi = xhi << shift;
  }
}
// One text iteration on record `rec` (the record of ln.a - 1): compares up to 64 bases, resolves the inline sites met on
// the way (gmx_types.h) and stops at the first mismatch (state dead), at a marker that needs its record (the lane becomes
// that marker's pending hit), at the record's lower end or at the stop position. Returns false — lane untouched since the
// last inline site it finished — when a path node could not be allocated (the general iteration reports the overflow).
template <class Ctx, class Reader>
GMX_HD bool gmx_dfs_text_apply(Ctx &ctx, GmxLane &ln, uint32_t stop, Reader &rd, const GmxTextRec &rec) {
  if (ln.a == 0) {  // PRG start: BWT holds the sentinel, no base extends the match
    ln.mode = GMX_MODE_DEAD;
    return true;
  }
  // the read's planes aligned with the record, extracted ONCE: behind an inline site the same planes serve, shifted down by
  // the PRG symbols the site took beyond the one read base it consumed (slot s then holds what slot s + shift held)
  uint64_t rlo, rhi;
  gmx_dfs_text_read_planes(ln, rd, rlo, rhi);
  for (;;) {
    const uint32_t t = (ln.a - 1u) & GMX_TEXT_MASK;
    const uint32_t avail = ln.pos - stop;
    const uint32_t n = avail < t + 1u ? avail : t + 1u;
    const uint64_t range = gmx_below64(n) << (t + 1u - n);
    const uint64_t events = (((rec.lo ^ rlo) | (rec.hi ^ rhi)) | rec.mk) & range;
    if (events == 0) {
      ln.a -= n;
      ln.pos -= n;
      return true;
    }
    const uint32_t e = 63u - (uint32_t)__builtin_clzll(events);  // nearest slot with a marker or a mismatch
    ln.a -= t - e;
    ln.pos -= t - e;
    if (!((rec.mk >> e) & 1ull)) {
      ln.mode = GMX_MODE_DEAD;
      return true;
    }
    if (((rec.hi >> e) & 1ull) && ln.pos - 1u > stop) {  // closing marker of an inline site, bases left behind the allele
      const uint64_t opens = rec.mk & rec.lo & gmx_below64(e);
      const uint32_t o = 63u - (uint32_t)__builtin_clzll(opens);  // its opening marker (in this record by construction)
      const uint64_t clo = ((rlo >> e) & 1ull) ? ~0ull : 0ull, chi = ((rhi >> e) & 1ull) ? ~0ull : 0ull;  // the next read base
      const uint64_t alleles = ~rec.mk & gmx_below64(e) & ~gmx_below64(o + 1u);
      const uint64_t match = ~((rec.lo ^ clo) | (rec.hi ^ chi)) & alleles;
      if (match == 0) {
        ln.mode = GMX_MODE_DEAD;  // no allele of the site is that base (every sub-record of the marker is a dead ENTER)
        --ln.pos;
        return true;
      }
      const uint32_t s = (uint32_t)__builtin_ctzll(match);
      const uint32_t site = 5u + 2u * (rec.srank + (uint32_t)__builtin_popcountll(rec.mk & rec.lo & gmx_below64(o)));
      const uint32_t nn = ctx.arena_new(site, (int32_t)((s - o - 1u) >> 1), ln.tvd);
      if (nn == GMX_NIL) {  // as a pending hit: the general iteration finds the arena full and reports it
        ln.a = rec.mrank + (uint32_t)__builtin_popcountll(rec.mk & gmx_below64(e));
        ln.b = 0;
        ln.mode = GMX_MODE_HIT;
        return false;
      }
      ln.tvd = nn;
      ln.a -= e - o + 1u;  // left of the opening marker
      --ln.pos;
      if (o == 0 || ln.pos <= stop) return true;  // the record is used up (or the read: cannot be, two bases were left)
      rlo >>= e - o;  // (2 <= e - o <= 63: the site's symbols between its markers, plus one)
      rhi >>= e - o;
      continue;
    }
    ln.a = rec.mrank + (uint32_t)__builtin_popcountll(rec.mk & gmx_below64(e));
    ln.b = 0;
    ln.mode = GMX_MODE_HIT;
    return true;
  }
}

// HIT: `hs` = the record's sub-record for the next read base (gmx_dfs_hit_sub)
template <class Reader>
GMX_HD const GmxHitSub *gmx_dfs_hit_sub(const GmxIndexView &ix, Reader &rd, const GmxLane &ln) {
  return &ix.hits[ln.a].sub[rd.at(ln.pos - 1) - 1u];
}
template <class Ctx>
GMX_HD bool gmx_dfs_fast_hit(Ctx &ctx, GmxLane &ln, uint32_t stop, const GmxHitSub &hs) {
  const uint32_t kind = hs.head & 3u;
  if (kind == GMX_HIT_EXIT) {  // update_variant_site_path + exiting_site_search_state, vBWT_jump.cpp:51-92
    uint32_t rest = GMX_NIL;  // the traversing path without its innermost site (a nested path: one node load)
    if (ln.tvg != GMX_NIL) {
      if (ctx.arena_site(ln.tvg) != hs.site) return false;  // the general path reports the inconsistency
      rest = ctx.arena_next(ln.tvg);
    }
    uint32_t nn = ctx.arena_new(hs.site, (int32_t)hs.y, ln.tvd);
    if (nn == GMX_NIL) return false;
    ln.tvg = rest;
    ln.tvd = nn;
    ln.a = hs.x;
    ln.b = GMX_TEXT_MARK;
  } else if (kind == GMX_HIT_FUSED && ln.pos - 1u > stop) {  // ENTER, then EXIT of the one-base allele entered
    uint32_t nn = ctx.arena_new(hs.site, (int32_t)hs.y, ln.tvd);
    if (nn == GMX_NIL) return false;
    ln.tvd = nn;
    ln.a = hs.x - (hs.head >> 4);
    ln.b = GMX_TEXT_MARK;
  } else if (kind == GMX_HIT_ENTER || kind == GMX_HIT_FUSED) {  // entering_site_search_state, vBWT_jump.cpp:29-44
    const uint32_t nn = ctx.arena_new(hs.site, -1, ln.tvg);  // (inline handle for a first site, a node for a nested one)
    if (nn == GMX_NIL) return false;
    ln.tvg = nn;
    ln.a = hs.x;
    ln.b = (hs.head & GMX_HITF_TEXT) ? GMX_TEXT_MARK : hs.y;
  } else
    return false;
  if (hs.head & GMX_HITF_ALIVE) {
    --ln.pos;
    ln.mode = GMX_MODE_STATE;
  } else
    ln.mode = GMX_MODE_DEAD;
  return true;
}

// WIDE: `w` = the 16 words of the rank block holding [a, b] (counts | low plane | high plane | marker plane).
// A marker inside the interval needs the general path. Same arithmetic as gmx_lf, on 32-bit words.
GMX_HD uint32_t gmx_prefix32(uint32_t r, uint32_t word) {  // bits of 32-bit word `word` that lie below block position r
  const int32_t t = (int32_t)r - (int32_t)(32u * word);
  return t <= 0 ? 0u : (t >= 32 ? ~0u : ((1u << t) - 1u));
}
template <class Reader>
GMX_HD bool gmx_dfs_fast_wide(const GmxIndexView &ix, Reader &rd, GmxLane &ln, const uint32_t *w) {
  const uint32_t r_lo = ln.a & GMX_BLK_MASK, r_hi = (ln.b & GMX_BLK_MASK) + 1u;
  const uint32_t a0 = gmx_prefix32(r_lo, 0), a1 = gmx_prefix32(r_lo, 1), a2 = gmx_prefix32(r_lo, 2), a3 = gmx_prefix32(r_lo, 3);
  const uint32_t z0 = gmx_prefix32(r_hi, 0), z1 = gmx_prefix32(r_hi, 1), z2 = gmx_prefix32(r_hi, 2), z3 = gmx_prefix32(r_hi, 3);
  if ((w[12] & z0 & ~a0) | (w[13] & z1 & ~a1) | (w[14] & z2 & ~a2) | (w[15] & z3 & ~a3)) return false;
  const uint32_t c = rd.at(ln.pos - 1);
  const uint32_t code = c - 1u;
  const uint32_t xl = (code & 1u) ? 0u : ~0u, xh = (code & 2u) ? 0u : ~0u, ka = c == 1 ? ~0u : 0u;
  const uint32_t m0 = (w[4] ^ xl) & (w[8] ^ xh) & ~(w[12] & ka);
  const uint32_t m1 = (w[5] ^ xl) & (w[9] ^ xh) & ~(w[13] & ka);
  const uint32_t m2 = (w[6] ^ xl) & (w[10] ^ xh) & ~(w[14] & ka);
  const uint32_t m3 = (w[7] ^ xl) & (w[11] ^ xh) & ~(w[15] & ka);
  const uint32_t blk = ln.a >> GMX_BLK_SHIFT;
  const uint32_t base = c == 1 ? w[0] : (c == 2 ? w[1] : (c == 3 ? w[2] : (blk << GMX_BLK_SHIFT) - w[0] - w[1] - w[2] - w[3]));
  uint32_t rank_lo = base + (uint32_t)__builtin_popcount(m0 & a0) + (uint32_t)__builtin_popcount(m1 & a1) +
                     (uint32_t)__builtin_popcount(m2 & a2) + (uint32_t)__builtin_popcount(m3 & a3);
  uint32_t rank_hi1 = base + (uint32_t)__builtin_popcount(m0 & z0) + (uint32_t)__builtin_popcount(m1 & z1) +
                      (uint32_t)__builtin_popcount(m2 & z2) + (uint32_t)__builtin_popcount(m3 & z3);
  if (c == 1) {  // the sentinel is stored as code 00, it is not a base
    if (ix.sentinel_pos < ln.a) rank_lo -= 1;
    if (ix.sentinel_pos <= ln.b) rank_hi1 -= 1;
  }
  if (rank_lo == rank_hi1) {
    ln.mode = GMX_MODE_DEAD;
    return true;
  }
  const uint32_t first = c == 1 ? ix.C[1] : (c == 2 ? ix.C[2] : (c == 3 ? ix.C[3] : ix.C[4]));
  ln.a = first + rank_lo;
  ln.b = first + rank_hi1 - 1u;
  --ln.pos;
  return true;
}

// One fast iteration with direct loads (host emulation, and the reference for the kernels' scheduled version).
template <class Ctx, class Reader>
GMX_HD bool gmx_dfs_fast_iter(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop, GmxLane &ln, uint32_t kind) {
  switch (kind) {
    case GMX_FAST_HIT:
      return gmx_dfs_fast_hit(ctx, ln, stop, *gmx_dfs_hit_sub(ix, rd, ln));
    case GMX_FAST_WIDE:
      return gmx_dfs_fast_wide(ix, rd, ln, reinterpret_cast<const uint32_t *>(ix.blocks + (ln.a >> GMX_BLK_SHIFT)));
    case GMX_FAST_CONVERT:
      ln.a = ix.sa[ln.a];
      ln.b = GMX_TEXT_MARK;
      return true;
    case GMX_FAST_EMIT:
      gmx_dfs_emit(ctx, ln);
      return true;
    case GMX_FAST_POP:
      gmx_dfs_pop(ctx, ln);
      return true;
    default:
      return gmx_dfs_text_apply(ctx, ln, stop, rd, ix.text[gmx_dfs_text_rec(ln)]);
  }
}

// Host-style driver (one lane at a time): the kernels interleave the same two functions with wave-level
// batching of the general path (gmx_engine.hip: dfs_run_wave).
template <class Ctx, class Reader>
GMX_HD void gmx_dfs_run(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop) {
  GmxLane ln;
  ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
  while (ln.have) {
    const uint32_t kind = gmx_dfs_fast_kind(ln, stop);
    if (kind != GMX_FAST_NONE && gmx_dfs_fast_iter(ix, ctx, rd, stop, ln, kind)) continue;
    gmx_dfs_slow_iter(ix, ctx, rd, stop, ln);
  }
}
