// gmx_dfs.h — the per-lane search loop of the HIP kernels: a depth-first work queue.
//
// search_read_backwards (libgramtools/src/genotype/quasimap/quasimap.cpp:227-256) advances ALL states of a
// read one base at a time. States never interact, so a lane may instead carry ONE state in registers down the
// read and keep the others (siblings created at variant markers, extra seed states) on a small LIFO stack:
//
//   every iteration = one 64-byte line fetch (a rank block, or a marker-hit record) + register arithmetic.
//
// No second dependent load inside an iteration, no per-wave serialisation of the rare marker path: a marker
// hit only pushes {hit rank, path handles, position}; it is resolved when popped — by then as the lane's
// one fetch of that iteration. Final coverage does not depend on the order in which states are explored
// (DESIGN.md §4), and every state is explored exactly as the reference would.
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
#define GMX_MODE_HIT 2u    // unresolved marker hit: a = marker rank

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
      uint32_t h = mbase + gmx_popc64(k0 & ((1ull << bit) - 1ull));
      if (!ctx.push(h, 0, tvd, tvg, pos, GMX_MODE_HIT)) ctx.fail(GMX_TASK_OVERFLOW);
    }
    uint32_t c0 = gmx_popc64(k0);
    while (s1) {
      uint32_t bit = (uint32_t)__builtin_ctzll(s1);
      s1 &= s1 - 1;
      uint32_t h = mbase + c0 + gmx_popc64(k1 & ((1ull << bit) - 1ull));
      if (!ctx.push(h, 0, tvd, tvg, pos, GMX_MODE_HIT)) ctx.fail(GMX_TASK_OVERFLOW);
    }
  }
}

// 64 raw bytes, fetched from a per-lane address; only ever indexed with constants (stays in registers)
struct GmxLine {
  uint32_t w[16];
};
GMX_HD GmxRankBlock gmx_line_as_block(const GmxLine &l) {
  GmxRankBlock b;
  b.cnt[0] = l.w[0];
  b.cnt[1] = l.w[1];
  b.cnt[2] = l.w[2];
  b.cnt[3] = l.w[3];
  b.lo[0] = (uint64_t)l.w[4] | ((uint64_t)l.w[5] << 32);
  b.lo[1] = (uint64_t)l.w[6] | ((uint64_t)l.w[7] << 32);
  b.hi[0] = (uint64_t)l.w[8] | ((uint64_t)l.w[9] << 32);
  b.hi[1] = (uint64_t)l.w[10] | ((uint64_t)l.w[11] << 32);
  b.mk[0] = (uint64_t)l.w[12] | ((uint64_t)l.w[13] << 32);
  b.mk[1] = (uint64_t)l.w[14] | ((uint64_t)l.w[15] << 32);
  return b;
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
  // --- the iteration's one line fetch ---
  const GmxLine *src = ln.mode == GMX_MODE_HIT ? reinterpret_cast<const GmxLine *>(ix.hits + ln.a)
                                               : reinterpret_cast<const GmxLine *>(ix.blocks + (ln.a >> GMX_BLK_SHIFT));
  const GmxLine line = *src;
  const uint32_t c = rd.at(ln.pos - 1);
  bool alive = false;
  if (ln.mode == GMX_MODE_HIT) {
    const uint32_t kind = line.w[0];
    if (kind == GMX_HIT_EXIT) {  // update_variant_site_path + exiting_site_search_state, vBWT_jump.cpp:51-92
      const uint32_t site = line.w[2];
      bool ok = true;
      if (ln.tvg != GMX_NIL) {
        if (ctx.arena_site(ln.tvg) != site) {
          ctx.fail(GMX_TASK_ERROR);
          ok = false;
        } else
          ln.tvg = ctx.arena_next(ln.tvg);
      }
      if (ok) {
        uint32_t nn = ctx.arena_new(site, (int32_t)line.w[3], ln.tvd);
        if (nn == GMX_NIL) {
          ctx.fail(GMX_TASK_OVERFLOW);
        } else {
          ln.tvd = nn;
          alive = line.w[4] == c;  // the only base that can precede the site marker
          ln.a = ln.b = line.w[5];
        }
      }
    } else if (kind == GMX_HIT_ENTER) {  // entering_site_search_state, vBWT_jump.cpp:29-44
      uint32_t nn = ctx.arena_new(line.w[2], -1, ln.tvg);
      if (nn == GMX_NIL) {
        ctx.fail(GMX_TASK_OVERFLOW);
      } else {
        ln.tvg = nn;
        ln.a = c == 1 ? line.w[4] : (c == 2 ? line.w[6] : (c == 3 ? line.w[8] : line.w[10]));
        ln.b = c == 1 ? line.w[5] : (c == 2 ? line.w[7] : (c == 3 ? line.w[9] : line.w[11]));
        alive = ln.a <= ln.b;
      }
    } else {  // general jump program: its outputs still need their LF step -> pushed as LF-only entries
      GmxDfsProgSink<Ctx> sink{ctx, ln.pos};
      gmx_run_program(ix, line.w[1], ln.tvd, ln.tvg, sink);
    }
  } else {
    const GmxRankBlock blk = gmx_line_as_block(line);
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

// The FAST iteration covers the cases that make up almost every step of a read on a flat PRG:
//   * a width-1 interval: LF step with 32-bit word arithmetic on one block; if a variant marker precedes the
//     position, the state turns into the pending hit itself (its own LF step would die: the BWT symbol is not a base);
//   * a pending hit whose record is pre-resolved (GMX_HIT_EXIT / GMX_HIT_ENTER) and whose traversing path is empty
//     or inline: path update + the precomputed LF result, no further fetch.
// No stack traffic, no calls, one 64-byte line per iteration. Preconditions: gmx_dfs_fast_ok().
// Returns false, leaving the lane untouched, when the general iteration is needed.
GMX_HD bool gmx_dfs_fast_ok(const GmxLane &ln, uint32_t stop) {
  return ln.have && ln.pos > stop && ((ln.mode == GMX_MODE_STATE && ln.a == ln.b) || ln.mode == GMX_MODE_HIT);
}
// address of the 64-byte line the lane's fast iteration consumes: a rank block or a hit record
GMX_HD const uint32_t *gmx_dfs_fast_src(const GmxIndexView &ix, const GmxLane &ln) {
  return ln.mode == GMX_MODE_HIT ? reinterpret_cast<const uint32_t *>(ix.hits + ln.a)
                                 : reinterpret_cast<const uint32_t *>(ix.blocks + (ln.a >> GMX_BLK_SHIFT));
}
// `w` = the 16 words of that line (the extend kernel fetches it quad-cooperatively, everything else directly)
template <class Ctx, class Reader>
GMX_HD bool gmx_dfs_fast_iter_line(const GmxIndexView &ix, Ctx &ctx, Reader &rd, GmxLane &ln, const uint32_t *w) {
  const bool is_hit = ln.mode == GMX_MODE_HIT;
  const uint32_t i = ln.a;
  const uint32_t bi = i >> GMX_BLK_SHIFT;
  // one 64-byte line: a rank block (counts | lo plane | hi plane | marker plane) or a hit record
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  const uint32_t l0 = w[4], l1 = w[5], l2 = w[6], l3 = w[7];
  const uint32_t h0 = w[8], h1 = w[9], h2 = w[10], h3 = w[11];
  const uint32_t k0 = w[12], k1 = w[13], k2 = w[14], k3 = w[15];
  const uint32_t c = rd.at(ln.pos - 1);
  if (is_hit) {
    (void)w1;
    bool alive;
    if (w0 == GMX_HIT_EXIT) {  // update_variant_site_path + exiting_site_search_state, vBWT_jump.cpp:51-92
      if (ln.tvg != GMX_NIL) {
        if (!gmx_h_inline(ln.tvg) || 5u + 2u * (ln.tvg & ~GMX_INLINE_FLAG) != w2) return false;  // general path (or error there)
      }
      uint32_t nn = ctx.arena_new(w2, (int32_t)w3, ln.tvd);
      if (nn == GMX_NIL) return false;
      ln.tvg = GMX_NIL;
      ln.tvd = nn;
      alive = l0 == c;  // lf[0]: the only base that can precede the site marker
      ln.a = ln.b = l1;
    } else if (w0 == GMX_HIT_ENTER) {  // entering_site_search_state, vBWT_jump.cpp:29-44
      if (ln.tvg != GMX_NIL) return false;  // nested entry: the general path materialises the list
      ln.tvg = GMX_INLINE_FLAG | ((w2 - 5u) >> 1);
      ln.a = c == 1 ? l0 : (c == 2 ? l2 : (c == 3 ? h0 : h2));
      ln.b = c == 1 ? l1 : (c == 2 ? l3 : (c == 3 ? h1 : h3));
      alive = ln.a <= ln.b;
    } else
      return false;
    if (alive) {
      --ln.pos;
      ln.mode = GMX_MODE_STATE;
    } else
      ln.mode = GMX_MODE_DEAD;
    return true;
  }
  const uint32_t j = (i >> 5) & 3u, t = i & 31u;
  const uint32_t below = (1u << t) - 1u;
  const uint32_t kj = j == 0 ? k0 : (j == 1 ? k1 : (j == 2 ? k2 : k3));
  if ((kj >> t) & 1u) {
    // a variant marker precedes this position: the state becomes its own (single) pending hit; marker rank =
    // block count + markers below the position (left_markers_search, vBWT_jump.cpp:94-117)
    uint32_t r = w3 + (j > 0 ? (uint32_t)__builtin_popcount(k0) : 0u) + (j > 1 ? (uint32_t)__builtin_popcount(k1) : 0u) +
                 (j > 2 ? (uint32_t)__builtin_popcount(k2) : 0u) + (uint32_t)__builtin_popcount(kj & below);
    ln.a = r;
    ln.b = 0;
    ln.mode = GMX_MODE_HIT;
    return true;
  }
  const uint32_t code = c - 1u;
  const uint32_t xl = (code & 1u) ? 0u : ~0u, xh = (code & 2u) ? 0u : ~0u, ka = c == 1 ? ~0u : 0u;
  const uint32_t m0 = (l0 ^ xl) & (h0 ^ xh) & ~(k0 & ka);
  const uint32_t m1 = (l1 ^ xl) & (h1 ^ xh) & ~(k1 & ka);
  const uint32_t m2 = (l2 ^ xl) & (h2 ^ xh) & ~(k2 & ka);
  const uint32_t m3 = (l3 ^ xl) & (h3 ^ xh) & ~(k3 & ka);
  const uint32_t mj = j == 0 ? m0 : (j == 1 ? m1 : (j == 2 ? m2 : m3));
  uint32_t rank = (j > 0 ? (uint32_t)__builtin_popcount(m0) : 0u) + (j > 1 ? (uint32_t)__builtin_popcount(m1) : 0u) +
                  (j > 2 ? (uint32_t)__builtin_popcount(m2) : 0u) + (uint32_t)__builtin_popcount(mj & below);
  const uint32_t cT = (bi << GMX_BLK_SHIFT) - w0 - w1 - w2 - w3;
  rank += c == 1 ? w0 : (c == 2 ? w1 : (c == 3 ? w2 : cT));
  bool hit = ((mj >> t) & 1u) != 0;
  if (c == 1) {  // the sentinel is stored as code 00
    if (ix.sentinel_pos < i) rank -= 1;
    if (ix.sentinel_pos == i) hit = false;
  }
  if (hit) {
    const uint32_t first = c == 1 ? ix.C[1] : (c == 2 ? ix.C[2] : (c == 3 ? ix.C[3] : ix.C[4]));
    ln.a = ln.b = first + rank;
    --ln.pos;
  } else {
    ln.mode = GMX_MODE_DEAD;
  }
  return true;
}

template <class Ctx, class Reader>
GMX_HD bool gmx_dfs_fast_iter(const GmxIndexView &ix, Ctx &ctx, Reader &rd, GmxLane &ln) {
  const uint32_t *src = gmx_dfs_fast_src(ix, ln);
  uint32_t w[16];
  for (int k = 0; k < 16; ++k) w[k] = src[k];
  return gmx_dfs_fast_iter_line(ix, ctx, rd, ln, w);
}

// Host-style driver (one lane at a time): the kernels interleave the same two functions with wave-level
// batching of the general path (gmx_engine.hip: dfs_run_wave).
template <class Ctx, class Reader>
GMX_HD void gmx_dfs_run(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop) {
  GmxLane ln;
  ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
  while (ln.have) {
    if (gmx_dfs_fast_ok(ln, stop) && gmx_dfs_fast_iter(ix, ctx, rd, ln)) continue;
    gmx_dfs_slow_iter(ix, ctx, rd, stop, ln);
  }
}
