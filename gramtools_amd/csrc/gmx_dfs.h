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

// Runs the lane's queue dry. `rd.at(i)` is the oriented base i of the read; states stop at position `stop`
// (0 = the whole read; > 0 = the probe phase parks survivors there).
template <class Ctx, class Reader>
GMX_HD void gmx_dfs_run(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop) {
  uint32_t a, b, tvd, tvg, pos, mode;
  bool have = ctx.pop(a, b, tvd, tvg, pos, mode);
  while (have) {
    if (pos <= stop) {  // parked / seed already at the stop position
      if (mode == GMX_MODE_HIT) {
        ctx.fail(GMX_TASK_ERROR);  // hits are only created for positions > stop
        break;
      }
      if (!ctx.emit(a, b, tvd, tvg)) ctx.fail(GMX_TASK_OVERFLOW);
      if (ctx.status != GMX_TASK_MAPPED) break;
      have = ctx.pop(a, b, tvd, tvg, pos, mode);
      continue;
    }
    // --- the iteration's one line fetch ---
    const GmxLine *src = mode == GMX_MODE_HIT ? reinterpret_cast<const GmxLine *>(ix.hits + a)
                                              : reinterpret_cast<const GmxLine *>(ix.blocks + (a >> GMX_BLK_SHIFT));
    const GmxLine line = *src;
    const uint32_t c = rd.at(pos - 1);
    bool alive = false;
    if (mode == GMX_MODE_HIT) {
      const uint32_t kind = line.w[0];
      if (kind == GMX_HIT_EXIT) {  // update_variant_site_path + exiting_site_search_state, vBWT_jump.cpp:51-92
        const uint32_t site = line.w[2];
        bool ok = true;
        if (tvg != GMX_NIL) {
          if (ctx.arena_site(tvg) != site) {
            ctx.fail(GMX_TASK_ERROR);
            ok = false;
          } else
            tvg = ctx.arena_next(tvg);
        }
        if (ok) {
          uint32_t nn = ctx.arena_new(site, (int32_t)line.w[3], tvd);
          if (nn == GMX_NIL) {
            ctx.fail(GMX_TASK_OVERFLOW);
          } else {
            tvd = nn;
            alive = line.w[4] == c;  // the only base that can precede the site marker
            a = b = line.w[5];
          }
        }
      } else if (kind == GMX_HIT_ENTER) {  // entering_site_search_state, vBWT_jump.cpp:29-44
        uint32_t nn = ctx.arena_new(line.w[2], -1, tvg);
        if (nn == GMX_NIL) {
          ctx.fail(GMX_TASK_OVERFLOW);
        } else {
          tvg = nn;
          a = c == 1 ? line.w[4] : (c == 2 ? line.w[6] : (c == 3 ? line.w[8] : line.w[10]));
          b = c == 1 ? line.w[5] : (c == 2 ? line.w[7] : (c == 3 ? line.w[9] : line.w[11]));
          alive = a <= b;
        }
      } else {  // general jump program: its outputs still need their LF step -> pushed as LF-only entries
        GmxDfsProgSink<Ctx> sink{ctx, pos};
        gmx_run_program(ix, line.w[1], tvd, tvg, sink);
      }
    } else {
      const GmxRankBlock blk = gmx_line_as_block(line);
      if (mode == GMX_MODE_STATE) gmx_dfs_push_hits(ix, a, b, tvd, tvg, pos, blk, ctx);
      alive = gmx_lf(ix, c, a, b, blk);
    }
    if (ctx.status != GMX_TASK_MAPPED) break;
    if (alive) {
      --pos;
      mode = GMX_MODE_STATE;
    } else {
      have = ctx.pop(a, b, tvd, tvg, pos, mode);
    }
  }
}
