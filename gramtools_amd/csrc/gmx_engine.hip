// gmx_engine.hip — HIP kernels (gfx950) and the device half of the C ABI.
//
// Execution model (HISTORY.md §2): one LANE per (read, orientation) task — 64 independent vBWT backward
// searches per wavefront. A lane carries one search state in registers and keeps the others on a small LIFO
// stack in LDS (gmx_dfs.h). A state that has narrowed to ONE suffix-array position is kept in text form and
// compares 32 read bases per step against a 16-byte record of the PRG itself; wide intervals use 64-byte rank
// blocks; a variant marker costs one 16-byte sub-record of its pre-resolved hit record. Final states go to
// the coverage kernels (gmx_cover.h): class selection with the seeded draw, then the coverage atomics.
//
// Kernels (launch order, HISTORY.md §2.4):
//   gmx_pack_kernel          flags reads with a non-ACGT byte (encode_dna_bases, utils.cpp:73-92), packs bases to bit planes
//   gmx_probe_kernel         seed look-up + the first steps of search_read_backwards (quasimap.cpp:227-256); survivors are parked
//   gmx_extend_kernel        the rest of the read for the compacted survivors
//   gmx_search_big_kernel    the same search for tasks that overflowed the per-lane pools (global-memory pools; side streams)
//   gmx_filter[_lds]_kernel  all_read_kmers_occur_in_index for tasks without final state (quasimap.cpp:212-225)
//   gmx_cover_single_kernel  coverage::record::search_states (coverage_common.cpp:179-197) for single-instance tasks
//   gmx_cover_kernel         the same in general (classes, seeded selection, hull), two scratch sizes
//   (QuasimapReadsStats, quasimap.hpp:17-24, are counted by the kernels that decide each task: SearchOut::stats)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gmx.h"
#include "gmx_core.h"
#ifdef GMX_LOOP_STATS  // debug build: wall time of the coverage routine's phases (tools/cover_stats.py)
#define GMX_COVER_PROF(env, k) (env).prof(k)
#define GMX_COVER_WHY(env, k) (env).why(k)
#endif
#include "gmx_cover.h"
#include "gmx_dfs.h"
#include "gmx_index.h"
#include "gmx_internal.h"

#ifndef GMX_BLOCK
#define GMX_BLOCK 256
#endif
#ifndef GMX_FAST_STATES
#define GMX_FAST_STATES 8     // final / parked states kept per task by the fast pass
#endif
#ifndef GMX_STACK_DEPTH
#define GMX_STACK_DEPTH 6
#endif
// GMX_STACK_DEPTH: pending entries (sibling states, unresolved marker hits) per lane, in LDS
#define GMX_STACK_WORDS 5
#define GMX_CNT_STRIDE 32      // device counters sit 128 B apart: same-line atomics would serialise in one L2 channel
#define GMX_N_COUNTERS 40      // per-batch queue counters (SearchOut::counters)
#define GMX_CNT_LOG_RETRY 30u       // entries of log_retry_list: coverage queue entries whose task found the grouped log full
#define GMX_CNT_LOG_RETRY_RECS 31u  // ... compact records (log_retry_recs: index into cover_recs)
#define GMX_CNT_LOG_RETRY_HUGE 33u  // ... tasks the last tier has to search again (log_retry_huge)
#define GMX_CNT_GENERAL_REST 34u    // entries gmx_cover_one_kernel left to the general instances (general_rest_list)
#define GMX_CNT_ALIVE2 35u          // stragglers of the extend kernel, parked for its next pass: [35 + pass], pass 0 .. GMX_EXTRA_PASSES - 1
#define GMX_EXTRA_PASSES 3          // (counters 35, 36, 37; task lists GMX_TL_ALIVE2 ..)
#define GMX_CNT_SINGLE_REST 38u     // compact records gmx_cover_jump_kernel left to gmx_cover_single_rest_kernel (single_rest_list)
#define GMX_CNT_REPLAY_RECS 32u     // replay: number of compact records to redo (gmx_cover_single_replay_kernel)
#ifndef GMX_FAST_ARENA
#define GMX_FAST_ARENA 48     // path arena nodes per task (fast pass): a read through an MSA region of configs[2] needs 25-40
                              // (24 sent 14 k of a million such reads to the large-capacity pass: 5.1 -> 3.1 ms per batch)
#endif
#define GMX_STATUS_MISSING_KMER 5u  // refinement of GMX_TASK_UNMAPPED by the k-mer filter
#define GMX_STATUS_IGNORED 7u       // reverse-complement task of a forward_only engine: not mapped, not counted

// ---------------------------------------------------------------------------
// read access: oriented base i of task (read r, orientation o)
// ---------------------------------------------------------------------------

struct __attribute__((aligned(8))) gmx_pair2 {
  uint32_t x, y, z, w;
};
struct ReadRef {
  const uint2 *w;      // bit planes of the base codes (A,C,G,T = 0..3): .x = low bits, .y = high bits of 32 bases
  uint32_t len;
  bool rc;
  uint32_t cur_idx;    // index of the cached pair (0xFFFFFFFF = none)
  uint2 cur;           // cached pair: the walk is sequential, so one load serves 32 steps
  __device__ __forceinline__ uint32_t at(uint32_t i) {
    uint32_t idx = rc ? len - 1 - i : i;  // reverse_complement_read, quasimap.cpp:273-298
    uint32_t wi = idx >> 5;
    if (wi != cur_idx) {
      cur = w[wi];
      cur_idx = wi;
    }
    uint32_t code = ((cur.x >> (idx & 31u)) & 1u) | (((cur.y >> (idx & 31u)) & 1u) << 1);
    return rc ? 4u - code : code + 1u;
  }
  // planes of raw bases start .. start + 31 (gmx_dfs.h, text-form iteration); the packed buffer has slack
  // behind the last read, bits past this read's end are never used
  __device__ __forceinline__ void planes(uint32_t start, uint32_t &lo, uint32_t &hi) const {
    const gmx_pair2 p = *reinterpret_cast<const gmx_pair2 *>(w + (start >> 5));  // one 16-byte load, 8-byte aligned
    lo = __builtin_amdgcn_alignbit(p.z, p.x, start & 31u);
    hi = __builtin_amdgcn_alignbit(p.w, p.y, start & 31u);
  }
};

// The same with the read held in registers (reads of up to 192 bases; longer ones fall back to memory). The search
// loop then issues no memory request for read bases at all: its kernels sit near the L2's request rate for
// scattered lines, and the read windows were about 40 % of the requests.
#define GMX_READ_REG_PAIRS 6
struct ReadRegs {
  uint2 p0, p1, p2, p3, p4, p5;
  const uint2 *w;
  uint32_t len;
  bool rc;
  bool in_regs;
  __device__ __forceinline__ uint2 sel(uint32_t d) const {
    uint2 r = make_uint2(0, 0);
    r = d == 0 ? p0 : r;
    r = d == 1 ? p1 : r;
    r = d == 2 ? p2 : r;
    r = d == 3 ? p3 : r;
    r = d == 4 ? p4 : r;
    r = d == 5 ? p5 : r;
    return r;
  }
  __device__ __forceinline__ void load(const uint2 *pairs, uint32_t length, bool reverse) {
    w = pairs;
    len = length;
    rc = reverse;
    in_regs = length <= 32u * GMX_READ_REG_PAIRS;
    p0 = p1 = p2 = p3 = p4 = p5 = make_uint2(0, 0);
    if (in_regs) {  // the packed buffer has slack behind the last read
      const gmx_pair2 a = *reinterpret_cast<const gmx_pair2 *>(pairs), b = *reinterpret_cast<const gmx_pair2 *>(pairs + 2),
                      c = *reinterpret_cast<const gmx_pair2 *>(pairs + 4);
      p0 = make_uint2(a.x, a.y);
      p1 = make_uint2(a.z, a.w);
      p2 = make_uint2(b.x, b.y);
      p3 = make_uint2(b.z, b.w);
      p4 = make_uint2(c.x, c.y);
      p5 = make_uint2(c.z, c.w);
    }
  }
  __device__ __forceinline__ void clear(const uint2 *pairs) {
    w = pairs;
    len = 0;
    rc = false;
    in_regs = true;
    p0 = p1 = p2 = p3 = p4 = p5 = make_uint2(0, 0);
  }
  __device__ __forceinline__ uint32_t at(uint32_t i) const {
    const uint32_t idx = rc ? len - 1 - i : i;
    const uint2 pr = in_regs ? sel(idx >> 5) : w[idx >> 5];
    const uint32_t code = ((pr.x >> (idx & 31u)) & 1u) | (((pr.y >> (idx & 31u)) & 1u) << 1);
    return rc ? 4u - code : code + 1u;
  }
  __device__ __forceinline__ void planes(uint32_t start, uint32_t &lo, uint32_t &hi) const {
    uint2 a, b;
    if (in_regs) {
      a = sel(start >> 5);
      b = sel((start >> 5) + 1);
    } else {
      a = w[start >> 5];
      b = w[(start >> 5) + 1];
    }
    lo = __builtin_amdgcn_alignbit(b.x, a.x, start & 31u);
    hi = __builtin_amdgcn_alignbit(b.y, a.y, start & 31u);
  }
};

// ---------------------------------------------------------------------------
// per-lane contexts
// ---------------------------------------------------------------------------
extern __shared__ uint32_t gmx_lds[];

// What the single-instance coverage kernel needs of a mapped task, in one 32-byte record written by the search
// kernel that finished it: the final state's PRG position, the read length, the traversing path (inline handle or
// nil) and the traversed loci, newest first, in one of two forms: up to three (site, allele) pairs, or — the sites
// along a read through a non-nested PRG are consecutive — up to GMX_REC_RUN loci as the first site and one allele byte
// each. The kernel reads its queue coalesced and touches neither the task's final states nor its path arena. Tasks
// that do not fit (several final states, an SA-form final state, longer or non-consecutive paths, large allele ids,
// reads >= 65536 bases) go to the general coverage queue as task ids.
#define GMX_REC_RUN 16u
#define GMX_REC_RUN_FLAG 0x80000000u
struct alignas(32) GmxCoverRec {
  uint32_t p;
  uint32_t len_n;  // read length | number of traversed loci << 16 | GMX_REC_RUN_FLAG (run form)
  uint32_t tvg;
  uint32_t site[3];  // pair form: the sites; run form: site[0] = site of locus 0 (locus i: site[0] + 2 i), then allele bytes
  uint32_t a01;    // pair form: allele 0 | allele 1 << 16; run form: allele bytes 8..11
  uint32_t a2;     // pair form: allele 2; run form: allele bytes 12..15
};

// A pending entry of a task handed from the probe kernel to the extend kernel (overlays the task's finals[]).
struct GmxParked {
  uint32_t a, b, tvd, tvg, pm;  // pm = read position | mode << 30, as on the stack
};
static_assert(GMX_STACK_DEPTH * sizeof(GmxParked) <= GMX_FAST_STATES * sizeof(GmxFinalState), "parked entries overlay finals[]");

// A state of a multi-state k-mer index entry as the DEVICE copy of the words holds it (gmx_seed_mark_kernel rewrites the
// host form [lo, hi, n_traversed, n_traversing, paths...] in place): a state over ONE suffix-array position is
// [PRG position, left context, n_traversed | GMX_SEEDST_TEXT, n_traversing, paths...]. Left context: the up to 14 base
// symbols left of the position (2 bits each, nearest first) up to the first marker or the PRG's start, and in bits 28..31
// how many there are: a seed state is rejected on it without any fetch (FastCtx::next_seed_screened).
#define GMX_SEEDST_TEXT 0x80000000u
#define GMX_SEEDST_CTX 14u
struct GmxSeedState {
  uint32_t lo, hi, nt, ng, ctx;
  __device__ __forceinline__ bool text() const { return hi == GMX_TEXT_MARK; }
  __device__ __forceinline__ uint32_t words() const { return 4u + 2u * nt + ng; }
  __device__ __forceinline__ uint32_t width() const { return text() ? 1u : hi - lo + 1u; }
};
// the left-context word of PRG position tp (GmxSeedState)
__device__ __forceinline__ uint32_t gmx_left_context(const GmxTextRec *text, uint32_t tp) {
  uint32_t ctx = 0, nv = 0;
  for (; nv < GMX_SEEDST_CTX && nv < tp; ++nv) {
    const uint32_t q = tp - 1u - nv;
    const GmxTextRec rec = text[q >> GMX_TEXT_SHIFT];
    const uint32_t bit = q & GMX_TEXT_MASK;
    if ((rec.mk >> bit) & 1ull) break;
    ctx |= ((uint32_t)((rec.lo >> bit) & 1ull) | ((uint32_t)((rec.hi >> bit) & 1ull) << 1)) << (2u * nv);
  }
  return ctx | (nv << 28);
}
__device__ __forceinline__ GmxSeedState gmx_seed_state(const uint32_t *p) {
  const uint32_t w2 = p[2];
  const bool text = (w2 & GMX_SEEDST_TEXT) != 0;
  return GmxSeedState{p[0], text ? GMX_TEXT_MARK : p[1], w2 & ~GMX_SEEDST_TEXT, p[3], text ? p[1] : 0u};
}

struct FastCtx {  // pending-entry stack in LDS (lane-strided), traversed-path arena and emitted states in global memory
  uint32_t sp;
  GmxPathNode *arena;
  uint32_t arena_n;
  uint32_t arena_stride;  // tasks the table was allocated for (SearchOut::arena_stride)
  uint32_t arena_first;   // handle of this lane's node 0 (0; instance lanes: their part of the task's slot pool)
  // Instance lanes (gmx_extend_inst_kernel): one of several lanes searching the same task. Final states go straight into
  // the task's large-capacity slot, each at a position drawn from the slot's counter.
  GmxFinalState *inst_states;  // non-null: instance mode
  uint32_t *inst_count;
  uint32_t inst_cap;
  uint32_t status;
  GmxFinalState *out;
  uint32_t n_out, out_cap;
  uint32_t first_pos; // PRG position of the first emitted text-form state (GMX_NIL if none): the task's coverage region
  uint32_t first_tvd, first_tvg;  // its path handles
  bool parking;       // probe kernel: "emitted" states are parked for the extend kernel (GmxParked, same memory)
  uint32_t park_pos;  // read position of states parked by emit()
  // Seed cursor: the states of a multi-state k-mer index entry are taken ONE AT A TIME from the index (seed_words)
  // whenever the stack runs empty, instead of being pushed all at once — a k-mer of a large or dense PRG has tens of
  // states, far more than the stack holds. Path nodes of a seed state whose descendants all died are released.
  uint32_t seed_left;               // states of the k-mer index entry not started yet
  uint64_t seed_off;                // word offset of the next one in seed_words (above 2^32 in a whole-genome index)
  uint32_t seed_pos;                // read position of the seed states
  uint32_t mark_arena, mark_out;    // arena / emitted-state counts when the current seed state started
  uint32_t seed_rctx, seed_rn = 0xFFFFFFFFu;  // the read's bases left of seed_pos as a left-context word, and how many (lazily)
  __device__ __forceinline__ bool more_seeds() const { return seed_left != 0 && status == GMX_TASK_MAPPED; }
  __device__ __forceinline__ bool next_seed(const GmxIndexView &ix, bool release, uint32_t &a, uint32_t &b, uint32_t &tvd,
                                            uint32_t &tvg, uint32_t &pos, uint32_t &mode) {
    // nothing emitted since the previous seed state started: all its descendants died, its path nodes are garbage
    if (release && n_out == mark_out) arena_n = mark_arena;
    mark_arena = arena_n;
    mark_out = n_out;
    const uint32_t *p = ix.seed_words + seed_off;
    const GmxSeedState ss = gmx_seed_state(p);
    const uint32_t lo = ss.lo, hi = ss.hi, nt = ss.nt, ng = ss.ng;
    p += 4;
    tvd = tvg = GMX_NIL;
    for (uint32_t j = 0; j < nt; ++j, p += 2) {
      tvd = arena_new(p[0], (int32_t)p[1], tvd);
      if (tvd == GMX_NIL) break;
    }
    bool ok = nt == 0 || tvd != GMX_NIL;
    for (uint32_t j = 0; ok && j < ng; ++j, ++p) {
      tvg = arena_new(p[0], -1, tvg);
      ok = tvg != GMX_NIL;
    }
    if (!ok) {
      fail(GMX_TASK_OVERFLOW);
      seed_left = 0;
      return false;
    }
    seed_off = (uint64_t)(p - ix.seed_words);
    --seed_left;
    a = lo;
    b = hi;
    pos = seed_pos;
    mode = GMX_MODE_STATE;
    return true;
  }
  // Would a text-form state at PRG position `tp`, read position `pos`, survive its first text step? A DRY RUN of that very
  // step — gmx_dfs_text_apply on the record of tp - 1 with a context that allocates nothing — so inline sites are walked
  // through as the real step walks them: a state next to a SNP site (a site every 36 bases in a whole-genome PRG) is
  // compared beyond it instead of passing for "alive at a marker" and costing three iterations of the wave loop to die.
  // One 32-byte fetch; dead here = dead there (the real step takes the same decisions; it can only add an arena overflow).
  struct DryCtx {
    __device__ __forceinline__ uint32_t arena_new(uint32_t, int32_t, uint32_t) { return 0u; }
  };
  template <class Reader>
  __device__ __forceinline__ bool seed_text_alive(const GmxIndexView &ix, Reader &rd, uint32_t tp, uint32_t pos, uint32_t stop) const {
    if (pos <= stop) return true;  // already final
    GmxLane t;
    t.a = tp, t.b = GMX_TEXT_MARK, t.tvd = t.tvg = GMX_NIL, t.pos = pos, t.mode = GMX_MODE_STATE, t.have = true;
    const GmxTextRec rec = ix.text[gmx_dfs_text_rec(t)];
    DryCtx dry;
    (void)gmx_dfs_text_apply(dry, t, stop, rd, rec);
    return t.mode != GMX_MODE_DEAD;
  }
  // The seed cursor with a screen in front (indexes whose k-mers have many states: a whole-genome PRG has ~12 occurrences
  // per 14-mer, a third of them across a site — and all but one of a read's seed states die at their first text step, after
  // three iterations of the wave loop each: next state, suffix-array look-up, compare). States over ONE suffix-array position
  // are tested here, in a tight per-lane loop, and only the survivors enter the wave loop — already in text form; a
  // path-less state over a few positions (the k-mer's occurrences outside sites) is taken apart into its occurrences, which
  // is the same search (load_seed_cursor, gmx_search_big_kernel), and screened likewise. Nothing changes for the states
  // that survive: they are searched by the same code from the same position.
  template <class Reader>
  __device__ __forceinline__ bool next_seed_screened(const GmxIndexView &ix, Reader &rd, uint32_t stop, uint32_t &a, uint32_t &b,
                                                     uint32_t &tvd, uint32_t &tvg, uint32_t &pos, uint32_t &mode) {
    if (seed_rn == 0xFFFFFFFFu) {  // the read's bases left of the seed, once per task, in the entries' left-context form
      seed_rn = seed_pos > stop ? min(seed_pos - stop, GMX_SEEDST_CTX) : 0u;
      seed_rctx = 0;
      for (uint32_t j = 0; j < seed_rn; ++j) seed_rctx |= (rd.at(seed_pos - 1u - j) - 1u) << (2u * j);
    }
    // a mismatch among the bases before the first marker: dead, without fetching anything
    auto ctx_dead = [&](uint32_t c) {
      const uint32_t n = min(c >> 28, seed_rn);
      return n != 0 && (((c ^ seed_rctx) << (32u - 2u * n)) != 0u);
    };
    // The WHOLE rest of the entry is screened in this one call: the first survivor becomes the lane's state, further ones
    // (rare) go on its stack. A lane that came back to the cursor after every survivor put the screening loop into most
    // iterations of its wave (lanes finish at different times), at the price of the loop's instructions for all 64.
    bool got = false;
    auto take = [&](uint32_t sa_, uint32_t sb_, uint32_t st_, uint32_t sg_) {
      if (!got) {
        a = sa_, b = sb_, tvd = st_, tvg = sg_, pos = seed_pos, mode = GMX_MODE_STATE;
        got = true;
      } else if (!push(sa_, sb_, st_, sg_, seed_pos, GMX_MODE_STATE)) {
        fail(GMX_TASK_OVERFLOW);
      }
    };
    while (status == GMX_TASK_MAPPED) {
      // Phase A, a loop of its own: skip the states the left context rejects (a header load and a dozen instructions each).
      // The lanes of a wave run it together and meet again behind it, so the heavy code below — text record, read planes,
      // path nodes — runs once per CANDIDATE of the slowest lane, not once per state: in one loop with the test, every
      // iteration found some lane with a candidate and the wave paid the heavy path ~30 times per entry.
      GmxSeedState ss;
      bool have = false;
      while (seed_left != 0) {
        ss = gmx_seed_state(ix.seed_words + seed_off);
        if (!(ss.text() && ctx_dead(ss.ctx))) {
          have = true;
          break;
        }
        seed_off += ss.words();
        --seed_left;
      }
      if (!have) break;
      if (got && sp + 2u > GMX_STACK_DEPTH) break;  // (no room for another survivor: the rest of the entry on a later visit)
      const uint32_t lo = ss.lo, hi = ss.hi, nt = ss.nt, ng = ss.ng;
      if (ss.text() || lo == hi) {  // (one position: in text form in the device copy of the entries)
        const uint32_t tp = ss.text() ? lo : ix.sa[lo];
        if (!seed_text_alive(ix, rd, tp, seed_pos, stop)) {
          seed_off += ss.words();
          --seed_left;
          continue;
        }
        uint32_t xa, xb, xt, xg, xp, xm;
        if (!next_seed(ix, !got, xa, xb, xt, xg, xp, xm)) break;  // (path nodes: arena full -> the task overflows)
        take(tp, GMX_TEXT_MARK, xt, xg);
        continue;
      }
      if (nt == 0 && ng == 0 && hi - lo < 32u && seed_pos > stop) {
        if (!got) {
          if (n_out == mark_out) arena_n = mark_arena;  // (as next_seed: the state before left nothing behind)
          mark_arena = arena_n;
          mark_out = n_out;
        }
        seed_off += 4u;
        --seed_left;
        for (uint32_t i = lo; i <= hi;) {
          if (ix.sa_ctx) {  // the same two phases over the occurrences: consecutive context words first
            while (i <= hi && ctx_dead(ix.sa_ctx[i])) ++i;
            if (i > hi) break;
          }
          const uint32_t tp = ix.sa[i++];
          if (seed_text_alive(ix, rd, tp, seed_pos, stop)) take(tp, GMX_TEXT_MARK, GMX_NIL, GMX_NIL);
        }
        continue;
      }
      uint32_t xa, xb, xt, xg, xp, xm;  // anything else (an interval state with paths): as it is
      if (!next_seed(ix, !got, xa, xb, xt, xg, xp, xm)) break;
      take(xa, xb, xt, xg);
    }
    return got && status == GMX_TASK_MAPPED;
  }
  __device__ __forceinline__ bool park(uint32_t a, uint32_t b, uint32_t tvd, uint32_t tvg, uint32_t pos, uint32_t mode) {
    if (n_out >= out_cap) return false;
    reinterpret_cast<GmxParked *>(out)[n_out++] = GmxParked{a, b, tvd, tvg, pos | (mode << 30)};
    return true;
  }
  __device__ __forceinline__ bool pop(uint32_t &a, uint32_t &b, uint32_t &tvd, uint32_t &tvg, uint32_t &pos, uint32_t &mode) {
    if (sp == 0) return false;
    --sp;
    const uint32_t *e = gmx_lds + (sp * GMX_STACK_WORDS) * GMX_BLOCK + threadIdx.x;
    a = e[0];
    b = e[GMX_BLOCK];
    tvd = e[2 * GMX_BLOCK];
    tvg = e[3 * GMX_BLOCK];
    uint32_t pm = e[4 * GMX_BLOCK];
    pos = pm & 0x3FFFFFFFu;
    mode = pm >> 30;
    return true;
  }
  __device__ __forceinline__ bool push(uint32_t a, uint32_t b, uint32_t tvd, uint32_t tvg, uint32_t pos, uint32_t mode) {
    if (sp >= GMX_STACK_DEPTH) return false;
    uint32_t *e = gmx_lds + (sp * GMX_STACK_WORDS) * GMX_BLOCK + threadIdx.x;
    e[0] = a;
    e[GMX_BLOCK] = b;
    e[2 * GMX_BLOCK] = tvd;
    e[3 * GMX_BLOCK] = tvg;
    e[4 * GMX_BLOCK] = pos | (mode << 30);
    ++sp;
    return true;
  }
  // The first final state of a task stays in registers when it is a text-form one (defer_first: extend kernel, flat
  // PRG): almost every task ends with exactly that one state and leaves as a compact record, which carries all the
  // coverage kernel needs — its copy in finals[] would be one scattered store per task that nobody reads. It is written
  // when a second state arrives (flush_first) or when the task turns out not to be compact (finish_lane).
  bool defer_first, first_deferred;
  __device__ __forceinline__ bool inst_put(const GmxFinalState &st) {
    const uint32_t at = atomicAdd(inst_count, 1u);
    if (at >= inst_cap) return false;
    inst_states[at] = st;
    return true;
  }
  __device__ __forceinline__ bool flush_first() {
    if (first_deferred) {
      first_deferred = false;
      const GmxFinalState st{first_pos, GMX_TEXT_MARK, first_tvd, first_tvg};
      if (inst_states) return inst_put(st);
      out[0] = st;
    }
    return true;
  }
  __device__ __forceinline__ bool emit(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
    if (parking) return park(lo, hi, tvd, tvg, park_pos, GMX_MODE_STATE);
    if (inst_states) {  // the first text-form state waits in registers like everywhere; the others go to the slot at once
      if (n_out == 0 && hi == GMX_TEXT_MARK) {
        first_pos = lo;
        first_tvd = tvd;
        first_tvg = tvg;
        first_deferred = true;
        n_out = 1;
        return true;
      }
      ++n_out;
      return flush_first() && inst_put(GmxFinalState{lo, hi, tvd, tvg});
    }
    if (n_out >= out_cap) return false;
    if (n_out == 0 && hi == GMX_TEXT_MARK) {
      first_pos = lo;
      first_tvd = tvd;
      first_tvg = tvg;
      if (defer_first) {
        first_deferred = true;
        n_out = 1;
        return true;
      }
    }
    flush_first();
    out[n_out++] = GmxFinalState{lo, hi, tvd, tvg};
    return true;
  }
  __device__ __forceinline__ uint32_t alloc_node(uint32_t site, int32_t allele, uint32_t next) {
    if (arena_n >= GMX_FAST_ARENA) return GMX_NIL;
    // node k of a task lives at arena[k * stride], arena = the table's base + task: node k of neighbouring tasks share
    // cache lines (a wave's 64 first-node stores touch ~24 lines instead of 64), and the handle is the offset itself, so
    // every reader keeps indexing arena[handle] from the task's base
    const uint32_t h = arena_first + arena_n * arena_stride;
    arena[h] = GmxPathNode{site, allele, next};
    ++arena_n;
    return h;
  }
  __device__ __forceinline__ uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) {
    if (allele == -1) {  // traversing path: a single entered site stays inline in the handle (no node, no load to pop)
      if (next == GMX_NIL) return GMX_INLINE_FLAG | ((site - 5u) >> 1);
      if (gmx_h_inline(next)) {
        next = alloc_node(gmx_h_site(arena, next), -1, GMX_NIL);
        if (next == GMX_NIL) return GMX_NIL;
      }
    }
    return alloc_node(site, allele, next);
  }
  __device__ __forceinline__ uint32_t arena_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ uint32_t arena_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ void fail(uint32_t s) {
    if (status == GMX_TASK_MAPPED || s == GMX_TASK_ERROR) status = s;
  }
};

#define GMX_BIG_LDS_DEPTH 16u  // pending entries of the large-capacity pass kept in LDS (most of its tasks need no more)
struct BigCtx {  // the same DFS queue with everything in global memory and runtime capacities (large-capacity pass)
  __device__ __forceinline__ bool more_seeds() const { return false; }
  __device__ __forceinline__ bool next_seed(const GmxIndexView &, bool, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &,
                                            uint32_t &) {
    return false;
  }
  template <class Reader>
  __device__ __forceinline__ bool next_seed_screened(const GmxIndexView &, Reader &, uint32_t, uint32_t &, uint32_t &, uint32_t &, uint32_t &,
                                                     uint32_t &, uint32_t &) {
    return false;
  }
  uint32_t sp, cap;
  uint32_t *stack;  // cap x GMX_STACK_WORDS; the first GMX_BIG_LDS_DEPTH entries live in LDS instead (lane-strided, 64-lane blocks)
  GmxPathNode *arena;
  uint32_t arena_n, arena_cap;
  uint32_t status;
  GmxFinalState *out;
  uint32_t n_out, out_cap;
  __device__ __forceinline__ bool pop(uint32_t &a, uint32_t &b, uint32_t &tvd, uint32_t &tvg, uint32_t &pos, uint32_t &mode) {
    if (sp == 0) return false;
    --sp;
    uint32_t pm;
    if (sp < GMX_BIG_LDS_DEPTH) {
      const uint32_t *e = gmx_lds + (sp * GMX_STACK_WORDS) * 64 + (threadIdx.x & 63);
      a = e[0];
      b = e[64];
      tvd = e[128];
      tvg = e[192];
      pm = e[256];
    } else {
      const uint32_t *e = stack + (size_t)sp * GMX_STACK_WORDS;
      a = e[0];
      b = e[1];
      tvd = e[2];
      tvg = e[3];
      pm = e[4];
    }
    pos = pm & 0x3FFFFFFFu;
    mode = pm >> 30;
    return true;
  }
  __device__ __forceinline__ bool push(uint32_t a, uint32_t b, uint32_t tvd, uint32_t tvg, uint32_t pos, uint32_t mode) {
    if (sp >= cap) return false;
    if (sp < GMX_BIG_LDS_DEPTH) {
      uint32_t *e = gmx_lds + (sp * GMX_STACK_WORDS) * 64 + (threadIdx.x & 63);
      e[0] = a;
      e[64] = b;
      e[128] = tvd;
      e[192] = tvg;
      e[256] = pos | (mode << 30);
    } else {
      uint32_t *e = stack + (size_t)sp * GMX_STACK_WORDS;
      e[0] = a;
      e[1] = b;
      e[2] = tvd;
      e[3] = tvg;
      e[4] = pos | (mode << 30);
    }
    ++sp;
    return true;
  }
  __device__ __forceinline__ bool emit(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
    if (n_out >= out_cap) return false;
    out[n_out++] = GmxFinalState{lo, hi, tvd, tvg};
    return true;
  }
  __device__ __forceinline__ uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) {
    if (arena_n >= arena_cap) return GMX_NIL;
    arena[arena_n] = GmxPathNode{site, allele, next};
    return arena_n++;
  }
  __device__ __forceinline__ uint32_t arena_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ uint32_t arena_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ void fail(uint32_t s) {
    if (status == GMX_TASK_MAPPED || s == GMX_TASK_ERROR) status = s;
  }
};

// seed-table index (gmx_types.h GmxSeed) of oriented positions [start, start + k): RIGHTMOST base most significant
template <class Reader>
__device__ __forceinline__ uint32_t kmer_code(Reader &r, uint32_t start, uint32_t k) {
  uint32_t code = 0;
  for (uint32_t j = 0; j < k; ++j) code |= (r.at(start + j) - 1u) << (2u * j);
  return code;
}

// k-mer code of the read's LAST k oriented bases (the seed, quasimap.cpp:235-241) from one 32-base window of the
// bit planes instead of k single-base extractions: forward reads take the last k raw bases as they lie (leftmost base
// least significant), reverse-complement reads the first k raw bases complemented and in reverse bit order.
__device__ __forceinline__ uint32_t spread_even(uint32_t x) {  // bit i -> bit 2i (i < 16)
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}
template <class Reader>
__device__ __forceinline__ uint32_t last_kmer_code(Reader &r, uint32_t k) {
  uint32_t lo, hi;
  const uint32_t mask = (1u << k) - 1u;  // k <= 15
  if (r.rc) {
    r.planes(0, lo, hi);
    lo = __builtin_bitreverse32(~lo & mask) >> (32u - k);
    hi = __builtin_bitreverse32(~hi & mask) >> (32u - k);
  } else {
    r.planes(r.len - k, lo, hi);
    lo &= mask;
    hi &= mask;
  }
  return spread_even(lo) | (spread_even(hi) << 1);
}

// all_read_kmers_occur_in_index (quasimap.cpp:212-225); `bitmap` is the presence bitmap in global memory or LDS
__device__ bool all_kmers_present(const uint32_t *bitmap, uint32_t k, ReadRef &r) {
  uint32_t code = kmer_code(r, 0, k);
  for (uint32_t o = 0;;) {  // four independent bitmap probes in flight per round
    uint32_t present = 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      present &= bitmap[code >> 5] >> (code & 31);
      if (o + k >= r.len) return present & 1u;
      code = (code >> 2) | ((r.at(o + k) - 1u) << (2u * (k - 1u)));
      ++o;
    }
    if (!(present & 1u)) return false;
  }
}

// seeds the context from the k-mer index entry of the read's last k-mer (quasimap.cpp:235-241);
// Device copies of the seed tables: in a multi-state entry (a = GMX_SEED_COMPLEX) the word offset b carries two flags
// set once at upload (gmx_seed_mark_kernel), so that gmx_seed_kernel decides without reading the entry's words:
//   GMX_SEEDF_BIG    the task goes to the large-capacity pass: the entry holds a path-less state over more positions
//                    than the fast pass takes apart (a repeat), or more than 65535 states
//   GMX_SEEDF_EMPTY  no state
#define GMX_SEEDF_BIG 0x80000000u
#define GMX_SEEDF_EMPTY 0x40000000u
#define GMX_SEED_OFF(b) ((b) & 0x3FFFFFFFu)
__device__ __forceinline__ const uint32_t *gmx_seed_entry(const GmxIndexView &ix, uint32_t b) {
  return ix.seed_words + ((size_t)GMX_SEED_OFF(b) << ix.seed_shift);
}
#define GMX_SEED_SPLIT_MAX ((uint32_t)GMX_STACK_DEPTH - 1u)  // a path-less seed state over 2 .. 4 positions is taken apart in the fast pass (stack of 5)
// A single path-less state over ONE suffix-array position is stored in text form — a = its PRG position, b =
// GMX_TEXT_MARK — in the device copies: the search needs no suffix-array look-up to start (one dependent, always-missing
// fetch per task less: 64 MB of the extend kernel's 390 MB of fabric-side fetch at config[1]).
// The same inside the multi-state entries (round 4): a state over one suffix-array position — with or without paths — is
// rewritten in the device copy of the words as (PRG position, left context, flag; GmxSeedState). A whole-genome index has
// ~16 states per k-mer and all but one of a read's seed states die at their first compare: the suffix-array look-up and
// the text record were two scattered fetches per state — at 160 GB of index the kernels ran at the memory system's rate of
// scattered lines — and the left context rejects almost all of them from the entry's own, consecutive, words. Every device
// reader of the entries goes through gmx_seed_state().
__global__ void gmx_seed_mark_kernel(GmxSeed *seeds, uint64_t n, uint32_t *seed_words, uint32_t seed_shift, const uint32_t *sa,
                                     const GmxTextRec *text) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const GmxSeed s = seeds[i];
    if (s.a != GMX_SEED_COMPLEX) {
      if (s.a == s.b) seeds[i] = GmxSeed{sa[s.a], GMX_TEXT_MARK};
      continue;
    }
    uint32_t *w = seed_words + ((size_t)s.b << seed_shift);
    const uint32_t ns = *w++;
    bool big = ns > 0xFFFFu;
    for (uint32_t j = 0; j < ns; ++j) {
      const uint32_t nt = w[2], ng = w[3];
      big = big || (w[1] >= w[0] + GMX_SEED_SPLIT_MAX && nt == 0 && ng == 0);
      if (w[0] == w[1]) {  // one position: PRG position + left context (GmxSeedState)
        const uint32_t tp = sa[w[0]];
        w[0] = tp;
        w[1] = gmx_left_context(text, tp);
        w[2] = nt | GMX_SEEDST_TEXT;
      }
      w += 4 + 2 * nt + ng;
    }
    seeds[i].b = s.b | (big ? GMX_SEEDF_BIG : 0u) | (ns == 0 ? GMX_SEEDF_EMPTY : 0u);
  }
}

// sa_ctx[i] = left context of text position sa[i] (GmxIndexView::sa_ctx): the occurrences [lo, hi] of a path-less seed state
// are screened from hi - lo + 1 CONSECUTIVE words instead of a suffix-array look-up and a text record each.
__global__ void gmx_sa_ctx_kernel(const uint32_t *sa, const GmxTextRec *text, uint64_t n, uint32_t *out) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = gmx_left_context(text, sa[i]);
}

// push(lo, hi, tvd, tvg) receives every seed state
template <class Ctx, class Push>
__device__ void load_seed(const GmxIndexView &ix, const GmxSeed *table, uint32_t code, Ctx &ctx, Push push) {
  GmxSeed s = table[code];
  if (s.a != GMX_SEED_COMPLEX) {
    if (s.a <= s.b) push(s.a, s.b, GMX_NIL, GMX_NIL);
    return;
  }
  const uint32_t *p = gmx_seed_entry(ix, s.b);
  uint32_t ns = *p++;
  for (uint32_t i = 0; i < ns; ++i) {
    const GmxSeedState ss = gmx_seed_state(p);
    uint32_t lo = ss.lo, hi = ss.hi, nt = ss.nt, ng = ss.ng;
    p += 4;
    uint32_t tvd = GMX_NIL, tvg = GMX_NIL;
    bool ok = true;
    for (uint32_t j = 0; j < nt; ++j, p += 2) {
      if (!ok) continue;
      uint32_t nn = ctx.arena_new(p[0], (int32_t)p[1], tvd);
      if (nn == GMX_NIL) ok = false; else tvd = nn;
    }
    for (uint32_t j = 0; j < ng; ++j, ++p) {
      if (!ok) continue;
      uint32_t nn = ctx.arena_new(p[0], -1, tvg);
      if (nn == GMX_NIL) ok = false; else tvg = nn;
    }
    if (!ok || !push(lo, hi, tvd, tvg)) {
      ctx.fail(GMX_TASK_OVERFLOW);
      return;
    }
  }
}

#define GMX_SEED_PUSH_MAX 4u  // multi-state k-mer entries up to this size are pushed at once, larger ones use the seed cursor
// FastCtx: a single path-less state is pushed, a small multi-state entry too, a large one arms the seed cursor.
// CURSOR = false (engines whose index has hardly any large entry): every entry is pushed; one that does not fit
// the stack overflows to the large-capacity pass.
template <bool CURSOR>
__device__ __forceinline__ void load_seed_cursor(const GmxIndexView &ix, const GmxSeed s, FastCtx &ctx, uint32_t from) {
  if (s.a != GMX_SEED_COMPLEX) {
    if (s.a < s.b && s.b != GMX_TEXT_MARK && s.b - s.a < GMX_SEED_SPLIT_MAX && from > 0) {
      // a few occurrences (a short repeat): position by position in text form — the same results (see
      // gmx_search_big_kernel), 32 bases per step instead of one rank block per base and 137 iterations of the wave
      for (uint32_t i = s.a; i <= s.b; ++i) ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
    } else if (s.a <= s.b) {  // (one occurrence: already in text form in the device copy, gmx_seed_mark_kernel)
      ctx.push(s.a, s.b, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
    }
    return;
  }
  const uint32_t ns = *gmx_seed_entry(ix, s.b);
  if (ns > 0xFFFFu) {
    ctx.fail(GMX_TASK_OVERFLOW);
    return;
  }
  ctx.seed_off = (uint64_t)(gmx_seed_entry(ix, s.b) - ix.seed_words) + 1;
  ctx.seed_pos = from;
  ctx.seed_left = ns;
  ctx.mark_arena = ctx.arena_n;
  ctx.mark_out = ctx.n_out;
  if (!CURSOR || ns <= GMX_SEED_PUSH_MAX) {  // all on the stack at once (no dependent index fetch between them)
    uint32_t a, b, tvd, tvg, pos, mode;
    while (ctx.seed_left && ctx.next_seed(ix, false, a, b, tvd, tvg, pos, mode)) {
      bool ok = true;
      if (tvd == GMX_NIL && tvg == GMX_NIL && b > a && b != GMX_TEXT_MARK && b - a < GMX_SEED_SPLIT_MAX && pos > 0) {
        for (uint32_t i = a; i <= b && ok; ++i) ok = ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, pos, mode);  // as above
      } else {
        ok = ctx.push(a, b, tvd, tvg, pos, mode);
      }
      if (!ok) {
        ctx.fail(GMX_TASK_OVERFLOW);
        ctx.seed_left = 0;
      }
    }
  }
}

// Wave-level driver of the DFS queue (gmx_dfs.h). All lanes spin in the cheap fast iteration; a lane that needs
// the general iteration (marker hit, state death/finish, wide interval) waits, and the general code runs for the
// whole wave only when GMX_SLOW_BATCH lanes are waiting or nobody can go fast — so its ~10x higher instruction
// count is amortised instead of being executed (mostly masked off) on every step.
#define GMX_SLOW_BATCH 12
#define GMX_WAVE_SEED 7u  // a light kind of the wave loop only (gmx_dfs.h kinds are 0..6)
#ifdef GMX_LOOP_STATS
// Debug build only (-DGMX_LOOP_STATS): iteration mix of the wave loop, summed over all kernels using it.
//   [0] fast iterations  [1] heavy TEXT  [2] heavy HIT  [3] heavy WIDE  [4] light only  [5] slow iterations
//   [6] lanes served by fast heavy kinds  [7] lanes served by slow iterations  [8] waves  [9] light lanes
//   [10..12] clocks of prologue / loop / epilogue  [13] lanes holding a state, summed over the fast iterations
__device__ unsigned long long gmx_loop_stats[48];  // x3: probe, extend, large-capacity kernel
extern "C" int gmx_debug_loop_stats(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_loop_stats), sizeof(gmx_loop_stats)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_loop_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#define GMX_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gmx_loop_stats[KID * 16 + (i)], (unsigned long long)(v)); } while (0)
#else
#define GMX_STAT(i, v) do { } while (0)
#endif
#ifdef GMX_LOOP_STATS
#define GMX_CLK() clock64()
#define GMX_TSTAT(kid, i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gmx_loop_stats[(kid) * 16 + (i)], (unsigned long long)(v)); } while (0)
#else
#define GMX_CLK() 0ll
#define GMX_TSTAT(kid, i, v) do { } while (0)
#endif
#ifndef GMX_KIND_SHARE
#define GMX_KIND_SHARE 64
#endif
// GMX_KIND_SHARE: a heavier kind runs in an iteration when it holds at least 1/GMX_KIND_SHARE of the heavy lanes.
// Measured on MI355X: the loop is latency-bound, so running every kind present (64) beats gathering lanes (4).
// `fuse`: transitions that need no fetch of their own do not cost the lane an iteration. A resolved marker hit (or a
// converted width-one interval) that continues in text form takes its text step in the SAME iteration (one more fetch
// for those lanes, the compare code runs once for all), and a state that died or reached the stop position is replaced
// by the lane's next pending entry at the end of the iteration. A lane's chain shrinks from (text steps + marker hits +
// emits + pops) iterations to about its text steps; the wave runs as long as its slowest lane.
template <int KID, bool CURSOR, class Ctx, class Reader>
__device__ void dfs_run_wave(const GmxIndexView &ix, Ctx &ctx, Reader &rd, uint32_t stop, bool active, uint32_t budget,
                             GmxLane &ln, bool fuse = false) {
  ln.a = ln.b = ln.tvd = ln.tvg = ln.pos = ln.mode = 0;
  ln.have = active && ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
  bool wait_slow = false;
  uint32_t iters = 0;
  GMX_STAT(8, 1);
  for (;;) {
    // ---- fast phase. Of the three heavier kinds an iteration runs those that hold a fair share of the lanes
    // (a wave-uniform choice: the code of the others is branched over, not masked off), plus the cheap kinds
    // (convert / emit / pop). Lanes of a kind with few takers wait until it has gathered more; lanes that need
    // the general iteration wait for the slow phase. `budget` (probe kernel) bounds the number of iterations:
    // whatever is still pending then is parked and continues in the compacted extend kernel.
    unsigned long long mf, ms;
    for (;;) {
      uint32_t kind = wait_slow ? GMX_FAST_NONE : gmx_dfs_fast_kind(ln, stop);
      if (CURSOR && !ln.have && ctx.more_seeds()) kind = GMX_WAVE_SEED;  // stack empty: the next state of the k-mer index entry
      const unsigned long long m_text = __ballot(kind == GMX_FAST_TEXT), m_hit = __ballot(kind == GMX_FAST_HIT),
                               m_wide = __ballot(kind == GMX_FAST_WIDE),
                               m_light = __ballot(kind == GMX_FAST_CONVERT || kind == GMX_FAST_EMIT || kind == GMX_FAST_POP ||
                                                  kind == GMX_WAVE_SEED);
      ms = __ballot(ln.have && kind == GMX_FAST_NONE);
      mf = m_text | m_hit | m_wide | m_light;
      if (mf == 0 || __popcll(ms) >= GMX_SLOW_BATCH) break;
      if (budget && iters >= budget) return;
      ++iters;
      const uint32_t n_text = (uint32_t)__popcll(m_text), n_hit = (uint32_t)__popcll(m_hit), n_wide = (uint32_t)__popcll(m_wide);
      const uint32_t n_heavy = n_text + n_hit + n_wide;
      const bool run_text = n_text && n_text * GMX_KIND_SHARE >= n_heavy, run_hit = n_hit && n_hit * GMX_KIND_SHARE >= n_heavy,
                 run_wide = n_wide && n_wide * GMX_KIND_SHARE >= n_heavy;
      GMX_STAT(0, 1);
      {
        const uint32_t n_live = (uint32_t)__popcll(__ballot(ln.have));  // lanes that hold a search state in this iteration
        GMX_STAT(13, n_live);
      }
      GMX_STAT(1, run_text);
      GMX_STAT(2, run_hit);
      GMX_STAT(3, run_wide);
      GMX_STAT(4, n_heavy == 0);
      GMX_STAT(6, (run_text ? n_text : 0) + (run_hit ? n_hit : 0) + (run_wide ? n_wide : 0));
      GMX_STAT(9, __popcll(m_light));
      // all fetches of the iteration are issued before any of them is consumed
      uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
      uint32_t sa_val = 0;
      if (m_light && kind == GMX_FAST_CONVERT) sa_val = ix.sa[ln.a];
      if (run_text && kind == GMX_FAST_TEXT) {  // one 32-byte record: 64 symbols of the PRG
        const uint4 *src = reinterpret_cast<const uint4 *>(ix.text + gmx_dfs_text_rec(ln));
        q0 = src[0];
        q1 = src[1];
      }
      if (run_hit && kind == GMX_FAST_HIT) q0 = *reinterpret_cast<const uint4 *>(gmx_dfs_hit_sub(ix, rd, ln));
      if (run_wide && kind == GMX_FAST_WIDE) {
        const uint4 *src = reinterpret_cast<const uint4 *>(ix.blocks + (ln.a >> GMX_BLK_SHIFT));
        q0 = src[0];
        q1 = src[1];
        q2 = src[2];
        q3 = src[3];
      }
      if (m_light) {
        if (kind == GMX_FAST_CONVERT) {
          ln.a = sa_val;
          ln.b = GMX_TEXT_MARK;
        } else if (kind == GMX_FAST_EMIT) {
          gmx_dfs_emit(ctx, ln);
        } else if (kind == GMX_FAST_POP) {
          gmx_dfs_pop(ctx, ln);
        } else if (CURSOR && kind == GMX_WAVE_SEED) {
          ln.have = ctx.next_seed_screened(ix, rd, stop, ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
        }
      }
      auto text_rec = [&]() {
        return GmxTextRec{(uint64_t)q0.x | ((uint64_t)q0.y << 32), (uint64_t)q0.z | ((uint64_t)q0.w << 32),
                          (uint64_t)q1.x | ((uint64_t)q1.y << 32), q1.z, q1.w};
      };
      if (!fuse) {
        if (run_text && kind == GMX_FAST_TEXT && !gmx_dfs_text_apply(ctx, ln, stop, rd, text_rec())) wait_slow = true;
        if (run_hit && kind == GMX_FAST_HIT && !gmx_dfs_fast_hit(ctx, ln, stop, GmxHitSub{q0.x, q0.y, q0.z, q0.w})) wait_slow = true;
      } else {
        // marker hits first: what they continue as takes its text step below
        if (run_hit && kind == GMX_FAST_HIT && !gmx_dfs_fast_hit(ctx, ln, stop, GmxHitSub{q0.x, q0.y, q0.z, q0.w})) wait_slow = true;
        bool text_now = run_text && kind == GMX_FAST_TEXT;
        const bool late = !wait_slow && (kind == GMX_FAST_HIT || kind == GMX_FAST_CONVERT) &&
                          gmx_dfs_fast_kind(ln, stop) == GMX_FAST_TEXT;
        const unsigned long long m_late = __ballot(late);
        GMX_STAT(6, __popcll(m_late));  // text steps taken in the iteration that resolved their marker hit
        if (m_late) {
          if (late) {
            const uint4 *src = reinterpret_cast<const uint4 *>(ix.text + gmx_dfs_text_rec(ln));
            q0 = src[0];
            q1 = src[1];
            text_now = true;
          }
        }
        if (text_now && !gmx_dfs_text_apply(ctx, ln, stop, rd, text_rec())) wait_slow = true;
      }
      if (run_wide && kind == GMX_FAST_WIDE) {
        const uint32_t w[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        if (!gmx_dfs_fast_wide(ix, rd, ln, w)) wait_slow = true;
      }
      if (fuse) {  // a state that died or reached the stop position: the lane's next pending entry, now
        const uint32_t k2 = wait_slow || !(kind == GMX_FAST_TEXT || kind == GMX_FAST_HIT || kind == GMX_FAST_WIDE || kind == GMX_FAST_CONVERT)
                                ? GMX_FAST_NONE
                                : gmx_dfs_fast_kind(ln, stop);
        const unsigned long long m_tail = __ballot(k2 == GMX_FAST_EMIT || k2 == GMX_FAST_POP);
        GMX_STAT(9, __popcll(m_tail));
        if (m_tail) {
          if (k2 == GMX_FAST_EMIT)
            gmx_dfs_emit(ctx, ln);
          else if (k2 == GMX_FAST_POP)
            gmx_dfs_pop(ctx, ln);
        }
      }
    }
    if ((mf | ms) == 0) break;
    if (budget && iters >= budget) return;
    ++iters;
    // ---- one general iteration for every waiting lane ----
    GMX_STAT(5, 1);
    GMX_STAT(7, __popcll(ms));
    if (ln.have && (wait_slow || gmx_dfs_fast_kind(ln, stop) == GMX_FAST_NONE)) {
      gmx_dfs_slow_iter(ix, ctx, rd, stop, ln);
      wait_slow = false;
    }
  }
}

struct BatchView {
  const uint8_t *reads;      // caller's buffer: one byte per base (null when the caller handed over bit planes)
  const uint64_t *offsets;   // n_reads + 1 base offsets; null when uniform_len != 0
  const uint32_t *seeds;
  const uint8_t *skip;       // per read: holds a non-ACGT byte (null: no such read in the batch)
  const uint2 *packed;       // bit planes: written by gmx_pack_kernel, or uploaded as they are (gmx_map_reads_packed_host);
                             // read r starts at pair pack_off(r)
  uint32_t n_reads;
  uint32_t forward_only;
  uint32_t uniform_len;      // != 0: every read has this many bases and starts at pair r * pairs_per_read (no offsets)
  uint32_t pairs_per_read;   // ceil(uniform_len / 32)
  uint32_t keep_states;      // test hook (gmx_engine_debug_keep_states): every task's final states stay readable in finals[] / n_final[]
};
// Layout of the bit planes (include/gmx.h, gmx_pack_reads): P(r) = (offsets[r] >> 5) + r pairs from P(0) — ceil(len/32)
// pairs fit between consecutive starts whatever the offsets are, and a sub-range of a packed batch is again a packed
// batch (the host feed uploads chunks of one); reads of one length are packed back to back.
__device__ __forceinline__ uint64_t pack_off(const BatchView &b, uint32_t read) {
  if (b.uniform_len) return (uint64_t)read * b.pairs_per_read;
  return ((b.offsets[read] >> 5) - (b.offsets[0] >> 5)) + read;
}
__device__ __forceinline__ uint32_t read_len(const BatchView &b, uint32_t read) {
  return b.uniform_len ? b.uniform_len : (uint32_t)(b.offsets[read + 1] - b.offsets[read]);
}
__device__ __forceinline__ bool read_skipped(const BatchView &b, uint32_t read) { return b.skip && b.skip[read]; }

struct SearchOut {
  uint32_t *status;          // per task
  uint32_t *n_final;         // per task
  GmxFinalState *finals;     // per task x GMX_FAST_STATES
  GmxPathNode *arena;        // GMX_FAST_ARENA x arena_stride: node k of task t at [k * arena_stride + t] (FastCtx::alloc_node)
  GmxCoverRec *cover_recs;   // GMX_REGIONS queues x region_cap records of single-instance mapped tasks, by PRG region;
  uint32_t *cover_rec_task;  // their task ids (error reporting); counters [16 + r]
  uint32_t region_cap;       // capacity of one region list
  uint32_t region_inv;       // ceil(2^32 * GMX_REGIONS / n_prg): region = umulhi(position, region_inv)
  // The six task-id queues finish_lane appends to are slices of ONE allocation, `task_lists` (slice q at q * list_stride,
  // GMX_TL_*), and finish_lane addresses them as base + integer index: a lane-divergent chain of selects between six
  // queue POINTERS held in spilled SGPRs is what the compiler got wrong in round 2 (HISTORY.md §4.5: the VGPR copy of the
  // cover_general_list pointer was emitted in a sibling block, under another exec mask). The named members below point
  // into the same allocation for the kernels that read one queue.
  uint32_t *task_lists;
  uint32_t list_stride;
  uint32_t *overflow_list;   // task ids to re-run with large capacities (from the probe kernel); counter [1]
  uint32_t *overflow2_list;  // the same from the extend kernel; counter [9]
  uint32_t *cover_overflow_list;  // mapped_list entries whose selection needs the large scratch
  uint32_t *big_mapped_list;  // big-pass slots with final states (bit 31 set); counter [7]
  uint32_t *cover_mid_list;      // general tasks whose selection did not fit the LDS scratch; counter [13]
  uint32_t *cover_general_list;  // mapped_list entries that are not single-instance tasks; counter [8]
#ifdef GMX_SEARCHOUT_ALT  // test build (tools/searchout_alt.sh): the member order that broke gmx_probe_kernel in round 2
  unsigned long long *stats;
#endif
  uint32_t *alive_list;      // tasks that survived the probe phase (states parked in `finals`)
  uint32_t *dead_list;       // tasks without final state, to be classified by the k-mer filter: the probe kernel's (counter [6])
  uint32_t *dead2_list;      // ... and the extend kernel's (counter [12]); one filter pass each
  uint64_t *seed_cursor;     // per task: word offset into seed_words of the next seed state (when n_final's bits 16.. > 0)
  uint32_t *error;           // [0] = first error status, [1] = its task (persist until gmx_engine_sync reads them)
  uint32_t *counters;        // [0] = n mapped_list, [1] = n overflow_list, [2] = first error status, [3] = error task,
                             // [4] = n cover_overflow_list, [5] = n alive_list, [6] = n dead_list
  GmxSeed *alive_seed;       // gmx_seed_kernel: the seed directory entry of alive_list[i]
  uint32_t *huge_list;       // tasks the large-capacity pass could not hold (pools or slots exhausted); counter [11]
  uint32_t *cover_huge_list; // entries whose selection exceeded the largest fixed scratch; counter [15]
  uint32_t *huge_retry;      // last tier: work items its 64-wide round could not finish (run again alone with the whole heap)
  uint32_t arena_stride;     // tasks the per-task tables were allocated for
  // Reads in short repeats: a path-less seed over 6 .. 64 suffix-array positions is taken apart into one INSTANCE per
  // position (gmx_seed_kernel), each searched by a lane of its own like any other task (gmx_extend_inst_kernel); the task
  // owns a large-capacity slot in which the instances' final states and path nodes meet. Counter [24] = instances.
  uint32_t *inst_list;       // per instance: slot << 6 | index of the instance within its task
  uint32_t *inst_sa;         // per instance: suffix-array index of its occurrence
  uint32_t *inst_remaining;  // per slot: instances still running; bit 31: one of them failed (pools exceeded)
  uint32_t inst_cap;         // capacity of inst_list / inst_sa
  uint32_t inst_slots;       // slots available (BigOut::max_slots)
  uint32_t *slot_n_final, *slot_task;  // BigOut::n_final / task_of_slot
  uint32_t *inst_mapped_list;          // GMX_ENTRY_INST | slot of the instance-searched tasks with final states; counter [25]
  // the instances' own pools, dense in the instance index (a task's instances are consecutive): GMX_FAST_ARENA path
  // nodes and GMX_INST_STATES final states per instance. (In the large-capacity slots — 40 KB apart, gigabytes of address
  // space — every lane paid TLB misses: an instance lane took ten times as long as a regular one.)
  GmxPathNode *inst_arena;
  GmxFinalState *inst_states;
  uint32_t *inst_first;                // per slot: instance index of the task's first instance
  uint32_t *inst_remaining_width;      // per slot: number of instances
  uint32_t *inst_serial_list;          // entries of inst_mapped_list the cooperative coverage kernel left to the serial one; counter [26]
  uint32_t *general_serial_list;       // the same for cover_general_list; counter [27]
  uint32_t *big_serial_list;           // ... and for the second part of big_mapped_list (coverage instance 2); counter [28]
  uint32_t *overflow3_list;            // tasks one lane has to search with a whole large-capacity slot (a group's parts did not suffice); counter [29]
  uint32_t split_twice;                // the extend kernel's overflow queue goes through the split search as well
  // A task that finds the grouped log full (sites with more than 8 alleles) has recorded nothing: its queue entry goes
  // to one of these lists, the host drains the log after the batch and has the entries redone (launch_log_replay).
  uint32_t *log_retry_list;            // coverage queue entries (task / large-capacity slot / instance slot); counter [30]
  uint32_t *log_retry_recs;            // compact records, as index into cover_recs; counter [31]
  uint32_t *log_retry_huge;            // tasks of the last tier's search; counter [33]
  uint32_t *general_rest_list;         // entries of cover_general_list that gmx_cover_one_kernel left to the general instances; counter [34]
  uint32_t *single_rest_list;          // compact records (index into cover_recs) gmx_cover_jump_kernel declined; counter [38]
  // Stragglers: the extend kernel's wave loop has an iteration budget; a lane with work left then (a read inside an MSA
  // region takes fifty iterations, its 63 neighbours five) parks its pending entries and goes to a second, compacted pass.
  GmxParked *park2;                    // per task: up to GMX_STACK_DEPTH pending entries (its final states stay in finals[])
  uint32_t *park2_n;                   // per task: how many
#ifndef GMX_SEARCHOUT_ALT
  unsigned long long *stats; // QuasimapReadsStats (quasimap.hpp:17-24), counted where each task's fate is decided:
#endif
                             // [0] all (pack kernel) [1] skipped (seed / probe kernel) [2] missing_kmer [3] no_extension
                             // (filter kernels, large-capacity passes) [4] exact_mapped (whoever finished the search)
};

#ifndef GMX_REGIONS
#define GMX_REGIONS 8
#endif
enum : uint32_t { GMX_TL_OVERFLOW = 0, GMX_TL_OVERFLOW2, GMX_TL_ALIVE, GMX_TL_DEAD, GMX_TL_DEAD2, GMX_TL_GENERAL, GMX_TL_ALIVE2, GMX_TL_N = GMX_TL_ALIVE2 + GMX_EXTRA_PASSES };

// stats[idx] += number of threads of the block with `flag` (one global atomic per block). Every thread of the block
// must call it. `scratch` is one uint32 of LDS per call site.
__device__ __forceinline__ void gmx_block_count(unsigned long long *stats, uint32_t idx, bool flag, uint32_t *scratch) {
  if (threadIdx.x == 0) *scratch = 0;
  __syncthreads();
  const unsigned long long m = __ballot(flag);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(scratch, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && *scratch) atomicAdd(&stats[idx], (unsigned long long)*scratch);
}

__device__ __forceinline__ ReadRef task_read(const BatchView &b, uint32_t task) {
  uint32_t read = task >> 1;
  ReadRef r;
  r.w = b.packed + pack_off(b, read);
  r.len = read_len(b, read);
  r.rc = task & 1;
  r.cur_idx = 0xFFFFFFFFu;
  r.cur = make_uint2(0, 0);
  return r;
}

__device__ __forceinline__ void task_read_regs(const BatchView &b, uint32_t task, ReadRegs &r) {
  const uint32_t read = task >> 1;
  r.load(b.packed + pack_off(b, read), read_len(b, read), (task & 1) != 0);
}

// Common epilogue of the probe and extend kernels: publish the task's emitted states and queue the task.
//   done  : the whole read has been consumed (the emitted states are final, not parked)
//   parked: the task's pending entries are in SearchOut::park2 (a straggler of the extend kernel): alive whatever n_out says
//   alive_pass: which of the extend kernel's straggler lists a parked task goes to (second phase)
__device__ __forceinline__ void finish_lane(const GmxIndexView &ix, const SearchOut &o, bool active, uint32_t task, FastCtx &ctx,
                                            uint32_t status, bool done, bool second_phase, uint32_t read_len, bool parked = false,
                                            uint32_t alive_pass = 0, bool b_keep = false) {
  bool mapped = false, alive = false, dead = false, over = false;
  if (active && status != GMX_TASK_SKIPPED && status != GMX_STATUS_IGNORED) {
    if (status == GMX_TASK_MAPPED) {
      if (parked)
        alive = true;
      else if (ctx.n_out == 0 && ctx.seed_left == 0)
        dead = true;
      else {
        mapped = done;
        alive = !done;
      }
    } else if (status == GMX_TASK_OVERFLOW) {
      over = true;
    } else if (atomicCAS(&o.error[0], 0u, status) == 0u) {
      o.error[1] = task;
    }
  }
  // (the read counters — skipped reads, tasks mapped here — are tallied in the queue append below: one pair of barriers
  //  for everything the block publishes, four barriers less than counting them separately)
  // a mapped task with ONE text-form final state and a short path leaves as a compact record (GmxCoverRec)
  GmxCoverRec rec{0, 0, GMX_NIL, {0, 0, 0}, 0, 0};
  bool compact = mapped && ctx.n_out == 1 && ctx.first_pos != GMX_NIL && read_len < 0x10000u &&
                 (ctx.first_tvg == GMX_NIL || gmx_h_inline(ctx.first_tvg));
  if (compact) {
    uint32_t n = 0, alleles[3] = {0, 0, 0};
    uint32_t x = ctx.first_tvd;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (x != GMX_NIL) {
        const GmxPathNode nd = ctx.arena[x];
        rec.site[i] = nd.site;
        alleles[i] = (uint32_t)nd.allele;
        compact = compact && alleles[i] < 0x10000u;
        x = nd.next;
        ++n;
      }
    rec.p = ctx.first_pos;
    rec.tvg = ctx.first_tvg;
    if (x == GMX_NIL) {
      rec.len_n = read_len | (n << 16);
      rec.a01 = alleles[0] | (alleles[1] << 16);
      rec.a2 = alleles[2];
    } else {  // more than three loci: the run form, if the sites are consecutive and the allele ids fit a byte
      const uint32_t site0 = rec.site[0];
      bool run = rec.site[1] == site0 + 2 && rec.site[2] == site0 + 4 && (alleles[0] | alleles[1] | alleles[2]) < 256u;
      uint32_t w0 = alleles[0] | (alleles[1] << 8) | (alleles[2] << 16), w1 = 0, w2 = 0, w3 = 0;
      while (x != GMX_NIL && n < GMX_REC_RUN && run) {
        const GmxPathNode nd = ctx.arena[x];
        const uint32_t a = (uint32_t)nd.allele;
        run = nd.site == site0 + 2 * n && a < 256u;
        const uint32_t v = a << (8 * (n & 3u));
        w0 |= (n >> 2) == 0 ? v : 0u;
        w1 |= (n >> 2) == 1 ? v : 0u;
        w2 |= (n >> 2) == 2 ? v : 0u;
        w3 |= (n >> 2) == 3 ? v : 0u;
        x = nd.next;
        ++n;
      }
      compact = compact && run && x == GMX_NIL;
      rec.len_n = read_len | (n << 16) | GMX_REC_RUN_FLAG;
      rec.site[1] = w0;
      rec.site[2] = w1;
      rec.a01 = w2;
      rec.a2 = w3;
    }
  }
  if (mapped && (!compact || b_keep)) ctx.flush_first();  // the general coverage routine reads finals[]
  // the state counts of a task are read by the extend kernel (parked tasks) and by the general coverage routine; a
  // compact record needs neither (on a nested PRG the single-instance kernel may still hand the task on)
  if (alive || (mapped && (!compact || ix.is_nested || b_keep))) o.n_final[task] = ctx.n_out | (ctx.arena_n << 8) | (ctx.seed_left << 16);
  // Every lane goes to at most one queue; all of them are appended in one pass (one barrier pair, one atomic per
  // queue and block). Compact mapped tasks are queued by the PRG region they map to: workgroup b of the coverage
  // kernel serves region b % 8, workgroups go round-robin over the 8 XCDs, so every XCD's L2 sees one eighth of the
  // graph tables and of the accumulators (they do not fit one 4 MiB L2 as a whole). The probe
  // kernel's overflow queue is separate from the extend kernel's: it is served while the extend kernel still runs.
  const uint32_t region = min(__umulhi(ctx.first_pos, o.region_inv), (uint32_t)(GMX_REGIONS - 1));
  enum : uint32_t { Q_OVER = GMX_REGIONS, Q_ALIVE, Q_DEAD, Q_GENERAL, Q_N, Q_SKIPPED = Q_N, Q_COLS };  // Q_SKIPPED: a count only
  const uint32_t cat = mapped ? (compact ? region : Q_GENERAL) : over ? Q_OVER : alive ? Q_ALIVE : dead ? Q_DEAD : 0xFFu;
  __shared__ uint32_t q_cnt[GMX_BLOCK / 64][Q_COLS];
  __shared__ uint32_t q_base[Q_N];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mine = 0;
#pragma unroll
  for (uint32_t c = 0; c < Q_N; ++c) {
    const unsigned long long m = __ballot(cat == c);
    if (lane == 0) q_cnt[wave][c] = (uint32_t)__popcll(m);
    if (cat == c) mine = m;
  }
  {
    const unsigned long long m = __ballot(active && status == GMX_TASK_SKIPPED);
    if (lane == 0) q_cnt[wave][Q_SKIPPED] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < Q_N) {
    const uint32_t c = threadIdx.x;
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < GMX_BLOCK / 64; ++w) total += q_cnt[w][c];
    const uint32_t counter = c < GMX_REGIONS ? 16 + c : c == Q_OVER ? (second_phase ? 9u : 1u) : c == Q_ALIVE ? (second_phase ? GMX_CNT_ALIVE2 + alive_pass : 5u) : c == Q_DEAD ? (second_phase ? 12u : 6u) : 8u;
    q_base[c] = total ? atomicAdd(&o.counters[counter * GMX_CNT_STRIDE], total) : 0;
  } else if (threadIdx.x == Q_N) {  // read counters: every task mapped here (the regional queues + the general one) ...
    uint32_t n_map = 0;
#pragma unroll
    for (uint32_t w = 0; w < GMX_BLOCK / 64; ++w) {
      n_map += q_cnt[w][Q_GENERAL];
#pragma unroll
      for (uint32_t c = 0; c < GMX_REGIONS; ++c) n_map += q_cnt[w][c];
    }
    if (n_map) atomicAdd(&o.stats[4], (unsigned long long)n_map);
  } else if (threadIdx.x == Q_N + 1) {  // ... and the skipped reads (probe pipeline: the seed kernel counts its own)
    uint32_t n_skip = 0;
#pragma unroll
    for (uint32_t w = 0; w < GMX_BLOCK / 64; ++w) n_skip += q_cnt[w][Q_SKIPPED];
    if (n_skip) atomicAdd(&o.stats[1], (unsigned long long)n_skip);
  }
  __syncthreads();
  if (cat != 0xFFu) {
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < GMX_BLOCK / 64; ++w) before += w < wave ? q_cnt[w][cat] : 0;
    const uint32_t at = q_base[cat] + before + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    if (cat < GMX_REGIONS) {
      o.cover_recs[(size_t)cat * o.region_cap + at] = rec;
      o.cover_rec_task[(size_t)cat * o.region_cap + at] = task;
    } else {
      const uint32_t q = cat == Q_OVER ? (second_phase ? GMX_TL_OVERFLOW2 : GMX_TL_OVERFLOW)
                         : cat == Q_ALIVE ? (second_phase ? GMX_TL_ALIVE2 + alive_pass : GMX_TL_ALIVE)
                         : cat == Q_DEAD  ? (second_phase ? GMX_TL_DEAD2 : GMX_TL_DEAD)
                                          : GMX_TL_GENERAL;
      o.task_lists[(size_t)q * o.list_stride + at] = task;
    }
  }
}

#define GMX_PROBE_ITERS 10  // default iteration budget of the probe kernel (GMX_PROBE_ITERS in the environment overrides)
#define GMX_PROBE_STEPS 6  // bases extended by the probe phase; a wrong-orientation task survives them with p ~ 1e-3

// Phase 1 — every (read, orientation): seed lookup + the first GMX_PROBE_STEPS extensions. Half of the tasks
// (the orientation that does not map) die here; the survivors are parked and compacted for the main phase.
template <bool CURSOR>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_probe_kernel(GmxIndexView ix, BatchView b, SearchOut o, uint32_t probe_iters) {
  uint32_t task = blockIdx.x * GMX_BLOCK + threadIdx.x;
  bool active = task < b.n_reads * 2;
  if (task == 0) atomicAdd(&o.stats[0], (unsigned long long)b.n_reads * (b.forward_only ? 1ull : 2ull));  // all_reads_count
  uint32_t status = GMX_TASK_SKIPPED;
  bool done = false;
  FastCtx ctx;
  ctx.sp = 0;
  ctx.arena_n = 0;
  ctx.status = GMX_TASK_MAPPED;
  ctx.arena = o.arena + task;
  ctx.arena_stride = o.arena_stride;
  ctx.arena_first = 0;
  ctx.inst_states = nullptr;
  ctx.inst_count = nullptr;
  ctx.inst_cap = 0;
  ctx.out = o.finals + (size_t)task * GMX_FAST_STATES;
  ctx.n_out = 0;
  ctx.out_cap = GMX_STACK_DEPTH;  // parked entries must fit the extend kernel's stack
  ctx.parking = true;
  ctx.park_pos = 0;
  ctx.defer_first = ctx.first_deferred = false;
  ctx.first_pos = ctx.first_tvd = ctx.first_tvg = GMX_NIL;
  ctx.seed_left = ctx.seed_off = ctx.seed_pos = ctx.mark_arena = ctx.mark_out = 0;
  ReadRegs r;
  r.clear(b.packed);
  bool run = false;
  uint32_t lane_stop = 0;
  if (active) {
    task_read_regs(b, task, r);
    if (b.forward_only && r.rc) {
      status = GMX_STATUS_IGNORED;
    } else if (!read_skipped(b, task >> 1) && r.len >= ix.kmer_size && r.len > 0) {
      // reads long enough are seeded from the longer table (gmx_index.cpp): fewer steps, and most reverse-complement
      // tasks end here because their last k2-mer does not occur in the PRG
      const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
      const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
      const uint32_t from = r.len - k;
      const uint32_t stop = from > GMX_PROBE_STEPS ? from - GMX_PROBE_STEPS : 0;
      load_seed_cursor<CURSOR>(ix, (longer ? ix.seeds2 : ix.seeds)[last_kmer_code(r, k)], ctx, from);
      run = ctx.status == GMX_TASK_MAPPED;
      status = ctx.status;
      done = stop == 0;
      lane_stop = stop;
      ctx.parking = !done;
      ctx.park_pos = stop;
      if (done) ctx.out_cap = GMX_FAST_STATES;
    }
  }
  GmxLane ln;
  dfs_run_wave<0, CURSOR>(ix, ctx, r, lane_stop, run, probe_iters, ln);  // every lane of the wave takes part in the ballots
  if (run) {
    // iteration budget spent with work left: park the lane's entry and its stack as they are; seed states not yet
    // started stay in the index, the extend kernel continues the cursor
    if ((ln.have || ctx.sp || ctx.seed_left) && ctx.status == GMX_TASK_MAPPED) {
      if (done) {
        ctx.fail(GMX_TASK_OVERFLOW);  // a short read whose states are final ones: redone by the large-capacity pass
      } else {
        while (ln.have || ctx.sp) {
          if (ln.have && ln.mode != GMX_MODE_DEAD && !ctx.park(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode)) {
            ctx.fail(GMX_TASK_OVERFLOW);
            break;
          }
          ln.have = ctx.pop(ln.a, ln.b, ln.tvd, ln.tvg, ln.pos, ln.mode);
        }
        if (ctx.seed_left) o.seed_cursor[task] = ctx.seed_off;
      }
    }
    status = ctx.status;
  }
  finish_lane(ix, o, active, task, ctx, status, done, false, r.len, false, 0, b.keep_states != 0);
}

// Phase 2 — the compacted survivors: all 64 lanes of a wave carry a live search for the rest of the read.
// With a longer seed table the probe phase has nothing left to thin out: a reverse-complement task almost always
// ends at the look-up (its last k2-mer does not occur in the PRG). This light kernel does only that look-up for
// every task and queues it as alive or dead; the extend kernel then runs the alive ones from their seed states
// (SEEDED) — no probe steps, no parking, no second pass over the tasks that die here.
#define GMX_INST_MAX 64u          // a path-less seed over up to this many positions becomes that many instance lanes
#define GMX_INST_STATES 2u   // final states an instance may add (per task: its instances x this)
#define GMX_ENTRY_BIG 0x80000000u   // coverage queue entry: a large-capacity slot
#define GMX_ENTRY_INST 0xC0000000u  // ... the slot of an instance-searched task (its states and nodes are in the instance pools)
#define GMX_INST_COMPLEX 0x80000000u  // inst_sa entry: (state index << 8 | occurrence) within a multi-state seed entry
#define GMX_INST_FLAG 0x80000000u  // overflow_list entry: the task was taken apart into instances (the split search skips it)
#define GMX_SEED_THREADS 1024  // large blocks: one atomic per block and queue, and the queue counters are contended ...
#define GMX_SEED_CHUNKS 4      // ... so every thread takes four tasks (1024 apart): 512 reservations per queue and batch of 1 M
                               // reads instead of 2048 (each costs 5-10 ns of the kernel's time: 60 -> 110 us with 256-thread blocks)
__global__ void __launch_bounds__(GMX_SEED_THREADS) gmx_seed_kernel(GmxIndexView ix, BatchView b, SearchOut o) {
  constexpr uint32_t CH = GMX_SEED_CHUNKS;
  const uint32_t task0 = blockIdx.x * (GMX_SEED_THREADS * CH) + threadIdx.x;  // chunk j: task0 + j * GMX_SEED_THREADS
  // all_reads_count (quasimap.cpp:104): both orientations of every read, or the one a forward_only engine maps
  if (task0 == 0) atomicAdd(&o.stats[0], (unsigned long long)b.n_reads * (b.forward_only ? 1ull : 2ull));
  enum : uint32_t { C_ALIVE = 0, C_DEAD = 1, C_OVER = 2, C_NONE = 3 };
  uint32_t cat[CH];
  GmxSeed sds[CH];
  uint32_t n_skipped = 0;
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    const uint32_t task = task0 + j * GMX_SEED_THREADS;
    const bool active = task < b.n_reads * 2;
    bool alive = false, dead = false, over = false;
    GmxSeed sd{1, 0};
    if (active) {
      const uint32_t read = task >> 1;
      ReadRegs r;  // planes fetched on demand: one or two pairs hold the last k-mer
      r.w = b.packed + pack_off(b, read);
      r.len = read_len(b, read);
      r.rc = (task & 1) != 0;
      r.in_regs = false;
      if (b.forward_only && r.rc) {
        // not mapped, not counted
      } else if (!read_skipped(b, read) && r.len >= ix.kmer_size && r.len > 0) {
        const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
        const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
        sd = (longer ? ix.seeds2 : ix.seeds)[last_kmer_code(r, k)];
        if (sd.a != GMX_SEED_COMPLEX) {
          // a k2-mer with more occurrences than the per-lane stack has entries lies in a repeat: its interval splits at
          // the copies' own sites, the task would overflow the extend kernel after holding its wave up — straight to the
          // large-capacity pass (with the extend kernel's overflow queue)
          over = sd.a <= sd.b && sd.b != GMX_TEXT_MARK && sd.b - sd.a >= GMX_SEED_SPLIT_MAX;
          alive = sd.a <= sd.b && !over;
        } else {
          // a multi-state entry with a path-less state over many positions (the k2-mer spans a site in one copy of a
          // repeat and occurs plainly in the others; flagged at upload): the large-capacity pass takes such a state apart
          over = (sd.b & GMX_SEEDF_BIG) != 0;
          alive = !over && !(sd.b & GMX_SEEDF_EMPTY);
        }
        dead = !alive && !over;
      } else {
        ++n_skipped;
      }
    }
    cat[j] = alive ? C_ALIVE : dead ? C_DEAD : over ? C_OVER : C_NONE;
    sds[j] = sd;
  }
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {  // skipped tasks (reads with a non-ACGT symbol, or shorter than k): rare, one atomic per block that has any
    __shared__ uint32_t n_skip;
    if (threadIdx.x == 0) n_skip = 0;
    __syncthreads();
    if (n_skipped) atomicAdd(&n_skip, n_skipped);
    __syncthreads();
    if (threadIdx.x == 0 && n_skip) atomicAdd(&o.stats[1], (unsigned long long)n_skip);
  }
  // block-aggregated appends to the alive, the dead and the large-capacity queue: one reservation per queue and block
  __shared__ uint32_t cnt[CH][GMX_SEED_THREADS / 64][3];
  __shared__ uint32_t chunk_base[CH][3];  // of a chunk's entries within the block's reservation
  __shared__ uint32_t base[3];
  unsigned long long mine[CH];
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    const unsigned long long m0 = __ballot(cat[j] == C_ALIVE), m1 = __ballot(cat[j] == C_DEAD), m2 = __ballot(cat[j] == C_OVER);
    if (lane == 0) {
      cnt[j][wave][0] = (uint32_t)__popcll(m0);
      cnt[j][wave][1] = (uint32_t)__popcll(m1);
      cnt[j][wave][2] = (uint32_t)__popcll(m2);
    }
    mine[j] = cat[j] == C_ALIVE ? m0 : cat[j] == C_DEAD ? m1 : m2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t total = 0;
    for (uint32_t j = 0; j < CH; ++j) {
      chunk_base[j][threadIdx.x] = total;
      for (uint32_t w = 0; w < GMX_SEED_THREADS / 64; ++w) total += cnt[j][w][threadIdx.x];
    }
    const uint32_t counter = threadIdx.x == 0 ? 5u : threadIdx.x == 1 ? 6u : 1u;
    base[threadIdx.x] = total ? atomicAdd(&o.counters[counter * GMX_CNT_STRIDE], total) : 0;
  }
  __syncthreads();
  uint32_t over_at[CH];
#pragma unroll
  for (uint32_t j = 0; j < CH; ++j) {
    over_at[j] = 0;
    const uint32_t c = cat[j];
    if (c == C_NONE) continue;
    const uint32_t task = task0 + j * GMX_SEED_THREADS;
    uint32_t before = chunk_base[j][c];
    for (uint32_t w = 0; w < wave; ++w) before += cnt[j][w][c];
    const uint32_t at = base[c] + before + (uint32_t)__popcll(mine[j] & ((1ull << lane) - 1ull));
    if (c == C_ALIVE) {
      o.alive_list[at] = task;
      o.alive_seed[at] = sds[j];
    } else if (c == C_DEAD) {
      o.dead_list[at] = task;
    } else {
      over_at[j] = at;
    }
  }
  // Instances of the tasks sent to the large-capacity pass whose seed is one path-less interval of at most 64 positions:
  // block-wide exclusive scan of the instance counts, one atomic per block for the instance list. (A chunk without
  // such a task — every chunk of a repeat-free batch — skips this: block-uniform test.)
  for (uint32_t j = 0; j < CH; ++j) {
    {
      uint32_t any = 0;
      for (uint32_t w = 0; w < GMX_SEED_THREADS / 64; ++w) any += cnt[j][w][2];
      if (any == 0) continue;
    }
    const uint32_t task = task0 + j * GMX_SEED_THREADS;
    const bool over = cat[j] == C_OVER;
    const GmxSeed sd = sds[j];
    uint32_t width = 0;  // instances the task splits into (0: not this way)
    if (over && sd.a != GMX_SEED_COMPLEX) {
      width = sd.b - sd.a + 1u;
    } else if (over) {  // multi-state entry: one instance per occurrence of its path-less states, one per path-bearing state
      const uint32_t *w = gmx_seed_entry(ix, sd.b);
      const uint32_t ns = *w++;
      bool fits = ns <= GMX_INST_MAX;
      for (uint32_t q = 0; q < ns && fits; ++q) {
        const GmxSeedState ss = gmx_seed_state(w);
        const uint32_t n_q = (ss.nt == 0 && ss.ng == 0) ? ss.width() : 1u;
        fits = n_q <= GMX_INST_MAX && width + n_q <= GMX_INST_MAX && 2 * ss.nt + ss.ng + 2 <= GMX_FAST_ARENA;
        width += n_q;
        w += ss.words();
      }
      if (!fits) width = 0;
    }
    bool expand = over && width != 0 && width <= GMX_INST_MAX && over_at[j] < o.inst_slots;
    __shared__ uint32_t wsum[GMX_SEED_THREADS / 64];
    __shared__ uint32_t inst_base;
    uint32_t incl = expand ? width : 0u;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if ((int)lane >= d) incl += up;
    }
    __syncthreads();  // (the chunk before is done with wsum and inst_base)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (uint32_t w = 0; w < GMX_SEED_THREADS / 64; ++w) {
        const uint32_t t = wsum[w];
        wsum[w] = total;
        total += t;
      }
      uint32_t got = total ? atomicAdd(&o.counters[24 * GMX_CNT_STRIDE], total) : 0u;
      if (got + total > o.inst_cap) {  // no room: this block's tasks stay with the split search
        if (total) atomicSub(&o.counters[24 * GMX_CNT_STRIDE], total);
        got = 0xFFFFFFFFu;
      }
      inst_base = got;
    }
    __syncthreads();
    if (inst_base == 0xFFFFFFFFu) expand = false;
    if (expand) {
      const uint32_t first = inst_base + wsum[wave] + incl - width;
      if (sd.a != GMX_SEED_COMPLEX) {
        for (uint32_t i = 0; i < width; ++i) {
          o.inst_list[first + i] = (over_at[j] << 6) | i;
          o.inst_sa[first + i] = sd.a + i;
        }
      } else {
        const uint32_t *w = gmx_seed_entry(ix, sd.b);
        const uint32_t ns = *w++;
        uint32_t i = 0;
        for (uint32_t q = 0; q < ns; ++q) {
          const GmxSeedState ss = gmx_seed_state(w);
          const uint32_t n_q = (ss.nt == 0 && ss.ng == 0) ? ss.width() : 1u;
          for (uint32_t x = 0; x < n_q; ++x, ++i) {
            o.inst_list[first + i] = (over_at[j] << 6) | i;
            o.inst_sa[first + i] = GMX_INST_COMPLEX | (q << 8) | x;
          }
          w += ss.words();
        }
      }
      o.inst_remaining[over_at[j]] = width;
      o.inst_remaining_width[over_at[j]] = width;
      o.inst_first[over_at[j]] = first;
      o.slot_n_final[over_at[j]] = 0;
      o.slot_task[over_at[j]] = task;
    }
    if (over) o.overflow_list[over_at[j]] = task | (expand ? GMX_INST_FLAG : 0u);  // unflagged: the split search serves it
  }
}

// One lane per instance (above): the search of gmx_extend_kernel for ONE text-form seed state, with the path nodes in the
// instance's part of the task's slot and the final states in the slot's array. The lane that finishes a task's last
// instance queues the task for the coverage instance of the large-capacity pass — or, if one of them ran out of its
// part, for the one-lane large-capacity search, which redoes the whole task.
struct InstPools {  // (unused members kept out: the pools are SearchOut::inst_arena / inst_states)
  uint32_t reserved;
};
// (A kernel of its own: run by the idle half of gmx_extend_kernel's grid it cost that kernel 18 VGPRs — a wave per SIMD,
// 3 % of the repeat-free headline.) Block `first` of `n_blocks`.
__device__ void gmx_inst_rounds(const GmxIndexView &ix, const BatchView &b, const SearchOut &o, const InstPools &pools, uint32_t first,
                                uint32_t n_blocks) {
  (void)pools;
  const uint32_t n_inst = min(o.counters[24 * GMX_CNT_STRIDE], o.inst_cap);
  for (uint32_t base = first * GMX_BLOCK; base < n_inst; base += n_blocks * GMX_BLOCK) {
    const uint32_t idx = base + threadIdx.x;
    const bool active = idx < n_inst;
    const uint32_t entry = active ? o.inst_list[idx] : 0u;
    const uint32_t slot = entry >> 6, j = entry & 63u;
    const uint32_t task = active ? o.slot_task[slot] : 0u;
    const uint32_t first = active ? o.inst_first[slot] : 0u;  // == idx - j
    FastCtx ctx;
    ctx.sp = 0;
    ctx.arena_n = 0;
    ctx.status = GMX_TASK_MAPPED;
    ctx.arena = o.inst_arena + (size_t)first * GMX_FAST_ARENA;  // the task's base: handles are j * GMX_FAST_ARENA + n
    ctx.arena_stride = 1;
    ctx.arena_first = j * GMX_FAST_ARENA;
    ctx.out = nullptr;
    ctx.n_out = 0;
    ctx.out_cap = 0;
    ctx.parking = false;
    ctx.park_pos = 0;
    ctx.defer_first = true;
    ctx.first_deferred = false;
    ctx.first_pos = ctx.first_tvd = ctx.first_tvg = GMX_NIL;
    ctx.seed_left = ctx.seed_off = ctx.seed_pos = ctx.mark_arena = ctx.mark_out = 0;
    ctx.inst_states = o.inst_states + (size_t)first * GMX_INST_STATES;
    ctx.inst_count = o.slot_n_final + slot;
    ctx.inst_cap = active ? (o.inst_remaining_width[slot] * GMX_INST_STATES) : 0u;
    ReadRegs r;
    r.clear(b.packed);
    if (active) {
      task_read_regs(b, task, r);
      const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
      const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
      const uint32_t what = o.inst_sa[idx];
      if (!(what & GMX_INST_COMPLEX)) {  // occurrence `what` of a path-less seed interval
        ctx.push(ix.sa[what], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, r.len - k, GMX_MODE_STATE);
      } else {  // state (what >> 8) of a multi-state seed entry; occurrence (what & 255) of it when it is path-less
        const GmxSeed sd = (longer ? ix.seeds2 : ix.seeds)[last_kmer_code(r, k)];
        const uint32_t *p = gmx_seed_entry(ix, sd.b) + 1;
        for (uint32_t st = (what >> 8) & 0x7FFFFFu; st > 0; --st) p += gmx_seed_state(p).words();
        const GmxSeedState ss = gmx_seed_state(p);
        const uint32_t lo = ss.lo, hi = ss.hi, nt = ss.nt, ng = ss.ng;
        p += 4;
        if (nt == 0 && ng == 0) {  // (one position: already in text form in the device copy, gmx_seed_mark_kernel)
          ctx.push(hi == GMX_TEXT_MARK ? lo : ix.sa[lo + (what & 255u)], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, r.len - k, GMX_MODE_STATE);
        } else {
          uint32_t tvd = GMX_NIL, tvg = GMX_NIL;
          bool ok = true;
          for (uint32_t q = 0; q < nt && ok; ++q, p += 2) ok = (tvd = ctx.arena_new(p[0], (int32_t)p[1], tvd)) != GMX_NIL;
          for (uint32_t q = 0; q < ng && ok; ++q, ++p) ok = (tvg = ctx.arena_new(p[0], -1, tvg)) != GMX_NIL;
          if (!ok || !ctx.push(lo, hi, tvd, tvg, r.len - k, GMX_MODE_STATE)) ctx.fail(GMX_TASK_OVERFLOW);
        }
      }
    }
    GmxLane ln;
    dfs_run_wave<1, false>(ix, ctx, r, 0, active && ctx.status == GMX_TASK_MAPPED, 0, ln, true);
    if (!active) continue;
    bool failed = ctx.status != GMX_TASK_MAPPED;
    if (!failed && !ctx.flush_first()) failed = true;
    if (ctx.status == GMX_TASK_ERROR && atomicCAS(&o.error[0], 0u, (uint32_t)GMX_TASK_ERROR) == 0u) o.error[1] = task;
    // (no fence: nobody reads the instances' states or nodes before this kernel ends; the counters are device-scope
    // atomics. A release fence per wave here wrote back the XCD's L2 over and over and slowed every kernel beside it.)
    if (failed) atomicOr(&o.inst_remaining[slot], 0x80000000u);
    const uint32_t before = atomicSub(&o.inst_remaining[slot], 1u);
    if ((before & 0x7FFFFFFFu) != 1u) continue;  // the task's last instance goes on
    if ((before >> 31) || failed) {
      if (ctx.status != GMX_TASK_ERROR) o.overflow2_list[atomicAdd(&o.counters[9 * GMX_CNT_STRIDE], 1u)] = task;
      continue;
    }
    const uint32_t total = atomicAdd(&o.slot_n_final[slot], 0u);
    if (total > 0) {
      atomicAdd(&o.stats[4], 1ull);
      o.inst_mapped_list[atomicAdd(&o.counters[25 * GMX_CNT_STRIDE], 1u)] = GMX_ENTRY_INST | slot;
    } else {
      ReadRef rr = task_read(b, task);
      atomicAdd(&o.stats[all_kmers_present(ix.kmer_bitmap, ix.kmer_size, rr) ? 3 : 2], 1ull);
    }
  }
}

__global__ void __launch_bounds__(GMX_BLOCK) gmx_extend_inst_kernel(GmxIndexView ix, BatchView b, SearchOut o, InstPools pools) {
  gmx_inst_rounds(ix, b, o, pools, blockIdx.x, gridDim.x);
}

// Five waves per SIMD (96 VGPRs) since the text step compares 64 symbols at a time and resolves inline sites in registers
// (round 3: at six waves - 80 VGPRs - 51 values spilled and the kernel lost 6 %; A/B in profiles/round3/ab_text64_inline.txt).
// Round 2 ran six (80 VGPRs, four spills) with the 32-symbol step. Five blocks per CU leave LDS for a six-entry stack.
#ifndef GMX_EXTEND_WAVES
#define GMX_EXTEND_WAVES 5
#endif
#define GMX_EXTEND_ATTR __attribute__((amdgpu_waves_per_eu(GMX_EXTEND_WAVES)))
// MODE 0: the tasks the probe kernel parked (index without a longer seed table); 1: the tasks gmx_seed_kernel queued, from
// their seed directory entries; 2: the stragglers of the launch before (`pass` 0: of the MODE 0 / 1 launch; 1, 2: of the MODE 2
// launch with pass - 1), compacted again. `budget`: iterations of the wave loop after which a lane with work left is
// parked for the next launch; 0 = none (the last pass). A wave takes as long as its slowest lane: on nested PRGs a few
// tasks need hundreds of iterations, and every launch packs what is left into full waves again.
template <bool CURSOR, int MODE>
__global__ void __launch_bounds__(GMX_BLOCK) GMX_EXTEND_ATTR gmx_extend_kernel(GmxIndexView ix, BatchView b, SearchOut o, uint32_t fuse,
                                                                               uint32_t budget, uint32_t pass) {
  constexpr bool SEEDED = MODE == 1;
  // bit 31 of `pass`: the last pass runs under a cap — a lane with work left after `budget` iterations is not parked again
  // but handed to the large-capacity route as an overflow (nested PRGs: a few tasks with hundreds of general iterations held
  // the main stream for 0.9 ms; the 16-lane split search spreads their states over lanes, on a side stream)
  const bool capped = (pass & 0x80000000u) != 0;
  pass &= 0x7FFFFFFFu;
  uint32_t n_alive = o.counters[(MODE == 2 ? GMX_CNT_ALIVE2 + pass : 5u) * GMX_CNT_STRIDE];
  if (blockIdx.x * GMX_BLOCK >= n_alive) return;
  const long long t0 = GMX_CLK();
  uint32_t slot = blockIdx.x * GMX_BLOCK + threadIdx.x;
  bool active = slot < n_alive;
  uint32_t task = active ? (MODE == 2 ? o.task_lists + (size_t)(GMX_TL_ALIVE2 + pass) * o.list_stride : o.alive_list)[slot] : 0;
  uint32_t status = GMX_TASK_MAPPED;
  FastCtx ctx;
  ctx.sp = 0;
  ctx.arena_n = 0;
  ctx.status = GMX_TASK_MAPPED;
  ctx.arena = o.arena + task;
  ctx.arena_stride = o.arena_stride;
  ctx.arena_first = 0;
  ctx.inst_states = nullptr;
  ctx.inst_count = nullptr;
  ctx.inst_cap = 0;
  ctx.out = o.finals + (size_t)task * GMX_FAST_STATES;
  ctx.n_out = 0;
  ctx.out_cap = GMX_FAST_STATES;
  ctx.parking = false;
  ctx.park_pos = 0;
  ctx.defer_first = !ix.is_nested && !b.keep_states;  // (on a nested PRG the single-instance kernel may hand a task on to the general one)
  ctx.first_deferred = false;
  ctx.first_pos = ctx.first_tvd = ctx.first_tvg = GMX_NIL;
  ctx.seed_left = ctx.seed_off = ctx.seed_pos = ctx.mark_arena = ctx.mark_out = 0;
  ReadRegs r;
  r.clear(b.packed);
  if (active && SEEDED) {
    task_read_regs(b, task, r);
    const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
    const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
    load_seed_cursor<CURSOR>(ix, o.alive_seed[slot], ctx, r.len - k);  // the entry gmx_seed_kernel looked up
    active = ctx.status == GMX_TASK_MAPPED;
  } else if (active) {
    task_read_regs(b, task, r);
    uint32_t packed = o.n_final[task];
    uint32_t n = packed & 0xFF;
    ctx.arena_n = (packed >> 8) & 0xFF;
    ctx.mark_arena = ctx.arena_n;  // the parked entries are this kernel's pending work: nothing of theirs is released
    ctx.seed_left = CURSOR ? packed >> 16 : 0;
    if (ctx.seed_left) {
      ctx.seed_off = o.seed_cursor[task];
      ctx.seed_pos = r.len - (ix.kmer_size2 != 0 && r.len >= ix.kmer_size2 ? ix.kmer_size2 : ix.kmer_size);
    }
    if (MODE == 2) {  // a straggler: its final states so far are in finals[], its pending entries in park2
      ctx.n_out = n;
      ctx.mark_out = n;
      if (n) {
        const GmxFinalState f0 = ctx.out[0];
        if (f0.hi == GMX_TEXT_MARK) {
          ctx.first_pos = f0.lo;
          ctx.first_tvd = f0.traversed;
          ctx.first_tvg = f0.traversing;
        }
      }
      const uint32_t np = o.park2_n[task];
      const GmxParked *parked = o.park2 + (size_t)task * GMX_STACK_DEPTH;
      for (uint32_t s = 0; s < np; ++s) {
        GmxParked f = parked[s];
        ctx.push(f.a, f.b, f.tvd, f.tvg, f.pm & 0x3FFFFFFFu, f.pm >> 30);
      }
    } else {
      const GmxParked *parked = reinterpret_cast<const GmxParked *>(ctx.out);  // all read before the first emit overwrites them
      for (uint32_t s = 0; s < n; ++s) {
        GmxParked f = parked[s];
        ctx.push(f.a, f.b, f.tvd, f.tvg, f.pm & 0x3FFFFFFFu, f.pm >> 30);
      }
    }
  }
  const long long t1 = GMX_CLK();
  GmxLane ln;
  dfs_run_wave<1, CURSOR>(ix, ctx, r, 0, active, budget, ln, fuse != 0);
  bool done = true;
  if (budget && active && ctx.status == GMX_TASK_MAPPED && (ln.have || ctx.sp || ctx.seed_left)) {
    // budget spent with work left: the lane's entry and its stack as they are, for the second pass; what it has emitted
    // stays in finals[] (the first, deferred state is written now). A full stack beside a live entry has no room to be
    // restored: that task goes to the large-capacity pass.
    const bool cur = ln.have && ln.mode != GMX_MODE_DEAD;
    if (capped || (cur && ctx.sp >= GMX_STACK_DEPTH)) {
      ctx.fail(GMX_TASK_OVERFLOW);
    } else {
      ctx.flush_first();
      GmxParked *parked = o.park2 + (size_t)task * GMX_STACK_DEPTH;
      uint32_t np = 0;
      uint32_t a, bb, tvd, tvg, pos, mode;
      // (restored by pushing in this order and popping: the live entry goes last so that it is the first one popped)
      while (ctx.pop(a, bb, tvd, tvg, pos, mode)) parked[np++] = GmxParked{a, bb, tvd, tvg, pos | (mode << 30)};
      // pop order is top first: reverse so that pushing restores the same stack
      for (uint32_t i = 0; i + i + 1 < np; ++i) {
        const GmxParked t = parked[i];
        parked[i] = parked[np - 1 - i];
        parked[np - 1 - i] = t;
      }
      if (cur) parked[np++] = GmxParked{ln.a, ln.b, ln.tvd, ln.tvg, ln.pos | (ln.mode << 30)};
      o.park2_n[task] = np;
      if (ctx.seed_left) o.seed_cursor[task] = ctx.seed_off;
      done = false;
    }
  }
  status = ctx.status;
  const long long t2 = GMX_CLK();
  finish_lane(ix, o, slot < n_alive, task, ctx, status, done, true, r.len, !done, MODE == 2 ? pass + 1u : 0u, b.keep_states != 0);
  const long long t3 = GMX_CLK();
  GMX_TSTAT(1, 10, t1 - t0);
  GMX_TSTAT(1, 11, t2 - t1);
  GMX_TSTAT(1, 12, t3 - t2);
}

// Phase 3 — tasks without a final state: all_read_kmers_occur_in_index decides between the
// missing_kmer and no_extension counters (quasimap.cpp:168-186); it never affects coverage.
// Two passes: pass 0 = the probe kernel's dead tasks, run beside the extend kernel; pass 1 = the extend kernel's.
__global__ void __launch_bounds__(GMX_BLOCK) gmx_filter_kernel(GmxIndexView ix, BatchView b, SearchOut o, int pass) {
  const uint32_t n_dead = o.counters[(pass ? 12 : 6) * GMX_CNT_STRIDE];
  if (blockIdx.x * GMX_BLOCK >= n_dead) return;
  uint32_t slot = blockIdx.x * GMX_BLOCK + threadIdx.x;
  bool present = false, missing = false;
  if (slot < n_dead) {
    uint32_t task = (pass ? o.dead2_list : o.dead_list)[slot];
    ReadRef r = task_read(b, task);
    present = all_kmers_present(ix.kmer_bitmap, ix.kmer_size, r);
    missing = !present;
  }
  __shared__ uint32_t n_miss, n_noext;
  gmx_block_count(o.stats, 2, missing, &n_miss);
  gmx_block_count(o.stats, 3, present, &n_noext);
}

// The same decision where almost every k-mer occurs in the PRG (a whole-genome PRG: 12 occurrences per 14-mer, a few hundred
// of the 4^14 k-mers absent): the ABSENT k-mers as a hash table in LDS instead of the presence bitmap in memory. With the
// bitmap (32 MB at k = 14: no LDS, no early exit because nothing is missing) the filter sent 137 scattered requests per dead
// task to the L2 — 137 M per pass, twice per batch, beside the search kernels that live on the same request path.
#define GMX_ABSENT_MAX 2048u
#define GMX_ABSENT_SLOTS 4096u
__global__ void __launch_bounds__(GMX_BLOCK) gmx_filter_absent_kernel(GmxIndexView ix, BatchView b, SearchOut o, const uint32_t *absent,
                                                                      uint32_t n_absent, int pass) {
  const uint32_t n_dead = o.counters[(pass ? 12 : 6) * GMX_CNT_STRIDE];
  if (blockIdx.x * GMX_BLOCK >= n_dead) return;
  __shared__ uint32_t table[GMX_ABSENT_SLOTS];
  for (uint32_t i = threadIdx.x; i < GMX_ABSENT_SLOTS; i += GMX_BLOCK) table[i] = 0xFFFFFFFFu;  // (k-mer codes are < 4^15)
  __syncthreads();
  auto slot_of = [](uint32_t code) { return (code * 2654435761u) >> 20; };  // 12 bits
  for (uint32_t i = threadIdx.x; i < n_absent; i += GMX_BLOCK) {
    const uint32_t code = absent[i];
    uint32_t h = slot_of(code);
    while (atomicCAS(&table[h], 0xFFFFFFFFu, code) != 0xFFFFFFFFu) h = (h + 1u) & (GMX_ABSENT_SLOTS - 1u);
  }
  __syncthreads();
  const uint32_t slot = blockIdx.x * GMX_BLOCK + threadIdx.x;
  bool present = false, missing = false;
  if (slot < n_dead) {
    const uint32_t task = (pass ? o.dead2_list : o.dead_list)[slot];
    ReadRef r = task_read(b, task);
    present = true;
    if (n_absent) {
      const uint32_t k = ix.kmer_size;
      uint32_t code = kmer_code(r, 0, k);
      for (uint32_t at = 0;; ++at) {
        uint32_t h = slot_of(code), v;
        while ((v = table[h]) != 0xFFFFFFFFu) {
          if (v == code) {
            present = false;
            break;
          }
          h = (h + 1u) & (GMX_ABSENT_SLOTS - 1u);
        }
        if (!present || at + k >= r.len) break;
        code = (code >> 2) | ((r.at(at + k) - 1u) << (2u * (k - 1u)));
      }
    }
    missing = !present;
  }
  __shared__ uint32_t n_miss, n_noext;
  gmx_block_count(o.stats, 2, missing, &n_miss);
  gmx_block_count(o.stats, 3, present, &n_noext);
}

// The same with the presence bitmap staged in LDS (k <= 10: 4^k bits <= 128 KB of the CU's 160 KB). The probes
// of a wave go to 64 unrelated words: from LDS that costs a few bank-conflict cycles, from L1/L2 one tag
// look-up per lane. One 1024-thread block per CU, persistent over the dead-task queue.
#define GMX_FILTER_LDS_THREADS 1024
// all_kmers_present on the bit planes of the read, for the LDS kernel: a k-mer is looked up by its PLANAR code (the k low
// bits of its bases, base j at bit j, below the k high bits) in a bitmap indexed that way (gmx_engine::d_kmer_planar), so
// a window of 32 bases yields its 33 - k k-mers by shift and mask. The reverse complement of the read has the
// complemented planes in reverse order: bit-reverse the inverted window and shift from the other end.
__device__ bool all_kmers_present_planar(const uint32_t *bitmap, uint32_t k, const ReadRef &r) {
  const uint32_t m = (1u << k) - 1u, per_window = 33u - k, n_kmers = r.len - k + 1u;
  for (uint32_t f0 = 0; f0 < n_kmers; f0 += per_window) {
    uint32_t lo, hi;
    r.planes(f0, lo, hi);
    if (r.rc) {
      lo = __builtin_bitreverse32(~lo);
      hi = __builtin_bitreverse32(~hi);
    }
    const uint32_t cnt = min(per_window, n_kmers - f0);
    for (uint32_t j0 = 0; j0 < cnt; j0 += 8) {  // eight independent probes in flight
      uint32_t present = 1;
#pragma unroll
      for (uint32_t d = 0; d < 8; ++d) {
        const uint32_t j = min(j0 + d, cnt - 1u);
        const uint32_t sh = r.rc ? 32u - k - j : j;
        const uint32_t code = (((hi >> sh) & m) << k) | ((lo >> sh) & m);
        present &= bitmap[code >> 5] >> (code & 31u);
      }
      if (!(present & 1u)) return false;
    }
  }
  return true;
}

__global__ void __launch_bounds__(GMX_FILTER_LDS_THREADS) gmx_filter_lds_kernel(GmxIndexView ix, BatchView b, SearchOut o,
                                                                                 const uint32_t *planar_bitmap,
                                                                                 uint32_t n_words, int pass) {
  const uint32_t n_dead = o.counters[(pass ? 12 : 6) * GMX_CNT_STRIDE];
  if (blockIdx.x * GMX_FILTER_LDS_THREADS >= n_dead) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(planar_bitmap);
  uint4 *dst = reinterpret_cast<uint4 *>(gmx_lds);
  for (uint32_t i = threadIdx.x; i < n_words / 4; i += GMX_FILTER_LDS_THREADS) dst[i] = src[i];
  __syncthreads();
  uint32_t c_miss = 0, c_noext = 0;
  for (uint32_t slot = blockIdx.x * GMX_FILTER_LDS_THREADS + threadIdx.x; slot < n_dead;
       slot += gridDim.x * GMX_FILTER_LDS_THREADS) {
    uint32_t task = (pass ? o.dead2_list : o.dead_list)[slot];
    ReadRef r = task_read(b, task);
    if (all_kmers_present_planar(gmx_lds, ix.kmer_size, r))
      ++c_noext;
    else
      ++c_miss;
  }
  // one atomic per counter and block
  for (int off = 32; off > 0; off >>= 1) {
    c_miss += __shfl_down(c_miss, off);
    c_noext += __shfl_down(c_noext, off);
  }
  __shared__ uint32_t tot[2];
  if (threadIdx.x < 2) tot[threadIdx.x] = 0;
  __syncthreads();  // (also: every probe of the bitmap in LDS is done)
  if ((threadIdx.x & 63) == 0) {
    if (c_miss) atomicAdd(&tot[0], c_miss);
    if (c_noext) atomicAdd(&tot[1], c_noext);
  }
  __syncthreads();
  if (threadIdx.x < 2 && tot[threadIdx.x]) atomicAdd(&o.stats[2 + threadIdx.x], (unsigned long long)tot[threadIdx.x]);
}

struct BigOut {
  GmxFinalState *states;   // slot x max_states (final states)
  uint32_t *stack;         // slot x max_states x GMX_STACK_WORDS (pending entries)
  GmxPathNode *arena;      // slot x max_path_nodes
  uint32_t *n_final;       // per slot
  uint32_t *task_of_slot;  // per slot
  uint32_t max_states, max_path_nodes, max_slots;
};

// Large-capacity pass: one lane per task that overflowed the LDS stack / parked-state / arena limits, whole read
// from the seed, same DFS loop with global-memory pools. Persistent over the device-side overflow list.
__global__ void __launch_bounds__(64) gmx_search_big_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, int second) {
  // instance 0 serves the probe kernel's overflow queue (index without a longer seed table); instance 1 what the
  // 16-lane split search could not finish within a group's parts of a slot (slots after both of its instances')
  // (second == 1, A/B runs without the second split search: the extend kernel's queue itself)
  const uint32_t n_over = o.counters[(second == 2 ? 29 : second ? 9 : 1) * GMX_CNT_STRIDE];
  const uint32_t slot_base = second == 2 ? o.counters[1 * GMX_CNT_STRIDE] + o.counters[9 * GMX_CNT_STRIDE] : second ? o.counters[1 * GMX_CNT_STRIDE] : 0;
  const uint32_t *queue = second == 2 ? o.overflow3_list : second ? o.overflow2_list : o.overflow_list;
  uint32_t rounds = (n_over + gridDim.x * 64 - 1) / (gridDim.x * 64);
  for (uint32_t rd = 0; rd < rounds; ++rd) {
    // interleaved: a short queue spreads over all waves (few active lanes each) instead of filling the first ones
    const uint32_t qi = rd * gridDim.x * 64 + threadIdx.x * gridDim.x + blockIdx.x;
    bool active = qi < n_over;
    uint32_t task = active ? queue[qi] : 0;
    const uint32_t slot = slot_base + qi;
    if (active && slot >= g.max_slots) {  // no slot left: the last tier takes the task
      o.huge_list[atomicAdd(&o.counters[11 * GMX_CNT_STRIDE], 1u)] = task;
      active = false;
    }
    BigCtx ctx;
    ctx.sp = 0;
    ctx.cap = g.max_states;
    ctx.stack = g.stack + (size_t)(active ? slot : 0) * g.max_states * GMX_STACK_WORDS;
    ctx.arena = g.arena + (size_t)(active ? slot : 0) * g.max_path_nodes;
    ctx.arena_n = 0;
    ctx.arena_cap = g.max_path_nodes;
    ctx.status = GMX_TASK_MAPPED;
    ctx.out = g.states + (size_t)(active ? slot : 0) * g.max_states;
    ctx.n_out = 0;
    ctx.out_cap = g.max_states;
    ReadRef r;
    r.w = b.packed;
    r.len = 0;
    r.rc = false;
    r.cur_idx = 0xFFFFFFFFu;
    r.cur = make_uint2(0, 0);
    bool run = false;
    if (active) {
      r = task_read(b, task);
      // seeded like the fast pass (the longer table when there is one). A path-less state over several suffix-array
      // positions — a read inside a repeat — is taken apart into its positions in text form: the same set of
      // (position, path) results (a marker hit concerns one position, and path-less final states are recorded position
      // by position, encapsulated_search.cpp:30-107), but 32 bases per step and state instead of one
      const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
      const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
      const uint32_t from = r.len - k;
      load_seed(ix, longer ? ix.seeds2 : ix.seeds, kmer_code(r, from, k), ctx,
                [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
                  if (tvd == GMX_NIL && tvg == GMX_NIL && from > 0 && hi > lo && hi - lo < 64u) {
                    bool ok = true;
                    for (uint32_t i = lo; i <= hi && ok; ++i) ok = ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
                    return ok;
                  }
                  return ctx.push(lo, hi, tvd, tvg, from, GMX_MODE_STATE);
                });
      run = ctx.status == GMX_TASK_MAPPED;
    }
    GmxLane ln;
    dfs_run_wave<2, false>(ix, ctx, r, 0, run, 0, ln);
    if (!active) continue;
    uint32_t status = ctx.status;
    uint32_t nf = 0;
    if (status == GMX_TASK_MAPPED) {
      nf = ctx.n_out;
      if (nf == 0) status = all_kmers_present(ix.kmer_bitmap, ix.kmer_size, r) ? GMX_TASK_UNMAPPED : GMX_STATUS_MISSING_KMER;
    } else if (status == GMX_TASK_OVERFLOW) {  // these pools are too small for it: the last tier (heap-backed) takes it
      o.huge_list[atomicAdd(&o.counters[11 * GMX_CNT_STRIDE], 1u)] = task;
    } else if (atomicCAS(&o.error[0], 0u, status) == 0u) {
      o.error[1] = task;
    }
    if (status == GMX_TASK_MAPPED || status == GMX_TASK_UNMAPPED || status == GMX_STATUS_MISSING_KMER)
      atomicAdd(&o.stats[status == GMX_TASK_MAPPED ? 4 : status == GMX_TASK_UNMAPPED ? 3 : 2], 1ull);  // few tasks: one atomic each
    o.n_final[task] = nf;
    g.n_final[slot] = nf;
    g.task_of_slot[slot] = task;
    if (status == GMX_TASK_MAPPED && nf > 0) {
      uint32_t at = atomicAdd(&o.counters[7 * GMX_CNT_STRIDE], 1u);
      o.big_mapped_list[at] = 0x80000000u | slot;
    }
  }
}

// The seed kernel's tasks (reads in repeats: a seed over many suffix-array positions) with GMX_SPLIT lanes per task: the
// mapping instances are independent text-form states, so lane `sub` of a task's group takes every GMX_SPLIT-th of
// them — a tenth of the dependent iterations one lane would run. Each lane has its own part of the slot's pools
// (pending entries, path nodes: handles stay slot-wide indices; final states in the upper half of the slot's array),
// and the group then moves its final states together to the front of the array, where the coverage instance expects
// them. A task one of whose lanes runs out of its part is handed to the second instance of gmx_search_big_kernel,
// which runs it in one lane with the whole slot.
#define GMX_SPLIT 16u
__global__ void __launch_bounds__(64) gmx_search_split_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, int second) {
  // instance 0: what gmx_seed_kernel sent here (reads in repeats); instance 1: the extend kernel's overflow queue and the
  // tasks whose instance lanes ran out of their pools (slots after instance 0's)
  const uint32_t n_over = o.counters[(second ? 9 : 1) * GMX_CNT_STRIDE];
  const uint32_t slot_base = second ? o.counters[1 * GMX_CNT_STRIDE] : 0u;
  const uint32_t *queue = second ? o.overflow2_list : o.overflow_list;
  const uint32_t groups = 64 / GMX_SPLIT, group = threadIdx.x / GMX_SPLIT, sub = threadIdx.x % GMX_SPLIT;
  const uint32_t per_round = gridDim.x * groups;
  const uint32_t part_states = g.max_states / (2 * GMX_SPLIT), part_nodes = g.max_path_nodes / GMX_SPLIT,
                 part_stack = g.max_states / GMX_SPLIT;
  for (uint32_t base = 0; base < n_over; base += per_round) {
    const uint32_t qi = base + group * gridDim.x + blockIdx.x;  // interleaved over the blocks
    bool active = qi < n_over;
    uint32_t task = active ? queue[qi] : 0;
    if (!second && (task & GMX_INST_FLAG)) {  // searched by instance lanes (gmx_extend_inst_kernel)
      active = false;
      task = 0;
    }
    const uint32_t slot = slot_base + qi;
    if (active && slot >= g.max_slots) {
      if (sub == 0) o.huge_list[atomicAdd(&o.counters[11 * GMX_CNT_STRIDE], 1u)] = task;
      active = false;
    }
    const size_t s0 = active ? slot : 0;
    BigCtx ctx;
    ctx.sp = 0;
    ctx.cap = part_stack;
    ctx.stack = g.stack + (s0 * g.max_states + (size_t)sub * part_stack) * GMX_STACK_WORDS;
    ctx.arena = g.arena + s0 * g.max_path_nodes;
    ctx.arena_n = sub * part_nodes;
    ctx.arena_cap = (sub + 1) * part_nodes;
    ctx.status = GMX_TASK_MAPPED;
    GmxFinalState *const slot_states = g.states + s0 * g.max_states;
    ctx.out = slot_states + g.max_states / 2 + sub * part_states;
    ctx.n_out = 0;
    ctx.out_cap = part_states;
    ReadRef r;
    r.w = b.packed;
    r.len = 0;
    r.rc = false;
    r.cur_idx = 0xFFFFFFFFu;
    r.cur = make_uint2(0, 0);
    bool run = false;
    if (active) {
      r = task_read(b, task);
      const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
      const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
      const uint32_t from = r.len - k;
      const GmxSeed sd = (longer ? ix.seeds2 : ix.seeds)[kmer_code(r, from, k)];
      bool ok = true;
      uint32_t turn = 0;  // states and positions are dealt out to the lanes of the group in turn
      auto mine = [&]() { return (turn++ % GMX_SPLIT) == sub; };
      auto state = [&](uint32_t lo, uint32_t hi, const uint32_t *paths, uint32_t nt, uint32_t ng) {
        if (nt == 0 && ng == 0 && from > 0 && hi > lo && hi - lo < 4096u) {
          for (uint32_t i = lo; i <= hi && ok; ++i)
            if (mine()) ok = ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
          return;
        }
        if (!mine()) return;
        uint32_t tvd = GMX_NIL, tvg = GMX_NIL;
        for (uint32_t j = 0; j < nt && ok; ++j) {
          tvd = ctx.arena_new(paths[2 * j], (int32_t)paths[2 * j + 1], tvd);
          ok = tvd != GMX_NIL;
        }
        for (uint32_t j = 0; j < ng && ok; ++j) {
          tvg = ctx.arena_new(paths[2 * nt + j], -1, tvg);
          ok = tvg != GMX_NIL;
        }
        ok = ok && ctx.push(lo, hi, tvd, tvg, from, GMX_MODE_STATE);
      };
      if (sd.a != GMX_SEED_COMPLEX) {
        if (sd.a <= sd.b) state(sd.a, sd.b, nullptr, 0, 0);
      } else {
        const uint32_t *w = gmx_seed_entry(ix, sd.b);
        const uint32_t ns = *w++;
        for (uint32_t i = 0; i < ns && ok; ++i) {
          const GmxSeedState ss = gmx_seed_state(w);
          state(ss.lo, ss.hi, w + 4, ss.nt, ss.ng);
          w += ss.words();
        }
      }
      if (!ok) ctx.fail(GMX_TASK_OVERFLOW);
      run = ctx.status == GMX_TASK_MAPPED;
    }
    GmxLane ln;
    dfs_run_wave<2, false>(ix, ctx, r, 0, run, 0, ln);
    // the group's verdict and the places of its final states (shuffles within the GMX_SPLIT lanes of the group)
    const unsigned long long bad = __ballot(active && ctx.status != GMX_TASK_MAPPED);
    const bool group_bad = ((bad >> (group * GMX_SPLIT)) & ((1ull << GMX_SPLIT) - 1ull)) != 0;
    uint32_t before = 0, total = 0;
    for (uint32_t i = 0; i < GMX_SPLIT; ++i) {
      const uint32_t n_i = __shfl(ctx.n_out, (int)(group * GMX_SPLIT + i));
      before += i < sub ? n_i : 0;
      total += n_i;
    }
    if (!active) continue;
    if (group_bad) {  // one lane's part did not suffice: the whole task again, in one lane with the whole slot
      if (sub == 0) {
        if (o.split_twice)
          o.overflow3_list[atomicAdd(&o.counters[29 * GMX_CNT_STRIDE], 1u)] = task;
        else
          o.overflow2_list[atomicAdd(&o.counters[9 * GMX_CNT_STRIDE], 1u)] = task;
        g.n_final[slot] = 0;
        g.task_of_slot[slot] = task;
      }
      continue;
    }
    for (uint32_t f = 0; f < ctx.n_out; ++f) slot_states[before + f] = ctx.out[f];  // the front half: disjoint from every part
    if (sub != 0) continue;
    uint32_t status = GMX_TASK_MAPPED;
    if (total == 0) status = all_kmers_present(ix.kmer_bitmap, ix.kmer_size, r) ? GMX_TASK_UNMAPPED : GMX_STATUS_MISSING_KMER;
    atomicAdd(&o.stats[status == GMX_TASK_MAPPED ? 4 : status == GMX_TASK_UNMAPPED ? 3 : 2], 1ull);
    o.n_final[task] = total;
    g.n_final[slot] = total;
    g.task_of_slot[slot] = task;
    if (total > 0) o.big_mapped_list[atomicAdd(&o.counters[7 * GMX_CNT_STRIDE], 1u)] = 0x80000000u | slot;
  }
}

// ---------------------------------------------------------------------------
// coverage kernel
// ---------------------------------------------------------------------------
struct CoverAcc {
  uint32_t *acc;        // the accumulator block (gmx_types.h: gmx_slot_*)
  uint32_t *log;        // grouped log words
  uint32_t *log_cursor; // [0] = words used
  uint32_t log_cap;
  uint32_t *scratch_big;
  uint32_t n_lanes_big;
  int rng_mode;
  uint32_t log_sites;   // the index has sites with more than 8 alleles (users of the log)
  uint32_t *heap;       // the last tier's memory (gmx_tail_stage)
  uint64_t heap_words;
  const uint32_t *status;      // per task, for the read counters tallied by the batch's last launch
  uint32_t n_tasks;
  unsigned long long *stats;   // QuasimapReadsStats counters
};

// The grouped log (sites without dense group counters): a task reserves ALL the words it will append with one atomic add,
// before it records anything (gmx_cover.h); a task that does not fit fails whole (GMX_TASK_LOGFULL), gives its words back
// and is redone after the host has drained the log (log_settle). GMX_LOG_PAD words (a reservation abandoned on an error)
// are skipped by every reader.
#define GMX_LOG_PAD 0xFFFFFFFFu
#ifdef GMX_LOOP_STATS
// per coverage instance (LIST): [0..7] wall time (10 ns units) per phase summed over tasks, [8..15] its maximum
__device__ unsigned long long gmx_cover_stats[6 * 16];
__device__ unsigned long long gmx_coop_stats[6 * 8];  // cooperative instances: wave-level wall time of the four phases, [7] rounds
extern "C" int gmx_debug_coop_stats(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_coop_stats), sizeof(gmx_coop_stats)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[6 * 8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_coop_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
__device__ unsigned long long gmx_cover_why[8 * 4];  // per coverage instance (6, 7: cooperative item / class scratch): tasks that exceeded loci / key sites / hull / items
extern "C" int gmx_debug_cover_why(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_cover_why), sizeof(gmx_cover_why)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8 * 4] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_cover_why), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int gmx_debug_cover_stats(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_cover_stats), sizeof(gmx_cover_stats)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[6 * 16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gmx_cover_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
struct CoverLogPart {
#ifdef GMX_LOOP_STATS
  long long prof_t = 0;
  int prof_list = 0;
  __device__ void why(int k) { atomicAdd(&gmx_cover_why[(prof_list & 7) * 4 + (k & 3)], 1ull); }
  __device__ void prof(int k) {
    const long long t = wall_clock64();
    atomicAdd(&gmx_cover_stats[prof_list * 16 + k], (unsigned long long)(t - prof_t));
    atomicMax(&gmx_cover_stats[prof_list * 16 + 8 + k], (unsigned long long)(t - prof_t));
    prof_t = t;
  }
#endif
  uint32_t *acc, *log, *log_cursor;
  uint32_t log_cap;
  uint32_t status;
  uint32_t log_at;
  uint32_t log_end = 0;  // end of this task's reservation
  uint32_t log_sites;  // the index has sites that use the log
  __device__ __forceinline__ bool has_log_sites() const { return log_sites != 0; }
  __device__ __forceinline__ bool log_reserve(uint32_t words) {
    // compare-and-swap: the cursor moves only for a reservation that fits, so it never exceeds the capacity, the words
    // below it are exactly the successful reservations back to back, and a failing task leaves no trace (an add that is
    // taken back later opens a window in which another task's words end up beyond the cursor).
    uint32_t cur = __hip_atomic_load(log_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      if (cur > log_cap || words > log_cap - cur) {
        log_at = log_end = 0;
        status = GMX_TASK_LOGFULL;
        return false;
      }
      const uint32_t seen = atomicCAS(log_cursor, cur, cur + words);
      if (seen == cur) break;
      cur = seen;
    }
    log_at = cur;
    log_end = log_at + words;
    return true;
  }
  // a task that failed AFTER its reservation (a condition on which the reference throws) leaves no unwritten words behind
  __device__ __forceinline__ void log_abandon() {
    if (status != GMX_TASK_MAPPED && status != GMX_TASK_LOGFULL)
      for (uint32_t i = log_at; i < log_end && i < log_cap; ++i) log[i] = GMX_LOG_PAD;
    log_at = log_end = 0;
  }
  __device__ __forceinline__ bool log_grouped_begin(uint32_t site_index, uint32_t n_ids) {
    log[log_at++] = site_index;
    log[log_at++] = n_ids;
    return true;
  }
  __device__ __forceinline__ void log_grouped_id(int32_t a) { log[log_at++] = (uint32_t)a; }
  __device__ __forceinline__ void log_grouped_end() {}
  __device__ __forceinline__ uint32_t single_loci() const { return GMX_SINGLE_LOCI; }
  __device__ __forceinline__ void add_allele_sum(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_per_base(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_hit(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_grouped_dense(uint32_t slot) { atomicAdd(&acc[slot], 1u); }
  __device__ __forceinline__ void add_allele_and_group(uint32_t slot) {  // slot is even: both counters in one 64-bit add
    atomicAdd(reinterpret_cast<unsigned long long *>(acc + slot), 0x100000001ull);
  }
  __device__ __forceinline__ void fail(uint32_t s) {
    if (status == GMX_TASK_MAPPED || s == GMX_TASK_ERROR) status = s;
  }
};

template <uint32_t I_, uint32_t B_, uint32_t LOC_, uint32_t H_, uint32_t P_ = GMX_PATH_CACHE>
struct CoverEnvT : CoverLogPart {
  static constexpr uint32_t I_MAX = I_, B_MAX = B_, LOC_MAX = LOC_, H_MAX = H_, P_MAX = P_;
  __device__ __forceinline__ static constexpr uint32_t i_max() { return I_; }
  __device__ __forceinline__ static constexpr uint32_t b_max() { return B_; }
  __device__ __forceinline__ static constexpr uint32_t loc_max() { return LOC_; }
  __device__ __forceinline__ static constexpr uint32_t h_max() { return H_; }
  uint32_t *scratch;  // already offset by the lane
  uint32_t stride;
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[(size_t)w * stride]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[(size_t)w * stride] = v; }
};

// The last tier: capacities decided per task from what the heap slice holds (gmx_tail_stage).
struct CoverEnvDyn : CoverLogPart {
  uint32_t cap_i, cap_b, cap_loc, cap_h;
  __device__ __forceinline__ uint32_t i_max() const { return cap_i; }
  __device__ __forceinline__ uint32_t b_max() const { return cap_b; }
  __device__ __forceinline__ uint32_t loc_max() const { return cap_loc; }
  __device__ __forceinline__ uint32_t h_max() const { return cap_h; }
  uint32_t *scratch;
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[w]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[w] = v; }
};

typedef CoverEnvT<4, 12, 24, 24> CoverEnvLds;         // first tier of the general pass: per-lane scratch in the block's LDS
typedef CoverEnvT<12, 12, 48, 48> CoverEnvMid;         // the large-capacity pass's tasks (a read in a 10-copy repeat has ~11 items)
typedef CoverEnvT<24, 16, 64, 64> CoverEnv;           // per-lane scratch of the regular pass
typedef CoverEnvT<1024, 32, 1024, 1024> CoverEnvBig;  // reads with many mapping instances (repeats)

// ---------------------------------------------------------------------------
// The last tier. Every pool above has a fixed size per task; a task that exceeds one of them — a read with thousands of
// mapping instances, or through more nested sites than the large-capacity pools hold — ends up here, where the only
// limit is the engine's heap (gmx_engine_opts::huge_heap_bytes): the reference has no limit either
// (encapsulated_search.cpp:30-107 and coverage_common.cpp:85-146 simply iterate). Work items are
//   * tasks of huge_list: searched again from the seed with pools carved from a heap slice, then recorded with a scratch
//     sized for what the search produced (gmx_cover_task over CoverEnvDyn);
//   * entries of cover_huge_list: their final states are where the search left them, only the scratch was too small.
// One wave runs the stage (the last block of the batch's last coverage launch): first every lane takes work items with
// one 64th of the heap each, then lane 0 alone redoes, with the whole heap, what did not fit. Nothing is recorded for a
// task before all of its capacity checks have passed, so redoing is safe. A task that does not fit the whole heap is
// reported (GMX_ECAP: raise huge_heap_bytes). Common batches have no work item and pay one counter read.
// ---------------------------------------------------------------------------
// What a coverage queue entry stands for: a task finished by the fast pass (its id), a large-capacity slot, or the slot
// of an instance-searched task.
struct GmxTaskStates {
  uint32_t task, nf;
  const GmxFinalState *finals;
  const GmxPathNode *arena;
};
__device__ __forceinline__ GmxTaskStates gmx_entry_states(uint32_t entry, const SearchOut &o, const BigOut &g) {
  GmxTaskStates t;
  if ((entry & GMX_ENTRY_INST) == GMX_ENTRY_INST) {
    const uint32_t slot = entry & 0x3fffffffu, first = o.inst_first[slot];
    t.task = o.slot_task[slot];
    t.nf = o.slot_n_final[slot];
    t.finals = o.inst_states + (size_t)first * GMX_INST_STATES;
    t.arena = o.inst_arena + (size_t)first * GMX_FAST_ARENA;
  } else if (entry & GMX_ENTRY_BIG) {
    const uint32_t slot = entry & 0x7fffffffu;
    t.task = g.task_of_slot[slot];
    t.nf = g.n_final[slot];
    t.finals = g.states + (size_t)slot * g.max_states;
    t.arena = g.arena + (size_t)slot * g.max_path_nodes;
  } else {
    t.task = entry;
    t.nf = o.n_final[entry] & 0xFF;
    t.finals = o.finals + (size_t)entry * GMX_FAST_STATES;
    t.arena = o.arena + entry;  // handles are offsets from the task's base (FastCtx::alloc_node)
  }
  return t;
}

__device__ uint32_t gmx_count_items(const GmxIndexView &ix, const GmxFinalState *finals, uint32_t nf) {
  uint32_t n = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    const GmxFinalState st = finals[f];
    if (st.traversed != GMX_NIL || st.traversing != GMX_NIL) {
      ++n;
      continue;
    }
    for (uint32_t i = st.lo;; ++i) {
      n += ix.nodes[ix.pos_node[gmx_occ_pos(ix, st.hi, i)]].site != 0;
      if (gmx_text_form(st.hi) || i == st.hi) break;
    }
  }
  return n;
}

// returns the status of the work item: MAPPED (done), OVERFLOW (the slice was too small, nothing recorded), or an error
__device__ uint32_t gmx_tail_item(const GmxIndexView &ix, const BatchView &b, const SearchOut &o, const BigOut &g, const CoverAcc &acc,
                                  bool active, uint32_t item, uint32_t n_search, uint32_t *slice, uint64_t slice_words, bool whole_heap,
                                  uint32_t &task_out) {
  const bool is_search = active && item < n_search;
  uint32_t task = 0, nf = 0;
  const GmxFinalState *finals = nullptr;
  const GmxPathNode *arena = nullptr;
  uint32_t *scratch = slice;
  uint64_t scratch_words = slice_words;
  uint32_t status = GMX_TASK_MAPPED;
  // --- search (all lanes of the wave take part in the loop's ballots) ---
  BigCtx ctx;
  const uint64_t S = std::min<uint64_t>(slice_words / 30, 0x3FFFFFFFull);  // states; half of the slice is left for the scratch
  ctx.sp = 0;
  ctx.cap = (uint32_t)S;
  ctx.out = reinterpret_cast<GmxFinalState *>(slice);
  ctx.stack = slice + 4 * S;
  ctx.arena = reinterpret_cast<GmxPathNode *>(slice + 9 * S);
  ctx.arena_n = 0;
  ctx.arena_cap = (uint32_t)(2 * S);
  ctx.status = GMX_TASK_MAPPED;
  ctx.n_out = 0;
  ctx.out_cap = (uint32_t)S;
  ReadRef r;
  r.w = b.packed;
  r.len = 0;
  r.rc = false;
  r.cur_idx = 0xFFFFFFFFu;
  r.cur = make_uint2(0, 0);
  bool run = false;
  if (is_search) {
    task = o.huge_list[item];
    r = task_read(b, task);
    const bool longer = ix.kmer_size2 != 0 && r.len >= ix.kmer_size2;
    const uint32_t k = longer ? ix.kmer_size2 : ix.kmer_size;
    const uint32_t from = r.len - k;
    load_seed(ix, longer ? ix.seeds2 : ix.seeds, kmer_code(r, from, k), ctx,
              [&](uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
                if (tvd == GMX_NIL && tvg == GMX_NIL && from > 0 && hi > lo && hi != GMX_TEXT_MARK) {  // position by position in text form (gmx_search_big_kernel)
                  bool ok = true;
                  for (uint32_t i = lo; ok; ++i) {
                    ok = ctx.push(ix.sa[i], GMX_TEXT_MARK, GMX_NIL, GMX_NIL, from, GMX_MODE_STATE);
                    if (i == hi) break;
                  }
                  return ok;
                }
                return ctx.push(lo, hi, tvd, tvg, from, GMX_MODE_STATE);
              });
    run = ctx.status == GMX_TASK_MAPPED;
  }
  GmxLane ln;
  dfs_run_wave<2, false>(ix, ctx, r, 0, run, 0, ln);
  if (!active) return GMX_TASK_MAPPED;
  uint32_t len;
  if (is_search) {
    task_out = task;
    status = ctx.status;
    if (status != GMX_TASK_MAPPED) return status;
    nf = ctx.n_out;
    if (nf == 0) {
      // its read counter
      atomicAdd(&acc.stats[all_kmers_present(ix.kmer_bitmap, ix.kmer_size, r) ? 3 : 2], 1ull);
      o.n_final[task] = 0;
      return GMX_TASK_MAPPED;
    }
    finals = ctx.out;
    arena = ctx.arena;
    scratch = slice + 15 * S;
    scratch_words = slice_words - 15 * S;
    len = r.len;
  } else {
    const GmxTaskStates ts = gmx_entry_states(o.cover_huge_list[item - n_search], o, g);
    task = ts.task;
    nf = ts.nf;
    finals = ts.finals;
    arena = ts.arena;
    task_out = task;
    const uint32_t read = task >> 1;
    len = read_len(b, read);
  }
  // --- coverage with a scratch sized for this task ---
  CoverEnvDyn env;
  const uint64_t n_items = std::max<uint32_t>(gmx_count_items(ix, finals, nf), 1u);
  uint64_t cap_b = std::min<uint64_t>(std::max<uint64_t>(len + 8u, 32u), 4096u);
  if (whole_heap) cap_b = std::max<uint64_t>(cap_b, std::min<uint64_t>(65536u, scratch_words / (4 * n_items)));
  const uint64_t fixed = n_items * (GmxScratch<CoverEnvDyn>::ITEM_W + 2 + cap_b) + 2 * GMX_PATH_CACHE + 1;
  if (fixed + 5 * 64 > scratch_words) return GMX_TASK_OVERFLOW;
  const uint64_t rest = std::min<uint64_t>((scratch_words - fixed) / 5, 0x0FFFFFFFull);
  env.cap_i = (uint32_t)n_items;
  env.cap_b = (uint32_t)cap_b;
  env.cap_loc = env.cap_h = (uint32_t)rest;
  env.scratch = scratch;
  env.arena = arena;
  env.acc = acc.acc;
  env.log = acc.log;
  env.log_cursor = acc.log_cursor;
  env.log_cap = acc.log_cap;
  env.log_sites = acc.log_sites;
  env.status = GMX_TASK_MAPPED;
  env.log_at = 0;
  gmx_cover_task(ix, env, finals, nf, len, b.seeds[task >> 1], acc.rng_mode);
  env.log_abandon();
  if (env.status == GMX_TASK_MAPPED && is_search) {
    atomicAdd(&acc.stats[4], 1ull);  // exact_mapped
    o.n_final[task] = nf;
  }
  return env.status;
}

// a work item of the last tier that found the grouped log full: redone after the host has drained the log
__device__ __forceinline__ void gmx_tail_log_retry(const SearchOut &o, uint32_t item, uint32_t n_search) {
  if (item < n_search)
    o.log_retry_huge[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE], 1u)] = o.huge_list[item];
  else
    o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = o.cover_huge_list[item - n_search];
}

__device__ void gmx_tail_stage(const GmxIndexView &ix, const BatchView &b, const SearchOut &o, const BigOut &g, const CoverAcc &acc) {
  const uint32_t n_search = o.counters[11 * GMX_CNT_STRIDE], n_cover = o.counters[15 * GMX_CNT_STRIDE];
  const uint32_t total = n_search + n_cover;
  if (total == 0) return;
  __shared__ uint32_t n_retry;
  if (threadIdx.x == 0) n_retry = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t slice_words = acc.heap_words / 64;
  for (uint32_t base = 0; base < total; base += 64) {  // every lane: one work item, one 64th of the heap
    const uint32_t item = base + lane;
    uint32_t task = 0;
    const uint32_t st = gmx_tail_item(ix, b, o, g, acc, item < total, item, n_search, acc.heap + (size_t)lane * slice_words, slice_words,
                                      false, task);
    if (item < total && st == GMX_TASK_OVERFLOW)
      o.huge_retry[atomicAdd(&n_retry, 1u)] = item;
    else if (item < total && st == GMX_TASK_LOGFULL)
      gmx_tail_log_retry(o, item, n_search);
    else if (item < total && st != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, st) == 0u)
      o.error[1] = task;
  }
  __syncthreads();
  __threadfence();
  const uint32_t retries = n_retry;
  for (uint32_t i = 0; i < retries; ++i) {  // lane 0 alone, the whole heap
    uint32_t task = 0;
    const uint32_t st = gmx_tail_item(ix, b, o, g, acc, lane == 0, o.huge_retry[i], n_search, acc.heap, acc.heap_words, true, task);
    if (lane == 0 && st == GMX_TASK_LOGFULL)
      gmx_tail_log_retry(o, o.huge_retry[i], n_search);
    else if (lane == 0 && st != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, st) == 0u)
      o.error[1] = task;
  }
}

// Four instances over four device-side queues (LIST):
//   3  tasks finished by the probe / extend kernels that gmx_cover_single_kernel passed on; scratch sized for the
//      few instances and loci most such tasks have
//   0  those whose selection exceeded it, regular scratch
//   2  tasks finished by the large-capacity search (runs on the engine's side stream), regular scratch
//   1  entries of 0 and 2 whose selection exceeded the regular scratch, redone with the large one after both
// An entry that exceeds a scratch has recorded nothing yet.  The per-lane scratch of 3, 0 and 2 lives in the block's
// LDS (a dependent chain of scratch accesses per task: LDS latency, not L2 latency, sets the pace); a block runs
// gmx_cover_lds_lanes<Env>() lanes, as many as copies of the scratch fit 64 KB.  Instance 1 uses global memory.
template <class Env>
constexpr uint32_t gmx_cover_lds_lanes() {
  return GmxScratchFixed<Env>::total * 64 * sizeof(uint32_t) <= 64 * 1024   ? 64u
         : GmxScratchFixed<Env>::total * 32 * sizeof(uint32_t) <= 64 * 1024 ? 32u
                                                                       : 16u;
}
template <class Env, int LIST>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g,
                                                              CoverAcc acc, uint32_t lanes_rt, uint32_t after_coop) {
  constexpr bool BIG = LIST == 1;
  constexpr bool LDS = LIST != 1;
  const uint32_t LANES = LDS ? lanes_rt : 64u;  // active lanes of a block (blockDim.x is 64)
  // LIST 4 and 2 share the large-capacity pass's queue: 4 takes what its first instance mapped and leaves the length
  // in counter [10], 2 starts there
  // (instances 3, 5 and 2 after the cooperative kernel: only what that one left, reject lists and counters [27], [26], [28])
  uint32_t n_mapped = o.counters[(LIST == 3   ? (after_coop ? 27 : GMX_CNT_GENERAL_REST)
                                 : LIST == 0 ? 13
                                 : LIST == 1 ? 4
                                 : LIST == 5 ? (after_coop ? 26 : 25)
                                 : LIST == 2 ? (after_coop ? 28 : 7)
                                             : 7) * GMX_CNT_STRIDE];
  const uint32_t m_start = LIST == 2 && !after_coop ? o.counters[10 * GMX_CNT_STRIDE] : 0u;
  const uint32_t *list = LIST == 3   ? (after_coop ? o.general_serial_list : o.general_rest_list)
                         : LIST == 0 ? o.cover_mid_list
                         : LIST == 1 ? o.cover_overflow_list
                         : LIST == 5 ? (after_coop ? o.inst_serial_list : o.inst_mapped_list)
                         : LIST == 2 ? (after_coop ? o.big_serial_list : o.big_mapped_list)
                                     : o.big_mapped_list;
  if (LIST == 4 && blockIdx.x == 0 && threadIdx.x == 0) o.counters[10 * GMX_CNT_STRIDE] = n_mapped;  // read by LIST 2 only
  if (threadIdx.x >= LANES) return;
#ifdef GMX_LOOP_STATS
  long long t_kernel = wall_clock64();
#endif
  const uint32_t lane_id = blockIdx.x * LANES + threadIdx.x;
  const uint32_t work_blocks = gridDim.x;
  // interleaved: a short queue spreads over all waves (few diverging lanes each) instead of filling the first ones
  for (uint32_t m = m_start + threadIdx.x * work_blocks + blockIdx.x; m < n_mapped; m += work_blocks * LANES) {
    uint32_t entry = list[m];
    uint32_t task, nf;
    const GmxFinalState *finals;
    const GmxPathNode *arena;
    {
      const GmxTaskStates ts = gmx_entry_states(entry, o, g);
      task = ts.task;
      nf = ts.nf;
      finals = ts.finals;
      arena = ts.arena;
    }
    uint32_t read = task >> 1;
    uint32_t len = read_len(b, read);
    Env env;
    env.scratch = LDS ? gmx_lds + threadIdx.x : acc.scratch_big + lane_id;
    env.stride = LDS ? LANES : acc.n_lanes_big;
    env.arena = arena;
    env.acc = acc.acc;
    env.log = acc.log;
    env.log_cursor = acc.log_cursor;
    env.log_cap = acc.log_cap;
    env.log_sites = acc.log_sites;
    env.status = GMX_TASK_MAPPED;
    env.log_at = 0;
#ifdef GMX_LOOP_STATS
    env.prof_list = LIST;
    env.prof_t = t_kernel;
    env.prof(6);  // from the start of the kernel (first task of the lane) or the end of the lane's previous task
    atomicAdd(&gmx_cover_stats[LIST * 16 + 7], 1ull);
#endif
    gmx_cover_task(ix, env, finals, nf, len, b.seeds[read], acc.rng_mode);
    env.log_abandon();
#ifdef GMX_LOOP_STATS
    env.prof(5);
    t_kernel = env.prof_t;
#endif
    if (env.status == GMX_TASK_OVERFLOW && LIST == 3) {  // nothing has been recorded for it yet: next scratch size
      o.cover_mid_list[atomicAdd(&o.counters[13 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_OVERFLOW && !BIG) {
      o.cover_overflow_list[atomicAdd(&o.counters[4 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_OVERFLOW) {  // beyond the largest fixed scratch: the last tier sizes one from its heap
      o.cover_huge_list[atomicAdd(&o.counters[15 * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status == GMX_TASK_LOGFULL) {  // nothing recorded: again once the host has drained the log
      o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
    } else if (env.status != GMX_TASK_MAPPED) {
      if (atomicCAS(&o.error[0], 0u, env.status) == 0u) o.error[1] = task;
    }
  }
  if (BIG) {  // this instance is the batch's last search / coverage launch: whichever block finishes last serves the last tier
    __shared__ uint32_t ticket;
    __threadfence();
    if (threadIdx.x == 0) ticket = atomicAdd(&o.counters[14 * GMX_CNT_STRIDE], 1u);
    __syncthreads();
    if (ticket == work_blocks - 1) {
      __threadfence();
      gmx_tail_stage(ix, b, o, g, acc);
    }
  }
}

// ---------------------------------------------------------------------------
// Single-instance tasks the compact path could not take — a nested traversing path, traversed sites that are not
// consecutive (children inside an MSA region), or more loci than the register slots of gmx_cover_single_nested hold —
// one lane per task with the loci in LDS (gmx_cover_single_nested_wide): no keys, no sort, no class merge, no draw.
// The general instances (cooperative, then serial) spent 0.8 ms of wall time per round on such tasks at configs[2]
// (profiles/round3/coop_phases_config2.txt): a single state of width one has ONE item, hence one class, and the draw
// cannot change the outcome (coverage_common.cpp:166-177 with one class and no non-variant instance selects it whatever
// the number drawn). What does not fit (several final states, wide intervals, more than 32 loci) goes on to them.
// ---------------------------------------------------------------------------
#define GMX_ONE_THREADS 64
struct OneEnv : CoverLogPart {
  uint32_t *scratch;  // this lane's words, GMX_ONE_THREADS apart
  const GmxPathNode *arena;
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const { return gmx_h_site(arena, h); }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const { return gmx_h_allele(arena, h); }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const { return gmx_h_next(arena, h); }
  __device__ __forceinline__ uint32_t sget(uint32_t w) const { return scratch[w * GMX_ONE_THREADS]; }
  __device__ __forceinline__ void sset(uint32_t w, uint32_t v) { scratch[w * GMX_ONE_THREADS] = v; }
};
__global__ void __launch_bounds__(GMX_ONE_THREADS) gmx_cover_one_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, CoverAcc acc,
                                                                        uint32_t enabled) {
  const uint32_t n = o.counters[8 * GMX_CNT_STRIDE];
  // (entry m = lane * blocks + block: a short queue — a few thousand entries among a million reads on a flat PRG — is spread
  //  over all workgroups, a handful of lanes each, instead of filling the first few waves with 64 divergent dependent-load
  //  chains apiece: the kernel's duration is that of its slowest wave)
  for (uint32_t m = threadIdx.x * gridDim.x + blockIdx.x; m < n; m += gridDim.x * GMX_ONE_THREADS) {
    const uint32_t entry = o.cover_general_list[m];
    const GmxTaskStates ts = gmx_entry_states(entry, o, g);
    bool taken = false;
    if (enabled && ts.nf == 1) {
      const GmxFinalState st = ts.finals[0];
      if (gmx_text_form(st.hi) || st.lo == st.hi) {
        OneEnv env;
        env.scratch = gmx_lds + threadIdx.x;
        env.arena = ts.arena;
        env.acc = acc.acc;
        env.log = acc.log;
        env.log_cursor = acc.log_cursor;
        env.log_cap = acc.log_cap;
        env.log_sites = acc.log_sites;
        env.status = GMX_TASK_MAPPED;
        env.log_at = 0;
        const uint32_t len = read_len(b, ts.task >> 1);
        if (!taken) taken = gmx_cover_single_nested_wide(ix, env, st, len);
        if (env.status == GMX_TASK_LOGFULL) {
          o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
        } else if (env.status != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, env.status) == 0u) {
          o.error[1] = ts.task;
        }
        env.log_abandon();
      }
    }
    if (!taken) o.general_rest_list[atomicAdd(&o.counters[GMX_CNT_GENERAL_REST * GMX_CNT_STRIDE], 1u)] = entry;
  }
}

// ---------------------------------------------------------------------------
// The general routine, cooperatively: 16 lanes per task, one lane per item. The serial instances above spend one lane
// on a whole task — a read with ten mapping instances is ten items' worth of loci, keys, a sort and a class search in
// one lane, and a wave of such lanes executes the union of all their branches: the SIMDs, not memory, set the pace.
// Here the items of a task are spread over lanes that all run the same short code:
//   units    a path-bearing final state is one unit (an item); a pathless one has one unit per occurrence, each a
//            non-variant instance or an allele-encapsulated item (encapsulated_search.cpp:30-107). Lanes load one final
//            state each, a prefix sum of the widths assigns units to lanes.
//   keys     every item lane runs gmx_item_loci + gmx_item_key on its own LDS scratch (capacity one item).
//   classes  every item lane compares its key with the group's other keys: the lanes whose key has no equal at a lower
//            lane lead a class; the rank of a class = the number of leaders with smaller keys (std::map order,
//            coverage_common.hpp:133).
//   draw     one seeded draw over non-variant instances + classes (coverage_common.cpp:95-108).
//   record   the leader of the drawn class merges its members' loci and hulls in the group's class scratch
//            (gmx_class_add_item: a set union, the order of the members is immaterial) and records.
// Tasks with more than 16 units, or exceeding a scratch capacity (nothing recorded by then), go to the serial instance
// of the same queue through a reject list. One wave per block, four tasks per wave, persistent over the queue.
// ---------------------------------------------------------------------------
// Scratch sizes per instance. An item's key holds its level-0 sites: a 150-base read inside an MSA region of configs[2]
// (a site every ~20 bases) has 9-12 of them — with room for 6, two thirds of that workload's tasks fell through to the
// one-lane instances (whose keys were as short: the last, global-memory one then took 9 of the batch's 15 ms).
template <int LIST>
struct CoopSizes {  // instances 3 and 2: the regular tasks' general instance, and what the large-capacity search mapped
  typedef CoverEnvT<1, 16, 24, 1, 24> Item;    // one item: its record, key, loci window (and the copy of its traversed list)
  typedef CoverEnvT<1, 1, 48, 48, 8> Class;    // the drawn class: union of loci, hull (no keys; the loci come from the members' windows)
};
template <>
struct CoopSizes<5> {  // instance 5: the instance-searched reads in repeats (many items, short paths)
  typedef CoverEnvT<1, 8, 12, 1> Item;
  typedef CoverEnvT<1, 1, 24, 24> Class;
};
template <int LIST>
constexpr uint32_t gmx_coop_lds_words() {  // + per group: the drawn item's traversed list as path nodes (the walk's handles)
  return 64u * GmxScratchFixed<typename CoopSizes<LIST>::Item>::total + 4u * GmxScratchFixed<typename CoopSizes<LIST>::Class>::total +
         4u * 3u * CoopSizes<LIST>::Item::P_MAX;
}

template <int LIST>
__global__ void __launch_bounds__(64) gmx_cover_coop_kernel(GmxIndexView ix, BatchView b, SearchOut o, BigOut g, CoverAcc acc) {
  typedef typename CoopSizes<LIST>::Item CoopItemEnv;
  typedef typename CoopSizes<LIST>::Class CoopClassEnv;
  typedef GmxScratch<CoopItemEnv> SI;
  typedef GmxScratch<CoopClassEnv> SC;
  const uint32_t n = o.counters[(LIST == 5 ? 25 : LIST == 2 ? 7 : GMX_CNT_GENERAL_REST) * GMX_CNT_STRIDE];
  const uint32_t n_first = LIST == 2 ? o.counters[10 * GMX_CNT_STRIDE] : 0u;  // instance 2 starts where instance 4 stopped
  const uint32_t *list = LIST == 5 ? o.inst_mapped_list : LIST == 2 ? o.big_mapped_list : o.general_rest_list;
  uint32_t *reject = LIST == 5 ? o.inst_serial_list : LIST == 2 ? o.big_serial_list : o.general_serial_list;
  uint32_t *reject_n = &o.counters[(LIST == 5 ? 26 : LIST == 2 ? 28 : 27) * GMX_CNT_STRIDE];
  const uint32_t lane = threadIdx.x, grp = lane >> 4, gl = lane & 15u, gbase = grp << 4;
  CoopItemEnv ie;
  ie.scratch = gmx_lds + lane;
  ie.stride = 64;
  CoopClassEnv ce;
  ce.scratch = gmx_lds + 64u * GmxScratchFixed<CoopItemEnv>::total + grp;
  ce.stride = 4;
  ie.acc = ce.acc = acc.acc;
  ie.log = ce.log = acc.log;
  ie.log_cursor = ce.log_cursor = acc.log_cursor;
  ie.log_cap = ce.log_cap = acc.log_cap;
  ie.log_sites = ce.log_sites = acc.log_sites;
  ie.log_at = ce.log_at = 0;
#ifdef GMX_LOOP_STATS
  ie.prof_list = 6;
  ce.prof_list = 7;
#endif
  const uint32_t kofs = SI::keys(ie);  // key word t of lane L: gmx_lds[(kofs + t) * 64 + L]
  for (uint32_t m0 = n_first + blockIdx.x * 4u; m0 < n; m0 += gridDim.x * 4u) {  // wave-uniform: every lane takes part in the shuffles
    const uint32_t m = m0 + grp;
    const bool have = m < n;
#ifdef GMX_LOOP_STATS  // wave-level phase times of this instance: [0] units, [1] loci + keys, [2] classes + draw, [3] class merge + record; [7] rounds
    long long tp = wall_clock64();
#define GMX_COOP_PHASE(k) do { const long long tq = wall_clock64(); if (lane == 0) atomicAdd(&gmx_coop_stats[LIST * 8 + (k)], (unsigned long long)(tq - tp)); tp = tq; } while (0)
    if (lane == 0) atomicAdd(&gmx_coop_stats[LIST * 8 + 7], 1ull);
#else
#define GMX_COOP_PHASE(k) do { } while (0)
#endif
    const uint32_t entry = have ? list[m] : 0u;
    GmxTaskStates ts{0u, 0u, nullptr, nullptr};
    if (have) ts = gmx_entry_states(entry, o, g);
    bool rejected = ts.nf > 16u;
    // --- units ---
    GmxFinalState st{0u, 0u, GMX_NIL, GMX_NIL};
    uint32_t w = 0;
    if (have && !rejected && gl < ts.nf) {
      st = ts.finals[gl];
      w = (st.traversed != GMX_NIL || st.traversing != GMX_NIL || gmx_text_form(st.hi)) ? 1u : min(st.hi - st.lo, 16u) + 1u;
    }
    uint32_t incl = w;
#pragma unroll
    for (uint32_t d = 1; d < 16; d <<= 1) {
      const uint32_t v = __shfl_up(incl, d, 16);
      if (gl >= d) incl += v;
    }
    const uint32_t start = incl - w, n_units = __shfl(incl, 15, 16);
    rejected = rejected || n_units > 16u;
#ifdef GMX_LOOP_STATS
    if (have && gl == 0 && rejected) ie.why(3);  // more than 16 units
#endif
    uint32_t f_lo = 0, f_hi = 0, f_tvd = GMX_NIL, f_tvg = GMX_NIL, f_start = 0;
    bool unit = false;
#pragma unroll 4
    for (uint32_t f = 0; f < 16; ++f) {
      const uint32_t s = __shfl(start, f, 16), ww = __shfl(w, f, 16);
      const uint32_t lo = __shfl(st.lo, f, 16), hi = __shfl(st.hi, f, 16), tvd = __shfl(st.traversed, f, 16), tvg = __shfl(st.traversing, f, 16);
      if (gl >= s && gl < s + ww) {
        unit = true;
        f_lo = lo;
        f_hi = hi;
        f_tvd = tvd;
        f_tvg = tvg;
        f_start = s;
      }
    }
    unit = unit && have && !rejected;
    bool is_item = false, nonvar = false;
    uint32_t i_lo = 0, i_hi = 0, enc_site = 0;
    int32_t enc_allele = -1;
    if (unit) {
      if (f_tvd != GMX_NIL || f_tvg != GMX_NIL) {
        is_item = true;
        i_lo = f_lo;
        i_hi = f_hi;
      } else {
        const uint32_t i = f_lo + (gl - f_start);
        const GmxNode &nd = ix.nodes[ix.pos_node[gmx_occ_pos(ix, f_hi, i)]];
        if (nd.site == 0) {
          nonvar = true;
        } else {
          is_item = true;
          i_lo = i;
          i_hi = gmx_text_form(f_hi) ? f_hi : i;
          enc_site = nd.site;
          enc_allele = nd.allele;
        }
      }
    }
    const uint32_t items16 = (uint32_t)(__ballot(is_item) >> gbase) & 0xFFFFu;
    const uint32_t nonvariant = __popc((uint32_t)(__ballot(nonvar) >> gbase) & 0xFFFFu);
    GMX_COOP_PHASE(0);
    // --- loci and key of the lane's item ---
    ie.arena = ts.arena;
    ie.status = GMX_TASK_MAPPED;
    if (is_item) {
      ie.sset(SI::items + 0, i_lo);
      ie.sset(SI::items + 1, i_hi);
      ie.sset(SI::items + 2, f_tvd);
      ie.sset(SI::items + 3, f_tvg);
      ie.sset(SI::items + 4, enc_site);
      ie.sset(SI::items + 5, (uint32_t)enc_allele);
      const uint32_t nl = gmx_item_loci(ix, ie, 0, 0);
      if (nl != 0xFFFFFFFFu) {
        gmx_item_key(ix, ie, 0, 0, nl);
        ie.sset(SI::order(ie), nl);  // (the order word is free with one item: the class's first lane reads the window's length here)
      }
    }
    uint32_t err = (is_item && ie.status != GMX_TASK_MAPPED && ie.status != GMX_TASK_OVERFLOW) ? ie.status : 0u;
    rejected = rejected || (((uint32_t)(__ballot(is_item && ie.status == GMX_TASK_OVERFLOW) >> gbase) & 0xFFFFu) != 0u);
    bool failed = (((uint32_t)(__ballot(err != 0u) >> gbase) & 0xFFFFu) != 0u);
    __syncthreads();  // the keys are in LDS
    GMX_COOP_PHASE(1);
    // --- classes ---
    uint32_t lt = 0, eq = 0;
    if (is_item && !rejected && !failed) {
      const uint32_t la = gmx_lds[kofs * 64u + lane];
      for (uint32_t rest = items16 & ~(1u << gl); rest; rest &= rest - 1u) {
        const uint32_t j = (uint32_t)__ffs(rest) - 1u, other = gbase + j;
        const uint32_t lb = gmx_lds[kofs * 64u + other];
        const uint32_t mlen = min(la, lb);
        int cmp = 0;  // sign of (other's key - mine)
        for (uint32_t t = 0; t < mlen && cmp == 0; ++t) {
          const uint32_t va = gmx_lds[(kofs + 1u + t) * 64u + lane], vb = gmx_lds[(kofs + 1u + t) * 64u + other];
          cmp = vb < va ? -1 : (vb > va ? 1 : 0);
        }
        if (cmp == 0) cmp = lb < la ? -1 : (lb > la ? 1 : 0);
        if (cmp < 0) lt |= 1u << j;
        if (cmp == 0) eq |= 1u << j;
      }
    }
    const bool leader = is_item && !rejected && !failed && (eq & ((1u << gl) - 1u)) == 0u;
    const uint32_t leaders16 = (uint32_t)(__ballot(leader) >> gbase) & 0xFFFFu;
    const uint32_t n_classes = __popc(leaders16), rank = __popc(lt & leaders16);
    // --- the draw ---
    bool member = false;
    if (have && !rejected && !failed && items16 != 0u) {
      uint32_t r = 0;
      if (!gmx_uniform_1_to_n(b.seeds[ts.task >> 1], nonvariant + n_classes, acc.rng_mode, r)) {
        err = GMX_TASK_ERROR;
      } else if (r > nonvariant) {
        member = is_item && rank == r - nonvariant - 1u;
      }
    }
    const uint32_t members16 = (uint32_t)(__ballot(member) >> gbase) & 0xFFFFu;
    GMX_COOP_PHASE(2);
    // --- the drawn class: its first lane merges the members and records ---
    bool class_overflow = false, class_logfull = false;
    if (member && gl == (uint32_t)__ffs(members16) - 1u) {
      const uint32_t read = ts.task >> 1;
      const uint32_t len = read_len(b, read);
      ce.arena = ts.arena;
      ce.status = GMX_TASK_MAPPED;
      ce.log_at = 0;
      uint32_t n_loci = 0, n_hull = 0;
      bool ok = true;
      for (uint32_t rest = members16; rest && ok; rest &= rest - 1u) {
        const uint32_t other = gbase + (uint32_t)__ffs(rest) - 1u;
#pragma unroll
        for (uint32_t t = 0; t < SI::ITEM_W; ++t) ce.sset(SC::items + t, gmx_lds[(SI::items + t) * 64u + other]);
        // the member's loci window as its lane left it (gmx_class_add_item would run gmx_item_loci again): set union
        const uint32_t nl_m = gmx_lds[SI::order(ie) * 64u + other], first = n_loci;
        for (uint32_t i = 0; i < nl_m && ok; ++i) {
          const uint32_t site = gmx_lds[(SI::loci(ie) + 2u * i) * 64u + other], al = gmx_lds[(SI::loci(ie) + 2u * i + 1u) * 64u + other];
          bool dup = false;
          for (uint32_t j = 0; j < first && !dup; ++j) dup = ce.sget(SC::loci(ce) + 2u * j) == site && ce.sget(SC::loci(ce) + 2u * j + 1u) == al;
          if (dup) continue;
          if (n_loci >= ce.loc_max()) {
            ce.fail(GMX_TASK_OVERFLOW);
            ok = false;
            break;
          }
          ce.sset(SC::loci(ce) + 2u * n_loci, site);
          ce.sset(SC::loci(ce) + 2u * n_loci + 1u, al);
          ++n_loci;
        }
        // The walk consumes the member's traversed list newest first, a dependent arena load per locus (the fast
        // pass's arena keeps a task's nodes n_tasks entries apart: every one a miss). The member's lane has copied
        // the list to its scratch: laid out as path nodes in LDS, handle = index, the walk never leaves the CU for it.
        const uint32_t nt_m = gmx_lds[(SI::path(ie) + 2u * CoopItemEnv::P_MAX) * 64u + other];
        if (nt_m != 0xFFFFFFFFu && nt_m != 0u && ce.sget(SC::items + 2) != GMX_NIL) {
          GmxPathNode *ln = reinterpret_cast<GmxPathNode *>(gmx_lds + 64u * GmxScratchFixed<CoopItemEnv>::total +
                                                            4u * GmxScratchFixed<CoopClassEnv>::total + grp * 3u * CoopItemEnv::P_MAX);
          for (uint32_t i = 0; i < nt_m; ++i)
            ln[i] = GmxPathNode{gmx_lds[(SI::path(ie) + 2u * i) * 64u + other], (int32_t)gmx_lds[(SI::path(ie) + 2u * i + 1u) * 64u + other],
                                i + 1u < nt_m ? i + 1u : GMX_NIL};
          ce.arena = ln;
          ce.sset(SC::items + 2, 0u);
        } else {
          ce.arena = ts.arena;
        }
        ok = ok && gmx_item_per_base(ix, ce, 0, len, n_hull);
      }
      ce.arena = ts.arena;
      if (ok) gmx_class_record(ix, ce, n_loci, n_hull);
      class_overflow = ce.status == GMX_TASK_OVERFLOW;
      class_logfull = ce.status == GMX_TASK_LOGFULL;
      if (ce.status != GMX_TASK_MAPPED && !class_overflow && !class_logfull) err = ce.status;
      ce.log_abandon();
    }
    rejected = rejected || (((uint32_t)(__ballot(class_overflow) >> gbase) & 0xFFFFu) != 0u);
    const bool logfull = (((uint32_t)(__ballot(class_logfull) >> gbase) & 0xFFFFu) != 0u);
    if (have && gl == 0 && logfull)
      o.log_retry_list[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], 1u)] = entry;
    if (have && gl == 0 && rejected) reject[atomicAdd(reject_n, 1u)] = entry;
    if (err != 0u && atomicCAS(&o.error[0], 0u, err) == 0u) o.error[1] = ts.task;
    __syncthreads();  // the scratch is reused by the next round
    GMX_COOP_PHASE(3);
  }
#undef GMX_COOP_PHASE
}

// Path handles of a GmxCoverRec: traversed loci are addressed by their index in the record (newest first), the
// traversing path is an inline handle (gmx_types.h) or nil.
struct CompactRec {  // a GmxCoverRec in scalars (with the array member the compiler kept the record in scratch memory and indexed it)
  uint32_t p, len_n, tvg, s0, s1, s2, a01, a2;
  __device__ __forceinline__ CompactRec &operator=(const GmxCoverRec &r) {
    p = r.p, len_n = r.len_n, tvg = r.tvg, s0 = r.site[0], s1 = r.site[1], s2 = r.site[2], a01 = r.a01, a2 = r.a2;
    return *this;
  }
};
struct CompactEnv : CoverLogPart {
  CompactRec rec;
  __device__ __forceinline__ uint32_t n_trav() const { return (rec.len_n >> 16) & 31u; }
  __device__ __forceinline__ bool run_form() const { return (rec.len_n & GMX_REC_RUN_FLAG) != 0; }
  __device__ __forceinline__ uint32_t h_site(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return 5u + 2u * (h & ~GMX_INLINE_FLAG);
    if (run_form()) return rec.s0 + 2u * h;
    return h == 0 ? rec.s0 : (h == 1 ? rec.s1 : rec.s2);
  }
  __device__ __forceinline__ int32_t h_allele(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return -1;
    if (run_form()) {
      const uint32_t q = h >> 2, w = q == 0 ? rec.s1 : q == 1 ? rec.s2 : q == 2 ? rec.a01 : rec.a2;
      return (int32_t)((w >> (8u * (h & 3u))) & 0xFFu);
    }
    return (int32_t)(h == 0 ? (rec.a01 & 0xFFFFu) : (h == 1 ? (rec.a01 >> 16) : rec.a2));
  }
  __device__ __forceinline__ uint32_t h_next(uint32_t h) const {
    if (h & GMX_INLINE_FLAG) return GMX_NIL;
    return h + 1 < n_trav() ? h + 1 : GMX_NIL;
  }
};

// The common case, one lane per compact record and no scratch (gmx_cover_single, gmx_cover.h): a task with ONE
// final state of width one. Few registers, a coalesced queue, region-local tables.
template <bool NESTED>  // (two kernels: the nested routine's locus arrays would cost the flat one registers and scratch)
__device__ __forceinline__ void gmx_cover_single_rec(const GmxIndexView &ix, const SearchOut &o, const CoverAcc &acc, size_t rec_idx,
                                                     uint32_t *handoff_list, uint32_t handoff_counter) {
  CompactEnv env;
  env.rec = o.cover_recs[rec_idx];
  env.acc = acc.acc;
  env.log = acc.log;
  env.log_cursor = acc.log_cursor;
  env.log_cap = acc.log_cap;
  env.log_sites = acc.log_sites;
  env.status = GMX_TASK_MAPPED;
  env.log_at = 0;
  const GmxFinalState st{env.rec.p, GMX_TEXT_MARK, env.n_trav() ? 0u : GMX_NIL, env.rec.tvg};
  if constexpr (!NESTED) {
    gmx_cover_single(ix, env, st, env.rec.len_n & 0xFFFFu);
  } else if (!gmx_cover_single_nested(ix, env, st, env.rec.len_n & 0xFFFFu)) {  // many loci: the general instance next
    handoff_list[atomicAdd(&o.counters[handoff_counter * GMX_CNT_STRIDE], 1u)] = o.cover_rec_task[rec_idx];
  }
  if (env.status == GMX_TASK_LOGFULL) {  // nothing recorded: again once the host has drained the log
    o.log_retry_recs[atomicAdd(&o.counters[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE], 1u)] = (uint32_t)rec_idx;
  } else if (env.status != GMX_TASK_MAPPED && atomicCAS(&o.error[0], 0u, env.status) == 0u) {
    o.error[1] = o.cover_rec_task[rec_idx];
  }
  env.log_abandon();
}

template <bool NESTED>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t region = blockIdx.x & (GMX_REGIONS - 1);  // = the XCD this workgroup runs on (round-robin dispatch)
  const uint32_t n_mapped = o.counters[(16 + region) * GMX_CNT_STRIDE];
  const uint32_t m = (blockIdx.x / GMX_REGIONS) * GMX_BLOCK + threadIdx.x;
  if (m >= n_mapped) return;
  gmx_cover_single_rec<NESTED>(ix, o, acc, (size_t)region * o.region_cap + m, o.cover_general_list, 8u);
}

// The same queue on a flat PRG whose sites have geometry records (GmxSiteGeo): gmx_cover_jump alone — no walk, no GmxSite, the
// increments staged in LDS between its check pass and the recording — and what it declines (a site of more than 8 alleles
// or an allele of 255+ bases on the path) goes to gmx_cover_single_rest_kernel, the routine above over a list. (Forced to 64
// registers for 8 waves per SIMD it spills and is no faster: GMX_JUMP_MIN_BLOCKS; profiles/round4/cover_jump_variants_config3.txt.)
struct StageLds {
  uint32_t *w;  // this lane's words, GMX_BLOCK apart
  __device__ __forceinline__ uint32_t cap() const { return GMX_STAGE_MAX; }
  __device__ __forceinline__ void put(uint32_t i, uint32_t v) { w[i * GMX_BLOCK] = v; }
  __device__ __forceinline__ uint32_t get(uint32_t i) const { return w[i * GMX_BLOCK]; }
};
#ifndef GMX_JUMP_MIN_BLOCKS
#define GMX_JUMP_MIN_BLOCKS 1  // (8 = 64 registers, 8 waves per SIMD with spills: measured slower, and its LDS crowds out the side streams' kernels)
#endif
__global__ void __launch_bounds__(GMX_BLOCK, GMX_JUMP_MIN_BLOCKS) gmx_cover_jump_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t region = blockIdx.x & (GMX_REGIONS - 1);
  const uint32_t n_mapped = o.counters[(16 + region) * GMX_CNT_STRIDE];
  const uint32_t m = (blockIdx.x / GMX_REGIONS) * GMX_BLOCK + threadIdx.x;
  if (m >= n_mapped) return;
  const size_t rec_idx = (size_t)region * o.region_cap + m;
  CompactEnv env;
  env.rec = o.cover_recs[rec_idx];
  env.acc = acc.acc;
  env.log = nullptr;
  env.log_cursor = nullptr;
  env.log_cap = 0;
  env.log_sites = 0;
  env.status = GMX_TASK_MAPPED;
  env.log_at = env.log_end = 0;
  const uint32_t p = env.rec.p, tvd = env.n_trav() ? 0u : GMX_NIL, tvg = env.rec.tvg;
  StageLds stage{gmx_lds + threadIdx.x};
  const bool done = gmx_cover_jump(ix, env, stage, p, tvd, tvg, env.rec.len_n & 0xFFFFu);
  if (!done) o.single_rest_list[atomicAdd(&o.counters[GMX_CNT_SINGLE_REST * GMX_CNT_STRIDE], 1u)] = (uint32_t)rec_idx;
}
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_rest_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc) {
  const uint32_t n = o.counters[GMX_CNT_SINGLE_REST * GMX_CNT_STRIDE];
  for (uint32_t i = blockIdx.x * GMX_BLOCK + threadIdx.x; i < n; i += gridDim.x * GMX_BLOCK)
    gmx_cover_single_rec<false>(ix, o, acc, o.single_rest_list[i], o.cover_general_list, 8u);
}

// ---- grouped log full: the batch's failed entries again, after the host has drained the log (launch_log_replay) ----
// moves the retry lists' lengths to where the replay kernels read them and empties the retry lists for this round
__global__ void gmx_log_replay_setup_kernel(SearchOut o, const uint32_t *retry_huge_in) {
  uint32_t *c = o.counters;
  const uint32_t n_entries = c[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE], n_recs = c[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE],
                 n_huge = c[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE];
  for (uint32_t i = threadIdx.x; i < n_huge; i += blockDim.x) o.huge_list[i] = retry_huge_in[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    c[4 * GMX_CNT_STRIDE] = n_entries;                 // gmx_cover_kernel<CoverEnvBig, 1> reads its queue length here
    c[GMX_CNT_REPLAY_RECS * GMX_CNT_STRIDE] = n_recs;
    c[11 * GMX_CNT_STRIDE] = n_huge;                   // the last tier's search items
    c[15 * GMX_CNT_STRIDE] = 0;
    c[14 * GMX_CNT_STRIDE] = 0;                        // the ticket counter of the last-tier stage
    c[GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE] = 0;
    c[GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE] = 0;
    c[GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE] = 0;
  }
}
template <bool NESTED>
__global__ void __launch_bounds__(GMX_BLOCK) gmx_cover_single_replay_kernel(GmxIndexView ix, BatchView b, SearchOut o, CoverAcc acc,
                                                                            const uint32_t *recs_in) {
  const uint32_t n = o.counters[GMX_CNT_REPLAY_RECS * GMX_CNT_STRIDE];
  for (uint32_t i = blockIdx.x * GMX_BLOCK + threadIdx.x; i < n; i += gridDim.x * GMX_BLOCK)
    gmx_cover_single_rec<NESTED>(ix, o, acc, recs_in[i], o.cover_overflow_list, 4u);  // (nested, many loci: the large scratch, which runs next)
}

// The five uint64 read counters <-> 16-bit limbs in uint32 words, so that they travel inside the one uint32
// all-reduce(sum) of the coverage block: limb sums of up to 65536 ranks cannot overflow (gmx_coverage_reduce_*).
__global__ void gmx_stats_limbs_kernel(unsigned long long *stats, uint32_t *limbs, int recombine) {
  const uint32_t t = threadIdx.x;
  if (!recombine) {
    if (t < 20) limbs[t] = (uint32_t)((stats[t >> 2] >> (16 * (t & 3))) & 0xFFFFull);
    else if (t < 32) limbs[t] = 0;
  } else if (t < 5) {
    unsigned long long v = 0;
    for (int l = 3; l >= 0; --l) v = (v << 16) + limbs[4 * t + l];  // limb sums carry into the limbs above
    stats[t] = v;
  }
}

// Validation + packing, one lane per read. Reads holding a byte outside 1..4 are skipped as a whole
// (encode_dna_bases, utils.cpp:73-92). The packed form is two bit planes per 32 bases (uint2: low bits, high
// bits of the codes 0..3): the search kernels compare 32 bases per step against the PRG's planes (GmxTextRec),
// and a single base is two bit extracts.
//
// A block owns GMX_PACK_READS consecutive reads, whose bytes and whose packed pairs are both contiguous:
// the bytes are staged through LDS with coalesced 16-byte loads, packed from LDS (aligned dwords joined with
// v_alignbyte), and written back from LDS with coalesced stores. Blocks whose reads do not fit the LDS
// window (very long reads) take the direct per-lane path.
#define GMX_PACK_READS 128
#define GMX_PACK_IN_BYTES (24 * 1024)
#define GMX_PACK_OUT_PAIRS (GMX_PACK_IN_BYTES / 32 + GMX_PACK_READS + 8)
typedef uint32_t __attribute__((aligned(1))) gmx_u32_unaligned;
// four bytes -> four bits of each plane (bit i = byte i), flagging bytes outside 1..4
__device__ __forceinline__ void pack4(uint32_t x, uint32_t &lo, uint32_t &hi, uint32_t &bad) {
  uint32_t y = x - 0x01010101u;                         // per-byte code 0..3 when every byte is in 1..4
  bad |= ((y & ~x & 0x80808080u) | (y & 0xFCFCFCFCu));  // a zero byte, or a byte > 4
  lo = (((y & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
  hi = ((((y >> 1) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint2 pack_tail(const uint8_t *p, uint32_t rem, uint32_t &bad) {
  uint2 out = make_uint2(0, 0);
  for (uint32_t j = 0; j < rem; ++j) {
    uint32_t x = p[j];
    if (x < 1 || x > 4) bad = 1;
    out.x |= ((x - 1u) & 1u) << j;
    out.y |= (((x - 1u) >> 1) & 1u) << j;
  }
  return out;
}
#define GMX_PACK_THREADS (2 * GMX_PACK_READS)  // two threads per read: twice the loads and stores in flight per LDS window
__global__ void __launch_bounds__(GMX_PACK_THREADS) gmx_pack_kernel(BatchView b, uint8_t *skip, uint2 *packed, uint32_t *counters,
                                                                    uint32_t *zero, uint32_t zero_words) {
  __shared__ uint4 in4[GMX_PACK_IN_BYTES / 16 + 2];
  __shared__ uint2 outp[GMX_PACK_OUT_PAIRS];
  // the queue counters are per batch: this is the batch's first kernel and everything that counts comes after it
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < GMX_N_COUNTERS * GMX_CNT_STRIDE; i += GMX_PACK_THREADS) counters[i] = 0;
  // a reset queued just ahead of this batch (gmx_engine_reset_async): the accumulator block, read counters and log
  // cursor zeroed here instead of by a memset of their own (nothing in this kernel touches them otherwise)
  for (uint32_t i = blockIdx.x * GMX_PACK_THREADS + threadIdx.x; i < zero_words; i += gridDim.x * GMX_PACK_THREADS) zero[i] = 0;
  const uint32_t r0 = blockIdx.x * GMX_PACK_READS;
  const uint32_t r1 = min(r0 + GMX_PACK_READS, b.n_reads);
  const uint32_t read = r0 + (threadIdx.x >> 1), half = threadIdx.x & 1u;  // the two threads of a read are neighbours
  const uint64_t s0 = b.offsets[r0], s1 = b.offsets[r1];
  const uintptr_t g0 = reinterpret_cast<uintptr_t>(b.reads + s0);
  const uint32_t shift = (uint32_t)(g0 & 15u);
  const uint64_t span = (s1 - s0) + shift;
  if (span <= GMX_PACK_IN_BYTES) {  // block-uniform
    const uint4 *src = reinterpret_cast<const uint4 *>(g0 - shift);
    const uint32_t n16 = (uint32_t)((span + 15) >> 4);
    {  // independent 16-byte loads in flight per thread and round
      uint32_t i = threadIdx.x;
      for (; i + 2 * GMX_PACK_THREADS < n16; i += 3 * GMX_PACK_THREADS) {
        const uint4 v0 = src[i], v1 = src[i + GMX_PACK_THREADS], v2 = src[i + 2 * GMX_PACK_THREADS];
        in4[i] = v0;
        in4[i + GMX_PACK_THREADS] = v1;
        in4[i + 2 * GMX_PACK_THREADS] = v2;
      }
      for (; i < n16; i += GMX_PACK_THREADS) in4[i] = src[i];
    }
    const uint64_t po0 = pack_off(b, r0);
    const uint32_t n_out = (uint32_t)(pack_off(b, r1) - po0);
    for (uint32_t i = threadIdx.x; i < n_out; i += GMX_PACK_THREADS) outp[i] = make_uint2(0, 0);
    __syncthreads();
    if (read < r1) {
      const uint64_t s = b.offsets[read];
      const uint32_t len = (uint32_t)(b.offsets[read + 1] - s);
      const uint32_t q = shift + (uint32_t)(s - s0);
      const uint32_t *w = reinterpret_cast<const uint32_t *>(in4);
      const uint8_t *bytes = reinterpret_cast<const uint8_t *>(in4);
      uint2 *out = outp + (uint32_t)(pack_off(b, read) - po0);
      const uint32_t full = len >> 5, first_half = (full + 1u) >> 1;
      const uint32_t c0 = half ? first_half : 0u, c1 = half ? full : first_half;  // this thread's pairs
      uint32_t idx = (q >> 2) + 8u * c0;
      const uint32_t sh = q & 3u;
      uint32_t bad = 0;
      uint32_t carry = w[idx];
      for (uint32_t c = c0; c < c1; ++c) {
        uint2 pair = make_uint2(0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          uint32_t nxt = w[++idx], lo, hi;
          pack4(__builtin_amdgcn_alignbyte(nxt, carry, sh), lo, hi, bad);
          pair.x |= lo << (4 * j);
          pair.y |= hi << (4 * j);
          carry = nxt;
        }
        out[c] = pair;
      }
      const uint32_t rem = len & 31u;
      if (half && rem) out[full] = pack_tail(bytes + q + full * 32, rem, bad);
      bad |= (uint32_t)__shfl_xor((int)bad, 1);
      if (!half) skip[read] = bad ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_out; i += GMX_PACK_THREADS) packed[po0 + i] = outp[i];
    return;
  }
  if (read >= r1 || half) return;
  uint64_t s = b.offsets[read], e = b.offsets[read + 1];
  uint32_t len = (uint32_t)(e - s);
  const uint8_t *p = b.reads + s;
  uint2 *out = packed + pack_off(b, read);
  uint32_t bad = 0;
  uint32_t full = len >> 5;
  for (uint32_t c = 0; c < full; ++c) {
    uint2 pair = make_uint2(0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t lo, hi;
      pack4(*reinterpret_cast<const gmx_u32_unaligned *>(p + c * 32 + j * 4), lo, hi, bad);
      pair.x |= lo << (4 * j);
      pair.y |= hi << (4 * j);
    }
    out[c] = pair;
  }
  uint32_t rem = len & 31u;
  if (rem) out[full] = pack_tail(p + full * 32, rem, bad);
  skip[read] = bad ? 1 : 0;
}

// ===========================================================================
// engine (host side of the device half)
// ===========================================================================
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gmx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
      return GMX_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

// kernels timed one by one besides gmx_extend_kernel (gmx_timing::kernel_ms; include/gmx.h lists them)
enum : int { GMX_TK_SEED = 0, GMX_TK_FILTER0, GMX_TK_FILTER1, GMX_TK_SINGLE, GMX_TK_EXTEND2, GMX_TK_UNPACK, GMX_TK_N };
static_assert(GMX_TK_N <= GMX_TIMED_KERNELS, "gmx_timing::kernel_ms holds GMX_TIMED_KERNELS entries");

struct gmx_engine {
  gmx_engine_opts opts;
  GmxIndexView dview;  // device pointers
  std::vector<void *> allocs;
  uint64_t index_bytes = 0;
  // accumulators
  uint32_t *d_fused = nullptr, *d_limbs = nullptr;  // accumulator block (n_acc words, gmx_types.h) | 32 counter-limb words
  size_t n_fused = 0, n_acc = 0;
  std::vector<uint32_t> phys_allele, phys_pb, phys_grouped;  // logical slot -> slot of the block (gmx_coverage_fetch)
  std::vector<uint32_t> hit_fix;                             // hit counters and the logical slots they count for
  unsigned long long *d_stats = nullptr;  // with d_log_cursor behind the coverage block: one memset resets all of it
  uint32_t *d_error = nullptr;
  uint32_t *d_log = nullptr, *d_log_cursor = nullptr;
  uint32_t log_cap = 0;
  uint32_t n_allele = 0, n_pb = 0, n_grouped = 0;
  // batch workspace (sized for max_batch_reads)
  uint64_t cap_reads = 0;
  uint8_t *d_skip = nullptr;
  uint2 *d_packed = nullptr;
  uint64_t cap_packed = 0;
  uint32_t *d_status = nullptr, *d_n_final = nullptr, *d_mapped = nullptr, *d_overflow = nullptr, *d_counters = nullptr;
  uint32_t *d_general_rest = nullptr;
  uint32_t *d_single_rest = nullptr;
  bool cover_jump = false;  // gmx_cover_jump_kernel + gmx_cover_single_rest_kernel instead of gmx_cover_single_kernel<false>
  uint32_t *d_task_lists = nullptr;  // SearchOut::task_lists: d_overflow, d_overflow2, d_alive, d_dead, d_dead2, d_cover_general are its slices
  GmxSeed *d_alive_seed = nullptr;
  uint32_t *d_alive = nullptr, *d_dead = nullptr, *d_dead2 = nullptr;
  uint64_t *d_seed_cursor = nullptr;
  bool seed_cursor = false;  // the index has many multi-state k-mer entries: kernels instantiated with the seed cursor
  GmxFinalState *d_finals = nullptr;
  GmxPathNode *d_arena = nullptr;
  GmxCoverRec *d_cover_recs = nullptr;
  BigOut big{};
  uint32_t *d_scratch_big = nullptr, *d_cover_overflow = nullptr;
  uint32_t cover_big_lanes = 0;
  uint32_t *d_big_mapped = nullptr, *d_cover_general = nullptr, *d_cover_mid = nullptr, *d_overflow2 = nullptr;
  uint32_t *d_huge = nullptr, *d_cover_huge = nullptr, *d_huge_retry = nullptr;  // the last tier's queues (gmx_tail_stage)
  uint32_t *d_inst_list = nullptr, *d_inst_sa = nullptr, *d_inst_remaining = nullptr, *d_inst_mapped = nullptr;  // instance lanes (gmx_extend_inst_kernel)
  uint32_t inst_cap = 0;
  GmxPathNode *d_inst_arena = nullptr;
  GmxFinalState *d_inst_states = nullptr;
  uint32_t *d_inst_first = nullptr, *d_inst_width = nullptr;
  uint32_t *d_inst_serial = nullptr, *d_general_serial = nullptr, *d_big_serial = nullptr, *d_overflow3 = nullptr;  // what gmx_cover_coop_kernel leaves to the serial instances
  bool coop = true;  // GMX_NO_COOP=1 in the environment: serial coverage instances only (A/B runs)
  uint32_t *d_heap = nullptr;      // ... and its memory
  uint64_t heap_words = 0;
  bool log_sites = false;          // the index has sites with more than 8 alleles
  // grouped log: drained into `log_counts` (records with counts) whenever the device log may run full, and at fetch time
  std::map<std::vector<uint32_t>, uint64_t> log_counts;  // key = [site_index, ids...]
  // Exact accounting (round 3): after every batch of an engine whose index uses the log, the log cursor and the lengths
  // of the three retry lists are copied to page-locked words; before the next batch (and before any reader of the
  // coverage) log_settle() looks at them: entries that found the log full are redone after a drain (launch_log_replay),
  // and the log is drained once it is half full. No assumed bound on what a read appends.
  uint32_t *h_log_state = nullptr;     // [cursor, retry entries, retry records, retry last-tier tasks]
  hipEvent_t ev_log_state = nullptr;
  bool log_state_pending = false;
  uint32_t *d_log_retry[2] = {nullptr, nullptr}, *d_log_retry_recs[2] = {nullptr, nullptr}, *d_log_retry_huge[2] = {nullptr, nullptr};
  int log_retry_side = 0;              // which of the two sets the kernels append to
  BatchView last_b{};
  SearchOut last_o{};
  CoverAcc last_acc{};
  size_t last_big_lds = 0;
  uint64_t log_replays = 0, log_replayed_entries = 0;  // statistics (tests)
  uint64_t log_known = 0;          // log words in use after the last drain / look ...
  uint64_t log_reads_since = 0;    // (unused since round 3: the fill is read back after every batch)
  // gmx_engine_reset_async leaves its memset pending: the next batch's pack kernel zeroes the block when it is launched
  // on the same stream (one command and one dependent-launch gap less per job); every other reader of the accumulators
  // issues the memset first (flush_reset)
  bool reset_pending = false;
  hipStream_t reset_stream = nullptr;
  hipStream_t side2_stream = nullptr;
  hipEvent_t ev_fork2 = nullptr, ev_side1 = nullptr, ev_filter = nullptr;
  hipStream_t side_stream = nullptr;  // large-capacity search + its coverage run beside filter/cover
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_wait = nullptr;  // blocking event of gmx_quiesce
  struct GmxDeviceIndex *shared_index = nullptr;  // the device copy of the index tables, shared with the other engines of this index on this device
  uint32_t filter_lds_words = 0;  // > 0: the k-mer presence bitmap fits LDS (gmx_filter_lds_kernel)
  const uint32_t *d_kmer_planar = nullptr;  // that bitmap indexed by planar k-mer code (all_kmers_present_planar)
  const uint32_t *d_absent = nullptr;       // the k-mers that do NOT occur, when they are few (gmx_filter_absent_kernel)
  uint32_t n_absent = 0;
  bool use_absent = false;
  uint32_t n_cus = 256;
  uint32_t probe_iters = GMX_PROBE_ITERS;  // wave-loop iterations before the probe kernel parks what is left
  uint32_t extend_cap = 0;      // iterations of the LAST pass after which a task goes to the large-capacity route (0: runs to the end)
  uint32_t extend_budget = 8;   // wave-loop iterations of the extend kernel before a lane with work left is parked for the second pass
  bool seeds_in_place = false;  // gmx_engine_seeds_in_place
  uint32_t extend_passes = 1;   // launches over the stragglers (<= GMX_EXTRA_PASSES); all but the last with a budget of their own
  uint32_t extend_budget2[GMX_EXTRA_PASSES] = {24, 96, 0};
                                // (GMX_EXTEND_BUDGET in the environment; 0 = one pass)
  GmxParked *d_park2 = nullptr;
  uint32_t *d_park2_n = nullptr;
  // test hook (gmx_engine_debug_keep_states, gmx_debug_final_states): the last batch's per-task search results stay readable
  bool keep_states = false;
  uint64_t keep_reads = 0;            // reads of that batch
  std::vector<uint32_t> debug_isa;    // inverse suffix array (text position -> SA index), fetched on first use
  uint32_t fuse = 1;  // fused transitions in the extend kernel's wave loop (GMX_NO_FUSE=1 in the environment: off, for A/B runs)
  // host staging for the _host entry point
  uint8_t *d_reads = nullptr;
  uint64_t *d_offsets = nullptr;
  uint32_t *d_seeds = nullptr;
  uint64_t cap_bases = 0, cap_stage_reads = 0;
  // gmx_map_reads_host, pipelined: two staging slots (device buffers + pinned offsets/seeds), a copy stream
  struct StageSlot {
    uint8_t *d_reads = nullptr;
    uint64_t *d_offsets = nullptr, *h_offsets = nullptr;
    uint32_t *d_seeds = nullptr, *h_seeds = nullptr;
    uint64_t cap_bases = 0, cap_reads = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
  } stage[2];
  hipStream_t copy_stream = nullptr;
  hipStream_t last_stream = nullptr;
  // gmx_map_reads_packed_host: three slots of device buffers for bit planes, offsets, seeds and skip flags; the upload of
  // a chunk (copy stream, straight from the caller's page-locked buffers) runs beside the kernels of the chunks before
  struct PackSlot {
    uint2 *d_planes = nullptr;
    uint64_t *d_offsets = nullptr;
    uint32_t *d_seeds = nullptr;
    uint8_t *d_skip = nullptr;
    uint64_t cap_pairs = 0, cap_reads = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
  } pslot[3];
  uint32_t pslot_next = 0;
  // releases a device buffer obtained from alloc() before the engine is destroyed (superseded staging buffers)
  void release(void *q) {
    if (!q) return;
    for (size_t i = 0; i < allocs.size(); ++i)
      if (allocs[i] == q) {
        allocs[i] = allocs.back();
        allocs.pop_back();
        (void)hipFree(q);
        return;
      }
  }
  // optional HIP-event timing of the kernels (bench.py roofline leg)
  bool timing = false;
  struct EvTriple { hipEvent_t s, a, b, c; uint64_t reads; hipEvent_t k[GMX_TIMED_KERNELS][2]; uint32_t timed; };
  std::vector<EvTriple> pending;
  double search_ms = 0, cover_ms = 0;
  uint64_t search_launches = 0, cover_launches = 0, timed_reads = 0;
  double kernel_ms[GMX_TIMED_KERNELS] = {0};        // gmx_timing::kernel_ms (GMX_TK_*)
  uint64_t kernel_launches[GMX_TIMED_KERNELS] = {0};

  template <class T>
  int alloc(T **p, size_t count, bool zero) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    if (zero) HIP_TRY(hipMemset(q, 0, bytes));
    allocs.push_back(q);
    *p = (T *)q;
    return GMX_OK;
  }
  template <class T, class A>
  int upload(const T **dst, const std::vector<T, A> &src) {
    T *q = nullptr;
    int rc = alloc(&q, src.size(), false);
    if (rc) return rc;
    if (!src.empty()) HIP_TRY(hipMemcpy(q, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    index_bytes += src.size() * sizeof(T);
    *dst = q;
    return GMX_OK;
  }
  int upload(const uint32_t **dst, const gmx::WordBuf &src) {
    uint32_t *q = nullptr;
    int rc = alloc(&q, src.size(), false);
    if (rc) return rc;
    const size_t piece = (size_t)1 << 28;  // (pageable memory, tens of GB at whole-genome scale: 1 GB per staged copy)
    for (size_t at = 0; at < src.size(); at += piece)
      HIP_TRY(hipMemcpy(q + at, src.data() + at, std::min(piece, src.size() - at) * sizeof(uint32_t), hipMemcpyHostToDevice));
    index_bytes += src.size() * sizeof(uint32_t);
    *dst = q;
    return GMX_OK;
  }
};

// Host-side wait for an event that costs no core: query, nap, query. hipEventSynchronize — also on an event created with
// hipEventBlockingSync — kept the calling thread AND a thread of the runtime at 100 % of a core each on the GPU boxes
// (tools/exp/host_call_cost.py: 1500 back-to-back calls, 1.03 s of wall time, 1.03 s of CPU in each of the two threads), so a
// feeder that runs ahead of its GPU cost two cores: eight of them, sixteen — the whole container. The nap (50 us) is far
// below a batch (0.4-3 ms) and three batches are in flight per engine, so the GPU never waits for the host's wake-up.
// GMX_WAIT_SPIN=1: hipEventSynchronize as before (A/B runs).
static hipError_t gmx_event_wait(hipEvent_t ev) {
  static const bool spin = getenv("GMX_WAIT_SPIN") != nullptr;
  if (spin) return hipEventSynchronize(ev);
  // (the first 60 us by querying alone: an event about to complete — the end of a job, the last of several streams — is
  //  not paid for with a nap's wake-up latency; a feeder ahead of its GPU waits ~0.5 ms per batch and naps through it)
  const auto t0 = std::chrono::steady_clock::now();
  for (bool napping = false;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady) return q;
    (void)hipGetLastError();  // (hipErrorNotReady is sticky for hipGetLastError)
    if (!napping) {
      napping = std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(60);
      continue;
    }
    struct timespec ts = {0, 50 * 1000};
    nanosleep(&ts, nullptr);
  }
}

// Wait for the engine's own streams WITHOUT spinning: a blocking event per stream (the thread sleeps until the interrupt).
// hipStreamSynchronize / hipDeviceSynchronize poll — one core per waiting thread; a node's eight feeder threads, each ahead
// of its GPU, cost eight cores that way (profiles/round4/feed_x8.txt: 2.5 ns of host CPU per read, 27 cores' worth at
// 8 x 1.34 G reads/s). The callers still issue their hipDeviceSynchronize afterwards: it then returns at once.
static int gmx_quiesce(gmx_engine *e) {
  if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming | hipEventBlockingSync));
  hipStream_t streams[4] = {e->last_stream, e->copy_stream, e->side_stream, e->side2_stream};
  for (int i = 0; i < 4; ++i) {
    if (i > 0 && !streams[i]) continue;  // ([0]: the null stream counts)
    bool seen = false;
    for (int j = 0; j < i; ++j) seen = seen || streams[j] == streams[i];
    if (seen) continue;
    HIP_TRY(hipEventRecord(e->ev_wait, streams[i]));
    HIP_TRY(gmx_event_wait(e->ev_wait));
  }
  return GMX_OK;
}

static int flush_reset(gmx_engine *e) {
  if (!e->reset_pending) return GMX_OK;
  e->reset_pending = false;
  HIP_TRY(hipMemsetAsync(e->d_fused, 0, (e->n_fused + 32) * 4, e->reset_stream));
  return GMX_OK;
}

// Grouped log -> host. Waits for the device, adds the log's records to e->log_counts when more than `keep_below` words are in
// use (and empties the device log), and notes how full it is. Records: [site_index, n_ids, ids...], each worth +1;
// GMX_LOG_PAD words are padding (CoverLogPart::log_reserve).
static int log_settle(gmx_engine *e);
static int gmx_log_drain(gmx_engine *e, uint64_t keep_below) {
  int frc = flush_reset(e);
  if (frc) return frc;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t used = 0;
  HIP_TRY(hipMemcpy(&used, e->d_log_cursor, 4, hipMemcpyDeviceToHost));
  used = std::min(used, e->log_cap);
  e->log_reads_since = 0;
  e->log_known = used;
  if (used <= keep_below) return GMX_OK;
  std::vector<uint32_t> w(used);
  HIP_TRY(hipMemcpy(w.data(), e->d_log, (size_t)used * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> key;
  for (size_t i = 0; i < w.size();) {
    if (w[i] == GMX_LOG_PAD) {
      ++i;
      continue;
    }
    if (i + 2 > w.size() || i + 2 + (size_t)w[i + 1] > w.size()) {  // reservations are exact (log_reserve): never expected
      gmx_set_error("grouped-allele-count log: malformed record at word " + std::to_string(i) + " of " + std::to_string(w.size()));
      return GMX_EREF;
    }
    key.assign(1, w[i]);
    key.insert(key.end(), w.begin() + i + 2, w.begin() + i + 2 + w[i + 1]);
    e->log_counts[key] += 1;
    i += 2 + w[i + 1];
  }
  HIP_TRY(hipMemset(e->d_log_cursor, 0, 4));
  e->log_known = 0;
  return GMX_OK;
}

// A coverage instance with its scratch in LDS: one wave per block, as many blocks per CU as scratch copies fit its LDS.
template <class Env, int LIST>
static void launch_cover_lds(gmx_engine *e, hipStream_t stream, const BatchView &b, const SearchOut &o, const CoverAcc &acc,
                             bool after_coop = false) {
  static const uint32_t lanes_env = getenv("GMX_COVER_LANES") ? (uint32_t)atoi(getenv("GMX_COVER_LANES")) : 0u;
  const uint32_t lanes = lanes_env ? std::min(lanes_env, gmx_cover_lds_lanes<Env>()) : gmx_cover_lds_lanes<Env>();
  const size_t lds = (size_t)GmxScratchFixed<Env>::total * lanes * sizeof(uint32_t);
  const uint32_t per_cu = std::min<uint32_t>((uint32_t)(160 * 1024 / lds), 32u);
  hipLaunchKernelGGL((gmx_cover_kernel<Env, LIST>), dim3(e->n_cus * per_cu), dim3(64), lds, stream, e->dview, b, o, e->big,
                     acc, lanes, after_coop ? 1u : 0u);
}

template <int LIST>
static void launch_cover_coop(gmx_engine *e, hipStream_t stream, const BatchView &b, const SearchOut &o, const CoverAcc &acc) {
  const size_t lds = (size_t)gmx_coop_lds_words<LIST>() * sizeof(uint32_t);
  const uint32_t per_cu = std::min<uint32_t>((uint32_t)(160 * 1024 / lds), 16u);
  hipLaunchKernelGGL((gmx_cover_coop_kernel<LIST>), dim3(e->n_cus * per_cu), dim3(64), lds, stream, e->dview, b, o, e->big, acc);
}

extern "C" {

void gmx_engine_default_opts(gmx_engine_opts *o) {
  o->device = 0;
  o->rng_mode = GMX_RNG_LEMIRE;
  o->max_states = 1024;
  o->max_path_nodes = 2048;
  o->max_batch_reads = 4u << 20;
  o->forward_only = 0;
  o->huge_heap_bytes = 512ull << 20;
  o->log_cap_words = 0;
}

static int ensure_batch_capacity(gmx_engine *e, uint64_t n_reads) {
  if (n_reads <= e->cap_reads) return GMX_OK;
  // (re)allocate: old buffers stay in `allocs` until destroy; growth is rare (first call sizes it)
  uint64_t cap = std::max<uint64_t>(n_reads, 1024);
  uint64_t n_tasks = cap * 2;
  int rc;
  if ((rc = e->alloc(&e->d_skip, cap, true))) return rc;
  if ((rc = e->alloc(&e->d_status, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_n_final, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_cover_recs, n_tasks * GMX_REGIONS, false))) return rc;
  if ((rc = e->alloc(&e->d_mapped, n_tasks * GMX_REGIONS, false))) return rc;
  if ((rc = e->alloc(&e->d_task_lists, (size_t)GMX_TL_N * n_tasks, false))) return rc;
  e->d_overflow = e->d_task_lists + (size_t)GMX_TL_OVERFLOW * n_tasks;
  e->d_overflow2 = e->d_task_lists + (size_t)GMX_TL_OVERFLOW2 * n_tasks;
  e->d_alive = e->d_task_lists + (size_t)GMX_TL_ALIVE * n_tasks;
  e->d_dead = e->d_task_lists + (size_t)GMX_TL_DEAD * n_tasks;
  e->d_dead2 = e->d_task_lists + (size_t)GMX_TL_DEAD2 * n_tasks;
  e->d_cover_general = e->d_task_lists + (size_t)GMX_TL_GENERAL * n_tasks;
  if ((rc = e->alloc(&e->d_cover_overflow, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_general_rest, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_single_rest, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_park2, (size_t)n_tasks * GMX_STACK_DEPTH, false))) return rc;
  if ((rc = e->alloc(&e->d_park2_n, n_tasks, false))) return rc;
  if (e->log_sites)
    for (int side = 0; side < 2; ++side) {
      if ((rc = e->alloc(&e->d_log_retry[side], n_tasks, false))) return rc;
      if ((rc = e->alloc(&e->d_log_retry_recs[side], n_tasks, false))) return rc;
      if ((rc = e->alloc(&e->d_log_retry_huge[side], n_tasks, false))) return rc;
    }
  if ((rc = e->alloc(&e->d_cover_mid, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_seed_cursor, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_finals, n_tasks * GMX_FAST_STATES, false))) return rc;
  if ((rc = e->alloc(&e->d_arena, n_tasks * GMX_FAST_ARENA, false))) return rc;
  if ((rc = e->alloc(&e->d_alive_seed, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_huge, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_cover_huge, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_huge_retry, 2 * n_tasks, false))) return rc;
  // large-capacity pass: one slot (~60 KB of pools at the default capacities) per task it may have to take; a 1 M-read
  // batch with 5 % of the genome in 10-copy repeats sends 59 k of its 2 M tasks there
  e->big.max_slots = (uint32_t)std::min<uint64_t>(n_tasks, std::min<uint64_t>(std::max<uint64_t>(n_tasks / 16, 4096), 262144));
  if ((rc = e->alloc(&e->big.states, (size_t)e->big.max_slots * e->big.max_states, false))) return rc;
  if ((rc = e->alloc(&e->big.stack, (size_t)e->big.max_slots * e->big.max_states * GMX_STACK_WORDS, false))) return rc;
  if ((rc = e->alloc(&e->big.arena, (size_t)e->big.max_slots * e->big.max_path_nodes, false))) return rc;
  if ((rc = e->alloc(&e->big.n_final, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->big.task_of_slot, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_big_mapped, e->big.max_slots, false))) return rc;
  e->inst_cap = (uint32_t)std::min<uint64_t>(n_tasks, 1u << 23);  // instance lanes of reads in short repeats (320 B of pools each)
  if ((rc = e->alloc(&e->d_inst_list, e->inst_cap, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_sa, e->inst_cap, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_remaining, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_mapped, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_first, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_width, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_serial, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_general_serial, n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_big_serial, e->big.max_slots, false))) return rc;
  if ((rc = e->alloc(&e->d_overflow3, 2 * n_tasks, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_arena, (size_t)e->inst_cap * GMX_FAST_ARENA, false))) return rc;
  if ((rc = e->alloc(&e->d_inst_states, (size_t)e->inst_cap * GMX_INST_STATES, false))) return rc;
  e->cap_reads = cap;
  return GMX_OK;
}

// The device copy of an index is shared by the engines made of it on one device (round 5): several engines per GPU keep batches
// in flight side by side — a nested PRG's batch is a 0.4 ms burst and then 2 ms of a few straggler tasks on a handful of CUs —
// and the second one must not cost a second upload and a second copy in HBM. Reference counted; GMX_NO_INDEX_SHARE=1: off.
struct GmxDeviceIndex {
  uint64_t serial = 0;  // gmx_index_serial of the index it was uploaded from
  int device = 0;
  GmxIndexView view{};
  std::vector<void *> allocs;
  uint64_t bytes = 0;
  int refs = 0;
};
static std::mutex g_dev_index_mu;
static std::vector<GmxDeviceIndex *> g_dev_indexes;
static void gmx_dev_index_release(GmxDeviceIndex *d) {
  if (!d) return;
  std::lock_guard<std::mutex> lk(g_dev_index_mu);
  if (--d->refs > 0) return;
  for (void *p : d->allocs) (void)hipFree(p);
  g_dev_indexes.erase(std::remove(g_dev_indexes.begin(), g_dev_indexes.end(), d), g_dev_indexes.end());
  delete d;
}

int gmx_engine_create(const gmx_index *ixh, const gmx_engine_opts *opts_in, gmx_engine **out) {
  if (!ixh || !out) {
    gmx_set_error("gmx_engine_create: null argument");
    return GMX_EINVAL;
  }
  gmx_engine_opts opts;
  if (opts_in)
    opts = *opts_in;
  else
    gmx_engine_default_opts(&opts);
  if (opts.max_states == 0) opts.max_states = 1024;
  if (opts.max_path_nodes == 0) opts.max_path_nodes = 2048;
  if (opts.max_batch_reads == 0) opts.max_batch_reads = 4u << 20;
  if (opts.huge_heap_bytes == 0) opts.huge_heap_bytes = 512ull << 20;
  if (const char *hb = getenv("GMX_HUGE_HEAP_BYTES")) opts.huge_heap_bytes = strtoull(hb, nullptr, 10);
  opts.huge_heap_bytes = std::max<uint64_t>(opts.huge_heap_bytes, 64 * 1024);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    gmx_set_error("no HIP device available: the quasimap engine has no CPU fallback");
    return GMX_ENODEV;
  }
  if (opts.device < 0 || opts.device >= ndev) {
    gmx_set_error("device ordinal out of range");
    return GMX_ENODEV;
  }
  HIP_TRY(hipSetDevice(opts.device));
  const gmx::HostIndex &h = gmx_index_host(ixh);
  if (h.kmer_size == 0) {
    gmx_set_error("the index was built without a k-mer seed table (kmer_size = 0)");
    return GMX_EINVAL;
  }
  gmx_engine *e = new gmx_engine();
  e->opts = opts;
  GmxIndexView v = h.view();
  int rc = 0;
  {
    std::lock_guard<std::mutex> share_lock(g_dev_index_mu);  // (engines of one group are created side by side: the second waits for the first one's upload)
    GmxDeviceIndex *found = nullptr;
    if (!getenv("GMX_NO_INDEX_SHARE"))
      for (GmxDeviceIndex *d : g_dev_indexes)
        if (d->serial == gmx_index_serial(ixh) && d->device == opts.device) found = d;
    if (found) {
      ++found->refs;
      v = found->view;
      e->index_bytes = found->bytes;
      e->shared_index = found;
    } else {
      rc |= e->upload(&v.blocks, h.blocks);
      rc |= e->upload(&v.hits, h.hits);
      rc |= e->upload(&v.hit_perm, h.hit_perm);
      rc |= e->upload(&v.hit_prog, h.hit_prog);
      rc |= e->upload(&v.text, h.text);
      rc |= e->upload(&v.prog, h.prog);
      rc |= e->upload(&v.sa, h.sa);
      rc |= e->upload(&v.pos_node, h.pos_node);
      rc |= e->upload(&v.nodes, h.nodes);
      rc |= e->upload(&v.edges, h.edges);
      rc |= e->upload(&v.sites, h.sites);
      rc |= e->upload(&v.site_geo, h.site_geo);
      rc |= e->upload(&v.seeds, h.seeds);
      if (h.kmer_size2) rc |= e->upload(&v.seeds2, h.seeds2);
      else v.seeds2 = nullptr;
      rc |= e->upload(&v.seed_words, h.seed_words);
      if (!rc) {  // flags in the multi-state entries of the device copies (GMX_SEEDF_*)
        if (((uint64_t)h.seed_words.size() >> h.seed_shift) >= (1u << 30)) {
          gmx_set_error("the seed tables hold more than 2^30 units of multi-state entries");
          rc = GMX_ECAP;
        } else {
          hipLaunchKernelGGL(gmx_seed_mark_kernel, dim3(4096), dim3(256), 0, nullptr, const_cast<GmxSeed *>(v.seeds), (uint64_t)h.seeds.size(), const_cast<uint32_t *>(v.seed_words), v.seed_shift, v.sa, v.text);
          if (h.kmer_size2)
            hipLaunchKernelGGL(gmx_seed_mark_kernel, dim3(4096), dim3(256), 0, nullptr, const_cast<GmxSeed *>(v.seeds2), (uint64_t)h.seeds2.size(), const_cast<uint32_t *>(v.seed_words), v.seed_shift, v.sa, v.text);
          rc |= hipDeviceSynchronize() != hipSuccess;
        }
      }
      rc |= e->upload(&v.kmer_bitmap, h.kmer_bitmap);

      if (!rc) {  // everything allocated so far is the index: it moves to the shared object
        GmxDeviceIndex *d = new GmxDeviceIndex();
        d->serial = gmx_index_serial(ixh);
        d->device = opts.device;
        d->view = v;
        d->allocs = std::move(e->allocs);
        e->allocs.clear();
        d->bytes = e->index_bytes;
        d->refs = 1;
        g_dev_indexes.push_back(d);
        e->shared_index = d;
      }
    }
  }
  e->dview = v;
  e->n_allele = h.n_allele_slots;
  e->n_pb = h.n_pb_slots;
  e->n_grouped = h.n_grouped_slots;
  {  // one contiguous block: a single all-reduce covers the whole coverage (gmx_coverage_device)
    e->n_acc = ((size_t)h.n_acc_slots + 63) / 64 * 64;
    e->n_fused = e->n_acc + 32;
    rc |= e->alloc(&e->d_fused, e->n_fused + 32, true);  // + 16 words of read counters + log cursor
    e->d_limbs = e->d_fused ? e->d_fused + e->n_acc : nullptr;
    e->d_stats = e->d_fused ? reinterpret_cast<unsigned long long *>(e->d_fused + e->n_fused) : nullptr;
    e->d_log_cursor = e->d_fused ? e->d_fused + e->n_fused + 16 : nullptr;
    rc |= e->alloc(&e->d_error, 2, true);
    e->phys_allele = h.phys_allele;
    e->phys_pb = h.phys_pb;
    e->phys_grouped = h.phys_grouped;
    e->hit_fix = h.hit_fix;
  }
  // The grouped log is used only by sites with more alleles than get dense group counters (gmx_index.cpp: 8). Between
  // batches the engine looks at its real fill (log_settle): drained when half full; entries that found it full are redone.
  for (const GmxSite &st : h.sites) e->log_sites = e->log_sites || st.grouped_off == GMX_GROUPED_LOG;
  {  // the lean single-instance coverage kernel where most sites have geometry records (GMX_NO_COVER_JUMP: A/B runs)
    uint64_t n_jump = 0;
    for (const GmxSiteGeo &g : h.site_geo) n_jump += (g.flags & GMX_SITE_JUMP) ? 1u : 0u;
    e->cover_jump = !h.is_nested && 2 * n_jump > h.site_geo.size() && !getenv("GMX_NO_COVER_JUMP");
    // GMX_NO_COVER_JUMP=1 (the escape hatch, and the walk side of tests/test_cover_jump_ab.py): no kernel
    // sees the geometry records, every single-instance read is recorded by the walk as the reference walks it
    if (getenv("GMX_NO_COVER_JUMP")) e->dview.site_geo = nullptr;
  }
  {
    uint64_t cap = opts.log_cap_words ? opts.log_cap_words
                   : e->log_sites     ? (1ull << 26)  // 256 MB; a batch that fills it is settled by drain + replay (log_settle)
                                      : 64;
    e->log_cap = (uint32_t)std::min<uint64_t>(cap, 0xFFFFFF00ull);
  }
  rc |= e->alloc(&e->d_log, e->log_cap, false);
  e->heap_words = opts.huge_heap_bytes / 4 / 64 * 64;
  rc |= e->alloc(&e->d_heap, e->heap_words, false);
  rc |= e->alloc(&e->d_counters, GMX_N_COUNTERS * GMX_CNT_STRIDE, true);
  // large-capacity pass
  e->big.max_states = opts.max_states;
  e->big.max_path_nodes = opts.max_path_nodes;
  e->big.max_slots = 0;  // its pools are sized with the batch (ensure_batch_capacity)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, opts.device) == hipSuccess && prop.multiProcessorCount > 0)
      e->n_cus = (uint32_t)prop.multiProcessorCount;
    const size_t words = h.kmer_bitmap.size();
    if (!getenv("GMX_FORCE_ABSENT_FILTER") && words >= 4 && words % 4 == 0 && words * 4 <= 128 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(gmx_filter_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(words * 4)) == hipSuccess)
      e->filter_lds_words = (uint32_t)words;
    (void)hipGetLastError();
    if (e->filter_lds_words) {  // re-index the presence bitmap: table index (base j from the left in bit pair j) -> planar
      const uint32_t k = h.kmer_size;
      std::vector<uint32_t> planar(words, 0);
      for (uint64_t code = 0; code < (1ull << (2 * k)); ++code) {
        if (!((h.kmer_bitmap[code >> 5] >> (code & 31)) & 1u)) continue;
        uint32_t lo = 0, hi = 0;
        for (uint32_t j = 0; j < k; ++j) {
          const uint32_t base = (uint32_t)(code >> (2 * j)) & 3u;
          lo |= (base & 1u) << j;
          hi |= (base >> 1) << j;
        }
        const uint32_t p = (hi << k) | lo;
        planar[p >> 5] |= 1u << (p & 31);
      }
      rc |= e->upload(&e->d_kmer_planar, planar);
    } else if (!getenv("GMX_NO_ABSENT_FILTER")) {  // a bitmap too large for LDS: few absent k-mers? (whole-genome PRGs)
      const uint64_t n_k = 1ull << (2 * h.kmer_size);
      if (n_k - std::min<uint64_t>(n_k, h.n_seed_kmers_present) <= GMX_ABSENT_MAX) {
        std::vector<uint32_t> absent;
        for (size_t w = 0; w < words && absent.size() <= GMX_ABSENT_MAX; ++w) {
          uint32_t zeros = ~h.kmer_bitmap[w];
          while (zeros) {
            const uint64_t code = (uint64_t)w * 32 + (uint32_t)__builtin_ctz(zeros);
            zeros &= zeros - 1;
            if (code < n_k) absent.push_back((uint32_t)code);
          }
        }
        if (absent.size() <= GMX_ABSENT_MAX) {
          e->n_absent = (uint32_t)absent.size();
          e->use_absent = true;
          if (absent.empty()) absent.push_back(0);
          rc |= e->upload(&e->d_absent, absent);
        }
      }
    }
  }
  if (const char *pi = getenv("GMX_PROBE_ITERS")) e->probe_iters = (uint32_t)std::max(0, atoi(pi));
  if (getenv("GMX_NO_FUSE")) e->fuse = 0;
  if (const char *eb = getenv("GMX_EXTEND_BUDGET")) e->extend_budget = (uint32_t)std::max(0, atoi(eb));
  e->extend_cap = 0u;  // (GMX_EXTEND_CAP: off by default — at configs[2] a cap of 40 iterations sent 45 k tasks per batch to the
                       //  large-capacity route and the step took 6.4 ms instead of 2.5; see profiles/round4/config2_cap_sweep.txt)
  if (const char *ec = getenv("GMX_EXTEND_CAP")) e->extend_cap = (uint32_t)std::max(0, atoi(ec));
  // passes over the stragglers and the iteration budgets of all but the last. ONE pass by default: at configs[2] (nested
  // MSA regions) three passes — budgets 24 and 96 — take 230 + 528 + 494 us where the single pass takes 901: what is left
  // after the first budget is a few tasks with hundreds of general iterations each (~5 us per iteration: dependent fetches
  // of jump programs and path nodes), and packing them into full waves again does not shorten any of them.
  // GMX_EXTEND_PASSES = "b0,b1": three passes, budgets b0 and b1 (experiments).
  e->extend_passes = 1u;
  e->extend_budget2[0] = 24;
  e->extend_budget2[1] = 96;
  if (const char *ep = getenv("GMX_EXTEND_PASSES")) {
    e->extend_passes = 1;
    for (const char *q = ep; *q && e->extend_passes < GMX_EXTRA_PASSES;) {
      e->extend_budget2[e->extend_passes - 1] = (uint32_t)std::max(1l, strtol(q, const_cast<char **>(&q), 10));
      ++e->extend_passes;
      if (*q == ',') ++q; else break;
    }
  }
  if (getenv("GMX_NO_COOP")) e->coop = false;
  // k-mer entries with many states (small k on a large or dense PRG) do not fit the per-lane stack: when they carry
  // more than 10 % of the seed states the kernels take them one state at a time (seed cursor, a few % slower), else
  // the rare large entry goes to the large-capacity pass
  e->seed_cursor = h.n_seed_states_large * 10 > h.n_seed_states;
  if (const char *sc = getenv("GMX_SEED_CURSOR")) e->seed_cursor = atoi(sc) != 0;
  if (e->seed_cursor && !rc && !getenv("GMX_NO_SA_CTX")) {  // left-context word per suffix-array position (GmxIndexView::sa_ctx): + 4 B per symbol
    uint32_t *sc = nullptr;
    if (e->alloc(&sc, h.sa.size(), false) == GMX_OK) {
      hipLaunchKernelGGL(gmx_sa_ctx_kernel, dim3(8192), dim3(256), 0, nullptr, e->dview.sa, e->dview.text, (uint64_t)h.sa.size(), sc);
      if (hipDeviceSynchronize() == hipSuccess) {
        e->dview.sa_ctx = sc;
        e->index_bytes += h.sa.size() * sizeof(uint32_t);
      }
    } else {
      (void)hipGetLastError();  // (no room: the occurrences are screened through the suffix array and the text, as before)
    }
  }
  // (stream priorities for the side streams — the few-task kernels first — were measured in round 4: no difference, the
  //  chains there wait for memory, not for wave slots)
  rc |= hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking) != hipSuccess;
  rc |= hipStreamCreateWithFlags(&e->side2_stream, hipStreamNonBlocking) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_fork2, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_side1, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_filter, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess;
  rc |= hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess;
  e->cover_big_lanes = 64 * 32;
  rc |= e->alloc(&e->d_scratch_big, (size_t)GmxScratchFixed<CoverEnvBig>::total * e->cover_big_lanes, false);
  if (rc) {
    gmx_engine_destroy(e);
    return GMX_EHIP;
  }
  *out = e;
  return GMX_OK;
}

void gmx_engine_destroy(gmx_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->opts.device);
  (void)hipDeviceSynchronize();
  if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
  if (e->side2_stream) (void)hipStreamDestroy(e->side2_stream);
  if (e->ev_fork2) (void)hipEventDestroy(e->ev_fork2);
  if (e->ev_side1) (void)hipEventDestroy(e->ev_side1);
  if (e->ev_filter) (void)hipEventDestroy(e->ev_filter);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_wait) (void)hipEventDestroy(e->ev_wait);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->h_log_state) (void)hipHostFree(e->h_log_state);
  if (e->ev_log_state) (void)hipEventDestroy(e->ev_log_state);
  for (auto &sl : e->pslot) {
    if (sl.copied) (void)hipEventDestroy(sl.copied);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  for (auto &sl : e->stage) {
    if (sl.copied) (void)hipEventDestroy(sl.copied);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
    if (sl.h_seeds) (void)hipHostFree(sl.h_seeds);
  }
  for (void *p : e->allocs) (void)hipFree(p);
  gmx_dev_index_release(e->shared_index);
  delete e;
}

int gmx_engine_reset(gmx_engine *e) {
  HIP_TRY(hipSetDevice(e->opts.device));
  e->reset_pending = false;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(e->d_fused, 0, (e->n_fused + 32) * 4));
  HIP_TRY(hipMemset(e->d_error, 0, 8));
  HIP_TRY(hipMemset(e->d_counters, 0, GMX_N_COUNTERS * GMX_CNT_STRIDE * 4));
  e->log_counts.clear();
  e->log_known = e->log_reads_since = 0;
  e->log_state_pending = false;
  return GMX_OK;
}

int gmx_engine_reset_async(gmx_engine *e, void *hip_stream) {
  HIP_TRY(hipSetDevice(e->opts.device));
  hipStream_t st = (hipStream_t)hip_stream;
  int frc = flush_reset(e);  // (an earlier one still pending, on whatever stream it named)
  if (frc) return frc;
  e->reset_pending = true;
  e->reset_stream = st;
  e->log_counts.clear();  // what earlier batches left in the device log goes with the cursor
  e->log_known = e->log_reads_since = 0;
  e->log_state_pending = false;
  return GMX_OK;
}

static void launch_filter(gmx_engine *e, hipStream_t st, dim3 task_grid, const BatchView &b, const SearchOut &o, int pass,
                          hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
  if (e->filter_lds_words)
    hipExtLaunchKernelGGL(gmx_filter_lds_kernel, dim3(e->n_cus), dim3(GMX_FILTER_LDS_THREADS), e->filter_lds_words * 4,
                          st, t0, t1, 0u, e->dview, b, o, e->d_kmer_planar, e->filter_lds_words, pass);
  else if (e->use_absent)
    hipExtLaunchKernelGGL(gmx_filter_absent_kernel, task_grid, dim3(GMX_BLOCK), 0, st, t0, t1, 0u, e->dview, b, o, e->d_absent, e->n_absent, pass);
  else
    hipExtLaunchKernelGGL(gmx_filter_kernel, task_grid, dim3(GMX_BLOCK), 0, st, t0, t1, 0u, e->dview, b, o, pass);
}

// ---- grouped log: exact accounting between batches (engines whose index has sites with more than 8 alleles) ----------
static int log_state_enqueue(gmx_engine *e, hipStream_t stream) {
  if (!e->h_log_state) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_log_state), 4 * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_log_state, hipEventDisableTiming));
  }
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 0, e->d_log_cursor, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 1, e->d_counters + GMX_CNT_LOG_RETRY * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 2, e->d_counters + GMX_CNT_LOG_RETRY_RECS * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(e->h_log_state + 3, e->d_counters + GMX_CNT_LOG_RETRY_HUGE * GMX_CNT_STRIDE, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(e->ev_log_state, stream));
  e->log_state_pending = true;
  return GMX_OK;
}

// The entries of the last batch that found the log full, again: setup (list lengths where the kernels read them), the
// compact records, then the large-scratch coverage instance, whose last block also serves the last tier.
static int launch_log_replay(gmx_engine *e, hipStream_t stream) {
  const int in = e->log_retry_side, out = in ^ 1;
  SearchOut o = e->last_o;
  o.log_retry_list = e->d_log_retry[out];
  o.log_retry_recs = e->d_log_retry_recs[out];
  o.log_retry_huge = e->d_log_retry_huge[out];
  o.cover_overflow_list = e->d_log_retry[in];  // the queue of gmx_cover_kernel<CoverEnvBig, 1>: this round's entries
  hipLaunchKernelGGL(gmx_log_replay_setup_kernel, dim3(1), dim3(1024), 0, stream, o, e->d_log_retry_huge[in]);
  if (e->dview.is_nested)
    hipLaunchKernelGGL(gmx_cover_single_replay_kernel<true>, dim3(e->n_cus * 4), dim3(GMX_BLOCK), 0, stream, e->dview, e->last_b, o, e->last_acc,
                     e->d_log_retry_recs[in]);
  else
    hipLaunchKernelGGL(gmx_cover_single_replay_kernel<false>, dim3(e->n_cus * 4), dim3(GMX_BLOCK), 0, stream, e->dview, e->last_b, o, e->last_acc,
                     e->d_log_retry_recs[in]);
  hipLaunchKernelGGL((gmx_cover_kernel<CoverEnvBig, 1>), dim3(e->cover_big_lanes / 64), dim3(64), e->last_big_lds, stream, e->dview,
                     e->last_b, o, e->big, e->last_acc, 64u, 0u);
  HIP_TRY(hipGetLastError());
  e->log_retry_side = out;
  e->last_o.log_retry_list = o.log_retry_list;
  e->last_o.log_retry_recs = o.log_retry_recs;
  e->last_o.log_retry_huge = o.log_retry_huge;
  return log_state_enqueue(e, stream);
}

static int log_settle(gmx_engine *e) {
  if (!e->log_state_pending) return GMX_OK;
  uint64_t before = ~0ull;
  for (int round = 0;; ++round) {
    HIP_TRY(hipEventSynchronize(e->ev_log_state));
    e->log_state_pending = false;
    const uint32_t used = std::min(e->h_log_state[0], e->log_cap);
    const uint64_t retries = (uint64_t)e->h_log_state[1] + e->h_log_state[2] + e->h_log_state[3];
    if (retries == 0) {
      if (used > e->log_cap / 2) return gmx_log_drain(e, 0);
      e->log_known = used;
      return GMX_OK;
    }
    if (round > 0 && retries >= before) {  // (every round starts with an empty log: each must get at least one entry through)
      // an emptied log did not hold one task's records: only more memory helps (gmx_engine_sync reports the read)
      HIP_TRY(hipDeviceSynchronize());
      const uint32_t err[2] = {GMX_TASK_LOGFULL, 0};
      HIP_TRY(hipMemcpy(e->d_error, err, 8, hipMemcpyHostToDevice));
      return GMX_OK;
    }
    before = retries;
    int rc = gmx_log_drain(e, 0);
    if (rc) return rc;
    e->log_replays++;
    e->log_replayed_entries += retries;
    if ((rc = launch_log_replay(e, e->last_stream))) return rc;
  }
}

// One batch as the kernels see it: reads as bytes (d_reads + d_offsets: gmx_pack_kernel makes the bit planes) or as bit
// planes already (d_planes; gmx_map_reads_packed_host).
struct BatchInput {
  const uint8_t *d_reads = nullptr;
  const uint64_t *d_offsets = nullptr;  // null with uniform_len
  const uint32_t *d_seeds = nullptr;
  const uint2 *d_planes = nullptr;      // non-null: packed input, no pack kernel
  const uint8_t *d_skip = nullptr;      // packed input: per-read skip flags, or null
  const uint32_t *d_twobit = nullptr;   // non-null: the reads as a 2-bit stream (gmx_map_reads_2bit_host); unpacked into d_packed
  uint32_t twobit_base0 = 0;            // ... whose first base sits at this base index of d_twobit (< 32)
  uint32_t uniform_len = 0;
  uint64_t n_reads = 0, total_bases = 0;
};

// first kernel of a batch whose reads arrive packed: what gmx_pack_kernel does besides packing (queue counters, a queued reset)
__global__ void gmx_batch_begin_kernel(uint32_t *counters, uint32_t *zero, uint32_t zero_words) {
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < GMX_N_COUNTERS * GMX_CNT_STRIDE; i += blockDim.x) counters[i] = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x) zero[i] = 0;
}

// Reads that arrive as a 2-bit stream (include/gmx.h, gmx_pack_reads_2bit: base j of the batch in bits 2j, 2j + 1) -> the bit
// planes the kernels read, in gmx_pack_kernel's layout. One thread per pair of planes (32 bases): three words of the
// stream, funnel-shifted to the pair's first base, even bits -> low plane, odd bits -> high plane.
__device__ __forceinline__ uint32_t gmx_even_bits(unsigned long long x) {  // bits 0, 2, 4, .. 62 of x, compacted
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (uint32_t)x;
}
__global__ void __launch_bounds__(256) gmx_unpack2_kernel(BatchView b, const uint32_t *stream, uint32_t base0, uint2 *packed) {
  const uint32_t ppr_uniform = b.pairs_per_read;
  for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;; t += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t read, pair;
    uint64_t first_base;  // of the read, in the batch's stream
    uint32_t len;
    if (b.uniform_len) {
      read = (uint32_t)(t / ppr_uniform);
      if (read >= b.n_reads) break;
      pair = (uint32_t)(t - (uint64_t)read * ppr_uniform);
      first_base = (uint64_t)read * b.uniform_len;
      len = b.uniform_len;
    } else {  // ragged: one thread per read walks its pairs (the plane layout needs the offsets anyway)
      read = (uint32_t)t;
      if (read >= b.n_reads) break;
      pair = 0;
      first_base = b.offsets[read] - b.offsets[0];
      len = (uint32_t)(b.offsets[read + 1] - b.offsets[read]);
    }
    uint2 *out = packed + pack_off(b, read);
    const uint32_t n_pairs = b.uniform_len ? pair + 1 : (len + 31u) / 32u;
    for (uint32_t p = pair; p < n_pairs; ++p) {
      const uint64_t j = base0 + first_base + 32ull * p;  // base index in the stream of the pair's first base
      const uint64_t w = j >> 4;                         // 16 bases per word
      const uint32_t sh = (uint32_t)(j & 15u) * 2u;
      const uint32_t w0 = stream[w], w1 = stream[w + 1], w2 = stream[w + 2];
      const unsigned long long bits = (unsigned long long)__builtin_amdgcn_alignbit(w1, w0, sh) |
                                      ((unsigned long long)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
      out[p] = make_uint2(gmx_even_bits(bits), gmx_even_bits(bits >> 1));
    }
  }
}

static int launch_batch(gmx_engine *e, const BatchInput &in, hipStream_t stream) {
  const uint64_t n_reads = in.n_reads, total_bases = in.total_bases;
  if (n_reads == 0) return GMX_OK;
  if (n_reads > (0x7fffffffull / GMX_FAST_ARENA) / 2) {  // path-node handles (offsets into the arena table) stay below 2^31
    gmx_set_error("batch too large: at most 44 M reads per launch (lower gmx_engine_opts.max_batch_reads)");
    return GMX_EINVAL;
  }
  // the batch before: redo what found the log full, drain when half full. FIRST: a replay reads that batch's queues and
  // retry lists, which a growing workspace (ensure_batch_capacity) replaces with fresh, uninitialised buffers.
  int rc = e->log_sites ? log_settle(e) : GMX_OK;
  if (rc) return rc;
  if ((rc = ensure_batch_capacity(e, n_reads))) return rc;
  const bool fold_reset = e->reset_pending && e->reset_stream == stream;
  if (e->reset_pending && !fold_reset && (rc = flush_reset(e))) return rc;
  e->reset_pending = false;
  if (!in.d_planes) {
    uint64_t need = total_bases / 32 + n_reads + 16;  // pairs; the slack covers the one-pair look-ahead of planes()
    if (need > e->cap_packed) {
      rc = e->alloc(&e->d_packed, need, false);
      if (rc) return rc;
      e->cap_packed = need;
    }
  }
  BatchView b{in.d_reads, in.d_offsets, in.d_seeds, (in.d_planes || in.d_twobit) ? in.d_skip : e->d_skip, in.d_planes ? in.d_planes : e->d_packed,
              (uint32_t)n_reads, (uint32_t)(e->opts.forward_only ? 1 : 0), in.uniform_len, (in.uniform_len + 31u) / 32u,
              e->keep_states ? 1u : 0u};
  const uint32_t region_inv = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (((uint64_t)GMX_REGIONS << 32) + e->dview.n_prg - 1) / std::max<uint32_t>(e->dview.n_prg, 1u));
  SearchOut o{};  // (member by member: the struct's order is not part of any contract)
  o.status = e->d_status;
  o.n_final = e->d_n_final;
  o.finals = e->d_finals;
  o.arena = e->d_arena;
  o.cover_recs = e->d_cover_recs;
  o.cover_rec_task = e->d_mapped;
  o.region_cap = (uint32_t)(e->cap_reads * 2);
  o.region_inv = region_inv;
  o.task_lists = e->d_task_lists;
  o.list_stride = (uint32_t)(e->cap_reads * 2);
  o.overflow_list = e->d_overflow;
  o.overflow2_list = e->d_overflow2;
  o.cover_overflow_list = e->d_cover_overflow;
  o.big_mapped_list = e->d_big_mapped;
  o.cover_mid_list = e->d_cover_mid;
  o.cover_general_list = e->d_cover_general;
  o.alive_list = e->d_alive;
  o.dead_list = e->d_dead;
  o.dead2_list = e->d_dead2;
  o.seed_cursor = e->d_seed_cursor;
  o.error = e->d_error;
  o.counters = e->d_counters;
  o.alive_seed = e->d_alive_seed;
  o.huge_list = e->d_huge;
  o.cover_huge_list = e->d_cover_huge;
  o.huge_retry = e->d_huge_retry;
  o.arena_stride = (uint32_t)(e->cap_reads * 2);
  o.inst_list = e->d_inst_list;
  o.inst_sa = e->d_inst_sa;
  o.inst_remaining = e->d_inst_remaining;
  o.inst_cap = e->inst_cap;
  o.inst_slots = !getenv("GMX_NO_INST") ? e->big.max_slots : 0u;
  o.slot_n_final = e->big.n_final;
  o.slot_task = e->big.task_of_slot;
  o.inst_mapped_list = e->d_inst_mapped;
  o.inst_arena = e->d_inst_arena;
  o.inst_states = e->d_inst_states;
  o.inst_first = e->d_inst_first;
  o.inst_remaining_width = e->d_inst_width;
  o.inst_serial_list = e->d_inst_serial;
  o.general_serial_list = e->d_general_serial;
  o.big_serial_list = e->d_big_serial;
  o.overflow3_list = e->d_overflow3;
  o.split_twice = getenv("GMX_NO_SPLIT2") ? 0u : 1u;
  o.general_rest_list = e->d_general_rest;
  o.single_rest_list = e->d_single_rest;
  o.park2 = e->d_park2;
  o.park2_n = e->d_park2_n;
  o.log_retry_list = e->d_log_retry[e->log_retry_side];
  o.log_retry_recs = e->d_log_retry_recs[e->log_retry_side];
  o.log_retry_huge = e->d_log_retry_huge[e->log_retry_side];
  o.stats = e->d_stats;
  uint32_t n_tasks = (uint32_t)n_reads * 2;
  if (e->keep_states) {  // test hook: a task that never reaches a kernel that writes its state count reads as "no state"
    HIP_TRY(hipMemsetAsync(e->d_n_final, 0, (size_t)n_tasks * sizeof(uint32_t), stream));
    e->keep_reads = n_reads;
  }
  gmx_engine::EvTriple ev{};
  if (e->timing) {
    HIP_TRY(hipEventCreate(&ev.s));
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventCreate(&ev.c));
    for (int k = 0; k < GMX_TK_N; ++k) {
      HIP_TRY(hipEventCreate(&ev.k[k][0]));
      HIP_TRY(hipEventCreate(&ev.k[k][1]));
    }
    ev.reads = n_reads;
    HIP_TRY(hipEventRecord(ev.s, stream));
  }
  // (timing leg: events attached to the dispatches themselves — their own start and end, as a kernel trace sees them)
  auto t_ev = [&](int k, int side) -> hipEvent_t {
    if (!e->timing) return nullptr;
    ev.timed |= 1u << k;
    return ev.k[k][side];
  };
  if (in.d_planes || in.d_twobit) {
    hipLaunchKernelGGL(gmx_batch_begin_kernel, dim3(fold_reset ? 256 : 1), dim3(1024), 0, stream, e->d_counters,
                       fold_reset ? e->d_fused : nullptr, fold_reset ? (uint32_t)(e->n_fused + 32) : 0u);
    if (in.d_twobit) {
      const uint64_t threads = in.uniform_len ? n_reads * ((in.uniform_len + 31u) / 32u) : n_reads;
      hipExtLaunchKernelGGL(gmx_unpack2_kernel, dim3((unsigned)std::min<uint64_t>((threads + 255) / 256, 1u << 20)), dim3(256), 0, stream,
                            t_ev(GMX_TK_UNPACK, 0), t_ev(GMX_TK_UNPACK, 1), 0u, b, in.d_twobit, in.twobit_base0, e->d_packed);
    }
  } else
    hipLaunchKernelGGL(gmx_pack_kernel, dim3((unsigned)((n_reads + GMX_PACK_READS - 1) / GMX_PACK_READS)), dim3(GMX_PACK_THREADS), 0, stream, b,
                       e->d_skip, e->d_packed, e->d_counters, fold_reset ? e->d_fused : nullptr, fold_reset ? (uint32_t)(e->n_fused + 32) : 0u);
  size_t lds = (size_t)GMX_STACK_DEPTH * GMX_STACK_WORDS * GMX_BLOCK * sizeof(uint32_t);
  const size_t big_lds = (size_t)GMX_BIG_LDS_DEPTH * GMX_STACK_WORDS * 64 * sizeof(uint32_t);
  dim3 task_grid((n_tasks + GMX_BLOCK - 1) / GMX_BLOCK);
  const bool seeded = e->dview.kmer_size2 != 0 && !getenv("GMX_NO_SEEDED");  // longer seed table: no probe phase (gmx_seed_kernel)
  if (seeded)
    hipExtLaunchKernelGGL(gmx_seed_kernel, dim3((n_tasks + GMX_SEED_THREADS * GMX_SEED_CHUNKS - 1) / (GMX_SEED_THREADS * GMX_SEED_CHUNKS)), dim3(GMX_SEED_THREADS), 0, stream,
                          t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o);
  else if (e->seed_cursor)
    hipExtLaunchKernelGGL(gmx_probe_kernel<true>, task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o,
                          e->probe_iters);
  else
    hipExtLaunchKernelGGL(gmx_probe_kernel<false>, task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, t_ev(GMX_TK_SEED, 0), t_ev(GMX_TK_SEED, 1), 0u, e->dview, b, o,
                          e->probe_iters);
  // fork 1: the probe kernel's overflow queue (few, long-running tasks) is served by the large-capacity kernel on a
  // side stream while the extend kernel runs, and so is the k-mer filter of the tasks the probe kernel found dead
  // (most reverse-complement tasks; the extend kernel queues its own dead tasks separately)
  HIP_TRY(hipEventRecord(e->ev_fork, stream));
  HIP_TRY(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
  CoverAcc acc{e->d_fused, e->d_log, e->d_log_cursor, e->log_cap, e->d_scratch_big, e->cover_big_lanes, e->opts.rng_mode,
               e->log_sites ? 1u : 0u, e->d_heap, e->heap_words, e->d_status, (uint32_t)n_reads * 2u, e->d_stats};
  if (seeded) {  // what gmx_seed_kernel sent to the large-capacity pass (reads in repeats), and its coverage: on side 2
    HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_fork, 0));
    const InstPools pools{0};
    // one lane per mapping instance of the reads in short repeats, then their coverage. (On a stream of its own this pair
    // gained nothing: the runtime then put two of the four streams on one hardware queue, and filter and extend kernel
    // ran one after the other.)
    hipLaunchKernelGGL(gmx_extend_inst_kernel, dim3(e->n_cus * 2), dim3(GMX_BLOCK), lds, e->side2_stream, e->dview, b, o, pools);
    if (e->coop) launch_cover_coop<5>(e, e->side2_stream, b, o, acc);
    launch_cover_lds<CoverEnvMid, 5>(e, e->side2_stream, b, o, acc, e->coop);
    hipLaunchKernelGGL(gmx_search_split_kernel, dim3(4096), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, 0);
    launch_cover_lds<CoverEnvMid, 4>(e, e->side2_stream, b, o, acc);
  } else {
    hipLaunchKernelGGL(gmx_search_big_kernel, dim3(1024), dim3(64), big_lds, e->side_stream, e->dview, b, o, e->big, 0);
  }
  HIP_TRY(hipEventRecord(e->ev_side1, e->side_stream));
  launch_filter(e, e->side_stream, task_grid, b, o, 0, t_ev(GMX_TK_FILTER0, 0), t_ev(GMX_TK_FILTER0, 1));
  // (timing leg: the events are attached to this very dispatch — its own start and end, as a kernel trace sees them —
  // instead of being recorded around it, where they add the gap to the kernel before and two barrier packets)
  hipEvent_t k0 = e->timing ? ev.a : nullptr, k1 = e->timing ? ev.b : nullptr;
  const uint32_t budget = e->extend_budget;
  if (seeded && e->seed_cursor)
    hipExtLaunchKernelGGL((gmx_extend_kernel<true, 1>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else if (seeded)
    hipExtLaunchKernelGGL((gmx_extend_kernel<false, 1>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else if (e->seed_cursor)
    hipExtLaunchKernelGGL((gmx_extend_kernel<true, 0>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  else
    hipExtLaunchKernelGGL((gmx_extend_kernel<false, 0>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, k0, k1, 0u, e->dview, b, o, e->fuse, budget, 0u);
  if (budget) {  // the stragglers, compacted (a block that finds its part of the queue empty returns at once)
    for (uint32_t pass = 0; pass < e->extend_passes; ++pass) {
      const bool last = pass + 1 >= e->extend_passes;
      // (the last pass runs to the end, or — nested PRGs — to its cap, beyond which a task goes to the split search)
      const uint32_t budget2 = last ? e->extend_cap : e->extend_budget2[pass];
      const uint32_t pass_arg = pass | (last && e->extend_cap ? 0x80000000u : 0u);
      hipEvent_t p0 = pass == 0 ? t_ev(GMX_TK_EXTEND2, 0) : nullptr, p1 = pass == 0 ? t_ev(GMX_TK_EXTEND2, 1) : nullptr;
      if (e->seed_cursor)
        hipExtLaunchKernelGGL((gmx_extend_kernel<true, 2>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, p0, p1, 0u, e->dview, b, o, e->fuse, budget2, pass_arg);
      else
        hipExtLaunchKernelGGL((gmx_extend_kernel<false, 2>), task_grid, dim3(GMX_BLOCK), (uint32_t)lds, stream, p0, p1, 0u, e->dview, b, o, e->fuse, budget2, pass_arg);
    }
  }
  // fork 2: the extend kernel's overflow queue, then the coverage of everything the large-capacity kernel mapped,
  // beside filter + coverage of the regular tasks
  HIP_TRY(hipEventRecord(e->ev_fork2, stream));
  HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_fork2, 0));
  HIP_TRY(hipStreamWaitEvent(e->side2_stream, e->ev_side1, 0));
  // the second filter pass (the tasks the extend kernel found dead) comes first here: side 1 is busy with the first pass
  // for most of the batch, and behind the few-lane kernels below it would end after the main stream's last kernel
  launch_filter(e, e->side2_stream, task_grid, b, o, 1, t_ev(GMX_TK_FILTER1, 0), t_ev(GMX_TK_FILTER1, 1));
  // the extend kernel's overflow queue (and the tasks whose instances ran out of their pools): the 16-lane split search
  // first, one lane with a whole slot for what that leaves
  static const bool split2 = getenv("GMX_NO_SPLIT2") == nullptr;
  if (split2) hipLaunchKernelGGL(gmx_search_split_kernel, dim3(1024), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, 1);
  hipLaunchKernelGGL(gmx_search_big_kernel, dim3(1024), dim3(64), big_lds, e->side2_stream, e->dview, b, o, e->big, split2 ? 2 : 1);
  if (e->coop) launch_cover_coop<2>(e, e->side2_stream, b, o, acc);
  launch_cover_lds<CoverEnvMid, 2>(e, e->side2_stream, b, o, acc, e->coop);
  // the general instances of the regular tasks: their queue is complete after the extend kernel unless the PRG is
  // nested (there gmx_cover_single_kernel hands tasks over), so they run on side 1, off the main stream (and not behind
  // the large-capacity pass's chain of few-lane kernels: with reads in repeats that chain is the batch's longest path)
  const bool general_on_side = !e->dview.is_nested;
  HIP_TRY(hipStreamWaitEvent(e->side_stream, e->ev_fork2, 0));
  const size_t one_lds = (size_t)4 * GMX_WIDE_LOCI * GMX_ONE_THREADS * sizeof(uint32_t);
  const bool one = !getenv("GMX_NO_COVER_ONE");
  auto launch_one = [&](hipStream_t st) {  // (with GMX_NO_COVER_ONE the kernel only passes its queue on: A/B runs)
    hipLaunchKernelGGL(gmx_cover_one_kernel, dim3(e->n_cus * 4), dim3(GMX_ONE_THREADS), one_lds, st, e->dview, b, o, e->big, acc, one ? 1u : 0u);
  };
  if (general_on_side) {
    launch_one(e->side_stream);
    if (e->coop) launch_cover_coop<3>(e, e->side_stream, b, o, acc);
    launch_cover_lds<CoverEnvLds, 3>(e, e->side_stream, b, o, acc, e->coop);
    launch_cover_lds<CoverEnv, 0>(e, e->side_stream, b, o, acc);
  }
  HIP_TRY(hipEventRecord(e->ev_filter, e->side_stream));
  // (Round 4 measured the records in PRG order — a radix sort of (position, record) pairs in front of this kernel, for the
  //  locality of the accumulator and table lines: at configs[3] the kernel took 508 us instead of 436 plus 120 us of sorting, at
  //  configs[4] 646 instead of 611: neighbouring lanes then hit the SAME accumulator words and their atomics serialise. Dropped.)
  if (e->dview.is_nested) {
    hipExtLaunchKernelGGL(gmx_cover_single_kernel<true>, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), 0, stream, t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u,
                          e->dview, b, o, acc);
  } else if (e->cover_jump) {  // most sites have geometry records: the lean kernel, then the few records it declined
    // (Measured and dropped: this kernel over the records queued by then BESIDE the extend kernel's passes over the stragglers,
    //  those moved to side 1 — at configs[3] the passes then took 335 us instead of 165 and the batch 1.13 ms instead of 1.07:
    //  the two kernels wait for the same thing, the memory system's rate of scattered accesses.)
    hipExtLaunchKernelGGL(gmx_cover_jump_kernel, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), (uint32_t)(GMX_STAGE_MAX * GMX_BLOCK * sizeof(uint32_t)), stream,
                          t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u, e->dview, b, o, acc);
    hipLaunchKernelGGL(gmx_cover_single_rest_kernel, dim3(e->n_cus * 2), dim3(GMX_BLOCK), 0, stream, e->dview, b, o, acc);
  } else {
    hipExtLaunchKernelGGL(gmx_cover_single_kernel<false>, dim3(task_grid.x * GMX_REGIONS), dim3(GMX_BLOCK), 0, stream, t_ev(GMX_TK_SINGLE, 0), t_ev(GMX_TK_SINGLE, 1), 0u,
                          e->dview, b, o, acc);
  }
  // The batch's last coverage instance (1: what exceeded the regular scratch; its last block also serves the last tier,
  // whose search keeps its first pending entries in LDS) needs every other instance done except gmx_cover_single_kernel,
  // which queues nothing on a non-nested PRG: there it runs at the end of side 2, beside that kernel.
  hipStream_t last = general_on_side ? e->side2_stream : stream;
  if (!general_on_side) {
    launch_one(stream);
    if (e->coop) launch_cover_coop<3>(e, stream, b, o, acc);
    launch_cover_lds<CoverEnvLds, 3>(e, stream, b, o, acc, e->coop);
    launch_cover_lds<CoverEnv, 0>(e, stream, b, o, acc);
    HIP_TRY(hipEventRecord(e->ev_join, e->side2_stream));
    HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
  }
  HIP_TRY(hipStreamWaitEvent(last, e->ev_filter, 0));
  hipLaunchKernelGGL((gmx_cover_kernel<CoverEnvBig, 1>), dim3(e->cover_big_lanes / 64), dim3(64), big_lds, last, e->dview,
                     b, o, e->big, acc, 64u, 0u);
  if (general_on_side) {
    HIP_TRY(hipEventRecord(e->ev_join, e->side2_stream));
    HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
  }
  // (no pass over per-task status words: the read counters are added where each task's fate is decided, SearchOut::stats)
  if (e->timing) {
    HIP_TRY(hipEventRecord(ev.c, stream));
    e->pending.push_back(ev);
  }
  HIP_TRY(hipGetLastError());
  e->last_stream = stream;
  if (e->log_sites) {  // what log_settle() looks at before the next batch, and what a replay needs of this one
    e->last_b = b;
    e->last_o = o;
    e->last_acc = acc;
    e->last_big_lds = big_lds;
    if ((rc = log_state_enqueue(e, stream))) return rc;
  }
  return GMX_OK;
}

int gmx_map_reads_device(gmx_engine *e, const uint8_t *d_reads, const uint64_t *d_offsets, const uint32_t *d_seeds,
                         uint64_t n_reads, uint64_t total_bases, void *hip_stream) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  hipStream_t stream = (hipStream_t)hip_stream;
  uint64_t done = 0;
  while (done < n_reads) {
    uint64_t n = std::min<uint64_t>(e->opts.max_batch_reads, n_reads - done);
    BatchInput in;
    in.d_reads = d_reads;
    in.d_offsets = d_offsets + done;
    in.d_seeds = d_seeds + done;
    in.n_reads = n;
    in.total_bases = total_bases;
    int rc = launch_batch(e, in, stream);
    if (rc) return rc;
    done += n;
  }
  // An index with log sites: a batch that found the grouped log full is replayed from ITS inputs (read lengths, seeds), and
  // the caller may reuse its device buffers in stream order after this call: settle now (waits for the batch; engines
  // without log sites — every dense-count index — stay asynchronous).
  if (e->log_sites) return log_settle(e);
  return GMX_OK;
}

// Is [p, p + bytes) page-locked memory the runtime can DMA from asynchronously (gmx_host_alloc, hipHostMalloc, registered)?
static bool gmx_is_pinned(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Large calls: chunks of <= 1 M reads through two staging slots; the upload of a chunk (copy stream, from the caller's
// buffer registered with the runtime for the duration of the call) runs beside the kernels of the one before.
static int map_reads_host_pipelined(gmx_engine *e, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                                    uint64_t n_reads, uint64_t chunk) {
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  const uint64_t first = offsets[0], total = offsets[n_reads] - first;
  const bool registered = !gmx_is_pinned(reads + first) &&
                          hipHostRegister(const_cast<uint8_t *>(reads + first), total, hipHostRegisterDefault) == hipSuccess;
  (void)hipGetLastError();
  int rc = GMX_OK;
  auto hip_ok = [&](hipError_t err, const char *what) {  // (no early return: the epilogue below always runs)
    if (err == hipSuccess) return true;
    gmx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    rc = GMX_EHIP;
    return false;
  };
  uint64_t done = 0;
  for (uint32_t i = 0; done < n_reads && rc == GMX_OK; ++i) {
    gmx_engine::StageSlot &sl = e->stage[i & 1];
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    const uint64_t b0 = offsets[done], bases = offsets[done + n] - b0;
    if (sl.busy) {  // the chunk that used this slot two rounds ago
      if (!hip_ok(gmx_event_wait(sl.done), "hipEventSynchronize")) break;
      sl.busy = false;
    }
    if (!sl.copied) {
      if (!hip_ok(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate") ||
          !hip_ok(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate"))
        break;
    }
    if (bases > sl.cap_bases) {  // (the slot is idle: its superseded buffer can go at once)
      const uint64_t cb = std::max<uint64_t>(bases + bases / 8, 1 << 16);
      e->release(sl.d_reads);
      sl.d_reads = nullptr;
      sl.cap_bases = 0;
      if ((rc = e->alloc(&sl.d_reads, cb + 16, false))) break;
      sl.cap_bases = cb;
    }
    if (n > sl.cap_reads) {
      const uint64_t cr = std::max<uint64_t>(n, 1024);
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, cr + 1, false)) || (rc = e->alloc(&sl.d_seeds, cr, false))) break;
      if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
      if (sl.h_seeds) (void)hipHostFree(sl.h_seeds);
      sl.h_offsets = nullptr;
      sl.h_seeds = nullptr;
      if (!hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sl.h_offsets), (cr + 1) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc") ||
          !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sl.h_seeds), cr * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc"))
        break;
      sl.cap_reads = cr;
    }
    for (uint64_t j = 0; j <= n; ++j) sl.h_offsets[j] = offsets[done + j] - b0;
    memcpy(sl.h_seeds, seeds + done, n * sizeof(uint32_t));
    if (!hip_ok(hipMemcpyAsync(sl.d_reads, reads + b0, bases, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(reads)") ||
        !hip_ok(hipMemcpyAsync(sl.d_offsets, sl.h_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(offsets)") ||
        !hip_ok(hipMemcpyAsync(sl.d_seeds, sl.h_seeds, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(seeds)") ||
        !hip_ok(hipEventRecord(sl.copied, e->copy_stream), "hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(nullptr, sl.copied, 0), "hipStreamWaitEvent"))
      break;
    {
      BatchInput in;
      in.d_reads = sl.d_reads;
      in.d_offsets = sl.d_offsets;
      in.d_seeds = sl.d_seeds;
      in.n_reads = n;
      in.total_bases = bases;
      rc = launch_batch(e, in, nullptr);
    }
    if (rc) break;
    if (!hip_ok(hipEventRecord(sl.done, nullptr), "hipEventRecord")) break;
    sl.busy = true;
    done += n;
  }
  // common epilogue, error or not: nothing in flight reads the caller's buffer, the slots are idle, the buffer is unregistered
  (void)hipStreamSynchronize(e->copy_stream);
  (void)hipDeviceSynchronize();
  e->stage[0].busy = e->stage[1].busy = false;
  if (registered) (void)hipHostUnregister(const_cast<uint8_t *>(reads + first));
  (void)hipGetLastError();
  return rc ? rc : gmx_engine_sync(e);
}

static uint64_t gmx_feed_chunk(const gmx_engine *e);  // reads per launch of the host feeds (below)
int gmx_map_reads_host(gmx_engine *e, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                       uint64_t n_reads) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  {
    const uint64_t chunk = gmx_feed_chunk(e);
    if (n_reads > chunk && !getenv("GMX_HOST_SERIAL")) return map_reads_host_pipelined(e, reads, offsets, seeds, n_reads, chunk);
  }
  uint64_t done = 0;
  while (done < n_reads) {
    uint64_t n = std::min<uint64_t>(e->opts.max_batch_reads, n_reads - done);
    uint64_t b0 = offsets[done], b1 = offsets[done + n];
    uint64_t bases = b1 - b0;
    if (bases > e->cap_bases) {
      uint64_t cb = std::max<uint64_t>(bases, 1 << 16);
      int rc = e->alloc(&e->d_reads, cb + 16, false);
      if (rc) return rc;
      e->cap_bases = cb;
    }
    if (n > e->cap_stage_reads) {
      uint64_t cr = std::max<uint64_t>(n, 1024);
      int rc = e->alloc(&e->d_offsets, cr + 1, false);
      if (rc) return rc;
      rc = e->alloc(&e->d_seeds, cr, false);
      if (rc) return rc;
      e->cap_stage_reads = cr;
    }
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t i = 0; i <= n; ++i) rel[i] = offsets[done + i] - b0;
    HIP_TRY(hipMemcpy(e->d_reads, reads + b0, bases, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_offsets, rel.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_seeds, seeds + done, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    BatchInput in;
    in.d_reads = e->d_reads;
    in.d_offsets = e->d_offsets;
    in.d_seeds = e->d_seeds;
    in.n_reads = n;
    in.total_bases = bases;
    int rc = launch_batch(e, in, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));  // staging buffers are reused by the next batch
    if (e->log_sites && (rc = log_settle(e))) return rc;  // ... and a replay of this batch reads them: before they are overwritten
    done += n;
  }
  return gmx_engine_sync(e);
}

// planes: the bit planes (twobit = false) or the 2-bit stream as 32-bit words (twobit = true; gmx_map_reads_2bit_host)
static int map_reads_packed_impl(gmx_engine *e, const uint64_t *planes, bool twobit, const uint64_t *offsets, uint32_t uniform_len,
                                 const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads) {
  if (!e || !planes || !seeds || (!offsets && !uniform_len)) {
    gmx_set_error("gmx_map_reads_packed_host / gmx_map_reads_2bit_host: null argument (offsets may be null only with uniform_len)");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  const uint32_t ppr = (uniform_len + 31u) / 32u;
  auto base_at = [&](uint64_t r) -> uint64_t { return uniform_len ? r * uniform_len : offsets[r] - offsets[0]; };
  auto pair_at = [&](uint64_t r) -> uint64_t {  // 8-byte units from the call's first read to read r (gmx.h: layout of `planes`;
    if (twobit) return (base_at(r) + 31) >> 5;  //  a 2-bit stream: 32 bases per unit, rounded up)
    return uniform_len ? r * ppr : ((offsets[r] >> 5) - (offsets[0] >> 5)) + r;
  };
  // buffers the runtime cannot DMA from are registered for the duration of the call, which then waits for its uploads
  struct Reg { const void *p; bool on; };
  Reg regs[4] = {{planes, false}, {offsets, false}, {seeds, false}, {skip, false}};
  const uint64_t reg_bytes[4] = {pair_at(n_reads) * 8, (n_reads + 1) * 8, n_reads * 4, n_reads};
  bool all_pinned = true;
  for (int i = 0; i < 4; ++i) {
    if (!regs[i].p || gmx_is_pinned(regs[i].p)) continue;
    regs[i].on = hipHostRegister(const_cast<void *>(regs[i].p), reg_bytes[i], hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    all_pinned = false;
  }
  // seeds in place (gmx_engine_seeds_in_place): the kernels read the few seeds they need — a read draws only when it has
  // several equally good mapping classes — from the caller's page-locked buffer over PCIe; nothing is uploaded
  const uint32_t *d_seeds_host = nullptr;
  if (e->seeds_in_place && gmx_is_pinned(seeds)) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<uint32_t *>(seeds), 0) == hipSuccess && dp) d_seeds_host = static_cast<const uint32_t *>(dp);
    else (void)hipGetLastError();
  }
  const uint64_t chunk = gmx_feed_chunk(e);
  int rc = GMX_OK;
  auto hip_ok = [&](hipError_t err, const char *what) {
    if (err == hipSuccess) return true;
    gmx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    rc = GMX_EHIP;
    return false;
  };
  for (uint64_t done = 0; done < n_reads && rc == GMX_OK;) {
    gmx_engine::PackSlot &sl = e->pslot[e->pslot_next];
    e->pslot_next = (e->pslot_next + 1) % 3;
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    uint64_t p0 = pair_at(done), pairs = pair_at(done + n) - p0;
    uint32_t twobit_base0 = 0;
    if (twobit) {  // the chunk's bases from the 8-byte unit holding its first one
      const uint64_t b0 = base_at(done), b1 = base_at(done + n);
      p0 = b0 >> 5;
      pairs = ((b1 + 31) >> 5) - p0;
      twobit_base0 = (uint32_t)(b0 & 31u);
    }
    if (sl.busy) {  // the batch that used this slot three chunks ago
      if (!hip_ok(gmx_event_wait(sl.done), "hipEventSynchronize")) break;
      sl.busy = false;
    }
    if (!sl.copied) {
      if (!hip_ok(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate") ||
          !hip_ok(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate"))
        break;
    }
    if (pairs + 16 > sl.cap_pairs) {  // (+ slack: the kernels fetch whole 16-byte pieces and one pair ahead)
      const uint64_t cp = pairs + pairs / 8 + 64;
      e->release(sl.d_planes);
      sl.d_planes = nullptr;
      sl.cap_pairs = 0;
      if ((rc = e->alloc(&sl.d_planes, cp, false))) break;
      sl.cap_pairs = cp;
    }
    if (n > sl.cap_reads) {
      const uint64_t cr = std::max<uint64_t>(n, 1024);
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      e->release(sl.d_skip);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.d_skip = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, cr + 1, false)) || (rc = e->alloc(&sl.d_seeds, cr, false)) || (rc = e->alloc(&sl.d_skip, cr, false))) break;
      sl.cap_reads = cr;
    }
    // (one copy stream: the planes split over two streams reach 31-37 GB/s instead of 51, and a kernel pulling the stream
    //  out of the caller's page-locked memory itself 34 GB/s — both measured in round 3 and removed)
    if (!hip_ok(hipMemcpyAsync(sl.d_planes, planes + p0, pairs * 8, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(planes)")) break;
    if (!uniform_len &&
        !hip_ok(hipMemcpyAsync(sl.d_offsets, offsets + done, (n + 1) * 8, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(offsets)"))
      break;
    if (!d_seeds_host &&
        !hip_ok(hipMemcpyAsync(sl.d_seeds, seeds + done, n * 4, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(seeds)"))
      break;
    if (skip && !hip_ok(hipMemcpyAsync(sl.d_skip, skip + done, n, hipMemcpyHostToDevice, e->copy_stream), "hipMemcpyAsync(skip)")) break;
    if (!hip_ok(hipEventRecord(sl.copied, e->copy_stream), "hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(nullptr, sl.copied, 0), "hipStreamWaitEvent"))
      break;
    BatchInput in;
    in.d_offsets = uniform_len ? nullptr : sl.d_offsets;
    in.d_seeds = d_seeds_host ? d_seeds_host + done : sl.d_seeds;
    in.d_planes = twobit ? nullptr : sl.d_planes;
    in.d_twobit = twobit ? reinterpret_cast<const uint32_t *>(sl.d_planes) : nullptr;
    in.twobit_base0 = twobit_base0;
    in.d_skip = skip ? sl.d_skip : nullptr;
    in.uniform_len = uniform_len;
    in.n_reads = n;
    in.total_bases = uniform_len ? n * (uint64_t)uniform_len : offsets[done + n] - offsets[done];
    if ((rc = launch_batch(e, in, nullptr))) break;
    if (!hip_ok(hipEventRecord(sl.done, nullptr), "hipEventRecord")) break;
    sl.busy = true;
    done += n;
  }
  // common epilogue: a failed call, or one that registered memory, leaves nothing in flight that reads the caller's buffers
  if (rc != GMX_OK || !all_pinned) {
    (void)hipStreamSynchronize(e->copy_stream);
    if (rc != GMX_OK) {
      (void)hipDeviceSynchronize();
      for (auto &sl : e->pslot) sl.busy = false;
    }
  }
  for (int i = 0; i < 4; ++i)
    if (regs[i].on) (void)hipHostUnregister(const_cast<void *>(regs[i].p));
  (void)hipGetLastError();
  return rc;
}

int gmx_map_reads_packed_host(gmx_engine *e, const uint64_t *planes, const uint64_t *offsets, uint32_t uniform_len,
                              const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads) {
  return map_reads_packed_impl(e, planes, false, offsets, uniform_len, seeds, skip, n_reads);
}

// Reads per launch of the host / device-plane feeds: the whole call, up to max_batch_reads (4 M). Every batch ends with a tail
// of few-lane kernels — on a NESTED PRG with ~2 ms of a few straggler tasks (reads inside MSA regions: hundreds of dependent
// general iterations each) whatever its size —, and the tail is paid per launch (round 5, tools/exp/engines_in_flight.py,
// kernel pipeline): configs[2] maps 138 M reads/s in launches of 250 k reads, 388 M at 1 M, 640 M at 4 M; configs[3] 914 M ->
// 1 173 M, configs[4] 350 -> 412 M, configs[1] 2.30 -> 2.63 G from 1 M to 4 M. (Until round 5 a launch took at most 2^20 reads;
// a call's first upload is now up to four times as long, the uploads behind it still hide behind the kernels.)
static uint64_t gmx_feed_chunk(const gmx_engine *e) {
  static const char *env = getenv("GMX_FEED_CHUNK");
  if (env) return std::max<uint64_t>(1, std::min<uint64_t>(e->opts.max_batch_reads, strtoull(env, nullptr, 10)));
  return e->opts.max_batch_reads;
}

// bit planes already in HBM (gmx_ingest_*): nothing to upload; seeds in device memory, or page-locked and read in place
int gmx_map_reads_packed_device(gmx_engine *e, const uint64_t *d_planes, const uint64_t *d_offsets, uint32_t uniform_len,
                                const uint32_t *seeds, const uint8_t *d_skip, uint64_t n_reads) {
  if (!e || !d_planes || !seeds || (!d_offsets && !uniform_len)) {
    gmx_set_error("gmx_map_reads_packed_device: null argument (d_offsets may be null only with uniform_len)");
    return GMX_EINVAL;
  }
  if (n_reads == 0) return GMX_OK;
  HIP_TRY(hipSetDevice(e->opts.device));
  const uint32_t *d_seeds = seeds;
  if (gmx_is_pinned(seeds)) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<uint32_t *>(seeds), 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      gmx_set_error("gmx_map_reads_packed_device: the page-locked seeds have no device address");
      return GMX_EHIP;
    }
    d_seeds = static_cast<const uint32_t *>(dp);
  }
  const uint64_t chunk = gmx_feed_chunk(e);
  if (!uniform_len && n_reads > chunk) {
    gmx_set_error("gmx_map_reads_packed_device: with d_offsets a call takes at most 2^20 reads (and at most max_batch_reads)");
    return GMX_EINVAL;
  }
  const uint64_t ppr = (uniform_len + 31u) / 32u;
  for (uint64_t done = 0; done < n_reads;) {
    const uint64_t n = std::min<uint64_t>(chunk, n_reads - done);
    BatchInput in;
    in.d_planes = reinterpret_cast<const uint2 *>(d_planes + done * ppr);
    in.d_offsets = uniform_len ? nullptr : d_offsets;
    in.d_seeds = d_seeds + done;
    in.d_skip = d_skip ? d_skip + done : nullptr;
    in.uniform_len = uniform_len;
    in.n_reads = n;
    in.total_bases = uniform_len ? n * (uint64_t)uniform_len : 0;  // (sizes the pack buffer of byte input only)
    int rc = launch_batch(e, in, nullptr);
    if (rc) return rc;
    done += n;
  }
  return GMX_OK;
}

int gmx_map_reads_2bit_host(gmx_engine *e, const uint64_t *stream, const uint64_t *offsets, uint32_t uniform_len, const uint32_t *seeds,
                            const uint8_t *skip, uint64_t n_reads) {
  return map_reads_packed_impl(e, stream, true, offsets, uniform_len, seeds, skip, n_reads);
}

int gmx_engine_seeds_in_place(gmx_engine *e, int on) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  e->seeds_in_place = on != 0;
  return GMX_OK;
}

int gmx_engine_sync_uploads(gmx_engine *e) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  if (e->copy_stream) {  // (sleeping, not polling: gmx_quiesce)
    if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventRecord(e->ev_wait, e->copy_stream));
    HIP_TRY(gmx_event_wait(e->ev_wait));
  }
  return GMX_OK;
}

// page-locked allocations are remembered so that gmx_host_free knows which call returns them. Freed page-locked blocks
// of 1 MB or more are kept (up to 16 of them) and handed out again: pinning and unpinning 100 MB costs 10-20 ms each
// way, which a reads feed would otherwise pay at its start and again at its end.
static std::mutex g_host_mu;
struct HostBlock {
  uint64_t bytes;
  bool pinned;
};
static std::map<void *, HostBlock> g_host_live;
static std::vector<std::pair<void *, uint64_t>> g_host_spare;  // pinned blocks waiting for reuse
void *gmx_host_alloc(uint64_t bytes) {
  bytes = std::max<uint64_t>(bytes, 1);
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t best = g_host_spare.size();
    for (size_t i = 0; i < g_host_spare.size(); ++i)
      if (g_host_spare[i].second >= bytes && g_host_spare[i].second <= 2 * bytes + (1u << 20) &&
          (best == g_host_spare.size() || g_host_spare[i].second < g_host_spare[best].second))
        best = i;
    if (best != g_host_spare.size()) {
      void *p = g_host_spare[best].first;
      g_host_live[p] = HostBlock{g_host_spare[best].second, true};
      g_host_spare.erase(g_host_spare.begin() + (long)best);
      return p;
    }
  }
  void *p = nullptr;
  static const unsigned alloc_flags = getenv("GMX_HOST_ALLOC_FLAGS") ? (unsigned)strtoul(getenv("GMX_HOST_ALLOC_FLAGS"), nullptr, 0) : hipHostMallocDefault;
  bool pinned = hipHostMalloc(&p, bytes, alloc_flags) == hipSuccess && p;
  if (!pinned) {
    (void)hipGetLastError();
    p = malloc(bytes);
  }
  if (p) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_live[p] = HostBlock{bytes, pinned};
  }
  return p;
}
void gmx_host_free(void *p) {
  if (!p) return;
  HostBlock blk{0, false};
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_live.find(p);
    if (it == g_host_live.end()) return;
    blk = it->second;
    g_host_live.erase(it);
    if (blk.pinned && blk.bytes >= (1u << 20) && g_host_spare.size() < 16) {
      g_host_spare.emplace_back(p, blk.bytes);
      return;
    }
  }
  if (blk.pinned)
    (void)hipHostFree(p);
  else
    free(p);
}

// Sizes the batch workspace and the staging buffers of the _host entry point ahead of the first call (otherwise the first
// call allocates them, and a later, larger call allocates them again).
int gmx_engine_reserve(gmx_engine *e, uint64_t n_reads, uint64_t n_bases) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  n_reads = std::min<uint64_t>(n_reads, e->opts.max_batch_reads);
  int rc = ensure_batch_capacity(e, n_reads);
  if (rc) return rc;
  const uint64_t need = n_bases / 32 + n_reads + 16;
  if (need > e->cap_packed) {
    if ((rc = e->alloc(&e->d_packed, need, false))) return rc;
    e->cap_packed = need;
  }
  if (n_bases > e->cap_bases) {
    if ((rc = e->alloc(&e->d_reads, n_bases + 16, false))) return rc;
    e->cap_bases = n_bases;
  }
  if (n_reads > e->cap_stage_reads) {
    if ((rc = e->alloc(&e->d_offsets, n_reads + 1, false)) || (rc = e->alloc(&e->d_seeds, n_reads, false))) return rc;
    e->cap_stage_reads = n_reads;
  }
  return GMX_OK;
}

// The same for gmx_map_reads_packed_host: the batch workspace, the copy stream and the three upload slots (bit planes,
// offsets, seeds, skip flags) for chunks of up to n_reads reads / n_pairs plane pairs.
int gmx_engine_reserve_packed(gmx_engine *e, uint64_t n_reads, uint64_t n_pairs) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  n_reads = std::min<uint64_t>(n_reads, gmx_feed_chunk(e));
  int rc = ensure_batch_capacity(e, n_reads);
  if (rc) return rc;
  if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (auto &sl : e->pslot) {
    if (sl.busy) continue;
    if (!sl.copied) {
      HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming | hipEventBlockingSync));
      HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
    }
    if (n_pairs + 16 > sl.cap_pairs) {
      e->release(sl.d_planes);
      sl.d_planes = nullptr;
      sl.cap_pairs = 0;
      if ((rc = e->alloc(&sl.d_planes, n_pairs + 64, false))) return rc;
      sl.cap_pairs = n_pairs + 64;
    }
    if (n_reads > sl.cap_reads) {
      e->release(sl.d_offsets);
      e->release(sl.d_seeds);
      e->release(sl.d_skip);
      sl.d_offsets = nullptr;
      sl.d_seeds = nullptr;
      sl.d_skip = nullptr;
      sl.cap_reads = 0;
      if ((rc = e->alloc(&sl.d_offsets, n_reads + 1, false)) || (rc = e->alloc(&sl.d_seeds, n_reads, false)) ||
          (rc = e->alloc(&sl.d_skip, n_reads, false)))
        return rc;
      sl.cap_reads = n_reads;
    }
  }
  return GMX_OK;
}

int gmx_engine_sync(gmx_engine *e) {
  HIP_TRY(hipSetDevice(e->opts.device));
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  {
    int qrc = gmx_quiesce(e);
    if (qrc) return qrc;
  }
  HIP_TRY(hipStreamSynchronize(e->last_stream));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t c[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(c + 2, e->d_error, 8, hipMemcpyDeviceToHost));
  if (c[2] != 0) {
    HIP_TRY(hipMemset(e->d_error, 0, 8));
    char msg[256];
    if (c[2] == GMX_TASK_LOGFULL) {
      snprintf(msg, sizeof(msg),
               "a read's records exceed the whole grouped-allele-count log (sites without dense group counters; %u words): "
               "nothing of it was recorded: raise gmx_engine_opts.log_cap_words",
               e->log_cap);
      gmx_set_error(msg);
      return GMX_ECAP;
    }
    if (c[2] == GMX_TASK_OVERFLOW) {
      snprintf(msg, sizeof(msg),
               "read %u (orientation %u) needs more memory for its search states or its mapping instances than the whole "
               "last-tier heap holds (%llu bytes); nothing of this read was recorded: raise gmx_engine_opts.huge_heap_bytes "
               "(GMX_HUGE_HEAP_BYTES)",
               c[3] >> 1, c[3] & 1, (unsigned long long)e->heap_words * 4);
      gmx_set_error(msg);
      return GMX_ECAP;
    }
    snprintf(msg, sizeof(msg),
             "read %u (orientation %u): inconsistent variant path (the reference throws/asserts here: a site "
             "traversed twice or an exit that does not match the entered site)",
             c[3] >> 1, c[3] & 1);
    gmx_set_error(msg);
    return GMX_EREF;
  }
  return GMX_OK;
}

int gmx_engine_enable_timing(gmx_engine *e, int on) {
  e->timing = on != 0;
  return GMX_OK;
}

int gmx_engine_timing(gmx_engine *e, gmx_timing *out) {
  HIP_TRY(hipSetDevice(e->opts.device));
  for (auto &ev : e->pending) {
    HIP_TRY(hipEventSynchronize(ev.c));
    float ms0 = 0, ms1 = 0, ms2 = 0;
    HIP_TRY(hipEventElapsedTime(&ms0, ev.s, ev.a));
    HIP_TRY(hipEventElapsedTime(&ms1, ev.a, ev.b));
    HIP_TRY(hipEventElapsedTime(&ms2, ev.b, ev.c));
    e->search_ms += ms1;
    e->cover_ms += ms0 + ms2;
    e->search_launches++;
    e->cover_launches++;
    e->timed_reads += ev.reads;
    for (int k = 0; k < GMX_TK_N; ++k) {
      if (ev.timed & (1u << k)) {  // (events of other streams: complete, ev.c is behind the batch's join)
        float ms = 0;
        HIP_TRY(hipEventSynchronize(ev.k[k][1]));
        HIP_TRY(hipEventElapsedTime(&ms, ev.k[k][0], ev.k[k][1]));
        e->kernel_ms[k] += ms;
        e->kernel_launches[k]++;
      }
      (void)hipEventDestroy(ev.k[k][0]);
      (void)hipEventDestroy(ev.k[k][1]);
    }
    (void)hipEventDestroy(ev.s);
    (void)hipEventDestroy(ev.a);
    (void)hipEventDestroy(ev.b);
    (void)hipEventDestroy(ev.c);
  }
  e->pending.clear();
  out->search_ms = e->search_ms;
  out->search_launches = e->search_launches;
  out->cover_ms = e->cover_ms;
  out->cover_launches = e->cover_launches;
  out->reads = e->timed_reads;
  for (int k = 0; k < GMX_TIMED_KERNELS; ++k) {
    out->kernel_ms[k] = e->kernel_ms[k];
    out->kernel_launches[k] = e->kernel_launches[k];
    e->kernel_ms[k] = 0;
    e->kernel_launches[k] = 0;
  }
  e->search_ms = e->cover_ms = 0;
  e->search_launches = e->cover_launches = e->timed_reads = 0;
  return GMX_OK;
}

int gmx_engine_queue_counts(gmx_engine *e, gmx_queue_counts *out) {
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t raw[GMX_N_COUNTERS * GMX_CNT_STRIDE];
  HIP_TRY(hipMemcpy(raw, e->d_counters, sizeof(raw), hipMemcpyDeviceToHost));
  auto c = [&](int i) { return (uint64_t)raw[i * GMX_CNT_STRIDE]; };
  out->mapped = 0;
  for (int r = 0; r < GMX_REGIONS; ++r) out->mapped += c(16 + r);
  out->mapped += c(8);
  out->alive = c(5);
  out->dead = c(6) + c(12);
  out->overflow_probe = c(1);
  out->overflow_extend = c(9);
  out->big_mapped = c(7);
  out->cover_general = c(8);
  out->cover_mid = c(13);
  out->cover_overflow = c(4);
  out->seed_cursor = e->seed_cursor ? 1 : 0;
  out->inst_mapped = c(25);
  out->huge_search = c(11);
  out->huge_cover = c(15);
  out->log_replays = e->log_replays;
  out->log_replayed_entries = e->log_replayed_entries;
  return GMX_OK;
}

int gmx_coverage_device(gmx_engine *e, gmx_device_coverage *out) {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  out->allele_sum = out->per_base = out->grouped = nullptr;  // interleaved in the block: use `fused`, or gmx_coverage_fetch
  out->n_allele_sum = e->n_allele;
  out->n_per_base = e->n_pb;
  out->n_grouped = e->n_grouped;
  out->stats = e->d_stats;
  out->n_stats = 5;
  out->fused = e->d_fused;
  out->n_fused = e->n_fused;
  return GMX_OK;
}

int gmx_coverage_reduce_begin(gmx_engine *e, void *hip_stream) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  hipLaunchKernelGGL(gmx_stats_limbs_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, e->d_stats, e->d_limbs, 0);
  HIP_TRY(hipGetLastError());
  return GMX_OK;
}

int gmx_coverage_reduce_end(gmx_engine *e, void *hip_stream) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  hipLaunchKernelGGL(gmx_stats_limbs_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, e->d_stats, e->d_limbs, 1);
  HIP_TRY(hipGetLastError());
  return GMX_OK;
}

int gmx_coverage_fetch(gmx_engine *e, uint32_t *allele_sum, uint32_t *per_base, uint32_t *grouped, gmx_stats *stats) {
  if (!e) {
    gmx_set_error("null engine");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<uint32_t> block(std::max<size_t>(e->n_acc, 1));
  if (e->n_acc) HIP_TRY(hipMemcpy(block.data(), e->d_fused, e->n_acc * 4, hipMemcpyDeviceToHost));
  if (allele_sum) for (size_t i = 0; i < e->phys_allele.size(); ++i) allele_sum[i] = block[e->phys_allele[i]];
  if (per_base) for (size_t i = 0; i < e->phys_pb.size(); ++i) per_base[i] = block[e->phys_pb[i]];
  if (grouped) for (size_t i = 0; i < e->phys_grouped.size(); ++i) grouped[i] = block[e->phys_grouped[i]];
  for (size_t i = 0; i + 3 < e->hit_fix.size(); i += 4) {  // a hit = one each of allele-sum, group {allele} and the base
    const uint32_t hits = block[e->hit_fix[i]];
    if (allele_sum) allele_sum[e->hit_fix[i + 1]] += hits;
    if (grouped) grouped[e->hit_fix[i + 2]] += hits;
    if (per_base) per_base[e->hit_fix[i + 3]] += hits;
  }
  if (stats) {
    unsigned long long s[5];
    HIP_TRY(hipMemcpy(s, e->d_stats, sizeof(s), hipMemcpyDeviceToHost));
    stats->all_reads_count = s[0];
    stats->skipped_reads_count = s[1];
    stats->missing_kmer_reads_count = s[2];
    stats->no_extension_reads_count = s[3];
    stats->exact_mapped_reads_count = s[4];
  }
  return GMX_OK;
}

int64_t gmx_coverage_fetch_grouped_log(gmx_engine *e, uint32_t *out, uint64_t cap_words) {
  if (!e) return GMX_EINVAL;
  if (hipSetDevice(e->opts.device) != hipSuccess) return GMX_EHIP;
  if (log_settle(e)) return GMX_EHIP;
  if (gmx_log_drain(e, 0)) return GMX_EHIP;
  uint64_t n = 0;
  for (auto const &kv : e->log_counts) {  // [site_index, n_ids | GMX_LOG_COUNTED, count lo, count hi, ids...]
    const uint64_t words = 4 + (kv.first.size() - 1);
    if (out && n + words <= cap_words) {
      out[n] = kv.first[0];
      out[n + 1] = (uint32_t)(kv.first.size() - 1) | GMX_LOG_COUNTED;
      out[n + 2] = (uint32_t)kv.second;
      out[n + 3] = (uint32_t)(kv.second >> 32);
      for (size_t j = 1; j < kv.first.size(); ++j) out[n + 3 + j] = kv.first[j];
    }
    n += words;
  }
  return (int64_t)n;
}

int gmx_coverage_import_grouped_log(gmx_engine *e, const uint32_t *records, uint64_t n_words, int replace) {
  if (!e || (!records && n_words)) {
    gmx_set_error("gmx_coverage_import_grouped_log: null argument");
    return GMX_EINVAL;
  }
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  HIP_TRY(hipSetDevice(e->opts.device));
  return gmx_engine_log_import(e, records, (size_t)n_words, replace != 0);
}

}  // extern "C"

void gmx_engine_raw(gmx_engine *e, GmxEngineRaw *out) {
  (void)flush_reset(e);
  out->device = e->opts.device;
  out->d_fused = e->d_fused;
  out->n_fused = e->n_fused;
  out->log_sites = e->log_sites;
}

int gmx_engine_log_export(gmx_engine *e, std::vector<uint32_t> &out) {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
    if ((frc = log_settle(e))) return frc;
  }
  const int64_t n = gmx_coverage_fetch_grouped_log(e, nullptr, 0);
  if (n < 0) return (int)n;
  out.assign((size_t)n, 0);
  if (n && gmx_coverage_fetch_grouped_log(e, out.data(), (uint64_t)n) < 0) return GMX_EHIP;
  return GMX_OK;
}

int gmx_engine_log_import(gmx_engine *e, const uint32_t *w, size_t n_words, bool replace) {
  {
    int frc = flush_reset(e);
    if (frc) return frc;
  }
  if (replace) {
    int rc = gmx_log_drain(e, 0);  // whatever is still on the device belongs to the totals being replaced
    if (rc) return rc;
    e->log_counts.clear();
  }
  std::vector<uint32_t> key;
  for (size_t i = 0; i < n_words;) {
    if (w[i] == GMX_LOG_PAD) {
      ++i;
      continue;
    }
    if (i + 2 > n_words) break;
    const uint32_t n = w[i + 1] & ~GMX_LOG_COUNTED;
    const size_t head = (w[i + 1] & GMX_LOG_COUNTED) ? 4 : 2;
    if (i + head + n > n_words) {
      gmx_set_error("corrupt grouped log");
      return GMX_EINVAL;
    }
    const uint64_t count = head == 4 ? ((uint64_t)w[i + 2] | ((uint64_t)w[i + 3] << 32)) : 1;
    key.assign(1, w[i]);
    key.insert(key.end(), w + i + head, w + i + head + n);
    e->log_counts[key] += count;
    i += head + n;
  }
  return GMX_OK;
}

#include "gmx_engine_debug.h"  // test hooks (final SearchStates of a task, the search loop on given states)
