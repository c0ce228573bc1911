// gmx_engine_search.h — part of the ONE translation unit gmx_engine.hip (included there, after its definitions; not a header
// to include elsewhere): read access, the per-lane contexts, the wave-level driver of the search loop (dfs_run_wave) and the
// search kernels — seed / probe / extend / instance lanes, the k-mer filter, the large-capacity and split searches.
#pragma once
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
#define GMX_SIDE_NONE 0xFFFFFFFFu  // screening side table (gmx_seed_side_kernel below): "this entry has no side words"
#define GMX_SIDE_CTX 6u            // ... bases of left context in a side word
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
  uint64_t seed_base = 0;           // word offset of the entry's FIRST state (0: not known — a cursor continued by a later kernel: no side words)
  uint32_t seed_ns = 0;             // ... and its number of states
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
      const uint32_t *side = ix.seed_side && seed_base ? ix.seed_side + ((seed_base - 1u) >> 2) : nullptr;
      if (side && side[0] == GMX_SIDE_NONE) side = nullptr;
      while (seed_left != 0) {
        if (side) {  // skip what the side words reject (consecutive words: a line or two per entry), land on the next candidate's header
          // (four side words per load, from the aligned 16 bytes that hold word j: a word per load was a chain of ~ns dependent
          //  L1 round trips per lane; what the block holds beyond the entry's words is never looked at)
          uint32_t j = seed_ns - seed_left, sw = 0;
          bool cand = false;
          while (j < seed_ns && !cand) {
            const uintptr_t at = reinterpret_cast<uintptr_t>(side + j);
            const uint4 q = *reinterpret_cast<const uint4 *>(at & ~(uintptr_t)15);
            const uint32_t qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) {
              if (cand || t < (uint32_t)((at & 15u) >> 2) || j >= seed_ns) continue;
              sw = qs[t];
              const uint32_t n6 = min((sw >> 12) & 7u, seed_rn);
              if (n6 == 0 || ((sw ^ seed_rctx) & ((1u << (2u * n6)) - 1u)) == 0u)
                cand = true;
              else
                ++j;
            }
          }
          seed_left = seed_ns - j;
          if (seed_left == 0) break;
          seed_off = seed_base + (sw >> 16);
        }
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

// The screening side table (round 6; GmxIndexView::seed_side). A whole-genome index has ~16 states per 14-mer and a task's screen
// (FastCtx::next_seed_screened, phase A) walked its entry header by header — 16 bytes of every state, 24 with paths: ~7 scattered
// 64-byte lines among 82 GB per task, and scattered lines are what the probe kernel is bound by (profiles/round4/
// config4_roofline.json: 19 G lines/s). All the screen needs of a state is its left context; here it is, ONE word per state, the
// entry's words side by side: bits 0-11 the six bases left of the position (nearest first), bits 12-14 how many there are (0: the
// state is never rejected here — an interval state, or a position right of a marker), bits 16-31 the state's word offset from the
// entry's first state. The states the six bases do not reject (one in 4 096 of the dead ones) are then judged as before, on
// their own header and text record. An entry at word offset W (its count word) of n states owns seed_side[W >> 2 .. (W >> 2) + n):
// it is at least 1 + 4 n words long, so the next entry's words start behind. GMX_SIDE_NONE in an entry's first word: no side
// words (a state lies more than 65 535 words into the entry): the entry is walked as before.
__global__ void gmx_seed_side_kernel(const GmxSeed *seeds, uint64_t n, const uint32_t *seed_words, uint32_t seed_shift, uint32_t *side) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const GmxSeed s = seeds[i];
    if (s.a != GMX_SEED_COMPLEX || (s.b & GMX_SEEDF_EMPTY)) continue;
    const uint64_t W = (uint64_t)GMX_SEED_OFF(s.b) << seed_shift;
    const uint32_t *w = seed_words + W;
    const uint32_t ns = *w++;
    uint32_t *out = side + (W >> 2);
    uint64_t rel = 0;
    bool ok = ns <= 0xFFFFu;
    for (uint32_t j = 0; ok && j < ns; ++j) {
      if (rel > 0xFFFFu) {
        ok = false;
        break;
      }
      const GmxSeedState ss = gmx_seed_state(w + rel);
      uint32_t word = (uint32_t)rel << 16;
      if (ss.text()) {
        const uint32_t cnt = min(ss.ctx >> 28, GMX_SIDE_CTX);
        word |= (ss.ctx & ((1u << (2u * cnt)) - 1u)) | (cnt << 12);
      }
      out[j] = word;
      rel += ss.words();
    }
    if (!ok && ns) out[0] = GMX_SIDE_NONE;
  }
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
  ctx.seed_base = ctx.seed_off;
  ctx.seed_ns = ns;
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
