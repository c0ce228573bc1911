// gmx_types.h — flat, GPU-resident index layout of the MI355X quasimap engine.
//
// Everything the mapping path reads is a plain array so that the same structs
// describe host memory (index builder) and HBM (kernels). Replaces the reference's
// PRG_Info bundle (libgramtools/include/prg/prg_info.hpp:22-59): SDSL csa_wt +
// 4 bit_vectors + 4 rank_support_v + marker mask + pointer-based coverage_Graph.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GMX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define GMX_HD inline
#endif

// ---------------------------------------------------------------------------
// Rank block: 128 BWT positions in one 64-byte line.
//   cnt[0..2] = number of A, C, G in BWT[0, 128*b)   (A count includes the sentinel
//               position, which is stored as code 00 with no marker bit; see gmx_rank())
//   cnt[3]    = number of marker positions (BWT symbol > 4) in BWT[0, 128*b)
//   lo/hi     = bit-sliced 2-bit base codes (A=00 C=01 G=10 T=11); markers/sentinel = 00
//   mk        = marker bit plane (reference: bwt_markers_mask, make_data_structures.cpp:158-163)
// T counts are derived: T = pos - A - C - G - M.
// Replaces 4 x (sdsl::bit_vector + rank_support_v<1>) = 8 cache lines per LF step
// (BWT_search.cpp:8-22, prg_info.cpp:20-26) by one 64-byte line.
// ---------------------------------------------------------------------------
struct alignas(64) GmxRankBlock {
  uint32_t cnt[4];
  uint64_t lo[2];
  uint64_t hi[2];
  uint64_t mk[2];
};
#define GMX_BLK_SHIFT 7
#define GMX_BLK_MASK 127u

// Graph node record (flattened coverage_Node, include/prg/coverage_graph.hpp:40-123).
struct GmxNode {
  uint32_t site;        // site marker (odd) or 0
  int32_t allele;       // allele id or -1 (boundary / outside sites)
  uint32_t seq_len;     // number of bases
  uint32_t first_pos;   // PRG position of the first base (sequence nodes) / of the marker (boundary nodes)
  uint32_t cov_off;     // slot of base 0 in the accumulator block (gmx_slot_*), 0xFFFFFFFF if the node owns none
  uint32_t edge_begin;  // first out-edge in edges[]; n_edges = next node's edge_begin - edge_begin
  uint32_t n_edges;     // copy of that difference, and
  uint32_t edge0;       // edges[edge_begin] (0xFFFFFFFF if none): a single-edge hop needs no second load
};
#define GMX_NO_COV 0xFFFFFFFFu

// Per-site record.
struct GmxSite {
  uint32_t parent_site;     // par_map[site].first, 0 when the site is level-0 (coverage_graph.cpp:193-196)
  int32_t parent_allele;    // par_map[site].second
  uint32_t n_alleles;       // edges of the bubble start
  uint32_t allele_sum_off;  // slot of the site's counter block in the accumulator block (even; see gmx_slot_*)
  uint32_t grouped_off;     // slot of its multi-allele group counters, or GMX_GROUPED_LOG (more than 8 alleles)
  uint32_t entry_node;      // bubble start node
  uint32_t exit_node;       // bubble end node
  uint32_t snp_kinds;       // bit 31: every allele is one base long or empty (below); bits 2a, 2a+1: GMX_ALLELE_* of allele a
};

#define GMX_GROUPED_LOG 0xFFFFFFFFu
// ONE accumulator block holds all three coverage structures, laid out per site so that what a read touches at a
// site sits in one cache line and the two counters every single-allele locus increments together are one 64-bit word:
//   site block (even slot): [allele-sum(a), group {a}] for a = 0 .. A-1 | groups of 2+ alleles in mask order (A <= 8)
//                           | per-base counters of the site's allele nodes
// An allele that is ONE base long (a SNP allele: one node between the site's entry and exit) gets a HIT counter after
// its per-base counter: a single-instance read through it adds 1 there instead of 1 to each of allele-sum(a),
// group {a} and the per-base counter — one atomic per SNP instead of two (the coverage kernel is bound by the rate of
// atomics). Such a node is recognised by an ODD cov_off (all others are even); its hit counter is at
// cov_off + 1 and is added to the three logical counters when coverage is fetched (HostIndex::hit_fix).
// The logical arrays of the C ABI (allele_sum, per_base, grouped_dense) are gathered from it (HostIndex::phys_*).
#define GMX_GROUPED_DENSE_MAX_ALLELES 8  // default; the upper limit; GMX_DENSE_MAX_ALLELES lowers it at index build (tests)

GMX_HD bool gmx_node_has_hit_counter(const GmxNode &n) { return n.cov_off != GMX_NO_COV && (n.cov_off & 1u); }
// A dense site whose alleles are all one base long (hit counter) or empty needs no walk to be recorded: the hit
// counters follow the site's pair and group counters in allele order (verified when the index is built).
#define GMX_ALLELE_LONG 0u   // needs the walk
#define GMX_ALLELE_HIT 1u    // one base, hit counter
#define GMX_ALLELE_EMPTY 2u  // no base: allele-sum and group {allele} only
#define GMX_SITE_WALK_FREE 0x80000000u
// Geometry of a flat site of up to 8 single-node alleles of up to 254 bases (GMX_SITE_JUMP): all that recording a read
// through the site needs, in ONE 32-byte sector of a table of its own — no walk of the coverage graph node by node, no
// look at GmxSite (gmx_cover_single). Sites without it (nested PRGs, more than 8 alleles, longer alleles) have flags = 0.
struct alignas(32) GmxSiteGeo {
  uint32_t flags;           // bits 0..15: GMX_ALLELE_* of allele a in bits 2a, 2a+1 (LONG: per-base counters); 16..19: alleles;
                            // 30: GMX_SITE_JUMP; 31: GMX_SITE_WALK_FREE (as GmxSite::snp_kinds)
  uint32_t allele_sum_off;  // GmxSite::allele_sum_off
  uint32_t entry_pos;       // PRG position of the site's entry marker; allele a starts at entry_pos + 1 + sum (len_b + 1), b < a
  uint32_t tail_len;        // base symbols between the site's end marker and the next marker (or the PRG's end)
  uint64_t allele_lens;     // bases of allele a in bits 8a .. 8a+7
  uint32_t reserved[2];
};
static_assert(sizeof(GmxSiteGeo) == 32, "one sector per site");
#define GMX_SITE_JUMP 0x40000000u
GMX_HD uint32_t gmx_geo_alleles(const GmxSiteGeo &g) { return (g.flags >> 16) & 15u; }
GMX_HD uint32_t gmx_geo_allele_len(const GmxSiteGeo &g, uint32_t allele) { return (uint32_t)(g.allele_lens >> (8u * allele)) & 0xFFu; }
GMX_HD uint32_t gmx_geo_exit_pos(const GmxSiteGeo &g) {  // PRG position of the end marker: lengths + separators
  const uint64_t m = 0x00FF00FF00FF00FFull;
  const uint64_t pairs = (g.allele_lens & m) + ((g.allele_lens >> 8) & m);
  return g.entry_pos + gmx_geo_alleles(g) + (uint32_t)((pairs * 0x0001000100010001ull) >> 48);
}
// cov_off of allele a's node by the accumulator block's layout rule (above; HostIndex verifies it against the nodes when it
// sets GMX_SITE_JUMP): the pairs, the groups of 2+ alleles, then the alleles' per-base counters in allele order — a one-base
// allele on an odd slot with its hit counter behind it, a longer one on an even slot. 0 bases: no counters.
GMX_HD uint32_t gmx_geo_cov_off(const GmxSiteGeo &g, uint32_t allele) {
  const uint32_t A = gmx_geo_alleles(g);
  uint32_t at = g.allele_sum_off + 2u * A + ((1u << A) - 1u - A), cov = 0;
  uint64_t lens = g.allele_lens;
  for (uint32_t b = 0; b <= allele; ++b, lens >>= 8) {
    const uint32_t len = (uint32_t)lens & 0xFFu;
    if (len == 1u) {
      at |= 1u;
      cov = at;
      at += 2u;
    } else if (len != 0u) {
      at += at & 1u;
      cov = at;
      at += len;
    }
  }
  return cov;
}
GMX_HD uint32_t gmx_slot_hit(const GmxSite &s, uint32_t allele) {  // walk-free sites, GMX_ALLELE_HIT alleles
  const uint32_t A = s.n_alleles;
  const uint32_t first = (s.allele_sum_off + 2u * A + ((1u << A) - 1u - A)) | 1u;
  const uint32_t before = (uint32_t)__builtin_popcount(s.snp_kinds & 0x55555555u & ((1u << (2u * allele)) - 1u));
  return first + 2u * before + 1u;
}
GMX_HD uint32_t gmx_slot_allele(const GmxSite &s, uint32_t allele) { return s.allele_sum_off + 2u * allele; }
// slot of the group counter of allele-id set `mask` (dense sites only)
GMX_HD uint32_t gmx_slot_grouped(const GmxSite &s, uint32_t mask) {
  if ((mask & (mask - 1u)) == 0) return s.allele_sum_off + 2u * (31u - (uint32_t)__builtin_clz(mask)) + 1u;
  const uint32_t bits = 32u - (uint32_t)__builtin_clz(mask);  // single-allele masks below `mask`
  return s.grouped_off + (mask - 1u) - bits;
}

// Seed directory entry (k-mer index, build/kmer_index/build.cpp:101-131), direct-addressed by the k-mer's table index:
// base j from the LEFT in bit pair j, i.e. the rightmost base is the most significant — the builder's tasks share the
// rightmost bases (suffixes) and so own contiguous ranges of the table (gmx_index.cpp, seed_walk).
//   a <= b               : exactly one path-less state [a, b]
//   a == 1, b == 0       : k-mer absent
//   a == 0xFFFFFFFF      : b << seed_shift = word offset into seed_words: [n_states, {lo, hi, n_traversed, n_traversing,
//                          (site, allele) x n_traversed (push order), site x n_traversing (push order)}*]
//                          (seed_shift = 0 unless the entries hold 2^30 words or more: whole-genome PRGs; entries then
//                          start on units of 2^seed_shift words)
struct GmxSeed {
  uint32_t a, b;
};
#define GMX_SEED_COMPLEX 0xFFFFFFFFu

// Jump program words (pre-resolved closure of search_state_vBWT_jumps, vBWT_jump.cpp:134-265, for one marker hit):
//   [n_outputs, { n_ops, (op, site, allele) x n_ops, lo, hi } x n_outputs]
#define GMX_OP_EXIT 1u   // exiting_site_search_state / update_variant_site_path (vBWT_jump.cpp:51-92)
#define GMX_OP_ENTER 2u  // entering_site_search_state (vBWT_jump.cpp:29-44)

// Marker hit record: one 64-byte line per variant marker of the PRG, in PRG (text) order; `hit_perm` maps a BWT
// marker rank to the record. The line holds four 16-byte sub-records, one per possible NEXT READ BASE c: what the
// hit does when the read continues with c. A hit therefore costs one 16-byte fetch at a known address.
//   kind PROG   site = word offset of the general jump program (pre-resolved closure of the marker's jumps)
//   kind EXIT   single output, single op EXIT(site, allele = y) -> interval [i, i]. ALIVE iff c is the base that
//               precedes the site marker; then x = SA[LF(i, c)]: the state continues in text form
//   kind ENTER  single output, single op ENTER(site) -> interval I. ALIVE iff LF(I, c) is non-empty; a width-one
//               result continues in text form (TEXT, x = its PRG position), a wider one as [x, y]
//   kind FUSED  an ENTER whose width-one result sits in a ONE-BASE allele of the same site (a SNP allele): the
//               symbol left of it is again a marker, whose hit is the EXIT of that allele. Sub-record = the ENTER
//               (x = PRG position of the allele base) plus y = allele id and head >> 4 = distance from the site's
//               opening marker to x, so that both hits resolve in one step: traversed += (site, allele), the
//               traversing path is unchanged (pushed by the ENTER, popped by the EXIT), the state continues in
//               text form at the opening marker. Used only when the read has bases left after the allele base.
// `hit_prog[h]` keeps the general program of every record (host-side lock-step search, seed table).
struct GmxHitSub {
  uint32_t head;  // kind | flags
  uint32_t site;
  uint32_t x, y;
};
struct alignas(64) GmxHit {
  GmxHitSub sub[4];
};
#define GMX_HIT_FUSED 3u
#define GMX_HITF_ALIVE 4u
#define GMX_HITF_TEXT 8u
#define GMX_HIT_PROG 0u
#define GMX_HIT_EXIT 1u
#define GMX_HIT_ENTER 2u

// The PRG itself, 64 symbols per 32-byte record, as the search consumes it once a state has narrowed to ONE
// suffix-array position i: its next backward step is decided by the symbol left of PRG position SA[i] alone
// (the LF step succeeds iff BWT[i] equals the read base, and BWT[i] = PRG[SA[i] - 1]; a marker there is the
// marker hit). Such a state is kept in TEXT FORM (a = PRG position, b = GMX_TEXT_MARK) and compares up to 64
// read bases per record against the PRG instead of fetching one rank block per base.
//
// INLINE sites (round 3). At marker positions the base planes are free; they carry two flags:
//   lo bit = the marker OPENS a site (an odd symbol);
//   hi bit = the marker CLOSES an inline site: every allele of the site is ONE base, the bases differ from each other,
//            the whole site (opening marker .. closing marker) lies inside this record, and the site's id is its ordinal
//            among the PRG's opening markers (5 + 2 * (srank + opening markers of the record below it)).
// Reaching such a closing marker with at least two read bases left, the search needs no marker record: the allele whose
// base equals the next read base is found in the planes (none: the state is dead), `traversed` gets (site, allele), and
// the state continues left of the opening marker — exactly what the pre-resolved FUSED sub-record of that marker says
// (the builder flags a site only where all four sub-records agree with this; gmx_index.cpp), without its fetch.
struct alignas(32) GmxTextRec {
  uint64_t lo, hi;  // bit planes of the base codes (A,C,G,T = 0..3); flags at marker positions (above)
  uint64_t mk;      // 1 = variant marker
  uint32_t mrank;   // markers of the PRG before this record (= index of its first marker in hits[])
  uint32_t srank;   // site-opening markers of the PRG before this record
};
#define GMX_TEXT_SHIFT 6
#define GMX_TEXT_MASK 63u
#define GMX_TEXT_MARK 0xFFFFFFFEu

// The device/host view of the index. All pointers are device pointers on the GPU side.
struct GmxIndexView {
  uint32_t n;             // text length including the sentinel (= BWT length)
  uint32_t n_prg;         // PRG length
  uint32_t sentinel_pos;  // BWT index holding the sentinel
  uint32_t kmer_size;
  uint32_t kmer_size2;    // longer seed table (0 = none), used for reads of at least that length
  uint32_t seed_shift;    // multi-state seed entries start at (offset << seed_shift) in seed_words
  uint32_t C[8];          // C[1..4]: first SA index of each base
  uint32_t n_blocks;
  uint32_t n_hits;
  uint32_t n_nodes;
  uint32_t n_sites;
  uint32_t n_allele_slots;   // allele-sum accumulator length
  uint32_t n_pb_slots;       // per-base accumulator length
  uint32_t n_grouped_slots;  // dense grouped accumulator length
  uint32_t n_acc_slots;      // length of the accumulator block all three are gathered from
  uint32_t is_nested;
  const GmxRankBlock *blocks;
  const GmxHit *hits;         // [n_hits] record of the h-th marker of the PRG (text order)
  const uint32_t *hit_perm;   // [n_hits] BWT marker rank -> index into hits[]
  const uint32_t *hit_prog;   // [n_hits] jump program of each record
  const GmxTextRec *text;     // [n_prg / 64 + 1]
  const uint32_t *prog;       // jump programs
  const uint32_t *sa;         // [n]
  const uint32_t *pos_node;   // [n_prg]
  const GmxNode *nodes;       // [n_nodes + 1] (sentinel record closes the last edge range)
  const uint32_t *edges;
  const GmxSite *sites;       // [n_sites]
  const GmxSiteGeo *site_geo;  // [n_sites] (flags = 0: no geometry)
  const GmxSeed *seeds;       // [4^k]
  const GmxSeed *seeds2;      // [4^k2] or null
  const uint32_t *seed_words;
  const uint32_t *kmer_bitmap;  // [4^k / 32] presence bits (all_read_kmers_occur_in_index, quasimap.cpp:212-225)
  const uint32_t *sa_ctx;       // [n] or null: left-context word of text position sa[i] (device only, engines with a seed cursor:
                                //   the occurrences of a path-less seed interval are screened from consecutive words; gmx_engine.hip)
  const uint32_t *seed_side;    // [n_seed_words / 4 + 1] or null: one word per state of every multi-state k-mer index entry, the states
                                //   of the entry at word offset W (its count word) at seed_side[W >> 2 ...] (device only, engines with a
                                //   seed cursor; gmx_seed_side_kernel, FastCtx::next_seed_screened)
};

// Status of one (read, orientation) task after the search kernel.
#define GMX_TASK_UNMAPPED 0u      // no final state (missing_kmer or no_extension, decided by the filter)
#define GMX_TASK_MAPPED 1u
#define GMX_TASK_OVERFLOW 2u      // capacity exceeded: must be re-run by the large-capacity kernel
#define GMX_TASK_SKIPPED 3u       // read holds a non-ACGT symbol (encode_dna_bases, utils.cpp:73-92)
#define GMX_TASK_ERROR 4u         // reference would have thrown / asserted (e.g. site traversed twice)
#define GMX_TASK_LOGFULL 6u       // the grouped-allele-count log is full (sites with more than 8 alleles)

#define GMX_NIL 0xFFFFFFFFu

// Arena node of the persistent path lists (VariantSitePath, search/types.hpp:13-15).
struct GmxPathNode {
  uint32_t site;
  int32_t allele;
  uint32_t next;  // GMX_NIL terminates
};

// A list handle is either GMX_NIL, an arena node index (< 2^31), or — for the common one-element traversing
// path of a flat PRG — the element itself, inline: GMX_INLINE_FLAG | site_index (no arena node, no load to pop it).
#define GMX_INLINE_FLAG 0x80000000u
GMX_HD bool gmx_h_inline(uint32_t h) { return h != GMX_NIL && (h & GMX_INLINE_FLAG) != 0; }
GMX_HD uint32_t gmx_h_site(const GmxPathNode *arena, uint32_t h) {
  return gmx_h_inline(h) ? 5u + 2u * (h & ~GMX_INLINE_FLAG) : arena[h].site;
}
GMX_HD int32_t gmx_h_allele(const GmxPathNode *arena, uint32_t h) { return gmx_h_inline(h) ? -1 : arena[h].allele; }
GMX_HD uint32_t gmx_h_next(const GmxPathNode *arena, uint32_t h) { return gmx_h_inline(h) ? GMX_NIL : arena[h].next; }

// Final-state record handed from the search kernel to the coverage kernel.
struct GmxFinalState {
  uint32_t lo, hi;
  uint32_t traversed;   // arena head (most recently pushed locus first) or GMX_NIL
  uint32_t traversing;  // arena head (innermost site first) or GMX_NIL
};
