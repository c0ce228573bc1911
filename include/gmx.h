/* gmx.h — C ABI of the MI355X-native quasimap engine (libgmx.so).
 *
 * Drop-in boundary for gramtools' quasimap path. The reference has no in-process
 * seam here: `gram genotype` calls quasimap_reads() directly
 * (libgramtools/src/genotype/genotype.cpp:45-46). Each entry point below names the
 * reference interface it replaces (paths relative to libgramtools/).
 *
 * Conventions: plain pointers and sizes; the caller owns every host buffer; the
 * library owns device memory. Every function returns 0 on success or a negative
 * GMX_E* code; gmx_last_error() returns the message of the calling thread's last
 * failure. No C++ exception leaves the library: every entry point is a function-try-block
 * (gmx_internal.h: GMX_GUARD_*), host memory exhaustion is GMX_ENOMEM (the reference's
 * process ends such a run with a message and a non-zero exit code,
 * gramtools/commands/genotype/genotype.py:106-107 — not with SIGABRT). One engine = one GPU = one host thread at a time.
 * Mapping entry points ALWAYS run on the GPU; there is no CPU fallback.
 */
#ifndef GMX_H
#define GMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMX_OK 0
#define GMX_EINVAL (-1)   /* bad argument / inconsistent PRG (reference: std::runtime_error in PRG_String / cov_Graph_Builder) */
#define GMX_ENODEV (-2)   /* no usable HIP device */
#define GMX_EHIP (-3)     /* HIP runtime error */
#define GMX_ECAP (-4)     /* a read needs more than the last tier's heap, or more of the grouped log than the whole log (raise gmx_engine_opts) */
#define GMX_EREF (-5)     /* a read hit a condition on which the reference throws/asserts */
#define GMX_ENOMEM (-6)

typedef struct gmx_index gmx_index;   /* host-side index: stands for PRG_Info + KmerIndex (include/prg/prg_info.hpp:22-59) */
typedef struct gmx_engine gmx_engine; /* index resident in HBM + coverage accumulators (Coverage, coverage/types.hpp:40-46) */

const char *gmx_last_error(void);

/* ---- index construction (host) ------------------------------------------------
 * Replaces load_prg_info() (src/prg/prg_info.cpp:6-29) + kmer_index::load()
 * (src/build/kmer_index/load.cpp:161-173): everything is re-derived from the integer
 * PRG (gram_dir/prg, little-endian uint32 per symbol) and the k-mer size. */
int gmx_index_build(const uint32_t *prg, uint64_t n_symbols, uint32_t kmer_size, int threads, gmx_index **out);
int gmx_index_build_from_file(const char *prg_path, uint32_t kmer_size, int threads, gmx_index **out);
/* Index cache (replaces what `gram build` leaves in gram_dir for the reference: fm_index, masks, kmer
 * index files, cov_graph — build/build.cpp:8-72). gmx_index_save writes every derived table to one file;
 * gmx_index_load reads it back after checking that it was built from the PRG at `prg_path` (length + hash of the
 * symbols) with the same kmer_size; any mismatch or damage is GMX_EINVAL and the caller rebuilds. */
int gmx_index_save(const gmx_index *ix, const char *path);
int gmx_index_load(const char *cache_path, const char *prg_path, uint32_t kmer_size, gmx_index **out);
void gmx_index_destroy(gmx_index *ix);

typedef struct gmx_index_info {
  uint64_t n_text;           /* PRG length + sentinel */
  uint32_t kmer_size;
  uint32_t n_sites;          /* prg_info.num_variant_sites */
  uint32_t is_nested;        /* coverage_graph.is_nested */
  uint32_t n_allele_slots;   /* length of the flat allele-sum array */
  uint32_t n_per_base_slots; /* length of the flat per-base array */
  uint32_t n_grouped_slots;  /* length of the dense grouped-counts array */
  uint32_t n_nodes;
  uint64_t n_kmers_present;
  uint64_t index_bytes;      /* bytes uploaded to HBM by gmx_engine_create */
  uint32_t kmer_size2;       /* length of the longer seed table the search is seeded from (0 = none); see DESIGN.md */
  uint32_t n_inline_sites;   /* sites the text-form search resolves without their marker record (one-base alleles inside one
                                text record; gmx_types.h) */
  uint32_t seed_shift;       /* multi-state k-mer index entries start on units of 2^seed_shift words (0 below 2^30 words;
                                whole-genome PRGs need more: build/kmer_index/build.cpp:101-131 has no such limit) */
  uint32_t n_jump_sites;     /* sites whose coverage is recorded from the site record alone (flat sites of up to 8 single-node
                                alleles of up to 254 bases; gmx_types.h GMX_SITE_JUMP) */
  uint64_t n_seed_words;     /* words of the multi-state entries (SearchStates with variant paths, several per k-mer) */
} gmx_index_info;
int gmx_index_get_info(const gmx_index *ix, gmx_index_info *out);

/* Per-site layout: for site index s (site marker 5 + 2s):
 *   n_alleles[s], allele_sum_off[s], grouped_off[s] (0xFFFFFFFF = site uses the grouped log),
 *   parent_site[s] (0 = level-0), parent_allele[s]. Any pointer may be NULL. */
int gmx_index_site_layout(const gmx_index *ix, uint32_t *n_alleles, uint32_t *allele_sum_off, uint32_t *grouped_off,
                          uint32_t *parent_site, int32_t *parent_allele);
/* Per-base layout of non-boundary allele nodes: for each (site index, allele id, k-th sequence node of that
 * allele) the slice [pb_off, pb_off+len). Records are (site_index, allele, first_prg_pos, pb_off, len) x n;
 * returns the number of records; pass out == NULL to query the count. */
int64_t gmx_index_per_base_layout(const gmx_index *ix, uint32_t *out, uint64_t cap_records);
/* allele_base_non_nested view (allele_base.cpp:10-38): for every site and allele the per-base slice of the
 * allele's single sequence node, or len 0 for a direct deletion. Records (pb_off, len) in site-major order. */
int gmx_index_allele_base_layout(const gmx_index *ix, uint32_t *pb_off, uint32_t *len);

/* Read-depth statistics of read_stats.json (AbstractReadStats::compute_coverage_depth, src/genotype/read_stats.cpp:72-160),
 * from the raw uint32 totals returned by gmx_coverage_fetch (+ the grouped log). */
typedef struct gmx_depth_stats {
  double mean_cov_depth, variance_cov_depth;
  uint64_t num_sites_noCov, num_sites_total;
} gmx_depth_stats;
int gmx_compute_coverage_depth(const gmx_index *ix, const uint32_t *per_base_raw, const uint32_t *grouped_dense_raw,
                               const uint32_t *grouped_log, uint64_t n_log_words, gmx_depth_stats *out);
/* Site markers in the iteration order of the reference's bubble_map (descending first-allele position, then
 * descending site id; prg/types.hpp:26, coverage_graph.cpp:381-389). */
int gmx_index_bubble_order(const gmx_index *ix, uint32_t *site_markers_out);

/* Introspection (tests, debugging): copies of the derived structures. */
/* The suffix-array builder instantiated with 16-bit indices (text of n < 65535 symbols ending in the only 0): texts
 * longer than 2^15 drive it with the index type's top bit in use, as whole-human PRGs (> 2^31 symbols) drive the
 * 32-bit instance the index uses (SA_Index is uint32_t in the reference, search/types.hpp:19). */
int gmx_debug_suffix_array_u16(const uint16_t *text, uint64_t n, uint16_t *out);
int gmx_index_copy_sa(const gmx_index *ix, uint32_t *out);             /* n_text entries (fm_index[i]) */
int gmx_index_copy_bwt(const gmx_index *ix, uint32_t *out);            /* n_text entries */
uint32_t gmx_index_rank(const gmx_index *ix, uint32_t upper, uint32_t base); /* dna_bwt_rank, BWT_search.cpp:8-22 */
int gmx_index_copy_pos_info(const gmx_index *ix, int64_t *out);        /* per PRG position: node site, node allele,
                                                                          offset in node, target marker, target allele */
int64_t gmx_index_copy_target_map(const gmx_index *ix, int64_t *out, uint64_t cap); /* [n, {key, n_t, (id, del)*}*] */
int64_t gmx_index_seed_states(const gmx_index *ix, const uint8_t *kmer, int64_t *out, uint64_t cap);
/* the same for a k-mer of length `len` = kmer_size or kmer_size2 (the longer seed table) */
int64_t gmx_index_seed_states_k(const gmx_index *ix, const uint8_t *kmer, uint32_t len, int64_t *out, uint64_t cap);
        /* k-mer index entry: [n_states, {lo, hi, n_tvd, (site, allele)*, n_tvg, (site, -1)*}*] or [-1] if absent */
int64_t gmx_index_jump_states(const gmx_index *ix, uint32_t lo, uint32_t hi, int64_t *out, uint64_t cap);
        /* search_state_vBWT_jumps of a path-less state [lo, hi] (vBWT_jump.cpp:134-183), same format */

/* ---- engine (device) ---------------------------------------------------------------*/
typedef struct gmx_engine_opts {
  int device;              /* HIP device ordinal */
  int rng_mode;            /* 0 = libstdc++ >= 11 uniform_int_distribution (default), 1 = libstdc++ <= 10 */
  uint32_t max_states;     /* per-read search-state capacity of the large-capacity pass (default 4096) */
  uint32_t max_path_nodes; /* per-read path arena of the large-capacity pass (default 8192) */
  uint64_t max_batch_reads;/* reads per internal launch (default 4M) */
  int forward_only;        /* 1 = map only the given orientation of each read (quasimap_read, quasimap.cpp:159-194,
                              as the reference's unit tests call it); 0 = forward + reverse complement (default) */
  uint64_t huge_heap_bytes;/* memory of the last tier (default 512 MiB): a read whose search states or mapping instances
                              exceed every fixed pool is redone with pools carved from this heap — the whole heap if need
                              be — so the only limit left is this number, as memory is the reference's only limit
                              (encapsulated_search.cpp:30-107, coverage_common.cpp:85-146). GMX_ECAP names the read that
                              does not fit it. */
  uint64_t log_cap_words;  /* device log of grouped counts of sites with more than 8 alleles, in uint32 words (0 = 2^26). Between
                              batches the engine reads its fill back: drained to the host when half full; tasks that found it
                              full (nothing of theirs recorded) are redone after a drain. GMX_ECAP only when ONE task's
                              records exceed the whole log */
} gmx_engine_opts;
void gmx_engine_default_opts(gmx_engine_opts *opts);

/* Uploads the index to HBM and allocates zeroed coverage accumulators.
 * Replaces coverage::generate::empty_structure (coverage_common.cpp:206-213). */
int gmx_engine_create(const gmx_index *ix, const gmx_engine_opts *opts, gmx_engine **out);
void gmx_engine_destroy(gmx_engine *e);
int gmx_engine_reset(gmx_engine *e); /* zero coverage + statistics (synchronises the device) */
int gmx_engine_reset_async(gmx_engine *e, void *hip_stream); /* the same, enqueued on hip_stream */

/* Quasimap a batch of reads, forward and reverse complement (replaces handle_reads_buffer +
 * quasimap_forward_reverse + quasimap_read, quasimap.cpp:82-194).
 *   reads   : concatenated encoded reads, one byte per base, A,C,G,T = 1,2,3,4 (encode_dna_bases,
 *             common/utils.cpp:73-92). A read holding any other value is skipped (both orientations counted
 *             as skipped), as is a read shorter than the k-mer size.
 *   offsets : n_reads + 1 byte offsets into `reads`
 *   seeds   : per-read selection seed (the i-th draw of the master generator, quasimap.cpp:136-137)
 * The _host variant copies the buffers to the device; the _device variant takes device pointers
 * (already resident in HBM) and enqueues on `hip_stream` (a hipStream_t, NULL = default stream)
 * without synchronising. Coverage accumulates on the device across calls. */
int gmx_map_reads_host(gmx_engine *e, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                       uint64_t n_reads);
int gmx_map_reads_device(gmx_engine *e, const uint8_t *d_reads, const uint64_t *d_offsets, const uint32_t *d_seeds,
                         uint64_t n_reads, uint64_t total_bases, void *hip_stream);
/* The same from reads that are ALREADY bit planes in host memory ("pre-encoded (2-bit packed ...) pinned-host"
 * input; replaces get_reads_buffer + encode_dna_bases, quasimap.cpp:65-76 / common/utils.cpp:73-92, whose result the
 * reference hands to handle_reads_buffer): a quarter of the bytes over PCIe and no packing pass on the device.
 *   planes : one uint64 per 32 bases, low word = bit 0 of the base codes (A,C,G,T = 0,1,2,3: encoded value - 1), high word =
 *            bit 1; base j of the 32 at bit j; bits past a read's end are ignored. Read r of the call starts at pair
 *              P(r) = r * ceil(uniform_len / 32)                        when uniform_len != 0 (every read that long), else
 *              P(r) = (offsets[r] >> 5) - (offsets[0] >> 5) + r         (ceil(len / 32) pairs always fit before P(r + 1);
 *            a sub-range of a packed batch is again a packed batch: pass planes + P(first), offsets + first).
 *            gmx_packed_pairs() gives the length of the array, gmx_pack_reads() makes it from encoded bytes.
 *   offsets: n_reads + 1 base offsets (only their differences and the layout above matter), or NULL with uniform_len
 *   skip   : per read, non-zero = the read holds a non-ACGT symbol: both orientations count as skipped
 *            (encode_dna_bases semantics); NULL = no such read
 * ASYNCHRONOUS when the buffers are page-locked (gmx_host_alloc): the call returns once every chunk is enqueued — uploads
 * on a copy stream beside the kernels of the chunks before — and the buffers must stay untouched until
 * gmx_engine_sync_uploads (uploads done; kernels may still run) or gmx_engine_sync / gmx_coverage_fetch. Pageable buffers
 * are registered for the duration of the call, which then waits for its uploads. */
int gmx_map_reads_packed_host(gmx_engine *e, const uint64_t *planes, const uint64_t *offsets, uint32_t uniform_len,
                              const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads);
int gmx_engine_sync_uploads(gmx_engine *e);
/* Seeds in place (off by default): gmx_map_reads_packed_host then uploads NO seeds when `seeds` lies in gmx_host_alloc memory —
 * a read's seed is consulted only when the read has several equally good mapping classes (coverage_common.cpp:166-177), and
 * the kernels read those few from the caller's buffer over PCIe (4 of the 44 bytes per 150 bp read stay on the host). The
 * price is a longer hold: the seeds must stay untouched until gmx_engine_sync / gmx_coverage_fetch, not just until
 * gmx_engine_sync_uploads. Results are the same either way. */
int gmx_engine_seeds_in_place(gmx_engine *e, int on);
uint64_t gmx_packed_pairs(const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads);
/* Encoded reads (one byte per base, 1..4; the input of gmx_map_reads_host) -> bit planes in the layout above, on `threads`
 * host threads (0 = all). skip[r] (may be NULL) = read r holds a byte outside 1..4. With uniform_len every read must be
 * that long (GMX_EINVAL otherwise). */
int gmx_pack_reads(const uint8_t *reads, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads, uint64_t *planes,
                   uint8_t *skip, int threads);
/* The same feed with the reads as a 2-bit STREAM: base j of the call (the reads back to back, in order) in bits 2j, 2j + 1 of
 * `stream` (little-endian: 32 bases per uint64, base 0 in the lowest bits; A,C,G,T = 0,1,2,3). A read starts wherever the
 * one before it ends — 37.5 bytes per 150 bp read over PCIe instead of the planes' 40, 25 instead of 32 per 100 bp read — and
 * the first kernel of the batch turns the stream into the bit planes (gmx_unpack2_kernel, HBM to HBM). offsets / uniform_len /
 * seeds / skip, asynchrony and gmx_engine_sync_uploads as for gmx_map_reads_packed_host; gmx_twobit_units() gives the
 * length of `stream` in uint64 (one unit of slack included), gmx_pack_reads_2bit() makes it from encoded bytes. */
int gmx_map_reads_2bit_host(gmx_engine *e, const uint64_t *stream, const uint64_t *offsets, uint32_t uniform_len,
                            const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads);
uint64_t gmx_twobit_units(const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads);
int gmx_pack_reads_2bit(const uint8_t *reads, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_reads, uint64_t *stream,
                        uint8_t *skip, int threads);
/* The same from bit planes that are already IN HBM (what gmx_ingest_* below leaves there): nothing is uploaded but the
 * seeds — `seeds` is a device pointer, or a gmx_host_alloc pointer, which the kernels then read in place over PCIe as with
 * gmx_engine_seeds_in_place. Layout of d_planes / d_offsets / d_skip as for gmx_map_reads_packed_host; with d_offsets the
 * call takes at most 2^20 reads (a longer batch is split by the caller at reads whose pair index it knows:
 * gmx_ingest_result::sub_pairs). Enqueues on the engine's stream without synchronising; the buffers must stay untouched
 * until gmx_engine_sync (or an event recorded behind the call on the NULL stream). */
int gmx_map_reads_packed_device(gmx_engine *e, const uint64_t *d_planes, const uint64_t *d_offsets, uint32_t uniform_len,
                                const uint32_t *seeds, const uint8_t *d_skip, uint64_t n_reads);
/* An engine of a nested PRG or of an index of 2 GB and more keeps TWO launches in flight: it hands consecutive launches of the
 * host feeds and of gmx_map_reads_packed_device to its two workspaces in turn, the first on the NULL stream, the second on a
 * stream of its own (a launch's tail of few-lane kernels beside the next launch's full-GPU kernels). This returns
 * that second stream (NULL: the engine has one workspace). A caller that orders its own work behind launches it has enqueued —
 * gmx_ingest_release_after — does so on both streams. gmx_engine_sync, gmx_coverage_fetch and the exchange wait for both. */
void *gmx_engine_second_stream(gmx_engine *e);

/* ---- reads files decoded on the device -----------------------------------------------------------
 * Replaces the reference's reads reader for gzipped FASTQ — SeqRead over zlib / htslib on one host thread
 * (include/sequence_read/seqread.hpp:94-180, quasimap.cpp:65-76) — for BGZF files (bgzip, htslib, BCL Convert: gzip members
 * of <= 64 KB that carry their size in a `BC` extra field, SAM spec §4.1): the compressed members are uploaded as they lie in
 * the file and HIP kernels inflate them (one wavefront per member, CRC-32 checked), find the four-line records and pack
 * their bases into the bit planes of gmx_map_reads_packed_host, all in HBM. A file is handed over in CHUNKS of whole
 * members, in order; a record cut by a chunk's end is carried into the next chunk on the device. THREE slots (0, 1, 2) take
 * the chunks in turn, so that two chunks upload and inflate — their inflate kernels side by side: a kernel's last wavefronts
 * leave most of the GPU idle — while the one before them is mapped:
 *     submit(0, chunk 0); submit(1, chunk 1); submit(2, chunk 2); wait(0) -> map -> release_after(0); submit(0, chunk 3); wait(1) ...
 * (a caller may also alternate between two of the slots, as until round 5: consecutive chunks must go to different slots).
 * There is no CPU fallback inside: a chunk the kernels cannot take (status != 0) is the caller's to handle — `gram` lets its
 * host reader take the whole file from its start (dropping the reads already mapped), which reports the damage or, if the
 * device decoder was at fault, delivers the rest; a caller may also inflate the chunk itself and pass the text through
 * gmx_ingest_submit_text (same kernels behind the inflate step). */
typedef struct gmx_ingest gmx_ingest;
typedef struct {
  uint64_t offset;   /* of the member's deflate data within the chunk's bytes (behind the gzip header and its extra field) */
  uint32_t size;     /* bytes of deflate data */
  uint32_t isize;    /* bytes of text it inflates to (the member's trailer; <= 65536) */
  uint32_t crc32;    /* of that text (the trailer) */
  uint32_t reserved;
} gmx_bgzf_member;
#define GMX_INGEST_BAD_RECORD 1u      /* not plain four-line FASTQ (blank / multi-line record, '@' or '+' missing, lengths differ) */
#define GMX_INGEST_BAD_MEMBER 2u      /* a member's deflate data could not be decoded */
#define GMX_INGEST_BAD_CRC 4u         /* a member's text does not match its trailer */
#define GMX_INGEST_TOO_MANY_LINES 8u  /* more lines / records than the ingest has room for (lines of a few bytes) */
typedef struct {
  uint32_t status;        /* 0, or GMX_INGEST_* bits: the reads below are then NOT to be used */
  uint32_t bad_member;    /* first member (index within the chunk) with GMX_INGEST_BAD_MEMBER / _BAD_CRC */
  uint64_t n_reads, n_bases, n_pairs;
  uint32_t uniform_len;   /* != 0: every read has this many bases (d_offsets is NULL) */
  uint32_t any_skip;      /* some read holds a non-ACGT letter (d_skip says which) */
  uint64_t text_bytes;    /* text of the chunk, the carried start of its first record included */
  uint64_t consumed_bytes, tail_bytes; /* taken by complete records / carried into the next chunk */
  uint64_t sub_pairs[16]; /* pair index of read i * 2^20 within d_planes (reads of different lengths: where a sub-batch starts) */
  const uint64_t *d_planes, *d_offsets; /* device memory of the slot, valid until the slot's next submit */
  const uint8_t *d_skip;
} gmx_ingest_result;
/* max_text_bytes: most text (inflated bytes) a chunk may hold; compressed bytes per chunk: up to gmx_ingest_max_compressed(). */
int gmx_ingest_create(int device, uint64_t max_text_bytes, gmx_ingest **out);
void gmx_ingest_destroy(gmx_ingest *g);
uint64_t gmx_ingest_max_text(const gmx_ingest *g);
uint64_t gmx_ingest_max_compressed(const gmx_ingest *g);
uint64_t gmx_ingest_max_members(const gmx_ingest *g); /* most members a chunk may hold (files of many tiny members) */
int gmx_ingest_reset(gmx_ingest *g); /* the next chunk is a file's first: nothing is carried into it */
/* `compressed` (host memory; page-locked = asynchronous upload) must stay untouched until the slot's gmx_ingest_wait. */
int gmx_ingest_submit_bgzf(gmx_ingest *g, int slot, const uint8_t *compressed, uint64_t n_bytes, const gmx_bgzf_member *members,
                           uint64_t n_members, int final_chunk);
int gmx_ingest_submit_text(gmx_ingest *g, int slot, const uint8_t *text, uint64_t n_bytes, int final_chunk);
int gmx_ingest_wait(gmx_ingest *g, int slot, gmx_ingest_result *out);
/* A file's chunks dealt over SEVERAL devices (one gmx_ingest each; SURVEY.md §8e: the path shards by chunk): the record cut by a
 * chunk's end continues on another device, so the caller carries it through the host. _deferred uploads a chunk and enqueues its
 * inflate kernel at once; gmx_ingest_scan enqueues the rest when the caller holds the end of the chunk before
 * (gmx_ingest_fetch_tail of that chunk's slot after its gmx_ingest_wait; nothing for a file's first chunk), then gmx_ingest_wait as
 * usual. Chunks are scanned in file order; their inflate kernels run ahead on all devices. */
int gmx_ingest_submit_bgzf_deferred(gmx_ingest *g, int slot, const uint8_t *compressed, uint64_t n_bytes, const gmx_bgzf_member *members,
                                    uint64_t n_members);
int gmx_ingest_submit_text_deferred(gmx_ingest *g, int slot, const uint8_t *text, uint64_t n_bytes); /* plain text dealt the same way */
int gmx_ingest_scan(gmx_ingest *g, int slot, const uint8_t *carry, uint64_t n_carry, int final_chunk);
int64_t gmx_ingest_fetch_tail(gmx_ingest *g, int slot, uint8_t *out, uint64_t cap); /* NULL out: its length */
/* The slot's planes are read by work enqueued on hip_stream (the mapping call): its next submit waits for that work. May be
 * called twice per chunk, for two streams (gmx_engine_second_stream): the next submit waits for both. */
int gmx_ingest_release_after(gmx_ingest *g, int slot, void *hip_stream);
/* test hooks: the chunk's text (NULL out: its length), its reads in the host layout of gmx_map_reads_packed_host */
int64_t gmx_ingest_fetch_text(gmx_ingest *g, int slot, uint8_t *out, uint64_t cap);
int gmx_ingest_fetch_reads(gmx_ingest *g, int slot, uint64_t *planes, uint64_t *offsets, uint8_t *skip);

/* Page-locked host memory for the buffers handed to gmx_map_reads_host: the upload is then one DMA at the PCIe rate
 * instead of being staged through small pinned chunks by the runtime. Falls back to plain memory without a device (the
 * parsers also run in tests without one). gmx_host_free takes only pointers gmx_host_alloc returned. */
void *gmx_host_alloc(uint64_t bytes);
void gmx_host_free(void *p);
/* Sizes the engine's batch workspace (and the device staging buffers of gmx_map_reads_host) for calls of up to n_reads
 * reads / n_bases bases ahead of the first call. Optional: the first call does it otherwise. */
int gmx_engine_reserve(gmx_engine *e, uint64_t n_reads, uint64_t n_bases);
/* The same ahead of gmx_map_reads_packed_host: workspace, copy stream and upload slots for chunks of up to n_reads reads
 * whose bit planes take up to n_pairs uint64. */
int gmx_engine_reserve_packed(gmx_engine *e, uint64_t n_reads, uint64_t n_pairs);
/* Waits for enqueued work and reports a read that overflowed / errored (GMX_ECAP, GMX_EREF). */
int gmx_engine_sync(gmx_engine *e);

/* Kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg). After
 * gmx_engine_enable_timing(e, 1) every internal launch of the search and coverage kernels is bracketed by events;
 * gmx_engine_timing() (call after gmx_engine_sync) returns the accumulated milliseconds and launch counts since
 * the last call and resets them. */
int gmx_engine_enable_timing(gmx_engine *e, int on);
#define GMX_TIMED_KERNELS 8
typedef struct gmx_timing {
  double search_ms;  uint64_t search_launches;   /* gmx_extend_kernel (the dominant kernel) */
  double cover_ms;   uint64_t cover_launches;    /* everything else: validate, probe, big pass, filter, coverage, stats */
  uint64_t reads;                                /* reads covered by those launches */
  /* further kernels, each with events attached to its own dispatch (start / end of that kernel alone, whichever stream):
   * [0] gmx_seed_kernel (gmx_probe_kernel without a longer seed table)   [1] k-mer filter, pass 0   [2] k-mer filter, pass 1
   * [3] the single-instance coverage kernel (gmx_cover_jump_kernel; gmx_cover_single_kernel on nested PRGs or with
   *     GMX_NO_COVER_JUMP)   [4] gmx_extend_kernel's first pass over the stragglers   [5] gmx_unpack2_kernel (2-bit feed) */
  double kernel_ms[GMX_TIMED_KERNELS];
  uint64_t kernel_launches[GMX_TIMED_KERNELS];
} gmx_timing;
int gmx_engine_timing(gmx_engine *e, gmx_timing *out);

/* ---- files of a stock gramtools gram_dir ------------------------------
 * Readers of the SDSL-lite 2.1.1 vectors a stock `gramtools build` writes: kmers (int_vector<3>), kmers_stats,
 * sa_intervals, paths (int_vector<>, bit-compressed) — read as build/kmer_index/load.cpp:71-173 reads them — and the four
 * {a,c,g,t}_base_bwt_mask bit_vectors (prg/make_data_structures.cpp:140-156). On-disk form: 64-bit length in bits, a
 * width byte for variable-width vectors, values packed LSB-first in 64-bit words. fm_index (csa_wt) and cov_graph (Boost
 * archive) are not read: both are functions of gram_dir/prg, from which the native index is built.
 * PARITY UNPINNED: no reference-written file exists here; checked against the format, hand-assembled bytes and a round
 * trip through the writers (tests/test_stock_files.py).
 * fixed_width: 0 = variable-width vector (width byte in the file), else the vector's compile-time width (3 for kmers,
 * 1 for a bit_vector). Returns the number of elements (also with out == null), or a negative code. */
int64_t gmx_stock_read_int_vector(const char *path, uint32_t fixed_width, uint64_t *out, uint64_t cap, uint32_t *width_out);
int gmx_stock_write_int_vector(const char *path, const uint64_t *values, uint64_t n, uint32_t width, int fixed);
/* the index's k-mer table and BWT masks in those formats (as dump.cpp:27-137 / make_data_structures.cpp:112-138 write them) */
int gmx_index_write_stock_files(const gmx_index *ix, const char *gram_dir);
typedef struct gmx_stock_report {
  uint64_t kmers, states;               /* read from the files */
  uint64_t kmer_mismatches;             /* k-mers whose states differ from the native index's (as sets of states) */
  uint64_t kmers_missing_in_files;      /* indexed natively, absent from the files (a stock build may index fewer k-mers) */
  uint64_t duplicate_kmers;
  uint64_t mask_bits, mask_mismatches;  /* over the four masks */
  /* cov_graph (a Boost binary archive of coverage_Graph, prg_info.cpp:13-15; only its head is looked at — the archive signature,
   * the library version, and the first collection count behind them, which is bubble_map's = the number of variant sites,
   * coverage_graph.hpp:220-233). 0 = no such file; 1 = signature found; 2 = ... and the site count equals the native index's;
   * 3 = signature found, a different count where the site count was expected */
  uint64_t cov_graph_state, cov_graph_library_version, cov_graph_sites;
  /* fm_index (sdsl::csa_wt over int_vector<>): bytes of the file, 0 = absent; not decoded (SA and BWT are re-derived from prg) */
  uint64_t fm_index_bytes;
} gmx_stock_report;
/* reads the files of gram_dir and compares them with the native index of the same PRG and k */
int gmx_index_check_stock_files(const gmx_index *ix, const char *gram_dir, gmx_stock_report *out);

/* ---- test hooks: SearchStates of the HIP path -----------------------------------------
 * The reference's unit tests pin the search at the level of SearchStates (search/types.hpp:31-57;
 * tests/genotype/quasimap/search/test_vBWT_jump.cpp:55-405, test_encapsulated_search.cpp:28-254, the
 * search_read_backwards cases of test_quasimap.cpp). These entry points expose the device's states so that
 * tests/ can run those vectors on the HIP kernels; none of them is on the mapping path.
 * States travel as words in the k-mer index's serialisation, SA intervals as the reference has them:
 *   [n_states, {lo, hi, n_traversed, n_traversing, (site, allele) x n_traversed, site x n_traversing}*]   (push order)
 * Every call sets *n_words to the words of the answer; GMX_ECAP when `out` (non-null) holds fewer.
 *
 * gmx_engine_debug_keep_states(e, 1): from the next batch on, every task's final states stay readable after the
 *   batch (the kernels then also write the copies a compact coverage record makes redundant).
 * gmx_debug_final_states: the final states of task 2 * read + orientation of the LAST launch (one call of a
 *   gmx_map_reads_* entry point with at most max_batch_reads reads), as search_read_backwards leaves them
 *   (quasimap.cpp:227-256, before handle_allele_encapsulated_states). *tier: 0 fast pass, 1 a large-capacity
 *   slot, 2 instance lanes. GMX_ECAP for a task the last tier searched (it keeps nothing).
 * gmx_debug_search: the device's search loop on ONE read (bases 1..4) from caller-given states at read position
 *   `from` (bases [0, from) are still to be matched) down to `stop`; lf_only: the given states take their first
 *   step without a marker pass (search_base_backwards, BWT_search.cpp:78-94), else marker pass + LF step
 *   (process_read_char_search_states, quasimap.cpp:258-268). from_seed_table != 0: seeded from the k-mer index
 *   entry of the read's last k-mer and run to the read's start instead (search_read_backwards).
 * gmx_debug_encapsulate: the device's handle_allele_encapsulated_states (encapsulated_search.cpp:30-107) on the
 *   given states: position by position; out = the states that stay mapping instances of a site, nonvariant_sa =
 *   the SA indices of the path-less positions outside every site (the reference keeps them as path-less states). */
int gmx_engine_debug_keep_states(gmx_engine *e, int on);
/* Allocation-failure injection (tests/test_alloc_failure.py): the nth call of operator new made by libgmx.so's own code
 * from now on throws std::bad_alloc, once; nth < 0: only count allocations; nth == 0: off, counting stops (the counter is one
 * cache line every allocating thread hits: it is on only between such calls); also GMX_TEST_FAIL_ALLOC=n in the environment
 * at load time. Returns the
 * number of allocations the library has made since its first call (counting starts with the first call or the variable; so a test can count the allocations of a call and walk n over them).
 * The HIP runtime's, RCCL's and the host program's allocations are not touched. */
uint64_t gmx_debug_fail_alloc(int64_t nth);
int gmx_debug_final_states(gmx_engine *e, uint64_t task, uint32_t *out, uint64_t cap_words, uint64_t *n_words, int *tier);
int gmx_debug_search(gmx_engine *e, const uint8_t *read, uint32_t read_len, int from_seed_table, const uint32_t *states,
                     uint64_t n_state_words, uint32_t from, uint32_t stop, int lf_only, uint32_t *out, uint64_t cap_words,
                     uint64_t *n_words);
int gmx_debug_encapsulate(gmx_engine *e, const uint32_t *states, uint64_t n_state_words, uint32_t *out, uint64_t cap_words,
                          uint64_t *n_words, uint32_t *nonvariant_sa, uint64_t cap_nonvariant, uint64_t *n_nonvariant);

/* Queue lengths of the LAST batch (after gmx_engine_sync): how many (read, orientation) tasks took which route.
 * Diagnostics for capacity planning; tasks_overflow_* are redone by the large-capacity kernel. */
typedef struct gmx_queue_counts {
  uint64_t mapped;            /* tasks with final states from the probe / extend kernels */
  uint64_t alive;             /* tasks parked by the probe kernel for the extend kernel */
  uint64_t dead;              /* tasks without final state (k-mer filter decides their counter) */
  uint64_t overflow_probe;    /* tasks that overflowed the per-lane pools in the probe kernel */
  uint64_t overflow_extend;   /* ... in the extend kernel */
  uint64_t big_mapped;        /* large-capacity tasks with final states */
  uint64_t cover_general;     /* mapped tasks that needed the general coverage kernel (first tier: scratch in LDS) */
  uint64_t cover_mid;         /* ... of which those that needed its global-memory scratch */
  uint64_t cover_overflow;    /* ... and those (large-capacity tasks included) that needed the large scratch */
  uint64_t seed_cursor;       /* 1: the engine runs the seed-cursor kernels (index with many multi-state k-mer entries) */
  uint64_t huge_search;       /* tasks searched again by the last tier (pools carved from the heap) */
  uint64_t inst_mapped;       /* tasks of reads in short repeats searched as one lane per mapping instance */
  uint64_t huge_cover;        /* tasks whose selection scratch the last tier sized from the heap */
  uint64_t log_replays;       /* since the engine was created: rounds in which entries that found the grouped log full were redone */
  uint64_t log_replayed_entries; /* ... and how many entries those rounds redid */
} gmx_queue_counts;
int gmx_engine_queue_counts(gmx_engine *e, gmx_queue_counts *out);

/* The i-th master-generator draws for reads_per_file (quasimap.cpp:120-141: 5000 draws per batch of
 * <= 5000 reads, one mt19937(master_seed) shared by all files). `out` receives sum(reads_per_file) seeds. */
int gmx_master_seeds(uint32_t master_seed, const uint64_t *reads_per_file, uint64_t n_files, uint32_t *out);

typedef struct gmx_stats { /* QuasimapReadsStats, quasimap.hpp:17-24 */
  uint64_t all_reads_count;
  uint64_t skipped_reads_count;
  uint64_t missing_kmer_reads_count;
  uint64_t no_extension_reads_count;
  uint64_t exact_mapped_reads_count;
} gmx_stats;

/* The device side of the multi-GPU exchange (reads shard, the index is replicated, ONE exchange at the
 * end; then gmx_coverage_fetch on every rank or on rank 0). `fused` is one contiguous uint32 block: the accumulator
 * block (allele-sum, grouped and per-base counters interleaved per site so that a read's updates at a site share a
 * cache line) followed by 32 words for the five uint64 read counters as 16-bit limbs. A single all-reduce(sum) of
 * `n_fused` uint32 between gmx_coverage_reduce_begin (counters -> limbs) and gmx_coverage_reduce_end (limb sums ->
 * counters) is the whole exchange, exact for up to 65536 ranks. The three per-array pointers are NULL (the logical
 * arrays are not contiguous on the device; gmx_coverage_fetch gathers them); their lengths are the logical lengths. */
typedef struct gmx_device_coverage {
  void *allele_sum;  uint64_t n_allele_sum;
  void *per_base;    uint64_t n_per_base;
  void *grouped;     uint64_t n_grouped;
  void *stats;       uint64_t n_stats;      /* 5 x uint64 */
  void *fused;       uint64_t n_fused;      /* uint32 words */
} gmx_device_coverage;
int gmx_coverage_device(gmx_engine *e, gmx_device_coverage *out);
int gmx_coverage_reduce_begin(gmx_engine *e, void *hip_stream);
int gmx_coverage_reduce_end(gmx_engine *e, void *hip_stream);

/* Copies the raw uint32 totals to the host (any pointer may be NULL). The reference's uint16 semantics are
 * functions of these totals: allele-sum and grouped counts wrap (mod 65536, data_types.hpp:52), per-base
 * saturates at 65535 (allele_base.cpp:239). gmx_finalize_u16 applies them. */
int gmx_coverage_fetch(gmx_engine *e, uint32_t *allele_sum, uint32_t *per_base, uint32_t *grouped_dense,
                       gmx_stats *stats);
/* Grouped counts of sites with more than 8 alleles (grouped_allele_counts.cpp:17-49 for sites without dense slots).
 * Returns the number of uint32 words of the log (and copies it when it fits cap_words). The log is a sequence of records
 *   [site_index, n_ids, ids...]                                     worth +1, or
 *   [site_index, n_ids | GMX_LOG_COUNTED, count_lo, count_hi, ids...] worth +count,
 * ids ascending; the same (site, ids) may appear in several records (their values add up — concatenating the logs of
 * several engines is their sum). This function returns counted records, one per distinct (site, ids). */
#define GMX_LOG_COUNTED 0x80000000u
int64_t gmx_coverage_fetch_grouped_log(gmx_engine *e, uint32_t *out, uint64_t cap_words);
/* Adds the records of a grouped log (either form) to the engine's totals, after emptying them when `replace`: the
 * receiving end of a log exchange done outside the library (gramtools_amd/distributed.py under torch.distributed). */
int gmx_coverage_import_grouped_log(gmx_engine *e, const uint32_t *records, uint64_t n_words, int replace);
/* The receiving end of the log exchange as a pure function (used by gmx_group_allreduce / gmx_comm_allreduce_coverage after
 * their payload all-gather; exported for the tests): `gathered` = world slices of `pad` words, slice r holding sizes[r] words
 * of rank r's log (an empty rank: 0). Returns the length of the merged log (counted records, one per distinct (site, ids))
 * and copies it when it fits cap_words. */
int64_t gmx_grouped_log_merge_gathered(const uint32_t *gathered, const uint64_t *sizes, int world, uint64_t pad, uint32_t *out,
                                       uint64_t cap_words);
void gmx_finalize_u16(uint32_t *values, uint64_t n, int saturate);

/* ---- infer stage: genotyping from the recorded coverage, on the host ------------------------------
 * Replaces LevelGenotyper + the three writers `gram genotype` runs after quasimap (src/genotype/genotype.cpp:72-118,
 * src/genotype/infer/): level genotyping of every site, most nested first, with the reference's likelihood model,
 * genotype confidence percentiles (lib/GCP/GCP.h), nested-site invalidation and the AMBIG filter. Inputs are the raw
 * totals of gmx_coverage_fetch (+ the grouped log); uint16 semantics are applied inside. ploidy is 1 or 2. */
typedef struct gmx_infer gmx_infer;
int gmx_infer_run(const gmx_index *ix, const uint32_t *per_base_raw, const uint32_t *grouped_dense_raw,
                  const uint32_t *grouped_log, uint64_t n_log_words, double mean_cov_depth, double variance_cov_depth,
                  double mean_pb_error, int ploidy, gmx_infer **out);
void gmx_infer_destroy(gmx_infer *inf);
/* genotype/genotyped.json (jVCF; output_specs/make_json.cpp), genotype/genotyped.vcf.gz (BGZF; output_specs/make_vcf.cpp:
 * level-1 sites) and genotype/personalised_reference.fasta (personalised_reference.cpp; deduplicated by sequence as
 * genotype.cpp:16-21). coords_path = gram_dir/prg_coords.tsv (segment names and sizes) or NULL for one segment. */
int gmx_infer_write_json(const gmx_infer *inf, const char *coords_path, const char *sample_id, const char *out_path);
int gmx_infer_write_vcf(const gmx_infer *inf, const char *coords_path, const char *sample_id, const char *out_path);
int gmx_infer_write_fasta(const gmx_infer *inf, const char *coords_path, const char *description, const char *out_path);
/* The text of site_gtyping_debug_info.txt (`gram genotype --debug`; genotype/parameters.cpp:98, level_genotyping/runner.cpp:66-75):
 * one line per site in genotyping order. Returns the length, copies when it fits cap. */
int64_t gmx_infer_debug_text(const gmx_infer *inf, char *out, uint64_t cap);
/* One site as its jVCF object (POS 1-based in PRG coordinates); returns the length, copies when it fits cap. */
int64_t gmx_infer_site_json(const gmx_infer *inf, uint32_t site_index, char *out, uint64_t cap);
/* The likelihood model on explicit alleles and grouped counts (the known answers of tests/genotype/infer/
 * level_genotyping/test_model.cpp): the resulting site as JSON with the model's extra alleles and thresholds. */
int64_t gmx_infer_model(uint32_t n_alleles, const char *const *seqs, const uint32_t *pb_off, const uint32_t *pb_cov,
                        const int32_t *haplogroups, const uint8_t *callable, uint32_t n_groups, const uint32_t *group_off,
                        const int32_t *group_ids, const uint32_t *group_counts, int ploidy, double mean_cov, double var_cov,
                        double mean_pb_error, char *out, uint64_t cap);

/* Test hook: the pieces of the likelihood model one by one (ops documented at the definition, gmx_infer.cpp), as the
 * reference's unit tests call them (tests/genotype/infer/level_genotyping/test_model.cpp). JSON out. */
int64_t gmx_infer_debug(int op, uint32_t n_alleles, const char *const *seqs, const uint32_t *pb_off, const uint32_t *pb_cov,
                        const int32_t *haplogroups, const uint8_t *callable, uint32_t n_groups, const uint32_t *group_off,
                        const int32_t *group_ids, const uint32_t *group_counts, int ploidy, double mean_cov, double var_cov,
                        double mean_pb_error, const int32_t *ids, uint32_t n_ids, const double *lik, const uint32_t *lik_off,
                        const int32_t *lik_gt, uint32_t n_lik, char *out, uint64_t cap);

/* Test hook: the segment tracker of the writers (contig names and offsets from prg_coords.tsv), driven by a small script (see the
 * definition), as tests/genotype/infer/test_segment_tracker.cpp drives it. JSON list out. */
int64_t gmx_infer_segments_debug(const char *coords, const char *script, char *out, uint64_t cap);
/* Test hook: the allele extracter on a PRG with mock genotyped child sites (text formats at the definition, gmx_infer.cpp), as
 * tests/genotype/infer/test_allele_extracter.cpp drives it. JSON out: [[sequence, [per-base], haplogroup, callable], ...]. */
int64_t gmx_infer_extract_debug(const gmx_index *ix, int op, uint32_t site_index, const uint32_t *per_base_raw, const char *existing,
                                const char *mocks, char *out, uint64_t cap);

/* ---- several GPUs: reads shard, the index is replicated, one exchange at the end ----------------------
 * Replaces the OpenMP loop over reads with shared coverage structures (quasimap.cpp:90-118; omp atomic / omp critical
 * in coverage/allele_sum.cpp:31-43 and grouped_allele_counts.cpp:17-49). Every GPU has an engine of its own; after the
 * last read the uint32 totals are summed: ONE RCCL all-reduce of each engine's fused block (the five read counters ride
 * in it as 16-bit limbs) plus the exchange of the grouped log (counted records) when the PRG has sites with more than
 * 8 alleles (the dense limit, GMX_GROUPED_DENSE_MAX_ALLELES). Afterwards every engine holds the totals of the whole job: gmx_coverage_fetch on any of them. */
typedef struct gmx_group gmx_group; /* N engines in ONE process, one per listed device (the `gram` executable) */
/* GPUs visible to this process (0 without one). `gram genotype` without --device / --devices takes all of them when the
 * reads files are large enough for sharding to pay (the front-end passes a fixed argument list, common.py:33-49, so the
 * unmodified Python command scales with the node). */
int gmx_device_count(void);
/* Brings the HIP runtime, the device's context and the library's kernels up (150-250 ms): `gram genotype` does it on a
 * thread of its own beside the index load. GMX_ENODEV without a usable device. */
int gmx_device_warmup(int device);
/* The engines are created side by side, one host thread per device (the index upload of a whole-genome PRG is minutes). */
int gmx_group_create(const gmx_index *ix, const gmx_engine_opts *opts, const int *devices, int n_devices, gmx_group **out);
void gmx_group_destroy(gmx_group *g);
int gmx_group_size(const gmx_group *g);
gmx_engine *gmx_group_engine(gmx_group *g, int i);
int gmx_group_uses_rccl(const gmx_group *g); /* 0: peer copies + add kernel (no RCCL, or two engines on one device) */
/* Deals the reads of this call out by read index — engine i maps a contiguous range, one host thread per engine — with
 * the caller's per-read seeds (the master stream is global, so the result does not depend on the number of GPUs). */
int gmx_group_map_reads_host(gmx_group *g, const uint8_t *reads, const uint64_t *offsets, const uint32_t *seeds,
                             uint64_t n_reads);
int gmx_group_map_reads_packed_host(gmx_group *g, const uint64_t *planes, const uint64_t *offsets, uint32_t uniform_len,
                                    const uint32_t *seeds, const uint8_t *skip, uint64_t n_reads);
int gmx_group_sync_uploads(gmx_group *g);
int gmx_group_allreduce(gmx_group *g); /* the exchange; synchronises every engine first */

typedef struct gmx_comm gmx_comm; /* one engine per PROCESS (torch.distributed.run, mpirun, ...): rank 0 makes the id,
                                     the launcher's own channel broadcasts its 128 bytes, every rank creates its comm */
int gmx_comm_unique_id(uint8_t *out128);
int gmx_comm_create(const uint8_t *id128, int world_size, int rank, gmx_engine *e, gmx_comm **out);
void gmx_comm_destroy(gmx_comm *c);
/* The same exchange, enqueued on hip_stream (the log part, if any, synchronises that stream). */
int gmx_comm_allreduce_coverage(gmx_comm *c, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
