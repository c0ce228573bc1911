"""ctypes loader for libgmx.so (include/gmx.h). Fails loudly if the HIP library is missing."""
import ctypes as C
import os

from .build import LIB, build_library


class GmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gmx error {code}: {msg}")
        self.code = code


class IndexInfo(C.Structure):
    _fields_ = [("n_text", C.c_uint64), ("kmer_size", C.c_uint32), ("n_sites", C.c_uint32), ("is_nested", C.c_uint32),
                ("n_allele_slots", C.c_uint32), ("n_per_base_slots", C.c_uint32), ("n_grouped_slots", C.c_uint32),
                ("n_nodes", C.c_uint32), ("n_kmers_present", C.c_uint64), ("index_bytes", C.c_uint64),
                ("kmer_size2", C.c_uint32), ("n_inline_sites", C.c_uint32), ("seed_shift", C.c_uint32),
                ("n_jump_sites", C.c_uint32), ("n_seed_words", C.c_uint64)]


class EngineOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("rng_mode", C.c_int), ("max_states", C.c_uint32),
                ("max_path_nodes", C.c_uint32), ("max_batch_reads", C.c_uint64), ("forward_only", C.c_int),
                ("huge_heap_bytes", C.c_uint64), ("log_cap_words", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("all_reads_count", C.c_uint64), ("skipped_reads_count", C.c_uint64),
                ("missing_kmer_reads_count", C.c_uint64), ("no_extension_reads_count", C.c_uint64),
                ("exact_mapped_reads_count", C.c_uint64)]


class QueueCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("mapped", "alive", "dead", "overflow_probe", "overflow_extend", "big_mapped",
                                          "cover_general", "cover_mid", "cover_overflow", "seed_cursor", "huge_search",
                                          "inst_mapped", "huge_cover", "log_replays", "log_replayed_entries")]


class StockReport(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("kmers", "states", "kmer_mismatches", "kmers_missing_in_files", "duplicate_kmers",
                                          "mask_bits", "mask_mismatches", "cov_graph_state", "cov_graph_library_version",
                                          "cov_graph_sites", "fm_index_bytes")]


class Timing(C.Structure):
    _fields_ = [("search_ms", C.c_double), ("search_launches", C.c_uint64), ("cover_ms", C.c_double),
                ("cover_launches", C.c_uint64), ("reads", C.c_uint64), ("kernel_ms", C.c_double * 8),
                ("kernel_launches", C.c_uint64 * 8)]


class BgzfMember(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("size", C.c_uint32), ("isize", C.c_uint32), ("crc32", C.c_uint32), ("reserved", C.c_uint32)]


class IngestResult(C.Structure):
    _fields_ = [("status", C.c_uint32), ("bad_member", C.c_uint32), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64),
                ("n_pairs", C.c_uint64), ("uniform_len", C.c_uint32), ("any_skip", C.c_uint32), ("text_bytes", C.c_uint64),
                ("consumed_bytes", C.c_uint64), ("tail_bytes", C.c_uint64), ("sub_pairs", C.c_uint64 * 16),
                ("d_planes", C.c_void_p), ("d_offsets", C.c_void_p), ("d_skip", C.c_void_p)]


class DepthStats(C.Structure):
    _fields_ = [("mean_cov_depth", C.c_double), ("variance_cov_depth", C.c_double), ("num_sites_noCov", C.c_uint64),
                ("num_sites_total", C.c_uint64)]


class DeviceCoverage(C.Structure):
    _fields_ = [("allele_sum", C.c_void_p), ("n_allele_sum", C.c_uint64), ("per_base", C.c_void_p),
                ("n_per_base", C.c_uint64), ("grouped", C.c_void_p), ("n_grouped", C.c_uint64),
                ("stats", C.c_void_p), ("n_stats", C.c_uint64), ("fused", C.c_void_p), ("n_fused", C.c_uint64)]


# every symbol include/gmx.h declares: (restype, argtypes)
_vp, _u64, _u32, _i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64
_u8p, _u32p, _i32p, _u64p, _i64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_int64))
SYMBOLS = {
    "gmx_last_error": (C.c_char_p, []),
    "gmx_index_build": (C.c_int, [_u32p, _u64, _u32, C.c_int, C.POINTER(_vp)]),
    "gmx_index_build_from_file": (C.c_int, [C.c_char_p, _u32, C.c_int, C.POINTER(_vp)]),
    "gmx_index_save": (C.c_int, [_vp, C.c_char_p]),
    "gmx_index_load": (C.c_int, [C.c_char_p, C.c_char_p, _u32, C.POINTER(_vp)]),
    "gmx_index_destroy": (None, [_vp]),
    "gmx_index_get_info": (C.c_int, [_vp, C.POINTER(IndexInfo)]),
    "gmx_index_site_layout": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u32p, _i32p]),
    "gmx_index_per_base_layout": (_i64, [_vp, _u32p, _u64]),
    "gmx_index_allele_base_layout": (C.c_int, [_vp, _u32p, _u32p]),
    "gmx_compute_coverage_depth": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u64, C.POINTER(DepthStats)]),
    "gmx_index_bubble_order": (C.c_int, [_vp, _u32p]),
    "gmx_debug_suffix_array_u16": (C.c_int, [C.POINTER(C.c_uint16), _u64, C.POINTER(C.c_uint16)]),
    "gmx_index_copy_sa": (C.c_int, [_vp, _u32p]),
    "gmx_index_copy_bwt": (C.c_int, [_vp, _u32p]),
    "gmx_index_rank": (_u32, [_vp, _u32, _u32]),
    "gmx_index_copy_pos_info": (C.c_int, [_vp, _i64p]),
    "gmx_index_copy_target_map": (_i64, [_vp, _i64p, _u64]),
    "gmx_index_seed_states": (_i64, [_vp, _u8p, _i64p, _u64]),
    "gmx_index_seed_states_k": (_i64, [_vp, _u8p, _u32, _i64p, _u64]),
    "gmx_index_jump_states": (_i64, [_vp, _u32, _u32, _i64p, _u64]),
    "gmx_engine_default_opts": (None, [C.POINTER(EngineOpts)]),
    "gmx_engine_create": (C.c_int, [_vp, C.POINTER(EngineOpts), C.POINTER(_vp)]),
    "gmx_engine_destroy": (None, [_vp]),
    "gmx_engine_reset": (C.c_int, [_vp]),
    "gmx_engine_reset_async": (C.c_int, [_vp, _vp]),
    "gmx_map_reads_host": (C.c_int, [_vp, _u8p, _u64p, _u32p, _u64]),
    "gmx_map_reads_device": (C.c_int, [_vp, _vp, _vp, _vp, _u64, _u64, _vp]),
    "gmx_map_reads_packed_host": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u64]),
    "gmx_engine_sync_uploads": (C.c_int, [_vp]),
    "gmx_engine_seeds_in_place": (C.c_int, [_vp, C.c_int]),
    "gmx_packed_pairs": (_u64, [_u64p, _u32, _u64]),
    "gmx_twobit_units": (_u64, [_u64p, _u32, _u64]),
    "gmx_pack_reads_2bit": (C.c_int, [_vp, _vp, _u32, _u64, _vp, _vp, C.c_int]),
    "gmx_map_reads_2bit_host": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u64]),
    "gmx_map_reads_packed_device": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u64]),
    "gmx_ingest_create": (C.c_int, [C.c_int, _u64, C.POINTER(_vp)]),
    "gmx_ingest_destroy": (None, [_vp]),
    "gmx_ingest_max_text": (_u64, [_vp]),
    "gmx_ingest_max_compressed": (_u64, [_vp]),
    "gmx_ingest_reset": (C.c_int, [_vp]),
    "gmx_ingest_submit_bgzf": (C.c_int, [_vp, C.c_int, _vp, _u64, C.POINTER(BgzfMember), _u64, C.c_int]),
    "gmx_ingest_submit_text": (C.c_int, [_vp, C.c_int, _vp, _u64, C.c_int]),
    "gmx_ingest_submit_text_deferred": (C.c_int, [_vp, C.c_int, _vp, _u64]),
    "gmx_ingest_max_members": (_u64, [_vp]),
    "gmx_ingest_wait": (C.c_int, [_vp, C.c_int, C.POINTER(IngestResult)]),
    "gmx_ingest_submit_bgzf_deferred": (C.c_int, [_vp, C.c_int, _vp, _u64, C.POINTER(BgzfMember), _u64]),
    "gmx_ingest_scan": (C.c_int, [_vp, C.c_int, _vp, _u64, C.c_int]),
    "gmx_ingest_fetch_tail": (_i64, [_vp, C.c_int, _vp, _u64]),
    "gmx_ingest_release_after": (C.c_int, [_vp, C.c_int, _vp]),
    "gmx_ingest_fetch_text": (_i64, [_vp, C.c_int, _vp, _u64]),
    "gmx_ingest_fetch_reads": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "gmx_pack_reads": (C.c_int, [_vp, _vp, _u32, _u64, _vp, _vp, C.c_int]),
    "gmx_engine_reserve": (C.c_int, [_vp, _u64, _u64]),
    "gmx_engine_reserve_packed": (C.c_int, [_vp, _u64, _u64]),
    "gmx_host_alloc": (_vp, [_u64]),
    "gmx_host_free": (None, [_vp]),
    "gmx_engine_sync": (C.c_int, [_vp]),
    "gmx_engine_enable_timing": (C.c_int, [_vp, C.c_int]),
    "gmx_engine_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "gmx_engine_queue_counts": (C.c_int, [_vp, C.POINTER(QueueCounts)]),
    "gmx_stock_read_int_vector": (_i64, [C.c_char_p, _u32, _u64p, _u64, C.POINTER(C.c_uint32)]),
    "gmx_stock_write_int_vector": (C.c_int, [C.c_char_p, _u64p, _u64, _u32, C.c_int]),
    "gmx_index_write_stock_files": (C.c_int, [_vp, C.c_char_p]),
    "gmx_index_check_stock_files": (C.c_int, [_vp, C.c_char_p, C.POINTER(StockReport)]),
    "gmx_engine_second_stream": (_vp, [_vp]),
    "gmx_engine_debug_keep_states": (C.c_int, [_vp, C.c_int]),
    "gmx_debug_fail_alloc": (C.c_uint64, [C.c_int64]),
    "gmx_debug_final_states": (C.c_int, [_vp, _u64, _u32p, _u64, _u64p, C.POINTER(C.c_int)]),
    "gmx_debug_search": (C.c_int, [_vp, _u8p, _u32, C.c_int, _u32p, _u64, _u32, _u32, C.c_int, _u32p, _u64, _u64p]),
    "gmx_debug_encapsulate": (C.c_int, [_vp, _u32p, _u64, _u32p, _u64, _u64p, _u32p, _u64, _u64p]),
    "gmx_master_seeds": (C.c_int, [_u32, _u64p, _u64, _u32p]),
    "gmx_coverage_device": (C.c_int, [_vp, C.POINTER(DeviceCoverage)]),
    "gmx_coverage_reduce_begin": (C.c_int, [_vp, _vp]),
    "gmx_coverage_reduce_end": (C.c_int, [_vp, _vp]),
    "gmx_coverage_fetch": (C.c_int, [_vp, _u32p, _u32p, _u32p, C.POINTER(Stats)]),
    "gmx_coverage_fetch_grouped_log": (_i64, [_vp, _u32p, _u64]),
    "gmx_coverage_import_grouped_log": (C.c_int, [_vp, _u32p, _u64, C.c_int]),
    "gmx_grouped_log_merge_gathered": (_i64, [_vp, _vp, C.c_int, _u64, _vp, _u64]),
    "gmx_finalize_u16": (None, [_u32p, _u64, C.c_int]),
    "gmx_infer_run": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u64, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(_vp)]),
    "gmx_infer_destroy": (None, [_vp]),
    "gmx_infer_write_json": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p]),
    "gmx_infer_write_vcf": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p]),
    "gmx_infer_write_fasta": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p]),
    "gmx_infer_debug_text": (_i64, [_vp, C.c_char_p, _u64]),
    "gmx_infer_site_json": (_i64, [_vp, _u32, C.c_char_p, _u64]),
    "gmx_infer_model": (_i64, [_u32, C.POINTER(C.c_char_p), _u32p, _u32p, _i32p, _u8p, _u32, _u32p, _i32p, _u32p, C.c_int,
                               C.c_double, C.c_double, C.c_double, C.c_char_p, _u64]),
    "gmx_infer_debug": (_i64, [C.c_int, _u32, C.POINTER(C.c_char_p), _u32p, _u32p, _i32p, _u8p, _u32, _u32p, _i32p, _u32p, C.c_int,
                               C.c_double, C.c_double, C.c_double, _i32p, _u32, C.POINTER(C.c_double), _u32p, _i32p, _u32, C.c_char_p, _u64]),
    "gmx_infer_segments_debug": (_i64, [C.c_char_p, C.c_char_p, C.c_char_p, _u64]),
    "gmx_infer_extract_debug": (_i64, [_vp, C.c_int, _u32, _u32p, C.c_char_p, C.c_char_p, C.c_char_p, _u64]),
    "gmx_device_count": (C.c_int, []),
    "gmx_device_warmup": (C.c_int, [C.c_int]),
    "gmx_group_create": (C.c_int, [_vp, C.POINTER(EngineOpts), C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "gmx_group_destroy": (None, [_vp]),
    "gmx_group_size": (C.c_int, [_vp]),
    "gmx_group_engine": (_vp, [_vp, C.c_int]),
    "gmx_group_uses_rccl": (C.c_int, [_vp]),
    "gmx_group_map_reads_host": (C.c_int, [_vp, _u8p, _u64p, _u32p, _u64]),
    "gmx_group_map_reads_packed_host": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u64]),
    "gmx_group_sync_uploads": (C.c_int, [_vp]),
    "gmx_group_allreduce": (C.c_int, [_vp]),
    "gmx_comm_unique_id": (C.c_int, [_u8p]),
    "gmx_comm_create": (C.c_int, [_u8p, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "gmx_comm_destroy": (None, [_vp]),
    "gmx_comm_allreduce_coverage": (C.c_int, [_vp, _vp]),
}

_lib = None


def load(build=True):
    """Load libgmx.so (building it in-tree with hipcc when absent). Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GMX_LIB", LIB)  # (tests: another build of the same sources, e.g. lib/libgmx_alt.so)
    if not os.path.exists(path):
        if not build or path != LIB:
            raise ImportError(f"{path} is missing: run `python -m gramtools_amd.build` (needs hipcc)")
        build_library()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(lib, name)  # AttributeError here = the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise GmxError(rc, load().gmx_last_error().decode(errors="replace"))
    return rc
