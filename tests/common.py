"""Shared helpers of the parity tests: canonical coverage views of the oracle, the HIP engine and the
test-only host emulation, so that they can be compared with ==."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle import Oracle
from gramtools_amd import Index, Coverage, QuasimapReadsStats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def flatten_reads(reads):
    reads = [np.asarray(r, dtype=np.uint8) for r in reads]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    flat = np.concatenate(reads) if reads else np.zeros(0, dtype=np.uint8)
    return flat, offs


def canonical_oracle(o: Oracle):
    pb = {n["first_pos"]: n["cov"] for n in o.per_base_nodes()}
    # depth: the Read_depth block of read_stats.json (read_stats.cpp:119-160) — doubles summed in bubble_map order on
    # both sides, so equality is exact
    return dict(allele_sum=o.allele_sum(), grouped=o.grouped(), per_base=pb, stats=o.stats(),
                allele_base=o.allele_base_non_nested(), depth=o.depth_stats())


def canonical_cov(cov: Coverage):
    return dict(allele_sum=cov.allele_sum_coverage, grouped=cov.grouped_allele_counts, per_base=cov.per_base_by_first_pos(),
                stats=cov.stats.as_dict(), allele_base=cov.allele_base_coverage, depth=cov.depth_stats())


def oracle_map(prg, k, reads, seeds, rng_mode=0, threads=1, kmers_of_reads=False):
    """kmers_of_reads: index only the k-mers of these (equal-length, clean) reads instead of all 4^k (k = 14)."""
    o = Oracle(prg, k, all_kmers=not kmers_of_reads, rng_mode=rng_mode)
    if isinstance(reads, tuple):  # (flat, offsets): ragged reads, possibly with errors and Ns
        flat, offs = reads
        if kmers_of_reads:
            o.index_kmers_of_read_list(flat, offs)
    else:
        if kmers_of_reads:
            o.index_kmers_of_reads(np.asarray(reads, dtype=np.uint8))
        flat, offs = flatten_reads(reads)
    o.map_reads(flat, offs, seeds, threads=threads)
    return canonical_oracle(o)


# ---------------------------------------------------------------------------------------------
# test-only host emulation of the device logic (tests/hostemu/hostemu.cpp)
# ---------------------------------------------------------------------------------------------
_emu = None


def _load_emu():
    global _emu
    if _emu is not None:
        return _emu
    so = os.path.join(HERE, "hostemu", "libhostemu.so")
    srcs = [os.path.join(HERE, "hostemu", "hostemu.cpp"), os.path.join(ROOT, "gramtools_amd", "csrc", "gmx_index.cpp")]
    deps = srcs + [os.path.join(ROOT, "gramtools_amd", "csrc", h) for h in
                   ("gmx_core.h", "gmx_cover.h", "gmx_dfs.h", "gmx_types.h", "gmx_index.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}.tmp"  # pytest-xdist workers may all find it stale: build aside, rename atomically
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + srcs + ["-lpthread"])
        os.replace(tmp, so)
    lib = C.CDLL(so)
    lib.hostemu_create.restype = C.c_void_p
    lib.hostemu_create.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint32, C.c_int, C.c_char_p, C.c_uint64]
    lib.hostemu_destroy.argtypes = [C.c_void_p]
    lib.hostemu_set_single_loci.argtypes = [C.c_void_p, C.c_uint32]
    lib.hostemu_set_wide.argtypes = [C.c_void_p, C.c_int]
    lib.hostemu_n_wide.restype = C.c_uint64
    lib.hostemu_n_wide.argtypes = [C.c_void_p]
    lib.hostemu_map.restype = C.c_int
    lib.hostemu_map.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64,
                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hostemu_set_stage.argtypes = [C.c_uint32]
    lib.hostemu_routes.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    lib.hostemu_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hostemu_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    _emu = lib
    return lib


def hostemu_map(prg, k, reads, seeds, rng_mode=0, fast_states=8, fast_arena=24, big_states=1024, big_arena=2048,
                return_raw=False, single_loci=None, wide=False, stats=None, stage=0):
    """Runs the device headers on the host. Returns (canonical coverage, n_overflow_tasks, rc)."""
    lib = _load_emu()
    arr = np.ascontiguousarray(prg, dtype=np.uint32)
    err = C.create_string_buffer(512)
    h = lib.hostemu_create(arr.ctypes.data_as(C.POINTER(C.c_uint32)), arr.size, k, rng_mode, err, 512)
    if not h:
        raise RuntimeError(err.value.decode())
    try:
        flat, offs = flatten_reads(reads)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.uint8)
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        if single_loci is not None:  # loci capacity of the nested single-instance routine (gmx_cover.h)
            lib.hostemu_set_single_loci(h, single_loci)
        lib.hostemu_set_stage(stage)  # operations gmx_cover_jump stages between its check pass and the recording (gmx_cover.h)
        if wide:  # single-instance tasks through gmx_cover_single_nested_wide (the routine of gmx_cover_one_kernel)
            lib.hostemu_set_wide(h, 1)
        rc = lib.hostemu_map(h, flat.ctypes.data_as(C.POINTER(C.c_uint8)), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                             s.ctypes.data_as(C.POINTER(C.c_uint32)), offs.size - 1, fast_states, fast_arena, big_states,
                             big_arena)
        if stats is not None:
            stats["n_wide"] = int(lib.hostemu_n_wide(h))
            routes = np.zeros(3, dtype=np.uint64)  # single-instance tasks by routine: walk-free, jump, walk (process-wide: reset here)
            lib.hostemu_routes(routes.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
            stats["routes"] = [int(x) for x in routes]
        sizes = np.zeros(7, dtype=np.uint64)
        lib.hostemu_sizes(h, sizes.ctypes.data_as(C.POINTER(C.c_uint64)))
        a = np.zeros(max(int(sizes[0]), 1), dtype=np.uint32)
        p = np.zeros(max(int(sizes[1]), 1), dtype=np.uint32)
        g = np.zeros(max(int(sizes[2]), 1), dtype=np.uint32)
        lg = np.zeros(max(int(sizes[3]), 1), dtype=np.uint32)
        st = np.zeros(5, dtype=np.uint64)
        lib.hostemu_fetch(h, *(x.ctypes.data_as(C.POINTER(C.c_uint32)) for x in (a, p, g, lg)),
                          st.ctypes.data_as(C.POINTER(C.c_uint64)))
        raw = dict(allele_sum=a[:int(sizes[0])], per_base=p[:int(sizes[1])], grouped=g[:int(sizes[2])],
                   grouped_log=lg[:int(sizes[3])], stats=st)
        if return_raw:
            return raw, (int(sizes[4]), int(sizes[5]), int(sizes[6])), rc
        ix = Index(prg, k, threads=1)
        cov = Coverage(ix, raw["allele_sum"], raw["per_base"], raw["grouped"], raw["grouped_log"],
                       QuasimapReadsStats(*(int(x) for x in st)))
        return canonical_cov(cov), (int(sizes[4]), int(sizes[5]), int(sizes[6])), rc
    finally:
        lib.hostemu_destroy(h)
