"""The C-ABI library loads and exports every symbol include/gmx.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from gramtools_amd import _lib, Index, Quasimapper, GmxError, master_seeds
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB) if os.path.exists(_lib.LIB) else _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_master_seeds_match_oracle():
    assert master_seeds(42, [3, 5001]).tolist() == Oracle.master_seeds(42, [3, 5001]).tolist()


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ix = Index([1, 2, 5, 1, 6, 2, 6, 3, 4], 2)
    with pytest.raises(GmxError) as e:
        Quasimapper(ix)
    assert e.value.code in (-2, -3)  # GMX_ENODEV / GMX_EHIP: no CPU fallback exists
    from gramtools_amd import Ingest
    with pytest.raises(GmxError) as e:  # the device-side reads decoder likewise: nothing is inflated or parsed on the host behind this API
        Ingest(max_text_bytes=1 << 20)
    assert e.value.code in (-2, -3)


@pytest.mark.gpu
def test_page_locked_blocks_are_kept_and_handed_out_again():
    """gmx_host_alloc / gmx_host_free: a freed page-locked block of 1 MB or more serves the next request it fits (the
    reads feed allocates its block buffers per file); small blocks are really freed."""
    lib = _lib.load()
    a = lib.gmx_host_alloc(8 << 20)
    assert a
    lib.gmx_host_free(a)
    b = lib.gmx_host_alloc(6 << 20)  # fits the spare block (at most twice the request + 1 MB)
    assert b == a
    c = lib.gmx_host_alloc(6 << 20)  # no spare left: a new block
    assert c and c != b
    lib.gmx_host_free(b)
    lib.gmx_host_free(c)
    d = lib.gmx_host_alloc(1 << 10)  # far smaller than any spare: not served from them
    assert d and d not in (b, c)
    lib.gmx_host_free(d)


@pytest.mark.gpu
def test_reserve_sizes_the_workspace_ahead_of_the_first_call():
    import ctypes as C
    from gramtools_amd import Index, Quasimapper, master_seeds
    from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg
    from common import canonical_cov
    ref = random_ref(20000, 3)
    prg, pos, alts, n_alts = snp_prg(ref, 200, 4)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, 3000, 150, 5)
    seeds = master_seeds(1, [3000])
    ix = Index(prg, 7)
    a = Quasimapper(ix)
    a.map_reads(reads.reshape(-1), flat_offsets(3000, 150), seeds)
    b = Quasimapper(ix)
    assert b.lib.gmx_engine_reserve(b.h, 5000, 5000 * 150) == 0
    assert b.lib.gmx_engine_reserve(b.h, 100, 100) == 0  # smaller: nothing to do
    b.map_reads(reads.reshape(-1), flat_offsets(3000, 150), seeds)
    assert canonical_cov(a.coverage()) == canonical_cov(b.coverage())


def test_two_bit_stream_packer_argument_checks():
    """gmx_pack_reads_2bit / gmx_twobit_units (host side of gmx_map_reads_2bit_host): sizes and GMX_EINVAL, no device needed."""
    import ctypes as C
    import numpy as np
    from gramtools_amd import _lib
    lib = _lib.load()
    offs = np.array([0, 150, 300, 450], dtype=np.uint64)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    assert lib.gmx_twobit_units(p64(offs), 0, 3) == (450 + 31) // 32 + 1
    assert lib.gmx_twobit_units(None, 150, 3) == (450 + 31) // 32 + 1
    reads = np.ones(450, dtype=np.uint8)
    out = np.zeros(20, dtype=np.uint64)
    assert lib.gmx_pack_reads_2bit(reads.ctypes.data, offs.ctypes.data, 150, 3, out.ctypes.data, None, 1) == 0
    assert lib.gmx_pack_reads_2bit(reads.ctypes.data, offs.ctypes.data, 151, 3, out.ctypes.data, None, 1) == -1   # GMX_EINVAL: another length
    assert b"another length" in lib.gmx_last_error()
    assert lib.gmx_pack_reads_2bit(None, offs.ctypes.data, 150, 3, out.ctypes.data, None, 1) == -1
    assert lib.gmx_map_reads_2bit_host(None, out.ctypes.data, None, 150, out.ctypes.data, None, 3) == -1
    assert lib.gmx_engine_seeds_in_place(None, 1) == -1
    # the device-plane feed and the device-side reads decoder: argument checks come before any device call
    assert lib.gmx_map_reads_packed_device(None, out.ctypes.data, None, 150, out.ctypes.data, None, 3) == -1
    assert lib.gmx_ingest_reset(None) == -1 and lib.gmx_ingest_max_text(None) == 0
    import ctypes as C
    from gramtools_amd import _lib as L
    res = L.IngestResult()
    assert lib.gmx_ingest_wait(None, 0, C.byref(res)) == -1 and lib.gmx_ingest_submit_text(None, 0, None, 0, 1) == -1
    h = C.c_void_p()
    assert lib.gmx_ingest_create(0, 1000, C.byref(h)) == -1 and b"64 KB" in lib.gmx_last_error()   # GMX_EINVAL before the device is looked for


def test_every_exported_function_is_guarded_against_exceptions():
    """include/gmx.h: "no C++ exception leaves the library". Every function the header declares is either closed by a GMX_GUARD_*
    macro (a function-try-block, gmx_internal.h) or is one of the one-line accessors that cannot throw."""
    csrc = os.path.join(ROOT, "gramtools_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".hip", ".cpp")))
    guarded = set(re.findall(r'GMX_GUARD_(?:INT|VOID|PTR|ZERO)\("(gmx_[a-z0-9_]+)"\)', text))
    trivial = {"gmx_last_error", "gmx_index_destroy", "gmx_infer_destroy", "gmx_group_size", "gmx_group_engine", "gmx_group_uses_rccl",
               "gmx_ingest_max_text", "gmx_ingest_max_compressed", "gmx_ingest_max_members", "gmx_engine_second_stream", "gmx_debug_fail_alloc"}
    missing = [n for n in declared_symbols() if n not in guarded and n not in trivial]
    assert not missing, missing
    for n in trivial & set(declared_symbols()):  # the accessors really are one-liners
        m = re.search(r"\b" + n + r"\([^)]*\)\s*(?:try\s*)?\{([^\n]*)", text)
        assert m, n
