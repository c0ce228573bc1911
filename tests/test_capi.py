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
