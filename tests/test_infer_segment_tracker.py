"""The segment tracker behind the jVCF / VCF / FASTA writers (contig of a PRG position from prg_coords.tsv) against the
reference's unit tests (libgramtools/tests/genotype/infer/test_segment_tracker.cpp, transcribed; no reference code)."""
import ctypes as C
import json

import pytest

from gramtools_amd import _lib, GmxError
from gramtools_amd._lib import check

COORDS = "chr1\t2200\nchr2\t400\n"


def run(coords, script):
    lib = _lib.load()
    n = check(lib.gmx_infer_segments_debug(coords.encode(), script.encode(), None, 0))
    buf = C.create_string_buffer(n + 1)
    check(lib.gmx_infer_segments_debug(coords.encode(), script.encode(), buf, n + 1))
    return json.loads(buf.value.decode())


def test_no_coords_default_id():  # :22-27
    assert run("", "id 1000;id 40000") == ["gramtools_prg", "gramtools_prg"]


def test_beyond_the_last_segment_fails():  # :29-31
    with pytest.raises(GmxError):
        run(COORDS, "id 40000")


def test_boundary_is_the_next_segment_and_no_backward_queries():  # :33-40
    assert run(COORDS, "id 2200") == ["chr2"]
    with pytest.raises(GmxError):
        run(COORDS, "id 2200;id 200")


def test_valid_queries():  # :42-51
    assert run(COORDS, "global_edge;edge;id 400;id 2500;edge") == [2599, 2199, "chr1", "chr2", 2599]


def test_reset_allows_querying_again():  # :53-58
    assert run(COORDS, "id 2500;reset;id 100") == ["chr2", None, "chr1"]


def test_relative_position():  # :60-64
    assert run(COORDS, "id 2500;rel 2500") == ["chr2", 300]
