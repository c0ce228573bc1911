"""Known answers of the reference's unit tests for the coverage dump formats, the grouped-count bookkeeping, the per-base
error rate of read_stats.json and the one binary PRG its tests hold (tests/golden/dumps_and_stats.json), against the
product's host side: the Python mirror of the writers (gramtools_amd.quasimap — `gram`'s C++ writers are compared with it
file for file in tests/test_dump_formats.py and tests/test_gram_cli.py), the `gram` executable, the index loader."""
import os
import subprocess

import numpy as np
import pytest

from golden_runner import host_cases, GOLDEN_DIR
from gramtools_amd import (Index, allele_base_json, hash_allele_groups, group_id_counts, group_id_alleles, grouped_json)
from gramtools_amd.build import build_gram
from oracle import Oracle
from oracle.prg_text import ints_to_prg_string

CASES = host_cases()


def _sites(spec):
    return [{tuple(ids): c for ids, c in site} for site in spec]


def _hash(spec):
    return {tuple(ids): g for ids, g in spec}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_host_side_known_answers(case, tmp_path):
    kind = case["kind"]
    if kind == "allele_base_json":
        assert allele_base_json(case["sites"]) == case["expect"]
    elif kind == "hash_allele_groups":
        h = hash_allele_groups(_sites(case["sites"]))
        assert set(h) == {tuple(g) for g in case["expect_groups"]}
        assert sorted(h.values()) == case["expect_ids"]  # distinct and 'full': from 0, increasing by one
    elif kind == "group_id_counts":
        got = group_id_counts(_sites(case["sites"]), _hash(case["hash"]))
        assert [list(d.items()) for d in got] == [[tuple(kv) for kv in site] for site in case["expect"]]
    elif kind == "group_id_alleles":
        got = group_id_alleles(_hash(case["hash"]))
        assert [[k, v] for k, v in got.items()] == case["expect"]
    elif kind == "grouped_json":
        got = grouped_json(_sites(case["sites"]), _hash(case["hash"]))
        if "expect" in case:
            assert got == case["expect"]
        else:
            assert case["expect_part"] in got
    elif kind == "read_stats":
        fq = tmp_path / "r.fastq"
        fq.write_text(case["fastq"])
        out = subprocess.run([build_gram(), "_read_stats", str(fq)], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 0, out.stdout
        got = dict(kv.split("=") for kv in out.stdout.split())
        for key, want in case["expect"].items():
            if isinstance(want, float):
                assert float(got[key]) == pytest.approx(want, rel=1e-6)  # EXPECT_FLOAT_EQ
            else:
                assert int(got[key]) == want
    elif kind == "prg_file":
        path = os.path.join(GOLDEN_DIR, case["file"])
        ix = Index(path, 3)  # gmx_index_build_from_file: the reference's own binary PRG (little-endian uint32 per symbol)
        sa, bwt = ix.sa(), ix.bwt()
        text = np.zeros(len(sa) - 1, dtype=np.uint32)  # BWT[i] = text[SA[i] - 1]: the symbols the index holds
        for i in range(len(sa)):
            if sa[i] > 0:
                text[sa[i] - 1] = bwt[i]
        assert ints_to_prg_string(text) == case["expect"]
        ints = np.fromfile(path, dtype="<u4")
        assert text.tolist() == ints.tolist()
        o = Oracle(ints.tolist(), 3)
        assert o.num_sites() == ix.info.n_sites == 5 and o.is_nested() and ix.info.is_nested


def test_golden_set_size():
    """Every known-answer case transcribed from the reference's suites (>= 150)."""
    from golden_runner import all_cases
    assert len(all_cases()) + len(CASES) >= 150
