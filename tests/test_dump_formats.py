"""Coverage file formats of the Python mirror against the reference's dump tests
(tests/genotype/quasimap/coverage/test_allele_base.cpp:12-46, test_grouped_allele_counts.cpp:171-242)."""
import json

import numpy as np

from gramtools_amd import Index, Coverage, QuasimapReadsStats, dump_allele_base, dump_allele_sum, dump_grouped_allele_counts
from oracle import encode_prg, prg_string_to_ints


def _cov(prg, pb=None, asum=None, grouped=None, log=None):
    ix = Index(prg, 1)
    z = lambda n: np.zeros(max(n, 0), dtype=np.uint32)
    a = z(ix.info.n_allele_slots) if asum is None else np.asarray(asum, dtype=np.uint32)
    p = z(ix.info.n_per_base_slots) if pb is None else np.asarray(pb, dtype=np.uint32)
    g = z(ix.info.n_grouped_slots) if grouped is None else np.asarray(grouped, dtype=np.uint32)
    l = np.zeros(0, dtype=np.uint32) if log is None else np.asarray(log, dtype=np.uint32)
    return ix, Coverage(ix, a, p, g, l, QuasimapReadsStats())


def test_allele_base_json_layout():
    # SitesAlleleBaseCoverage {{1,12},{0,3,0}}, {{0},{0,19,0}} -> test_allele_base.cpp:12-28
    ix, cov = _cov(encode_prg("a5gg6ccc6t7c8ggg8"), pb=[1, 12, 0, 3, 0, 0, 0, 19, 0])
    assert dump_allele_base(cov) == '{"allele_base_counts":[[[1,12],[0,3,0]],[[0],[0,19,0]]]}\n'


def test_allele_base_json_empty_for_nested_prg():
    ix, cov = _cov(prg_string_to_ints("[ac[tg,cc]t,t]a"))
    assert dump_allele_base(cov) == '{"allele_base_counts":[]}\n'


def test_per_base_saturates_and_allele_sum_wraps():
    ix, cov = _cov(encode_prg("a5g6c6t"), pb=[70000, 65535], asum=[65536 + 7, 65535])
    assert dump_allele_base(cov) == '{"allele_base_counts":[[[65535],[65535]]]}\n'
    assert dump_allele_sum(cov) == "7 65535\n"


def test_grouped_json_layout():
    # dense slots of a 2-allele site: mask 1 = {0}, 2 = {1}, 3 = {0,1}
    ix, cov = _cov(encode_prg("a5g6c6t7a8c8"), grouped=[2, 0, 19, 0, 5, 0])
    doc = json.loads(dump_grouped_allele_counts(cov))["grouped_allele_counts"]
    groups = {g: tuple(ids) for g, ids in doc["allele_groups"].items()}
    got = [{groups[g]: c for g, c in site.items()} for site in doc["site_counts"]]
    assert got == [{(0,): 2, (0, 1): 19}, {(1,): 5}]
    assert dump_grouped_allele_counts(cov).startswith('{"grouped_allele_counts":{"allele_groups":{')
    assert " " not in dump_grouped_allele_counts(cov)


def test_grouped_log_sites_with_many_alleles(monkeypatch):
    monkeypatch.setenv("GMX_DENSE_MAX_ALLELES", "5")  # (sites of up to 8 alleles have dense group counters by default)
    ix, cov = _cov(prg_string_to_ints("a[a,c,g,t,aa,cc]t"), log=[0, 2, 1, 4, 0, 2, 1, 4, 0, 1, 5])
    assert int(ix.grouped_off[0]) == 0xFFFFFFFF
    assert cov.grouped_allele_counts == [{(1, 4): 2, (5,): 1}]
