"""The likelihood model's pieces, one by one, against the reference's own unit tests (inputs and expectations transcribed from
libgramtools/tests/genotype/infer/level_genotyping/test_model.cpp; no reference code): coverage bookkeeping, diploid
coverage dispatch, credible positions, permutations, genotype rescaling, likelihood counts, the max-likelihood choice and
the extra alleles handed to the parent site. Through gmx_infer_debug (include/gmx.h)."""
import math

import pytest

from gramtools_amd import genotyping_model_debug as dbg, GmxError
from gramtools_amd.quasimap import DBG_INTERNALS, DBG_DIPLOID, DBG_NONCREDIBLE, DBG_PERMUTATIONS, DBG_RESCALE, DBG_CALL

FOUR = [("", [], h) for h in range(4)]   # four haplogroups


def test_haploid_and_singleton_coverages_singletons_only():  # test_model.cpp:14-23
    r = dbg(DBG_INTERNALS, FOUR, {(0,): 5, (1,): 10, (3,): 1})
    assert r["HAPLOID"] == [5, 10, 0, 1] and r["SINGLETON"] == [5, 10, 0, 1]


def test_haploid_and_singleton_coverages_multi_allelic_classes():  # :25-38
    r = dbg(DBG_INTERNALS, FOUR, {(0,): 5, (0, 1): 4, (1,): 10, (2, 3): 1})
    assert r["HAPLOID"] == [9, 14, 1, 1] and r["SINGLETON"] == [5, 10, 0, 0]


def test_diploid_coverages_shared_units_dispatched_by_unique_ratio():  # :40-57 (EXPECT_FLOAT_EQ)
    gp = {(0,): 7, (0, 1): 4, (1,): 20, (0, 3): 3, (2, 3): 1}
    c = dbg(DBG_DIPLOID, FOUR, gp, ids=[0, 1, 4, 0, 0, 0, 0])["C"]
    assert math.isclose(c[0], 10 + 4 / 3., rel_tol=1e-6) and math.isclose(c[1], 20 + 8 / 3., rel_tol=1e-6)


def test_diploid_coverages_only_multi_allelic_classes():  # :59-74
    c = dbg(DBG_DIPLOID, FOUR, {(0, 1): 3, (2, 3): 1}, ids=[0, 1, 4, 0, 0, 0, 0])["C"]
    assert c == [1.5, 1.5]


def test_direct_deletion_allele_gets_its_haplogroups_coverage():  # :76-96
    als = [("C", [8], 0), ("G", [8], 0), ("", [], 1)]
    r = dbg(DBG_INTERNALS, als, {(0,): 8, (1,): 8, (0, 1): 1}, ids=[2])
    assert r["EMPTY_PB"] == [[8], [8], [9]]


def test_diploid_coverages_one_dominating_class():  # :98-135
    gp = {(0,): 8, (0, 1): 4}
    two = [("", [], 0), ("", [], 1)]
    assert dbg(DBG_DIPLOID, two, gp, ids=[0, 1, 2, 0, 0])["C"] == [12.0, 0.0]   # no unique coverage on haplogroup 1
    assert dbg(DBG_DIPLOID, two, gp, ids=[0, 0, 2, 1])["C"] == [6.0, 6.0]        # the same haplogroup twice (nested site within)


def test_fraction_of_noncredible_positions():  # :137-146
    al = [("ATCGCCG", [0, 0, 2, 3, 3, 5, 4, 4], 0)]
    assert dbg(DBG_NONCREDIBLE, al, {}, ids=[0, 3])["F"] == 0.375


def test_total_coverage():  # :148-156
    assert dbg(DBG_INTERNALS, FOUR, {})["TOTAL_COV"] == 0
    assert dbg(DBG_INTERNALS, FOUR, {(0,): 5, (0, 1): 4, (1,): 10, (2, 3): 1})["TOTAL_COV"] == 20


def test_haplogroup_multiplicities():  # :158-179
    assert dbg(DBG_INTERNALS, [("", [], 0), ("", [], 0)], {})["MULT"] == [1]
    assert dbg(DBG_INTERNALS, [("", [], 0), ("", [], 1), ("", [], 1)], {})["MULT"] == [0, 1]


def test_permutations():  # :181-199
    assert dbg(DBG_PERMUTATIONS, ids=[2, 1, 4, 5])["P"] == [[1, 4], [1, 5], [4, 5]]
    assert sorted(dbg(DBG_PERMUTATIONS, ids=[2, 4, 3, 2])["P"]) == [[2, 3], [2, 4], [3, 4]]
    assert dbg(DBG_PERMUTATIONS, ids=[2, 1])["P"] == []


def test_rescale_genotypes():  # :201-216
    assert dbg(DBG_RESCALE, ids=[1, 3])["G"] == [1, 2]
    assert dbg(DBG_RESCALE, ids=[0, 4, 4])["G"] == [0, 1, 1]
    assert dbg(DBG_RESCALE, ids=[4, 2])["G"] == [1, 2]


def test_number_of_likelihoods_with_an_ignored_ref():  # :383-400
    als = [("A", [10], 0, False), ("C", [9], 1), ("G", [10], 2)]
    gp = {(0,): 20, (1,): 9, (2,): 10}
    assert dbg(DBG_INTERNALS, als, gp, 1, 10, 0, 0.01)["N_LIKELIHOODS"] == 2
    assert dbg(DBG_INTERNALS, als, gp, 2, 10, 0, 0.01)["N_LIKELIHOODS"] == 3      # two homozygous and one heterozygous


def test_number_of_genotypes_four_alleles():  # :478-505
    als = [("AATAA", [8] * 5, 0), ("AAGAA", [7] * 5, 0), ("GGTGG", [15, 15, 15, 16, 16], 1), ("GGCGG", [14, 14, 14, 15, 15], 1)]
    gp = {(0,): 15, (1,): 30}
    assert dbg(DBG_INTERNALS, als, gp, 1, 30, 0, 0.01)["N_LIKELIHOODS"] == 4
    assert dbg(DBG_INTERNALS, als, gp, 2, 30, 0, 0.01)["N_LIKELIHOODS"] == 10     # 4 homozygous + (4 choose 2)


AG = [("A", [0], 0), ("G", [0], 1)]
DIFFERENT = [(-4, (0,)), (-2, (1,))]   # allele 1 has the highest log likelihood


def test_extra_alleles_none_with_large_coverage():  # :349-353
    r = dbg(DBG_CALL, AG, [1, 39, 1], 1, 40, 0, 0.01, ids=[0, 0], likelihoods=DIFFERENT)
    assert r["HAS_EXTRA"] is False and r["GT"] == [[1]]


def test_extra_alleles_all_best_alleles_when_confidence_is_zero():  # :355-364
    r = dbg(DBG_CALL, AG, [1, 39], 1, 40, 0, 0.01, ids=[0, 0], likelihoods=[(-2, (0,)), (-2, (1,))])
    assert sorted(r["EXTRA"]) == ["A", "G"] and all(r["EXTRA_CALLABLE"]) and r["GT"] == [[None]]


def test_extra_alleles_in_low_coverage_situations():  # :366-381
    r = dbg(DBG_CALL, AG, [1, 5], 1, 40, 0, 0.01, ids=[0, 0], likelihoods=DIFFERENT)       # low total coverage against mean 40
    assert r["EXTRA"] == ["A"] and r["EXTRA_CALLABLE"] == [False]
    r = dbg(DBG_CALL, AG, [20, 21], 1, 40, 0, 0.01, ids=[0, 0], likelihoods=DIFFERENT)     # low relative difference
    assert r["EXTRA"] == ["A"] and r["EXTRA_CALLABLE"] == [False]


ABCD = [("A", [], 0), ("B", [], 0), ("C", [], 0), ("D", [], 0)]
LIKS = [(-1, (0,)), (-2, (1,)), (-3, (2,)), (-4, (3,))]


def _abcd(callable_flags):
    return [(s, pb, h, c) for (s, pb, h), c in zip(ABCD, callable_flags)]


def test_max_likelihood_choice():  # :507-560
    with pytest.raises(GmxError):                                                         # one likelihood only
        dbg(DBG_CALL, _abcd([1, 1, 1, 1]), [20, 15, 12, 8], 1, 20, 5, 0.01, ids=[0], likelihoods=LIKS[:1])
    r = dbg(DBG_CALL, _abcd([1, 1, 1, 1]), [20, 15, 12, 8], 1, 20, 5, 0.01, ids=[0], likelihoods=LIKS)
    assert r["GT"] == [[0]] and r["ALS"][0] == "A"                                       # the highest likelihood
    r = dbg(DBG_CALL, _abcd([1, 0, 1, 1]), [20, 15, 12, 8], 1, 20, 5, 0.01, ids=[0], likelihoods=LIKS)
    assert r["GT"] == [[0]]                                                              # an inconsistent SECOND best: no skipping
    with pytest.raises(GmxError):                                                         # fewer than two consistent alleles
        dbg(DBG_CALL, _abcd([0, 0, 0, 1]), [20, 15, 12, 8], 1, 20, 5, 0.01, ids=[0], likelihoods=LIKS)


def test_nesting_inconsistent_best_allele_is_not_called():  # :562-581
    r = dbg(DBG_CALL, _abcd([0, 1, 1, 1]), [20, 15, 12, 8], 1, 20, 5, 0.01, ids=[0], likelihoods=LIKS)
    assert r["ALS"] == ["A", "B"] and r["GT"] == [[1]]
