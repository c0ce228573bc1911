"""The allele extracter of the infer stage (gmx_infer.cpp: Genotyper::extract / combine / ref_allele) against the reference's own
unit tests (inputs and expectations transcribed from libgramtools/tests/genotype/infer/test_allele_extracter.cpp; no
reference code). Child sites are mocked as the reference mocks them: genotype, alleles, extra alleles."""
import ctypes as C
import json

from gramtools_amd import Index, _lib
from gramtools_amd._lib import check
from gramtools_amd.synth import bracket_to_ints


def A(seq, pb, hapg=0, callable_=True):
    return f"{seq}/{','.join(str(c) for c in pb)}/{hapg}/{1 if callable_ else 0}"


def mock(site_index, gt, alleles, extra=None):
    return f"{site_index}|{','.join(str(g) for g in gt)}|{';'.join(alleles)}|{';'.join(extra) if extra else '-'}"


def run(prg_text, op, site_index, existing=(), mocks=()):
    lib = _lib.load()
    ix = Index(bracket_to_ints(prg_text), 1, threads=1)
    ex = ";".join(existing).encode() if existing else None
    mk = "\n".join(mocks).encode() if mocks else None
    n = check(lib.gmx_infer_extract_debug(ix.h, op, site_index, None, ex, mk, None, 0))
    buf = C.create_string_buffer(n + 1)
    check(lib.gmx_infer_extract_debug(ix.h, op, site_index, None, ex, mk, buf, n + 1))
    return [tuple(a[:1] + [a[1], a[2], a[3]]) for a in json.loads(buf.value.decode())]


EXTRACT, REF, COMBINE = 0, 1, 2
NESTED = "AT[GCC[C,A,G]T,TTA]T"        # site 5 (index 0) encloses site 7 (index 1)
EXISTING = [A("ATTG", [0, 1, 2, 3]), A("ATCG", [0, 0, 1, 1])]


def test_ref_allele_of_a_site_with_nested_sites():  # :12-20
    (seq, pb, hapg, _), = run("AT[[C,A,G]T[G[,C]C,T],TTA]T", REF, 0)
    assert (seq, hapg) == ("CTGC", 0)


def test_combine_one_called_allele_keeps_the_left_haplogroup():  # :37-48
    got = run(NESTED, COMBINE, 1, EXISTING[:1], [mock(1, [0], [A("CCC", [1, 1, 1], 2)])])
    assert got == [("ATTGCCC", [0, 1, 2, 3, 1, 1, 1], 0, True)]


def test_combine_includes_extra_alleles_and_their_nesting_inconsistency():  # :50-73
    got = run(NESTED, COMBINE, 1, EXISTING[:1],
              [mock(1, [1], [A("CCC", [1, 1, 1]), A("GGG", [2, 2, 2])], [A("AAA", [2, 1, 0], 2, False)])])
    assert got == [("ATTGGGG", [0, 1, 2, 3, 2, 2, 2], 0, True), ("ATTGAAA", [0, 1, 2, 3, 2, 1, 0], 0, False)]


def test_combine_null_genotype_takes_the_first_allele():  # :75-87
    got = run(NESTED, COMBINE, 1, EXISTING[:1], [mock(1, [-1], [A("TTT", [1, 1, 1]), A("CCC", [0, 1, 1])])])
    assert got == [("ATTGTTT", [0, 1, 2, 3, 1, 1, 1], 0, True)]


def test_combine_heterozygous_genotype_gives_all_four_combinations():  # :89-112
    got = run(NESTED, COMBINE, 1, EXISTING, [mock(1, [0, 1], [A("CCC", [1, 1, 1], 0), A("TTT", [5, 5, 5], 1)])])
    assert got == [("ATTGCCC", [0, 1, 2, 3, 1, 1, 1], 0, True), ("ATTGTTT", [0, 1, 2, 3, 5, 5, 5], 0, True),
                   ("ATCGCCC", [0, 0, 1, 1, 1, 1, 1], 0, True), ("ATCGTTT", [0, 0, 1, 1, 5, 5, 5], 0, True)]


def test_nested_bubble_alleles():  # :152-161
    got = run(NESTED, EXTRACT, 1)
    assert got == [("C", [0], 0, True), ("A", [0], 1, True), ("G", [0], 2, True)]


def test_outer_bubble_around_a_haploid_nested_call():  # :163-175
    got = run(NESTED, EXTRACT, 0, mocks=[mock(1, [0], [A("C", [0], 0)])])
    assert [a[:3] for a in got] == [("GCCCT", [0] * 5, 0), ("TTA", [0] * 3, 1)]


def test_outer_bubble_around_a_triploid_nested_call():  # :177-195
    got = run(NESTED, EXTRACT, 0, mocks=[mock(1, [0, 1, 2], [A("C", [0], 0), A("A", [0], 1), A("G", [0], 2)])])
    assert [a[:3] for a in got] == [("GCCCT", [0] * 5, 0), ("GCCAT", [0] * 5, 0), ("GCCGT", [0] * 5, 0), ("TTA", [0] * 3, 1)]
    assert got[0][3] is True


def test_outer_bubble_non_ref_nested_call_still_produces_the_ref_uncallable():  # :197-213
    got = run(NESTED, EXTRACT, 0, mocks=[mock(1, [1], [A("C", [0], 0), A("G", [0], 2)])])
    assert [a[:3] for a in got] == [("GCCCT", [0] * 5, 0), ("GCCGT", [0] * 5, 0), ("TTA", [0] * 3, 1)]
    assert got[0][3] is False


def test_nested_next_best_allele_is_produced():  # :215-231
    got = run(NESTED, EXTRACT, 0, mocks=[mock(1, [1], [A("C", [0], 0), A("G", [0], 2)], [A("A", [0], 1)])])
    assert [a[:3] for a in got] == [("GCCCT", [0] * 5, 0), ("GCCGT", [0] * 5, 0), ("GCCAT", [0] * 5, 0), ("TTA", [0] * 3, 1)]


def test_direct_deletion_allele_is_present():  # :233-245
    got = run("AT[GCC,TTA,]T", EXTRACT, 0)
    assert [a[:3] for a in got] == [("GCC", [0] * 3, 0), ("TTA", [0] * 3, 1), ("", [], 2)]
