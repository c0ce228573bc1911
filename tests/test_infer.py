"""The infer stage: gmx_infer_* against the known answers of the reference's own tests
(libgramtools/tests/genotype/infer/**, inputs and expectations transcribed; no reference code), on the CPU: coverage comes
from the test-only host emulation of the device logic (the same raw arrays the GPU engine returns)."""
import gzip
import json
import math

import numpy as np
import pytest

from common import hostemu_map, flatten_reads
from golden_runner import prg_ints, seq
from gramtools_amd import Index, Coverage, QuasimapReadsStats, Genotyped, genotyping_model


# ---- level_genotyping/test_model.cpp, test_probabilities.cpp ------------------------------------------------------------
def test_null_genotypes():  # test_model.cpp:236-283
    als = [("A", [0], 0), ("G", [0], 1)]
    r = genotyping_model(als + [("A", [1], 1)], {}, 1, 15, 0, 0.01)      # duplicated allele: null + AMBIG
    assert r["GT"] == [[None]] and r["FT"] == [["AMBIG"]]
    assert genotyping_model(als, {}, 1, 0, 0, 0.01)["GT"] == [[None]]     # zero mean coverage
    assert genotyping_model(als, {}, 1, 15, 0, 0.01)["GT"] == [[None]]    # no coverage on any allele
    r = genotyping_model(als, {(0,): 5, (1,): 5}, 1, 15, 0, 0.01)        # same coverage everywhere: gt_conf 0
    assert r["GT"] == [[None]] and r["GT_CONF"] == [0.0] and r["ALS"] == ["A"]
    assert sorted(r["EXTRA"]) == ["A", "G"] and all(r["EXTRA_CALLABLE"])  # all best alleles go up to the parent site


def test_calls():  # test_model.cpp:285-334
    als = [("ATC", [0, 0, 1], 0), ("GGGCC", [10, 12, 12, 14, 14], 1)]
    gp = {(0,): 1, (1,): 13}
    assert genotyping_model(als, gp, 2, 15, 0, 0.01)["GT"] == [[1, 1]]
    r = genotyping_model(als, gp, 1, 15, 0, 0.01)
    assert r["GT"] == [[1]] and r["ALS"] == ["ATC", "GGGCC"] and r["HAPG"] == [[1]] and r["DP"] == [14]
    assert r["COV"] == [[1.0, 13.0]] and r["GT_CONF"][0] > 0
    assert genotyping_model(als, gp, 1, 15, 16, 0.01)["GT"] == [[1]]      # negative binomial when variance > mean


def test_ignored_ref_and_rescaled_indices():  # test_model.cpp:383-432
    als = [("A", [10], 0, False), ("C", [9], 1), ("G", [10], 2)]
    gp = {(0,): 20, (1,): 9, (2,): 10}
    r = genotyping_model(als, gp, 1, 10, 0, 0.01)
    assert r["ALS"] == ["A", "G"] and r["GT"] == [[1]]
    r = genotyping_model(als, gp, 2, 10, 0, 0.01)
    assert r["ALS"] == ["A", "C", "G"] and r["GT"] == [[1, 2]]


def test_homozygous_and_gap_penalty():  # test_model.cpp:434-476
    r = genotyping_model([("AA", [0, 1], 0), ("TT", [20, 19], 1)], {(0,): 2, (0, 1): 1, (1,): 20}, 2, 20, 0, 0.01)
    assert r["GT"] == [[1, 1]]
    r = genotyping_model([("AAAACAG", [0, 20, 20, 20, 20, 20, 0], 0), ("TAAACAT", [20] * 7, 0)], {(0,): 20}, 1, 20, 200, 0.01)
    assert r["GT"] == [[1]]


def test_likelihood_statistics():  # test_probabilities.cpp:56-121
    als, gp = [("A", [3], 0), ("C", [1], 1)], {(0,): 3, (1,): 1}
    assert genotyping_model(als, gp, 1, 2, 0, 0.01)["LOG_ZERO"] == -2.0           # ln Poisson(lambda = 2)(0)
    for mean, err, want in ((10, 0.0001, 1), (10, 0.001, 2), (100, 0.001, 10)):
        assert genotyping_model(als, gp, 1, mean, 0, err)["CREDIBLE_COV_T"] == want
    r = genotyping_model(als, gp, 1, 10, 20, 0.01)                               # negative binomial: k = 10, p = 0.5
    assert math.isclose(r["LOG_ZERO"], 10 * math.log(0.5)) and math.isclose(r["LOG_NO_ZERO"], math.log(1 - 0.5 ** 10))


def test_more_than_one_likelihood_is_needed():  # test_model.cpp:223-231 (EXPECT_DEATH there; an error code here)
    from gramtools_amd import GmxError
    with pytest.raises(GmxError):
        genotyping_model([("ACGT", [1, 1, 1, 1], 0)], {(0,): 3}, 1, 10, 0, 0.01)


# ---- level_genotyping/test_runner.cpp: PRG + reads -> genotyped sites --------------------------------------------------
def _genotype(prg_spec, reads, ploidy="haploid", k=2, err=0.001):
    prg = prg_ints(prg_spec)
    rd = [seq(r) for r in reads]
    seeds = np.arange(len(rd), dtype=np.uint32)
    raw, _, rc = hostemu_map(prg, k, rd, seeds, return_raw=True)
    assert rc == 0
    ix = Index(prg, k, threads=1)
    cov = Coverage(ix, raw["allele_sum"], raw["per_base"], raw["grouped"], raw["grouped_log"], QuasimapReadsStats(*(int(x) for x in raw["stats"])))
    return ix, Genotyped(cov, err, ploidy)


def test_two_site_non_nested_prg():  # test_runner.cpp:14-40
    ix, g = _genotype({"numbered": "AATAA5C6G6AA7C8G8AA"}, ["AATAACAACAA"] * 5 + ["AATAAGAACAA"])
    assert g.called_alleles(0) == ["C"] and g.site(0)["HAPG"] == [[0]] and g.site(0)["COV"] == [[5.0]]  # five reads through 5:1
    assert g.called_alleles(1) == ["C"] and g.site(1)["COV"] == [[6.0]]                               # all six through 7:1


def test_two_site_nested_prg():  # test_runner.cpp:42-68
    ix, g = _genotype({"bracketed": "AATAA[CCC[A,G],T]AA"}, ["AATAACCCGAA"] * 5 + ["AATAATAA"])
    assert g.called_alleles(1) == ["G"] and g.site(1)["HAPG"] == [[1]]
    assert g.called_alleles(0) == ["CCCG"] and g.site(0)["HAPG"] == [[0]]


def test_direct_deletion_is_called():  # test_runner.cpp:70-91
    ix, g = _genotype({"bracketed": "GGGGG[CCC,]GG"}, ["GGGGGG"] * 5)
    assert g.called_alleles(0) == [""] and g.site(0)["HAPG"] == [[1]]


def test_snps_nested_in_two_haplotypes():  # test_runner.cpp:93-160
    spec = {"bracketed": "ATCGGC[TC[A,G]TC,GG[T,G]GG]AT"}
    ix, g = _genotype(spec, [])
    assert all(g.site(s)["GT"] == [[None]] for s in range(3))
    ix, g = _genotype(spec, ["ATCGGCTCGTCAT"] * 7 + ["ATCGGCGGG"])
    assert g.called_alleles(0) == ["TCGTC"] and g.site(0)["HAPG"] == [[0]]
    assert g.called_alleles(1) == ["G"] and g.site(1)["HAPG"] == [[1]]
    s9 = g.site(2)                               # lives on the haplogroup that was not called: invalidated
    assert s9["GT"] == [[None]] and s9["GT_CONF"] == [0.0]


# ---- outputs ------------------------------------------------------------------------------------------------------------
def _fasta(path):
    recs, name = [], None
    for line in open(path).read().splitlines():
        if line.startswith(">"):
            name = line[1:]
            recs.append([name, ""])
        else:
            recs[-1][1] += line
    return recs


@pytest.mark.parametrize("coords,expected", [
    (None, ["ATCGCTTTATC"]),
    ("chr1\t2\nchr2\t9\n", ["AT", "CGCTTTATC"]),
    ("chr1\t6\nchr2\t5\n", ["ATCGCT", "TTATC"]),
    ("chr1\t10\nchr2\t1\n", ["ATCGCTTTAT", "C"]),
    ("chr1\t7\nchr2\t4\n", ["ATCGCTT", "TATC"]),
], ids=["one-segment", "to-edge", "from-edge", "adjacent-sites", "inside-sequence"])
def test_personalised_reference_of_null_genotypes(tmp_path, coords, expected):  # test_personalised_reference.cpp:132-217
    ix, g = _genotype({"bracketed": "AT[CG[C,G]T,C]TT[AT,TT][C,G]"}, [])
    cp = None
    if coords:
        cp = str(tmp_path / "prg_coords.tsv")
        open(cp, "w").write(coords)
    g.write(str(tmp_path), "sample", cp)
    recs = _fasta(tmp_path / "personalised_reference.fasta")
    assert sorted(r[1] for r in recs) == sorted(expected)
    assert all(r[0].endswith("sample personalised reference made by gramtools genotype") for r in recs)
    if coords:
        assert sorted(r[0].split()[0] for r in recs) == ["chr1", "chr2"]


def test_outputs_of_a_nested_diploid_run(tmp_path):
    """jVCF, VCF and FASTA of one run, read back: consistent with each other and with the site records."""
    spec = {"bracketed": "ATCGGC[TC[A,G]TC,GG[T,G]GG]AT[C,G,T]AA"}
    reads = ["ATCGGCTCGTCATCAA"] * 6 + ["ATCGGCGGTGGATGAA"] * 6 + ["GGCTCGTCATC"] * 2
    ix, g = _genotype(spec, reads, ploidy="diploid")
    open(tmp_path / "prg_coords.tsv", "w").write("chrA\t11\nchrB\t5\n")  # chrA ends with the first site (first-allele coordinates)
    g.write(str(tmp_path), "s1", str(tmp_path / "prg_coords.tsv"))
    j = json.loads(open(tmp_path / "genotyped.json").read())
    assert sorted(j) == ["Child_Map", "Filters", "Lvl1_Sites", "Model", "Samples", "Site_Fields", "Sites"]
    assert j["Model"] == "LevelGenotyping" and j["Samples"] == [{"Desc": "made by gramtools genotype", "Name": "s1"}]
    assert j["Lvl1_Sites"] == [0, 3] and j["Child_Map"] == {"0": {"0": [1], "1": [2]}}
    assert len(j["Sites"]) == 4
    for i, site in enumerate(j["Sites"]):
        mine = g.site(i)
        for key in ("ALS", "GT", "HAPG", "COV", "DP", "FT", "GT_CONF", "GT_CONF_PERCENTILE"):
            assert site[key] == mine[key]
    assert j["Sites"][0]["ALS"] == ["TCATC", "TCGTC", "GGTGG"] and j["Sites"][0]["GT"] == [[1, 2]]  # REF first, het call
    assert j["Sites"][0]["HAPG"] == [[0, 1]]                                                     # across the two haplogroups
    assert j["Sites"][0]["SEG"] == "chrA" and j["Sites"][0]["POS"] == 7
    assert j["Sites"][3]["SEG"] == "chrB" and j["Sites"][3]["POS"] == 3                  # 6 + 5 (first allele) + 2 = 13 -> 13 - 11 + 1
    # VCF: BGZF container (gzip members with the BC extra field, then the EOF block), level-1 sites only
    blob = open(tmp_path / "genotyped.vcf.gz", "rb").read()
    assert blob[:4] == b"\x1f\x8b\x08\x04" and blob[12:14] == b"BC" and blob.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    lines = gzip.decompress(blob).decode().splitlines()
    assert lines[0] == "##fileformat=VCFv4.2" and "##contig=<ID=chrA,length=11,Source=\"gramtools\">" in lines
    recs = [l.split("\t") for l in lines if not l.startswith("#")]
    assert [r[0] for r in recs] == ["chrA", "chrB"] and [r[1] for r in recs] == ["7", "3"]
    for r, si in zip(recs, (0, 3)):
        site = j["Sites"][si]
        assert [r[3]] + (r[4].split(",") if r[4] != "." else []) == site["ALS"]
        fmt = dict(zip(r[8].split(":"), r[9].split(":")))
        assert fmt["GT"] == "/".join(str(x) for x in site["GT"][0]) and fmt["DP"] == str(site["DP"][0]) and fmt["FT"] == "PASS"
        assert math.isclose(float(fmt["GT_CONF"]), site["GT_CONF"][0], rel_tol=1e-5)
    # FASTA: two haplotypes of chrA, (one or two) of chrB
    names = [r[0].split()[0] for r in _fasta(tmp_path / "personalised_reference.fasta")]
    assert set(names) <= {"chrA_1", "chrA_2", "chrB_1", "chrB_2"} and any(n.startswith("chrA") for n in names)


def test_gt_conf_percentiles_are_reproducible_and_monotone():
    """GT_CONF_PERCENTILE: empirical confidences + simulations from std::default_random_engine(42) as lib/GCP/GCP.h; the
    same input gives the same numbers, and a larger confidence never gets a smaller percentile."""
    spec = {"numbered": "AATAA5C6G6AA7C8G8AAGT9A10C10TTG"}
    reads = ["AATAACAACAAGTATTG"] * 12 + ["AATAAGAACAAGTCTTG"] * 3
    a = [_genotype(spec, reads)[1].site(i) for i in range(3)]
    b = [_genotype(spec, reads)[1].site(i) for i in range(3)]
    assert a == b
    pairs = sorted((s["GT_CONF"][0], s["GT_CONF_PERCENTILE"][0]) for s in a)
    assert all(0 <= p <= 100 for _, p in pairs) and all(x[1] <= y[1] for x, y in zip(pairs, pairs[1:]))


def test_threads_change_nothing_in_the_calls_or_the_files(tmp_path, monkeypatch):
    """(round 5) On a non-nested PRG the sites are genotyped side by side and the jVCF / VCF records are formatted and the BGZF
    members compressed on several threads: calls and files are byte-identical to the one-thread run (the reference's loops:
    level_genotyping/runner.cpp:29-103, make_json.cpp, make_vcf.cpp)."""
    import numpy as np
    from gramtools_amd import Index, Coverage, QuasimapReadsStats
    from gramtools_amd.quasimap import Genotyped
    from gramtools_amd.synth import random_ref, snp_prg
    ref = random_ref(300_000, 71)
    prg, pos, alts, n_alts = snp_prg(ref, 9000, 72, multi_allelic_frac=0.2)
    ix = Index(prg, 5)
    info = ix.info
    rng = np.random.default_rng(73)
    cov = Coverage(ix, rng.integers(0, 40, info.n_allele_slots, dtype=np.uint32), rng.integers(0, 40, info.n_per_base_slots, dtype=np.uint32),
                   rng.integers(0, 25, max(info.n_grouped_slots, 1), dtype=np.uint32), np.zeros(0, dtype=np.uint32),
                   QuasimapReadsStats(0, 0, 0, 0, 0))
    open(tmp_path / "coords.tsv", "w").write("chr1\t100000\nchr2\t120000\nchr3\t80000\n")
    outs = []
    for threads in ("1", "7", "16"):
        monkeypatch.setenv("GMX_INFER_THREADS", threads)
        d = tmp_path / f"t{threads}"
        d.mkdir()
        g = Genotyped(cov, 0.001, depth=dict(mean=18.0, variance=30.0))
        g.write(str(d), "s", str(tmp_path / "coords.tsv"))
        outs.append({n: open(d / n, "rb").read() for n in ("genotyped.json", "genotyped.vcf.gz", "personalised_reference.fasta")} |
                    {"sites": [g.site(i) for i in (0, 1, 4500, 8999)]})
        g.close()
    assert outs[0] == outs[1] == outs[2]
    assert len(gzip.decompress(outs[0]["genotyped.vcf.gz"]).splitlines()) > 9000 and len(outs[0]["genotyped.vcf.gz"]) > 3 * 0xff00 // 10
