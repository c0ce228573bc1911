"""libgramtools/tests/genotype/infer/test_json_spec.cpp (17 cases) against gramtools_amd/jvcf.py: the merge rules of
genotyped sites and PRGs of several samples (Json_Site::combine_with, Json_Prg::combine_with)."""
import copy
import json

import pytest

from gramtools_amd.jvcf import JsonSite, JsonPrg, JSONCombineException, JSONConsistencyException, empty_prg


def mock_site(als, gts, hapgs, covs, dps, pos, seg):
    """MockJsonSite: one sample (flat lists) or several (lists of lists); test_json_spec.cpp:30-57."""
    s = JsonSite()
    j = s.get_site()
    j["SEG"], j["POS"], j["ALS"] = seg, pos, list(als)
    if gts and not isinstance(gts[0], list):
        gts, hapgs, covs, dps = [gts], [hapgs], [covs], [dps]
    for g, h, c, d in zip(gts, hapgs, covs, dps):
        j["GT"].append(list(g)); j["HAPG"].append(list(h)); j["COV"].append(list(c)); j["DP"].append(d)
    return s


class Data:  # JSON_data_store, test_json_spec.cpp:59-103
    def __init__(self):
        self.site1 = [mock_site(["CTCCT", "CTT"], [0, 0], [0, 0], [10, 2], 11, 3, "gene1"),
                      mock_site(["CTCCT", "CTT"], [1, 1], [1, 1], [2, 10], 11, 3, "gene1"),
                      mock_site(["CTCCT", "GTT"], [0, 1], [0, 2], [5, 5], 12, 3, "gene1")]
        self.site2 = [mock_site(["AAAAAAA", "AAA"], [1], [1], [20, 1], 23, 50, "gene2"),
                      mock_site(["AAAAAAA", "A"], [1], [4], [0, 18], 24, 50, "gene2")]
        self.prg1, self.prg2 = JsonPrg(), JsonPrg()
        self.prg1.set_sample_info("Gazorp", "")
        self.prg1.add_site(self.site1[0]); self.prg1.add_site(self.site2[0])
        self.prg2.set_sample_info("Dorp", "")
        self.prg2.add_site(self.site1[1]); self.prg2.add_site(self.site2[1])


@pytest.fixture
def fixed_and_json():
    the_site = copy.deepcopy(Data().site1[0].get_site())
    fixed = JsonSite()
    fixed.set_site(the_site)
    return fixed, the_site


def _test_site(j):
    t = JsonSite()
    t.set_site(j)
    return t


def test_site_combine_same_jsons_no_fail(fixed_and_json):
    fixed, j = fixed_and_json
    fixed.combine_with(_test_site(j))


@pytest.mark.parametrize("change", [lambda j: j["ALS"].__setitem__(0, "NOTSAME"), lambda j: j.__setitem__("SEG", "another_gene"),
                                    lambda j: j.__setitem__("POS", 100)], ids=["REF", "SEG", "POS"])
def test_site_combine_different_singletons_fail(fixed_and_json, change):
    fixed, j = fixed_and_json
    change(j)
    with pytest.raises(JSONCombineException):
        fixed.combine_with(_test_site(j))


def test_site_combine_inconsistent_hapgs_do_not_fail(fixed_and_json):
    fixed, j = fixed_and_json
    j["HAPG"][0][0] = 1
    fixed.combine_with(_test_site(j))


def test_site_combine_different_cov_and_als_cardinality_fails(fixed_and_json):
    fixed, j = fixed_and_json
    j["COV"][0] = [10]
    with pytest.raises(JSONConsistencyException):
        fixed.combine_with(_test_site(j))


def test_combi_map_of_two_samples():
    d, m = Data(), {}
    JsonSite.build_allele_combi_map(d.site1[0].get_site(), m)
    JsonSite.build_allele_combi_map(d.site1[1].get_site(), m)
    assert m == {"CTCCT": [0, 0], "CTT": [1, 1]}


def test_rescale_entries_given_combi_map():
    m = {"CTCCT": [0, 0], "CCC": [1, 2], "CTT": [2, 1]}
    s = Data().site1[1]
    s.rescale_entries(m)
    assert s.get_site() == mock_site(["CTCCT", "CTT"], [2, 2], [1, 1], [2, 0, 10], 11, 3, "gene1").get_site()


def test_append_entries_of_two_genotyped_sites():
    d = Data()
    d.site1[0].combine_with(d.site1[1])
    want = mock_site(["CTCCT", "CTT"], [[0, 0], [1, 1]], [[0, 0], [1, 1]], [[10, 2], [2, 10]], [11, 11], 3, "gene1")
    assert d.site1[0].get_site() == want.get_site()


def test_combine_with_one_null_genotyped_site():
    d = Data()
    null = _test_site(d.site1[0].get_site())
    null.get_site()["GT"][0] = [None]
    d.site1[0].combine_with(null)
    assert d.site1[0].get_site()["GT"] == [[0, 0], [None]]


def test_three_sites_combined_and_associativity():
    want = mock_site(["CTCCT", "CTT", "GTT"], [[0, 0], [1, 1], [0, 2]], [[0, 0], [1, 1], [0, 2]],
                     [[10, 2, 0], [2, 10, 0], [5, 0, 5]], [11, 11, 12], 3, "gene1").get_site()
    d = Data()
    d.site1[0].combine_with(d.site1[1])
    d.site1[0].combine_with(d.site1[2])
    assert d.site1[0].get_site() == want
    d = Data()
    d.site1[1].combine_with(d.site1[2])
    d.site1[0].combine_with(d.site1[1])
    assert d.site1[0].get_site() == want


@pytest.fixture
def prg_and_json():
    the_prg = empty_prg()
    the_prg["Model"] = "M1"
    the_prg["Child_Map"] = {"0": {"1": [2, 3]}}
    the_prg["Lvl1_Sites"].append(0)
    p = JsonPrg()
    p.set_prg(the_prg)
    return p, the_prg


def _prg(j):
    p = JsonPrg()
    p.set_prg(j)
    return p


def test_prg_combine_different_models_fails(prg_and_json):
    p1, j = prg_and_json
    j["Model"] = "A_different_model"
    with pytest.raises(JSONCombineException):
        p1.combine_with(_prg(j))


def test_prg_combine_different_prgs_fails(prg_and_json):
    p1, j = prg_and_json
    keep = copy.deepcopy(j)
    j["Child_Map"] = {}
    with pytest.raises(JSONCombineException):
        p1.combine_with(_prg(j))
    assert keep == p1.get_prg()
    keep["Lvl1_Sites"].append("all")
    with pytest.raises(JSONCombineException):
        p1.combine_with(_prg(keep))


def test_prg_combine_different_site_specs_fails(prg_and_json):
    p1, j = prg_and_json
    j["Site_Fields"]["GT"]["Desc"] = "Greater Than"
    with pytest.raises(JSONCombineException):
        p1.combine_with(_prg(j))


def test_prg_combine_different_number_of_sites_fails(prg_and_json):
    p1, j = prg_and_json
    p2 = _prg(j)
    p2.add_site(JsonSite())
    with pytest.raises(JSONCombineException):
        p1.combine_with(p2)


def test_sample_names_can_be_forced():
    d = Data()
    p1, p2 = JsonPrg(), JsonPrg()
    p1.add_site(d.site1[0]); p2.add_site(d.site1[1])
    p1.set_sample_info("Sample1", "I am sample1")
    p2.set_sample_info("Sample1", "I am another sample but I was named the same. Sorry.")
    with pytest.raises(JSONConsistencyException):
        p1.add_samples(p2)
    want = [copy.deepcopy(p1.get_prg()["Samples"][0]), copy.deepcopy(p2.get_prg()["Samples"][0])]
    want[1]["Name"] = "Sample1_1"
    p1.add_samples(p2, True)
    assert p1.get_prg()["Samples"] == want


def test_two_prgs_combined_site_by_site(tmp_path):
    d = Data()
    s1a, s1b = _test_site(d.site1[0].get_site()), _test_site(d.site1[1].get_site())
    s2a, s2b = _test_site(d.site2[0].get_site()), _test_site(d.site2[1].get_site())
    d.prg1.combine_with(d.prg2)
    s1a.combine_with(s1b)
    s2a.combine_with(s2b)
    assert d.prg1.get_prg()["Sites"][0] == s1a.get_site()
    assert d.prg1.get_prg()["Sites"][1] == s2a.get_site()
    assert [s["Name"] for s in d.prg1.get_prg()["Samples"]] == ["Gazorp", "Dorp"]
    json.dumps(d.prg1.get_prg())


def test_two_samples_of_this_engine_combine(tmp_path):
    """genotyped.json of two samples as `gram genotype` writes them (gmx_infer_write_json) through the merge: the file form
    of submods/combine_jvcfs.cpp. Coverage from the host emulation of the device logic (== the oracle's)."""
    import numpy as np
    from common import hostemu_map
    from gramtools_amd import Index, Coverage, QuasimapReadsStats, Genotyped, master_seeds
    from gramtools_amd.jvcf import combine_jvcf_files
    from gramtools_amd.synth import random_ref, mixed_variant_prg, simulate_haplotype_reads
    ref = random_ref(3000, 11)
    prg, sites = mixed_variant_prg(ref, 60, 12, max_alleles=3)
    ix = Index(prg, 7, threads=1)
    paths = []
    for sample in range(2):
        reads = simulate_haplotype_reads(ref, sites, 600, 60, 100, 20 + sample)
        raw, _, rc = hostemu_map(prg, 7, reads, master_seeds(42, [len(reads)]), return_raw=True)
        assert rc == 0
        cov = Coverage(ix, raw["allele_sum"], raw["per_base"], raw["grouped"], raw["grouped_log"], QuasimapReadsStats(*(int(x) for x in raw["stats"])))
        d = tmp_path / f"s{sample}"
        d.mkdir()
        Genotyped(cov, 0.01, "haploid").write(str(d), f"sample{sample}")
        paths.append(str(d / "genotyped.json"))
    one = json.load(open(paths[0]))
    both = combine_jvcf_files(paths, str(tmp_path / "combined.json")).get_prg()
    assert [s["Name"] for s in both["Samples"]] == ["sample0", "sample1"] and len(both["Sites"]) == len(one["Sites"])
    for merged, first in zip(both["Sites"], one["Sites"]):
        assert len(merged["GT"]) == len(merged["DP"]) == len(merged["GT_CONF"]) == 2 and merged["ALS"][0] == first["ALS"][0]
        assert merged["DP"][0] == first["DP"][0] and merged["GT_CONF"][0] == first["GT_CONF"][0]
        if first["GT"][0][0] is not None:  # the first sample's call names the same allele after the rescaling
            assert merged["ALS"][merged["GT"][0][0]] == first["ALS"][first["GT"][0][0]]
    with pytest.raises(JSONConsistencyException):
        combine_jvcf_files([paths[0], paths[0]], str(tmp_path / "dup.json"))
    combine_jvcf_files([paths[0], paths[0]], str(tmp_path / "dup.json"), force=True)
