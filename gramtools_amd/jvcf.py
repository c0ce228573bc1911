"""Merging genotyped jVCF JSONs of several samples over one PRG — what the reference's `combine_jvcfs` tool does with
`Json_Prg::combine_with` / `Json_Site::combine_with` (libgramtools/src/genotype/infer/output_specs/json_prg_spec.cpp:62-98,
json_site_spec.cpp:8-140; submods/combine_jvcfs.cpp:61). Host-side bookkeeping on the files `gram genotype` writes; it is
not part of the quasimap path and touches no GPU code. Same names, argument meaning and error behaviour as the reference,
so tests/test_jvcf_combine.py reads like libgramtools/tests/genotype/infer/test_json_spec.cpp."""
import copy
import sys

TRIVIALLY_MERGED_ENTRIES = ("GT", "HAPG", "COV", "DP", "FT")   # output_specs/fields.hpp:107
SINGLETON_ENTRIES = ("POS", "SEG")                             # fields.hpp:108
LEVEL_GENOTYPING_ENTRIES = ("GT_CONF", "GT_CONF_PERCENTILE")   # LevelGenotypedSite::site_model_specific_entries, site.cpp:23-33
SITE_FIELDS = ("POS", "SEG", "ALS", "HAPG", "GT", "DP", "COV", "FT")  # spec::site_fields(), fields.hpp:124-142


class JSONCombineException(RuntimeError):
    pass


class JSONConsistencyException(RuntimeError):
    pass


def _is_null_gt(gt):
    return gt[0] is None


class JsonSite:
    def __init__(self, json_site=None):
        if json_site is None:  # Json_Site(): every site field an empty array, SEG an empty string (json_site_spec.hpp:20-26)
            json_site = {f: [] for f in SITE_FIELDS}
            json_site["SEG"] = ""
        self.json_site = json_site

    def get_site(self):
        return self.json_site

    def set_site(self, json_site):
        self.json_site = copy.deepcopy(json_site)

    # allele -> [index among the combined alleles, haplogroup]; insertion-ordered like the reference's use of the map
    @staticmethod
    def build_allele_combi_map(json_site, m):
        insertion_index = len(m)
        for gts, hapgs in zip(json_site["GT"], json_site["HAPG"] + [None] * len(json_site["GT"])):
            if _is_null_gt(gts):
                continue
            if hapgs is None or len(gts) != len(hapgs):
                raise JSONConsistencyException("Different number of GT and HAPG entries")
            for gt, hapg in zip(gts, hapgs):
                allele = json_site["ALS"][gt]
                if allele not in m:
                    m[allele] = [insertion_index, hapg]
                    insertion_index += 1
                elif m[allele][1] != hapg:
                    sys.stderr.write(f"Warning: Allele {allele} has two HAPG values: {hapg} vs {m[allele][1]}")

    @staticmethod
    def get_all_alleles(m):
        out = [None] * len(m)
        for allele, (index, _) in m.items():
            out[index] = allele
        return out

    def rescale_entries(self, m):
        site = self.json_site
        alleles = site["ALS"]
        for s, gts in enumerate(site["GT"]):
            if _is_null_gt(gts):
                continue
            covs = site["COV"][s]
            if len(alleles) != len(covs):
                raise JSONConsistencyException("Different number of ALS and COV entries")
            new_covs = [0] * len(m)
            for j, cov in enumerate(covs):
                if alleles[j] in m:  # (an allele called in no sample is dropped)
                    new_covs[m[alleles[j]][0]] = cov
            site["GT"][s] = [m[alleles[gt]][0] for gt in gts]
            site["COV"][s] = new_covs

    def combine_with(self, other, gtyping_model=""):
        mine, theirs = self.json_site, other.json_site
        for entry in SINGLETON_ENTRIES:
            if mine[entry] != theirs[entry]:
                raise JSONCombineException(f"Sites do not have same {entry}: ")
        ref = mine["ALS"][0]
        if ref != theirs["ALS"][0]:
            raise JSONCombineException(f"Sites do not have same 'reference' allele: {ref} vs {theirs['ALS'][0]}")
        m = {ref: [0, 0]}  # the REF always first
        self.build_allele_combi_map(mine, m)
        self.build_allele_combi_map(theirs, m)
        self.rescale_entries(m)
        mine["ALS"] = self.get_all_alleles(m)
        other.rescale_entries(m)
        for entry in TRIVIALLY_MERGED_ENTRIES:
            mine[entry].extend(theirs[entry])
        if gtyping_model == "LevelGenotyping":
            for entry in LEVEL_GENOTYPING_ENTRIES:
                mine[entry].extend(theirs[entry])


def empty_prg():
    """spec::json_prg (fields.hpp:154-158) without the descriptive texts, which only ever compare equal to themselves."""
    return {"Model": "UNKNOWN", "Site_Fields": {f: {"Desc": f} for f in SITE_FIELDS}, "Filters": {}, "Samples": [], "Sites": [],
            "Lvl1_Sites": [], "Child_Map": {}}


class JsonPrg:
    def __init__(self, json_prg=None):
        self.json_prg = empty_prg() if json_prg is None else json_prg
        self.sites = [JsonSite(s) for s in self.json_prg["Sites"]]

    def get_prg(self):
        return self.json_prg

    def set_prg(self, json_prg):
        self.json_prg = copy.deepcopy(json_prg)
        self.sites = [JsonSite(s) for s in self.json_prg["Sites"]]

    def set_sample_info(self, name, desc):
        if len(self.json_prg["Samples"]) > 1:
            raise JSONConsistencyException("This JSON already contains > 1 samples")
        self.json_prg["Samples"] = [{"Name": name, "Desc": desc}]

    def add_site(self, site):
        self.sites.append(site)
        self.json_prg["Sites"].append(site.get_site())

    def add_samples(self, other, force=False):
        theirs = other.json_prg
        if len(theirs["Sites"][0]["GT"]) != len(theirs["Samples"]):
            raise JSONConsistencyException("Merged in JSON does not have number of GT arrays consistent with its number of Samples")
        seen = {e["Name"]: 1 for e in self.json_prg["Samples"]}
        for entry in theirs["Samples"]:
            name = used = entry["Name"]
            if name in seen:
                if not force:
                    raise JSONConsistencyException(f"Duplicate sample name found: {name}")
                used = f"{name}_{seen[name]}"
                seen[name] += 1
            else:
                seen[name] = 1
            entry["Name"] = used
            self.json_prg["Samples"].append(entry)

    def combine_with(self, other, force=False):
        mine, theirs = self.json_prg, other.json_prg
        if mine["Model"] != theirs["Model"]:
            raise JSONCombineException("JSONs have different models")
        if mine["Lvl1_Sites"] != theirs["Lvl1_Sites"] or mine["Child_Map"] != theirs["Child_Map"]:
            raise JSONCombineException("Incompatible PRGs (Check Child_Map and Lvl1_Sites)")
        if mine["Site_Fields"] != theirs["Site_Fields"]:
            raise JSONCombineException("Incompatible Site Fields")
        if len(self.sites) != len(other.sites):
            raise JSONCombineException("JSONs do not have the same number of sites")
        self.add_samples(other, force)
        for j, site in enumerate(self.sites):
            site.combine_with(other.sites[j], mine["Model"])
            mine["Sites"][j] = site.get_site()


def combine_jvcf_files(paths, out_path, force=False):
    """submods/combine_jvcfs.cpp: the first file's PRG, every further one merged in."""
    import json
    combined = JsonPrg(json.load(open(paths[0])))
    for p in paths[1:]:
        combined.combine_with(JsonPrg(json.load(open(p))), force)
    with open(out_path, "w") as fh:
        json.dump(combined.get_prg(), fh)
    return combined
