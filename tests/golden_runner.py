"""Interpreter for the tests/golden/*.json known-answer vectors.

Each case names a PRG, a k-mer size and a list of ops. `run_case(case, backend)`
executes the ops against a backend object exposing the oracle's method names,
so the same vectors pin the CPU oracle (tests/test_oracle_golden.py) and — for the
ops the product implements — the HIP path (tests/test_gpu_golden.py).
"""
import json
import os

import numpy as np

from oracle import encode_prg, prg_string_to_ints, encode_dna_bases

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASES = {"a": 1, "c": 2, "g": 3, "t": 4, "A": 1, "C": 2, "G": 3, "T": 4}


def load_cases(fname):
    with open(os.path.join(GOLDEN_DIR, fname)) as fh:
        doc = json.load(fh)
    return doc["cases"]


def all_cases():
    out = []
    for f in sorted(os.listdir(GOLDEN_DIR)):
        if f.endswith(".json"):
            for c in load_cases(f):
                if "kind" not in c:  # (dumps_and_stats.json: host-side formats, driven by tests/test_golden_dumps.py)
                    out.append((f, c))
    return out


def host_cases():
    return [c for c in load_cases("dumps_and_stats.json")]


def prg_ints(spec):
    if "numbered" in spec:
        return encode_prg(spec["numbered"])
    if "bracketed" in spec:
        return prg_string_to_ints(spec["bracketed"])
    return list(spec["ints"])


def seq(x):
    if isinstance(x, str):
        return encode_dna_bases(x)
    return np.asarray(x, dtype=np.uint8)


def st(s):
    return (s[0], s[1], [tuple(x) for x in s[2]], [tuple(x) for x in s[3]])


def norm_states(states):
    return [(int(s[0]), int(s[1]), [tuple(map(int, x)) for x in s[2]], [tuple(map(int, x)) for x in s[3]]) for s in states]


def check_front(state, exp):
    lo, hi, tvd, tvg = state
    if "sa" in exp:
        assert [lo, hi] == exp["sa"]
    if "traversed" in exp:
        assert tvd == [tuple(x) for x in exp["traversed"]]
    if "traversing" in exp:
        assert tvg == [tuple(x) for x in exp["traversing"]]
    if "traversed_front" in exp:
        assert tvd[0] == tuple(exp["traversed_front"])


def grouped_key(d):
    return {",".join(str(i) for i in k): v for k, v in d.items()}


def check_states(res, op):
    res = norm_states(res)
    if "expect" in op:
        assert res == norm_states([st(s) for s in op["expect"]]), (res, op["expect"])
    if "expect_size" in op:
        assert len(res) == op["expect_size"], res
    if "expect_front" in op:
        check_front(res[0], op["expect_front"])
    if "expect_back" in op:
        check_front(res[-1], op["expect_back"])
    if "expect_front_width" in op:
        assert res[0][1] - res[0][0] + 1 == op["expect_front_width"]
    if "expect_front_traversing_back_allele" in op:
        assert res[0][3][-1][1] == op["expect_front_traversing_back_allele"]


def run_case(case, make_backend):
    """make_backend(prg_ints, k, all_kmers) -> backend; raises on inconsistent PRGs."""
    ints = prg_ints(case["prg"])
    k = case.get("k", 2)
    if case.get("expect_build_error"):
        try:
            make_backend(ints, k, case.get("all_kmers", True))
        except Exception:
            return
        raise AssertionError("expected PRG construction to fail")
    o = make_backend(ints, k, case.get("all_kmers", True))
    for op in case["ops"]:
        run_op(o, op, ints)


def run_op(o, op, ints):
    kind = op["op"]
    if kind == "quasimap_read":
        o.quasimap_read(seq(op["read"]), op.get("seed", 42))
    elif kind == "map_reads":
        reads = [seq(r) for r in op["reads"]]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
        seeds = o.master_seeds(op["master_seed"], [len(reads)])
        o.map_reads(np.concatenate(reads), offs, seeds)
    elif kind == "expect_stats":
        s = o.stats()
        for key, v in op["value"].items():
            assert s[key] == v, s
    elif kind == "expect_allele_sum":
        assert o.allele_sum() == op["value"], o.allele_sum()
    elif kind == "expect_allele_base":
        assert o.allele_base_non_nested() == op["value"], o.allele_base_non_nested()
    elif kind == "expect_grouped":
        got = [grouped_key(d) for d in o.grouped()]
        assert got == op["value"], got
    elif kind == "expect_node_cov":
        got = [o.node_coverage_at(p) for p in op["positions"]]
        assert got == op["value"], got
    elif kind == "expect_num_sites":
        assert o.num_sites() == op["value"]
    elif kind == "expect_is_nested":
        assert o.is_nested() == op["value"]
    elif kind == "expect_sa":
        assert o.sa().tolist() == op["value"], o.sa().tolist()
    elif kind == "expect_bwt":
        assert o.bwt().tolist() == op["value"], o.bwt().tolist()
    elif kind == "expect_prg":
        assert list(ints) == op["value"]
    elif kind == "rank":
        assert o.rank(op["upper"], op["base"]) == op["expect"]
    elif kind == "marker_sa_interval":
        assert list(o.marker_sa_interval(op["marker"])) == op["expect"]
    elif kind == "base_next_sa_interval":
        assert list(o.base_next_sa_interval(op["next_char"], op["first_sa"], *op["sa"])) == op["expect"]
    elif kind == "reverse_complement":
        assert o.reverse_complement(seq(op["read"])).tolist() == op["expect"]
    elif kind == "all_kmers_in_index":
        assert o.all_kmers_in_index(seq(op["read"])) == op["expect"]
    elif kind == "index_kmers":
        o.index_kmers([seq(km) for km in op["kmers"]])
    elif kind == "index_kmer_diffs":
        o.index_kmer_diffs([seq(d) for d in op["diffs"]])
    elif kind == "expect_kmer_index_size":
        assert o.kmer_index_size() == op["value"]
    elif kind == "expect_kmer":
        res = o.kmer_states(seq(op["kmer"]))
        if op.get("expect_absent"):
            assert res is None
        else:
            assert res is not None
            check_states(res, op)
    elif kind == "process_read_char_from_kmer":
        states = o.kmer_states(seq(op["kmer"]))
        check_states(o.process_read_char(BASES[op["base"]], states), op)
    elif kind == "search_read_backwards":
        check_states(o.search_read_backwards(seq(op["read"])), op)
    elif kind == "search_base_backwards":
        check_states(o.search_base_backwards(BASES[op["base"]], [st(s) for s in op["states"]]), op)
    elif kind == "process_read_char":
        check_states(o.process_read_char(BASES[op["base"]], [st(s) for s in op["states"]]), op)
    elif kind == "left_markers_search":
        res = o.left_markers_search(*op["sa"])
        if "expect" in op:
            assert res == [tuple(x) for x in op["expect"]], res
        if "expect_first_marker_parity" in op:
            assert (res[0][0] % 2 == 0) == (op["expect_first_marker_parity"] == "even")
    elif kind == "vbwt_jumps":
        check_states(o.vbwt_jumps(st(op["state"])), op)
    elif kind == "encapsulated":
        check_states(o.encapsulated([st(s) for s in op["states"]]), op)
    elif kind == "locus_finder":
        base, loci = o.locus_finder(st(op["state"]))
        assert base == op["expect_base"] and loci == [tuple(x) for x in op["expect_loci"]], (base, loci)
    elif kind == "locus_finder_traversing_only":
        # assign_traversing_loci alone: same as the full finder on a state whose traversed path is dropped
        s = st(op["state"])
        base, loci = o.locus_finder((s[0], s[1], [], s[3]))
        assert base == op["expect_base"] and loci == [tuple(x) for x in op["expect_loci"]], (base, loci)
    elif kind == "set_par_map":
        o.set_par_map({int(k): tuple(v) for k, v in op["value"].items()})
    elif kind == "check_site_uniqueness":
        assert o.check_site_uniqueness_throws(st(op["state"])) == op["expect_throws"]
    elif kind == "assign_loci":
        base, used, loci = o.assign_loci([tuple(x) for x in op["loci"]], [st(s) for s in op.get("traversed_of", [])])
        assert base == op["expect_base"], base
        if "expect_used" in op:
            assert used == op["expect_used"], used
        assert loci == [tuple(x) for x in op["expect_loci"]], loci
    elif kind == "unique_site_paths":
        nonvar, entries = o.unique_site_paths([st(s) for s in op["states"]])
        if "expect_nonvariant" in op:
            assert nonvar == op["expect_nonvariant"], nonvar
        if "expect_sites" in op:
            assert [e[0] for e in entries] == op["expect_sites"], entries
        if "expect_states" in op:
            assert [[list(x) for x in e[1]] for e in entries] == op["expect_states"], entries
        if "expect_loci" in op:
            assert [[list(x) for x in e[2]] for e in entries] == op["expect_loci"], entries
    elif kind == "select_forced":
        res = o.select_forced([st(s) for s in op["states"]], op["forced"])
        exp = dict(op["expect"])
        exp["loci"] = [tuple(x) for x in exp["loci"]]
        assert res == exp, res
    elif kind == "rng_raw":
        assert o.rng_raw(op["seed"], len(op["expect"])).tolist() == op["expect"]
    elif kind == "rng_generate":
        for mode in (0, 1):  # both libstdc++ variants agree on the reference's known answers
            assert o.rng_generate(op["seed"], op["min"], op["max"], len(op["expect"]), mode).tolist() == op["expect"]
    elif kind == "traverse":
        nodes, remaining, final = o.traverse(op["pos"], [tuple(x) for x in op["path"]], op["read_size"])
        if "expect_nodes" in op:
            got = [{k: n[k] for k in e} for n, e in zip(nodes, op["expect_nodes"])]
            assert got == op["expect_nodes"] and len(nodes) == len(op["expect_nodes"]), nodes
        if "expect_loci" in op:
            assert [(n["site"], n["allele"]) for n in nodes] == [tuple(x) for x in op["expect_loci"]], nodes
        if "expect_last_coords" in op:
            assert list(final) == op["expect_last_coords"], (nodes, final)
        if "expect_remaining" in op:
            assert remaining == op["expect_remaining"]
    elif kind == "traverse_first":
        nodes, _, _ = o.traverse(op["pos"], [tuple(x) for x in op["path"]], op["read_size"])
        assert [nodes[0]["start"], nodes[0]["end"]] == op["expect_coords"], nodes
    elif kind == "record_per_base":
        for _ in range(op.get("times", 1)):
            o.record_per_base([st(s) for s in op["states"]], op["read_size"])
    elif kind == "dummy_cov_nodes":
        d = o.dummy_cov_nodes([st(s) for s in op["states"]], op["read_size"])
        ra = o.random_access()
        got = []
        for p in op["positions"]:
            node = int(ra[p][0])
            got.append(list(d[node]) if node in d else None)
        assert got == op["expect"], got
    elif kind == "record_loci":
        for _ in range(op.get("times", 1)):
            o.record_loci([tuple(x) for x in op["loci"]])
    elif kind == "extract_max_cov_allele":
        seq_, cov = o.extract_max_cov_allele(op["site"])
        assert seq_ == op["expect_sequence"] and cov == op["expect_cov"], (seq_, cov)
    elif kind == "expect_max_haplogroup":
        assert list(o.max_cov_haplogroup(op["site_index"])) == op["expect"]
    elif kind == "expect_depth":
        d = o.depth_stats()
        assert d["mean"] == op["mean"] and d["variance"] == op["variance"], d
        assert d["num_sites_noCov"] == op["noCov"] and d["num_sites_total"] == op["total"], d
    elif kind == "expect_target_map":
        got = {str(k): [list(t) for t in v] for k, v in o.target_map().items()}
        assert got == op["value"], got
    elif kind == "expect_par_map":
        got = {str(k): list(v) for k, v in o.par_map().items()}
        assert got == op["value"], got
    elif kind == "expect_targets":
        ra = o.random_access()
        assert ra[:, 2].tolist() == op["sites"], ra[:, 2].tolist()
        assert ra[:, 3].tolist() == op["alleles"], ra[:, 3].tolist()
    elif kind == "expect_node_ids":
        ra = o.random_access()
        assert [[int(r[4]), int(r[5])] for r in ra] == op["value"]
    elif kind == "expect_bubble_site_indices":
        assert [(s - 5) // 2 for s, _, _ in o.bubble_order()] == op["value"]
    elif kind == "expect_bubble_pos":
        got = {str(s): p for s, p, _ in o.bubble_order()}
        assert got == op["value"], got
    elif kind == "prefix_diffs":
        assert o.prefix_diffs(op["kmers"]) == op["expect"]
    elif kind == "all_kmers_prefix":
        km = o.all_kmers(op["k"]).tolist()
        assert km[:len(op["expect_first"])] == op["expect_first"]
        assert km[-1] == op["expect_last"] and len(km) == op["expect_count"]
    else:
        raise AssertionError("unknown op " + kind)
