"""State-level parity of the HIP path: the reference's SearchState vectors (tests/genotype/quasimap/search/
test_vBWT_jump.cpp:55-405, test_encapsulated_search.cpp:28-254, test_BWT_search.cpp, the search_read_backwards cases of
test_quasimap.cpp — transcribed in tests/golden/search.json and quasimap.json) run on the device's own search code through
the test hooks of the C ABI (gmx_debug_*), and the per-task final states of the PRODUCTION pipeline (seed / probe / extend
kernels, instance lanes, large-capacity passes) against the oracle's search_read_backwards on random nested PRGs.

States are compared as order-insensitive sets of mapping instances: a state over [lo, hi] stands for hi - lo + 1 instances
(SA index, traversed path, traversing path). The reference merges neighbouring instances of one allele into an interval
(encapsulated_search.cpp:64-85) and the device keeps single positions in text form; the instances are the same."""
import numpy as np
import pytest

from golden_runner import BASES, load_cases, prg_ints, seq, st
from gramtools_amd import Index, Quasimapper
from gramtools_amd.synth import bracket_to_ints, nested_prg, simulate_graph_reads
from oracle import Oracle

pytestmark = pytest.mark.gpu


def instances(states):
    out = []
    for s in states:
        lo, hi = int(s[0]), int(s[1])
        tvd = tuple((int(m), int(a)) for m, a in (s[2] if len(s) > 2 else []))
        tvg = tuple(int(x[0]) for x in (s[3] if len(s) > 3 else []))
        out += [(i, tvd, tvg) for i in range(lo, hi + 1)]
    return sorted(out)


def encapsulated_instances(qm, states):
    """the device's handle_allele_encapsulated_states on `states`, as instances"""
    if not states:
        return []
    inside, outside = qm.debug_encapsulate(states)
    return sorted(instances(inside) + [(i, (), ()) for i in outside])


STATE_OPS = {"search_read_backwards", "search_base_backwards", "process_read_char", "process_read_char_from_kmer", "expect_kmer",
             "vbwt_jumps", "encapsulated"}


def _state_cases():
    out = []
    for f in ("search.json", "quasimap.json"):
        for c in load_cases(f):
            if c.get("expect_build_error") or not any(op["op"] in STATE_OPS | {"quasimap_read"} for op in c["ops"]):
                continue
            out.append((f, c))
    return out


CASES = _state_cases()


@pytest.mark.parametrize("fname,case", CASES, ids=[f"{f}:{c['name']}" for f, c in CASES])
def test_state_level_golden_vectors_on_gpu(fname, case):
    ints = prg_ints(case["prg"])
    k = case.get("k", 2) or 1
    ix = Index(ints, k)
    qm = Quasimapper(ix, forward_only=True)
    qm.debug_keep_states()
    o = Oracle(ints, k, all_kmers=True)  # (pinned by the same vectors: tests/test_oracle_golden.py)
    ran = 0
    for op in case["ops"]:
        kind = op["op"]
        if kind in ("search_read_backwards", "quasimap_read"):
            r = seq(op["read"])
            if r.size < k or r.size == 0:
                continue
            want = instances(o.search_read_backwards(r))
            if kind == "search_read_backwards" and "expect" in op:
                assert want == instances([st(s) for s in op["expect"]])
            # (1) the search loop from the k-mer index entry, then the device's encapsulation
            got = encapsulated_instances(qm, qm.debug_search(r))
            assert got == want, (op, got, want)
            # (2) the production pipeline: one launch, the task's final states wherever the kernels left them
            qm.map_reads(r, np.array([0, r.size], dtype=np.uint64), np.array([op.get("seed", 42)], dtype=np.uint32))
            states, tier = qm.debug_final_states(0, 0)
            assert encapsulated_instances(qm, states) == want, (op, tier, states, want)
        elif kind in ("search_base_backwards", "process_read_char"):
            given = [st(s) for s in op["states"]]
            base = BASES[op["base"]]
            lf_only = kind == "search_base_backwards"
            want = (o.search_base_backwards if lf_only else o.process_read_char)(base, given)
            if "expect" in op:
                assert instances(want) == instances([st(s) for s in op["expect"]])
            got = qm.debug_search(np.array([base], dtype=np.uint8), given, lf_only=lf_only)
            assert instances(got) == instances(want), (op, got, want)
        elif kind == "process_read_char_from_kmer":
            km = seq(op["kmer"])
            want = o.process_read_char(BASES[op["base"]], o.kmer_states(km))
            got = qm.debug_search(np.concatenate([[BASES[op["base"]]], km]).astype(np.uint8))
            assert instances(got) == instances(want), (op, got, want)
        elif kind == "expect_kmer":
            km = seq(op["kmer"])
            want = o.kmer_states(km)
            got = qm.debug_search(km)  # nothing left to match: the k-mer index entry's states as the kernels load them
            assert instances(got) == instances(want or []), (op, got, want)
        elif kind == "vbwt_jumps":
            state = st(op["state"])
            jumped = o.vbwt_jumps(state)
            if "expect" in op:
                assert instances(jumped) == instances([st(s) for s in op["expect"]])
            for base in (1, 2, 3, 4):
                # marker pass + LF step of the device == LF step (oracle) of the state and of what the GOLDEN jump produced
                want = o.search_base_backwards(base, [state] + jumped)
                got = qm.debug_search(np.array([base], dtype=np.uint8), [state])
                assert instances(got) == instances(want), (op, base, got, want)
                assert instances(want) == instances(o.process_read_char(base, [state]))
        elif kind == "encapsulated":
            given = [st(s) for s in op["states"]]
            want = o.encapsulated(given)
            if "expect" in op:
                assert instances(want) == instances([st(s) for s in op["expect"]])
            assert encapsulated_instances(qm, given) == instances(want), op
        else:
            continue
        ran += 1
    assert ran > 0


def _random_case(seed):
    rng = np.random.default_rng(seed)
    s = nested_prg(seed, n_top=int(rng.integers(1, 6)), max_depth=int(rng.integers(1, 4)), seq_max=int(rng.integers(1, 7)))
    if seed % 3 == 0:  # low-complexity PRGs: repeats, many states per read
        s = s.replace("t", "a").replace("g", "c")
    prg = bracket_to_ints(s)
    L, k = int(rng.integers(4, 25)), int(rng.integers(1, 5))
    reads = simulate_graph_reads(prg, 50, L, seed + 100)
    reads += [rng.integers(1, 5, size=L).astype(np.uint8) for _ in range(6)]
    return prg, k, [r for r in reads if len(r) >= k]


@pytest.mark.parametrize("seed", range(24))
def test_final_states_of_the_production_pipeline_match_oracle(seed):
    """Every (read, orientation) task of one launch over a random nested PRG: the final SearchStates the kernels left —
    in the fast pass's finals[], the instance lanes' pools or a large-capacity slot — equal the oracle's
    search_read_backwards (quasimap.cpp:227-256) after the device's own encapsulation step."""
    prg, k, reads = _random_case(seed)
    ix = Index(prg, k)
    caps = dict(max_states=64, max_path_nodes=128) if seed % 4 == 3 else {}
    qm = Quasimapper(ix, **caps)
    qm.debug_keep_states()
    o = Oracle(prg, k, all_kmers=True)
    flat = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    qm.map_reads(flat, offs, np.arange(len(reads), dtype=np.uint32))
    tiers = {}
    for i, r in enumerate(reads):
        for ori in (0, 1):
            read = o.reverse_complement(r) if ori else r
            want = instances(o.search_read_backwards(read))
            try:
                states, tier = qm.debug_final_states(i, ori)
            except Exception as exc:  # the last tier keeps no states (GMX_ECAP)
                assert getattr(exc, "code", None) == -4, exc
                tiers[3] = tiers.get(3, 0) + 1
                continue
            tiers[tier] = tiers.get(tier, 0) + 1
            assert encapsulated_instances(qm, states) == want, (seed, i, ori, tier, states, want)
    assert tiers.get(0, 0) > 0
