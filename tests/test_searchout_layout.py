"""The probe pipeline (gmx_probe_kernel -> gmx_extend_kernel; GMX_NO_SEEDED=1 forces it on every index) with BOTH member
orders of SearchOut compiled: the regular library and lib/libgmx_alt.so (-DGMX_SEARCHOUT_ALT), each in a process of its own.

Round 2 found gmx_probe_kernel appending mapped tasks to dead_list with one member order. Round 3's reading of the ISA
(profiles/round3/searchout_layout_bug/, HISTORY.md §4.5): a compiler defect, not undefined behaviour — the exec-masked
VGPR copy of the cover_general_list pointer (a spilled SGPR pair) was emitted in a sibling block. finish_lane now
addresses its queues as one base pointer + an integer index, which leaves no divergent pointer select to get wrong;
these tests keep both layouts honest against the oracle (IT2 / IT3 are the cases that caught it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from common import oracle_map, canonical_cov, flatten_reads
from golden_runner import all_cases, prg_ints, seq
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import nested_prg, bracket_to_ints, simulate_graph_reads
bad = []
# the CLI-level golden cases (IT1-IT3): map_reads ops against the oracle
for f, c in all_cases():
    if not any(op["op"] == "map_reads" for op in c["ops"]) or c.get("expect_build_error"):
        continue
    ints = prg_ints(c["prg"])
    for op in c["ops"]:
        if op["op"] != "map_reads":
            continue
        reads = [seq(r) for r in op["reads"]]
        seeds = master_seeds(op["master_seed"], [len(reads)])
        qm = Quasimapper(Index(ints, c["k"]))
        flat, offs = flatten_reads(reads)
        qm.map_reads(flat, offs, seeds)
        if canonical_cov(qm.coverage()) != oracle_map(ints, c["k"], reads, seeds):
            bad.append(c["name"])
# random nested PRGs: short reads are finished by the probe kernel itself (the queue that was mis-addressed)
for seed in range(12):
    rng = np.random.default_rng(seed)
    prg = bracket_to_ints(nested_prg(seed, n_top=int(rng.integers(2, 8)), max_depth=int(rng.integers(1, 4)), seq_max=6))
    k = int(rng.integers(2, 5))
    reads = simulate_graph_reads(prg, 80, int(rng.integers(k + 1, 30)), seed + 50)
    seeds = master_seeds(seed, [len(reads)])
    qm = Quasimapper(Index(prg, k))
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    if canonical_cov(qm.coverage()) != oracle_map(prg, k, reads, seeds):
        bad.append("nested seed %d" % seed)
print(json.dumps({"bad": bad}))
"""


@pytest.mark.parametrize("lib", ["libgmx.so", "libgmx_alt.so"])
@pytest.mark.parametrize("no_seeded", ["1", ""])
def test_probe_pipeline_with_both_searchout_layouts(lib, no_seeded):
    path = os.path.join(ROOT, "gramtools_amd", "lib", lib)
    assert os.path.exists(path), f"{path} missing: run python -m gramtools_amd.build"
    env = dict(os.environ, GMX_LIB=path)
    if no_seeded:
        env["GMX_NO_SEEDED"] = "1"
    else:
        env.pop("GMX_NO_SEEDED", None)
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"bad": []}
