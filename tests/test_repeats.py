"""Multi-mapping at scale ("replace 5 % of the reference by copies of 1-5 kb segments"): reads inside the
copies have several mapping instances, so the equivalence classes, the seeded draw (both libstdc++ algorithms) and the
non-variant-instance rule decide what is recorded. Host emulation (not gpu) and HIP (gpu) against the oracle."""
import numpy as np
import pytest

from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg

from common import canonical_cov, hostemu_map, oracle_map


def _repeat_workload(G, n_sites, n_reads, seed, copies=6, seg=1500):
    rng = np.random.default_rng(seed)
    ref = random_ref(G, seed)
    src = int(rng.integers(0, G - seg))
    for c in range(copies):  # paste the same segment at several places
        dst = int(rng.integers(0, G - seg))
        ref[dst:dst + seg] = ref[src:src + seg]
    prg, pos, alts, n_alts = snp_prg(ref, n_sites, seed + 1, multi_allelic_frac=0.1)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, n_reads, 150, seed + 2)
    return prg, reads


@pytest.mark.parametrize("seed,mode", [(1, 0), (2, 1)])
def test_host_emulation_matches_oracle(seed, mode):
    prg, reads = _repeat_workload(30000, 400, 1500, seed)
    seeds = master_seeds(seed, [1500])
    want = oracle_map(prg, 8, list(reads), seeds, rng_mode=mode, threads=4)
    got, _, rc = hostemu_map(prg, 8, list(reads), seeds, rng_mode=mode)
    assert rc == 0
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mode", [(3, 0), (4, 1)])
def test_gpu_matches_oracle(seed, mode):
    prg, reads = _repeat_workload(200000, 2700, 20000, seed, copies=10, seg=3000)
    seeds = master_seeds(seed, [20000])
    want = oracle_map(prg, 10, list(reads), seeds, rng_mode=mode, threads=8)
    qm = Quasimapper(Index(prg, 10), rng_mode=mode)
    qm.map_reads(reads.reshape(-1), flat_offsets(20000, 150), seeds)
    assert canonical_cov(qm.coverage()) == want


@pytest.mark.gpu
@pytest.mark.parametrize("copies", [5, 17, 40, 63, 64, 70])
def test_gpu_routes_by_number_of_copies(copies):
    """The routes a read inside a repeat can take, by its number of mapping instances: up to 5 on the lane's own stack,
    6..64 as instance lanes (up to 16 of them recorded by the cooperative coverage instance, more by the one-lane
    instances up to their scratch sizes, then the large scratch), more than 64 through the split search."""
    prg, reads = _repeat_workload(400000, 5000, 6000, 40 + copies, copies=copies - 1, seg=2000)  # the source + its pastes
    seeds = master_seeds(copies, [6000])
    want = oracle_map(prg, 10, list(reads), seeds, threads=8)
    qm = Quasimapper(Index(prg, 10))
    qm.map_reads(reads.reshape(-1), flat_offsets(6000, 150), seeds)
    assert canonical_cov(qm.coverage()) == want
    q = qm.queue_counts()
    if 6 <= copies <= 64:
        assert q["inst_mapped"] > 0, q
    if copies > 64:
        assert q["big_mapped"] > 0, q
