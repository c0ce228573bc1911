"""The N > 1 path under gloo with world_size 2 on CPU: shard by global read index with the global seed
stream, map each shard, all-reduce the uint32 totals, finalise — result must equal the single-process oracle.
The per-shard mapper is the test-only host emulation of the device logic (no GPU in this container); on GPUs
the same function runs with gramtools_amd.distributed.gpu_map_shard."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import oracle_map, hostemu_map, canonical_cov, flatten_reads
from gramtools_amd import Index
from gramtools_amd.distributed import quasimap_reads_sharded, shard_range, global_seeds
from gramtools_amd.synth import (nested_prg, bracket_to_ints, simulate_graph_reads, random_ref, snp_prg, simulate_snp_reads,
                                 mixed_variant_prg, simulate_haplotype_reads)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, prg, k, flat, offs, master_seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reads_of = lambda f, o: [f[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]

        def map_shard(r, o, seeds):
            raw, _, rc = hostemu_map(prg, k, reads_of(r, o), seeds, return_raw=True)
            assert rc == 0
            return raw
        cov = quasimap_reads_sharded(Index(prg, k, threads=1), flat, offs, master_seed, map_shard, dist=dist)
        q.put((rank, canonical_cov(cov)))
    finally:
        dist.destroy_process_group()


def _run(prg, k, reads, master_seed, world=2):
    flat, offs = flatten_reads(reads)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, prg, k, flat, offs, master_seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return dict(res)


def test_shard_ranges_partition_the_reads():
    for n in (0, 1, 7, 10000, 10001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, i) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_two_ranks_equal_single_process_oracle_snp_prg():
    ref = random_ref(5000, 3)
    prg, pos, alts, n_alts = snp_prg(ref, 60, 4, multi_allelic_frac=0.3)
    reads = list(simulate_snp_reads(ref, pos, alts, n_alts, 401, 150, 5))
    want = oracle_map(prg, 6, reads, global_seeds(42, [len(reads)]))
    got = _run(prg, 6, reads, 42)
    assert got[0] == want and got[1] == want


def test_two_ranks_equal_single_process_oracle_nested_repeats():
    s = nested_prg(9, n_top=4).replace("t", "a").replace("g", "c")  # multi-mapping: exercises the seeded selection
    prg = bracket_to_ints(s)
    reads = simulate_graph_reads(prg, 120, 12, 77)
    want = oracle_map(prg, 3, reads, global_seeds(7, [len(reads)]))
    got = _run(prg, 3, reads, 7)
    assert got[0] == want and got[1] == want


def test_two_ranks_exchange_the_grouped_log_of_many_allele_sites(monkeypatch):
    """Sites with 6-7 alleles keep their grouped counts in the append log (index built with the round-2 limit of 5 dense
    alleles), which every rank must receive from every other: the totals equal the single-process oracle
    and the log really was in play."""
    monkeypatch.setenv("GMX_DENSE_MAX_ALLELES", "5")
    ref = random_ref(4000, 11)
    prg, sites = mixed_variant_prg(ref, 80, 12, max_alleles=7)
    assert any(len(al) >= 6 for _, _, al in sites)
    reads = simulate_haplotype_reads(ref, sites, 500, 60, 150, 13)
    want = oracle_map(prg, 6, reads, global_seeds(42, [len(reads)]))
    got = _run(prg, 6, reads, 42)
    assert got[0] == want and got[1] == want
    assert any(len(ids) >= 1 and max(ids) >= 5 for site in want["grouped"] for ids in site)


@pytest.mark.parametrize("n_reads", [5, 203])
def test_eight_ranks_ragged_shards_empty_ranks_and_a_grouped_log(monkeypatch, n_reads):
    """World size 8 as on a full MI355X node: 203 reads give shards of 26 and 25 reads; 5 reads leave three ranks with
    nothing to map — their all-reduce contribution is zeros and their log is empty — and the PRG's 6-7-allele sites keep
    their grouped counts in the append log every rank gathers from every other. Every rank ends with the single-process
    oracle's coverage."""
    monkeypatch.setenv("GMX_DENSE_MAX_ALLELES", "5")
    ref = random_ref(3000, 21)
    prg, sites = mixed_variant_prg(ref, 60, 22, max_alleles=7)
    assert any(len(al) >= 6 for _, _, al in sites)
    reads = simulate_haplotype_reads(ref, sites, n_reads, 60, 150, 23)
    want = oracle_map(prg, 6, reads, global_seeds(42, [len(reads)]))
    sizes = [shard_range(n_reads, 8, r)[1] - shard_range(n_reads, 8, r)[0] for r in range(8)]
    assert (0 in sizes) == (n_reads < 8) and len(set(sizes)) >= 2
    got = _run(prg, 6, reads, 42, world=8)
    assert all(got[r] == want for r in range(8))
