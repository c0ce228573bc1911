"""The grouped log of sites without dense group counters, driven to its limits (no assumed bound per read).
After every batch the engine reads the log's real fill back; a task that found the log full has recorded nothing, its
queue entry is redone after a drain (gmx_engine.hip: log_settle, launch_log_replay). The reference has no such limit
(grouped_allele_counts.cpp:17-49 is a map insert), so a tiny log must change nothing but the number of drains."""
import numpy as np
import pytest

from common import oracle_map, canonical_cov, flatten_reads
from gramtools_amd import Index, Quasimapper, master_seeds, GmxError
from gramtools_amd.synth import (random_ref, mixed_variant_prg, simulate_haplotype_reads, nested_prg, bracket_to_ints,
                                 simulate_graph_reads)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_multi_allelic_site_uses_the_log(monkeypatch):
    monkeypatch.setenv("GMX_DENSE_MAX_ALLELES", "2")


@pytest.mark.parametrize("cap", [16, 300, 1 << 20])
def test_tiny_log_gives_the_oracles_counts_flat_prg(cap):
    ref = random_ref(6000, 3)
    prg, sites = mixed_variant_prg(ref, 150, 4, max_alleles=7)
    reads = simulate_haplotype_reads(ref, sites, 3000, 60, 150, 5)
    seeds = master_seeds(42, [len(reads)])
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    ix = Index(prg, 7)
    assert ix.uses_grouped_log
    qm = Quasimapper(ix, log_cap_words=cap, max_batch_reads=1000)  # three batches per call
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    q = qm.queue_counts()
    assert (q["log_replays"] > 0) == (cap < 1 << 20), q


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tiny_log_nested_prg_with_multi_mapping(seed):
    s = nested_prg(seed + 20, n_top=8, max_depth=3).replace("t", "a")   # repeats: the general and cooperative instances record too
    prg = bracket_to_ints(s)
    reads = simulate_graph_reads(prg, 600, 14, seed)
    seeds = master_seeds(seed, [len(reads)])
    want = oracle_map(prg, 3, reads, seeds)
    qm = Quasimapper(Index(prg, 3), log_cap_words=96, max_batch_reads=250)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert qm.queue_counts()["log_replays"] > 0


def test_a_task_that_needs_more_than_the_whole_log_is_reported():
    ref = random_ref(3000, 8)
    prg, sites = mixed_variant_prg(ref, 100, 9, max_alleles=5)
    reads = simulate_haplotype_reads(ref, sites, 400, 100, 150, 10)
    seeds = master_seeds(1, [len(reads)])
    qm = Quasimapper(Index(prg, 7), log_cap_words=2)   # one record is three words
    flat, offs = flatten_reads(reads)
    with pytest.raises(GmxError) as e:
        qm.map_reads(flat, offs, seeds)
        qm.coverage()
    assert e.value.code == -4


def _wide_site_case(n_reads, seed):
    ref = random_ref(6000, seed)
    prg, sites = mixed_variant_prg(ref, 150, seed + 1, max_alleles=7)
    reads = simulate_haplotype_reads(ref, sites, n_reads, 60, 150, seed + 2)
    seeds = master_seeds(42, [len(reads)])
    return prg, reads, seeds


def test_a_larger_batch_after_a_full_log_replays_the_earlier_batchs_own_lists():
    """batch N + 1 grows the workspace; the replay of batch N must still read batch N's retry lists and inputs
    (log_settle runs before ensure_batch_capacity)."""
    prg, reads, seeds = _wide_site_case(3000, 11)
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    qm = Quasimapper(Index(prg, 7), log_cap_words=64)
    cuts = [0, 300, 1100, 3000]  # every call larger than the one before: the workspace is reallocated twice
    for a, b in zip(cuts[:-1], cuts[1:]):
        flat, offs = flatten_reads(reads[a:b])
        qm.map_reads(flat, offs, seeds[a:b])
    assert canonical_cov(qm.coverage()) == want
    assert qm.queue_counts()["log_replays"] > 0


def test_serial_host_loop_settles_before_its_staging_buffers_are_reused(monkeypatch):
    """GMX_HOST_SERIAL reuses one staging buffer per batch; a batch that found the log full is replayed from
    its own reads and seeds, not the next batch's."""
    monkeypatch.setenv("GMX_HOST_SERIAL", "1")
    prg, reads, seeds = _wide_site_case(2500, 21)
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    qm = Quasimapper(Index(prg, 7), log_cap_words=48, max_batch_reads=400)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert qm.queue_counts()["log_replays"] > 0


def test_device_entry_point_settles_before_it_returns():
    """gmx_map_reads_device with a log-using index: the caller may overwrite its device buffers in stream order
    after the call."""
    import torch
    prg, reads, seeds = _wide_site_case(2000, 31)
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    qm = Quasimapper(Index(prg, 7), log_cap_words=48)
    for a, b in ((0, 900), (900, 2000)):
        flat, offs = flatten_reads(reads[a:b])
        d_r = torch.from_numpy(flat.copy()).cuda()
        d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_s = torch.from_numpy(np.ascontiguousarray(seeds[a:b]).view(np.int32).copy()).cuda()
        qm.map_reads_device(d_r, d_o, d_s, b - a)
        d_r.zero_()   # stream-ordered reuse of the caller's buffers
        d_s.zero_()
        d_o.zero_()
    assert canonical_cov(qm.coverage()) == want
    assert qm.queue_counts()["log_replays"] > 0
