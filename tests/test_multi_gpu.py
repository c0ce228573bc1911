"""The several-GPU path of the product (gmx.h: gmx_group_*, gmx_comm_*; `gram genotype --devices`), exercised on ONE GPU:
two engines on device 0 shard the reads and exchange their totals through the very routine `--devices 0-7` uses
(peer copies instead of RCCL when both engines share a device). The PRG has sites with 6-7 alleles, whose grouped
counts live in the append log that must be exchanged too, and the result must equal the
single-process oracle — whatever the number of engines (the seeds are the global master stream's)."""
import json
import subprocess

import numpy as np
import pytest

from common import oracle_map, canonical_cov, flatten_reads
from gramtools_amd import Index, Quasimapper, QuasimapperGroup, master_seeds
from gramtools_amd.build import build_gram
from gramtools_amd.synth import random_ref, mixed_variant_prg, simulate_haplotype_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _log_for_six_alleles(monkeypatch):
    """Sites of up to 8 alleles get dense group counters since round 3; these tests are about the append log and its
    exchange, so the index is built with the round-2 limit (sites with 6-7 alleles use the log)."""
    monkeypatch.setenv("GMX_DENSE_MAX_ALLELES", "5")


def _workload(seed=3, n_reads=3000):
    ref = random_ref(6000, seed)
    prg, sites = mixed_variant_prg(ref, 120, seed + 1, max_alleles=7)
    assert any(len(al) >= 6 for _, _, al in sites)
    reads = simulate_haplotype_reads(ref, sites, n_reads, 60, 150, seed + 2)
    return prg, reads, master_seeds(42, [len(reads)])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["2-engines", "3-engines"])
def test_group_of_engines_equals_single_process_oracle(devices):
    prg, reads, seeds = _workload()
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    ix = Index(prg, 7)
    assert ix.uses_grouped_log
    grp = QuasimapperGroup(ix, devices)
    flat, offs = flatten_reads(reads)
    grp.map_reads(flat[:int(offs[1000])], offs[:1001], seeds[:1000])          # two calls: shards differ per call
    grp.map_reads(flat[int(offs[1000]):], offs[1000:] - offs[1000], seeds[1000:])
    part = grp.coverage(1)
    assert 0 < part.stats.all_reads_count < 2 * len(reads)                     # a shard, before the exchange
    grp.allreduce()
    for member in range(len(devices)):                                         # afterwards every engine holds the totals
        assert canonical_cov(grp.coverage(member)) == want
    assert len(grp.coverage(0).raw_grouped_log) > 0


def test_library_communicator_world_size_1():
    """gmx_comm_*: RCCL driven from inside the library (what bench.py --gpus N uses on every rank). World size 1 here:
    id, communicator, in-place all-reduce of the fused block and the log exchange run; the coverage must not change."""
    import os
    import torch
    import torch.distributed as dist
    from gramtools_amd.distributed import CoverageComm
    torch.cuda.init()
    prg, reads, seeds = _workload(5, 1500)
    qm = Quasimapper(Index(prg, 7))
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    before = canonical_cov(qm.coverage())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = CoverageComm(qm, dist)
        comm.allreduce()
        qm.sync()
        comm.close()
    finally:
        dist.destroy_process_group()
    assert canonical_cov(qm.coverage()) == before
    assert before == oracle_map(prg, 7, reads, seeds, threads=8)


def test_gram_devices_option_gives_the_same_files(tmp_path):
    """`gram genotype --devices 0,0` (two engines, two host threads, one exchange) writes byte-identical coverage files
    and the same counters as `--device 0`."""
    prg, reads, _ = _workload(7, 6000)
    (tmp_path / "prg").write_bytes(np.asarray(prg, dtype="<u4").tobytes())
    with open(tmp_path / "r.fq", "w") as fh:
        for i, r in enumerate(reads):
            s = "".join("ACGT"[b - 1] for b in r)
            fh.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
    outs = []
    for name, dev in (("one", ["--device", "0"]), ("two", ["--devices", "0,0"])):
        out = tmp_path / name
        r = subprocess.run([build_gram(), "genotype", "--gram_dir", str(tmp_path), "--reads", str(tmp_path / "r.fq"),
                            "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", "7", "--genotype_dir", str(out),
                            "--seed", "42", "--max_threads", "4"] + dev, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        counts = [l for l in r.stdout.splitlines() if l.startswith("Count ")]
        files = {f: (out / "coverage" / f).read_bytes() for f in
                 ("allele_sum_coverage", "allele_base_coverage.json", "grouped_allele_counts_coverage.json")}
        outs.append((counts, files, json.loads((out / "read_stats.json").read_text())["Read_depth"]))
    assert outs[0] == outs[1]
    want = oracle_map(prg, 7, reads, master_seeds(42, [len(reads)]), threads=8)
    asum = [[int(x) for x in l.split()] for l in outs[1][1]["allele_sum_coverage"].decode().splitlines()]
    assert asum == want["allele_sum"]


def test_engines_of_one_index_share_its_device_copy():
    """Engines made of one index on one device share the device copy of its tables (reference counted, gmx_engine.hip:
    GmxDeviceIndex): the second engine maps as the first, keeps working when the first is destroyed, and an engine of ANOTHER
    index that happens to be built afterwards does not pick the old copy up (the key is a serial, not an address)."""
    import gc
    prg, reads, seeds = _workload(seed=5, n_reads=2000)
    flat, offs = flatten_reads(reads)
    want = oracle_map(prg, 5, reads, seeds, threads=8)
    ix = Index(prg, 5)
    a, b = Quasimapper(ix), Quasimapper(ix)
    a.map_reads(flat, offs, seeds)
    b.map_reads(flat, offs, seeds)
    assert canonical_cov(a.coverage()) == want and canonical_cov(b.coverage()) == want
    a.close()
    b.reset()
    b.map_reads(flat, offs, seeds)
    assert canonical_cov(b.coverage()) == want
    b.close()
    del ix
    gc.collect()
    prg2, reads2, seeds2 = _workload(seed=9, n_reads=1500)
    flat2, offs2 = flatten_reads(reads2)
    ix2 = Index(prg2, 5)
    c = Quasimapper(ix2)
    c.map_reads(flat2, offs2, seeds2)
    assert canonical_cov(c.coverage()) == oracle_map(prg2, 5, reads2, seeds2, threads=8)
