"""Two launches in flight inside one engine (round 6; gmx_engine::twin): the host feeds hand consecutive launches to
the engine's two workspaces in turn — each with streams of its own — which record into the SAME accumulators. Forced on here
(GMX_TWIN=1; by default: nested PRGs and indexes of 2 GB and more) with launches of a few hundred reads, so that every call is a
dozen launches alternating between the workspaces: against the oracle, against one workspace (GMX_TWIN=0), across resets (the
launch that carries a queued reset, the twin's launch behind it), with timing on, and through `gram` on BGZF and plain FASTQ
(the ingest's slots are released behind BOTH streams)."""
import json

import numpy as np
import pytest

from common import canonical_cov, flatten_reads, oracle_map
from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads, pack_reads_2bit, _lib
from gramtools_amd.synth import (bracket_to_ints, flat_offsets, nested_prg, random_ref, simulate_graph_reads, simulate_snp_reads,
                                 snp_prg)

pytestmark = pytest.mark.gpu


def _raw(cov):
    return (cov.raw_allele_sum.copy(), cov.raw_per_base.copy(), cov.raw_grouped.copy(), cov.stats.as_dict())


def _same(a, b):
    return all((x == y).all() for x, y in zip(a[:3], b[:3])) and a[3] == b[3]


@pytest.mark.parametrize("feed", ["planes", "2bit"])
def test_flat_prg_many_launches_alternating(monkeypatch, feed):
    ref = random_ref(40000, 5)
    prg, pos, alts, n_alts = snp_prg(ref, 500, 6, multi_allelic_frac=0.2)
    n = 20000
    reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 7)
    seeds = master_seeds(11, [n])
    offs = flat_offsets(n, 150)
    flat = np.ascontiguousarray(reads.reshape(-1))
    want = oracle_map(prg, 8, list(reads), seeds, threads=8)
    ix = Index(prg, 8)
    pk = (pack_reads if feed == "planes" else pack_reads_2bit)(flat, offs, uniform_len=150, pinned=True)
    raws = {}
    for twin in ("1", "0"):
        monkeypatch.setenv("GMX_TWIN", twin)
        qm = Quasimapper(ix, max_batch_reads=1500)  # 14 launches per call
        assert bool(_lib.load().gmx_engine_second_stream(qm.h)) == (twin == "1")
        qm.map_reads_packed(pk, seeds)
        cov = qm.coverage()
        assert canonical_cov(cov) == want, f"GMX_TWIN={twin}"
        raws[twin] = _raw(cov)
        # a second job behind a queued (asynchronous) reset, timing on: the same arrays again
        qm.enable_timing(True)
        qm.reset(stream=0)
        qm.map_reads_packed(pk, seeds)
        qm.map_reads_packed(pk, seeds)
        again = qm.coverage()
        tm = qm.timing()
        assert tm["search_launches"] == 28
        assert again.stats.as_dict()["all"] == 2 * 2 * n
        assert (again.raw_allele_sum == 2 * cov.raw_allele_sum).all() and (again.raw_per_base == 2 * cov.raw_per_base).all()
    assert _same(raws["1"], raws["0"])
    pk.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_nested_prg_has_a_twin_by_default_and_matches_the_oracle(monkeypatch, seed):
    prg = bracket_to_ints(nested_prg(seed + 60, n_top=12, max_depth=3))
    reads = simulate_graph_reads(prg, 3000, 20, seed)
    seeds = master_seeds(seed, [len(reads)])
    want = oracle_map(prg, 4, reads, seeds, threads=8)
    flat, offs = flatten_reads(reads)
    ix = Index(prg, 4)
    monkeypatch.delenv("GMX_TWIN", raising=False)
    qm = Quasimapper(ix, max_batch_reads=400)
    assert _lib.load().gmx_engine_second_stream(qm.h)  # (nested: two workspaces without being asked)
    pk = pack_reads(flat, offs, pinned=True)
    qm.map_reads_packed(pk, seeds)
    assert canonical_cov(qm.coverage()) == want
    qm.reset()
    qm.map_reads(flat, offs, seeds)  # (the byte feed stays on the first workspace)
    assert canonical_cov(qm.coverage()) == want
    pk.close()


def test_gram_device_feeds_with_two_workspaces(tmp_path, monkeypatch):
    """`gram genotype` on BGZF and plain FASTQ decoded on the device, chunks of a few members / 3000 bytes, GMX_TWIN=1: files and
    counters identical to GMX_TWIN=0 and to the host parser."""
    from test_ingest import _gram, bgzf
    rng = np.random.default_rng(2)
    ref = random_ref(3000, 4)
    prg, pos, alts, n_alts = snp_prg(ref, 40, 5, multi_allelic_frac=0.3)
    (tmp_path / "prg").write_bytes(np.array(prg, dtype="<u4").tobytes())
    reads = simulate_snp_reads(ref, pos, alts, n_alts, 6000, 60, 6)
    txt = ["".join("ACGT"[b - 1] for b in r)[:int(rng.integers(25, 61))] for r in reads]
    fq = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(txt)).encode()
    (tmp_path / "a.fq").write_bytes(fq)
    (tmp_path / "a.fq.gz").write_bytes(bgzf(fq, block=5000))
    outs = {}
    for name, f, env in (("host", "a.fq", {"GMX_HOST_FASTQ": "1", "GMX_TWIN": "0"}), ("text-twin", "a.fq", {"GMX_TEXT_CHUNK": "3000", "GMX_TWIN": "1"}),
                         ("bgzf-twin", "a.fq.gz", {"GMX_INGEST_MEMBERS": "2", "GMX_TWIN": "1"}), ("bgzf-one", "a.fq.gz", {"GMX_INGEST_MEMBERS": "2", "GMX_TWIN": "0"})):
        out = tmp_path / name
        r = _gram("genotype", "--gram_dir", str(tmp_path), "--reads", str(tmp_path / f), "--sample_id", "s", "--ploidy", "diploid", "--kmer_size", "6",
                  "--genotype_dir", str(out), "--seed", "99", env=env)
        assert r.returncode == 0, (name, r.stdout)
        outs[name] = ([(out / "coverage" / g).read_bytes() for g in ("allele_sum_coverage", "allele_base_coverage.json", "grouped_allele_counts_coverage.json")],
                      [l for l in r.stdout.splitlines() if l.startswith("Count ")], json.loads((out / "read_stats.json").read_text())["Read_depth"])
    for name in ("text-twin", "bgzf-twin", "bgzf-one"):
        assert outs[name] == outs["host"], name
