"""Parity tests proper: the HIP path (through the C-ABI) against the oracle and the golden vectors."""
import numpy as np
import pytest

from common import oracle_map, canonical_cov, flatten_reads
from golden_runner import all_cases, prg_ints, seq, grouped_key
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import (nested_prg, bracket_to_ints, simulate_graph_reads, random_ref, snp_prg,
                                 simulate_snp_reads, flat_offsets)

pytestmark = pytest.mark.gpu

GPU_OPS = {"quasimap_read", "map_reads", "expect_allele_sum", "expect_allele_base", "expect_grouped", "expect_node_cov",
           "expect_stats", "expect_depth"}


def _gpu_cases():
    out = []
    for f, c in all_cases():
        kinds = {op["op"] for op in c["ops"]}
        if c.get("expect_build_error") or not kinds or not kinds <= GPU_OPS:
            continue
        out.append((f, c))
    return out


CASES = _gpu_cases()


@pytest.mark.parametrize("fname,case", CASES, ids=[f"{f}:{c['name']}" for f, c in CASES])
def test_golden_vectors_on_gpu(fname, case):
    """Every coverage-level known answer of the reference's tests, mapped by the HIP kernels."""
    ints = prg_ints(case["prg"])
    ix = Index(ints, case["k"] or 1)  # structure-only vectors carry k = 0; the engine needs a seed table
    fwd = Quasimapper(ix, forward_only=True)   # quasimap_read (one orientation), as the reference's unit tests call it
    both = Quasimapper(ix)                      # quasimap_forward_reverse, as `gram genotype` runs it
    used = fwd
    for op in case["ops"]:
        kind = op["op"]
        if kind == "quasimap_read":
            r = seq(op["read"])
            fwd.map_reads(r, np.array([0, len(r)], dtype=np.uint64), np.array([op.get("seed", 42)], dtype=np.uint32))
            used = fwd
        elif kind == "map_reads":
            reads = [seq(r) for r in op["reads"]]
            flat, offs = flatten_reads(reads)
            both.map_reads(flat, offs, master_seeds(op["master_seed"], [len(reads)]))
            used = both
        else:
            cov = used.coverage()
            if kind == "expect_allele_sum":
                assert cov.allele_sum_coverage == op["value"]
            elif kind == "expect_allele_base":
                assert cov.allele_base_coverage == op["value"]
            elif kind == "expect_grouped":
                assert [grouped_key(d) for d in cov.grouped_allele_counts] == op["value"]
            elif kind == "expect_node_cov":
                pb = cov.per_base_by_first_pos()
                pi = ix.pos_info()
                got = []
                for p in op["positions"]:
                    first = p - int(pi[p][2])
                    got.append(pb.get(first, []))
                assert got == op["value"]
            elif kind == "expect_depth":  # gmx_compute_coverage_depth against test_read_stats.cpp:140-182
                d = cov.depth_stats()
                assert d["mean"] == op["mean"] and d["variance"] == op["variance"], d
                assert d["num_sites_noCov"] == op["noCov"] and d["num_sites_total"] == op["total"], d
            elif kind == "expect_stats":
                s = cov.stats.as_dict()
                for key, v in op["value"].items():
                    assert s[key] == v


def _random_case(seed):
    rng = np.random.default_rng(seed)
    s = nested_prg(seed, n_top=int(rng.integers(1, 6)), max_depth=int(rng.integers(1, 4)), seq_max=int(rng.integers(1, 7)))
    if seed % 3 == 0:
        s = s.replace("t", "a").replace("g", "c")
    prg = bracket_to_ints(s)
    L, k = int(rng.integers(4, 25)), int(rng.integers(1, 5))
    reads = simulate_graph_reads(prg, 60, L, seed + 100)
    reads += [rng.integers(1, 5, size=L).astype(np.uint8) for _ in range(6)]
    reads.append(np.array([1, 2, 0, 3] * 3, dtype=np.uint8))
    reads.append(np.array([4, 4, 1, 2, 5, 1, 1, 2, 3, 4, 1, 78], dtype=np.uint8))
    reads.append(np.zeros(0, dtype=np.uint8))
    reads = [r for r in reads if len(r) >= k or len(r) == 0]
    seeds = rng.integers(0, 2 ** 32, size=len(reads), dtype=np.uint64).astype(np.uint32)
    return prg, k, reads, seeds


@pytest.mark.parametrize("seed", range(40))
def test_random_nested_prgs_match_oracle(seed):
    prg, k, reads, seeds = _random_case(seed)
    mode = seed % 2
    want = oracle_map(prg, k, reads, seeds, rng_mode=mode)
    qm = Quasimapper(Index(prg, k), rng_mode=mode)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want


def _snp_workload(G, n_sites, n_reads, seed, multi=0.0):
    ref = random_ref(G, seed)
    prg, pos, alts, n_alts = snp_prg(ref, n_sites, seed + 1, multi_allelic_frac=multi)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, n_reads, 150, seed + 2)
    return prg, reads


def test_config1_1kb_10snps_10k_reads():
    """BASELINE.json configs[0]: 1 kb ref + 10 SNPs, 10k x 150 bp, k = 5, --seed 42."""
    prg, reads = _snp_workload(1000, 10, 10000, 1)
    seeds = master_seeds(42, [10000])
    offs = flat_offsets(10000, 150)
    want = oracle_map(prg, 5, list(reads), seeds, threads=8)
    qm = Quasimapper(Index(prg, 5))
    qm.map_reads(reads.reshape(-1), offs, seeds)
    got = canonical_cov(qm.coverage())
    assert got == want
    assert got["stats"]["exact_mapped"] >= 10000


@pytest.mark.parametrize("n_sites", [1500, 3000, 6000, 7400])
def test_dense_sites_matches_oracle(n_sites):
    """A SNP every 20 / 10 / 5 / 4 bp: 7, 15, 30 and 37 loci per read (beyond the 32 the one-lane routine of the general queue holds). The search kernels hand single-instance tasks to the
    coverage kernel as compact records — three (site, allele) pairs, or a run of up to 16 consecutive sites — and
    longer paths as task ids for the general instance; all three routes must give the oracle's coverage."""
    prg, reads = _snp_workload(30000, n_sites, 3000, 40 + n_sites, multi=0.1)
    seeds = master_seeds(13, [3000])
    offs = flat_offsets(3000, 150)
    want = oracle_map(prg, 7, list(reads), seeds, threads=8)
    qm = Quasimapper(Index(prg, 7))
    qm.map_reads(reads.reshape(-1), offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert want["stats"]["exact_mapped"] >= 3000
    counts = qm.queue_counts()
    if n_sites == 1500:
        assert counts["cover_general"] < counts["mapped"] // 4  # runs of up to 16 sites stay compact
    if n_sites == 6000:  # 30 loci fit neither the record nor the per-lane path arena: large-capacity pass
        assert counts["cover_general"] + counts["big_mapped"] > 1500


def test_mtb_like_sample_matches_oracle():
    """A 200 kb slice of the configs[1] recipe (SNP every ~73 bp, k = 10), 20k reads, oracle-checked."""
    prg, reads = _snp_workload(200000, 2700, 20000, 5, multi=0.05)
    seeds = master_seeds(7, [20000])
    offs = flat_offsets(20000, 150)
    want = oracle_map(prg, 10, list(reads), seeds, threads=8)
    qm = Quasimapper(Index(prg, 10))
    qm.map_reads(reads.reshape(-1), offs, seeds)
    assert canonical_cov(qm.coverage()) == want


def test_batching_and_accumulation_are_equivalent():
    """Coverage is additive over calls and independent of the internal batch size (size-independent property)."""
    prg, reads = _snp_workload(50000, 600, 6000, 9)
    seeds = master_seeds(3, [6000])
    offs = flat_offsets(6000, 150)
    ix = Index(prg, 8)
    a = Quasimapper(ix)
    a.map_reads(reads.reshape(-1), offs, seeds)
    b = Quasimapper(ix, max_batch_reads=1000)
    b.map_reads(reads[:2500].reshape(-1), flat_offsets(2500, 150), seeds[:2500])
    b.map_reads(reads[2500:].reshape(-1), flat_offsets(3500, 150), seeds[2500:])
    assert canonical_cov(a.coverage()) == canonical_cov(b.coverage())


def test_device_resident_entry_point_matches_host_entry_point():
    import torch
    prg, reads = _snp_workload(50000, 600, 4000, 21)
    seeds = master_seeds(5, [4000])
    offs = flat_offsets(4000, 150)
    ix = Index(prg, 8)
    a = Quasimapper(ix)
    a.map_reads(reads.reshape(-1), offs, seeds)
    b = Quasimapper(ix)
    d_reads = torch.from_numpy(reads.reshape(-1).copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
    b.map_reads_device(d_reads, d_offs, d_seeds, 4000)
    b.sync()
    assert canonical_cov(a.coverage()) == canonical_cov(b.coverage())


def test_async_reset_is_applied_before_the_next_batch_and_before_any_read_back():
    """gmx_engine_reset_async leaves its memset pending for the next batch's first kernel (same stream) or for whoever
    reads the accumulators first: either way nothing of the batches before the reset may show."""
    import torch
    prg, reads = _snp_workload(50000, 600, 3000, 29)
    seeds = master_seeds(9, [3000])
    ix = Index(prg, 8)
    first, second = slice(0, 1500), slice(1500, 3000)
    ref = Quasimapper(ix)
    ref.map_reads(reads[second].reshape(-1), flat_offsets(1500, 150), seeds[second])
    want = canonical_cov(ref.coverage())
    stream = torch.cuda.current_stream().cuda_stream
    dev = [(torch.from_numpy(reads[s].reshape(-1).copy()).cuda(), torch.from_numpy(flat_offsets(1500, 150).astype(np.int64)).cuda(),
            torch.from_numpy(seeds[s].astype(np.int64)).to(torch.int32).cuda()) for s in (first, second)]
    qm = Quasimapper(ix)
    qm.map_reads_device(*dev[0], 1500, stream=stream)
    qm.reset(stream=stream)  # folded into the next launch
    qm.map_reads_device(*dev[1], 1500, stream=stream)
    qm.sync()
    assert canonical_cov(qm.coverage()) == want
    qm.reset(stream=stream)  # nothing follows: the read-back has to issue it
    st = qm.coverage().stats.as_dict()
    assert st["all"] == 0 and st["exact_mapped"] == 0
    qm.map_reads(reads[second].reshape(-1), flat_offsets(1500, 150), seeds[second])  # host entry point after a pending reset
    assert canonical_cov(qm.coverage()) == want


def test_full_size_properties_mtb_scale():
    """configs[1] scale (4.4 Mb, 60k SNPs, k = 10): too large for the oracle in seconds, so check
    size-independent properties: every error-free read maps exactly once per read (one orientation),
    reversing every read's strand leaves coverage unchanged, and a checksum-of-checksums over batches."""
    prg, reads = _snp_workload(4411532, 60000, 200000, 101)
    n = reads.shape[0]
    seeds = master_seeds(42, [n])
    offs = flat_offsets(n, 150)
    ix = Index(prg, 10)
    a = Quasimapper(ix)
    a.map_reads(reads.reshape(-1), offs, seeds)
    ca = a.coverage()
    st = ca.stats.as_dict()
    assert st["all"] == 2 * n and st["skipped"] == 0
    assert st["exact_mapped"] >= n            # each read maps in its own orientation (plus rare palindromic hits)
    assert st["exact_mapped"] + st["missing_kmer"] + st["no_extension"] == 2 * n
    # strand symmetry: mapping the reverse complements gives identical coverage
    rc = np.ascontiguousarray((5 - reads)[:, ::-1])
    b = Quasimapper(ix)
    b.map_reads(rc.reshape(-1), offs, seeds)
    cb = b.coverage()
    assert (ca.raw_allele_sum == cb.raw_allele_sum).all()
    assert (ca.raw_per_base == cb.raw_per_base).all()
    assert (ca.raw_grouped == cb.raw_grouped).all()
    # every mapped read crossing a site adds exactly one grouped count per level-0 site it covers, and
    # allele-sum totals dominate grouped totals (a group holds >= 1 allele)
    assert int(ca.raw_allele_sum.sum()) >= int(ca.raw_grouped.sum()) > 0


def test_device_accumulators_alias_as_torch_tensors_and_allreduce_in_place():
    """The multi-GPU exchange: ONE torch tensor aliasing the engine's coverage block, RCCL all-reduce in place
    (world_size 1 here: the mechanics, not the scaling)."""
    import os
    import torch
    import torch.distributed as dist
    from gramtools_amd.distributed import allreduce_device_coverage, fused_coverage_tensor
    prg, reads = _snp_workload(20000, 250, 2000, 33)
    seeds = master_seeds(5, [2000])
    qm = Quasimapper(Index(prg, 7))
    qm.map_reads(reads.reshape(-1), flat_offsets(2000, 150), seeds)
    before = qm.coverage()
    fused = fused_coverage_tensor(qm)
    # the block holds every logical count exactly once (plus padding and the counter limbs) — directly, or as a hit
    # of a one-base allele that stands for one count in each of the three structures (gmx_types.h)
    total = int(before.raw_allele_sum.sum()) + int(before.raw_per_base.sum()) + int(before.raw_grouped.sum())
    in_block = int(fused[:-32].sum().item())
    assert in_block <= total <= 3 * in_block and (total - in_block) % 2 == 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        allreduce_device_coverage(qm, dist)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    after = qm.coverage()
    assert canonical_cov(after) == canonical_cov(before)
    # aliasing, not a copy — and what three ranks with identical totals would leave after the exchange: every word of
    # the block tripled between begin and end
    qm.reduce_begin()
    fused.mul_(3)
    qm.reduce_end()
    torch.cuda.synchronize()
    tripled = qm.coverage()
    assert (tripled.raw_allele_sum == 3 * before.raw_allele_sum).all()
    assert (tripled.raw_per_base == 3 * before.raw_per_base).all()
    assert (tripled.raw_grouped == 3 * before.raw_grouped).all()
    assert tripled.stats.as_dict() == {k: 3 * v for k, v in before.stats.as_dict().items()}


@pytest.mark.parametrize("k2", ["0", "12"])
def test_results_do_not_depend_on_the_seed_table_length(monkeypatch, k2):
    """The search is seeded from a longer k-mer table when the PRG is large enough (HISTORY.md §2); without it
    (GMX_SEED_K2=0) or with another length the coverage is the same, and equal to the oracle's."""
    prg, reads = _snp_workload(200000, 2700, 8000, 15, multi=0.05)
    seeds = master_seeds(11, [8000])
    offs = flat_offsets(8000, 150)
    want = oracle_map(prg, 10, list(reads), seeds, threads=8)
    monkeypatch.setenv("GMX_SEED_K2", k2)
    ix = Index(prg, 10)
    assert ix.info.kmer_size2 == (0 if k2 == "0" else int(k2))
    qm = Quasimapper(ix)
    qm.map_reads(reads.reshape(-1), offs, seeds)
    assert canonical_cov(qm.coverage()) == want


def test_probe_pipeline_behind_a_longer_seed_table(monkeypatch):
    """With a longer seed table the engine runs gmx_seed_kernel + the seeded extend kernel; GMX_NO_SEEDED=1 keeps the
    probe / park / extend pipeline (the one indexes without such a table use) on the same index: same coverage,
    different queue lengths (the probe kernel also removes the tasks that die within its first steps)."""
    prg, reads = _snp_workload(200000, 2700, 8000, 16, multi=0.05)
    seeds = master_seeds(12, [8000])
    offs = flat_offsets(8000, 150)
    ix = Index(prg, 10)
    assert ix.info.kmer_size2 > 10
    seeded = Quasimapper(ix)
    seeded.map_reads(reads.reshape(-1), offs, seeds)
    monkeypatch.setenv("GMX_NO_SEEDED", "1")
    probed = Quasimapper(ix)
    probed.map_reads(reads.reshape(-1), offs, seeds)
    assert canonical_cov(probed.coverage()) == canonical_cov(seeded.coverage())
    assert canonical_cov(seeded.coverage()) == oracle_map(prg, 10, list(reads), seeds, threads=8)
