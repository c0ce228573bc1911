"""The receiving end of the grouped-log exchange (gmx_multi.hip: sizes all-gather, payload all-gather padded to the largest
log, gmx_grouped_log_merge_gathered on every rank) as a pure function: ragged sizes, an empty rank, both record forms,
padding words — against a plain dictionary sum (CPU). And, on the GPU box, two ranks in two PROCESSES on device 0: RCCL
refuses two ranks on one device, so every rank must notice, say so and take the fallback exchange together."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gramtools_amd import _lib
from gramtools_amd.quasimap import iter_grouped_log, LOG_COUNTED, LOG_PAD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_log(rng, n_records, counted_frac=0.5, pad_frac=0.1):
    words, want = [], {}
    for _ in range(n_records):
        site = int(rng.integers(0, 6))
        ids = sorted(set(int(x) for x in rng.integers(0, 9, size=int(rng.integers(1, 5)))))
        if rng.random() < counted_frac:
            count = int(rng.integers(1, 1 << 34))
            words += [site, len(ids) | LOG_COUNTED, count & 0xFFFFFFFF, count >> 32] + ids
        else:
            count = 1
            words += [site, len(ids)] + ids
        want[(site, tuple(ids))] = want.get((site, tuple(ids)), 0) + count
        if rng.random() < pad_frac:
            words.append(LOG_PAD)
    return words, want


def test_merge_of_gathered_logs_ragged_sizes_and_an_empty_rank():
    lib = _lib.load()
    rng = np.random.default_rng(7)
    for world, sizes_of in ((1, [12]), (2, [0, 30]), (3, [17, 0, 5]), (4, [0, 0, 0, 0]), (5, [40, 1, 0, 9, 40])):
        logs, want = [], {}
        for n_rec in sizes_of:
            w, d = _random_log(rng, n_rec)
            logs.append(w)
            for k, v in d.items():
                want[k] = want.get(k, 0) + v
        pad = max((len(w) for w in logs), default=0)
        gathered = np.full(max(world * pad, 1), 0xDEADBEEF, dtype=np.uint32)   # the slices' tails are whatever was there
        for r, w in enumerate(logs):
            gathered[r * pad:r * pad + len(w)] = w
        sizes = np.asarray([len(w) for w in logs], dtype=np.uint64)
        n = lib.gmx_grouped_log_merge_gathered(gathered.ctypes.data, sizes.ctypes.data, world, pad, None, 0)
        assert n >= 0
        out = np.zeros(max(n, 1), dtype=np.uint32)
        assert lib.gmx_grouped_log_merge_gathered(gathered.ctypes.data, sizes.ctypes.data, world, pad, out.ctypes.data, n) == n
        got = {(s, ids): c for s, ids, c in iter_grouped_log(out[:n])}
        assert got == want
        assert len(got) == len(list(iter_grouped_log(out[:n])))   # one counted record per distinct (site, ids)


def test_merge_refuses_a_truncated_record():
    lib = _lib.load()
    bad = np.asarray([3, 2, 1], dtype=np.uint32)  # announces two ids, carries one
    sizes = np.asarray([3], dtype=np.uint64)
    assert lib.gmx_grouped_log_merge_gathered(bad.ctypes.data, sizes.ctypes.data, 1, 3, None, 0) < 0


CHILD = r"""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, GMX_DENSE_MAX_ALLELES="5")
dist.init_process_group("gloo", rank=rank, world_size=world)
from common import oracle_map, canonical_cov, flatten_reads
from gramtools_amd import Index, Quasimapper, master_seeds, GmxError
from gramtools_amd.distributed import CoverageComm, shard_range, allreduce_raw, coverage_from_raw
from gramtools_amd.synth import random_ref, mixed_variant_prg, simulate_haplotype_reads
ref = random_ref(5000, 3)
prg, sites = mixed_variant_prg(ref, 100, 4, max_alleles=7)
reads = simulate_haplotype_reads(ref, sites, 1500, 60, 150, 5)
seeds = master_seeds(42, [len(reads)])
ix = Index(prg, 7)
qm = Quasimapper(ix, device=0)          # BOTH ranks on device 0
lo, hi = shard_range(len(reads), world, rank)
flat, offs = flatten_reads(reads[lo:hi])
qm.map_reads(flat, offs, seeds[lo:hi])
how, err = "library", ""
try:
    comm = CoverageComm(qm, dist)
    ok = 1
except GmxError as e:                      # RCCL: two ranks on one device
    ok, err = 0, str(e)
flag = torch.tensor([ok]); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if int(flag.item()) == 1:
    comm.allreduce(None); qm.sync()
    cov = qm.coverage()
else:                                      # every rank falls back together: host totals through the launcher's group
    how = "fallback"
    c = qm.coverage(); s = c.stats
    raw = dict(allele_sum=c.raw_allele_sum, per_base=c.raw_per_base, grouped=c.raw_grouped, grouped_log=c.raw_grouped_log,
               stats=np.array([s.all_reads_count, s.skipped_reads_count, s.missing_kmer_reads_count, s.no_extension_reads_count,
                               s.exact_mapped_reads_count], dtype=np.uint64))
    cov = coverage_from_raw(ix, allreduce_raw(raw, dist))
want = oracle_map(prg, 7, reads, seeds, threads=4)
print(json.dumps({"rank": rank, "how": how, "equal": canonical_cov(cov) == want, "err": err[:200]}))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_processes_on_one_device_exchange_or_fall_back_together():
    port = str(29600 + os.getpid() % 200)
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(r), "2", port], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-1500:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))  # (RCCL prints its banner on stdout too)
    assert all(o["equal"] for o in outs), outs
    assert outs[0]["how"] == outs[1]["how"]            # both ranks took the same exchange
    if outs[0]["how"] == "fallback":
        assert any(o["err"] for o in outs)             # ... and the library said why it could not
