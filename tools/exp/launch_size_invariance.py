"""Many asynchronous calls of the packed feeds with launches smaller than a call (GMX_FEED_CHUNK): the read counters must not depend
on the launch size (round 6: bench.py with GMX_FEED_CHUNK=125000 showed 1 118 of 20 M dead tasks in the other counter)."""
import os, sys, subprocess, json
CHILD = r'''
import sys, json, os, numpy as np
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads, pack_reads_2bit, PinnedArray
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast, flat_offsets
ref = random_ref(4411532, 1); prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
ix = Index(prg, 10)
N, NB = 1000000, 4
offs = flat_offsets(N, 150)
mode = os.environ.get("MODE", "2bit-async")
batches = []
for j in range(NB):
    reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, N, 150, 1000 + 97 * j)
    pk = (pack_reads if "planes" in mode else pack_reads_2bit)(reads.reshape(-1), offs, uniform_len=150, pinned=True)
    sd = PinnedArray(N, np.uint32); sd.array[:] = master_seeds(42 + j, [N])
    batches.append((pk, sd))
qm = Quasimapper(ix)
if "inplace" in mode: qm.seeds_in_place(True)
for rep in range(2):
    qm.reset()
    for s in range(12):
        pk, sd = batches[s % NB]
        qm.map_reads_packed(pk, sd.array, use_skip=False)
        if "sync" in mode and "async" not in mode: qm.sync()
    cov = qm.coverage()
print(json.dumps([cov.stats.as_dict(), int(cov.raw_allele_sum.sum()), int(cov.raw_per_base.sum())]))
'''
for mode in ("2bit-async", "2bit-sync", "planes-async", "2bit-async-inplace"):
    for chunk in ("", "125000"):
        env = dict(os.environ, MODE=mode)
        if chunk: env["GMX_FEED_CHUNK"] = chunk
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(mode, "chunk", chunk or "default", r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-500:], flush=True)
