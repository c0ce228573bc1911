"""Kernel-pipeline A/B on one box: configs[1], reads resident in HBM (bytes) or packed in pinned host memory; prints ms per
step and the extend kernel's own time for every library given. Usage: python tools/kbench.py [lib.so ...] (default: the
regular build). Each library runs in a process of its own (GMX_LIB)."""
import json
import os
import subprocess
import sys

CHILD = r"""
import json, os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast, flat_offsets
G, S, K, N = 4411532, 60000, 10, 1000000
ref = random_ref(G, 1)
prg, pos, alts, n_alts = snp_prg(ref, S, 2)
ix = Index(prg, K)
reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, N, 150, 1000)
seeds = master_seeds(42, [N])
qm = Quasimapper(ix)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_offs = torch.from_numpy(flat_offsets(N, 150).astype(np.int64)).cuda()
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
def loop(steps):
    qm.reset(stream=0)
    for _ in range(steps):
        qm.map_reads_device(d_reads, d_offs, d_seeds, N, stream=0)
    qm.sync(); torch.cuda.synchronize()
loop(10)
qm.enable_timing(True); loop(5); tm = qm.timing(); qm.enable_timing(False)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); loop(30); best = min(best, (time.perf_counter() - t0) / 30)
st = qm.coverage().stats.as_dict()
print(json.dumps({"ms_per_step": best * 1e3, "extend_ms": tm["search_ms"] / tm["search_launches"], "inline_sites": int(ix.info.n_inline_sites),
                  "exact_mapped": st["exact_mapped"], "stats": st, "queues": qm.queue_counts()}))
"""
libs = sys.argv[1:] or [""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["GMX_LIB"] = os.path.abspath(lib)
    for rep in range(2):
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if out.returncode:
            print(lib or "default", "FAILED", out.stderr[-600:])
            break
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{lib or 'default':40s} {d['ms_per_step']:.4f} ms/step  extend {d['extend_ms']:.4f} ms  inline {d['inline_sites']}  {d['stats']}", flush=True)
