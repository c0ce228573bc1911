"""Index build phases at a given scale (GMX_BUILD_TRACE=1). Usage: python tools/build_trace.py GENOME N_SITES K [THREADS ...]"""
import os
import sys
import time

os.environ.setdefault("GMX_BUILD_TRACE", "1")
sys.path.insert(0, ".")
from gramtools_amd import Index  # noqa: E402
from gramtools_amd.synth import random_ref, snp_prg  # noqa: E402

G, n_sites, k = (int(x) for x in sys.argv[1:4])
ref = random_ref(G, 1)
prg, pos, alts, n_alts = snp_prg(ref, n_sites, 2, multi_allelic_frac=0.05)
for threads in ([int(x) for x in sys.argv[4:]] or [0]):
    t0 = time.time()
    ix = Index(prg, k, threads=threads)
    print(f"index ({threads or 'all'} threads): {time.time() - t0:.1f} s, {ix.info.index_bytes / 1e9:.2f} GB, k2 = {ix.info.kmer_size2}", flush=True)
    ix.close()
