"""`gram genotype` on a FASTQ of N reads of the bench workload: the timer report with the feed breakdown.
Usage: python tools/cli_feed.py [N_READS=4000000] [THREADS=64]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from bench import write_fastq, GENOME, N_SITES, KMER  # noqa: E402
from gramtools_amd.build import build_gram  # noqa: E402
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
threads = sys.argv[2] if len(sys.argv) > 2 else "64"
ref = random_ref(GENOME, 1)
prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
gram = build_gram()
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
    write_fastq(os.path.join(d, "r.fq"), [reads])
    subprocess.run([gram, "build", "--gram_dir", d, "--kmer_size", str(KMER), "--max_threads", threads], stdout=subprocess.DEVNULL)
    for rep in range(2):
        t0 = time.time()
        g = subprocess.run([gram, "genotype", "--gram_dir", d, "--reads", os.path.join(d, "r.fq"), "--sample_id", "s", "--ploidy", "haploid",
                            "--kmer_size", str(KMER), "--genotype_dir", os.path.join(d, f"run{rep}"), "--max_threads", threads, "--seed", "42"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(f"--- run {rep}: {time.time() - t0:.2f} s wall, rc {g.returncode}")
        print("\n".join(l for l in g.stdout.splitlines() if "Timer" in l or l.startswith("  ") or "feed" in l or "Count exact" in l))
