"""End-to-end `gram genotype` on a FASTQ file (the process boundary of the reference): wall-clock by stage.
Usage: python tools/cli_throughput.py N_READS [--gz]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from gramtools_amd.build import build_gram  # noqa: E402
from gramtools_amd.synth import random_ref, simulate_snp_reads, snp_prg  # noqa: E402

n = int(sys.argv[1])
gz = "--gz" in sys.argv
gram = build_gram()
ref = random_ref(4411532, 1)
prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
tmp = tempfile.mkdtemp(prefix="gmx_cli_")
gram_dir = os.path.join(tmp, "gram")
os.makedirs(gram_dir)
prg.astype("<u4").tofile(os.path.join(gram_dir, "prg"))
lut = np.frombuffer(b"NACGT", dtype=np.uint8)
seq = lut[reads]  # n x 150 ASCII
qual = b"I" * 150
fq = os.path.join(tmp, "reads.fastq" + (".gz" if gz else ""))
t0 = time.time()
with (gzip.open(fq, "wb", compresslevel=1) if gz else open(fq, "wb")) as fh:
    for i in range(0, n, 50000):
        block = seq[i:i + 50000]
        fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i + j, block[j].tobytes(), qual) for j in range(block.shape[0])))
print(f"wrote {fq}: {os.path.getsize(fq) / 1e6:.0f} MB in {time.time() - t0:.1f} s", flush=True)
out = os.path.join(tmp, "geno")
cmd = [gram, "genotype", "--gram_dir", gram_dir, "--reads", fq, "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", "10",
       "--genotype_dir", out, "--max_threads", "16", "--seed", "42"]
if "--cache" in sys.argv:
    t0 = time.time()
    b = subprocess.run([gram, "build", "--gram_dir", gram_dir, "--kmer_size", "10"], stdout=subprocess.PIPE, text=True)
    print(b.stdout.strip().splitlines()[-1], f"({time.time() - t0:.2f} s)", flush=True)
t0 = time.time()
p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
dt = time.time() - t0
print(p.stdout[-1500:])
print(f"gram genotype: rc={p.returncode}, {dt:.2f} s wall for {n} reads = {n / dt / 1e6:.2f} M reads/s end to end "
      f"(index build + FASTQ parse + H2D + kernels + output files)", flush=True)
