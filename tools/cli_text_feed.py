"""`gram genotype` on a PLAIN FASTQ (round 6): device text feed against the host parser, by chunk size and thread count.
Usage: python tools/cli_text_feed.py [N_READS] — configs[1] PRG, 150 bp reads, 316 B per record. Prints one line per variant:
the `Quasimap (parse + map)` time gram reports, reads/s, and the coverage files' equality with the host parser's."""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import write_fastq, GENOME, N_SITES, KMER  # noqa: E402
from gramtools_amd.build import build_gram  # noqa: E402
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
gram = build_gram()
ref = random_ref(GENOME, 1)
prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
per = 1_000_000
batches = [simulate_snp_reads_fast(ref, pos, alts, n_alts, min(per, n - i), 150, 1000 + i) for i in range(0, n, per)]
d = tempfile.mkdtemp(prefix="gmx_text_", dir=os.environ.get("TMPDIR", "/tmp"))
np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
fq = os.path.join(d, "reads.fastq")
write_fastq(fq, batches)
print(f"{fq}: {os.path.getsize(fq) / 1e6:.0f} MB, {n} reads", flush=True)
subprocess.run([gram, "build", "--gram_dir", d, "--kmer_size", str(KMER), "--max_threads", "16"], stdout=subprocess.DEVNULL, check=True)


def run(tag, env_extra, threads):
    env = dict(os.environ)
    env.update(env_extra)
    out = os.path.join(d, "run_" + tag.replace(" ", "_").replace("=", "_"))
    best = None
    for rep in range(3):
        t0 = time.time()
        g = subprocess.run([gram, "genotype", "--gram_dir", d, "--reads", fq, "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", str(KMER),
                            "--genotype_dir", out, "--max_threads", str(threads), "--seed", "42"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        wall = time.time() - t0
        if g.returncode:
            print(tag, "FAILED", g.stdout[-600:])
            return None
        t_map = next(float(l.rsplit(":", 1)[1]) for l in g.stdout.splitlines() if "Quasimap (parse + map" in l)
        if best is None or t_map < best[0]:
            best = (t_map, wall, g.stdout)
    h = hashlib.sha256()
    for f in ("allele_sum_coverage", "allele_base_coverage.json", "grouped_allele_counts_coverage.json"):
        h.update(open(os.path.join(out, "coverage", f), "rb").read())
    counters = [l for l in best[2].splitlines() if l.startswith("Count ")]
    print(f"{tag:42s} threads {threads:3d}: parse+map {best[0] * 1e3:7.2f} ms = {n / best[0] / 1e6:7.1f} M reads/s  (whole call {best[1]:.2f} s)  files {h.hexdigest()[:12]} {'|'.join(c.split(':')[-1].strip() for c in counters)}", flush=True)
    if env_extra.get("GMX_FEED_TRACE"):
        print("\n".join(l for l in best[2].splitlines() if l.startswith("[feed")))
    return h.hexdigest()


for rep in range(2):
    for threads in (1, 2, 4, 8, 16, 32, 64):
        ref_hash = run("host parser (GMX_HOST_FASTQ=1)", {"GMX_HOST_FASTQ": "1"}, threads)
        run("device text feed, 64 MB chunks", {"GMX_DEVICE_FASTQ": "1"}, threads)
        run("device text feed, 128 MB chunks", {"GMX_DEVICE_FASTQ": "1", "GMX_TEXT_CHUNK": str(128 << 20)}, threads)
