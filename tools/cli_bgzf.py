"""`gram genotype` on a BGZF FASTQ (Illumina-style headers, binned qualities, reads simulated from the configs[1] PRG): the
whole call with the file decoded on the GPU (default) and with GMX_HOST_GZ=1 (inflated by the host's threads), coverage
files compared byte for byte. Usage: python tools/cli_bgzf.py [N_READS]"""
import os, struct, subprocess, sys, tempfile, time, zlib
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, ".")
from gramtools_amd.build import build_gram
from gramtools_amd.synth import random_ref, simulate_snp_reads_fast, snp_prg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
L = 150


def piece(args):
    seed, first, reads = args
    rng = np.random.default_rng(seed)
    n = reads.shape[0]
    bases = np.frombuffer(b"NACGT", dtype=np.uint8)[reads]
    lvl = np.frombuffer(b"F:,#", dtype=np.uint8)
    q = lvl[np.repeat(rng.choice(4, size=(n, L // 5), p=[0.9, 0.06, 0.03, 0.01]), 5, axis=1)]
    xs, ys = rng.integers(1000, 30000, n), rng.integers(1000, 30000, n)
    text = b"".join(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n%s\n+\n%s\n" % (1101 + (first + i) // 40000, xs[i], ys[i], bases[i].tobytes(), q[i].tobytes())
                    for i in range(n))
    out = bytearray()
    for i in range(0, len(text), 65280):
        p = text[i:i + 65280]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(p) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p))
    return bytes(out), len(text)


if __name__ == "__main__":
    gram = build_gram()
    ref = random_ref(4411532, 1)
    prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
    reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, N, L, 1000)
    tmp = tempfile.mkdtemp(prefix="gmx_bgzf_")
    gram_dir = os.path.join(tmp, "gram")
    os.makedirs(gram_dir)
    prg.astype("<u4").tofile(os.path.join(gram_dir, "prg"))
    t0 = time.time()
    per = 50000
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        parts = list(ex.map(piece, [(7 + i, i * per, reads[i * per:(i + 1) * per]) for i in range((N + per - 1) // per)]))
    fq = os.path.join(tmp, "reads.fastq.gz")
    with open(fq, "wb") as fh:
        for p, _ in parts:
            fh.write(p)
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    print(f"{N} reads: {sum(t for _, t in parts) / 1e6:.0f} MB of text, {os.path.getsize(fq) / 1e6:.0f} MB of BGZF, written in {time.time() - t0:.0f} s", flush=True)
    subprocess.run([gram, "build", "--gram_dir", gram_dir, "--kmer_size", "10"], stdout=subprocess.DEVNULL)
    outs = {}
    legs = [("device", {}), ("host", {"GMX_HOST_GZ": "1"}), ("device", {}), ("host", {"GMX_HOST_GZ": "1"})]
    if os.environ.get("CLI_BGZF_DEVICES"):  # e.g. 0,0: two engines on one GPU, the file's chunks dealt over their ingests
        legs += [("dealt", {}), ("dealt", {})]
    for name, env in legs:
        out = os.path.join(tmp, "geno_" + name)
        cmd = [gram, "genotype", "--gram_dir", gram_dir, "--reads", fq, "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", "10", "--genotype_dir", out,
               "--max_threads", "16", "--seed", "42"] + (["--devices", os.environ["CLI_BGZF_DEVICES"]] if name == "dealt" else [])
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, GMX_PHASE_TRACE="1", **env))
        dt = time.time() - t0
        ph = [l for l in p.stdout.splitlines() if l.startswith("[phase")]
        t_map = [float(l.split()[1]) for l in ph if "quasimap done" in l or "base error rate" in l]
        counts = [l for l in p.stdout.splitlines() if l.startswith("Count ")]
        outs[name] = [open(os.path.join(out, "coverage", f), "rb").read() for f in ("allele_sum_coverage", "allele_base_coverage.json", "grouped_allele_counts_coverage.json")] + counts
        q = (t_map[1] - t_map[0]) / 1e3 if len(t_map) == 2 else float("nan")
        print(f"{name:7s} rc={p.returncode}  whole call {dt:.2f} s; reads decoded + mapped in {q:.3f} s = {N / q / 1e6:.1f} M reads/s  ({counts[-1] if counts else p.stdout[-300:]})", flush=True)
    print("coverage files and counters identical:", outs["device"] == outs["host"] and outs.get("dealt", outs["host"]) == outs["host"])
    if os.environ.get("CLI_BGZF_TRACE"):
        out = os.path.join(tmp, "geno_trace")
        cmd = [gram, "genotype", "--gram_dir", gram_dir, "--reads", fq, "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", "10", "--genotype_dir", out,
               "--max_threads", "16", "--seed", "42"] + (["--devices", os.environ["CLI_BGZF_DEVICES"]] if os.environ.get("CLI_BGZF_DEVICES") else [])
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, GMX_PHASE_TRACE="1", GMX_FEED_TRACE="1"))
        print("\n".join(l for l in p.stdout.splitlines() if l.startswith("[phase") or l.startswith("[feed")))
