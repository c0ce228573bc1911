"""What eight feeders cost inside one GPU box's container (16 cores of CPU time, 300 GiB), measured without a second GPU
(VERDICT r3 item 8b): a group of N engines on device 0 — the code path `gram --devices` takes on a node, peer-copy exchange
instead of RCCL — fed with the packed feed (bit planes from page-locked memory), and `gram genotype --devices 0,0,...` on a
FASTQ file for the parser's CPU seconds. On a real node every engine has its own PCIe link: the aggregate H2D rate here is
ONE link's; what carries over is the host side — the threads, the page-locked staging, the parser — per million reads.
Usage: python tools/feed_x8.py [N_ENGINES=8] [READS_PER_ENGINE=1000000] [STEPS=6]"""
import os
import resource
import subprocess
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, QuasimapperGroup, Quasimapper, master_seeds, pack_reads, PinnedArray  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402
from gramtools_amd.build import build_gram  # noqa: E402
from oracle.prg_text import ints_to_prg_bytes  # noqa: E402  (tool: writes gram_dir/prg)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
G, n_sites, k = 4411532, 60000, 10
ref = random_ref(G, 1)
prg, pos, alts, n_alts = snp_prg(ref, n_sites, 2)
ix = Index(prg, k)
n = N * per
reads = simulate_snp_reads(ref, pos, alts, n_alts, min(n, 2_000_000), 150, 1000)
reads = np.concatenate([reads] * (-(-n // reads.shape[0])))[:n]
_sd = PinnedArray(n, np.uint32)  # page-locked like the planes (round 5: pageable seeds were registered and unregistered by every
_sd.array[:] = master_seeds(42, [n])  # call — 1 ms of host CPU per 1 M reads that round 4's figures included)
seeds = _sd.array
offs = flat_offsets(n, 150)
pk = pack_reads(np.ascontiguousarray(reads).reshape(-1), offs, uniform_len=150, pinned=True)


def cpu_seconds():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def thread_times():
    """CPU seconds per thread of this process (feeder threads come and go per call: summed by name is not possible, so the
    runtime's long-lived threads are listed and the rest — the feeders — is the difference to the process total)"""
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK")
        except OSError:
            pass
    return out


for n_eng in sorted({1, 2, N}):
    grp = QuasimapperGroup(ix, [0] * n_eng)
    m = n_eng * per
    sub = pk if m == n else pack_reads(np.ascontiguousarray(reads[:m]).reshape(-1), flat_offsets(m, 150), uniform_len=150, pinned=True)
    for rep in range(2):
        th0 = thread_times()
        c0, t0 = cpu_seconds(), time.perf_counter()
        for _ in range(steps):
            grp.map_reads_packed(sub, seeds[:m], use_skip=False)
        grp.allreduce()
        dt, cpu = (time.perf_counter() - t0) / steps, (cpu_seconds() - c0) / steps
        th1 = thread_times()
    lived = sorted(((th1[t] - th0[t]) / steps for t in th1 if t in th0 and t != os.getpid()), reverse=True)
    runtime_thread = lived[0] if lived else 0.0  # the HIP runtime's event thread: busy whenever work is in flight, ONE per process
    print(f"   of which: the runtime's busiest long-lived thread {runtime_thread * 1e3:.2f} ms per step (a cost per second of wall time and PROCESS, "
          f"not per read), the calling thread {(th1[os.getpid()] - th0[os.getpid()]) / steps * 1e3:.2f} ms, feeder threads + rest "
          f"{(cpu - runtime_thread - (th1[os.getpid()] - th0[os.getpid()]) / steps) * 1e3:.2f} ms per step = "
          f"{(cpu - runtime_thread) / m * 1e9:.2f} ns per read without the runtime thread", flush=True)
    print(f"{n_eng} engine(s) on device 0, {per} reads each per step: {dt * 1e3:.2f} ms per step = {m / dt / 1e6:.0f} M reads/s aggregate, "
          f"H2D {m * 40 / dt / 1e9:.1f} GB/s (40 B per read as planes), host CPU {cpu * 1e3:.1f} ms per step = {cpu / m * 1e9:.1f} ns per read", flush=True)
    grp.close()
    if sub is not pk:
        sub.close()

# the executable: parser + feed + N engines, CPU seconds of the whole process
tmp = os.environ.get("TMPDIR", "/tmp")
d = os.path.join(tmp, "feed_x8")
os.makedirs(d, exist_ok=True)
open(os.path.join(d, "prg"), "wb").write(ints_to_prg_bytes(prg))
from bench import write_fastq  # noqa: E402
fq = os.path.join(d, "reads.fq")
write_fastq(fq, [reads[:8_000_000]])
gram = build_gram()
subprocess.run([gram, "build", "--gram_dir", d, "--kmer_size", str(k)], stdout=subprocess.DEVNULL)
for devs in ("0", ",".join(["0"] * N)):
    for rep in range(2):
        t0 = time.perf_counter()
        c0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        out = subprocess.run([gram, "genotype", "--gram_dir", d, "--reads", fq, "--sample_id", "s", "--ploidy", "haploid", "--kmer_size", str(k),
                              "--genotype_dir", os.path.join(d, "run"), "--max_threads", "64", "--seed", "42", "--devices", devs],
                             stdout=subprocess.PIPE, text=True, env={"LD_LIBRARY_PATH": ""})
        c1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        wall = time.perf_counter() - t0
    line = [x for x in out.stdout.splitlines() if "quasimap" in x.lower() or "reads/s" in x.lower()]
    print(f"gram genotype --devices {devs} on {min(n, 8_000_000)} reads (plain FASTQ, 64 threads): wall {wall:.2f} s, CPU {(c1.ru_utime - c0.ru_utime) + (c1.ru_stime - c0.ru_stime):.2f} s "
          f"(rc {out.returncode}); {' | '.join(line[-2:])}", flush=True)
