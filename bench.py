#!/usr/bin/env python
"""bench.py — 150 bp reads quasimapped per second (BASELINE.json metric), whole job over N GPUs.

Workload (config.workload): BASELINE.json configs[1] — M. tuberculosis scale: 4 411 532 bp random reference +
60 000 SNP sites written as a PRG, k = 10, 1 M x 150 bp error-free reads per GPU (50 % reverse strand), synthetic
(no real genomes offline). One "step" = one pass of the hot path (search + selection + coverage atomics, forward and
reverse complement) over the rank's 1 M reads, which are resident in HBM (one byte per base, the reference's
encode_dna_bases form) before the timed region starts; for N > 1 every step ends with the exchange of the coverage
(one RCCL all-reduce of the fused block, driven from inside the library: gmx_comm_allreduce_coverage — the routine
`gram genotype --devices` uses).

`value` is a KERNEL-PIPELINE rate: no PCIe, no parsing, no file output. What a user of `gram genotype` sees is in the
extra keys (rank 0, N = 1 only):
  host_inclusive   SURVEY.md §8(d)'s timed region: host buffers in (H2D of the reads), coverage arrays final on the
                   host out (D2H + gather), through gmx_map_reads_host / gmx_coverage_fetch
  sustained        the `value` loop run for >= 1 s (clocks and thermals settle; thousands of steps)
  cli_end_to_end   the `gram` executable on a FASTQ file: parse + upload + map + exchange + the three coverage files
  cpu_baseline     the oracle (CPU restatement of the reference algorithm, "port"): all host threads and one thread
  roofline         gmx_extend_kernel, the dominant kernel: see DESIGN.md §8 for the byte model
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME = 4411532
N_SITES = 60000
KMER = 10
READ_LEN = 150
READS_PER_GPU = 1_000_000
B_NOMINAL_PER_READ = 128 * (READ_LEN - KMER) + READ_LEN   # 18 070 B/read at k = 10
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8.0 TB/s spec
# Algorithmic bytes gmx_extend_kernel must move per mapped read with text-form states (DESIGN.md §8 derives each term):
# queue entry 4 + seed directory entry 8 + packed read planes 48 + PRG text records 6 x 16 + marker sub-records 3 x 16 +
# path nodes 2 x 12 + coverage record 32 + task id 4
B_DESIGN_PER_READ = 4 + 8 + 48 + 6 * 16 + 3 * 16 + 2 * 12 + 32 + 4
PROFILE_DIR = os.path.join(ROOT, "profiles", "round2")


def profile_json(name):
    try:
        with open(os.path.join(PROFILE_DIR, name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (FETCH_SIZE + WRITE_SIZE, separate --pmc passes; tools/pmc_hbm.sh)."""
    table = profile_json("hbm_traffic.json")
    try:
        k = next(v for name, v in table.items() if name.startswith(kernel))  # template instances: gmx_extend_kernel<...>
        return int(k["fetch_bytes"] + k["write_bytes"])
    except (TypeError, KeyError, AttributeError, StopIteration):
        return None


def cpu_baseline(prg, reads, seeds, max_seconds=12.0):
    """Oracle (port of the reference algorithm, OpenMP over reads as quasimap.cpp:90) on a bounded sample of the same reads."""
    from oracle import Oracle
    from gramtools_amd.synth import flat_offsets
    cores = os.cpu_count() or 1
    o = Oracle(prg, KMER)
    n1 = 3000
    t0 = time.time()
    o.map_reads(reads[:n1].reshape(-1), flat_offsets(n1, READ_LEN), seeds[:n1], threads=1)
    one = n1 / max(time.time() - t0, 1e-6)
    n1 = int(min(reads.shape[0], max(n1, one * 5.0)))
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n1].reshape(-1), flat_offsets(n1, READ_LEN), seeds[:n1], threads=1)
    dt1 = time.time() - t0
    n = 4000
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n].reshape(-1), flat_offsets(n, READ_LEN), seeds[:n], threads=cores)
    rate = n / max(time.time() - t0, 1e-6)
    n2 = int(min(reads.shape[0], 400_000, max(n, rate * max_seconds * 0.6)))  # OpenMP scaling is sub-linear: keep it bounded
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n2].reshape(-1), flat_offsets(n2, READ_LEN), seeds[:n2], threads=cores)
    dt = time.time() - t0
    return {"value": n2 / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n2} of the rank-0 reads, same PRG/k/seeds, OpenMP over reads ({cores} threads), {dt:.1f} s",
            "single_thread": {"value": n1 / dt1, "unit": "reads/s", "cores": 1, "sample": f"first {n1} reads, {dt1:.1f} s"}}


def write_fastq(path, reads):
    """Four-line FASTQ of uint8 reads (1..4), fixed-width names, quality 'I': numpy only."""
    n, L = reads.shape
    name_w = 10
    row = np.empty((n, 1 + name_w + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    row[:, 0] = ord("@")
    idx = np.arange(n)
    for d in range(name_w):
        row[:, 1 + d] = (idx // 10 ** (name_w - 1 - d)) % 10 + ord("0")
    row[:, 1 + name_w] = ord("\n")
    row[:, 2 + name_w:2 + name_w + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[reads - 1]
    o = 2 + name_w + L
    row[:, o] = ord("\n")
    row[:, o + 1] = ord("+")
    row[:, o + 2] = ord("\n")
    row[:, o + 3:o + 3 + L] = ord("I")
    row[:, o + 3 + L] = ord("\n")
    row.tofile(path)


def cli_end_to_end(prg, reads, threads):
    """`gram build` + `gram genotype` on a FASTQ of these reads: what the Python front-end's subprocess call costs."""
    from gramtools_amd.build import build_gram
    gram = build_gram()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
        fq = os.path.join(d, "reads.fastq")
        write_fastq(fq, reads)
        t0 = time.time()
        b = subprocess.run([gram, "build", "--gram_dir", d, "--kmer_size", str(KMER), "--max_threads", str(threads)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_build = time.time() - t0
        t0 = time.time()
        g = subprocess.run([gram, "genotype", "--gram_dir", d, "--reads", fq, "--sample_id", "bench", "--ploidy", "haploid",
                            "--kmer_size", str(KMER), "--genotype_dir", os.path.join(d, "run"), "--max_threads", str(threads),
                            "--seed", "42"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_all = time.time() - t0
        if b.returncode or g.returncode:
            return {"error": (b.stdout + g.stdout)[-400:]}
        t_map = t_load = None
        for line in g.stdout.splitlines():
            if "Quasimap (parse + map" in line:
                t_map = float(line.rsplit(":", 1)[1])
            if "Load data" in line:
                t_load = float(line.rsplit(":", 1)[1])
        n = reads.shape[0]
        return {"reads": n, "fastq_bytes": os.path.getsize(fq), "host_threads": threads,
                "parse_and_map_s": t_map, "value": n / t_map if t_map else None, "unit": "reads/s",
                "whole_call_s": t_all, "whole_call_reads_per_s": n / t_all, "index_load_s": t_load, "gram_build_s": t_build,
                "note": "plain four-line FASTQ -> coverage files; parse_and_map = parser threads + H2D + kernels"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host_inclusive / sustained / cli_end_to_end")
    ap.add_argument("--torch-exchange", action="store_true", help="N > 1: all-reduce through torch.distributed instead of the library")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gramtools_amd import Index, Quasimapper, master_seeds
    from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads, flat_offsets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the quasimap engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node <gpus>"

    # ---- workload (index replicated; reads sharded by global read index) -----------------------
    t0 = time.time()
    ref = random_ref(GENOME, 1)
    prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
    ix = Index(prg, KMER)
    t_index = time.time() - t0
    n = args.reads
    reads = simulate_snp_reads(ref, pos, alts, n_alts, n, READ_LEN, 1000 + rank)
    all_seeds = master_seeds(42, [n * world])          # one master stream for the whole job
    seeds = all_seeds[rank * n:(rank + 1) * n]          # identical whatever the GPU count
    offs = flat_offsets(n, READ_LEN)
    qm = Quasimapper(ix, device=local_rank)
    d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    from gramtools_amd.distributed import allreduce_device_coverage, fused_coverage_tensor, CoverageComm
    exchange = "none"
    comm = cov_t = None
    if world > 1:
        ok = 0
        if not args.torch_exchange:
            try:
                comm = CoverageComm(qm, dist)
                ok = 1
            except Exception as exc:  # RCCL could not be driven from the library on this rank: every rank falls back
                print(f"[rank {rank}] library exchange unavailable ({exc}); using torch.distributed", file=sys.stderr)
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            exchange = "library (gmx_comm_allreduce_coverage: RCCL all-reduce from C++)"
        else:
            if comm is not None:
                comm.close()
                comm = None
            cov_t = fused_coverage_tensor(qm)
            exchange = "torch.distributed all_reduce on the aliased block"

    def exchange_coverage():
        if comm is not None:
            comm.allreduce(stream)
        elif cov_t is not None:
            allreduce_device_coverage(qm, dist, cov_t, stream)

    def job(steps, exchange_every_step=False):
        # the job of BASELINE.json: zeroed accumulators -> `steps` batches of the rank's reads (a step = one batch through
        # the whole kernel pipeline) -> ONE sum-exchange of the coverage at the end, as `gram genotype --devices` does it
        qm.reset(stream=stream)
        for _ in range(steps):
            qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
            if exchange_every_step:
                exchange_coverage()
                qm.reset(stream=stream)
        if not exchange_every_step:
            exchange_coverage()

    def fence():
        qm.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(steps, exchange_every_step=False):
        fence()
        t0 = time.perf_counter()
        job(steps, exchange_every_step)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # ---- side legs first (the device has idled through the index build: they also bring its clocks up) ----
    # roofline leg: the kernels bracketed by HIP events inside the library (a few extra steps, not part of the timed region)
    job(10)
    qm.enable_timing(True)
    fence()
    job(5)
    fence()
    tm = qm.timing()
    qm.enable_timing(False)
    dt_each = timed(args.steps, exchange_every_step=True)  # side figure: every step a job of its own (reset + exchange per step)

    # ---- W warm-up steps, then THE timed region: exactly `steps` steps, max over ranks ----
    if args.warmup:
        job(args.warmup)
    dt = timed(args.steps)
    st = qm.coverage().stats.as_dict() if rank == 0 else None

    total_reads = n * world * args.steps
    value = total_reads / dt
    out = None
    if rank == 0:
        search_s = tm["search_ms"] / 1e3 / max(tm["search_launches"], 1)
        reads_per_launch = tm["reads"] / max(tm["search_launches"], 1)
        achieved = B_DESIGN_PER_READ * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0
        k_seed = max(KMER, int(ix.info.kmer_size2))
        b_nominal_kernel = 128 * (READ_LEN - k_seed) + READ_LEN  # the part of the nominal figure this kernel is credited with
        sq = profile_json("sq_extend.json") or {}
        out = {
            "metric": "150bp reads quasimapped/sec (whole node); bit-exact coverage",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "value_is": "kernel-pipeline rate: reads resident in HBM, coverage left in HBM (see host_inclusive, cli_end_to_end)",
            "config": {"workload": "configs[1]: M. tuberculosis scale, 4411532 bp random ref + 60000 SNP PRG, k=10, "
                                   f"{n} x 150 bp reads per GPU per step, fwd+rc, reads resident in HBM",
                       "reads_per_gpu": n, "read_len": READ_LEN, "kmer_size": KMER, "parallelism": f"reads sharded x{world}, "
                       "index replicated, one RCCL all-reduce of coverage per step", "exchange": exchange,
                       "index_build_s": round(t_index, 2), "index_bytes": int(ix.info.index_bytes)},
            "roofline": {"bound": "hbm", "kernel": "gmx_extend_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic("gmx_extend_kernel"),
                         "traffic_source": "profiles/round2/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 1 M reads per launch)",
                         "alg_bytes_per_read": B_DESIGN_PER_READ,
                         "alg_bytes_model": "text-form states: 16 B of PRG per 32 bases + one 16 B sub-record per marker (DESIGN.md §8)",
                         "reads_per_launch": reads_per_launch, "avg_launch_ms": search_s * 1e3,
                         "other_kernels_ms_per_launch": tm["cover_ms"] / max(tm["cover_launches"], 1),
                         "what_bounds_it": "the rate of 64-byte transactions behind the XCD L2 (scattered 12-16 byte payloads), not HBM "
                                           "bandwidth and not instruction issue (DESIGN.md §4)",
                         "issue": {"valu_busy": sq.get("valu_busy"), "active_lane_share": sq.get("active_lane_share"),
                                   "frac": (sq.get("valu_busy") or 0) * (sq.get("active_lane_share") or 0) or None,
                                   "iterations_per_wave": sq.get("iterations_per_wave"),
                                   "heavy_steps_per_lane": sq.get("heavy_steps_per_lane"),
                                   "source": "profiles/round2/sq_extend.json (rocprofv3 --pmc SQ counters + GMX_LOOP_STATS build)"},
                         "nominal": {"bytes_per_read": b_nominal_kernel, "bytes_per_read_whole_path": B_NOMINAL_PER_READ,
                                     "achieved": b_nominal_kernel * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0,
                                     "note": "SURVEY §8(d) prices a 128 B rank block per base (the reference's algorithm); "
                                             "exceeds the HBM peak because the kernel does not move those bytes"}},
            "stats_job": st,
            "every_step_its_own_job": {"value": total_reads / dt_each, "unit": "reads/s",
                                       "note": "the same steps with zeroed accumulators before and a coverage exchange after EVERY step"},
        }
    if world == 1 and not args.no_extras:
        # ---- sustained: the same loop for >= 1 s --------------------------------------------------------------
        per_step = dt / args.steps
        k = max(int(1.25 / per_step), args.steps)
        dts = timed(k)
        out["sustained"] = {"seconds": dts, "steps": k, "reads": k * n, "value": k * n / dts, "unit": "reads/s"}
        # ---- host inclusive (SURVEY §8(d) timed region): host buffers -> coverage arrays on the host --------------
        flat = np.ascontiguousarray(reads.reshape(-1))
        qm.reset()
        qm.map_reads(flat, offs, seeds)              # warm-up (staging buffers, registration)
        reps = 4
        qm.reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            qm.map_reads(flat, offs, seeds)
        cov = qm.coverage()                           # D2H of the accumulator block + gather into the three arrays
        dth = time.perf_counter() - t0
        out["host_inclusive"] = {"value": reps * n / dth, "unit": "reads/s", "reads": reps * n, "seconds": dth,
                                 "includes": "H2D of 1 byte per base (pageable numpy -> staged), kernels, D2H of the coverage block",
                                 "exact_mapped": cov.stats.as_dict()["exact_mapped"]}
        # ---- the executable on a FASTQ file -----------------------------------------------------------------------
        big = reads if n >= 2_000_000 else np.concatenate([reads, simulate_snp_reads(ref, pos, alts, n_alts, n, READ_LEN, 77)])
        out["cli_end_to_end"] = cli_end_to_end(prg, big, min(os.cpu_count() or 8, 64))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prg, reads, seeds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
