#!/usr/bin/env python
"""bench.py — 150 bp reads quasimapped per second (BASELINE.json metric), whole job over N GPUs.

Workload (config.workload): BASELINE.json configs[1] — M. tuberculosis scale: 4 411 532 bp random
reference + 60 000 SNP sites written as a PRG, k = 10, 1 M x 150 bp error-free reads per GPU (50 % reverse
strand), synthetic (no real genomes offline). One "step" = one pass of the hot path (search + selection +
coverage atomics, forward and reverse complement) over the rank's 1 M reads, which are resident in HBM
before the timed region starts; for N > 1 every step ends with the RCCL all-reduce of the coverage arrays.

Extra objects on the JSON line:
  roofline     gmx_extend_kernel (the per-base extension of the mapping orientation, the kernel SURVEY.md §8(d)'s
               algorithmic-byte figure describes): nominal algorithmic bytes per launch / HIP-event duration measured
               inside the library on the launch stream. `frac` exceeds 1 by design: the figure prices one 128-byte
               rank block per base, the kernel compares 32 bases per 16-byte PRG record once a state has narrowed to
               one suffix-array position. `traffic` = HBM-side bytes per launch of that kernel from
               the committed rocprofv3 --pmc passes (profiles/round1/hbm_traffic.json), `design_*` = what the kernel
               itself has to move per read.
  cpu_baseline the oracle (CPU restatement of the reference algorithm, "port") on a bounded read sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME = 4411532
N_SITES = 60000
KMER = 10
READ_LEN = 150
READS_PER_GPU = 1_000_000
B_ALG_PER_READ = 128 * (READ_LEN - KMER) + READ_LEN   # 18 070 B/read at k = 10
PROBE_STEPS = 6                                        # bases done by gmx_probe_kernel, not by the dominant kernel
B_ALG_DOMINANT = 128 * (READ_LEN - KMER - PROBE_STEPS) + READ_LEN  # what gmx_extend_kernel itself is credited with
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8.0 TB/s spec
# bytes gmx_extend_kernel itself moves per mapped read: parked entry 20 + packed read 48 + PRG records
# ~6 x 16 + marker sub-records ~3 x 16 + final state 16 + path nodes 2 x 12 + queue/status words 16
B_DESIGN_PER_READ = 20 + 48 + 6 * 16 + 3 * 16 + 16 + 2 * 12 + 16
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "round1", "hbm_traffic.json")


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (FETCH_SIZE + WRITE_SIZE, separate --pmc passes; tools/pmc_hbm.sh)."""
    try:
        with open(TRAFFIC_FILE) as fh:
            table = json.load(fh)
        k = next(v for name, v in table.items() if name.startswith(kernel))  # template instances: gmx_extend_kernel<...>
        return int(k["fetch_bytes"] + k["write_bytes"])
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def cpu_baseline(prg, reads, seeds, max_seconds=20.0):
    """Oracle (port of the reference algorithm, OpenMP over reads) on a bounded sample of the same reads."""
    from oracle import Oracle
    from gramtools_amd.synth import flat_offsets
    cores = os.cpu_count() or 1
    o = Oracle(prg, KMER)
    n = 4000
    t0 = time.time()
    o.map_reads(reads[:n].reshape(-1), flat_offsets(n, READ_LEN), seeds[:n], threads=cores)
    rate = n / max(time.time() - t0, 1e-6)
    n2 = int(min(reads.shape[0], 400_000, max(n, rate * max_seconds * 0.6)))  # OpenMP scaling is sub-linear: keep it bounded
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n2].reshape(-1), flat_offsets(n2, READ_LEN), seeds[:n2], threads=cores)
    dt = time.time() - t0
    return {"value": n2 / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n2} of the rank-0 reads, same PRG/k/seeds, OpenMP over reads ({cores} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    from gramtools_amd.distributed import allreduce_device_coverage, fused_coverage_tensor
    cov_t = fused_coverage_tensor(qm) if world > 1 else None

    def step():
        # a step is a whole job: zeroed accumulators -> map the rank's reads -> one sum-exchange of the coverage
        qm.reset(stream=stream)
        qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
        if cov_t is not None:                             # THE exchange: one all-reduce of the fused coverage block
            allreduce_device_coverage(qm, dist, cov_t, stream)

    def fence():
        qm.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    qm.enable_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tm = qm.timing()
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    total_reads = n * world * args.steps
    value = total_reads / dt
    if rank == 0:
        st = qm.coverage().stats.as_dict()
        search_s = tm["search_ms"] / 1e3 / max(tm["search_launches"], 1)
        reads_per_launch = tm["reads"] / max(tm["search_launches"], 1)
        k_seed = max(KMER, int(ix.info.kmer_size2))     # the search is seeded after k2 >= k bases
        # with a longer seed table there is no probe phase (gmx_seed_kernel): the extend kernel does every step
        probe_steps = 0 if int(ix.info.kmer_size2) else PROBE_STEPS
        b_alg_dominant = 128 * (READ_LEN - k_seed - probe_steps) + READ_LEN
        achieved = b_alg_dominant * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0
        out = {
            "metric": "150bp reads quasimapped/sec (whole node); bit-exact coverage",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: M. tuberculosis scale, 4411532 bp random ref + 60000 SNP PRG, k=10, "
                                   f"{n} x 150 bp reads per GPU per step, fwd+rc, reads resident in HBM",
                       "reads_per_gpu": n, "read_len": READ_LEN, "kmer_size": KMER, "parallelism": f"reads sharded x{world}, "
                       "index replicated, one RCCL all-reduce of coverage per step",
                       "index_build_s": round(t_index, 2), "index_bytes": int(ix.info.index_bytes)},
            "roofline": {"bound": "hbm", "kernel": "gmx_extend_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic("gmx_extend_kernel"),
                         "traffic_source": "profiles/round1/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 1 M reads per launch)",
                         "alg_bytes_per_read": b_alg_dominant, "alg_bytes_per_read_whole_path": B_ALG_PER_READ, "reads_per_launch": reads_per_launch,
                         "avg_launch_ms": search_s * 1e3,
                         "other_kernels_ms_per_launch": tm["cover_ms"] / max(tm["cover_launches"], 1),
                         "design_bytes_per_read": B_DESIGN_PER_READ,
                         "design_achieved": B_DESIGN_PER_READ * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0,
                         "note": "frac > 1: the nominal figure prices a 128 B rank block per base; text-form states read 16 B per 32 bases"},
            "stats_last_step": st,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prg, reads, seeds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
