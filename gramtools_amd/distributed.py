"""Multi-GPU quasimap: reads shard across ranks, the index is replicated, one sum-reduction at the end.

The path partitions by read: every rank maps a contiguous range of the global read index
with the seeds that range would get in a single-process run (the master stream is global), accumulates
uint32 totals, and a single all-reduce(sum) per flat array (RCCL over xGMI on GPUs; gloo in the CPU tests)
yields the totals every rank finalises identically: allele-sum and grouped counts wrap mod 65536, per-base
saturates at 65535 — both functions of the total, so the result equals the single-thread reference.
"""
import numpy as np

from .quasimap import Coverage, QuasimapReadsStats, master_seeds


def shard_range(n_reads: int, world: int, rank: int):
    """Contiguous [lo, hi) of the global read index owned by `rank`."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_seeds(master_seed: int, reads_per_file):
    """The per-read selection seeds of the whole job (quasimap.cpp:120-141), independent of the GPU count."""
    return master_seeds(master_seed, reads_per_file)


def allreduce_raw(raw: dict, dist=None, device=None):
    """Sum-reduce the raw uint32 totals over all ranks. `raw` holds numpy arrays 'allele_sum', 'per_base',
    'grouped' (uint32), 'stats' (5 x uint64) and 'grouped_log' (uint32 words). Returns the reduced dict."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return raw
    import torch
    out = {}
    for key in ("allele_sum", "per_base", "grouped", "stats"):
        a = np.ascontiguousarray(raw[key])
        t = torch.from_numpy(a.astype(np.int64))  # gloo lacks uint32; int64 holds any uint32/uint64 total here
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        out[key] = t.cpu().numpy().astype(a.dtype)
    logs = [None] * dist.get_world_size()
    dist.all_gather_object(logs, np.ascontiguousarray(raw["grouped_log"]))
    out["grouped_log"] = np.concatenate([np.asarray(l, dtype=np.uint32) for l in logs]) if logs else raw["grouped_log"]
    return out


def coverage_from_raw(index, raw) -> Coverage:
    st = QuasimapReadsStats(*(int(x) for x in raw["stats"][:5]))
    return Coverage(index, np.asarray(raw["allele_sum"], dtype=np.uint32), np.asarray(raw["per_base"], dtype=np.uint32),
                    np.asarray(raw["grouped"], dtype=np.uint32), np.asarray(raw["grouped_log"], dtype=np.uint32), st)


def quasimap_reads_sharded(index, reads_flat, offsets, master_seed, map_shard, dist=None, device=None) -> Coverage:
    """Maps this rank's shard with `map_shard(reads_flat, offsets, seeds) -> raw dict` and reduces.

    On GPUs `map_shard` is :func:`gpu_map_shard`; the CPU tests pass a stand-in that produces the same raw
    arrays, so the sharding / seeding / reduction / finalisation logic is exercised under gloo."""
    n = len(offsets) - 1
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    seeds = global_seeds(master_seed, [n])
    lo, hi = shard_range(n, world, rank)
    offs = np.asarray(offsets, dtype=np.uint64)
    sub_offs = offs[lo:hi + 1] - offs[lo]
    sub_reads = np.asarray(reads_flat, dtype=np.uint8)[int(offs[lo]):int(offs[hi])]
    raw = map_shard(sub_reads, sub_offs, seeds[lo:hi])
    return coverage_from_raw(index, allreduce_raw(raw, dist, device))


def gpu_map_shard(index, device=0, rng_mode=0):
    """map_shard for :func:`quasimap_reads_sharded` backed by the HIP engine on `device`."""
    from .quasimap import Quasimapper

    def run(reads, offs, seeds):
        qm = Quasimapper(index, device=device, rng_mode=rng_mode)
        qm.map_reads(reads, offs, seeds)
        cov = qm.coverage()
        qm.close()
        s = cov.stats
        return dict(allele_sum=cov.raw_allele_sum, per_base=cov.raw_per_base, grouped=cov.raw_grouped,
                    grouped_log=cov.raw_grouped_log,
                    stats=np.array([s.all_reads_count, s.skipped_reads_count, s.missing_kmer_reads_count,
                                    s.no_extension_reads_count, s.exact_mapped_reads_count], dtype=np.uint64))
    return run


class _DevArray:
    """Zero-copy view of a device allocation for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def fused_coverage_tensor(qm):
    """One int32 tensor aliasing the engine's whole coverage block (allele_sum | per_base | grouped | counter limbs)."""
    import torch
    dc = qm.device_coverage()
    return torch.as_tensor(_DevArray(dc.fused, dc.n_fused, "<i4"), device="cuda")


def allreduce_device_coverage(qm, dist, tensor=None, stream=None):
    """The exchange through torch.distributed (RCCL): one all-reduce(sum), in place on the engine's coverage block
    (uint32 totals wrap exactly like int32 sums; the uint64 read counters travel as 16-bit limbs), and — when the PRG
    has sites with more than 8 alleles — an all-gather of the grouped logs (counted records: one per distinct
    (site, allele set), small), after which every rank's engine holds the sum of all logs.
    :class:`CoverageComm` does the same inside the library (the implementation `gram --devices` uses)."""
    t = fused_coverage_tensor(qm) if tensor is None else tensor
    qm.reduce_begin(stream)
    dist.all_reduce(t)
    qm.reduce_end(stream)
    if qm.index.uses_grouped_log and dist.get_world_size() > 1:
        logs = [None] * dist.get_world_size()
        dist.all_gather_object(logs, qm.grouped_log())
        for r, log in enumerate(logs):
            qm.import_grouped_log(np.asarray(log, dtype=np.uint32), replace=(r == 0))


class CoverageComm:
    """The library's own exchange (gmx.h: gmx_comm_*; RCCL called from C++, the code path `gram --devices` uses) for
    one engine per process. The 128-byte RCCL id is made on rank 0 and broadcast over the launcher's process group."""

    def __init__(self, qm, dist):
        import ctypes as C
        from . import _lib
        self.lib = _lib.load()
        self.qm = qm
        world, rank = dist.get_world_size(), dist.get_rank()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self.lib.gmx_comm_unique_id(ident))
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.h = C.c_void_p()
        _lib.check(self.lib.gmx_comm_create(ident, world, rank, qm.h, C.byref(self.h)))

    def allreduce(self, stream=None):
        import ctypes as C
        from . import _lib
        _lib.check(self.lib.gmx_comm_allreduce_coverage(self.h, C.c_void_p(stream) if stream else None))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_comm_destroy(self.h)
            self.h = None

    __del__ = close
