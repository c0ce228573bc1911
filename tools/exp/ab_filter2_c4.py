"""configs[4] built once; the kernel-pipeline loop with the second filter pass on the main stream (default) and on side 2
(GMX_FILTER2_ON_SIDE, read at every launch), alternating. Usage: python tools/exp/ab_filter2_c4.py [4|4s]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, genome_recipe_file  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "4"
n, steps = 1_000_000, 8
G, S, seed = (3_100_000_000, 85_000_000, 61) if which == "4" else (400_000_000, 11_000_000, 51)
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gmx_ab.prg")
_, reads = genome_recipe_file(path, G, S, n, seed)
t0 = time.time()
ix = Index(path, 14)
os.remove(path)
print(f"index {ix.info.index_bytes / 1e9:.1f} GB in {time.time() - t0:.0f} s", flush=True)
seeds = master_seeds(42, [n])
offs = flat_offsets(n, reads.shape[1])
qm = Quasimapper(ix)
d_r = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda()
d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
d_s = torch.from_numpy(np.ascontiguousarray(seeds).view(np.int32).copy()).cuda()
stream = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for side in (False, True):
        if side:
            os.environ["GMX_FILTER2_ON_SIDE"] = "1"
        else:
            os.environ.pop("GMX_FILTER2_ON_SIDE", None)
        for warm in range(2):
            qm.reset(stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                qm.map_reads_device(d_r, d_o, d_s, n, stream=stream)
            qm.sync()
            dt = (time.perf_counter() - t0) / steps
        print(f"second filter pass on {'side 2' if side else 'main  '}: {dt * 1e3:.3f} ms per {n} reads = {n / dt / 1e6:.1f} M reads/s", flush=True)
