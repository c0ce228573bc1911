"""Kernel-pipeline A/B of environment switches on one box: configs[1], reads resident in HBM; ms per step (best of 3 x 30 steps) and
the live per-kernel timers, one process per setting. Usage: python tools/kab.py "VAR=1 VAR2=x" "..."   ('' = default)."""
import json
import os
import subprocess
import sys

_src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench.py")).read()
CHILD = _src[_src.index('CHILD = r"""') + len('CHILD = r"""'):_src.index('"""\nlibs = ')]  # (kbench.py's child program, without running its main loop)

CHILD2 = CHILD.replace('print(json.dumps({"ms_per_step": best * 1e3,', 'print(json.dumps({"kernels": {k: (v["ms"] / v["launches"] if v["launches"] else None) for k, v in tm["kernels"].items()}, "ms_per_step": best * 1e3,')
settings = sys.argv[1:] or [""]
for rep in range(2):
    for st in settings:
        env = dict(os.environ)
        for kv in st.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD2], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if out.returncode:
            print(st or "default", "FAILED", out.stderr[-600:])
            continue
        d = json.loads(out.stdout.strip().splitlines()[-1])
        ks = " ".join(f"{k}={v * 1e3:.0f}" for k, v in d["kernels"].items() if v)
        print(f"{st or 'default':44s} {d['ms_per_step']:.4f} ms/step  extend {d['extend_ms'] * 1e3:.0f} us  [{ks}]  miss/noext {d['stats']['missing_kmer']}/{d['stats']['no_extension']}", flush=True)
