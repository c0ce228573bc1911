#!/bin/bash
# bench several compile-time variants.  gpurun --timeout 1200 -- 'bash tools/sweep_build.sh "-DA=1" "-DB=2" ...'
mkdir -p gpurun_out
for flags in "$@"; do
  echo "== $flags"
  (cd gramtools_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -shared -o ../lib/libgmx.so gmx_engine.hip gmx_capi.cpp gmx_index.cpp -lpthread) || continue
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],3), 'ms; extend', round(d['roofline']['avg_launch_ms'],3), 'other', round(d['roofline']['other_kernels_ms_per_launch'],3), d['stats_job'])"
done
