#!/bin/bash
# Debug build with the coverage routine's phase timers, one batch, print.  gpurun --timeout 600 -- 'bash tools/cover_stats.sh 0.002'
set -eu
cd gramtools_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGMX_LOOP_STATS -shared -o ../lib/libgmx.so gmx_engine.hip gmx_multi.hip gmx_capi.cpp gmx_index.cpp gmx_infer.cpp -lpthread -ldl -lz
cd ../..
mkdir -p gpurun_out
python tools/cover_stats.py ${1:-0.002} | tee gpurun_out/cover_stats_${1:-0.002}.txt
