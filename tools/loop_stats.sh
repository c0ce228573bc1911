#!/bin/bash
# Debug build with loop statistics, one batch, print.  gpurun --timeout 600 -- 'bash tools/loop_stats.sh'
set -eu
cd gramtools_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGMX_LOOP_STATS -shared -o ../lib/libgmx.so gmx_engine.hip gmx_multi.hip gmx_capi.cpp gmx_index.cpp gmx_infer.cpp -lpthread -ldl -lz
cd ../..
mkdir -p gpurun_out
python tools/loop_stats.py | tee gpurun_out/loop_stats.txt
