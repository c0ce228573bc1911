#!/bin/bash
# Debug build, configs[2] recipe, one batch: overflow reasons and phase times of the coverage instances.
set -eu
cd gramtools_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGMX_LOOP_STATS -shared -o ../lib/libgmx.so gmx_engine.hip gmx_multi.hip gmx_capi.cpp gmx_index.cpp gmx_infer.cpp -lpthread -ldl -lz
cd ../..
mkdir -p gpurun_out
python tools/cover_why.py ${1:-1000000} 2>&1 | grep -v amdgpu | tee gpurun_out/cover_why.txt
