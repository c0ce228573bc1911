#!/bin/bash
# Round 5: the round-4 wrong-result build of gmx_cover_jump, reproduced and explained (profiles/round5/jump_ptr_form_*).
# Applies tools/exp/jump_ptr_form.patch (made against the sources of commit 0e49958, when gmx_engine.hip was still one file: run it on a
# checkout of that commit — `git worktree add /tmp/jp 0e49958` —, later sources need the patch refreshed)
# — (the form that took the caller's first node by pointer, plus counters) to a scratch copy
# of the sources, builds P (pointer form) and G (the node fetched inside, same call sites) and runs the 37-loci workload and
# the flat fuzz slice with each. Usage (on a GPU box): tools/exp/jump_ptr_form.sh
set -e
cd "$(dirname "$0")/../.." && root=$PWD && mkdir -p gpurun_out && out=$root/gpurun_out/jump_ptr_form.txt && : > $out
work=$(mktemp -d) && cp -r gramtools_amd include $work/ && (cd $work && patch -p1 -s < $root/tools/exp/jump_ptr_form.patch)
src=$work/gramtools_amd/csrc
for v in P G; do
  flag=$([ $v = P ] && echo -DGMX_JUMP_PTR_FORM || echo -DGMX_JUMP_GENERAL)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flag -DGMX_JUMP_DBG -shared -o $work/libgmx_jump$v.so \
    $src/gmx_engine.hip $src/gmx_multi.hip $src/gmx_seedwalk.hip $src/gmx_suffixsort.hip $src/gmx_capi.cpp $src/gmx_index.cpp $src/gmx_infer.cpp $src/gmx_stock.cpp -lpthread -ldl -lz
  echo "== build $v ($flag)" >> $out
  GMX_LIB=$work/libgmx_jump$v.so timeout 600 python tools/exp/jump_dbg.py 2>&1 | tail -3 >> $out
  GMX_LIB=$work/libgmx_jump$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "dense_sites" 2>&1 | tail -3 >> $out
  GMX_NO_COVER_JUMP=1 GMX_LIB=$work/libgmx_jump$v.so timeout 900 python -m pytest tests/test_fuzz_slice.py -q -m gpu -k "fuzz_flat" 2>&1 | tail -3 >> $out
done
cat $out
