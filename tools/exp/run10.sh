export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_ingest.py tests/test_gram_cli.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/exp/engines_in_flight.py 3 1 4000000 2>&1 | tail -1
timeout 600 python tools/exp/engines_in_flight.py 1 1 4000000 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline --configs 2 > gpurun_out/bench_c2.log 2> gpurun_out/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_c2.log").read().strip().splitlines()[-1])
c=d["configs"]["2"]
print({k:c[k] for k in ("kernel_pipeline","packed_host_feed","kernel_pipeline_large_batch","all_reads_mapped")})
PY
