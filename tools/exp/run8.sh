export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ingest.py -x -q 2>&1 | tail -2
INGEST_MAP=0 timeout 200 python tools/ingest_bench.py 4000000 binned 7168 2>&1 | tail -2
INGEST_MAP=0 timeout 200 python tools/ingest_bench.py 4000000 binned 3584 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_r5.log 2> gpurun_out/bench_r5.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench_r5.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/bench_r5.log").read().strip().splitlines()[-1])
    print("value", d["value"], "kernel_pipeline", d.get("kernel_pipeline",{}).get("value"), "bgzf", json.dumps(d.get("bgzf_device_feed"))[:600])
except Exception as e: print("no json", e)
PY
for v in "GMX_DUMMY=1" "GMX_FILTER1_GLOBAL=1" "GMX_FILTER1_LAST=1" "GMX_FILTER1_GLOBAL=1 GMX_FILTER1_LAST=1"; do echo "== $v"; env $v python tools/kbench.py 2>&1 | cut -c1-75; done
