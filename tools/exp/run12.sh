export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ingest.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
for q in binned wide; do INGEST_MAP=0 timeout 200 python tools/ingest_bench.py 4000000 $q 5120 2>&1 | tail -1; done
INGEST_MAP=0 timeout 200 python tools/ingest_bench.py 4000000 binned 10240 2>&1 | tail -1
GMX_INGEST_VALU=0 bash tools/exp/ingest_pmc.sh gpurun_out/ingest_pmc2 2>&1 | grep -E "SQ_INSTS_SALU|SQ_INSTS_VALU|SQ_INSTS_BRANCH|SQ_BUSY_CYCLES|SQ_INST_CYCLES_SALU"
timeout 900 python -m pytest tests/test_capi.py tests/test_configs.py tests/test_gram_cli.py tests/test_depth_limits.py -m gpu -x -q -k "not config4_full" 2>&1 | grep -E "passed|failed" | tail -2
