export INGEST_MAP=0 GMX_INGEST_STATS=1 GMX_INGEST_VALU=0
for st in 256 1024 7168; do echo "== members per chunk $st"; python tools/ingest_bench.py 1000000 binned $st 2>&1 | tail -2 | cut -c1-330; done
echo "== all vector mode, 256"; GMX_INGEST_VALU=1 python tools/ingest_bench.py 1000000 binned 256 2>&1 | tail -1 | cut -c1-330
