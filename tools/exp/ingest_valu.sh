export INGEST_MAP=0
for v in 0 1 2 3; do echo "== GMX_INGEST_VALU=$v"; GMX_INGEST_VALU=$v python tools/ingest_bench.py 4000000 binned 7168 2>&1 | tail -1 | cut -c1-200; done
GMX_INGEST_VALU=2 python tools/ingest_bench.py 4000000 wide 7168 2>&1 | tail -1
