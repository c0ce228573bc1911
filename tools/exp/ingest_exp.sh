export GMX_INGEST_STATS=1 GMX_INGEST_NO_CRC=1 INGEST_MAP=0
for m in 0 1 2 4 7; do echo "== exp $m"; GMX_INGEST_EXP=$m python tools/ingest_bench.py 1000000 binned 7680 2>&1 | tail -2 | cut -c1-330; done
