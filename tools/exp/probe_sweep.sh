# probe budget sweep on the 400 Mb / k = 12 proxy of configs[4]:  gpurun -- 'bash tools/exp/probe_sweep.sh "GMX_PROBE_ITERS=10" "GMX_PROBE_ITERS=1"'
for v in "$@"; do
  env $v bash tools/profile_round4.sh 4k12 trace > /dev/null 2>&1
  echo "== $v: $(grep -E 'kernel pipeline' gpurun_out/r4/config4k12/run_trace.txt)"
  python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/config4k12/kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if n.startswith(('gmx_','void gmx_')) and float(r['AverageNs'])>200000 and 'mark' not in n and 'sa_ctx' not in n and int(r['Calls'])>=12:
        print(f"  {n[:60]:60s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
