for L in 32 16 8 4; do
  export GMX_COVER_LANES=$L
  echo "== lanes $L"
  bash tools/repeat_timeline.sh 0.002 gpurun_out/r3a 2>&1 | grep "cover_kernel\|device-resident"
  bash tools/repeat_timeline.sh 0.05 gpurun_out/r3a 2>&1 | grep "cover_kernel\|device-resident"
done
