#!/bin/bash
# Runs tests/test_two_process_gpu.py N times in a row on one box and reports how many runs were green, with the failing runs' output kept:
#   tools/soak_two_process.sh 50 gpurun_out/r05_soak
# (the harness stalled twice in 25 suite runs in round 4; the test no longer retries, every collective carries a
# timeout and each rank logs its progress, so a repeat shows where the pair stood.)
n=${1:-50}
out=${2:-gpurun_out/soak}
mkdir -p $out
ok=0
bad=0
t0=$(date +%s)
for i in $(seq 1 $n); do
  if timeout 600 python -m pytest tests/test_two_process_gpu.py -x -q > $out/run_$i.log 2>&1; then ok=$((ok+1)); rm -f $out/run_$i.log; else bad=$((bad+1)); fi
done
echo "{\"runs\": $n, \"green\": $ok, \"failed\": $bad, \"seconds\": $(( $(date +%s) - t0 ))}" | tee $out/summary.json
