#!/bin/bash
# kernel us per step (HIP events, HIP-graph replays) of every bench workload at small batches:
#   tools/small_batch.sh "<envs list>" <workload>...
set -u
mkdir -p gpurun_out
envs="$1"; shift
for w in "$@"; do
  line="$w"
  for n in $envs; do
    timeout -k 10 150 python bench.py --workload "$w" --envs "$n" --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs > gpurun_out/small_batch.log 2>&1 || { echo "FAILED $w $n"; tail -5 gpurun_out/small_batch.log; exit 1; }
    us=$(tail -1 gpurun_out/small_batch.log | python -c 'import sys,json; print("%.2f" % (1e3*json.loads(sys.stdin.read())["roofline"]["avg_launch_ms"]))') || exit 1
    line="$line  $n:$us"
  done
  echo "$line"
done
