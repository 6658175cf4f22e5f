#!/bin/bash
# Runs bench.py for each workload given on the command line (no CPU baseline) and prints one summary
# line per workload; stops at the first failure (no further GPU step after a fault or a timeout).
set -u
mkdir -p gpurun_out
for w in "$@"; do
  timeout -k 10 150 python bench.py --workload "$w" --no-cpu-baseline --no-other-configs > "gpurun_out/bench_$w.log" 2>&1 || { echo "FAILED: $w"; tail -5 "gpurun_out/bench_$w.log"; exit 1; }
  tail -1 "gpurun_out/bench_$w.log" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%-32s %.3e agent-steps/s  %8.1f us  %6.0f GB/s  frac %.3f  %d B/env-step" % (d["config"]["workload"], d["value"], 1e3*r["avg_launch_ms"], r["achieved"], r["frac"], r["bytes_per_env_step"]))' || exit 1
done
