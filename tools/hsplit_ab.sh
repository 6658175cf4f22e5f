#!/bin/bash
# A/B of the two register-kernel forms of the command-hierarchy worlds (one env per lane: MPE_FLAGS=2; one agent per
# wavefront: MPE_FLAGS=4) over batch sizes; prints kernel us per launch (HIP events inside bench.py, HIP-graph replays).
# usage: tools/hsplit_ab.sh "<workloads>" "<envs list>" [lib.so ...]
set -u
mkdir -p gpurun_out
libs=("${@:3}"); [ ${#libs[@]} -eq 0 ] && libs=(libmpe_hip.so)
for w in $1; do for n in $2; do
  line="$w envs=$n"
  for l in "${libs[@]}"; do for f in 2 4; do
    MARL_AMD_LIB=$PWD/marl_amd/$l MPE_FLAGS=$f timeout -k 10 150 python bench.py --workload "$w" --envs "$n" --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs > gpurun_out/hsplit_ab.log 2>&1 || { echo "FAILED $w $n $l $f"; tail -5 gpurun_out/hsplit_ab.log; exit 1; }
    us=$(tail -1 gpurun_out/hsplit_ab.log | python -c 'import sys,json; print("%.2f" % (1e3*json.loads(sys.stdin.read())["roofline"]["avg_launch_ms"]))') || exit 1
    line="$line  ${l#libmpe_}:$f=$us"
  done; done
  echo "$line"
done; done
