#!/bin/bash
# usage: tools/envs_sweep.sh <workload> <envs> [<envs> ...]   (default kernel choice; kernel time per launch and per env)
W="$1"; shift
for n in "$@"; do
  timeout -k 10 200 python -u bench.py --workload $W --envs $n --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-26s envs %8d  kernel %.2f us  %.0f GB/s  ns/env %.3f  ms/step %.4f' % ('$W', $n, 1e3*r['avg_launch_ms'], r['achieved'], 1e6*r['avg_launch_ms']/$n, d['ms_per_step']))" || exit 1
done
