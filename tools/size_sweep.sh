#!/bin/bash
# usage: tools/size_sweep.sh <workload> <envs> [<envs> ...]
# one env per lane (MPE_FLAG_WIDE = 2) vs one agent per wavefront (MPE_FLAG_SPLIT = 4) vs the LDS-tile kernel (1)
W="$1"; shift
for n in "$@"; do
  for ft in 2 4 1; do
    MPE_FLAGS=$ft timeout -k 10 200 python -u bench.py --workload $W --envs $n --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-26s envs %8d flags=%-1s  kernel %.2f us  %.0f GB/s  ms/step %.4f' % ('$W', $n, '$ft', 1e3*r['avg_launch_ms'], r['achieved'], d['ms_per_step']))" || exit 1
  done
done
