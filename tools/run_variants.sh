#!/bin/bash
# usage: tools/run_variants.sh "<bench args>" lib1.so lib2.so ...   (sequential, one process each)
ARGS="$1"; shift
for lib in "$@"; do
  if [ "$lib" = base ]; then unset MARL_AMD_LIB; else export MARL_AMD_LIB="$PWD/$lib"; fi
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-32s value %.3e  kernel %.4f ms  %.0f GB/s  ms/step %.4f' % ('$lib', d['value'], r['avg_launch_ms'], r['achieved'], d['ms_per_step']))" || exit 1
done
