#!/bin/bash
# simple_tag 1 Mi envs f32 with u_noise = 0.3: the normals drawn inside the step kernel against torch's generator
# filling an [A, 2, B] buffer per step (the round-2 path); run on the GPU box from the repo root.
#   tools/noise_ab.sh <tag>  ->  gpurun_out/noise_ab_<tag>.txt + rocprofv3 kernel stats of both runs
set -e
TAG="$1"; ROOT="$PWD"; OUT="$ROOT/gpurun_out/noise_ab_$TAG.txt"
export TMPDIR=/tmp
: > "$OUT"
for mode in none device host; do
  case $mode in none) ARGS="";; device) ARGS="--u-noise 0.3";; host) ARGS="--u-noise 0.3 --host-noise";; esac
  for dt in simple_tag_1M_f32 simple_tag_1M_f64; do
    timeout -k 10 200 python3 bench.py --workload $dt --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs $ARGS \
      > "$ROOT/gpurun_out/noise_ab_$TAG.$mode.$dt.json" 2> "$ROOT/gpurun_out/noise_ab_$TAG.$mode.$dt.err"
    python3 - "$ROOT/gpurun_out/noise_ab_$TAG.$mode.$dt.json" "$mode" "$dt" >> "$OUT" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('%-8s %-20s ms_per_step %.4f  kernel_ms %.4f  value %.4e  launch: %s  noise: %s' % (
    sys.argv[2], sys.argv[3], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'], d['config']['launch'],
    d['config']['noise_source']))
PY
  done
done
cd /tmp
for mode in device host; do
  case $mode in device) ARGS="--u-noise 0.3";; host) ARGS="--u-noise 0.3 --host-noise";; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_noise_${mode}_$TAG" -o bench --output-format csv -- \
    python3 "$ROOT/bench.py" --workload simple_tag_1M_f32 --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs $ARGS \
    > "$ROOT/gpurun_out/prof_noise_${mode}_$TAG.log" 2>&1
  echo "== rocprofv3 kernel stats, $mode noise" >> "$OUT"
  head -6 "$ROOT/gpurun_out/prof_noise_${mode}_$TAG/bench_kernel_stats.csv" >> "$OUT"
done
cat "$OUT"
