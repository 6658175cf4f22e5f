#!/bin/bash
# simple_tag, 1 Mi envs, four learners in the loop: full replay records against compact ones (obs_tp1 stored once);
# wall time per step and the rocprofv3 kernel stats of both.  Run on the GPU box from the repo root: tools/train_demo_ab.sh <tag> [modes]
# (modes: default "full compact"; e.g. "compact store" for the step store, which has no append launch; a + joins the demo's
# arguments: "store store+f16actor")
set -e
TAG="$1"; MODES="${2:-full compact}"; ROOT="$PWD"; OUT="$ROOT/gpurun_out/train_demo_ab_$TAG.txt"
export TMPDIR=/tmp
: > "$OUT"
for mode in $MODES $MODES; do
  timeout -k 10 300 python3 tools/train_demo.py simple_tag 1048576 400 graph ${mode//+/ } 2>&1 | tail -1 >> "$OUT"
done
cd /tmp
for mode in $MODES; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_train_${mode}_$TAG" -o demo --output-format csv -- \
    python3 "$ROOT/tools/train_demo.py" simple_tag 1048576 400 graph ${mode//+/ } > "$ROOT/gpurun_out/prof_train_${mode}_$TAG.log" 2>&1
  echo "== rocprofv3 kernel stats, $mode records (name, calls, total ns, average ns)" >> "$OUT"
  python3 - "$ROOT/gpurun_out/prof_train_${mode}_$TAG/demo_kernel_stats.csv" >> "$OUT" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 8: break
    n = r['Name']
    n = n if len(n) < 100 else n[:60] + ' ... ' + n[-30:]
    print('%s, %s, %s, %.1f' % (n, r['Calls'], r['TotalDurationNs'], float(r['AverageNs'])))
PY
done
cat "$OUT"
