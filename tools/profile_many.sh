#!/bin/bash
# tools/profile_many.sh <workload>...: tools/profile.sh for each (kernel stats + the three PMC passes); stops at the first failure
set -e
for w in "$@"; do
  bash tools/profile.sh "$w" "${TAGPREFIX:-r3}_$w" > "gpurun_out/profile_$w.txt" 2>&1 || { echo "FAILED: $w"; tail -5 "gpurun_out/profile_$w.txt"; exit 1; }
  echo "done $w"
done
