#!/bin/bash
# rocprofv3 evidence for one bench workload (run on the GPU box from the repo root):
#   tools/profile.sh <workload> <tag>        (BENCH_EXTRA="--envs 16384": further bench.py options, e.g. another batch size)
# 1. --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/bench_kernel_stats.csv
# 2. three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ issue counters + GRBM_GUI_ACTIVE; never
#    combined with a trace) on a short run
#    -> gpurun_out/pmc_<tag>/{fetch,write}_counter_collection.csv -> profiles/pmc_<workload>.json
# Copy the summaries you want judged into profiles/.
set -e
W="$1"; TAG="$2"; ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_$TAG" -o bench --output-format csv -- \
  python3 "$ROOT/bench.py" --workload "$W" --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs ${BENCH_EXTRA:-} > "$ROOT/gpurun_out/prof_$TAG.log" 2>&1
for c in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  timeout -s KILL 240 rocprofv3 --pmc ${c#*:} -d "$ROOT/gpurun_out/pmc_$TAG" -o "${c%%:*}" --output-format csv -- \
    python3 "$ROOT/bench.py" --workload "$W" --steps 75 --warmup 0 --no-cpu-baseline --no-other-configs --no-events ${BENCH_EXTRA:-} > "$ROOT/gpurun_out/pmc_$TAG.${c%%:*}.log" 2>&1
done
cd "$ROOT"
python3 tools/summarize_pmc.py "gpurun_out/pmc_$TAG" "gpurun_out/pmc_$W.json" "mpe::" > /dev/null
head -4 "gpurun_out/prof_$TAG/bench_kernel_stats.csv"
