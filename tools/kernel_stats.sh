#!/bin/bash
# kernel-trace stats of the default bench command -> gpurun_out/prof/$TAG ; prints the per-kernel table
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-kt}
mkdir -p $OUT; cd /tmp
timeout ${KSTATS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --workers 1 $BENCH_ARGS > $OUT/bench.json 2>$OUT/bench.err
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/k_kernel_stats.csv")):
    print("%-60s calls %3s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
