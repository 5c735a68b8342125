#!/bin/bash
# The round's closing gpurun call: GPU tests, the PMC passes (traffic stamped with the K1 code hash, copied to where bench.py
# looks for it), the default bench line, the per-kernel table, the BED (configs[4]) line.  Everything lands in gpurun_out/$TAG.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r4final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
TAG=$TAG/pmc PMC_TIMEOUT=240 bash tools/pmc.sh > $OUT/pmc.txt 2>&1
cp gpurun_out/prof/$TAG/pmc/k1_traffic.json profiles/k1_traffic.json && cat profiles/k1_traffic.json
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 1500 $OUT/bench_default.json
TAG=$TAG/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1; head -24 $OUT/kstats.txt
timeout 300 python bench.py --bed --no-e2e --cpu-sample 0 > $OUT/bench_bed.json 2> $OUT/bench_bed.err; echo "bed rc $?"; tail -c 1200 $OUT/bench_bed.json
