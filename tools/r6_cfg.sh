#!/bin/bash
# one configuration of bench.py (CFG="--legacy" / "--fasta" / "--bed"): the GPU tests named by TESTK, the bench line on the tree's build and on
# every gpurun_variants/<name>, the kernel table of the tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r6af}; mkdir -p $OUT
N=$(echo $CFG | tr -d ' -')
if [ -n "$TESTK" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$TESTK" > $OUT/tests_$N.log 2>&1; tail -1 $OUT/tests_$N.log; fi
run() { RSQC_LIB=$2 timeout 300 python bench.py --no-e2e --cpu-sample 0 $CFG 2> $OUT/bench_${N}_$1.err | tail -1 > $OUT/bench_${N}_$1.json
  python -c "
import json; d=json.loads(open('$OUT/bench_${N}_$1.json').read()); print('$CFG $1', round(d['ms_per_step'],3), d['ms_per_step_spread']['median'], d['stage_ms'])"; }
run tree ""
for v in $(ls gpurun_variants 2>/dev/null); do [ -f gpurun_variants/$v/lib/librnaseqc_amd.so ] && run $v $GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so; done
BENCH_ARGS="$CFG" TAG=${TAG:-r6af}/k$N bash tools/kernel_stats.sh > $OUT/kstats_$N.txt 2>&1; head -${HEAD:-8} $OUT/kstats_$N.txt
