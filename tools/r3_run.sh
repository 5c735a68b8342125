#!/bin/bash
# One gpurun call of round 3: GPU parity tests, the bench (kernel tier only) on the tree's build and on every
# gpurun_variants/<name>/lib build, and the per-kernel table of the tree's build.  Everything lands in gpurun_out/$TAG.
# usage (on the GPU box): TAG=r3a TESTS=1 tools/r3_run.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r3}
mkdir -p $OUT
if [ "${TESTS:-1}" = "1" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${TEST_ARGS} > $OUT/tests.log 2>&1
  echo "tests rc $?"; tail -4 $OUT/tests.log
fi
B="--no-e2e --cpu-sample 0 --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS}"
timeout 300 python bench.py $B > $OUT/bench_tree.json 2> $OUT/bench_tree.err
echo "tree: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_tree.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])" 2>&1 | tail -1)"
for v in gpurun_variants/*/; do
  [ -d "$v" ] || continue
  name=$(basename $v)
  RSQC_LIB=$GRAFT_REPO_ROOT/$v/lib/librnaseqc_amd.so timeout 300 python bench.py $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])" 2>&1 | tail -1)"
done
if [ "${PMC:-0}" = "1" ]; then TAG=${TAG:-r3}/pmc PMC_TIMEOUT=200 bash tools/pmc.sh > $OUT/pmc.txt 2>&1; tail -60 $OUT/pmc.txt; fi
if [ "${KSTATS:-1}" = "1" ]; then
  TAG=${TAG:-r3}/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1
  head -24 $OUT/kstats.txt
fi
