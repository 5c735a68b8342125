#!/bin/bash
# One gpurun call of round 6: optional GPU tests, the kernel-tier bench on the tree's build and on every gpurun_variants/<name>/lib
# build (ONE change per line), optional SQ counters / kernel table.  Everything lands in gpurun_out/$TAG.
#   usage (on the GPU box): TAG=r5a TESTS=0 PMC=0 KSTATS=0 tools/r5_run.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r6}
mkdir -p $OUT
T0=$(date +%s)
if [ "${TESTS:-0}" = "1" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${TEST_ARGS} > $OUT/tests.log 2>&1
  echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -4 $OUT/tests.log
fi
B="--no-e2e --cpu-sample 0 --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS}"
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'K1', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'], d.get('stage_ms'))" 2>&1 | tail -1; }
run() { # name lib
  local t=$(date +%s)
  RSQC_LIB=$2 timeout 300 python bench.py $B > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  echo "$1: $(line $OUT/bench_$1.json)   [$(( $(date +%s) - t )) s]"
}
run tree ""
for v in ${VARIANTS:-$(ls gpurun_variants 2>/dev/null)}; do
  [ -f gpurun_variants/$v/lib/librnaseqc_amd.so ] || continue
  run $v $GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so
done
# ENV_RUNS="name:VAR=value,VAR2=value ..." : the tree's build under other environment settings (one bench run each)
for er in $ENV_RUNS; do
  name=${er%%:*}; envs=$(echo ${er#*:} | tr ',' ' ')
  t=$(date +%s)
  env $envs timeout 300 python bench.py $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name ($envs): $(line $OUT/bench_$name.json)   [$(( $(date +%s) - t )) s]"
done
if [ "${REPEAT_TREE:-1}" = "1" ]; then run tree2 ""; fi
if [ "${PMC:-0}" = "1" ]; then
  TAG=${TAG:-r6}/pmc PMC_SQ_ONLY=${PMC_SQ_ONLY:-1} PMC_TIMEOUT=200 bash tools/pmc.sh > $OUT/pmc.txt 2>&1; grep -A14 "classify_ei" $OUT/pmc.txt | head -40
  for name in $PMC_VARIANTS; do
    RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/$name/lib/librnaseqc_amd.so TAG=${TAG:-r6}/pmc_$name PMC_SQ_ONLY=1 PMC_TIMEOUT=200 bash tools/pmc.sh > $OUT/pmc_$name.txt 2>&1
    echo "== $name"; grep -A14 "classify_ei" $OUT/pmc_$name.txt | head -20
  done
fi
if [ "${KSTATS:-0}" = "1" ]; then
  TAG=${TAG:-r6}/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1
  head -24 $OUT/kstats.txt
fi
echo "total $(( $(date +%s) - T0 )) s"
