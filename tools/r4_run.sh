#!/bin/bash
# One gpurun call of round 4: optional GPU tests, the kernel-tier bench on the tree's build and on every gpurun_variants/<name>/lib
# build, SQ instruction counters (PMC) for the tree and for the variants listed in PMC_VARIANTS, the per-kernel table of the tree.
# Everything lands in gpurun_out/$TAG.   usage (on the GPU box): TAG=r4a TESTS=1 PMC=1 PMC_VARIANTS="abl1" tools/r4_run.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r4}
mkdir -p $OUT
if [ "${TESTS:-1}" = "1" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${TEST_ARGS} > $OUT/tests.log 2>&1
  echo "tests rc $?"; tail -4 $OUT/tests.log
fi
B="--no-e2e --cpu-sample 0 --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS}"
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline']['frac'], d.get('stage_ms'))" 2>&1 | tail -1; }
timeout 300 python bench.py $B > $OUT/bench_tree.json 2> $OUT/bench_tree.err
echo "tree: $(line $OUT/bench_tree.json)"
for v in gpurun_variants/*/; do
  [ -d "$v" ] || continue
  name=$(basename $v)
  RSQC_LIB=$GRAFT_REPO_ROOT/$v/lib/librnaseqc_amd.so timeout 300 python bench.py $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(line $OUT/bench_$name.json)"
done
if [ "${PMC:-0}" = "1" ]; then
  TAG=${TAG:-r4}/pmc PMC_SQ_ONLY=${PMC_SQ_ONLY:-1} PMC_TIMEOUT=200 bash tools/pmc.sh > $OUT/pmc.txt 2>&1; grep -A14 "classify_ei" $OUT/pmc.txt | head -40
  for name in $PMC_VARIANTS; do
    RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/$name/lib/librnaseqc_amd.so TAG=${TAG:-r4}/pmc_$name PMC_SQ_ONLY=1 PMC_TIMEOUT=200 bash tools/pmc.sh > $OUT/pmc_$name.txt 2>&1
    echo "== $name"; grep -A14 "classify_ei" $OUT/pmc_$name.txt | head -20
  done
fi
if [ "${KSTATS:-1}" = "1" ]; then
  TAG=${TAG:-r4}/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1
  head -24 $OUT/kstats.txt
fi
