#!/bin/bash
# K1 grid sweep on the tree's build (RSQC_K1_GRID: workgroups of the per-record kernel; default 4096).  usage: GRIDS="3840 5120" TAG=r5f tools/r5_grid.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5grid}; mkdir -p $OUT
B="--no-e2e --cpu-sample 0 --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS}"
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'K1', d['roofline'].get('kernel_ms'), 'frac', round(d['roofline']['frac'],4), 'fin', d['stage_ms']['finalize_kernels'])" 2>&1 | tail -1; }
for g in ${GRIDS:-4096}; do
  for v in "" ${VARIANTS}; do
    lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so
    RSQC_LIB=$lib RSQC_K1_GRID=$g timeout 300 python bench.py $B > $OUT/bench_${v:-tree}_g$g.json 2> $OUT/bench_${v:-tree}_g$g.err
    echo "${v:-tree} grid $g: $(line $OUT/bench_${v:-tree}_g$g.json)"
  done
done
