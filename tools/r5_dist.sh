#!/bin/bash
# the sharded shape on one GPU (bench.py --dist-selftest: one resident batch per contig, RCCL group of one) against the single batch of the same records
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5dist}; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],3), 'K1', round(d['roofline'].get('kernel_ms'),3), 'launches', d['stage_ms']['classify_launches_per_step'], 'fin', round(d['stage_ms']['finalize_kernels'],3), 'coll', d.get('collective_ms'), 'records', d['config']['records'])" 2>&1 | tail -1; }
for args in "" "--dist-selftest" "--pairs 10000000" "--pairs 10000000 --dist-selftest"; do
  tag=$(echo "x$args" | tr -c 'a-z0-9' '_')
  for lib in "" ${VARIANTS}; do
    l=""; [ -n "$lib" ] && l=$GRAFT_REPO_ROOT/gpurun_variants/$lib/lib/librnaseqc_amd.so
    RSQC_LIB=$l timeout 400 python bench.py --no-e2e --cpu-sample 0 --steps 10 --warmup 2 $args > $OUT/b_${lib:-tree}$tag.json 2> $OUT/b_${lib:-tree}$tag.err
    echo "${lib:-tree} [$args]: $(line $OUT/b_${lib:-tree}$tag.json)"
  done
done
