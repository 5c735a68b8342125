#!/bin/bash
# Where a pass spends what the kernels' events do not show: the tree, the build before the page-locked Read-Length mirror, a build that
# polls for the end of the pass, each three times (processes differ on one box), then the host clock trace of five passes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5q}
mkdir -p $OUT
B="--no-e2e --cpu-sample 0 --steps 20 --warmup 3"
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],3), d.get('ms_per_step_spread'), 'K1', round(d['stage_ms']['classify_k1'],3), 'fin', round(d['stage_ms']['finalize_kernels'],3))" 2>&1 | tail -1; }
run() { RSQC_LIB=$2 timeout 300 python bench.py $B > $OUT/bench_$1.json 2> $OUT/bench_$1.err; echo "$1: $(line $OUT/bench_$1.json)"; }
for k in 1 2 3; do
  run tree$k ""
  run pageable$k $GRAFT_REPO_ROOT/gpurun_variants/pageable/lib/librnaseqc_amd.so
  run spin$k $GRAFT_REPO_ROOT/gpurun_variants/spin/lib/librnaseqc_amd.so
done
RSQC_HOST_TRACE=1 RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/trace/lib/librnaseqc_amd.so timeout 300 python bench.py --no-e2e --cpu-sample 0 --steps 5 --warmup 2 > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace: $(line $OUT/bench_trace.json)"; grep "^\[host\]" $OUT/trace.err | tail -36
nproc; lscpu | grep -i "numa\|socket\|model name" | head -8
