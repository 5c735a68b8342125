#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=r6d TESTS=1 PMC=1 KSTATS=1 VARIANTS="fragkey16" ENV_RUNS="k3x0:RSQC_K3_XLARGE=0 grid4096:RSQC_K1_GRID=4096" tools/r6_run.sh
for v in tree bed4; do
  lib=""; [ "$v" != "tree" ] && lib=$GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so
  RSQC_LIB=$lib timeout 300 python bench.py --no-e2e --cpu-sample 0 --bed --steps 10 --warmup 2 > gpurun_out/r6d/bench_bed_$v.json 2> gpurun_out/r6d/bench_bed_$v.err
  python -c "import json; d=json.loads(open('gpurun_out/r6d/bench_bed_$v.json').read().strip().splitlines()[-1]); print('bed $v: step', d['ms_per_step'], d['stage_ms'])"
done
TAG=r6d_traffic VARIANTS="abl4 abl8 abl1 abl17" tools/r6_traffic.sh > gpurun_out/r6d/traffic_stdout.txt 2>&1; tail -30 gpurun_out/r6d/traffic_stdout.txt
