#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=r6f TESTS=1 PMC=0 KSTATS=1 VARIANTS="nt1 nt2 nt3" REPEAT_TREE=1 tools/r6_run.sh
timeout 300 python bench.py --no-e2e --cpu-sample 0 --bed --steps 10 --warmup 2 > gpurun_out/r6f/bench_bed.json 2> gpurun_out/r6f/bench_bed.err
python -c "import json; d=json.loads(open('gpurun_out/r6f/bench_bed.json').read().strip().splitlines()[-1]); print('bed: step', d['ms_per_step'], d['stage_ms'])"
BENCH_ARGS="--bed" TAG=r6f/kbed bash tools/kernel_stats.sh > gpurun_out/r6f/kstats_bed.txt 2>&1; head -12 gpurun_out/r6f/kstats_bed.txt
TAG=r6f_traffic VARIANTS="nt3" tools/r6_traffic.sh > gpurun_out/r6f/traffic_stdout.txt 2>&1; grep "classify_ei" gpurun_out/r6f/traffic_stdout.txt
