#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=r6e TESTS=1 PMC=1 PMC_SQ_ONLY=0 KSTATS=1 VARIANTS="none" REPEAT_TREE=1 tools/r6_run.sh
timeout 300 python bench.py --no-e2e --cpu-sample 0 --bed --steps 10 --warmup 2 > gpurun_out/r6e/bench_bed.json 2> gpurun_out/r6e/bench_bed.err
python -c "import json; d=json.loads(open('gpurun_out/r6e/bench_bed.json').read().strip().splitlines()[-1]); print('bed: step', d['ms_per_step'], d['stage_ms'])"
TAG=r6e_traffic VARIANTS="abl4 abl8 abl1 abl17" tools/r6_traffic.sh > gpurun_out/r6e/traffic_stdout.txt 2>&1; tail -30 gpurun_out/r6e/traffic_stdout.txt
