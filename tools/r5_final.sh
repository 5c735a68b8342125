#!/bin/bash
# the round's closing measurements: GPU suite, the driver-shaped default bench (kernel tier + whole node + cpu_baseline), PMC passes
# (SQ counters + FETCH_SIZE / WRITE_SIZE -> k1_traffic.json stamped with the K1 source hash), per-kernel tables (default, --bed, --dist-selftest)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5final}; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? ($(( $(date +%s) - T0 )) s)"; tail -c 1500 $OUT/bench_default.json | head -c 1500; echo
if [ "${PMC:-1}" = "1" ]; then TAG=${TAG:-r5final}/pmc PMC_TIMEOUT=300 bash tools/pmc.sh > $OUT/pmc.txt 2>&1; grep -A16 "classify_ei" $OUT/pmc.txt | head -24; echo "pmc done ($(( $(date +%s) - T0 )) s)"; fi   # (PMC=0: the K1 sources are those profiles/k1_traffic.json was measured on)
TAG=${TAG:-r5final}/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1; head -20 $OUT/kstats.txt
BENCH_ARGS="--bed" TAG=${TAG:-r5final}/kbed bash tools/kernel_stats.sh > $OUT/kstats_bed.txt 2>&1
BENCH_ARGS="--dist-selftest" TAG=${TAG:-r5final}/kdist bash tools/kernel_stats.sh > $OUT/kstats_dist.txt 2>&1
timeout 300 python bench.py --no-e2e --cpu-sample 0 --bed > $OUT/bench_bed.json 2> $OUT/bench_bed.err
echo "total $(( $(date +%s) - T0 )) s"
