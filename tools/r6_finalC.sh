#!/bin/bash
# closing call A: the counters of the final build -- SQ counters + FETCH_SIZE / WRITE_SIZE (k1_traffic.json stamped with the K1 source hash), the traffic of the
# ablation builds, the floor model (k1_model.json) -- written under gpurun_out/r6finalC, copied into profiles/ afterwards
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r6finalC; mkdir -p $OUT
T0=$(date +%s)
TAG=r6finalC/pmc PMC_TIMEOUT=300 bash tools/pmc.sh > $OUT/pmc.txt 2>&1; grep -A17 "classify_ei" $OUT/pmc.txt | head -22; echo "pmc done ($(( $(date +%s) - T0 )) s)"
TAG=r6finalC/traffic VARIANTS="abl4 abl8 abl1 abl17" tools/r6_traffic.sh > $OUT/traffic_stdout.txt 2>&1; grep "classify_ei" $OUT/traffic_stdout.txt
TAG=r6finalC/kstats bash tools/kernel_stats.sh > $OUT/kstats.txt 2>&1; head -24 $OUT/kstats.txt
KMS=$(python -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof/r6finalC/kstats/k_kernel_stats.csv')):
    if 'classify_ei_kernel' in r['Name']: print(float(r['AverageNs'])/1e6)")
python tools/k1_model.py gpurun_out/prof/r6finalC/pmc/summary.txt --traffic gpurun_out/r6finalC/traffic/traffic.txt --kernel-ms $KMS --out $OUT/k1_model.json | tail -12
cp gpurun_out/prof/r6finalC/pmc/k1_traffic.json $OUT/k1_traffic.json; cp gpurun_out/prof/r6finalC/pmc/summary.txt $OUT/pmc_summary.txt
echo "total $(( $(date +%s) - T0 )) s"
