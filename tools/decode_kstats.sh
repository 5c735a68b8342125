#!/bin/bash
# kernel-trace stats of the CLI in device-decode mode on a synthetic BAM -> gpurun_out/prof/$TAG ; prints the per-kernel table
# usage: TAG=x PAIRS=10000000 SEQ_MODE=0 tools/decode_kstats.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-dk}
mkdir -p $OUT
python - <<PY
import os, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from rnaseqc_amd import bamio, synth, hostinfo
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, int("${PAIRS:-10000000}"), seed=2, workers=min(16, hostinfo.effective_cpus()))
bamio.write_gtf("/tmp/dk.gtf", ann)
bamio.write_bam_fast("/tmp/dk.bam", [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=int("${SEQ_MODE:-0}"))
print("records", batch.n)
PY
cd /tmp
RSQC_DECODE=device RSQC_DECODE_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- $GRAFT_REPO_ROOT/rnaseqc_amd/bin/rnaseqc /tmp/dk.gtf /tmp/dk.bam /tmp/dk_out -vv > $OUT/cli.out 2>$OUT/cli.err
grep -E "Average Reads|decode" $OUT/cli.out $OUT/cli.err
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/k_kernel_stats.csv")):
    print("%-60s calls %4s total %10.2f ms avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
