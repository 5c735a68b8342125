#!/bin/bash
# PMC passes (counters only, each its own rocprofv3 run) of the CLI in device-decode mode on a synthetic BAM
# -> gpurun_out/prof/$TAG/summary.txt (mean per launch, per kernel).
# usage: TAG=x PAIRS=3000000 SEQ_MODE=1 SET1="A B" SET2="C D" [SET3="E F"] tools/decode_pmc.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-dpmc}
mkdir -p $OUT
timeout ${GEN_TIMEOUT:-40} python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from rnaseqc_amd import bamio, synth, hostinfo
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, int("${PAIRS:-3000000}"), seed=2, workers=min(16, hostinfo.effective_cpus()))
bamio.write_gtf("/tmp/dk.gtf", ann)
bamio.write_bam_fast("/tmp/dk.bam", [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=int("${SEQ_MODE:-1}"))
print("records", batch.n)
PY
cd /tmp
i=0
for set in "${SET1:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES}" "${SET2:-SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY}" ${SET3:+"$SET3"}; do
  i=$((i+1))
  RSQC_DECODE=device RSQC_DECODE_CPU_THREADS=0 timeout ${PMC_TIMEOUT:-25} rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- $GRAFT_REPO_ROOT/rnaseqc_amd/bin/rnaseqc /tmp/dk.gtf /tmp/dk.bam /tmp/dk_out > $OUT/p$i.log 2>$OUT/p$i.err
  echo "pass $i ($set): rc $?" >> $OUT/passes.txt
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r.get("Kernel_Name", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    o.write("# rocprofv3 --pmc (separate passes) -- rnaseqc gtf bam out, device decode, no CPU share; ${PAIRS:-3000000} pairs, seq_mode ${SEQ_MODE:-1}; mean per launch\n")
    for k, d in sorted(agg.items()):
        o.write(k + "\n")
        for c, v in sorted(d.items()): o.write("   %-24s n=%d mean=%.6g sum=%.6g\n" % (c, len(v), sum(v) / len(v), sum(v)))
print(open("$OUT/summary.txt").read()[:6000])
PY
cat $OUT/passes.txt
