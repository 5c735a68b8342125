#!/bin/bash
# Round 5: threads that read one chunk of the file side by side (RSQC_FEED_READ_THREADS), the CLI as the product runs it, both files.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5ac}; mkdir -p $OUT
for mode in ${MODES:-0 1}; do
timeout 300 python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from rnaseqc_amd import bamio, synth, hostinfo
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, int("${PAIRS:-25000000}"), seed=2, workers=min(16, hostinfo.effective_cpus()))
bamio.write_gtf("/tmp/ck.gtf", ann)
bamio.write_bam_fast("/tmp/ck.bam", [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=$mode)
print("records", batch.n, "seq_mode $mode")
PY
for rt in ${THREADS:-8 4 12 16}; do
  for rep in 1 2 3; do
    (cd /tmp && env RSQC_DECODE=device RSQC_DECODE_PROFILE=1 RSQC_FEED_READ_THREADS=$rt timeout 60 $GRAFT_REPO_ROOT/rnaseqc_amd/bin/rnaseqc /tmp/ck.gtf /tmp/ck.bam /tmp/ck_out -vv > /tmp/ck.out 2> /tmp/ck.err
     echo "mode $mode read threads $rt rep $rep: $(grep -o 'Average Reads/Sec: [0-9.e+]*' /tmp/ck.out) | $(grep -o 'inflate [0-9.]* ms ([0-9.]* GB/s out)' /tmp/ck.err | tail -1) | $(grep -o 'CPU share[^;]*' /tmp/ck.err) | $(grep -o '[0-9.]* ms waiting for file chunks[^;]*' /tmp/ck.err)")
  done
done 2>&1 | tee -a $OUT/feed_mode$mode.txt
done
