#!/bin/bash
# Round 5: bytes of the file per rsqc_decode_submit call (RSQC_DECODE_CHUNK).  One wave inflates one BGZF block and a call of a
# 128 MB chunk of a realistic file holds about as many blocks (5 300) as the chip has wave slots for the kernel (5 120): the blocks
# behind the first 5 120 run on a nearly empty chip.  The CLI as the product runs it (CPU share on), three runs per size.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r5u}; mkdir -p $OUT
for mode in ${MODES:-1 0}; do
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
ls -la /tmp/ck.bam | awk '{print "bam bytes", $5}'
for mb in ${SIZES:-64 128 192 256 384 512 768}; do
 for mo in ${MAXOUT:-1024}; do
  for rep in 1 2 3; do
    (cd /tmp && t0=$(date +%s%N) && env RSQC_DECODE=device RSQC_DECODE_PROFILE=1 $( [ "$mb" != "default" ] && echo RSQC_DECODE_CHUNK=$((mb << 20)) RSQC_DECODE_MAX_OUT=$((mo << 20)) ) ${EXTRA_ENV} timeout 60 $GRAFT_REPO_ROOT/rnaseqc_amd/bin/rnaseqc /tmp/ck.gtf /tmp/ck.bam /tmp/ck_out_$mb -vv > /tmp/ck.out 2> /tmp/ck.err
     echo "mode $mode chunk $mb MB max_out $mo MB rep $rep rc $? wall $(( ($(date +%s%N) - t0) / 1000000 )) ms: $(grep -o 'Average Reads/Sec: [0-9.e+]*' /tmp/ck.out) | $(grep -o '[0-9]* calls, [0-9.]* MB in, [0-9.]* MB inflated' /tmp/ck.err | tail -1) | $(grep -o 'inflate [0-9.]* ms ([0-9.]* GB/s out)' /tmp/ck.err | tail -1) | $(grep -o 'CPU share[^;]*' /tmp/ck.err) | $(grep -o '[0-9.]* ms waiting for file chunks[^;]*' /tmp/ck.err)")
  done
 done
done 2>&1 | tee -a $OUT/chunk_mode$mode.txt
for mb in ${SIZES:-64 128 192 256 384 512 768}; do cmp /tmp/ck_out_$mb/ck.bam.metrics.tsv /tmp/ck_out_128/ck.bam.metrics.tsv > /dev/null && cmp /tmp/ck_out_$mb/ck.bam.gene_reads.gct /tmp/ck_out_128/ck.bam.gene_reads.gct > /dev/null && echo "chunk $mb == chunk 128: metrics and gene_reads identical"; done | tee -a $OUT/chunk_mode$mode.txt
done
