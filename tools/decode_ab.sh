#!/bin/bash
# A/B of whole builds in device-decode mode on one box: every directory gpurun_variants/<name>/{bin,lib} (a private copy of
# the CLI and its libraries -- an LD_PRELOADed second library does NOT switch the kernels) and the tree's own build ("tree")
# run the same synthetic BAM.  Prints the CLI's reads/s and the inflate kernel's GB/s (RSQC_DECODE_PROFILE).
# usage: PAIRS=5000000 SEQ_MODE=1 REPS=2 CPU_THREADS=0 tools/decode_ab.sh
export TMPDIR=/tmp
timeout ${GEN_TIMEOUT:-60} python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from rnaseqc_amd import bamio, synth, hostinfo
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, int("${PAIRS:-5000000}"), seed=2, workers=min(16, hostinfo.effective_cpus()))
bamio.write_gtf("/tmp/ab.gtf", ann)
bamio.write_bam_fast("/tmp/ab.bam", [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=int("${SEQ_MODE:-1}"))
print("records", batch.n, "seq_mode ${SEQ_MODE:-1}")
PY
cd /tmp
for v in $GRAFT_REPO_ROOT/gpurun_variants/*/ $GRAFT_REPO_ROOT/rnaseqc_amd/; do
  name=$(basename $v); [ "$v" = "$GRAFT_REPO_ROOT/rnaseqc_amd/" ] && name=tree
  for rep in $(seq ${REPS:-2}); do
    RSQC_DECODE=device RSQC_DECODE_PROFILE=1 RSQC_DECODE_CPU_THREADS=${CPU_THREADS:-0} timeout ${RUN_TIMEOUT:-30} $v/bin/rnaseqc /tmp/ab.gtf /tmp/ab.bam /tmp/ab_out_$name -vv > /tmp/ab.out 2> /tmp/ab.err
    echo "$name rep $rep rc $?: $(grep -o 'Average Reads/Sec: [0-9.e+]*' /tmp/ab.out)  $(grep -o 'inflate [0-9.]* ms ([0-9.]* GB/s out)' /tmp/ab.err)"
  done
done
for v in $GRAFT_REPO_ROOT/gpurun_variants/*/; do
  name=$(basename $v)
  cmp /tmp/ab_out_$name/ab.bam.metrics.tsv /tmp/ab_out_tree/ab.bam.metrics.tsv && cmp /tmp/ab_out_$name/ab.bam.gene_reads.gct /tmp/ab_out_tree/ab.bam.gene_reads.gct && echo "$name == tree: metrics and gene_reads identical"
done
