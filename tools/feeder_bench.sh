#!/bin/bash
# Feeder-only scaling (no GPU work): N BgzfFeeder instances of ONE process stream a synthetic realistic-entropy BAM side by side,
# under the CPUs the process may use.  usage: PAIRS=10000000 tools/feeder_bench.sh  -> prints one line per (feeders, read threads)
ROOT=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $ROOT
g++ -O2 -std=c++17 tools/feeder_bench.cpp rnaseqc_amd/csrc/host/bgzf_feed.cpp rnaseqc_amd/csrc/host/bam.cpp -o /tmp/feeder_bench -Lrnaseqc_amd/lib -lrnaseqc_amd -lz -ldl -lpthread -Wl,-rpath,$ROOT/rnaseqc_amd/lib || exit 1
python - <<PY
import sys
sys.path.insert(0, "$ROOT")
from rnaseqc_amd import bamio, synth, hostinfo
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, int("${PAIRS:-10000000}"), seed=2, workers=min(16, hostinfo.effective_cpus()))
bamio.write_bam_fast("/tmp/fb.bam", [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=int("${SEQ_MODE:-1}"))
print("records", batch.n, "CPUs the process may use:", hostinfo.effective_cpus())
PY
ls -la /tmp/fb.bam | awk '{print "file bytes", $5}'
for rt in ${READ_THREADS:-1 2 4}; do for n in ${FEEDERS:-1 2 4 8}; do /tmp/feeder_bench /tmp/fb.bam $n $rt ${CHUNK_MIB:-128}; done; done
