#!/bin/bash
# Round-4 evidence for the other configurations, one gpurun call: configs[1] (chr1), the N > 1 path on one rank, --legacy, --fasta,
# and the per-kernel table of the CLI from a BAM (device decode).  Everything lands in gpurun_out/r4x.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r4x; mkdir -p $OUT
B="--no-e2e --cpu-sample 0"
timeout 200 python bench.py $B --chr1 --pairs 5000000 > $OUT/bench_chr1.json 2> $OUT/bench_chr1.err; echo "chr1 rc $?"
timeout 300 python bench.py $B --dist-selftest --pairs 10000000 > $OUT/bench_dist.json 2> $OUT/bench_dist.err; echo "dist rc $?"
timeout 300 python bench.py $B --legacy --pairs 10000000 > $OUT/bench_legacy.json 2> $OUT/bench_legacy.err; echo "legacy rc $?"
timeout 300 python bench.py $B --fasta --pairs 10000000 > $OUT/bench_fasta.json 2> $OUT/bench_fasta.err; echo "fasta rc $?"
for f in chr1 dist legacy fasta; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), d['value'], d['roofline'].get('kernel_ms'), d.get('stage_ms'), d.get('collective_ms'))" 2>&1 | tail -1; done
