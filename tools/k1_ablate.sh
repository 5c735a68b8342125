#!/bin/bash
# K1 ablations (diagnostic only: results are wrong by construction when RSQC_DEBUG_MASK != 0).
# bits: 8 gate cascade only | 4 feature stage computed, nothing committed | 1 no coverage atomics | 2 no exon add, no gene hits
#       1024 no exon LDS add | 2048 no pair store | 4096 no gene LDS add
PAIRS=${PAIRS:-10000000}
run() { python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-finalize --pairs $PAIRS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s k1 %.3f ms  (%d records)' % ('$1', d['stage_ms']['classify_k1'], d['config']['records']))"; }
for m in ${MASKS:-0 8 4 1 7}; do RSQC_DEBUG_MASK=$m run "variant${RSQC_K1_VARIANT:-41} dbg=$m"; done
