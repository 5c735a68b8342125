#!/bin/bash
# K1 ablations (diagnostic only: results are wrong by construction when RSQC_DEBUG_MASK != 0)
run() { python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-finalize | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s k1 %.3f ms' % ('$1', d['stage_ms']['classify_k1']))"; }
for m in ${MASKS:-0 8 1 2 3 4 7}; do RSQC_DEBUG_MASK=$m run "variant${RSQC_K1_VARIANT:-3} dbg=$m"; done
