#!/bin/bash
# GPU-box profiling recipe (run through gpurun); writes under gpurun_out/
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
PAIRS=${PAIRS:-5000000}
TAG=${TAG:-run}
one() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 5 --warmup 2 --pairs $PAIRS --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', {k: round(v,4) for k,v in d['stage_ms'].items()}, 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))" >> gpurun_out/${TAG}_variants.log
}
if [ "${VARIANTS:-1}" = "1" ]; then
  one w4_g2048 RSQC_K1_VARIANT=4 RSQC_K1_GRID=2048
  one w6_g2048 RSQC_K1_VARIANT=6 RSQC_K1_GRID=2048
  one w8_g2048 RSQC_K1_VARIANT=8 RSQC_K1_GRID=2048
  one w4_g1024 RSQC_K1_VARIANT=4 RSQC_K1_GRID=1024
  one w8_g4096 RSQC_K1_VARIANT=8 RSQC_K1_GRID=4096
  one w4_nocov RSQC_K1_VARIANT=4 RSQC_DEBUG_MASK=1
  one w4_noscatter RSQC_K1_VARIANT=4 RSQC_DEBUG_MASK=3
  one w4_gateonly RSQC_K1_VARIANT=4 RSQC_DEBUG_MASK=8
fi
cat gpurun_out/${TAG}_variants.log
