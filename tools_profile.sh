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
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  one $name $(echo $envs | tr ',' ' ')
done
cat gpurun_out/${TAG}_variants.log
