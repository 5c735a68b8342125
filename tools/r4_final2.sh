#!/bin/bash
# Closing call after the decode guess fix: GPU tests + the default bench line (K1 sources unchanged: profiles/k1_traffic.json stays valid)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r4final2; mkdir -p $OUT
timeout 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" $OUT/tests.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['stage_ms']); print(d['whole_node'])"
