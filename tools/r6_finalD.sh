#!/bin/bash
# closing call B: GPU suite, the driver-shaped default bench (kernel tier + whole node + cpu_baseline; reads profiles/k1_traffic.json and k1_model.json of call C),
# the per-kernel tables of --bed and --dist-selftest, and the other configurations' lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r6finalD; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? ($(( $(date +%s) - T0 )) s)"
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_note'), d['stage_ms'])
print('model current', (d['roofline'].get('model') or {}).get('current'), 'whole_node', {k: d['whole_node'][k] for k in ('value','wall_value','realistic_entropy_value','parity','realistic_entropy_parity','inflate_GBps')} if d.get('whole_node') else None)
print('cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])"
: > $OUT/other_configs.jsonl
for cfg in "--bed" "--chr1" "--legacy" "--fasta" "--dist-selftest"; do
  t=$(date +%s)
  timeout 600 python bench.py --no-e2e --cpu-sample 0 $cfg 2> $OUT/bench_other.err | tail -1 >> $OUT/other_configs.jsonl
  python -c "
import json; d=json.loads(open('$OUT/other_configs.jsonl').read().strip().splitlines()[-1])
print('$cfg', 'step', round(d['ms_per_step'],3), 'value %.3g' % d['value'], d['stage_ms'], 'collective', d.get('collective_ms'), d.get('collective_rccl_exposed_ms'), d.get('collective_host_merge_ms'))" 2>&1 | tail -1
  echo "   [$(( $(date +%s) - t )) s]"
done
BENCH_ARGS="--bed" TAG=r6finalD/kbed bash tools/kernel_stats.sh > $OUT/kstats_bed.txt 2>&1; head -14 $OUT/kstats_bed.txt
BENCH_ARGS="--dist-selftest" TAG=r6finalD/kdist bash tools/kernel_stats.sh > $OUT/kstats_dist.txt 2>&1; head -8 $OUT/kstats_dist.txt
echo "total $(( $(date +%s) - T0 )) s"
