#!/bin/bash
# instruction-count ablation of K1 under PMC
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-pmc2}
mkdir -p $OUT
cd /tmp
for m in 0 8 4 3 1; do
  RSQC_DEBUG_MASK=$m rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/m$m -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/m$m.log 2>&1
done
python - <<PY
import csv, glob, collections
for m in [0, 8, 4, 3, 1]:
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/m%d/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "classify_count" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mask", m, {c: "%.3g" % (sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
