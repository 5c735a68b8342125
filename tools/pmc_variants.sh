#!/bin/bash
# SQ instruction counters (one rocprofv3 --pmc pass each) of classify_ei_kernel for every gpurun_variants/<name>/lib build listed in $VARIANTS
# -> gpurun_out/$TAG/insts.txt   (VALU / SALU / LDS wave-instructions per launch and per 64-record tile)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-pmcv}
mkdir -p $OUT
for name in $VARIANTS; do
  ( cd /tmp && RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/$name/lib/librnaseqc_amd.so timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p_$name -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-e2e --workers 1 > $OUT/p_$name.log 2>$OUT/p_$name.err )
done
python - <<PY
import csv, glob, collections, os
out = open("$OUT/insts.txt", "w")
for name in "$VARIANTS".split():
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/p_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "classify_ei" in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    tiles = 102499973 / 64.0
    line = "%-10s VALU %.4g (%.0f/tile)  SALU %.4g (%.0f/tile)  LDS %.4g (%.1f/tile)  wait/wave-cycles %.2f" % (name, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_VALU", 0) / tiles, m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_SALU", 0) / tiles, m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_LDS", 0) / tiles, m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1))
    print(line); out.write(line + "\n")
PY
