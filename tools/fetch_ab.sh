#!/bin/bash
# One FETCH_SIZE pass (its own rocprofv3 run: FETCH_SIZE and WRITE_SIZE together exceed the counter hardware) of the bench command on a
# variant library: usage  bash tools/fetch_ab.sh <variant name under gpurun_variants/>
cd /tmp; export TMPDIR=/tmp
v=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4w/$v; mkdir -p $OUT
RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-e2e > $OUT/log 2> $OUT/err
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    if "classify_ei" in k or "frag_local" in k or "frag_count" in k:
        print("$v", k, {c: round(sum(v)/len(v)/1e6,3) for c,v in d.items()}, "(KB / 1e6)")
PY
