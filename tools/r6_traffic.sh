#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of classify_ei_kernel on the tree's build and on the ablation builds under gpurun_variants/ (VERDICT r5 item 2b):
# the differences are the traffic of the coverage atomics (abl4), the pairs (abl8), the feature stages (abl1, abl17).  -> gpurun_out/$TAG/traffic.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r6traffic}; mkdir -p $OUT
cd /tmp
for v in tree ${VARIANTS:-abl4 abl8 abl1 abl17}; do
  lib=""; [ "$v" != "tree" ] && lib=$GRAFT_REPO_ROOT/gpurun_variants/$v/lib/librnaseqc_amd.so
  for c in FETCH_SIZE WRITE_SIZE; do
    RSQC_LIB=$lib timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$v/$c -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --workers 1 > $OUT/$v.$c.log 2> $OUT/$v.$c.err
  done
done
python - <<PY
import csv, glob, collections, os
out = open("$OUT/traffic.txt", "w")
out.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) of python bench.py --steps 1 --warmup 0 --no-e2e; KB as reported x 1024, per launch\n")
out.write("%-10s %-28s %14s %14s\n" % ("build", "kernel", "FETCH_SIZE_B", "WRITE_SIZE_B"))
for v in sorted(os.listdir("$OUT")):
    if not os.path.isdir(os.path.join("$OUT", v)): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        if "classify" in k or "frag_local" in k or "frag_count" in k:
            f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", [])
            out.write("%-10s %-28s %14.4g %14.4g\n" % (v, k.replace("void rsqc::", "").replace("rsqc::", "").replace(" ", "")[:28], (sum(f) / len(f) * 1024) if f else float("nan"), (sum(w) / len(w) * 1024) if w else float("nan")))
out.close()
print(open("$OUT/traffic.txt").read())
PY
