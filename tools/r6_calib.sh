#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/fetch_calib's known-byte kernels -> gpurun_out/$TAG/calib.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r6calib}; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o c -- $GRAFT_REPO_ROOT/tools/fetch_calib > $OUT/expected_$c.txt 2> $OUT/err_$c.txt
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
exp = {}
for line in open("$OUT/expected_FETCH_SIZE.txt"):
    p = line.split()
    if p and p[0] == "expected": exp[p[1]] = (float(p[2]), float(p[3]), p[4])
with open("$OUT/calib.txt", "w") as o:
    o.write("# tools/fetch_calib under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs); counters in KB as reported, per dispatch (1st, 2nd)\n")
    o.write("%-20s %14s %14s %10s | %14s %14s %10s | %s\n" % ("kernel", "read bytes", "FETCH_SIZE*1024", "ratio", "write bytes", "WRITE_SIZE*1024", "ratio", "pattern"))
    for k, (rb, wb, note) in exp.items():
        d = agg.get(k, {})
        f = [v * 1024 for v in d.get("FETCH_SIZE", [])]; w = [v * 1024 for v in d.get("WRITE_SIZE", [])]
        fm = sum(f) / len(f) if f else float("nan"); wm = sum(w) / len(w) if w else float("nan")
        o.write("%-20s %14.4g %14.4g %10.3f | %14.4g %14.4g %10.3f | %s  %s %s\n" % (k, rb, fm, fm / rb if rb else float("nan"), wb, wm, wm / wb if wb else float("nan"), note,
                ["%.4g" % x for x in f], ["%.4g" % x for x in w]))
print(open("$OUT/calib.txt").read())
PY
