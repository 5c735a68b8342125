#!/bin/bash
# PMC passes for the default bench command (each its own rocprofv3 run, counters only; the generator runs in-process:
# forked generator workers under counter collection hung in round 2).  Writes gpurun_out/prof/$TAG/summary.txt and k1_traffic.json.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-pmc}
mkdir -p $OUT
cd /tmp
i=0
for set in ${SETS:-"FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"}; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-420} rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-e2e --workers 1 $BENCH_ARGS > $OUT/p$i.log 2>$OUT/p$i.err
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    o.write("# rocprofv3 --pmc (separate passes) -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-e2e --workers 1 $BENCH_ARGS ; mean per launch\n")
    for k, d in sorted(agg.items()):
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-24s n=%d mean=%.5g\n" % (c, len(v), sum(v) / len(v)))
k1 = [k for k in agg if "classify_count" in k]
if k1 and "FETCH_SIZE" in agg[k1[0]] and "WRITE_SIZE" in agg[k1[0]]:
    d = agg[k1[0]]
    fetch_kb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); write_kb = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    cfg = {}
    try:
        for line in open("$OUT/p1.log"):
            if line.startswith("{"): cfg = json.loads(line).get("config", {})
    except Exception: pass
    # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE uncalibrated, taken as is
    out = {"kernel": k1[0], "records": cfg.get("records"), "genes": cfg.get("genes"), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": fetch_kb * 1024 * 2.0 + write_kb * 1024,
           "note": "FETCH_SIZE x2 (gfx950 wide-load correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; separate --pmc passes"}
    json.dump(out, open("$OUT/k1_traffic.json", "w"), indent=1)
    print(out)
print(open("$OUT/summary.txt").read()[:3000])
PY
