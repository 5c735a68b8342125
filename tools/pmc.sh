#!/bin/bash
# PMC passes for the default bench command (each its own rocprofv3 run, counters only; the generator runs in-process:
# forked generator workers under counter collection hung in round 2).  Writes gpurun_out/prof/$TAG/summary.txt and k1_traffic.json.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-pmc}
mkdir -p $OUT
cd /tmp
i=0
if [ "${PMC_SQ_ONLY:-0}" = "1" ]; then SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"); fi
if [ -z "$SETS" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"); fi
for set in "${SETS[@]}"; do
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
# K1 = classify_ei_kernel + classify_multi_kernel since round 3 (the bench's timer brackets both): their traffic is summed
k1 = sorted(k for k in agg if "classify_ei" in k or "classify_multi" in k) or [k for k in agg if "classify_count" in k]
if k1 and all("FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k] for k in k1):
    fetch_kb = sum(sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]) for k in k1)
    write_kb = sum(sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"]) for k in k1)
    cfg = {}
    try:
        for line in open("$OUT/p1.log"):
            if line.startswith("{"): cfg = json.loads(line).get("config", {})
    except Exception: pass
    # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE uncalibrated, taken as is
    import sys
    sys.path.insert(0, "$GRAFT_REPO_ROOT")
    from rnaseqc_amd.hostinfo import k1_code_hash
    out = {"kernel": " + ".join(k1), "k1_code_hash": k1_code_hash(), "records": cfg.get("records"), "genes": cfg.get("genes"), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": fetch_kb * 1024 * 2.0 + write_kb * 1024,
           "note": "FETCH_SIZE x2 (gfx950 wide-load correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; separate --pmc passes"}
    json.dump(out, open("$OUT/k1_traffic.json", "w"), indent=1)
    print(out)
print(open("$OUT/summary.txt").read()[:3000])
PY
