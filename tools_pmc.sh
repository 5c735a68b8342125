#!/bin/bash
# PMC passes (each its own rocprofv3 run, counters only -- never combined with trace flags)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof/${TAG:-pmc}
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-24s n=%d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$OUT/summary.txt").read())
PY
