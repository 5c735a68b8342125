#!/bin/bash
# Round 5, the inflate kernel's walk (INF_VWALK_CFG): GPU decode tests, then the tree against gpurun_variants/* on the realistic-entropy
# file and on the SURVEY 8(d) file, GPU inflate only (no CPU share), then the tree with the CPU share as the product runs it.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-r5r}; mkdir -p $OUT
T0=$(date +%s)
if [ "${TESTS:-1}" = "1" ]; then timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_zz_gpu_decode_levels.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "decode tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -3 $OUT/tests.log; fi
for mode in 1 0; do
  echo "== seq_mode $mode, GPU inflate only"
  PAIRS=${PAIRS:-10000000} SEQ_MODE=$mode REPS=${REPS:-3} CPU_THREADS=0 GEN_TIMEOUT=200 bash tools/decode_ab.sh 2>&1 | tee $OUT/ab_mode$mode.txt
  echo "== seq_mode $mode, the product's CPU share (tree only)"
  for rep in 1 2 3; do
    (cd /tmp && RSQC_DECODE=device RSQC_DECODE_PROFILE=1 timeout 60 $GRAFT_REPO_ROOT/rnaseqc_amd/bin/rnaseqc /tmp/ab.gtf /tmp/ab.bam /tmp/ab_out_share -vv > /tmp/ab.out 2> /tmp/ab.err
     echo "tree+share rep $rep: $(grep -o 'Average Reads/Sec: [0-9.e+]*' /tmp/ab.out)  $(grep -o 'inflate [0-9.]* ms ([0-9.]* GB/s out)' /tmp/ab.err) $(grep -o 'CPU share[^;]*' /tmp/ab.err)") | tee -a $OUT/ab_mode$mode.txt
  done
done
echo "total $(( $(date +%s) - T0 )) s"
