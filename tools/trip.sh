cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( RSQC_DECODE_PROFILE=1 timeout 1200 python tools/decode_modes.py --pairs 50000000 --modes device --reps 2 ) > gpurun_out/t35_modes.log 2>&1
