cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/t31_tests.log
( TAG=t31_s0 PAIRS=25000000 SEQ_MODE=0 tools/decode_kstats.sh 2>&1 | head -14 ) > gpurun_out/t31_kstats.log 2>&1
( RSQC_DECODE_PROFILE=1 timeout 900 python tools/decode_modes.py --pairs 25000000 --modes host,device --reps 2 ) > gpurun_out/t31_modes.log 2>&1
rm -f gpurun_out/prof/t31_s0/*.db gpurun_out/prof/t31_s0/*trace.csv
