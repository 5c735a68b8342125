cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/t60_tests.log
( TAG=t60_s0 PAIRS=50000000 SEQ_MODE=0 tools/decode_kstats.sh 2>&1 | head -12 ) > gpurun_out/t60_kstats.log 2>&1
rm -f gpurun_out/prof/t60_s0/*.db gpurun_out/prof/t60_s0/*trace.csv
