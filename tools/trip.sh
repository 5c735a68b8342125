cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/t54_tests.log
( RSQC_DECODE_PROFILE=1 timeout 900 python tools/decode_modes.py --pairs 50000000 --modes device --reps 2 ) > gpurun_out/t54_modes.log 2>&1
