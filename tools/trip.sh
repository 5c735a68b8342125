cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/t26_tests.log
( for w in 4 1; do echo "== WPW $w"; RSQC_INFLATE_WPW=$w RSQC_DECODE_PROFILE=1 timeout 900 python tools/decode_modes.py --pairs 10000000 --modes device --reps 1; done ) > gpurun_out/t26_modes.log 2>&1
