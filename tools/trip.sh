cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python tools/decode_sweep.py 32:64 32:64 ) > gpurun_out/t16_sweep.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_contract.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/t16_tests.log
