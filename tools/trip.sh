cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/t21_tests.log
( MASKS="0" tools/k1_ablate.sh ) > gpurun_out/t21_k1.log 2>&1
( timeout 200 python tools/k1_prof.py --pairs 10000000 ) > gpurun_out/t21_prof.log 2>&1
