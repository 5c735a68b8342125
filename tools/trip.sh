cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/t14_tests.log
( timeout 200 python tools/k1_prof.py --pairs 10000000 ) > gpurun_out/t14_prof.log 2>&1
( MASKS="0 1" tools/k1_ablate.sh ) > gpurun_out/t14_ablate.log 2>&1
