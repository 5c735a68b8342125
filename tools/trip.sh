cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 ) > gpurun_out/t4_tests.log
( timeout 300 python bench.py --pairs 10000000 --dist-selftest --no-e2e --cpu-sample 0 > gpurun_out/t4_selftest.json 2> gpurun_out/t4_selftest.err )
