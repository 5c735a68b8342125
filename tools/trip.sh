cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py > gpurun_out/t59_bench.json 2> gpurun_out/t59_bench.err; tail -3 gpurun_out/t59_bench.err ) > gpurun_out/t59_bench.log 2>&1
( timeout 600 python -m pytest tests/test_cli.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/t59_tests.log
