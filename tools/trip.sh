cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/t11_tests.log
( TAG=t11_kt BENCH_ARGS="" timeout 500 tools/kernel_stats.sh ) > gpurun_out/t11_kt.log 2>&1
ls gpurun_out/prof/t11_kt >> gpurun_out/t11_kt.log
