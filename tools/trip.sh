cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/t40_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/t40_smoke.log
( timeout 900 python bench.py > gpurun_out/t40_bench.json 2> gpurun_out/t40_bench.err; tail -5 gpurun_out/t40_bench.err ) > gpurun_out/t40_bench.log 2>&1
