cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/t57_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/t57_smoke.log
( timeout 900 python bench.py > gpurun_out/t57_bench.json 2> gpurun_out/t57_bench.err; tail -3 gpurun_out/t57_bench.err ) > gpurun_out/t57_bench.log 2>&1
