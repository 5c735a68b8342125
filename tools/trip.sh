# scratch: what one gpurun trip runs (edit per trip).  This version = the round-end validation.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/trip_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/trip_smoke.log
( timeout 900 python bench.py > gpurun_out/trip_bench.json 2> gpurun_out/trip_bench.err; tail -3 gpurun_out/trip_bench.err ) > gpurun_out/trip_bench.log 2>&1
