cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/t49_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/t49_smoke.log
( timeout 900 python bench.py > gpurun_out/t49_bench.json 2> gpurun_out/t49_bench.err; tail -5 gpurun_out/t49_bench.err ) > gpurun_out/t49_bench.log 2>&1
( TAG=t49_s0 PAIRS=50000000 SEQ_MODE=0 tools/decode_kstats.sh 2>&1 | head -24 ) > gpurun_out/t49_kstats.log 2>&1
rm -f gpurun_out/prof/t49_s0/*.db gpurun_out/prof/t49_s0/*trace.csv
