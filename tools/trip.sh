cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/t2_tests.log
( timeout 300 python bench.py --chr1 --no-e2e --cpu-sample 0 > gpurun_out/t2_chr1.json 2> gpurun_out/t2_chr1.err )
( timeout 300 python bench.py --pairs 10000000 --no-e2e --cpu-sample 0 > gpurun_out/t2_g10.json 2> gpurun_out/t2_g10.err )
( BENCH_ARGS="--pairs 10000000" TAG=t2_pmc timeout 900 tools/pmc.sh ) > gpurun_out/t2_pmc.log 2>&1
