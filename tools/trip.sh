cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/t5_tests.log
( timeout 300 python bench.py --no-e2e --cpu-sample 0 --steps 5 > gpurun_out/t5_bench.json 2> gpurun_out/t5_bench.err )
( timeout 300 python bench.py --no-e2e --cpu-sample 0 --steps 5 --bed --pairs 10000000 > gpurun_out/t5_bed.json 2> gpurun_out/t5_bed.err )
