cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/t10_tests.log
run() { python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-finalize --pairs 10000000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s k1 %.3f ms' % ('$1', d['stage_ms']['classify_k1']))"; }
( run base4096; RSQC_K1_GRID=2048 run grid2048; RSQC_K1_GRID=8192 run grid8192; RSQC_K1_GRID=16384 run grid16384 ) > gpurun_out/t10_k1.log 2>&1
( timeout 200 python tools/k1_prof.py --pairs 10000000 ) > gpurun_out/t10_prof.log 2>&1
( TAG=t10_kt timeout 400 tools/kernel_stats.sh ) > gpurun_out/t10_kt.log 2>&1
