cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/t19_tests.log
run() { python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-30s step %.3f ms  k1 %.3f  fin %.3f' % ('$1', d['ms_per_step'], d['stage_ms']['classify_k1'], d['stage_ms']['finalize_kernels']))"; }
( run base; RSQC_K3_FORCE=4 run allxl ) > gpurun_out/t19_fin.log 2>&1
