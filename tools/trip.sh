cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python tools/decode_sweep.py 16:64 16:64:11 16:64:12 16:128:11 18:64:12 14:64:9 20:64:13 ) > gpurun_out/t9_sweep.log 2>&1
