cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python tools/decode_sweep.py 32:64:21 32:64:23 36:64:23 40:64:25 40:64:27 32:32:20 ) > gpurun_out/t17_sweep.log 2>&1
