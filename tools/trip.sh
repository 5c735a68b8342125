cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 200 python tools/k1_prof.py --pairs 10000000 ) > gpurun_out/t3_prof.log 2>&1
( timeout 200 python tools/k1_prof.py --chr1 --pairs 5000000 ) > gpurun_out/t3_prof_chr1.log 2>&1
( TAG=t3_kt timeout 400 tools/kernel_stats.sh ) > gpurun_out/t3_kt.log 2>&1
