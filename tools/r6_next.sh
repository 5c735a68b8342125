#!/bin/bash
# round 6, second session: K1 A/B builds (gpurun_variants/*) through tools/r6_run.sh, then the stage-mark section timers of the `smark` build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r6o} VARIANTS="${VARIANTS}" TESTS=${TESTS:-0} PMC=${PMC:-0} KSTATS=${KSTATS:-0} bash tools/r6_run.sh
if [ -f gpurun_variants/smark/lib/librnaseqc_amd.so ] && [ "${SMARK:-1}" = "1" ]; then
  K1_STAGE_MARKS=1 RSQC_LIB=$GRAFT_REPO_ROOT/gpurun_variants/smark/lib/librnaseqc_amd.so timeout 400 python tools/k1_prof.py --pairs ${SMARK_PAIRS:-20000000} > gpurun_out/${TAG:-r6o}/stage_marks.txt 2>&1
  cat gpurun_out/${TAG:-r6o}/stage_marks.txt | tail -22
fi
