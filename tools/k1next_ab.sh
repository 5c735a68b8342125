#!/bin/bash
# Builds the k1-next changes one at a time as variant libraries (gpurun_variants/<name>/lib), for ONE gpurun call:
#   tools/k1next_ab.sh && gpurun --timeout 600 -- "TAG=r5a TESTS=0 PMC=0 KSTATS=0 tools/r4_run.sh"
# tree = all four on; every other line turns ONE of them off; `none` = the round-4 kernel.
set -e
cd "$(dirname "$0")/../rnaseqc_amd/csrc"
make variant NAME=no_constmask DEFS=-DK1E_CONSTMASK=0
make variant NAME=no_opaque DEFS=-DK1E_NO_OPAQUE_SWITCHES
make variant NAME=no_lazypair DEFS=-DK1E_LAZYPAIR=0
make variant NAME=no_uniform DEFS=-DK1E_NO_UNIFORM
make variant NAME=none DEFS="-DK1E_CONSTMASK=0 -DK1E_NO_OPAQUE_SWITCHES -DK1E_LAZYPAIR=0 -DK1E_NO_UNIFORM"
