#!/bin/bash
# the -m gpu suite under each developer switch: every switch selects between kernels that compute
# the same layer, so the parity tests must stay green
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/switches; mkdir -p $out; cd $root
for sw in ${SWITCHES:-"SE_ATT_V1=1" "SE_RCONV16=0" "SE_RCONV16_TILE=16" "SE_RCONV96=0" "SE_RTILE=0" "SE_LL_STAGES=4" "SE_WINOGRAD=0" "SE_WINOGRAD48=0" "SE_WINO48_TILES=128" "SE_WINOUP_TILES=128" "SE_WINOGRAD_UP=0" "SE_WINOGRAD_UP48=0" "SE_XCD_REMAP=0" "SE_GCONV_FAST=0"}; do
  env $sw timeout 900 python -m pytest tests -m gpu -q --tb=line -x > $out/$sw.log 2>&1
  echo "$sw: $(tail -n 1 $out/$sw.log)"
done
