#!/bin/bash
tag=${1:-r5d}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
(cd /tmp && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma32_rate $root/tools/ubench/mfma32_rate.hip && /tmp/mfma32_rate) > $out/mfma32_rate.txt 2>&1; cat $out/mfma32_rate.txt
