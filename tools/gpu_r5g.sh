#!/bin/bash
tag=${1:-r5g}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
bash tools/ab_variants.sh "new w24p1 w24p2" 2>&1 | tee $out/ab_prio.txt
