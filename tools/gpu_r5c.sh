#!/bin/bash
# round 5, third GPU call: mid-step barrier pipeline in rconv16b / rconv96 (bf16): parity, then same-box A/B on config 5
tag=${1:-r5c}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "bf16 or saturate" > $out/pytest.log 2>&1; tail -n 8 $out/pytest.log
bash tools/ab_variants.sh "rc16old rc96old new" --dtype bf16 --size 512 --batch 16 2>&1 | tee $out/ab_c5.txt
