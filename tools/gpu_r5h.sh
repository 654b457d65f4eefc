#!/bin/bash
# round 5: 24 -> 96 layers (xconv3, pmconv3) on the 48-channel Winograd kernel's CIN = 24 form -- parity, then same-box A/B
# (SE_WINOGRAD48=0 would switch the 48 -> 96 layers off too, so the A/B is against the previous build, tools/_build/lib_prev.so)
tag=${1:-r5h}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "winograd or inference_64 or inference_256 or netG_64 or sample or 512 or weights_w or fuzz or flag_variants" > $out/pytest.log 2>&1; tail -n 8 $out/pytest.log
bash tools/ab_lib.sh --layers 2>&1 | tee $out/ab_c2.txt
bash tools/ab_lib.sh --size 512 --batch 8 2>&1 | tee $out/ab_c3.txt
