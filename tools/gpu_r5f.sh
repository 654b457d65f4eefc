#!/bin/bash
# round 5: the whole -m gpu suite on the final build, then longer randomised sweeps (end-to-end sizes, op shapes, attention)
tag=${1:-r5f}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short > $out/pytest.log 2>&1; tail -n 8 $out/pytest.log
timeout 900 python tools/fuzz_sizes.py 300 51 > $out/fuzz_sizes.log 2>&1; tail -n 6 $out/fuzz_sizes.log
timeout 600 python tools/fuzz_ops.py 3000 52 > $out/fuzz_ops.log 2>&1; tail -n 4 $out/fuzz_ops.log
timeout 600 python tools/fuzz_attention.py 300 53 > $out/fuzz_att.log 2>&1; tail -n 4 $out/fuzz_att.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
