#!/bin/bash
# same-box A/B of two builds of the library: tools/_build/lib_prev.so (SKETCHEDIT_HIP_LIB) against the in-tree one, alternating
#   usage: tools/ab_lib.sh [bench args...]
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --steps 40"
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v == prev ]; then export SKETCHEDIT_HIP_LIB=$root/tools/_build/lib_prev.so; else unset SKETCHEDIT_HIP_LIB; fi
    python bench.py $Q "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v', round(d['ms_per_step'],3), ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms_per_step'])[:8]))"
  done
done
