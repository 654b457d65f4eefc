#!/bin/bash
# same-box comparison of several builds of the library (tools/_build/lib_<name>.so; "new" = the in-tree one), alternating
#   usage: tools/ab_variants.sh "name1 name2 ..." [bench args...]
names=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --steps 30"
for rep in 1 2; do
  for v in $names; do
    if [ $v == new ]; then unset SKETCHEDIT_HIP_LIB; else export SKETCHEDIT_HIP_LIB=$root/tools/_build/lib_$v.so; fi
    python bench.py $Q "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s' % '$v', round(d['ms_per_step'],3), ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms_per_step'])[:6]))"
  done
done
