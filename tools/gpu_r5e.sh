#!/bin/bash
# round 5: bf16 pair-of-taps dense-K first layers -- parity, then same-box A/B on config 5 (SE_RTILE_DENSE=0: the 8-channel-granule K)
tag=${1:-r5e}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "bf16" > $out/pytest.log 2>&1; tail -n 8 $out/pytest.log
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --steps 30 --dtype bf16 --size 512 --batch 16 --layers"
for rep in 1 2 3; do
  for v in 0 1; do
    SE_RTILE_DENSE=$v python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; L=d['layers']
print('dense=$v', round(d['ms_per_step'],3), ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms_per_step'])[:5]), ' | ', ' '.join('%s %.3f' % (n.split(':')[1], v['ms']) for n, v in L.items() if n.split(':')[1] in ('conv1','wconv1','xconv1','pmconv1')))"
  done
done 2>&1 | tee $out/ab_c5_dense.txt
