#!/bin/bash
# Same-box A/B of the attention forms (run on the GPU box): the three bench configurations with the streaming passes / E GEMM
# switched by environment (se_attention.hip: SE_ATT_FUSED, SE_ATT_FUSED_BF16, SE_ATT_PTILDE_LDS, SE_ATT_STATS_LDS, SE_ATT_E16,
# SE_ATT_SYM), alternating, per-label times from the in-library profiler.   usage: tools/att_ab.sh [reps]
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --steps 20"
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py $Q "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-16s' % '$label', round(d['ms_per_step'],3), ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('att_prep','att_score','att_softmax','att_boxsum','att_pv') if n in k))"
}
for rep in $(seq ${1:-2}); do
  run c2_three SE_ATT_FUSED=0 --
  run c2_r3fused SE_ATT_PTILDE_LDS=0 --
  run c2_allE SE_ATT_SYM=0 --
  run c2_default X=1 --
  run c3_three SE_ATT_FUSED=0 -- --size 512 --batch 8
  run c3_r3fused SE_ATT_FUSED=1 SE_ATT_PTILDE_LDS=0 -- --size 512 --batch 8
  run c3_r3stats SE_ATT_STATS_LDS=0 -- --size 512 --batch 8
  run c3_allE SE_ATT_SYM=0 -- --size 512 --batch 8
  run c3_default X=1 -- --size 512 --batch 8
  run c5_three SE_ATT_FUSED_BF16=0 -- --size 512 --batch 16 --dtype bf16
  run c5_fp32E SE_ATT_E16=0 -- --size 512 --batch 16 --dtype bf16
  run c5_default X=1 -- --size 512 --batch 16 --dtype bf16
done
