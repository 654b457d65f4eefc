#!/bin/bash
# usage: tools/sweep.sh "ENV=val ENV2=val" ...   -- one bench run per argument, prints the per-kernel table
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s %.1f  ms %.2f  stack %.1f TF' % (d['value'], d['ms_per_step'], d['roofline']['gated_conv_stack_tflops']))
for k,v in d['kernels'].items():
    if v['ms_per_step']>0.3: print('   %-12s %7.3f ms  %s TF' % (k, v['ms_per_step'], round(v['tflops'],1) if v['tflops'] else None))
"
done
