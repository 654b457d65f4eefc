#!/bin/bash
# usage: tools/layers.sh [layer-name-substring]  -- per-layer timing table of one bench step (developer aid)
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --layers 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
flt=sys.argv[1] if len(sys.argv)>1 else ''
print('img/s %.1f  ms %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['layers'].items():
    if flt in k: print('   %-22s %7.4f ms x%d  %6.1f TF' % (k, v['ms'], v['n'], v['tflops']))
" "$1"
