#!/bin/bash
# low-latency vs default mode over small call sizes (gpurun): where _lib.LOW_LATENCY_MAX_PIXELS should sit
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
for s in 256 512; do for b in 1 2 3 4 6 8; do for m in on off; do
 [ $s = 512 ] && [ $b -gt 3 ] && continue
 python bench.py --size $s --batch $b --low-latency $m $Q --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$s', '$b', '$m', round(d['ms_per_step'],3), round(d['value'],1))"
done; done; done
