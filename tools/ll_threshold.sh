#!/bin/bash
# low-latency vs default mode over small call sizes (gpurun): where _lib.LOW_LATENCY_MAX_PIXELS should sit
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
for cfg in "256 1" "256 2" "256 3" "256 4" "256 5" "256 6" "384 1" "384 2" "384 3" "512 1" "512 2"; do set -- $cfg
 for m in on off; do
 python bench.py --size $1 --batch $2 --low-latency $m $Q --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1 x$2 low-latency=$m', round(d['ms_per_step'],3), round(d['value'],1))"
done; done
