#!/bin/bash
# low-latency mode: from how many workgroups on should a layer take its Winograd kernel (SE_LL_WINO_MIN_WG)?  (gpurun)
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
for cfg in "256 1" "256 2" "256 3" "512 1" "384 1" "512 2"; do set -- $cfg
 for t in 0 16 32 64 128 256; do
  SE_LL_WINO_MIN_WG=$t python bench.py --size $1 --batch $2 --low-latency on $Q --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1 x$2 ll min_wg=$t', round(d['ms_per_step'],3))"
 done
 python bench.py --size $1 --batch $2 --low-latency off $Q --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1 x$2 default', round(d['ms_per_step'],3))"
done
