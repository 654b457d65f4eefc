#!/bin/bash
# batch-1 latency A/B (low-latency mode): usage tools/gpu_ll.sh  [env assignments to compare against the default]
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/ll
Q="--no-cpu-baseline --no-parity --no-traffic --steps 50 --batch 1"
run() { tag=$1; shift; env "$@" python bench.py $Q --size 256 > gpurun_out/ll/${tag}_256.json 2>/dev/null; env "$@" python bench.py $Q --size 512 > gpurun_out/ll/${tag}_512.json 2>/dev/null
python - $tag <<'PY'
import json,sys
for s in (256,512):
    d=json.loads([l for l in open("gpurun_out/ll/%s_%d.json"%(sys.argv[1],s)).read().splitlines() if l.startswith("{")][-1])
    k=d["kernels"]; print(sys.argv[1], s, round(d["ms_per_step"],3), {n:k[n]["ms_per_step"] for n in k if n.startswith("gconv")})
PY
}
run default X=1
for e in "$@"; do run "$e" $e; done
