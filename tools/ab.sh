#!/bin/bash
# same-box A/B of the fp32 bench line under environment settings: tools/ab.sh "A=1" "B=2 C=3" ...   (first run: default)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/ab
Q="--no-cpu-baseline --no-parity --no-traffic ${BENCH_ARGS:-}"
run() { tag=$1; shift; env "$@" python bench.py $Q > gpurun_out/ab/"$tag".json 2>/dev/null
python - "$tag" <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/ab/%s.json"%sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
k=d["kernels"]; print(sys.argv[1], round(d["ms_per_step"],3), {n:k[n]["ms_per_step"] for n in k if k[n]["ms_per_step"]>0.3})
PY
}
run default X=1
for e in "$@"; do run "$e" $e; done
run default2 X=1
