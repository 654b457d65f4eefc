#!/bin/bash
# per-layer timing tables (in-library HIP events) of the three headline configurations
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=$root/gpurun_out/layers; mkdir -p $out
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --layers --steps 20"
python bench.py $Q > $out/c2.json 2>/dev/null
python bench.py $Q --dtype bf16 --size 512 --batch 16 > $out/c5.json 2>/dev/null
python - $out <<'PY'
import json, sys
for f in ("c2", "c5"):
    d = json.loads(open("%s/%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    L = d["layers"]
    print("==", f, round(d["ms_per_step"], 3), "ms")
    for n, v in sorted(L.items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-40s n=%d  %.3f ms  %s TF" % (n, v["n"], v["ms"], v["tflops_executed"]))
PY
