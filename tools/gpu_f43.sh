#!/bin/bash
# GPU check of the hybrid Winograd kernel: its op tests, then the c2 bench line with SE_WINOGRAD_F43 = 0 / 1 alternating
#   usage: tools/gpu_f43.sh <tag> [pytest -k expr] [reps]
tag=${1:-f43}; kexpr=${2:-winograd}; reps=${3:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "$kexpr" > $out/pytest.log 2>&1; tail -n 25 $out/pytest.log
Q="--no-cpu-baseline --no-traffic --no-secondary --steps 30"
for rep in $(seq $reps); do
  for v in 0 1; do
    SE_WINOGRAD_F43=$v timeout 300 python bench.py $Q > $out/c2_f43_${v}_$rep.json 2> $out/c2_f43_${v}_$rep.err
    python - $out/c2_f43_${v}_$rep.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    k = d["kernels"]
    print("F43=%s" % sys.argv[2], round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms  frac", d["roofline"]["frac"], "parity", d.get("parity"))
    print("   " + "  ".join("%s %.3f(%d x %.0fus)" % (n, v["ms_per_step"], v["launches_per_step"], v["avg_us"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]))
except Exception as e:
    print("unreadable", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
  done
done
