#!/bin/bash
# wall time of the default bench.py invocation (the driver's N = 1 command)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
s=$(date +%s.%N)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
e=$(date +%s.%N)
echo "bench.py default: $(echo "$e - $s" | bc) s, rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_algorithmic"], d["cpu_baseline"]["all_cores"]["note"][:70], [s["value"] for s in d["secondary"]])
PY
