#!/bin/bash
# quick GPU A/B: a pytest -k selection and the three headline bench lines with their attention / per-label times
#   usage: tools/gpu_quick.sh <tag> "<pytest -k expr>" [labels...]
tag=$1; kexpr=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "$kexpr" > $out/pytest.log 2>&1; tail -n 15 $out/pytest.log
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
python bench.py $Q > $out/c2.json 2>$out/c2.err
python bench.py --size 512 --batch 8 $Q --steps 20 > $out/c3.json 2>$out/c3.err
python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/c5.json 2>$out/c5.err
python - $out "$@" <<'PY'
import json, sys
out, labels = sys.argv[1], sys.argv[2:]
for f in ("c2", "c3", "c5"):
    try:
        d = json.loads([l for l in open("%s/%s.json" % (out, f)).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e, open("%s/%s.err" % (out, f)).read()[-400:]); continue
    k = d["kernels"]
    sel = labels or sorted(k, key=lambda n: -k[n]["ms_per_step"])[:10]
    print(f, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms", {n: k[n]["ms_per_step"] for n in sel if n in k})
PY
