#!/bin/bash
# quick GPU check of a change: a pytest -k selection, then short bench lines (c2 always; c3 / c5 / b1 on request)
#   usage: tools/gpu_ab.sh <tag> "<pytest -k expr or - for none>" [c3] [c5] [b1] [ENV=VAL ...]
tag=$1; kexpr=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root
cfgs="c2"
for a in "$@"; do case $a in c3|c5|b1|noc2) cfgs="$cfgs $a";; *=*) export "$a";; esac; done
if [ "$kexpr" != "-" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "$kexpr" > $out/pytest.log 2>&1; tail -n 12 $out/pytest.log
fi
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
for c in $cfgs; do
  case $c in
    c2) [[ "$cfgs" == *noc2* ]] || python bench.py $Q --steps 30 > $out/c2.json 2>$out/c2.err ;;
    c3) python bench.py --size 512 --batch 8 $Q --steps 20 > $out/c3.json 2>$out/c3.err ;;
    c5) python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/c5.json 2>$out/c5.err ;;
    b1) python bench.py --batch 1 $Q --steps 50 > $out/b1.json 2>$out/b1.err ;;
  esac
done
python - $out <<'PY'
import json, sys, os
out = sys.argv[1]
for f in ("c2", "c3", "c5", "b1"):
    pth = "%s/%s.json" % (out, f)
    if not os.path.exists(pth):
        continue
    try:
        d = json.loads([l for l in open(pth).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e, open("%s/%s.err" % (out, f)).read()[-600:]); continue
    k = d["kernels"]
    print(f, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms  frac", d["roofline"]["frac"], "fwd", d["roofline"]["forward_executed_frac"])
    print("   " + "  ".join("%s %.3f(%d x %.0fus)" % (n, v["ms_per_step"], v["launches_per_step"], v["avg_us"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])))
PY
