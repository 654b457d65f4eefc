#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite on the new build, the netM F(4,3) flip statistics, the c2 line per F43 mode,
# and the 32x32x16 MFMA microbenchmark
tag=${1:-r5a}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $out/pytest.log 2>&1; tail -n 12 $out/pytest.log
timeout 400 python tools/f43_flips.py 12 > $out/f43_flips.json 2> $out/f43_flips.err; tail -c 1500 $out/f43_flips.json
Q="--no-cpu-baseline --no-traffic --no-secondary --no-parity --steps 40"
for rep in 1 2; do
  for v in 1 3 2 0; do
    SE_WINOGRAD_F43=$v timeout 200 python bench.py $Q > $out/c2_f43_${v}_$rep.json 2> $out/c2_f43_${v}_$rep.err
    python - $out/c2_f43_${v}_$rep.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("F43=%s %.1f img/s %.3f ms wino_n192 %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernels"]["wino_n192"]["ms_per_step"]))
except Exception as e:
    print("unreadable", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
(cd /tmp && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma32_rate $root/tools/ubench/mfma32_rate.hip && /tmp/mfma32_rate) > $out/mfma32_rate.txt 2>&1; cat $out/mfma32_rate.txt
