#!/bin/bash
# low-latency mode A/B: parity tests of the small-grid shapes + batch-1 bench lines (ring depth 2 vs 4)
tag=$1; root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "low_latency or lowlat or two_source or graph or shard or bf16" > $out/pytest.log 2>&1; tail -n 12 $out/pytest.log
Q="--no-cpu-baseline --no-parity --no-traffic --steps 50"
for s in 256 512; do
  SE_LL_STAGES=2 python bench.py --size $s --batch 1 --low-latency on $Q > $out/b1_${s}_s2.json 2>/dev/null
  python bench.py --size $s --batch 1 --low-latency on $Q > $out/b1_${s}_s4.json 2>/dev/null
  python bench.py --size $s --batch 1 --low-latency on --dtype bf16 $Q > $out/b1_${s}_s4_bf16.json 2>/dev/null
done
python bench.py --size 256 --batch 2 --low-latency on $Q > $out/b2_256_s4.json 2>/dev/null
python bench.py --size 256 --batch 4 --low-latency on $Q > $out/b4_256_ll.json 2>/dev/null
python bench.py --size 256 --batch 4 --low-latency off $Q > $out/b4_256_def.json 2>/dev/null
python - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); k = d["kernels"]
        print(os.path.basename(f), round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms occ", d["roofline"].get("cu_occupancy_by_grid"),
              {n: (k[n]["ms_per_step"], k[n]["workgroups_per_launch"]) for n in ("gconv_n192", "gconv_n96", "gconv_n48", "gconv_n24") if n in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
