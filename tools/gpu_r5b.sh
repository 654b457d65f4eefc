#!/bin/bash
# round 5, second GPU call: conflict-free LDS layouts (rtilew2, rtile_dense5w) -- per-op parity, same-box A/B against the build
# before them (tools/_build/lib_prev.so), LDS conflict / MFMA-busy counters; the netM F(4,3) flip statistics
tag=${1:-r5b}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$tag; mkdir -p $out; cd $root; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "conv24 or first_layer or only_netM or netM_64 or inference_64 or inference_256 or flag_variants or fuzz" > $out/pytest.log 2>&1; tail -n 6 $out/pytest.log
timeout 400 python tools/f43_flips.py 12 > $out/f43_flips.json 2> $out/f43_flips.err; python - $out/f43_flips.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for m, r in d["modes"].items():
        print("F43 mode", m, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in r.items()})
    print(d["per_weight_set_and_size"])
except Exception as e:
    print("f43_flips unreadable", e)
PY
bash tools/ab_lib.sh 2>&1 | tee $out/ab_c2.txt
cd /tmp
P="python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-traffic --no-secondary"
for v in prev new; do
  if [ $v == prev ]; then export SKETCHEDIT_HIP_LIB=$root/tools/_build/lib_prev.so; else unset SKETCHEDIT_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_$v --output-format csv -- $P > /dev/null 2> $out/pmc_$v.err
  dd=$(dirname $(find $out/pmc_$v -name "*counter_collection.csv" | head -1))
  mkdir -p $out/pmc_${v}_flat; cp $dd/*counter_collection.csv $dd/*kernel_trace.csv $out/pmc_${v}_flat/
  (cd $root; python tools/pmc_summary.py $out/pmc_${v}_flat > $out/pmc_$v.txt); grep -E "rtile|wino24_kernel<3" $out/pmc_$v.txt
  rm -rf $out/pmc_$v $out/pmc_${v}_flat
done
