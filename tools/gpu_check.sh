#!/bin/bash
# Run on the GPU box (gpurun): the -m gpu test suite plus a set of short bench lines, everything under gpurun_out/<tag>/.
#   usage: tools/gpu_check.sh <tag> [pytest -k expression] [nobench]
tag=${1:-check}
kexpr=${2:-}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
if [ -n "$kexpr" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -k "$kexpr" > $out/pytest.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > $out/pytest.log 2>&1
fi
tail -n 40 $out/pytest.log
if [ "$3" == "nobench" ]; then exit 0; fi
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
# batch-1 latency: round-1 kernels (SE_ATT_V1 patch-form attention, default launch shapes) vs the low-latency mode
SE_ATT_V1=1 timeout 300 python bench.py --batch 1 --low-latency off $Q --steps 30 > $out/b1_256_r01.json 2> $out/b1_256_r01.err
timeout 300 python bench.py --batch 1 --low-latency off $Q --steps 30 > $out/b1_256_default.json 2> $out/b1_256_default.err
timeout 300 python bench.py --batch 1 --low-latency on $Q --steps 30 --layers > $out/b1_256_lowlat.json 2> $out/b1_256_lowlat.err
timeout 300 python bench.py --batch 1 --low-latency on --graph $Q --steps 30 > $out/b1_256_graph.json 2> $out/b1_256_graph.err
SE_ATT_V1=1 timeout 300 python bench.py --size 512 --batch 1 --low-latency off $Q --steps 20 > $out/b1_512_r01.json 2> $out/b1_512_r01.err
timeout 300 python bench.py --size 512 --batch 1 --low-latency off $Q --steps 20 > $out/b1_512_default.json 2> $out/b1_512_default.err
timeout 300 python bench.py --size 512 --batch 1 --low-latency on $Q --steps 20 > $out/b1_512_lowlat.json 2> $out/b1_512_lowlat.err
timeout 300 python bench.py --size 512 --batch 1 --low-latency on --graph $Q --steps 20 > $out/b1_512_graph.json 2> $out/b1_512_graph.err
timeout 300 python bench.py --size 512 --batch 1 --low-latency off --graph $Q --steps 20 > $out/b1_512_default_graph.json 2> $out/b1_512_default_graph.err
# config 2 / config 3 lines, old vs new attention
SE_ATT_V1=1 timeout 300 python bench.py $Q --layers > $out/c2_attv1.json 2> $out/c2_attv1.err
timeout 300 python bench.py $Q --layers > $out/c2.json 2> $out/c2.err
SE_ATT_V1=1 timeout 300 python bench.py --size 512 --batch 8 $Q --steps 20 > $out/c3_attv1.json 2> $out/c3_attv1.err
timeout 300 python bench.py --size 512 --batch 8 $Q --steps 20 --layers > $out/c3.json 2> $out/c3.err
# config 5 (bf16): 512x512 batch 16, and the config-2 shape for comparison
timeout 300 python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 --layers > $out/c5_bf16.json 2> $out/c5_bf16.err
timeout 300 python bench.py --dtype bf16 $Q --layers > $out/c2_bf16.json 2> $out/c2_bf16.err
timeout 300 python bench.py --size 512 --batch 16 $Q --steps 20 > $out/c5shape_f32.json 2> $out/c5shape_f32.err
# the RCCL branch with the one rank a 1-GPU box offers (init_process_group("nccl"), all_gather_into_tensor, side stream)
timeout 300 python bench.py --force-dist $Q --steps 10 > $out/c2_nccl_w1.json 2> $out/c2_nccl_w1.err; echo "nccl w1 rc=$?" >> $out/c2_nccl_w1.err
timeout 300 python bench.py --force-dist --no-overlap $Q --steps 10 > $out/c2_nccl_w1_noov.json 2> $out/c2_nccl_w1_noov.err; echo "nccl w1 no-overlap rc=$?" >> $out/c2_nccl_w1_noov.err
for f in $out/*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
    k = d.get("kernels") or {}
    print("  %.1f img/s  %.3f ms/step  exec=%s" % (d["value"], d["ms_per_step"], d["config"].get("execution")))
    r = d.get("roofline") or {}
    print("  dominant %s frac %.3f  forward_executed_frac %s" % (r.get("kernel"), r.get("frac", 0), r.get("forward_executed_frac")))
    print("  " + "  ".join("%s %.3f" % (n, v["ms_per_step"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]))
except Exception as e:
    print("  unreadable:", e)
    print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
