#!/bin/bash
# Round-6 GPU calls (gpurun), one sub-command each; everything lands under gpurun_out/r6_<cmd>/.
#   tools/gpu_round6.sh suite      full -m gpu suite + the default bench invocation (timed)
#   tools/gpu_round6.sh e2e        the files-to-files leg alone: own decoder processes, then DataLoader workers
#   tools/gpu_round6.sh ablate     fusion / two-stream ablations: per-layer times over batch sizes, SE_FORK_DEFAULT A/B
#   tools/gpu_round6.sh probe      where the e2e loop loses time (tools/e2e_probe.py)
cmd=${1:-suite}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_$cmd; mkdir -p $out; cd $root
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
case $cmd in
suite)
  timeout 1700 python -m pytest tests -m gpu -q --tb=short -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 6 $out/pytest.log
  ( time python bench.py > $out/c2_bench.json 2> $out/c2_bench.err ) 2> $out/c2_time.txt; echo "bench rc=$?"; tail -c 600 $out/c2_bench.json; tail -3 $out/c2_time.txt ;;
e2e)
  for v in "" "--e2e-dataloader"; do
    timeout 300 python bench.py --e2e $v > $out/e2e$v.json 2> $out/e2e$v.err; echo "e2e $v rc=$?"
    python - "$out/e2e$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().splitlines()[-1])
for k, v in d["writers"].items():
    print(k, v["e2e_images_per_sec"], v["decode_workers"], v["encode_workers"], v["stage_images_per_sec"], v["e2e_over_slowest_stage"], v["main_thread_s"])
PY
  done ;;
ablate)
  for b in 4 8 16 32 64; do python bench.py --batch $b --low-latency off $Q --layers --steps 20 > $out/f32_256_b$b.json 2> $out/f32_256_b$b.err; done
  for b in 2 4 8 16; do python bench.py --dtype bf16 --size 512 --batch $b --low-latency off $Q --layers --steps 20 > $out/bf16_512_b$b.json 2> $out/bf16_512_b$b.err; done
  for i in 1 2; do
    SE_FORK_DEFAULT=1 python bench.py $Q --steps 30 > $out/fork1_c2_$i.json 2> $out/fork1_c2_$i.err
    SE_FORK_DEFAULT=0 python bench.py $Q --steps 30 > $out/fork0_c2_$i.json 2> $out/fork0_c2_$i.err
  done
  SE_FORK_DEFAULT=1 python bench.py --size 512 --batch 8 $Q --steps 20 > $out/fork1_c3.json 2> $out/fork1_c3.err
  SE_FORK_DEFAULT=0 python bench.py --size 512 --batch 8 $Q --steps 20 > $out/fork0_c3.json 2> $out/fork0_c3.err
  SE_FORK_DEFAULT=1 python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/fork1_c5.json 2> $out/fork1_c5.err
  SE_FORK_DEFAULT=0 python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/fork0_c5.json 2> $out/fork0_c5.err
  python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
def load(f):
    try: return json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception: return None
tail = ("conv15_upsample_conv", "conv16", "conv17", "allconv15_upsample_conv", "allconv16", "allconv17", "conv1", "conv2_downsample")
for f in sorted(glob.glob(out + "/f32_256_b*.json")) + sorted(glob.glob(out + "/bf16_512_b*.json")):
    d = load(f)
    if not d: print(f, "unreadable"); continue
    B = d["config"]["per_gpu_batch"]
    row = {k.split(":")[1]: round(1e3 * v["ms"] / v["n"] / B, 2) for k, v in d["layers"].items() if k.split(":")[1] in tail}
    print(os.path.basename(f), "%.1f img/s" % d["value"], "us per image and launch:", row)
for f in sorted(glob.glob(out + "/fork*.json")):
    d = load(f)
    print(os.path.basename(f), None if not d else (round(d["value"], 1), round(d["ms_per_step"], 3)))
PY
  ;;
probe) timeout 300 python tools/e2e_probe.py 2>&1 | grep "^[0-9]" ;;
esac
