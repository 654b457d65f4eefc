#!/bin/bash
# round 6, call C: ablations for the fusion / concurrency questions
#  (1) per-layer times of the decoder tail at batch sizes whose intermediates do / do not fit the 256 MB Infinity Cache
#  (2) SE_FORK_DEFAULT=1: independent branches of netG on two streams in the default mode
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6c; mkdir -p $out; cd $root
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary --layers"
for b in 4 8 16 32 64; do python bench.py --batch $b --low-latency off $Q --steps 20 > $out/f32_256_b$b.json 2> $out/f32_256_b$b.err; done
for b in 2 4 8 16; do python bench.py --dtype bf16 --size 512 --batch $b --low-latency off $Q --steps 20 > $out/bf16_512_b$b.json 2> $out/bf16_512_b$b.err; done
for i in 1 2; do
SE_FORK_DEFAULT=1 python bench.py $Q --steps 30 > $out/fork1_c2_$i.json 2> $out/fork1_c2_$i.err
python bench.py $Q --steps 30 > $out/fork0_c2_$i.json 2> $out/fork0_c2_$i.err
done
SE_FORK_DEFAULT=1 python bench.py --size 512 --batch 8 $Q --steps 20 > $out/fork1_c3.json 2> $out/fork1_c3.err
python bench.py --size 512 --batch 8 $Q --steps 20 > $out/fork0_c3.json 2> $out/fork0_c3.err
SE_FORK_DEFAULT=1 python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/fork1_c5.json 2> $out/fork1_c5.err
python bench.py --dtype bf16 --size 512 --batch 16 $Q --steps 20 > $out/fork0_c5.json 2> $out/fork0_c5.err
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
def load(f):
    try: return json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e: return None
tail = ("conv15_upsample_conv", "conv16", "conv17", "allconv15_upsample_conv", "allconv16", "allconv17", "conv1", "conv2_downsample", "conv4_downsample", "conv5")
for f in sorted(glob.glob(out + "/f32_256_b*.json")) + sorted(glob.glob(out + "/bf16_512_b*.json")):
    d = load(f)
    if not d: print(f, "unreadable"); continue
    B = d["config"]["per_gpu_batch"]
    L = d["layers"]
    row = {}
    for k, v in L.items():
        n = k.split(":")[1]
        if n in tail: row[n] = round(1e3 * v["ms"] / v["n"] / B, 2)     # us per image and launch
    print(os.path.basename(f), "%.1f img/s" % d["value"], "us/img:", row)
for f in sorted(glob.glob(out + "/fork*.json")):
    d = load(f)
    print(os.path.basename(f), None if not d else (round(d["value"], 1), round(d["ms_per_step"], 3)))
PY
