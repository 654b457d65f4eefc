cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lay
Q="--dtype bf16 --size 512 --batch 16 --no-cpu-baseline --no-parity --no-traffic --steps 10 --warmup 3 --layers"
python bench.py $Q > gpurun_out/lay/r.json 2>/dev/null
SE_RCONV16=0 python bench.py $Q > gpurun_out/lay/g.json 2>/dev/null
python - <<'PY'
import json
def L(f):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); return d
r=L("gpurun_out/lay/r.json"); g=L("gpurun_out/lay/g.json")
print(r["ms_per_step"], g["ms_per_step"])
for k,v in r["layers"].items():
    if k.startswith("gconv_n192"):
        print(k, v, g["layers"].get(k))
PY
