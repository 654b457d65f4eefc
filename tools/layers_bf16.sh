cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lay
Q="--dtype bf16 --size 512 --batch 16 --no-cpu-baseline --no-parity --no-traffic --steps 10 --warmup 3 --layers"
python bench.py $Q > gpurun_out/lay/r.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/lay/r.json").read().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"])
for k,v in d["layers"].items(): print(k, v)
PY
