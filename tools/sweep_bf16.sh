cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
Q="--dtype bf16 --size 512 --batch 16 --no-cpu-baseline --no-parity --no-traffic --steps 10 --warmup 3"
run() { tag=$1; shift; env "$@" python bench.py $Q > gpurun_out/sweep/$tag.json 2>/dev/null; python - gpurun_out/sweep/$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); k=d["kernels"]
print(sys.argv[2], round(d["ms_per_step"],3), {n:k[n]["ms_per_step"] for n in ("gconv_n192","gconv_n96","gconv_n48","gconv_n24")})
PY
}
run base A=1
run n192v1 SE_GCONV_VARIANT_N192=1
run n96v1 SE_GCONV_VARIANT_N96=1
run n96v2 SE_GCONV_VARIANT_N96=2
run n48v1 SE_GCONV_VARIANT_N48=1
run n48v2 SE_GCONV_VARIANT_N48=2
run n24v1 SE_GCONV_VARIANT_N24=1
run n24v2 SE_GCONV_VARIANT_N24=2
