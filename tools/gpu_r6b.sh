#!/bin/bash
# round 6, call B: the I/O pipeline -- new model-API tests, the e2e leg alone
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6b; mkdir -p $out; cd $root
timeout 1200 python -m pytest tests/test_gpu_model_api.py -m gpu -q --tb=short -x -k "pipelined or dequantize or test_py or celeb" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 5 $out/pytest.log
timeout 300 python tools/e2e_probe.py 2>&1 | grep "^[0-9]"
timeout 300 python bench.py --e2e > $out/e2e.json 2> $out/e2e.err; echo "e2e rc=$?"; python -c "
import json,sys; d=json.loads(open('$out/e2e.json').read().splitlines()[-1]); print(d['value'], d['host'])
for k,v in d['writers'].items(): print(k, json.dumps(v))"; tail -2 $out/e2e.err
