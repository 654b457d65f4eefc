#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6d; mkdir -p $out; cd $root
timeout 900 python -m pytest tests/test_gpu_model_api.py -m gpu -q --tb=short -x -k "pipelined or test_py or celeb" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 5 $out/pytest.log
for v in "" "--e2e-dataloader"; do
timeout 300 python bench.py --e2e $v > $out/e2e$v.json 2> $out/e2e$v.err; echo "e2e $v rc=$?"; python -c "
import json,sys; d=json.loads(open('$out/e2e$v.json').read().splitlines()[-1])
for k,v in d['writers'].items(): print(k, v['e2e_images_per_sec'], v['decode_workers'], v['encode_workers'], v['stage_images_per_sec'], v['e2e_over_slowest_stage'], v['main_thread_s'])"; tail -2 $out/e2e$v.err
done
