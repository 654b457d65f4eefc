#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6d; mkdir -p $out; cd $root
for v in 1 0; do
SE_LOADER_NUMPY=$v timeout 300 python bench.py --e2e > $out/e2e_np$v.json 2> $out/e2e_np$v.err; echo "e2e numpy=$v rc=$?"; python -c "
import json,sys; d=json.loads(open('$out/e2e_np$v.json').read().splitlines()[-1])
for k,v in d['writers'].items(): print(k, v['e2e_images_per_sec'], v['stage_images_per_sec'], v['e2e_over_slowest_stage'], v['main_thread_s'])"; tail -2 $out/e2e_np$v.err
done
