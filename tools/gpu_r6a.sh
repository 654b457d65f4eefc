#!/bin/bash
# round 6, call A: full -m gpu suite and the default bench invocation (rate check, PMC traffic labels)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6a; mkdir -p $out; cd $root
timeout 1700 python -m pytest tests -m gpu -q --tb=short -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 25 $out/pytest.log
( time python bench.py > $out/c2_bench.json 2> $out/c2_bench.err ) 2> $out/c2_time.txt; echo "bench rc=$?"; tail -c 1500 $out/c2_bench.json; tail -3 $out/c2_time.txt
nproc; lscpu | head -20 > $out/lscpu.txt
