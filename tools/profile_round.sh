#!/bin/bash
# Run on the GPU box (gpurun): the round's bench line, the rocprofv3 kernel-trace summary and the PMC passes.
# Everything lands under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (tools/rocprof_summary.py,
# tools/pmc_summary.py).   usage: tools/profile_round.sh <tag>
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
B="python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- $B > $out/bench_under_rocprof.json 2> $out/prof.err
P="python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d $out/pmc1 --output-format csv -- $P > /dev/null 2> $out/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc2 --output-format csv -- $P > /dev/null 2> $out/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc3 --output-format csv -- $P > /dev/null 2> $out/pmc3.err
cd $root
find $out -name "*.db" -o -name "*counter_collection.csv" | head
# the databases are large: keep only what the summaries need
python tools/rocprof_summary.py $(find $out/prof -name "*.db" | head -1) $out/kernel_stats.csv
for d in pmc1 pmc2 pmc3; do
  dd=$(dirname $(find $out/$d -name "*counter_collection.csv" | head -1))
  mkdir -p $out/${d}_flat; cp $dd/*counter_collection.csv $dd/*kernel_trace.csv $out/${d}_flat/ 2>/dev/null
done
python tools/pmc_summary.py --json $out/pmc_traffic.json $out/pmc1_flat $out/pmc2_flat $out/pmc3_flat > $out/pmc_summary.txt
find $out -name "*.db" -delete; rm -rf $out/pmc1 $out/pmc2 $out/pmc3
du -sh $out
