#!/bin/bash
# Run on the GPU box (gpurun): the round's bench lines with their rocprofv3 evidence.  For each configuration:
#   <cfg>_bench.json          python bench.py ... (cpu_baseline + parity + live PMC traffic for the headline config)
#   <cfg>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (per-kernel calls / avg duration)
#   <cfg>_pmc_summary.txt     three separate --pmc passes (SQ busy counters, FETCH_SIZE, WRITE_SIZE), tools/pmc_summary.py
#   <cfg>_kernel_seq.txt      per-launch durations of one forward in launch order (tools/kernel_seq.py)
# Everything lands under gpurun_out/<tag>/; copy into profiles/ afterwards.   usage: tools/profile_round.sh <tag> [cfg ...]
tag=${1:-r03}; shift
cfgs=${@:-"c2 c3 c5 b1"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-parity --no-traffic --no-secondary"
profile() {   # name, bench args
  name=$1; shift
  cd /tmp
  # one un-instrumented trace of the DEFAULT (two-stream) plan first: launch order and overlap as the timed region runs them
  timeout 300 rocprofv3 --kernel-trace -d $out/${name}_seq2 --output-format csv -- python $root/bench.py $* --steps 1 --warmup 1 $Q > /dev/null 2> $out/${name}_seq2.err
  python $root/tools/kernel_seq.py $(find $out/${name}_seq2 -name "*kernel_trace.csv" | head -1) > $out/${name}_kernel_seq_two_streams.txt 2>> $out/${name}_seq2.err
  rm -rf $out/${name}_seq2
  # every per-kernel measurement below runs the same kernels on ONE stream (SE_FORK_DEFAULT=0): a launch duration / byte count is
  # a property of one kernel only when nothing else shares the chip
  export SE_FORK_DEFAULT=0
  B="python $root/bench.py $* --steps 5 --warmup 2 $Q"
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/${name}_prof -o $name -- $B > $out/${name}_bench_under_rocprof.json 2> $out/${name}_prof.err
  P="python $root/bench.py $* --steps 1 --warmup 1 $Q"
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $out/${name}_pmc1 --output-format csv -- $P > /dev/null 2> $out/${name}_pmc1.err
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${name}_pmc2 --output-format csv -- $P > /dev/null 2> $out/${name}_pmc2.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${name}_pmc3 --output-format csv -- $P > /dev/null 2> $out/${name}_pmc3.err
  # launch sequence of one forward (no in-library events between the launches): tools/kernel_seq.py
  timeout 300 rocprofv3 --kernel-trace -d $out/${name}_seq --output-format csv -- $P > /dev/null 2> $out/${name}_seq.err
  cd $root
  python tools/kernel_seq.py $(find $out/${name}_seq -name "*kernel_trace.csv" | head -1) > $out/${name}_kernel_seq.txt 2>> $out/${name}_seq.err
  rm -rf $out/${name}_seq
  python tools/rocprof_summary.py $(find $out/${name}_prof -name "*.db" | head -1) $out/${name}_kernel_stats.csv
  flat=""
  for d in pmc1 pmc2 pmc3; do
    dd=$(dirname $(find $out/${name}_$d -name "*counter_collection.csv" | head -1))
    mkdir -p $out/${name}_${d}_flat; cp $dd/*counter_collection.csv $dd/*kernel_trace.csv $out/${name}_${d}_flat/ 2>/dev/null
    flat="$flat $out/${name}_${d}_flat"
  done
  python tools/pmc_summary.py --json $out/${name}_pmc_traffic.json $flat > $out/${name}_pmc_summary.txt
  find $out -name "*.db" -delete; rm -rf $out/${name}_pmc1 $out/${name}_pmc2 $out/${name}_pmc3 $out/${name}_pmc?_flat $out/${name}_prof
  unset SE_FORK_DEFAULT
}
cd $root
for cfg in $cfgs; do
  case $cfg in
    c2) python bench.py > $out/c2_bench.json 2> $out/c2_bench.err; tail -c 400 $out/c2_bench.json; profile c2 ;;
    c3) python bench.py --size 512 --batch 8 --steps 30 --no-cpu-baseline > $out/c3_bench.json 2> $out/c3_bench.err; profile c3 --size 512 --batch 8 ;;
    c5) python bench.py --dtype bf16 --size 512 --batch 16 --steps 30 --no-cpu-baseline > $out/c5_bf16_bench.json 2> $out/c5_bf16_bench.err; profile c5_bf16 --dtype bf16 --size 512 --batch 16 ;;
    b1) for s in 256 512; do
          SE_ATT_V1=1 python bench.py --size $s --batch 1 --low-latency off $Q --steps 50 > $out/b1_${s}_round1_kernels.json 2>/dev/null
          python bench.py --size $s --batch 1 --low-latency off $Q --steps 50 > $out/b1_${s}_default.json 2>/dev/null
          python bench.py --size $s --batch 1 --low-latency on $Q --steps 50 --layers > $out/b1_${s}_lowlat.json 2>/dev/null
          python bench.py --size $s --batch 1 --low-latency on --graph $Q --steps 50 > $out/b1_${s}_lowlat_graph.json 2>/dev/null
        done
        python bench.py --force-dist $Q --steps 20 > $out/c2_rccl_world1.json 2> $out/c2_rccl_world1.err ;;
  esac
done
du -sh $out; ls $out
