#!/bin/bash
# usage (GPU box): tools/small_batch_prof.sh <tag> <precision> [bench args]  -> gpurun_out/<tag>_gaps.txt, <tag>_bench.json
# cfg4's per-GPU share (4 cfg3 tiles): kernel trace of the graph replays, durations and dependent-launch gaps per kernel
tag=$1; prec=$2; shift; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
out=/tmp/sb_$tag
rm -rf $out
python3 bench.py --workload cfg3 --batch ${BATCH:-4} --precision $prec --no-cpu-baseline --no-roofline --no-cascade --steps 10 --warmup 3 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --workload cfg3 --batch ${BATCH:-4} --precision $prec --no-cpu-baseline --no-roofline --no-cascade --steps 4 --warmup 1 "$@" > /tmp/sb_$tag.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$f" 0.4 > gpurun_out/${tag}_gaps.txt
python3 tools/show_bench.py gpurun_out/${tag}_bench.json
head -30 gpurun_out/${tag}_gaps.txt
