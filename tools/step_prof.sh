#!/bin/bash
# usage: tools/step_prof.sh <tag> [step_prof.py args...]   -> gpurun_out/step_<tag>.csv (kernel stats) + .log
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/stepprof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/step_prof.py "$@" > gpurun_out/step_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > gpurun_out/step_$tag.csv
python3 -c "import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if len(r)>3: print(\"%-84s %6s %10s\" % (r[0][:84], r[1], r[3][:9]))" gpurun_out/step_$tag.csv
tail -1 gpurun_out/step_$tag.log
