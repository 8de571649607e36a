#!/bin/bash
# usage: tools/feat_prof.sh <tag> [feat_prof.py args...]  -> gpurun_out/feat_<tag>.csv (kernel stats) + .log
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/featprof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/feat_prof.py "$@" > gpurun_out/feat_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" > gpurun_out/feat_$tag.csv
python3 -c "import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if len(r)>3: print(\"%-100s %6s %10s %6s\" % (r[0][:100], r[1], r[3][:9], r[4]))" gpurun_out/feat_$tag.csv
grep FeatureNet0 gpurun_out/feat_$tag.log
