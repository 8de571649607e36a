#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag> [bench.py args...]
#   -> gpurun_out/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py <args>`
#   -> gpurun_out/<tag>_traffic.json/.txt  FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes, no trace domains)
#   -> gpurun_out/<tag>_bench.json         the bench line of an unprofiled run
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf /tmp/prof_$tag
python3 bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/k -o p -- python3 bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 "$@" > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag/k -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$tag/f -o p -- python3 bench.py --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 1 "$@" >> /tmp/prof_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$tag/w -o p -- python3 bench.py --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 1 "$@" >> /tmp/prof_$tag.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_$tag/f /tmp/prof_$tag/w > gpurun_out/${tag}_traffic.txt
python3 tools/make_traffic_json.py "${WORKLOAD:-cfg2}" "${TILES:-32}" gpurun_out/${tag}_traffic.json /tmp/prof_$tag/f /tmp/prof_$tag/w
python3 tools/show_bench.py gpurun_out/${tag}_bench.json
head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
