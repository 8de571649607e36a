#!/bin/bash
# per-kernel time of the recurrence under each schedule: tools/r02_slotprof.sh <tag> [bench args] -> gpurun_out/<tag>_mode*_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for mode in ${MODES:-2 1 0}; do
  rm -rf /tmp/sp_$tag$mode
  ADAMVS_RECUR_MODE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$tag$mode -o p -- python3 bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 "$@" > /tmp/sp_$tag$mode.log 2>&1
  f=$(find /tmp/sp_$tag$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -30 "$f" > gpurun_out/${tag}_mode${mode}_kernel_stats.csv
  echo "== mode $mode policy ${ADAMVS_SLOT_POLICY:-default}"; python3 tools/show_kernel_stats.py gpurun_out/${tag}_mode${mode}_kernel_stats.csv "k_slot|k_conv_small|k_cand1|k_decoder|k_gru1"
done
