#!/bin/bash
# usage (GPU box): tools/bench_pmc.sh <tag> [bench.py args...] -> gpurun_out/<tag>_sq_counters.txt
# One --pmc pass (no trace domains) over an eager bench step; MFMA utilisation per kernel =
# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz).
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \
  --output-format csv -d /tmp/pmc_$tag/a -o p -- python3 bench.py --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 1 "$@" > /tmp/pmc_$tag.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_$tag/a > gpurun_out/${tag}_sq_counters.txt
python3 - gpurun_out/${tag}_sq_counters.txt <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split("\n")
for i in range(0, len(lines) - 1, 2):
    m = re.match(r"(.*?)\s+avg ([0-9.]+) us x (\d+)", lines[i])
    if not m: continue
    c = dict((k, float(v)) for k, v in re.findall(r"(\w+)=([0-9.e+]+)", lines[i + 1]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * float(m.group(2)) * 1e-6 * 2.4e9)
        print("%-72s %9.1f us  MFMA pipe busy %5.1f %%  VALU/MFMA instr %.2f" % (m.group(1)[:72], float(m.group(2)), 100 * util,
              (c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0)) / max(c.get("SQ_INSTS_MFMA", 1), 1)))
PY
