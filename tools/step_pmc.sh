#!/bin/bash
# usage: tools/step_pmc.sh <tag> [step_prof.py args...]  -> gpurun_out/pmc_<tag>.txt  (two counter passes, no trace domains)
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d /tmp/pmc_$tag/a -o p -- python3 tools/step_prof.py "$@" > /tmp/pmc_$tag.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL \
  --output-format csv -d /tmp/pmc_$tag/b -o p -- python3 tools/step_prof.py "$@" >> /tmp/pmc_$tag.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_$tag/a /tmp/pmc_$tag/b > gpurun_out/pmc_$tag.txt
cat gpurun_out/pmc_$tag.txt
