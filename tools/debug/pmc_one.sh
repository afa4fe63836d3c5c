#!/bin/bash
# SQ counters of ONE conv layer of tools/bench_conv.py (development): tools/debug/pmc_one.sh <layer> <resnet|gan> <tag>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
L=$1; W=$2; T=$3
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc1
  RG_BENCH_ONLY=$L rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc1 -- python3 tools/bench_conv.py 32 $W > /dev/null 2> gpurun_out/${T}_pmc.err || { tail -5 gpurun_out/${T}_pmc.err; exit 1; }
  python3 tools/prof_summary.py /tmp/pmc1 /tmp/pmc1.csv > /dev/null && cat /tmp/pmc1.csv >> gpurun_out/${T}_pmc.csv
done
