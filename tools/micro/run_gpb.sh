#!/bin/bash
# runs the gemm_pl_bench variants on the GPU box: tools/micro/run_gpb.sh <out file>
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
out=${1:-gpurun_out/gpb.txt}
: > $out
for v in full xcd pf2 pf2_xcd pf2_xcd_pad pf2_xcd_prio pf2_same nogload; do
  for cfg in "0 1 2048" "0 2 2048" "4 1 2048" "4 2 2048" "4 2 256"; do
    echo -n "$v: " >> $out
    timeout -k 5 60 tools/micro/gpb_$v $cfg >> $out 2>&1 || { echo "FAILED $v $cfg" >> $out; exit 1; }
  done
done
