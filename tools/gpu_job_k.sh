#!/bin/bash
# round-2 GPU job K: full parity suite, default bench, ATen-caller report, serial kernel summaries of configs 5 / 4a / 2
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -v > gpurun_out/r02k_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02k_status.txt
grep -E "FAILED|passed|failed" gpurun_out/r02k_tests.log | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/r02k_bench.json 2> gpurun_out/r02k_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02k_status.txt
for c in 2 5 4a; do
  timeout -k 10 300 python tools/debug/aten_callers.py $c > gpurun_out/r02k_aten_$c.txt 2>&1; echo "aten $c rc=$?" | tee -a gpurun_out/r02k_status.txt
done
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 5 4a 2; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02k_prof_$c.json 2> gpurun_out/r02k_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02k_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02k_status.txt
done
