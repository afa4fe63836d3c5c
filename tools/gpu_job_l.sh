#!/bin/bash
# round-2 GPU job L: targeted tests for the quantiser rewrite / GeM / full-size fp8 properties, then configs 5 and 3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_f8_gpu.py tests/test_fullsize_gpu.py tests/test_dptn_gpu.py tests/test_ops_gpu.py tests/test_cc_gpu.py -m gpu -q > gpurun_out/r02l_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02l_status.txt
tail -5 gpurun_out/r02l_tests.log
for c in 5 3; do
  python bench.py --config $c --no-others --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02l_bench_$c.json 2> gpurun_out/r02l_bench_$c.err; echo "bench $c rc=$?" | tee -a gpurun_out/r02l_status.txt
done
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 5; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02l_prof_$c.json 2> gpurun_out/r02l_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02l_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02l_status.txt
done
