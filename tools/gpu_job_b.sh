#!/bin/bash
# round-2 GPU job B: the new fp8 / DPTN tests first (fresh kernels), then bench config 5
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_f8_gpu.py -x -q -m gpu > gpurun_out/r02b_f8.log 2>&1; echo "f8 rc=$?" | tee gpurun_out/r02b_status.txt
timeout -k 10 600 python -m pytest tests/test_dptn_gpu.py -q -m gpu > gpurun_out/r02b_dptn.log 2>&1; echo "dptn rc=$?" | tee -a gpurun_out/r02b_status.txt
timeout -k 10 300 python bench.py --config 5 --no-others --no-cpu-baseline > gpurun_out/r02b_bench5.json 2> gpurun_out/r02b_bench5.err; echo "bench5 rc=$?" | tee -a gpurun_out/r02b_status.txt
tail -5 gpurun_out/r02b_f8.log gpurun_out/r02b_dptn.log
