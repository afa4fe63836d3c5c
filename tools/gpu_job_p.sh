#!/bin/bash
# round-2 GPU job P: full suite with the native binding, host cost with / without it, default bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02p_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02p_status.txt
tail -2 gpurun_out/r02p_tests.log
python __graft_entry__.py smoke > gpurun_out/r02p_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r02p_status.txt
for c in 5 4a 2; do
  python tools/debug/host_cost.py $c 2>&1 | grep "config $c at" | sed 's/^/native /' | tee -a gpurun_out/r02p_status.txt
  RG_NATIVE_BIND=0 python tools/debug/host_cost.py $c 2>&1 | grep "config $c at" | sed 's/^/ctypes /' | tee -a gpurun_out/r02p_status.txt
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02p_status.txt
