#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/debug/f8_grad_depth.py fp8 jit nokink > gpurun_out/r02c_depth.log 2>&1; echo "rc=$?"
grep "cos" gpurun_out/r02c_depth.log | tail -120
