#!/bin/bash
# 8-wave workgroup experiment (tools/micro/gemm_pipe_bench.hip v8) against today's scheme (v0)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_pipe_bench.hip -o /tmp/gpb > /dev/null 2>&1 || exit 1
: > gpurun_out/r02_micro_v8.txt
for K in 512 2304; do for per in 1 2 3; do for v in 0 816 832; do
  timeout -k 5 60 /tmp/gpb $v $per $K >> gpurun_out/r02_micro_v8.txt 2>&1 || echo "v$v per $per K $K failed" >> gpurun_out/r02_micro_v8.txt
done; done; done
cat gpurun_out/r02_micro_v8.txt
