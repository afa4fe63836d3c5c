#!/bin/bash
# The whole GPU suite including the duplicates kept out of the default `-m gpu` run (tests/conftest.py, marker `slow`):
#     /usr/local/graft/bin/gpurun --timeout 1200 -- tools/gpu_tests_all.sh
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
RG_RUN_SLOW=1 python -m pytest tests -q -m gpu --durations=40 2>&1 | tee gpurun_out/gpu_tests_all.log | tail -60
