#!/bin/bash
# round-2 GPU job D: whole parity suite, default bench line, serial-mode rocprof summaries (csv)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02d_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02d_status.txt
tail -4 gpurun_out/r02d_tests.log
python bench.py > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02d_status.txt
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 2 3 5; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02d_prof_$c.json 2> gpurun_out/r02d_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02d_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02d_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02d_status.txt
done
