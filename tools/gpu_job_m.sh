#!/bin/bash
# round-2 GPU job M: full parity suite + smoke, default bench, A/B of the fused BatchNorm (RG_BN_FUSED=0), config-3 kernel summary
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -v > gpurun_out/r02m_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02m_status.txt
grep -E "FAILED|passed|failed" gpurun_out/r02m_tests.log | tail -12
python __graft_entry__.py smoke > gpurun_out/r02m_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r02m_status.txt
tail -1 gpurun_out/r02m_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02m_bench.json 2> gpurun_out/r02m_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02m_status.txt
for c in 3 4a 2; do
  RG_BN_FUSED=0 python bench.py --config $c --no-others --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02m_bench_${c}_unfused.json 2>/dev/null; echo "unfused $c rc=$?" | tee -a gpurun_out/r02m_status.txt
  python bench.py --config $c --no-others --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02m_bench_${c}_fused.json 2>/dev/null; echo "fused $c rc=$?" | tee -a gpurun_out/r02m_status.txt
done
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 3; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02m_prof_$c.json 2> gpurun_out/r02m_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02m_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02m_status.txt
done
