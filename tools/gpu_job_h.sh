#!/bin/bash
# round-2 GPU job H: full parity suite (all failures listed), PMC traffic passes for configs 3 / 4b / 5, config-5 kernel summary
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02h_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02h_status.txt
tail -5 gpurun_out/r02h_tests.log
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 5; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02h_prof_$c.json 2> gpurun_out/r02h_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02h_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02h_status.txt
done
cp profiles/r02_pmc_traffic.json gpurun_out/r02_pmc_traffic.json 2>/dev/null
for c in 5 3 4b; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/r02h_pmc_${c}_$ctr.err
    echo "pmc $c $ctr rc=$?" | tee -a gpurun_out/r02h_status.txt
    python tools/prof_summary.py /tmp/pmc_${c}_$ctr gpurun_out/r02_c${c}_pmc_$ctr.csv | tee -a gpurun_out/r02h_status.txt
  done
  python tools/pmc_traffic.py gpurun_out/r02_c${c}_pmc_FETCH_SIZE.csv gpurun_out/r02_c${c}_pmc_WRITE_SIZE.csv 2 gpurun_out/r02_pmc_traffic.json $c | tee -a gpurun_out/r02h_status.txt
done
for c in 5 2 4a; do
  timeout -k 10 300 python tools/debug/aten_callers.py $c > gpurun_out/r02h_aten_$c.txt 2>&1; echo "aten $c rc=$?" | tee -a gpurun_out/r02h_status.txt
  timeout -k 10 300 python tools/debug/host_cost.py $c > gpurun_out/r02h_host_$c.txt 2>&1; echo "host $c rc=$?" | tee -a gpurun_out/r02h_status.txt
done
