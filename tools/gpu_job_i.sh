#!/bin/bash
# round-2 GPU job I: full parity suite (verbose, so an abort names its test), default bench, single-rank RCCL pass of the
# distributed path, config-5 kernel summary + PMC traffic with the 32x32x64 fp8 MFMA
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -v > gpurun_out/r02i_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02i_status.txt
grep -E "passed|failed" gpurun_out/r02i_tests.log | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02i_status.txt
RG_FORCE_REDUCE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02i_bench_rccl1.json 2> gpurun_out/r02i_bench_rccl1.err; echo "rccl1 rc=$?" | tee -a gpurun_out/r02i_status.txt
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 5; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02i_prof_$c.json 2> gpurun_out/r02i_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02i_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02i_status.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/r02i_pmc_${c}_$ctr.err
    echo "pmc $c $ctr rc=$?" | tee -a gpurun_out/r02i_status.txt
    python tools/prof_summary.py /tmp/pmc_${c}_$ctr gpurun_out/r02_c${c}_pmc_$ctr.csv | tee -a gpurun_out/r02i_status.txt
  done
  cp profiles/r02_pmc_traffic.json gpurun_out/r02_pmc_traffic.json
  python tools/pmc_traffic.py gpurun_out/r02_c${c}_pmc_FETCH_SIZE.csv gpurun_out/r02_c${c}_pmc_WRITE_SIZE.csv 2 gpurun_out/r02_pmc_traffic.json $c | tee -a gpurun_out/r02i_status.txt
done
timeout -k 10 300 python tools/debug/aten_callers.py 5 > gpurun_out/r02i_aten_5.txt 2>&1; echo "aten 5 rc=$?" | tee -a gpurun_out/r02i_status.txt
