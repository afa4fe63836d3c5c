#!/bin/bash
# round-2 GPU job N (final collection): full parity suite + smoke, default bench, serial kernel summaries of every configuration,
# PMC traffic passes for configs 2 / 3 / 4b / 5, single-rank RCCL pass
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02n_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02n_status.txt
tail -3 gpurun_out/r02n_tests.log
python __graft_entry__.py smoke > gpurun_out/r02n_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r02n_status.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r02n_status.txt
RG_FORCE_REDUCE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02n_bench_rccl1.json 2> gpurun_out/r02n_bench_rccl1.err; echo "rccl1 rc=$?" | tee -a gpurun_out/r02n_status.txt
export RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
for c in 2 3 4a 4b 5; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02n_prof_$c.json 2> gpurun_out/r02n_prof_$c.err
  echo "prof $c rc=$?" | tee -a gpurun_out/r02n_status.txt
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/r02_c${c}_kernel_summary_serial.csv --steps 17 | tee -a gpurun_out/r02n_status.txt
done
rm -f gpurun_out/r02_pmc_traffic.json
for c in 2 3 4b 5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/r02n_pmc_${c}_$ctr.err
    echo "pmc $c $ctr rc=$?" | tee -a gpurun_out/r02n_status.txt
    python tools/prof_summary.py /tmp/pmc_${c}_$ctr gpurun_out/r02_c${c}_pmc_$ctr.csv | tee -a gpurun_out/r02n_status.txt
  done
  python tools/pmc_traffic.py gpurun_out/r02_c${c}_pmc_FETCH_SIZE.csv gpurun_out/r02_c${c}_pmc_WRITE_SIZE.csv 2 gpurun_out/r02_pmc_traffic.json $c | tee -a gpurun_out/r02n_status.txt
done
