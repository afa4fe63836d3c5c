#!/bin/bash
# Reproduces the per-configuration files under profiles/ on a GPU box (run through gpurun from the repo root):
#     tools/gpu_profile.sh <tag, e.g. r03> <config: 2|3|4a|4b|5> [kernel|pmc|all]
#   kernel: rocprofv3 --kernel-trace --stats of `bench.py --config C` with the kernels launched back to back on one stream
#           -> gpurun_out/<tag>_c<C>_kernel_summary_serial.csv   (per-kernel calls / avg / ms per step)
#   pmc:    two SEPARATE counter passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains)
#           -> gpurun_out/<tag>_c<C>_pmc_{FETCH,WRITE}_SIZE.csv and the per-family fold gpurun_out/<tag>_pmc_traffic.json
#   sq:     one SQ counter pass (matrix-pipe busy cycles, VALU / MFMA instruction counts, wave wait cycles)
#           -> gpurun_out/<tag>_c<C>_pmc_sq.csv and the per-kernel occupancy table gpurun_out/<tag>_c<C>_mfma_busy.txt
# Copy what should be judged from gpurun_out/ into profiles/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=$1; c=$2; what=${3:-all}
export TMPDIR=/tmp RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
mkdir -p gpurun_out
# the first-call kernel choice (csrc/conv_igemm.hip, choose_impl) is measured in an un-profiled run and read back by the profiled
# ones: no measuring launches inside the traces, the same kernels in every pass
export RG_CONV_TUNE_CACHE=/tmp/rg_tune_$c.txt
[ -f $RG_CONV_TUNE_CACHE ] || python3 bench.py --config $c --no-others --no-cpu-baseline --steps 2 --warmup 2 --profile-steps 0 > /dev/null 2>&1
if [ "$what" = kernel ] || [ "$what" = all ]; then
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${tag}_c${c}_bench_under_rocprof.json 2> gpurun_out/${tag}_c${c}_prof.err || exit 1
  # 10 timed + 3 warm-up + 1 serial warm-up + 3 profiled steps + 1 preparation step (4 capture steps for the configurations with graphed networks)
  nsteps=18; case $c in 4a|5) nsteps=21;; esac
  python tools/prof_summary.py /tmp/prof_$c gpurun_out/${tag}_c${c}_kernel_summary_serial.csv --steps $nsteps || exit 1
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  [ -f profiles/${tag}_pmc_traffic.json ] && [ ! -f gpurun_out/${tag}_pmc_traffic.json ] && cp profiles/${tag}_pmc_traffic.json gpurun_out/
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/${tag}_c${c}_pmc_$ctr.err || exit 1
    python tools/prof_summary.py /tmp/pmc_${c}_$ctr gpurun_out/${tag}_c${c}_pmc_$ctr.csv || exit 1
  done
  # steps per pass: 1 warm-up + 1 timed + 1 preparation step (4 capture steps for the configurations with graphed networks)
  psteps=3; case $c in 4a|5) psteps=6;; esac
  python tools/pmc_traffic.py gpurun_out/${tag}_c${c}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_c${c}_pmc_WRITE_SIZE.csv $psteps gpurun_out/${tag}_pmc_traffic.json $c
fi
if [ "$what" = sq ] || [ "$what" = all ]; then
  rm -rf /tmp/pmc_${c}_sq
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_${c}_sq -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/${tag}_c${c}_pmc_sq.err || exit 1
  python tools/prof_summary.py /tmp/pmc_${c}_sq gpurun_out/${tag}_c${c}_pmc_sq.csv || exit 1
  python tools/pmc_busy.py gpurun_out/${tag}_c${c}_pmc_sq.csv gpurun_out/${tag}_c${c}_mfma_busy.txt 20
fi
