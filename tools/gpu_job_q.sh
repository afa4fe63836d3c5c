set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp RG_WGRAD_STREAM=0 RG_AUX_STREAM=0
mkdir -p gpurun_out
cp profiles/r02_pmc_traffic.json gpurun_out/r02_pmc_traffic.json
c=4a
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${c}_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -- python3 bench.py --config $c --no-others --no-cpu-baseline --steps 1 --warmup 1 --profile-steps 0 > /dev/null 2> gpurun_out/r02q_pmc_${c}_$ctr.err
  echo "pmc $c $ctr rc=$?"
  python tools/prof_summary.py /tmp/pmc_${c}_$ctr gpurun_out/r02_c${c}_pmc_$ctr.csv
done
python tools/pmc_traffic.py gpurun_out/r02_c${c}_pmc_FETCH_SIZE.csv gpurun_out/r02_c${c}_pmc_WRITE_SIZE.csv 2 gpurun_out/r02_pmc_traffic.json $c
