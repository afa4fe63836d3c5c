#!/bin/bash
# Regenerates the per-round profile set on a GPU box: tools/gpu_profile_all.sh <tag> [configs...]   (default: r04, all five)
# The five configurations do not fit one 20-minute gpurun call: run e.g. `gpu_profile_all.sh r04 2 3`, copy
# gpurun_out/<tag>_pmc_traffic.json into profiles/, then `gpu_profile_all.sh r04 4a 4b 5 final` (`final`: the default bench line —
# with the fresh traffic file copied into profiles/ first — and the per-layer conv tables).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04}; shift
cfgs="$*"; [ -z "$cfgs" ] && cfgs="2 3 4a 4b 5 final"
for c in $cfgs; do
  if [ "$c" = final ]; then
    [ -f gpurun_out/${tag}_pmc_traffic.json ] && cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
    python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err && echo bench ok
    python tools/bench_conv.py 32 resnet > gpurun_out/${tag}_conv_shapes_resnet_n32.txt 2>&1 && python tools/bench_conv.py 32 gan > gpurun_out/${tag}_conv_shapes_gan_n32.txt 2>&1 && echo shapes ok
    continue
  fi
  tools/gpu_profile.sh $tag $c all > gpurun_out/${tag}_prof_$c.log 2>&1 || { echo "profile $c failed"; tail -5 gpurun_out/${tag}_prof_$c.log; exit 1; }
  echo "profiled $c"
done
