set -o pipefail
cd "$GRAFT_REPO_ROOT"
for c in 2 3 4a 4b 5; do
  tag=${1:-r04}
  tools/gpu_profile.sh $tag $c all > gpurun_out/${tag}_prof_$c.log 2>&1 || { echo "profile $c failed"; tail -5 gpurun_out/${tag}_prof_$c.log; exit 1; }
  echo "profiled $c"
done
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err && echo bench ok
python tools/bench_conv.py 32 resnet > gpurun_out/${tag}_conv_shapes_resnet_n32.txt 2>&1 && python tools/bench_conv.py 32 gan > gpurun_out/${tag}_conv_shapes_gan_n32.txt 2>&1
