set -e
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "splitk" > gpurun_out/r04s_test_splitk.log 2>&1 || { tail -30 gpurun_out/r04s_test_splitk.log; exit 1; }
tail -3 gpurun_out/r04s_test_splitk.log
timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r04s_bench_finishkernel.json 2> gpurun_out/r04s_bench_finishkernel.err
RG_SPLITK_INKERNEL=1 timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r04s_bench_inkernel.json 2> gpurun_out/r04s_bench_inkernel.err
timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r04s_bench_finishkernel2.json 2> gpurun_out/r04s_bench_finishkernel2.err
RG_SPLITK_INKERNEL=1 timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r04s_bench_inkernel2.json 2> gpurun_out/r04s_bench_inkernel2.err
python - <<'P'
import json
for n in ("finishkernel","inkernel","finishkernel2","inkernel2"):
    d=json.loads(open("gpurun_out/r04s_bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"])
P
timeout -k 10 240 python tools/bench_conv.py 32 resnet > gpurun_out/r04s_conv_resnet_finishkernel.txt 2>&1
RG_SPLITK_INKERNEL=1 timeout -k 10 240 python tools/bench_conv.py 32 resnet > gpurun_out/r04s_conv_resnet_inkernel.txt 2>&1
tail -n 3 gpurun_out/r04s_conv_resnet_finishkernel.txt; tail -n 3 gpurun_out/r04s_conv_resnet_inkernel.txt
