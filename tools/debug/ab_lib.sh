# same-box A/B of two builds of the library: tools/debug/ab_lib.sh <other .so>   (RG_LIB_PATH selects the build)
set -e
mkdir -p gpurun_out
other=$1
for i in 1 2 3; do
RG_LIB_PATH=$other timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/ab_lib_other$i.json 2> gpurun_out/ab_lib_other$i.err
timeout -k 10 200 python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/ab_lib_head$i.json 2> gpurun_out/ab_lib_head$i.err
done
python - <<'P'
import json
for n in ("other1","head1","other2","head2","other3","head3"):
    d=json.loads(open("gpurun_out/ab_lib_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_step"])
P
