# the default bench under torch.distributed.run with ONE rank and the gradient reducers forced on (RG_FORCE_REDUCE=1): every reducer
# path (RCCL init, broadcasts, stage-bucketed all-reduce from inside the trunk backward) on hardware -> gpurun_out/<tag>_bench_rccl_single_rank.json
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=${1:-r04}
mkdir -p gpurun_out
RG_FORCE_REDUCE=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_rccl_single_rank.json 2> gpurun_out/${tag}_bench_rccl_single_rank.err || { tail -n 20 gpurun_out/${tag}_bench_rccl_single_rank.err; exit 1; }
python - <<P
import json
d=json.loads(open("gpurun_out/${tag}_bench_rccl_single_rank.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:v.get("ms_per_step") for k,v in d.get("other_configs",{}).items()})
P
