# same-box A/B: plain step vs the gradient reducers active on one rank (RG_FORCE_REDUCE=1) against the number of hardware queues the
# HIP streams are mapped onto (GPU_MAX_HW_QUEUES, default 4).  RG_AB_EXTRA: extra bench.py flags, RG_AB_QUEUES: the queue counts.
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
B="python bench.py --no-others --no-cpu-baseline --steps 30 --warmup 5 ${RG_AB_EXTRA}"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("host_enqueue_ms_per_step"))'
for i in 1 2; do
for q in ${RG_AB_QUEUES:-2 3 4 8}; do
echo plain-q$q; GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | python -c "$P"
echo rccl-q$q; GPU_MAX_HW_QUEUES=$q RG_FORCE_REDUCE=1 $B 2>/dev/null | python -c "$P"
echo hp-rccl-q$q; RG_NCCL_HIGH_PRIO=1 GPU_MAX_HW_QUEUES=$q RG_FORCE_REDUCE=1 $B 2>/dev/null | python -c "$P"
done
done
echo "D_pd's weight gradients on a side stream of the auxiliary stream (a fourth compute stream: the state before):"
for q in ${RG_AB_QUEUES:-2 3 4 8}; do
echo aux-side-plain-q$q; RG_AUX_SIDE=1 GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | python -c "$P"
echo aux-side-rccl-q$q; RG_AUX_SIDE=1 GPU_MAX_HW_QUEUES=$q RG_FORCE_REDUCE=1 $B 2>/dev/null | python -c "$P"
done
echo "unprobed streams:"
for q in ${RG_AB_QUEUES:-2 3 4 8}; do
echo unprobed-rccl-q$q; RG_STREAM_PROBE=0 RG_AUX_SIDE=1 GPU_MAX_HW_QUEUES=$q RG_FORCE_REDUCE=1 $B 2>/dev/null | python -c "$P"
done
