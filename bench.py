#!/usr/bin/env python
"""bench.py — train-step images/sec of the FD-GAN step on MI355X (BASELINE.json configs[1]).

One "step" = FDGANModel.set_input + optimize_parameters (FD/fdgan/model.py:127-147,216-229) on one batch of
synthetic, HBM-resident input: batch_size 16 pairs = 32 crops of 256x128 + 18-channel pose maps per GPU,
stage-2 wiring (E, G, D_id, D_pd all trainable; E / D_id BatchNorm in eval mode), reference defaults
(drop 0.2, lambda 1/1/1), fp32 end to end, random-init weights.  Weak scaling: every rank runs that batch,
gradients are all-reduced over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded by the library around every
conv implicit-GEMM launch (on the launch stream) during `--profile-steps` extra steps of the same workload
that follow the timed region (so the per-launch events do not perturb `value`); `cpu_baseline` times the CPU
oracle restatement of the reference step on a bounded sample (rank 0, N=1 only).
"""
from __future__ import absolute_import, print_function

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
CROPS_PER_GPU = 32                # batch_size 16 pairs
# algorithmic conv/linear FLOPs of the step per crop (BASELINE.md §2 / SURVEY §8d config 2): fwd 51.71 + necessary bwd 84.02
NECESSARY_GFLOP_PER_CROP = 135.7


def fdgan_opt(**kw):
    d = dict(stage=2, checkpoints="/tmp/rg_bench_ckpt", name="bench", norm="batch", drop=0.2, connect_layers=0,
             fuse_mode="cat", pose_feature_size=128, noise_feature_size=256, arch="resnet50", lr=0.001, niter=50,
             niter_decay=50, lambda_recon=1.0, lambda_veri=1.0, lambda_sp=1.0, smooth_label=False, random_init=True,
             quiet=True, batch_size=16)
    d.update(kw)
    return argparse.Namespace(**d)


def synth_inputs(b, dev, seed):
    """Device-resident synthetic batch in the dataloader's format (two dicts, FD/reid/utils/data/preprocessor.py)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)

    def imgs(n):
        return (torch.rand(n, 3, 256, 128, generator=g, device=dev) - mean) / std

    def poses(n):
        ys = torch.arange(256, device=dev, dtype=torch.float32).view(1, 1, 256, 1)
        xs = torch.arange(128, device=dev, dtype=torch.float32).view(1, 1, 1, 128)
        cy = torch.randint(0, 256, (n, 18, 1, 1), generator=g, device=dev).float()
        cx = torch.randint(0, 128, (n, 18, 1, 1), generator=g, device=dev).float()
        present = (torch.rand(n, 18, 1, 1, generator=g, device=dev) >= 0.1).float()
        return torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 50.0) * present
    pid1 = torch.arange(b, device=dev)
    same = (torch.arange(b, device=dev) % 4 == 0)           # RandomPairSampler(neg_pos_ratio=3) pattern
    pid2 = torch.where(same, pid1, pid1 + 100000)
    in1 = dict(pid=pid1, origin=imgs(b), target=imgs(b), posemap=poses(b))
    in2 = dict(pid=pid2, origin=imgs(b), target=imgs(b), posemap=poses(b))
    return in1, in2


def host_cores():
    """CPU threads this process may really use: affinity mask, then the cgroup CPU quota (the GPU boxes expose
    256 logical CPUs but grant a 16-CPU share; oversubscribing oneDNN's thread pool makes it crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("RG_CPU_THREADS", "16"))))


def cpu_baseline(sample_pairs=8):
    """The reference step on the host cores: oracle restatement (checked against the reference's modules by
    tests/golden) at a bounded sample so the default run stays within minutes."""
    from oracle import ref_torch as O
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    oE = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 2))
    oDi = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 1))
    oG = O.OPoseGenerator(128, 2048, 256, dropout=0.2)
    oG.apply(O.o_weights_init_normal)
    oDp = O.OPatchDiscriminator(21)
    oDp.apply(O.o_weights_init_normal)
    step = O.OFDGANStep(oE, oG, oDi, oDp, lr=0.001, stage=2)
    batch = O.synth_fdgan_batch(sample_pairs, seed=1)
    step.step(*batch)                                       # warm-up (allocator, oneDNN primitive caches)
    log("cpu baseline warm-up step done")
    t0 = time.time()
    n = 3
    for _ in range(n):
        step.step(*batch)
        log("cpu baseline step done")
    dt = (time.time() - t0) / n
    return {"value": 2 * sample_pairs / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "FD-GAN step (oracle/ref_torch.OFDGANStep, torch CPU fp32) at batch_size %d pairs = %d crops, "
                      "mean of %d steps after 1 warm-up (%.1f s/step)" % (sample_pairs, 2 * sample_pairs, n, dt)}


def log(*a):
    print("[bench %s]" % time.strftime("%H:%M:%S"), *a, file=sys.stderr, flush=True)


def pmc_traffic(launches_per_step):
    """HBM bytes per conv launch from the committed PMC passes (profiles/r01_pmc_traffic.json; produced on the GPU
    box by tools/prof_summary.py + tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this script).  PMC counters cannot be read from inside the process, so the figure is not live; None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            fam = json.load(f)["families"]["conv"]
    except (OSError, KeyError, ValueError):
        return None, "no PMC pass committed"
    per_step = fam["read_bytes_per_step"] + fam["write_bytes_per_step"]
    return round(per_step / max(launches_per_step, 1)), ("profiles/r01_pmc_traffic.json: %.1f GB read + %.1f GB written per "
                                                         "step by the conv family incl. split-K finish/reduce and KRSC "
                                                         "repack kernels" % (fam["read_bytes_per_step"] / 1e9,
                                                                             fam["write_bytes_per_step"] / 1e9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("RG_FORCE_REDUCE") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        raise SystemExit("bench.py --gpus %d must be launched through torch.distributed.run (one rank per GPU)" % args.gpus)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from fdgan.model import FDGANModel
    from rg_hip import ops

    torch.manual_seed(1234)                                  # identical replicas on every rank
    opt = fdgan_opt()
    model = FDGANModel(opt)
    model.reset_model_status()
    data = synth_inputs(opt.batch_size, dev, seed=100 + rank)
    torch.manual_seed(99 + rank)                             # noise z / dropout seeds differ per rank

    def one_step():
        model.set_input(data)
        model.optimize_parameters()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log("model built; warm-up %d steps" % args.warmup)
    for i in range(args.warmup):
        one_step()
        if rank == 0:
            torch.cuda.synchronize()
            log("warm-up step %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    losses = model.get_current_errors()
    if rank == 0:
        log("timed %d steps: %.1f ms/step" % (args.steps, 1e3 * elapsed / args.steps))

    # ---- roofline: per-launch HIP events on the conv implicit-GEMM kernels ------------------------------
    roof = None
    fam = None
    if rank == 0 and args.profile_steps > 0:
        # per-launch durations are taken with the kernels launched back to back on ONE stream: in the timed region the
        # weight-gradient kernels and the D_pd passes run concurrently on side streams (rg_hip.ops.side_*,
        # FDGANModel._aux_stream), which lengthens each overlapped launch without saying anything about the kernel
        ops.side_enable(False)
        os.environ["RG_AUX_STREAM"] = "0"
        one_step()
        torch.cuda.synchronize()
        ops.profile_reset()
        ops.profile_enable(True)
        for _ in range(args.profile_steps):
            one_step()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        ops.side_enable(True)
        os.environ["RG_AUX_STREAM"] = "1"
        fam = ops.profile_collect()
        conv = [fam[k] for k in ("conv_fwd", "conv_dgrad", "conv_wgrad")]
        ms = sum(f["ms"] for f in conv)
        fl = sum(f["flops"] for f in conv)
        calls = sum(f["calls"] for f in conv)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic(calls // args.profile_steps)
        roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "bytes per launch", "traffic_source": traffic_note,
                "kernel": "conv implicit-GEMM family (conv_fwd/dgrad/wgrad_kernel, v_mfma_f32_32x32x2_f32)",
                "launches_per_step": calls // args.profile_steps,
                "avg_launch_us": round(1e3 * ms / max(calls, 1), 2),
                "algorithmic_tflop_per_step": round(fl / args.profile_steps / 1e12, 4),
                "kernel_ms_per_step": round(ms / args.profile_steps, 3),
                "measured_over": "%d profiled steps after the timed region, single-stream launch order (the timed region "
                                 "overlaps wgrad / D_pd kernels on side streams)" % args.profile_steps,
                "by_family": {k: {"ms_per_step": round(v["ms"] / args.profile_steps, 3),
                                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2) if v["flops"] else None,
                                  "gbps_algorithmic": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None,
                                  "launches": v["calls"] // args.profile_steps} for k, v in fam.items() if v["calls"]}}
    if use_dist:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle step on %d host threads) ..." % host_cores())
        cpu = cpu_baseline()
        log("cpu baseline done: %.3f images/s" % cpu["value"])

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * CROPS_PER_GPU * args.steps / elapsed
        out = {
            "metric": "train-step images/sec, 256\u00d7128 ReID batch, 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (random-init weights, seeded on-device 256x128 crops + 18-ch pose maps)",
            "config": {"workload": "FD-GAN G + D_id/D_pd fwd/bwd step (FDGANModel.optimize_parameters), batch 16 pairs = "
                                   "32 crops of 256x128 + 18-ch pose map per GPU, stage 2, drop 0.2",
                       "global_batch_crops": world * CROPS_PER_GPU, "parallelism": "dp%d" % world},
            "step_tflops_necessary": round(NECESSARY_GFLOP_PER_CROP * CROPS_PER_GPU / 1e3, 3),
            "step_frac_of_f32_mfma_peak": round(NECESSARY_GFLOP_PER_CROP * CROPS_PER_GPU / 1e3 / (ms_step * 1e-3)
                                                / F32_MFMA_PEAK_TFLOPS, 4),
            "losses": {k: round(v, 5) for k, v in losses.items()},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
