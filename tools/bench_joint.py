#!/usr/bin/env python
"""Throughput of the other step configurations of BASELINE.json on one MI355X (bench.py stays on the headline config 2):
  cc      — config 3: cluster-contrast step, ResNet-50 (layer4 stride 1) + GeM, B crops of 256x128, K = 2048 clusters
  joint4a — config 4a: joint ReID + GAN step as committed (AEModel 'Pose' generator at 128x64 + spectral-norm D)
  joint4b — config 4b: the same trainer step with FDGANModel in the GAN role (B crops = B/2 FD-GAN pairs)
usage: python tools/bench_joint.py [cc|joint4a] [--batch 32] [--steps 10] [--warmup 3]
Prints one JSON line per run (images/s, ms/step, per-family kernel time from the library's HIP-event profiler)."""
from __future__ import absolute_import, print_function

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["cc", "joint4a", "joint4b"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--cprofile", action="store_true", help="print the host-side hot spots of 3 steps")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from rg_hip import ops, optim as roptim
    import clustercontrast.models as M
    from clustercontrast.models.cm import ClusterMemory
    from clustercontrast.trainers import ClusterContrastTrainer, ClusterContrastWithGANTrainer

    torch.manual_seed(0)
    B, K = args.batch, 2048
    enc = M.create('resnet50', pretrained=False, pooling_type="gem").to(dev).train()
    mem = ClusterMemory(enc.num_features, K, temp=0.05, momentum=0.1).to(dev)
    mem.features = F.normalize(torch.randn(K, enc.num_features, device=dev), dim=1)
    opt = roptim.Adam([{"params": [p]} for p in enc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
    g = torch.Generator(device=dev).manual_seed(1)
    imgs = torch.randn(B, 3, 256, 128, generator=g, device=dev)
    labels = torch.randint(0, K, (max(B // 16, 1),), generator=g, device=dev).repeat_interleave(min(16, B))[:B]
    indexes = torch.arange(B, device=dev)

    if args.workload == "cc":
        trainer = ClusterContrastTrainer(enc, mem)

        def step():
            return trainer.step(imgs, labels, opt)
        gflop_per_crop = 24.34
    elif args.workload == "joint4b":
        sys.path.insert(0, REPO)
        import bench as HB
        from fdgan.model import FDGANModel
        from fdgan.adaptor import FDGANAdaptor
        model = FDGANModel(HB.fdgan_opt(batch_size=B // 2))
        model.reset_model_status()
        gan = FDGANAdaptor(model)
        pair = HB.synth_inputs(B // 2, dev, seed=1234)
        trainer = ClusterContrastWithGANTrainer(enc, GAN=gan, memory=mem)

        def step():
            gan.set_input(pair)
            return trainer.joint_step(imgs, labels, indexes, opt)
        gflop_per_crop = 160.0
    else:
        from dual_gan.models.models import create_model
        gopt = argparse.Namespace(
            model="AE", gan_train=True, checkpoints_dir="/tmp/rg_ckpt", name="b", load_pretrain="", model_gen="Pose",
            num_feats=256, layers_g=3, image_nc=3, pose_nc=18, norm="instance", use_spect_g=False, use_spect_d=True,
            use_coord=False, num_blocks=3, nhead=2, num_CABs=2, num_TTBs=2, dis_layers=3, init_type="orthogonal",
            verbose=False, pool_size=0, gan_lr=2e-4, gan_mode="lsgan", no_vgg_loss=True, beta1=0.5, ratio_g2d=0.1,
            lambda_rec=2.0, lambda_g=5.0, gan_lr_policy="lambda", iter_start=0, niter=100, niter_decay=100,
            continue_train=False, which_epoch="latest", bipath_gan=False, use_adp=False)
        gan = create_model(gopt)
        xs = (torch.rand(B, 3, 128, 64, generator=g, device=dev) - 0.5) / 0.5
        ys = torch.arange(128, device=dev, dtype=torch.float32).view(1, 1, 128, 1)
        xx = torch.arange(64, device=dev, dtype=torch.float32).view(1, 1, 1, 64)
        cy = torch.randint(0, 128, (B, 18, 1, 1), generator=g, device=dev).float()
        cx = torch.randint(0, 64, (B, 18, 1, 1), generator=g, device=dev).float()
        ps = torch.exp(-((ys - cy) ** 2 + (xx - cx) ** 2) / 72.0)
        trainer = ClusterContrastWithGANTrainer(enc, GAN=gan, memory=mem)
        gan_in = {"Xs": xs, "Ps": ps}

        def step():
            gan.set_input(gan_in)
            return trainer.joint_step(imgs, labels, indexes, opt)
        gflop_per_crop = 32.0

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.cprofile:
        import cProfile, pstats, io
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
        torch.cuda.synchronize()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32)
        print(st.getvalue()[:7000], file=sys.stderr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_host = (time.perf_counter() - t0) / args.steps        # host enqueue time (no sync inside the loop)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.profile_reset()
    ops.profile_enable(True)
    for _ in range(args.profile_steps):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(False)
    fam = ops.profile_collect()
    out = {"workload": args.workload, "batch_crops": B, "ms_per_step": round(1e3 * dt, 3), "host_enqueue_ms_per_step": round(1e3 * t_host, 3), "images_per_s": round(B / dt, 1),
           "loss": round(float(loss), 5), "gflop_per_crop_algorithmic": gflop_per_crop,
           "step_tflops": round(gflop_per_crop * B / dt / 1e3, 2),
           "by_family": {k: {"ms_per_step": round(v["ms"] / args.profile_steps, 3), "launches": v["calls"] // args.profile_steps,
                             "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2) if v["flops"] else None}
                         for k, v in fam.items() if v["calls"]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
