#!/usr/bin/env python
"""bench.py — train-step images/sec of the ReID-GAN training steps on MI355X (BASELINE.json).

    python bench.py [--config 2|3|4a|4b|5] [--gpus N] [--steps K] [--warmup W]

Default (`--config 2`, BASELINE.json configs[1], the configuration the metric is quoted on): one "step" =
FDGANModel.set_input + optimize_parameters (FD/fdgan/model.py:127-147,216-229) on one batch of synthetic, HBM-resident
input: batch_size 16 pairs = 32 crops of 256x128 + 18-channel pose maps per GPU, stage-2 wiring, reference defaults
(drop 0.2), fp32 end to end, random-init weights.  The other BASELINE configurations are selected with --config:

    3   cluster-contrast step, ResNet-50 (layer4 stride 1) + GeM + ClusterMemory(2048 x 2048), 64 crops   (trainers.py:229-249)
    4a  joint ReID + dual_gan AEModel('Pose') step as committed, 32 crops + GAN at 128x64           (trainers_b.py:617-774)
    4b  joint step with FDGANModel in the GAN role (north-star wording), 32 crops = 16 pairs          (fdgan/adaptor.py)
    5   dual_gan DPTNModel step (two generator branches), fp8 MFMA convolutions, 64 crops at 128x64    (DPTN_model.py:216-225)

and, unless --no-others is given, the default single-GPU run also measures each of them briefly and reports them under
`other_configs` of the same JSON line, so one driver-run line covers every configuration (multi-GPU runs: the headline
only, unless --others asks for the DDP configurations 4a / 4b / 5 behind it).

Multi-GPU: weak scaling, one rank per GPU, gradients all-reduced over RCCL.  `--gpus N` without a torch.distributed
environment starts the N ranks itself (a `python -m torch.distributed.run` child process, created before this process
makes any GPU call) and relays rank 0's line; under `torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded by the library around every conv
implicit-GEMM launch (on the launch stream) during `--profile-steps` extra steps that follow the timed region, run by
EVERY rank (their collectives must match) with the kernels launched back to back on one stream; `cpu_baseline` times the
CPU oracle restatement of the same step on the GPU box's host cores (rank 0, N=1 only).
"""
from __future__ import absolute_import, print_function

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 peak (same guide).  The fp32 convolutions evaluate every fp32 product as SIX bf16 MFMA
                                  # products of exactly split operands (csrc/conv_igemm.hip), so the matrix pipe bounds them at
                                  # 2500 / 6 = 416.7 fp32-equivalent TFLOP/s: THAT is `roofline.peak` (the roof of the
                                  # instruction actually issued); the figure against the fp32-MFMA peak that
                                  # BASELINE.json's metric names is kept beside it as `frac_vs_fp32_mfma`
SPLIT_BF16_PEAK_TFLOPS = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
F8_MFMA_PEAK_TFLOPS = 5000.0      # dense fp8: v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 / e5m2 operands runs at twice the
                                  # bf16 rate (same guide, "Matrix cores"); the non-scaled 32x32x16 fp8 forms only reach 2.5 PF
METRIC = "train-step images/sec, 256×128 ReID batch, 1/2/4/8 MI355X"


def log(*a):
    print("[bench %s]" % time.strftime("%H:%M:%S"), *a, file=sys.stderr, flush=True)


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


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs, generated on the device
# ---------------------------------------------------------------------------------------------------------------------
def fdgan_opt(**kw):
    d = dict(stage=2, checkpoints="/tmp/rg_bench_ckpt", name="bench", norm="batch", drop=0.2, connect_layers=0,
             fuse_mode="cat", pose_feature_size=128, noise_feature_size=256, arch="resnet50", lr=0.001, niter=50,
             niter_decay=50, lambda_recon=1.0, lambda_veri=1.0, lambda_sp=1.0, smooth_label=False, random_init=True,
             quiet=True, batch_size=16)
    d.update(kw)
    return argparse.Namespace(**d)


def synth_inputs(b, dev, seed):
    """Device-resident synthetic FD-GAN batch in the dataloader's format (two dicts, FD/reid/utils/data/preprocessor.py)."""
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


def synth_reid(B, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    imgs = torch.randn(B, 3, 256, 128, generator=g, device=dev)
    labels = torch.randint(0, K, (max(B // 16, 1),), generator=g, device=dev).repeat_interleave(min(16, B))[:B]
    return imgs, labels, torch.arange(B, device=dev)


def synth_dualgan(B, dev, seed, with_target=False):
    """Xs in [-1, 1] and 18-channel Gaussian pose maps at 128x64 (CC/.../preprocessor.py:124-137, pose_utils.py:52-70)."""
    g = torch.Generator(device=dev).manual_seed(seed)

    def img():
        return (torch.rand(B, 3, 128, 64, generator=g, device=dev) - 0.5) / 0.5

    def pose():
        ys = torch.arange(128, device=dev, dtype=torch.float32).view(1, 1, 128, 1)
        xx = torch.arange(64, device=dev, dtype=torch.float32).view(1, 1, 1, 64)
        cy = torch.randint(0, 128, (B, 18, 1, 1), generator=g, device=dev).float()
        cx = torch.randint(0, 64, (B, 18, 1, 1), generator=g, device=dev).float()
        return torch.exp(-((ys - cy) ** 2 + (xx - cx) ** 2) / 72.0)
    d = {"Xs": img(), "Ps": pose()}
    if with_target:
        d.update({"Xt": img(), "Pt": pose()})
    return d


def dualgan_opt(**kw):
    d = dict(model="AE", gan_train=True, checkpoints_dir="/tmp/rg_ckpt", name="b", load_pretrain="", model_gen="Pose",
             num_feats=256, layers_g=3, image_nc=3, pose_nc=18, norm="instance", use_spect_g=False, use_spect_d=True,
             use_coord=False, num_blocks=3, nhead=2, num_CABs=2, num_TTBs=2, dis_layers=3, init_type="orthogonal",
             verbose=False, pool_size=0, gan_lr=2e-4, gan_mode="lsgan", no_vgg_loss=True, beta1=0.5, ratio_g2d=0.1,
             lambda_rec=2.0, lambda_g=5.0, gan_lr_policy="lambda", iter_start=0, niter=100, niter_decay=100,
             continue_train=False, which_epoch="latest", bipath_gan=False, use_adp=False)
    d.update(kw)
    return argparse.Namespace(**d)


# ---------------------------------------------------------------------------------------------------------------------
# workloads: build(dev, rank) -> step(); every rank runs the same per-GPU batch (weak scaling)
# ---------------------------------------------------------------------------------------------------------------------
class Workload(object):
    key = "?"
    crops = 32
    gflop_per_crop = 0.0           # algorithmic conv / linear FLOPs per crop and step (BASELINE.md §2 / SURVEY §8d)
    dtype = "f32"
    peak = SPLIT_BF16_PEAK_TFLOPS     # 2500 / 6: six v_mfma_f32_32x32x16_bf16 products per fp32 product
    conv_families = ("conv_fwd", "conv_dgrad", "conv_wgrad")
    kernel_note = ("conv implicit-GEMM family (conv_fwd/dgrad/wgrad_kernel and their bf16-plane twins conv_*_pl_kernel — the faster of "
                   "the two is measured once per geometry —, conv3x3_halo_kernel): fp32 in / fp32 out, every operand split exactly "
                   "into 3 bf16 pieces, 6 of the 9 partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped "
                   "terms <= 2^-23 per product: fp32-grade, tests at the round-2 2e-5 bounds); peak = 2500 / 6 at the nominal 2.4 GHz "
                   "(the chip holds 1.4-1.6 GHz under this load: profiles/r04_micro_gemm_pl.txt, r04_mfma_shape.txt)")
    mfma_products = 6                 # bf16 matrix products issued per algorithmic fp32 product
    launch_note = "eager (one hipLaunchKernel per kernel, issued through the C ABI)"
    capture_steps = 0
    describe = ""

    def build(self, dev, rank):
        raise NotImplementedError

    def step(self):
        raise NotImplementedError

    def losses(self):
        return {}

    def serial(self, on):
        """launch order for the profiled steps: every kernel back to back on one stream"""
        from rg_hip import ops
        ops.side_enable(not on)
        os.environ["RG_AUX_STREAM"] = "0" if on else "1"

    def cpu_baseline(self):
        return None


class FDGANStep(Workload):
    key = "2"
    crops = 32
    gflop_per_crop = 135.7         # fwd 51.71 + necessary bwd 84.02
    describe = ("FD-GAN G + D_id/D_pd fwd/bwd step (FDGANModel.optimize_parameters), batch 16 pairs = 32 crops of 256x128 "
                "+ 18-ch pose map per GPU, stage 2, drop 0.2")

    def build(self, dev, rank):
        from fdgan.model import FDGANModel
        torch.manual_seed(1234)
        opt = fdgan_opt()
        self.model = FDGANModel(opt)
        # Conditioned random weights for the two ResNet-50 trunks (the reference starts them from ImageNet / stage-1 checkpoints; with
        # raw random filters behind frozen unit BatchNorm statistics the verification logits hit BCE's +-100 clamp on step 1 and the
        # D_id -> G gradient of the timed step degenerates): the conditioning of tests/test_modules_gpu.py.  Same kernels, same FLOPs.
        g = torch.Generator().manual_seed(4242)
        with torch.no_grad():
            for net in (self.model.net_E.module, self.model.net_Di.module):
                cw = net.embed_model.classifier.weight
                cw.copy_(torch.randn(cw.shape, generator=g) * 0.05)
                for m in net.modules():
                    if hasattr(m, "running_mean") and m.running_mean is not None and hasattr(m, "weight"):
                        m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.05)
                        m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.4 + 0.8)
                        m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.2 + 0.4)
        self.model.reset_model_status()
        self.data = synth_inputs(opt.batch_size, dev, seed=100 + rank)
        torch.manual_seed(99 + rank)                             # noise z / dropout seeds differ per rank

    def step(self):
        self.model.set_input(self.data)
        self.model.optimize_parameters()

    def losses(self):
        return self.model.get_current_errors()

    def cpu_baseline(self, pairs=16, warm=2, timed=5):
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
        batch = O.synth_fdgan_batch(pairs, seed=1)
        dt = _time_cpu(lambda: step.step(*batch), warm, timed)
        return {"value": round(2 * pairs / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": "FD-GAN step (oracle/ref_torch.OFDGANStep, torch CPU fp32) at batch_size %d pairs = %d crops, "
                          "mean of %d steps after %d warm-up (%.1f s/step)" % (pairs, 2 * pairs, timed, warm, dt)}


def _time_cpu(fn, warm, timed):
    for i in range(warm):
        fn()
        log("cpu baseline warm-up step %d done" % i)
    t0 = time.time()
    for i in range(timed):
        fn()
        log("cpu baseline step %d done" % i)
    return (time.time() - t0) / timed


def _cc_parts(dev, B, K=2048):
    from rg_hip import optim as roptim
    import clustercontrast.models as M
    from clustercontrast.models.cm import ClusterMemory
    torch.manual_seed(0)
    enc = M.create('resnet50', pretrained=False, pooling_type="gem").to(dev).train()
    mem = ClusterMemory(enc.num_features, K, temp=0.05, momentum=0.1).to(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    mem.features = F.normalize(torch.randn(K, enc.num_features, generator=g, device=dev), dim=1)
    opt = roptim.Adam([{"params": [p]} for p in enc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
    return enc, mem, opt


class CCStep(Workload):
    key = "3"
    crops = 64
    gflop_per_crop = 24.34
    describe = ("cluster-contrast step (ClusterContrastTrainer.step): ResNet-50 (layer4 stride 1, train-mode BN) + GeM + "
                "ClusterMemory(2048 centroids x 2048), 64 crops of 256x128 per GPU, Adam")

    def build(self, dev, rank):
        from clustercontrast.trainers import ClusterContrastTrainer
        self.enc, self.mem, self.opt = _cc_parts(dev, self.crops)
        self.imgs, self.labels, self.indexes = synth_reid(self.crops, 2048, dev, 1 + rank)
        self.trainer = ClusterContrastTrainer(self.enc, self.mem)
        self.loss = None

    def step(self):
        self.loss = self.trainer.step(self.imgs, self.labels, self.opt)

    def losses(self):
        return {"loss": float(self.loss)}

    def cpu_baseline(self, warm=2, timed=5):
        from oracle import ref_torch as O
        cores = host_cores()
        torch.set_num_threads(cores)
        torch.manual_seed(0)
        B, K = self.crops, 2048
        enc = O.OCCResNet(50, pooling_type="gem")
        enc.train()
        mem = O.OClusterMemory(2048, K, temp=0.05, momentum=0.1)
        mem.features = F.normalize(torch.randn(K, 2048), dim=1)
        opt = torch.optim.Adam([{"params": [p]} for p in enc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
        imgs = torch.randn(B, 3, 256, 128)
        labels = torch.randint(0, K, (B // 16,)).repeat_interleave(16)
        dt = _time_cpu(lambda: O.o_cc_step(enc, mem, opt, imgs, labels), warm, timed)
        return {"value": round(B / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": "cluster-contrast step (oracle/ref_torch.o_cc_step, torch CPU fp32) at %d crops, mean of %d steps "
                          "after %d warm-up (%.1f s/step)" % (B, timed, warm, dt)}


class Joint4a(Workload):
    key = "4a"
    crops = 32
    gflop_per_crop = 32.0
    launch_note = ("the GAN networks' forward / backward programs replayed as single-stream hipGraphs captured per network "
                   "(rg_hip/netgraph.py; RG_NET_GRAPHS=0 = eager); everything else eager, one hipLaunchKernel per kernel through the C ABI")
    capture_steps = 4
    describe = ("joint ReID + GAN step as committed (ClusterContrastWithGANTrainer.joint_step, trainers_b.py:617-774): "
                "cluster-contrast encoder + dual_gan AEModel('Pose', layers 3) at 128x64 + spectral-norm D, 32 crops per GPU")

    def build(self, dev, rank):
        from clustercontrast.trainers import ClusterContrastWithGANTrainer
        from dual_gan.models.models import create_model
        self.enc, self.mem, self.opt = _cc_parts(dev, self.crops)
        self.imgs, self.labels, self.indexes = synth_reid(self.crops, 2048, dev, 1 + rank)
        self.gan = create_model(dualgan_opt())
        self.gan_in = synth_dualgan(self.crops, dev, 11 + rank)
        self.trainer = ClusterContrastWithGANTrainer(self.enc, GAN=self.gan, memory=self.mem)
        self.loss = None

    def step(self):
        self.gan.set_input(self.gan_in)
        self.loss = self.trainer.joint_step(self.imgs, self.labels, self.indexes, self.opt)

    def losses(self):
        return {"loss": float(self.loss)}


class Joint4b(Workload):
    key = "4b"
    crops = 32
    gflop_per_crop = 160.0
    describe = ("joint FD-GAN + cluster-contrast step (ClusterContrastWithGANTrainer.joint_step with FDGANModel behind "
                "fdgan.adaptor.FDGANAdaptor): cluster-contrast encoder step + FD-GAN step on 32 crops = 16 pairs per GPU")

    def build(self, dev, rank):
        from clustercontrast.trainers import ClusterContrastWithGANTrainer
        from fdgan.adaptor import FDGANAdaptor
        from fdgan.model import FDGANModel
        self.enc, self.mem, self.opt = _cc_parts(dev, self.crops)
        self.imgs, self.labels, self.indexes = synth_reid(self.crops, 2048, dev, 1 + rank)
        torch.manual_seed(1234)
        self.model = FDGANModel(fdgan_opt(batch_size=self.crops // 2))
        self.model.reset_model_status()
        self.gan = FDGANAdaptor(self.model)
        self.pair = synth_inputs(self.crops // 2, dev, seed=1234 + rank)
        self.trainer = ClusterContrastWithGANTrainer(self.enc, GAN=self.gan, memory=self.mem)
        torch.manual_seed(99 + rank)
        self.loss = None

    def step(self):
        self.gan.set_input(self.pair)
        self.loss = self.trainer.joint_step(self.imgs, self.labels, self.indexes, self.opt)

    def losses(self):
        d = {"loss": float(self.loss)}
        d.update(self.model.get_current_errors())
        return d


class DPTNStep(Workload):
    key = "5"
    crops = 64
    gflop_per_crop = 20.6
    launch_note = ("the GAN networks' forward / backward programs replayed as single-stream hipGraphs captured per network "
                   "(rg_hip/netgraph.py; RG_NET_GRAPHS=0 = eager); everything else eager, one hipLaunchKernel per kernel through the C ABI")
    capture_steps = 4
    dtype = "fp8"
    peak = F8_MFMA_PEAK_TFLOPS
    mfma_products = 1
    conv_families = ("conv_f8",)
    kernel_note = ("fp8 conv implicit-GEMM family (conv_f8_*_kernel: e4m3 activations / weights, e5m2 gradients, per-tensor "
                   "scales, fp32 accumulate, v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales)")
    describe = ("dual_gan two-generator path: DPTNModel.optimize_parameters (DPTN_model.py:216-225; source->source and "
                "source->target branches of DPTNGenerator, ResDiscriminator on the target branch), fp8 MFMA convolutions, "
                "64 crops of 128x64 per GPU, hinge GAN loss, perceptual loss off (VGG-19 weights need a download)")

    def build(self, dev, rank):
        from dual_gan.models.models import create_model
        torch.manual_seed(4321)
        self.gan = create_model(dualgan_opt(model="DPTN", model_gen="DPTN", gan_mode="hinge", layers_g=3, conv_dtype="fp8",
                                            lambda_rec=5.0, lambda_g=2.0, t_s_ratio=0.5, dis_layers=4, ratio_g2d=0.1))
        self.gan_in = synth_dualgan(self.crops, dev, 21 + rank, with_target=True)

    def step(self):
        self.gan.set_input(self.gan_in)
        self.gan.optimize_parameters()

    def losses(self):
        return {k: float(v) for k, v in self.gan.get_current_errors().items()}

    def cpu_baseline(self, warm=1, timed=3):
        from oracle import ref_dualgan as OD
        cores = host_cores()
        torch.set_num_threads(cores)
        torch.manual_seed(0)
        B = self.crops
        model = OD.ODPTNModel(gan_mode="hinge")
        d = OD.synth_dptn_inputs(B, seed=2)
        dt = _time_cpu(lambda: model.step(d), warm, timed)
        return {"value": round(B / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": "DPTNModel step (oracle/ref_dualgan.ODPTNModel, torch CPU fp32 — the reference's own arithmetic "
                          "type) at %d crops of 128x64, mean of %d steps after %d warm-up (%.1f s/step)" % (B, timed, warm, dt)}


WORKLOADS = {w.key: w for w in (FDGANStep, CCStep, Joint4a, Joint4b, DPTNStep)}


# ---------------------------------------------------------------------------------------------------------------------
def pmc_traffic(key, launches_per_step):
    """HBM bytes per conv launch from the committed PMC passes (profiles/r04_pmc_traffic.json, else an earlier round's; produced on the GPU box by
    tools/prof_summary.py + tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    script).  PMC counters cannot be read from inside the process, so the figure is not live; None if absent."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(REPO, "profiles", name)
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        fam = doc.get("configs", {}).get(key, {}).get("families", {}).get("conv_f8" if key == "5" else "conv")
        if fam is None and key == "2":
            fam = doc.get("families", {}).get("conv")
        if fam is None:
            continue
        per_step = fam["read_bytes_per_step"] + fam["write_bytes_per_step"]
        return round(per_step / max(launches_per_step, 1)), (
            "profiles/%s: %.1f GB read + %.1f GB written per step by the conv family incl. split-K finish/reduce, filter "
            "re-layout and (fp8) quantisation kernels" % (name, fam["read_bytes_per_step"] / 1e9, fam["write_bytes_per_step"] / 1e9))
    return None, "no PMC pass committed for this configuration"


def measure(w, args, dev, rank, world, use_dist, headline):
    """warm-up, timed region (barrier + synchronize on both sides, MAX over ranks), then the profiled steps on every rank"""
    from rg_hip import ops
    steps, warmup = (args.steps, args.warmup) if headline else (args.other_steps, 3)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # configurations whose networks replay captured launch programs capture them in their first calls (rg_hip.netgraph.WARMUP
    # eager calls per key, then one capturing call): those preparation steps come BEFORE the W warm-up steps of the contract
    # ... and ONE preparation step for every configuration: the first call of each convolution geometry measures its kernel /
    # plan candidates (csrc/conv_igemm.hip, choose_impl: ~0.5 s per configuration) and the second streams are probed
    # (rg_hip.ops.concurrent_stream) — neither belongs into a timed step should the caller ask for zero warm-up steps
    for i in range(max(1, w.capture_steps)):
        w.step()
    for i in range(warmup):
        w.step()
        if rank == 0 and headline:
            torch.cuda.synchronize()
            log("warm-up step %d done" % i)
    run = w.step
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    t_host = time.perf_counter() - t0                       # host enqueue time: no synchronisation inside the loop
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    losses = w.losses()
    if rank == 0:
        log("config %s: timed %d steps: %.2f ms/step (host enqueue %.2f)" % (w.key, steps, 1e3 * elapsed / steps,
                                                                             1e3 * t_host / steps))
        if w.capture_steps and os.environ.get("RG_BENCH_DEBUG") == "1":
            from rg_hip import netgraph as NG
            for name in ("net_G", "net_D"):
                net = getattr(getattr(w, "gan", None), name, None)
                net = getattr(net, "module", net)
                if net is not None:
                    log("   %s graphs: %s" % (name, [(c, r, ok) for c, r, ok in NG.stats(net).values()]))

    # ---- roofline: per-launch HIP events on the conv implicit-GEMM kernels; EVERY rank runs these steps (collectives) ----
    roof = None
    psteps = args.profile_steps if headline else min(args.profile_steps, 2)
    if psteps > 0:
        # per-launch durations are taken with the kernels launched back to back on ONE stream: in the timed region the
        # weight-gradient kernels and independent network passes run concurrently on side streams, which lengthens each
        # overlapped launch without saying anything about the kernel
        w.serial(True)
        w.step()
        torch.cuda.synchronize()
        ops.profile_reset()
        ops.profile_enable(True)
        for _ in range(psteps):
            w.step()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        w.serial(False)
        fam = ops.profile_collect()
        conv = [fam[k] for k in w.conv_families if k in fam]
        ms = sum(f["ms"] for f in conv)
        fl = sum(f["flops"] for f in conv)
        by = sum(f["bytes"] for f in conv)
        calls = sum(f["calls"] for f in conv)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        lps = calls // psteps
        traffic, traffic_note = pmc_traffic(w.key, lps)
        alg_bytes = round(by / max(calls, 1))
        roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": w.peak, "unit": "TFLOP/s",
                "frac": round(achieved / w.peak, 4), "traffic": traffic,
                "peak_note": ("2500 TFLOP/s dense bf16 MFMA / %d bf16 products per fp32 product" % w.mfma_products
                              if getattr(w, "mfma_products", 1) > 1 else "dense MFMA peak of the operand type"),
                "frac_vs_fp32_mfma": round(achieved / F32_MFMA_PEAK_TFLOPS, 4) if w.dtype == "f32" else None,
                "issued_mfma_tflops": round(achieved * getattr(w, "mfma_products", 1), 2) if getattr(w, "mfma_products", 1) > 1 else None,
                "issued_mfma_peak": BF16_MFMA_PEAK_TFLOPS if getattr(w, "mfma_products", 1) > 1 else None,
                "issued_mfma_frac": (round(achieved * w.mfma_products / BF16_MFMA_PEAK_TFLOPS, 4)
                                     if getattr(w, "mfma_products", 1) > 1 else None),
                "traffic_unit": "bytes per launch", "traffic_source": traffic_note,
                "algorithmic_bytes": alg_bytes,
                "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if traffic and alg_bytes else None,
                "kernel": w.kernel_note,
                "launches_per_step": lps,
                "avg_launch_us": round(1e3 * ms / max(calls, 1), 2),
                "algorithmic_tflop_per_step": round(fl / psteps / 1e12, 4),
                "kernel_ms_per_step": round(ms / psteps, 3),
                "measured_over": "%d profiled steps after the timed region, single-stream launch order (the timed region "
                                 "overlaps independent kernels on side streams)" % psteps,
                "by_family": {k: {"ms_per_step": round(v["ms"] / psteps, 3),
                                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2) if v["flops"] else None,
                                  "gbps_algorithmic": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None,
                                  "launches": v["calls"] // psteps} for k, v in fam.items() if v["calls"]}}
    if use_dist:
        dist.barrier()
    ms_step = 1e3 * elapsed / steps
    return {"value": round(world * w.crops * steps / elapsed, 2), "ms_per_step": round(ms_step, 3), "steps": steps,
            "warmup": warmup, "capture_steps": max(1, w.capture_steps), "host_enqueue_ms_per_step": round(1e3 * t_host / steps, 3),
            "launch": w.launch_note,
            "losses": {k: round(float(v), 5) for k, v in losses.items()}, "roofline": roof,
            "step_tflops_algorithmic": round(w.gflop_per_crop * w.crops / 1e3 / (ms_step * 1e-3), 2)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the ranks as a fresh child process tree (this
    process has made no GPU call: importing torch and parsing arguments initialise nothing) and relay its output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + \
          [a for a in sys.argv[1:] if a != "--dry-run"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    if args.dry_run:
        print(json.dumps({"dry_run": True, "ranks": args.gpus, "cmd": cmd,
                          "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")},
                          "configs_measured": [args.config] + (OTHERS_DIST if args.others and not args.no_others and args.config == "2"
                                                                 else [])}),
              file=_real_stdout)
        _real_stdout.flush()
        return 0
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


_real_stdout = sys.stdout
OTHERS_SINGLE = ["3", "4a", "4b", "5"]
OTHERS_DIST = ["4a", "4b", "5"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=sorted(WORKLOADS))
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--other-steps", type=int, default=8, help="timed steps of each configuration under other_configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="measure only --config")
    ap.add_argument("--others", action="store_true",
                    help="multi-GPU runs measure only the headline unless this is given (then BASELINE.json's DDP configurations 4a / 4b / "
                         "5 follow it under other_configs); single-GPU runs measure the other configurations by default")
    ap.add_argument("--dry-run", action="store_true",
                    help="with --gpus N outside torch.distributed.run: print the child command that would start the ranks (one "
                         "JSON line) and exit without touching a GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("RG_FORCE_REDUCE") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        from rg_hip.parallel import init_process_group          # RCCL, collectives on a high-priority stream (own hardware queue)
        init_process_group(rank, world, local_rank)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    w = WORKLOADS[args.config]()
    w.build(dev, rank)
    if rank == 0:
        log("config %s built; warm-up %d steps" % (w.key, args.warmup))
    head = measure(w, args, dev, rank, world, use_dist, headline=True)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle step on %d host threads) ..." % host_cores())
        try:
            cpu = w.cpu_baseline()
        except Exception as e:                                   # the GPU measurement stands without it; say why
            cpu = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if cpu and cpu.get("value"):
            log("cpu baseline done: %.3f images/s" % cpu["value"])

    others = None
    # N > 1: the headline only, unless --others — the line the driver's scaling curve is computed from must not depend on the graphed
    # configurations (4a, 5) getting through their first multi-rank run (no N > 1 run exists yet: DESIGN section 5)
    if not args.no_others and args.config == "2" and (world == 1 or args.others):
        # the other BASELINE configurations, measured in the same process behind the headline.  Multi-GPU: the DDP
        # configurations of BASELINE.json — 4 (the joint trainer: 4a as committed in trainers_b.py:617-814, 4b with FDGANModel
        # in the GAN role) and 5 (dual_gan DPTNModel, fp8) — so that their scaling curves come out of the same runs
        others = {}
        keys = OTHERS_SINGLE if world == 1 else OTHERS_DIST
        del w
        torch.cuda.empty_cache()
        for k in keys:
            try:
                ow = WORKLOADS[k]()
                ow.build(dev, rank)
                r = measure(ow, args, dev, rank, world, use_dist, headline=False)
                r.update({"workload": ow.describe, "dtype": ow.dtype, "crops_per_gpu": ow.crops, "unit": "images/s",
                          "n_gpus": world})
                others[k] = r
                del ow
                torch.cuda.empty_cache()
            except Exception as e:
                if use_dist:
                    raise                                         # a rank that skips a workload would desynchronise the others
                others[k] = {"error": "%s: %s" % (type(e).__name__, e)}
                log("config %s failed: %s" % (k, others[k]["error"]))
        w = WORKLOADS[args.config]()

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": head["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic (random-init weights, seeded on-device crops + 18-ch pose maps)",
            "config": {"workload": w.describe, "baseline_config": w.key, "crops_per_gpu": w.crops,
                       "global_batch_crops": world * w.crops, "parallelism": "dp%d" % world},
            "step_tflops_algorithmic": head["step_tflops_algorithmic"],
            "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"], "launch": head["launch"],
            "losses": head["losses"],
            "roofline": head["roofline"], "cpu_baseline": cpu,
        }
        if others is not None:
            out["other_configs"] = others
        print(json.dumps(out), file=_real_stdout)
        _real_stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries the ONE JSON line and nothing else: the networks' constructors print like the reference's do ("pooling_type:
    # gem", parameter counts, "initialize network with ..."), which goes to stderr together with the progress log
    _real_stdout = sys.stdout
    sys.stdout = sys.stderr
    main()
