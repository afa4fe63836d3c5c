"""hipGraph capture of whole training steps (rg_hip/graph.py): a replayed step launches the kernels of the eager step with the
same arguments, so losses, memory bank and parameters are BIT-IDENTICAL to the eager run; host-side bookkeeping (Adam step
counts, BatchNorm num_batches_tracked, arena epochs) follows the replays; dropout masks change from replay to replay through the
device clock.  (Capture is opt-in: on ROCm 7.2 the replay of these 1 000+ node multi-stream graphs costs the host as much as the
eager launches — profiles/r02_graph_vs_eager.txt — and the FD-GAN step, whose two discriminator backward passes run on
different streams inside autograd, is not captured reliably; the joint and cluster-contrast steps are.)"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_cluster_contrast_step_graph_and_bn_counters(dev):
    from rg_hip import optim as roptim
    from rg_hip.graph import CapturedStep
    import reid.models as RM
    from clustercontrast.models.cm import ClusterMemory
    from clustercontrast.trainers import ClusterContrastTrainer

    def run(graph):
        torch.manual_seed(0)
        # (the cluster-contrast wrapper's layer4 stride edit only fits Bottleneck depths — tests/test_cc_gpu.py — so the shallow
        # encoder is the FD-GAN ReID wrapper; the trainer takes any encoder)
        enc = RM.create('resnet18', pretrained=False).to(dev).train()
        mem = ClusterMemory(enc.num_features, 32, temp=0.05, momentum=0.1).to(dev)
        g = torch.Generator(device=dev).manual_seed(1)
        mem.features = F.normalize(torch.randn(32, enc.num_features, generator=g, device=dev), dim=1)
        opt = roptim.Adam([{"params": [p]} for p in enc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
        imgs = torch.randn(8, 3, 128, 64, generator=g, device=dev)
        labels = torch.tensor([1, 1, 4, 4, 1, 9, 9, 4], device=dev)
        tr = ClusterContrastTrainer(enc, mem)
        out = {}

        def step():
            out["loss"] = tr.step(imgs, labels, opt)
            return out["loss"]
        runner = CapturedStep(step, warmup=2) if graph else step
        losses = []
        for _ in range(5):
            r = runner()
            losses.append(float(r))
        return enc, mem, losses

    enc_e, mem_e, l_e = run(False)
    enc_g, mem_g, l_g = run(True)
    assert l_e == l_g, (l_e, l_g)
    assert torch.equal(mem_e.features, mem_g.features)
    for (n, p), (_, q) in zip(enc_e.state_dict().items(), enc_g.state_dict().items()):
        assert torch.equal(p, q), n                       # includes num_batches_tracked == 5 on both sides
    counters = [k for k in enc_g.state_dict() if k.endswith("num_batches_tracked")]
    assert counters and all(int(enc_g.state_dict()[k]) == 5 for k in counters)


def test_dropout_mask_changes_between_replays(dev):
    from rg_hip.graph import CapturedStep
    from rg_hip import nn as rnn
    from rg_hip.tape import Tape
    drop = rnn.Dropout(0.5).to(dev).train()
    x = torch.ones(4096, device=dev)
    torch.manual_seed(3)

    def step():
        return drop.tf(Tape(record=False), x)
    runner = CapturedStep(step, warmup=1)
    runner()
    a = runner().clone()
    b = runner().clone()
    c = runner().clone()
    assert not torch.equal(a, b) and not torch.equal(b, c)
    assert abs(float(a.mean()) - 1.0) < 0.1 and abs(float(b.mean()) - 1.0) < 0.1
