import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "reid-gan_amd")); sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from rg_hip import nn as rnn, ops
from rg_hip.tape import Tape
dev = torch.device("cuda:0")
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return ((a-b).abs().max()/b.abs().max().clamp_min(1e-12)).item()
g = torch.Generator().manual_seed(0)
for shape in ((3,128,64,32),(3,256,32,16),(3,512,31,15),(2,8,5,3)):
    x = torch.randn(shape, generator=g)*2+0.7
    dy = torch.randn(shape, generator=g)
    xd = x.double().requires_grad_(True)
    y = F.leaky_relu(F.instance_norm(xd, eps=1e-5), 0.2)
    y.backward(dy.double())
    m = rnn.InstanceNorm2d(shape[1]).to(dev)
    t = Tape()
    yy = m.tf(t, x.to(dev), act=ops.ACT_LEAKY, slope=0.2)
    dx = m.tb(t, dy.to(dev))
    print(shape, "fwd", rel(yy, y), "bwd", rel(dx, xd.grad))
