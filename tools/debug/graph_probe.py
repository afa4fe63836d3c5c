"""How fast does hipGraph replay a SINGLE-STREAM capture of one network forward (no side streams, no autograd)?  Round 2 measured
12 us per node for whole-step multi-stream graphs; this probe separates the node cost from the multi-stream cost.
usage: graph_probe.py [5|4a]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RG_WGRAD_STREAM"] = "0"
import torch
import bench

key = sys.argv[1] if len(sys.argv) > 1 else "5"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cls = bench.WORKLOADS[key]
cls.crops = 4
w = cls()
w.build(dev, 0)
for _ in range(3):
    w.step()
torch.cuda.synchronize()
gan = w.gan
net = gan.net_G
net.eval()
from rg_hip import lib as L
calls = [0]
x_in = None


def fwd():
    with torch.no_grad():
        return gan.forward() if hasattr(gan, "forward") else None


# count launches of one generator forward
import rg_hip.lib as RL
orig = {}
fwd(); fwd()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    fwd()
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print("eager generator forward: host %.3f ms, wall %.3f ms" % (1e3 * t_host, 1e3 * t_all))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd(); fwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd()
torch.cuda.synchronize()
g.replay(); g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print("graph replay:            host %.3f ms, wall %.3f ms" % (1e3 * t_host, 1e3 * t_all))
