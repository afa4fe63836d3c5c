import os, sys, time
sys.path.insert(0, '/root/repo'); 
import torch, bench
key = sys.argv[1] if len(sys.argv) > 1 else "5"
def run(graphs, steps=6):
    os.environ["RG_NET_GRAPHS"] = "1" if graphs else "0"
    import rg_hip.netgraph as NG
    NG.ENABLED = bool(graphs)
    torch.manual_seed(0)
    cls = bench.WORKLOADS[key]; cls.crops = 8
    w = cls(); w.build(torch.device("cuda",0), 0)
    out = []
    for i in range(steps):
        w.step(); out.append({k: float(v) for k, v in w.losses().items()})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10): w.step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("graphs", graphs, "host ms/step %.2f wall %.2f" % (100*th, 100*(time.perf_counter()-t0)))
    return out
a = run(False); b = run(True)
for i,(x,y) in enumerate(zip(a,b)):
    d = {k: (x[k], y[k]) for k in x if x[k] != y[k]}
    print("step", i, "identical" if not d else d)
