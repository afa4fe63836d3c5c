"""How far ahead of the GPU does the host run?  Times the Python/ctypes launch path of the FD-GAN step alone (no sync
inside the loop) against the synchronised step time."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd")); sys.path.insert(0, REPO)
import torch
import bench as HB
from fdgan.model import FDGANModel
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = FDGANModel(HB.fdgan_opt())
model.reset_model_status()
data = HB.synth_inputs(16, dev, 1)
def step():
    model.set_input(data); model.optimize_parameters()
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(5): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.1f ms/step, gpu-synchronised %.1f ms/step" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(3): step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
