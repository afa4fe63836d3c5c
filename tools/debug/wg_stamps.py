"""In-kernel phase stamps of the forward conv kernel's k-loop (development; needs a libreidgan_st.so built from a stamped copy of
csrc/conv_igemm.hip, see profiles/r03_conv_ablation.txt): per k-tile iteration of wave 0 of each workgroup, in shader cycles:
load issue | fragment reads + split + MFMA issue | LDS stores of the next tile (incl. the wait for its global loads) | barrier wait."""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
import torch
import numpy as np
dll = ctypes.CDLL(os.path.join(ROOT, "reid-gan_amd/lib/libreidgan_st.so"))
dev = torch.device("cuda:0")
def run(N, C, H, W, K, k=1):
    x = torch.randn(N, C, H, W, device=dev); w = torch.randn(K, C, k, k, device=dev) * 0.05
    y = torch.empty(N, K, H, W, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = dll.rg_conv2d_fwd
    f.argtypes = [ctypes.c_void_p]*4 + [ctypes.c_int]*13 + [ctypes.c_void_p]*3 + [ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    call = lambda: f(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, C, H, W, K, k, k, 1, 1, k//2, k//2, H, W, None, None, None, 0, 0.0, ws.data_ptr(), ws.numel(), st)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    n = 512
    buf = (ctypes.c_ulonglong * (64 * n))()
    dll.rg_debug_stamps(buf, 64 * n)
    a = np.array(buf, dtype=np.uint64).reshape(n, 64)[:, :40].reshape(n, 8, 5).astype(np.float64)
    a = a[a[:, 0, 0] > 0]
    d = np.diff(a, axis=2)            # [wg][iter][4]: load issue, mma (reads+split+mfma issue), stores, barrier
    it = np.median(a[:, 1:, 0] - a[:, :-1, 0], axis=0)
    print("%dx%dx%dx%d -> %d k%d, %d WGs: cycles per iteration (median over WGs), iterations 1..7: %s" % (N, C, H, W, K, k, len(a), np.round(it).astype(int)))
    for name, j in (("load issue", 0), ("reads+split+mfma issue", 1), ("lds stores (+vm wait)", 2), ("barrier wait", 3)):
        print("   %-26s %s" % (name, np.round(np.median(d[:, :7, j], axis=0)).astype(int)))
run(32, 128, 32, 16, 512)
run(32, 256, 64, 32, 128)
