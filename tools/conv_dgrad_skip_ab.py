"""conv2's input gradient at the step's shape: standard rows against frame-major rows without / with the temporal tap skipping
(stream-K, weighted plan).    python tools/conv_dgrad_skip_ab.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import _lib, kernels as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
B = 32
torch.manual_seed(3)
dyp = torch.zeros(B, 16, 9, 9, 768, device=dev, dtype=torch.bfloat16)
dyp[:, 4:12, 1:8, 1:8] = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
w2 = (torch.randn(768, 5, 3, 3, 768, device=dev) * 0.01).bfloat16()
inv = K.conv_row_table_inv(B, 16, 7, 7, dev, order=2)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    return ts[len(ts) // 2]


for rnd in range(2):
    for order, sw in ((0, 126), (2, 62), (2, 126)):
        _lib.set_tuning("conv_k_order", sw)
        t = timed(lambda: K.conv3d_k533_dgrad(dyp, w2, out_rows=inv if order else None, order=order))
        print("rows %s, temporal tap skipping %s: conv2 input gradient %6.1f us" % ("frame-major" if order else "standard   ", "on " if sw & 64 and order else "off", t), flush=True)
_lib.set_tuning("conv_k_order", 126)
