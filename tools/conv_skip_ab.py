"""Conv weight gradients with position-major rows: zero-border skipping on (conv_k_order bit 3) / off, and the standard row order,
isolated at the step's shapes; conv1 forward in both row orders.

    python tools/conv_skip_ab.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import _lib, kernels as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
B = 32
torch.manual_seed(3)
x_cl = torch.zeros(B, 16, 9, 9, 2048, device=dev, dtype=torch.bfloat16)
x_cl[:, :, 1:8, 1:8] = torch.randn(B, 16, 7, 7, 2048, device=dev).bfloat16()
y1 = torch.zeros(B, 12, 9, 9, 768, device=dev, dtype=torch.bfloat16)
y1[:, :, 1:8, 1:8] = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()
w1 = (torch.randn(768, 5, 3, 3, 2048, device=dev) * 0.01).bfloat16()
b1 = torch.zeros(768, device=dev)
dy1 = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()
dy2 = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
dw1 = torch.zeros(768, 5, 3, 3, 2048, device=dev)
dw2 = torch.zeros(768, 5, 3, 3, 768, device=dev)
ss = torch.zeros(1, dtype=torch.float64, device=dev)


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
    for order, sw in ((0, 30), (1, 6), (1, 14), (1, 30)):
        _lib.set_tuning("conv_k_order", sw)
        t1 = timed(lambda: K.conv3d_k533_wgrad_sumsq(x_cl, dy1, dw1, ss, order=order))
        t2 = timed(lambda: K.conv3d_k533_wgrad_sumsq(y1, dy2, dw2, ss, order=order))
        t3 = timed(lambda: K.conv3d_k533_wgrad(x_cl, dy1, dw1, accumulate=True, c0=0, cn=512, order=order))
        tf = timed(lambda: K.conv3d_k533_fwd(x_cl, w1, b1, 1, pad_out=True, out=y1, order=order))
        y1[:, :, 1:8, 1:8] = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()
        print("rows %s, skip %-24s: conv1 weight gradient %7.1f us   conv2 weight gradient %6.1f us   conv1 slice of 512 channels %7.1f us   conv1 forward %7.1f us"
              % ("position-major" if order else "standard      ", ("on, longest taps first" if sw & 16 else "on ") if sw & 8 and order else "off", t1, t2, t3, tf), flush=True)
_lib.set_tuning("conv_k_order", 126)
