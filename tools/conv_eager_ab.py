"""Isolated timing of the convolution kernels with eager launches and events (stream-K launches need their per-stream
workspace, which a hipGraph capture on a fresh stream does not have).  Run with SHG_STREAMK=0 / 1 / 3 / 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from shg_vqa_amd import _lib

dev = "cuda"


def timed(fn, iters=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = 32
x_cl = torch.randn(B, 16, 9, 9, 2048, device=dev).bfloat16()
w1 = (torch.randn(768, 5, 3, 3, 2048, device=dev) * 0.01).bfloat16()
b1 = torch.zeros(768, device=dev)
y1 = torch.zeros(B, 12, 9, 9, 768, device=dev, dtype=torch.bfloat16)
pre = torch.empty(B, 12, 7, 7, 768, device=dev, dtype=torch.bfloat16)
w2 = (torch.randn(768, 5, 3, 3, 768, device=dev) * 0.01).bfloat16()
d1 = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()
dw1 = torch.zeros(768, 5, 3, 3, 2048, device=dev)
d2 = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
dw2 = torch.zeros(768, 5, 3, 3, 768, device=dev)
d2p = torch.nn.functional.pad(d2, (0, 0, 1, 1, 1, 1, 4, 4))
K.conv_workspace(B, 16, 7, 7, torch.device(dev))
n0 = _lib.lib().shg_gemm_streamk_launches()
rows = [("conv1 fwd", lambda: K.conv3d_k533_fwd(x_cl, w1, b1, 1, pad_out=True, out=y1, want_pre=True, pre_out=pre), 12 * 2048),
        ("conv2 fwd", lambda: K.conv3d_k533_fwd(y1, w2, b1, 1, pad_out=False, want_pre=True), 8 * 768),
        ("conv1 wgrad", lambda: K.conv3d_k533_wgrad(x_cl, d1, dw1, accumulate=True), 12 * 2048),
        ("conv2 wgrad", lambda: K.conv3d_k533_wgrad(y1, d2, dw2, accumulate=True), 8 * 768),
        ("conv2 dgrad", lambda: K.conv3d_k533_dgrad(d2p, w2), 12 * 768)]
for name, fn, tc in rows:
    t = timed(fn)
    print("  %-12s %8.1f us %6.0f TF" % (name, t, 2.0 * B * 49 * 768 * 45 * tc / t / 1e6), flush=True)
print("  stream-K launches: %d" % (_lib.lib().shg_gemm_streamk_launches() - n0))
