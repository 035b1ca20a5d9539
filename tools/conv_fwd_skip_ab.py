"""Conv forwards at the step's shapes: standard rows / position-major rows without and with the zero-border tap skipping
(stream-K, weighted plan).    python tools/conv_fwd_skip_ab.py [iters]"""
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
w1 = (torch.randn(768, 5, 3, 3, 2048, device=dev) * 0.01).bfloat16()
w2 = (torch.randn(768, 5, 3, 3, 768, device=dev) * 0.01).bfloat16()
b1 = torch.zeros(768, device=dev)
inv2 = K.conv_row_table_inv(B, 12, 7, 7, dev)


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


ref = {}
for rnd in range(2):
    for order, sw in ((0, 30), (1, 30), (1, 62)):
        _lib.set_tuning("conv_k_order", sw)
        pre = torch.empty(B, 12, 7, 7, 768, device=dev, dtype=torch.bfloat16)
        t1 = timed(lambda: K.conv3d_k533_fwd(x_cl, w1, b1, 1, pad_out=True, out=y1, want_pre=True, pre_out=pre, order=order))
        out2 = [None]

        def f2():
            out2[0] = K.conv3d_k533_fwd(y1, w2, b1, 1, pad_out=False, want_pre=True, order=order, y_rows=inv2 if order else None)
        t2 = timed(f2)
        key = "y1"
        if rnd == 0:
            ref[(order, sw)] = (y1.clone(), out2[0][0].clone())
        print("rows %s, tap skipping %s: conv1 forward %7.1f us (%4.0f TFLOP/s algorithmic)   conv2 forward %6.1f us" % (
            "position-major" if order else "standard      ", "on " if sw & 32 and order else "off", t1, 2.0 * B * 12 * 49 * 768 * 45 * 2048 / t1 / 1e6, t2),
            flush=True)
_lib.set_tuning("conv_k_order", 126)
a, b, c = ref[(0, 30)], ref[(1, 30)], ref[(1, 62)]
print("conv1 output equal across the three: %s %s; conv2 output equal: %s %s" % (torch.equal(a[0], b[0]), torch.equal(a[0], c[0]),
                                                                              torch.equal(a[1], b[1]), torch.equal(a[1], c[1])))
