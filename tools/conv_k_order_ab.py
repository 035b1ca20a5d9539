"""Convolution forward / input gradient with the K-tiles in storage order (tap-major, `conv_k_order=0`) against channel-block-major
(`conv_k_order=1`): time per launch, and that the two orders give the same sums (they differ by fp32 rounding of the partial sums only).

    python tools/conv_k_order_ab.py [iters]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/conv_k_order_ab.py 3
    python tools/conv_k_order_ab.py pmc <dir>          per-launch FETCH_SIZE of the traced run, in launch order

Launch order of the run: for order in ORDERS: conv1 forward x (1 + iters), conv2 forward x (1 + iters), conv2 input gradient x (1 + iters)."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "pmc":
    d = sys.argv[2]
    f = (glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv")))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "Conv" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r["Counter_Name"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    for (i, n, c), v in sorted(per.items()):
        print("%6d %-60s %-10s %8.3f GB" % (i, n, c, v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9))
    sys.exit(0)

import torch
from shg_vqa_amd import _lib, kernels as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ORDERS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
dev = "cuda"
B = 32
torch.manual_seed(3)
x_cl = torch.zeros(B, 16, 9, 9, 2048, device=dev, dtype=torch.bfloat16)
x_cl[:, :, 1:8, 1:8] = torch.randn(B, 16, 7, 7, 2048, device=dev).bfloat16()
w1 = (torch.randn(768, 5, 3, 3, 2048, device=dev) * 0.01).bfloat16()
w2 = (torch.randn(768, 5, 3, 3, 768, device=dev) * 0.01).bfloat16()
b1 = torch.zeros(768, device=dev)
dyp = torch.zeros(B, 16, 9, 9, 768, device=dev, dtype=torch.bfloat16)          # conv2's output gradient, padded by 4 / 1 / 1
dyp[:, 4:12, 1:8, 1:8] = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
dy1 = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()                      # conv1's / conv2's output gradients, unpadded
dy2 = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
dw1 = torch.zeros(768, 5, 3, 3, 2048, device=dev)
dw2 = torch.zeros(768, 5, 3, 3, 768, device=dev)


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


res = {}
for order in ORDERS:
    _lib.set_tuning("conv_k_order", order)
    y1 = torch.zeros(B, 12, 9, 9, 768, device=dev, dtype=torch.bfloat16)
    t1 = timed(lambda: K.conv3d_k533_fwd(x_cl, w1, b1, 1, pad_out=True, out=y1))
    y2 = [None]

    def f2():
        y2[0] = K.conv3d_k533_fwd(y1, w2, b1, 1, pad_out=False)
    t2 = timed(f2)
    dx = [None]

    def f3():
        dx[0] = K.conv3d_k533_dgrad(dyp, w2)
    t3 = timed(f3)
    t4 = timed(lambda: K.conv3d_k533_wgrad(x_cl, dy1, dw1, accumulate=True))
    t5 = timed(lambda: K.conv3d_k533_wgrad(y1, dy2, dw2, accumulate=True))
    dw1.zero_(); dw2.zero_()
    K.conv3d_k533_wgrad(x_cl, dy1, dw1, accumulate=True)
    K.conv3d_k533_wgrad(y1, dy2, dw2, accumulate=True)
    res[order] = (y1.float(), y2[0].float(), dx[0].float(), dw1.clone(), dw2.clone())
    print("conv_k_order=%d: conv1 forward %7.1f us (%4.0f TFLOP/s)   conv2 forward %6.1f us   conv2 input gradient %6.1f us   "
          "conv1 weight gradient %7.1f us   conv2 weight gradient %6.1f us" % (
              order, t1, 2.0 * B * 12 * 49 * 768 * 45 * 2048 / t1 / 1e6, t2, t3, t4, t5), flush=True)
for name, a, b in zip(("conv1 forward", "conv2 forward", "conv2 input gradient", "conv1 weight gradient", "conv2 weight gradient"),
                      res[ORDERS[0]], res[ORDERS[-1]]):
    print("%-22s max |first - last order| = %.3e  (max |value| %.3e)" % (name, float((a - b).abs().max()), float(a.abs().max())))
