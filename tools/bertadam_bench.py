"""Isolated timing of the optimiser sweep (shg_bertadam_arena) at the step's size (289 M parameters: 34 bytes each = 9.83 GB)
over its variants ("bertadam_mode" bits: 1 two vectors per lane, 8 four, 2 non-temporal stores of the zeroed gradient and the
shadow too, 4 no non-temporal accesses at all) and grid sizes.  Each timed launch follows a gradient-norm pass, as in the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import _lib, kernels as K

dev = "cuda"
n = 289038100
p = torch.randn(n, device=dev) * 0.02
g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev)
v = torch.zeros(n, device=dev)
sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
step = torch.zeros(1, dtype=torch.int64, device=dev)
bytes_ = 34.0 * n
for mode in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,3,9,11,5,13,0".split(","))]:
    for blocks in (4096, 16384, 65536, 1 << 20):
        _lib.set_tuning("bertadam_mode", mode)
        _lib.set_tuning("bertadam_blocks", blocks)
        ts = []
        for it in range(6):
            norm = K.grad_norm(g)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.bertadam_arena(p, g, m, v, sh, norm, 5.0, 1e-5, 0.1, 10000, step, zero_grad=True)
            e1.record()
            torch.cuda.synchronize()
            g.normal_(0, 1e-3)
            if it >= 2:
                ts.append(e0.elapsed_time(e1))
        t = sum(ts) / len(ts)
        print("mode %2d blocks %8d: %.3f ms  %.2f TB/s" % (mode, blocks, t, bytes_ / t / 1e9), flush=True)
