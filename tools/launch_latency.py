"""GPU-side cost of a dependent kernel on one stream when the host is far ahead: N tiny kernels back to back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
dev = "cuda"
x = torch.zeros(64, device=dev)
c = torch.zeros(1, dtype=torch.int64, device=dev)
for name, fn in (("torch add_", lambda: x.add_(1.0)), ("shg_add_i64 (1 thread)", lambda: K.add_i64(c, 1))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    n = 3000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n): fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-24s host %.2f us/launch, GPU %.2f us/kernel" % (name, (t1 - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3))
# the same inside a captured graph (no host at all)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x.add_(1.0); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(1000): x.add_(1.0)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("graph of 1000 add_        GPU %.2f us/kernel" % (e0.elapsed_time(e1)))
