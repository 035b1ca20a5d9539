"""Fixed cost vs per-K-step cost of shg_gemm on a small grid (GPU-paced through a hipGraph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from gemm_shapes import bench  # noqa

SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1280, 768), (4096, 768), (12576, 768)]
for (M, N) in SHAPES:
    for Kd in (64, 128, 256, 512, 768, 1536, 3072, 6144):
        x = torch.randn(M, Kd, device="cuda").bfloat16()
        w = torch.randn(N, Kd, device="cuda").bfloat16()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = bench(lambda: K.gemm(x, w, y, None, True, True))
        t2 = bench(lambda: torch.matmul(x, w.t()))
        print("M=%6d N=%5d K=%5d  %7.1f us   torch %7.1f us" % (M, N, Kd, t, t2), flush=True)
