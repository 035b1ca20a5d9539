import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from shg_vqa_amd import kernels as K
from gemm_shapes import bench
for (M, N, Kd) in [(12576, 768, 768), (12576, 768, 3072), (12576, 2304, 768), (12576, 3072, 768), (8192, 2048, 768), (8192, 2048, 64*4)]:
    x = (torch.randn(M, Kd, device="cuda") / 8).bfloat16(); w = (torch.randn(N, Kd, device="cuda") / 8).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(x, w, y, None, True, True))
    print("M=%d N=%d K=%d: %.1f us" % (M, N, Kd, t), flush=True)
