"""Cost of the GEMM epilogue variants on the step's big shapes (GPU-paced through a hipGraph): plain store, bias, bias + GELU +
saved pre-activation (gemm_act), bf16 accumulate (residual-gradient sums), activation backward + bias column sums (gemm_dact)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from tools.gemm_shapes import bench

dev = "cuda"
seed = torch.zeros(2, dtype=torch.int64, device=dev)
for (M, N, Kd) in ((12576, 3072, 768), (12576, 768, 3072), (12576, 2304, 768), (4096, 2048, 768)):
    x = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()          # forward weight [out, in]
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_plain = bench(lambda: K.gemm(x, w, y, None, True, True))
    t_bias = bench(lambda: K.gemm(x, w, y, b, True, True))
    t_act = bench(lambda: K.gemm_act(x, w, y, b, 1, pre))
    t_actd = bench(lambda: K.gemm_act(x, w, y, b, 1, pre, 0.1, seed, 3))
    t_acc = bench(lambda: K.gemm(x, w, y, None, True, True, accumulate=True))
    # input-gradient form: dx[M, Kd] = dy[M, N] . w[N, Kd]
    dy = torch.randn(M, N, device=dev).bfloat16()
    dx = torch.empty(M, Kd, device=dev, dtype=torch.bfloat16)
    prex = torch.randn(M, Kd, device=dev).bfloat16()
    db = torch.zeros(Kd, device=dev)
    t_dg = bench(lambda: K.gemm(dy, w, dx, None, True, False))
    t_dgacc = bench(lambda: K.gemm(dy, w, dx, None, True, False, accumulate=True))
    t_dact = bench(lambda: K.gemm_dact(dy, w, dx, prex, db, 1))
    t_dactd = bench(lambda: K.gemm_dact(dy, w, dx, prex, db, 1, 0.1, seed, 3))
    t_dact_nob = bench(lambda: K.gemm_dact(dy, w, dx, prex, None, 1))
    t_drelu = bench(lambda: K.gemm_dact(dy, w, dx, prex, db, 2))
    print("M=%6d N=%5d K=%5d fwd: plain %6.1f bias %6.1f act+pre %6.1f +drop %6.1f accum %6.1f | dgrad [M,K]: plain %6.1f accum %6.1f dact %6.1f +drop %6.1f no-dbias %6.1f relu %6.1f us"
          % (M, N, Kd, t_plain, t_bias, t_act, t_actd, t_acc, t_dg, t_dgacc, t_dact, t_dactd, t_dact_nob, t_drelu), flush=True)
