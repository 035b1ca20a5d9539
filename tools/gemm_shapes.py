"""Micro-benchmark of shg_gemm on the shapes of one HGQA step (run on the GPU box).
Prints time and TFLOP/s per shape and layout; used to tune tile configurations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K

dev = "cuda"
shapes = [(12576, 768, 768), (12576, 2304, 768), (12576, 3072, 768), (12576, 768, 3072), (12576, 1536, 768), (4096, 768, 768),
          (4096, 1536, 768), (4096, 2048, 768), (4096, 768, 2048), (1536, 768, 768), (1280, 768, 768), (1280, 3072, 768),
          (5664, 768, 768), (5664, 3072, 768)]


def bench(fn, iters=20):
    """GPU time per launch: the launches are captured into a hipGraph so that Python does not pace them."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


for (M, N, Kd) in (shapes if __name__ == "__main__" else []):
    x = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, Kd, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, Kd, device=dev)
    fl = 2.0 * M * N * Kd
    t_f = bench(lambda: K.gemm(x, w, y, None, True, True))
    t_d = bench(lambda: K.gemm(dy, w, dx, None, True, False))
    t_w = bench(lambda: K.gemm(dy, x, dw, None, False, False, accumulate=True))
    t_ref = bench(lambda: torch.matmul(x, w.t())) if not os.environ.get("WGRAD_ONLY") else 0.0
    if os.environ.get("DGRAD_FORMS"):
        # the input-gradient forms of the step: plain, accumulated onto the residual gradient, and with the activation backward
        # + bias-gradient column sums in the epilogue (shg_gemm_dact)
        pre = torch.randn(M, Kd, device=dev).bfloat16()
        db = torch.zeros(Kd, device=dev)
        from shg_vqa_amd import _lib
        res = []
        for side in (0, 1):
            _lib.set_tuning("epilogue_side", side)
            res.append((bench(lambda: K.gemm(dy, w, dx, None, True, False, accumulate=True)), bench(lambda: K.gemm_dact(dy, w, dx, pre, db, 1)),
                        bench(lambda: K.gemm_dact(dy, w, dx, pre, db, 3))))
        t_rd = bench(lambda: torch.matmul(dy, w))
        print("M=%6d N=%5d K=%5d  dgrad (out %d cols, K %d): plain %7.1f us | accumulate %7.1f -> %7.1f us | dact+csum %7.1f -> %7.1f us (loads up front off -> on) | with the stored derivative %7.1f us | torch.mm %7.1f us"
              % (M, N, Kd, Kd, N, t_d, res[0][0], res[1][0], res[0][1], res[1][1], res[1][2], t_rd), flush=True)
        continue
    if os.environ.get("WGRAD_ONLY"):
        print("M=%6d N=%5d K=%5d wgrad %7.1f us %6.0f TF" % (M, N, Kd, t_w, fl / t_w / 1e6), flush=True)
        continue
    print("M=%6d N=%5d K=%5d  fwd %7.1f us %6.0f TF | dgrad %7.1f us %6.0f TF | wgrad %7.1f us %6.0f TF | torch.mm fwd %7.1f us %6.0f TF"
          % (M, N, Kd, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6, t_ref, fl / t_ref / 1e6), flush=True)
