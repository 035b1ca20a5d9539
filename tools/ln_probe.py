"""LayerNorm forward at the r-layer shape against plain streaming kernels of the same traffic (hipGraph-timed, no host pacing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from gemm_shapes import bench
dev = "cuda"
bf = torch.bfloat16
for rows, cols in [(12576, 768), (4096, 768)]:
    x = torch.randn(rows, cols, device=dev).to(bf)
    r = torch.randn(rows, cols, device=dev).to(bf)
    g, b_, bias = torch.ones(cols, device=dev), torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    seed = torch.tensor([1, 2], dtype=torch.int64, device=dev)
    y = torch.empty_like(x); z = torch.empty_like(x)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    from shg_vqa_amd import _lib
    def ln(res, zz, p):
        _lib.call("shg_bias_act_drop_res_ln_fwd", x.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, g.data_ptr(), b_.data_ptr(),
                  y.data_ptr(), zz.data_ptr() if zz is not None else None, mean.data_ptr(), rstd.data_ptr(), K._dt(x), rows, cols, 0, 1e-12, p,
                  seed.data_ptr(), 5, K._stream())
    mb = rows * cols * 2 / 1e6
    for name, fn, n in [("ln res+z p0.1", lambda: ln(r, z, 0.1), 4), ("ln res+z p0", lambda: ln(r, z, 0.0), 4), ("ln res, no z", lambda: ln(r, None, 0.0), 3),
                        ("ln plain", lambda: ln(None, None, 0.0), 2), ("torch add(x,r,out=y)", lambda: torch.add(x, r, out=y), 3),
                        ("torch copy", lambda: y.copy_(x), 2)]:
        t = bench(fn)
        print("%dx%d %-24s %.1f us  (%.2f TB/s)" % (rows, cols, name, t, n * mb / t), flush=True)
