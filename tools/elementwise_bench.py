"""Isolated timing of the row-wise kernels on the r-layer shapes (GB/s against the bytes they must move)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from tools.gemm_shapes import bench
dev = "cuda"
R = 12576
for cols in (768, 3072):
    x = torch.randn(R, cols, device=dev).bfloat16()
    dy = torch.randn(R, cols, device=dev).bfloat16()
    bias = torch.randn(cols, device=dev)
    t = bench(lambda: K.bias_act_fwd(x, bias, 1))
    print("bias_act_fwd  %5d  %7.1f us  %6.0f GB/s" % (cols, t, 2 * x.numel() * 2 / t / 1e3))
    t = bench(lambda: K.bias_act_bwd(x, None, dy, 1, want_dbias=True))
    print("bias_act_bwd  %5d  %7.1f us  %6.0f GB/s" % (cols, t, 3 * x.numel() * 2 / t / 1e3))
    part = torch.randn(K.colsum_partials(R), cols, device=dev)
    out = torch.zeros(cols, device=dev)
    t = bench(lambda: K.colsum_finish(part, out, True))
    print("colsum_finish %5d  %7.1f us  %6.0f GB/s (partials %d)" % (cols, t, part.numel() * 4 / t / 1e3, part.shape[0]))
cols = 768
x = torch.randn(R, cols, device=dev).bfloat16()
res = torch.randn(R, cols, device=dev).bfloat16()
g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev); bias = torch.randn(cols, device=dev)
seed = torch.tensor([1, 0], dtype=torch.int64, device=dev)
for p in (0.0, 0.1):
    y, z, mean, rstd = K.ln_fwd(x, bias, res, g, b, 1e-12, 0, p, seed, 3)
    t = bench(lambda: K.ln_fwd(x, bias, res, g, b, 1e-12, 0, p, seed, 3))
    print("ln_fwd p=%.1f          %7.1f us  %6.0f GB/s" % (p, t, 4 * x.numel() * 2 / t / 1e3))
    t = bench(lambda: K.ln_bwd(x, z, None, bias, g, mean, rstd, 0, p, seed, 3))
    print("ln_bwd p=%.1f          %7.1f us  %6.0f GB/s" % (p, t, 4 * x.numel() * 2 / t / 1e3))

# attention, r-layer shape (self, S=393) and decoder cross shape (Sq=128, Sk=393)
for (Sq, Sk) in ((393, 393), (128, 393)):
    B, H = 32, 12
    q = torch.randn(B, Sq, H * 64, device=dev).bfloat16()
    k = torch.randn(B, Sk, H * 64, device=dev).bfloat16()
    v = torch.randn(B, Sk, H * 64, device=dev).bfloat16()
    for p in (0.0, 0.1):
        o, lse = K.attention_fwd(q, k, v, H, K.MASK_NONE, None, 0.125, p, seed, 5)
        do = torch.randn_like(o)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        fl = 4.0 * B * H * Sq * Sk * 64
        t = bench(lambda: K.attention_fwd(q, k, v, H, K.MASK_NONE, None, 0.125, p, seed, 5))
        print("attn fwd  Sq=%d Sk=%d p=%.1f  %7.1f us  %5.0f TF" % (Sq, Sk, p, t, fl / t / 1e6))
        t = bench(lambda: K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, K.MASK_NONE, None, 0.125, p, seed, 5))
        print("attn bwd  Sq=%d Sk=%d p=%.1f  %7.1f us  %5.0f TF" % (Sq, Sk, p, t, 2.5 * fl / t / 1e6))

# BertAdam over the model's 289 M gradient-receiving parameters
n = 289038112
pa = torch.randn(n, device=dev); ga = torch.randn(n, device=dev) * 1e-3; ma = torch.zeros(n, device=dev); va = torch.zeros(n, device=dev)
sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
gn = torch.tensor([10.0], device=dev); st = torch.zeros(1, dtype=torch.int64, device=dev)
t = bench(lambda: K.bertadam_arena(pa, ga, ma, va, sh, gn, 5.0, 1e-5, 0.1, 10000, step_state=st, bump_step=False), iters=5)
print("bertadam 289M            %7.1f us  %6.0f GB/s" % (t, n * 30 / t / 1e3))
