"""Isolated timings of the step's non-GEMM kernels at the benchmark's shapes (events, 20 launches after 3 warm-ups):
attention forward / backward, LayerNorm forward / backward, split-K weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def timeit(fn, n=50, reps=5):
    """us per launch: the median of `reps` blocks of `n` back-to-back launches (single blocks of 20 launches scattered by +-8 % on
    one box: clocks move with what ran before)."""
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


which = sys.argv[1:] or ["attn", "ln", "wgrad"]
if "attn" in which or "attn393" in which:
    for (B, H, Sq, Sk, mk, p) in [(32, 12, 393, 393, K.MASK_KEY, 0.1)] if "attn393" in which else [(32, 12, 393, 393, K.MASK_KEY, 0.0), (32, 12, 393, 393, K.MASK_KEY, 0.1), (32, 12, 128, 128, K.MASK_FULL, 0.15),
                                  (32, 12, 128, 393, K.MASK_NONE, 0.15), (32, 12, 48, 393, K.MASK_NONE, 0.15), (32, 12, 40, 40, K.MASK_KEY, 0.1),
                                  (32, 12, 177, 40, K.MASK_KEY, 0.1), (32, 12, 40, 177, K.MASK_NONE, 0.1)]:
        qkv = torch.randn(B, Sq, 3 * H * 64, device=dev).to(bf)
        kv = torch.randn(B, Sk, 2 * H * 64, device=dev).to(bf) if Sk != Sq else None
        q = qkv[:, :, :H * 64]
        k = qkv[:, :, H * 64:2 * H * 64] if kv is None else kv[:, :, :H * 64]
        v = qkv[:, :, 2 * H * 64:] if kv is None else kv[:, :, H * 64:]
        mask = None
        if mk == K.MASK_KEY:
            mask = torch.zeros(B, Sk, device=dev)
        elif mk == K.MASK_FULL:
            mask = torch.zeros(Sq, Sk, device=dev)
        seed = torch.tensor([1, 2], dtype=torch.int64, device=dev)
        o, lse = K.attention_fwd(q, k, v, H, mk, mask, 0.125, p, seed, 3)
        do = torch.randn_like(o)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        flop = 4.0 * B * H * Sq * Sk * 64
        tf = timeit(lambda: K.attention_fwd(q, k, v, H, mk, mask, 0.125, p, seed, 3))
        tb = timeit(lambda: K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, mk, mask, 0.125, p, seed, 3))
        print("attn B%d H%d %dx%d mask%d p%.2f: fwd %.1f us (%.0f TFLOP/s)  bwd %.1f us (%.0f TFLOP/s)" % (
            B, H, Sq, Sk, mk, p, tf, flop / tf / 1e6, tb, 2.5 * flop / tb / 1e6), flush=True)
if "ln" in which:
    for rows, cols in [(12576, 768), (4096, 768), (1280, 768), (12576, 1536)]:
        x = torch.randn(rows, cols, device=dev).to(bf)
        r = torch.randn(rows, cols, device=dev).to(bf)
        g, b_, bias = torch.ones(cols, device=dev), torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
        seed = torch.tensor([1, 2], dtype=torch.int64, device=dev)
        for p in (0.0, 0.1):
            y, z, mean, rstd = K.ln_fwd(x, bias, r, g, b_, 1e-12, 0, p, seed, 5)
            tf = timeit(lambda: K.ln_fwd(x, bias, r, g, b_, 1e-12, 0, p, seed, 5))
            tb = timeit(lambda: K.ln_bwd(y, z, None, bias, g, mean, rstd, 0, p, seed, 5))
            byf, byb = rows * cols * 2 * 4, rows * cols * 2 * 4
            print("ln %dx%d p%.1f: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (rows, cols, p, tf, byf / tf / 1e6, tb, byb / tb / 1e6), flush=True)
if "wgrad" in which:
    for rows, n_out, n_in in [(12576, 2304, 768), (12576, 768, 768), (12576, 3072, 768), (12576, 768, 3072), (4096, 1536, 768),
                              (4096, 768, 768), (4096, 2048, 768), (4096, 768, 2048), (12576, 1536, 768), (1536, 768, 768), (1280, 3072, 768)]:
        dy = torch.randn(rows, n_out, device=dev).to(bf)
        x = torch.randn(rows, n_in, device=dev).to(bf)
        gw = torch.zeros(n_out, n_in, device=dev)
        t = timeit(lambda: K.gemm(dy, x, gw, None, False, False, accumulate=True))
        print("wgrad rows %d  %d x %d: %.1f us (%.0f TFLOP/s)" % (rows, n_out, n_in, t, 2.0 * rows * n_out * n_in / t / 1e6), flush=True)
if "colsum" in which:
    for rows, cols in [(12576, 768), (12576, 2304), (12576, 3072), (4096, 768), (4096, 2048), (1536, 768), (640, 768)]:
        x = torch.randn(rows, cols, device=dev).to(bf)
        out = torch.zeros(cols, device=dev)
        t = timeit(lambda: K.colsum(x, out, True))
        print("colsum_accumulate %dx%d: %.1f us (%.2f TB/s)" % (rows, cols, t, rows * cols * 2 / t / 1e6), flush=True)
if "hungarian" in which:
    for frames, per, C in [(512, 8, 457), (512, 3, 158), (512, 8, 158), (4096, 8, 457)]:
        logits = torch.randn(frames, per, C, device=dev).to(bf)
        tgt = torch.randint(0, C - 1, (frames, per), device=dev)
        lens = torch.randint(0, per + 1, (frames,), device=dev, dtype=torch.int32)
        t = timeit(lambda: K.hungarian_per_frame(logits, tgt, lens))
        print("hungarian_per_frame %d frames x %d queries x %d classes: %.1f us" % (frames, per, C, t), flush=True)
