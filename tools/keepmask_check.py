"""Decodes the attention forward's keep-mask buffer (csrc/attention.hip layout) and compares it with the kept set read back from the
kernel's own outputs (one-hot V probes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from shg_vqa_amd import kernels as K
import test_dropout_parity_gpu as T

dev = "cuda"
for dtype in (torch.float32, torch.bfloat16):
    for (B, H, Sq, Sk, mk, p) in [(2, 12, 393, 393, 1, 0.1), (2, 12, 128, 128, 2, 0.15), (2, 12, 40, 40, 1, 0.1)]:
        q, k, v, do, md, seed, sid = T._attn_case(K, dtype, B, H, Sq, Sk, mk, p)
        pd = T._probe_dropped_probs(K, q, k, H, mk, md, p, seed, sid)
        live = T._attn_ref_probs(q.float(), k.float(), H, mk, md) > 1e-30
        o, lse = K.attention_fwd(q, k, v, H, mk, md, 0.125, p, seed, sid)
        words = lse._shg_keep.cpu()
        nq16, nkt = (Sq + 15) // 16, (Sk + 63) // 64
        w = words.view(B * H, nq16, nkt, 16)
        dense = torch.zeros(B * H, nq16 * 16, nkt * 64, dtype=torch.bool)
        for e in range(16):
            kt, r = e // 4, e % 4
            for g in range(4):
                for li in range(16):
                    bit = (w[:, :, :, e] >> (16 * g + li)) & 1
                    dense[:, li::16, :][:, :, (16 * kt + 4 * g + r)::64] = bit.bool()
        dense = dense[:, :Sq, :Sk].view(B, H, Sq, Sk)
        kept = (pd != 0).cpu()
        lv = live.cpu()
        bad = (dense != kept) & lv
        print(dtype, (B, H, Sq, Sk, mk), "mismatches among live elements:", int(bad.sum()), "of", int(lv.sum()))
        if bad.any():
            idx = bad.nonzero()
            print("   first:", idx[:5].tolist(), " per key tile:", [int(bad[..., t * 64:(t + 1) * 64].sum()) for t in range(nkt)],
                  " per query block (first 8):", [int(bad[:, :, t * 16:(t + 1) * 16].sum()) for t in range(min(nq16, 8))])
