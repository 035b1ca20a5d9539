"""Numerical parity of every DROPOUT instantiation the benchmarked step runs, at its shape and in both dtypes.

The counter-based masks are a pure function of (seed state, call-site id, element index), so a kernel's OWN mask can
be read back from the kernel itself - attention: forward passes whose V holds one-hot columns return the dropped
probabilities 64 keys at a time; LayerNorm: the saved pre-norm z minus the residual is zero exactly where an element
was dropped - and an fp32 torch statement of the reference arithmetic that applies THAT mask then checks the outputs
and all gradients, including the last (partial) key tile, the key-mask / full-mask variants and the two-row-block
instantiations.  Reference arithmetic: modeling_capsbert.py:394-418 (attention probabilities -> dropout -> context),
:431-435 / :494-503 (dense -> dropout -> LayerNorm(x + residual)), transformer.py:212-233 (decoder, p = 0.15).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd import kernels
    return kernels


def _close(a, b, dtype, scale=1.0, what=""):
    rt, at = (2e-4, 2e-4) if dtype == torch.float32 else (3e-2, 3e-2)
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rt, atol=at * scale), f"{what}: max abs err {err}, ref max {b.abs().max().item()}"


# ------------------------------------------------------------------------------------------ attention
def _attn_masks(B, Sq, Sk, mask_kind):
    """The additive masks of the model: key padding (mc:1826-1842) or the decoder's block-causal frame mask (entry.py:114-121)."""
    if mask_kind == 1:
        m01 = torch.ones(B, Sk)
        m01[0, Sk // 2:] = 0
        if B > 1:
            m01[1, Sk - 3:] = 0
        return (1.0 - m01) * -10000.0
    if mask_kind == 2:
        from oracle import shg_ref
        return shg_ref.frame_causal_mask(16, Sq // 16)
    return None


def _probe_dropped_probs(K, q, k, H, mask_kind, md, p, seed, sid):
    """P_dropped [B, H, Sq, Sk] as the kernel computes it: forward passes with V = one-hot columns over 64 keys each."""
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    out = torch.empty(B, H, Sq, Sk, dtype=torch.float32, device=DEV)
    for t0 in range(0, Sk, 64):
        n = min(64, Sk - t0)
        v = torch.zeros(B, Sk, H, 64, dtype=q.dtype, device=DEV)
        idx = torch.arange(n, device=DEV)
        v[:, t0 + idx, :, idx] = 1.0
        o, _ = K.attention_fwd(q, k, v.view(B, Sk, H * 64), H, mask_kind, md, 0.125, p, seed, sid)
        out[:, :, :, t0:t0 + n] = o.view(B, Sq, H, 64)[..., :n].permute(0, 2, 1, 3).float()
    return out


def _attn_case(K, dtype, B, H, Sq, Sk, mask_kind, p, seed_vals=(99, 3), sid=5):
    gen = torch.Generator().manual_seed(Sq * 1000 + Sk + mask_kind)
    q = torch.randn(B, Sq, H * 64, generator=gen).to(dtype).to(DEV)
    k = torch.randn(B, Sk, H * 64, generator=gen).to(dtype).to(DEV)
    v = torch.randn(B, Sk, H * 64, generator=gen).to(dtype).to(DEV)
    do = torch.randn(B, Sq, H * 64, generator=gen).to(dtype).to(DEV)
    mask = _attn_masks(B, Sq, Sk, mask_kind)
    md = mask.to(DEV).contiguous() if mask is not None else None
    seed = torch.tensor(list(seed_vals), dtype=torch.int64, device=DEV)
    return q, k, v, do, md, seed, sid


def _attn_ref_probs(q, k, H, mask_kind, md):
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    qh = q.view(B, Sq, H, 64).transpose(1, 2)
    kh = k.view(B, Sk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * 0.125
    if mask_kind == 1:
        s = s + md[:, None, None, :]
    elif mask_kind == 2:
        s = s + md[None, None]
    return s.softmax(-1)


ATTN_SHAPES = [
    # (B, H, Sq, Sk, mask, p): the step's dropout call sites
    (2, 12, 393, 393, 1, 0.1),      # relation layers (mc:394-418), the benchmarked instantiation: 6 full key tiles + a 9-key tail
    (2, 12, 40, 40, 1, 0.1),        # language layers
    (2, 12, 40, 393, 0, 0.1),       # x-layers, language <- vision
    (2, 12, 393, 40, 1, 0.1),       # x-layers, vision <- language
    (2, 12, 177, 40, 1, 0.1),       # hyper-graph cross encoder
    (2, 12, 128, 128, 2, 0.15),     # relation decoder self-attention (block-causal full mask)
    (2, 12, 48, 48, 2, 0.15),       # action decoder self-attention
    (2, 12, 128, 393, 0, 0.15),     # relation decoder <- memory
    (2, 12, 48, 393, 0, 0.15),      # action decoder <- memory
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Sk,mask_kind,p", ATTN_SHAPES)
def test_attention_with_dropout_matches_fp32_reference_under_the_kernels_own_mask(K, dtype, B, H, Sq, Sk, mask_kind, p):
    q, k, v, do, md, seed, sid = _attn_case(K, dtype, B, H, Sq, Sk, mask_kind, p)
    pd = _probe_dropped_probs(K, q, k, H, mask_kind, md, p, seed, sid)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    probs = _attn_ref_probs(qr, kr, H, mask_kind, md)
    live = probs.detach() > 1e-30                          # masked-out keys have P = 0: their mask bit cannot matter
    keep = (pd != 0) | ~live
    rate = 1.0 - (keep & live).float().sum().item() / live.float().sum().item()
    assert abs(rate - p) < 0.01, rate
    # the kernel's dropped probabilities are the reference's under that mask (this pins the 1 / (1 - p) scale)
    pd_ref = probs * keep / (1.0 - p)
    _close(pd, pd_ref.detach(), dtype, what="dropped probabilities")
    ref_o = (pd_ref @ vr.view(B, Sk, H, 64).transpose(1, 2)).transpose(1, 2).reshape(B, Sq, H * 64)
    ref_o.backward(do.float())
    o, lse = K.attention_fwd(q, k, v, H, mask_kind, md, 0.125, p, seed, sid)
    _close(o, ref_o.detach(), dtype, what="o")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, mask_kind, md, 0.125, p, seed, sid)
    _close(dq, qr.grad, dtype, scale=2.0, what="dq")
    _close(dk, kr.grad, dtype, scale=2.0, what="dk")
    _close(dv, vr.grad, dtype, scale=2.0, what="dv")
    # the backward kernels regenerate the mask on their own: dV with dO = 1 and V-independent -> column sums of P_dropped,
    # exact in fp32 up to summation order - a single element dropped differently in the tail tile shows here
    if dtype == torch.float32:
        ones = torch.ones_like(do)
        K.attention_bwd(q, k, v, o, ones, lse, dq, dk, dv, H, mask_kind, md, 0.125, p, seed, sid)
        got = dv.view(B, Sk, H, 64)[..., 0].permute(0, 2, 1)
        assert torch.allclose(got, pd.sum(2), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,H,Sq,Sk,mask_kind,p", [ATTN_SHAPES[0], ATTN_SHAPES[5], ATTN_SHAPES[8]])
def test_attention_dropout_mask_is_the_same_in_both_dtypes(K, B, H, Sq, Sk, mask_kind, p):
    """fp32 (parity mode) and bf16 (benchmarked mode) draw the same kept set for the same (seed, step, call site)."""
    keeps = []
    for dtype in (torch.float32, torch.bfloat16):
        q, k, v, do, md, seed, sid = _attn_case(K, dtype, B, H, Sq, Sk, mask_kind, p)
        pd = _probe_dropped_probs(K, q, k, H, mask_kind, md, p, seed, sid)
        live = _attn_ref_probs(q.float(), k.float(), H, mask_kind, md) > 1e-30
        keeps.append(((pd != 0) | ~live, live))
    both = keeps[0][1] & keeps[1][1]
    assert torch.equal(keeps[0][0] & both, keeps[1][0] & both)
    # another call site / another step: another mask
    q, k, v, do, md, seed, sid = _attn_case(K, torch.float32, B, H, Sq, Sk, mask_kind, p)
    other = _probe_dropped_probs(K, q, k, H, mask_kind, md, p, seed, sid + 1) != 0
    assert not torch.equal(other & both, keeps[0][0] & both)
    seed2 = seed.clone()
    seed2[1] += 1
    other = _probe_dropped_probs(K, q, k, H, mask_kind, md, p, seed2, sid) != 0
    assert not torch.equal(other & both, keeps[0][0] & both)


@pytest.mark.parametrize("which", ["dq", "dkv"])
def test_attention_backward_two_row_block_instantiations_with_dropout(K, which):
    """The NB = 2 dQ / dK,dV instantiations (off by default, a tuning switch) under dropout give the default kernels' results."""
    from shg_vqa_amd import _lib
    B, H, Sq, Sk, mask_kind, p = ATTN_SHAPES[0]
    q, k, v, do, md, seed, sid = _attn_case(K, torch.bfloat16, B, H, Sq, Sk, mask_kind, p)
    o, lse = K.attention_fwd(q, k, v, H, mask_kind, md, 0.125, p, seed, sid)
    res = {}
    key = "attn_nb_" + which
    try:
        for nb in (1, 2):
            _lib.set_tuning(key, nb)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, mask_kind, md, 0.125, p, seed, sid)
            res[nb] = (dq, dk, dv)
    finally:
        _lib.set_tuning(key, 1)
    for a, b in zip(res[1], res[2]):
        _close(a, b, torch.bfloat16, scale=0.25, what=which)       # same arithmetic per element; the sums associate differently


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols,act,eps,res,p", [(768, 0, 1e-12, True, 0.1),      # BertAttOutput / BertOutput (mc:431-435, 494-503): bench
                                                 (768, 0, 1e-5, True, 0.15),      # decoder norms (transformer.py:212-233)
                                                 (1536, 1, 1e-12, False, 0.1),    # the 1 536-column chunk template, GELU
                                                 (1536, 0, 1e-12, True, 0.1),
                                                 (2048, 0, 1e-5, True, 0.15)])
def test_layernorm_with_dropout_matches_fp32_reference_under_the_kernels_own_mask(K, dtype, cols, act, eps, res, p):
    gen = torch.Generator().manual_seed(cols + act + int(res))
    rows = 1031                                            # not a multiple of the 16 rows of a backward partial
    sign = torch.where(torch.rand(rows, cols, generator=gen) < 0.5, -1.0, 1.0)
    x = (sign * (0.75 + torch.rand(rows, cols, generator=gen))).to(dtype)      # |act(x + b)| stays well away from 0
    r = torch.randn(rows, cols, generator=gen).to(dtype) if res else None
    b = torch.randn(cols, generator=gen) * 0.05
    gam = 1 + 0.1 * torch.randn(cols, generator=gen)
    bet = 0.1 * torch.randn(cols, generator=gen)
    dy = torch.randn(rows, cols, generator=gen).to(dtype)
    seed = torch.tensor([4321, 6], dtype=torch.int64, device=DEV)
    sid = 17
    xd, rd = x.to(DEV), (r.to(DEV) if res else None)
    y, z, mean, rstd = K.ln_fwd(xd, b.to(DEV), rd, gam.to(DEV), bet.to(DEV), eps, act, p, seed, sid)
    # the kernel's own mask: z = dropout(act(x + b)) + residual
    d = z.float().cpu() - (r.float() if res else 0.0)
    keep = d != 0
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    xr = x.float().requires_grad_(True)
    rr = r.float().requires_grad_(True) if res else None
    br, gr, ber = (t.clone().requires_grad_(True) for t in (b, gam, bet))
    u = xr + br
    if act == 1:
        u = F.gelu(u)
    zr = u * keep / (1.0 - p)
    if res:
        zr = zr + rr
    ref = F.layer_norm(zr, (cols,), gr, ber, eps)
    ref.backward(dy.float())
    _close(z, zr.detach(), dtype, what="z")
    _close(y, ref.detach(), dtype, what="y")
    dx, dres, dg, db, dbi = K.ln_bwd(dy.to(DEV), z, xd, b.to(DEV), gam.to(DEV), mean, rstd, act, p, seed, sid, want_dres=res)
    _close(dx, xr.grad, dtype, what="dx")
    assert torch.equal(dx.cpu() != 0, keep), "backward regenerated a different mask"
    if res:
        _close(dres, rr.grad, dtype, what="dres")
    for part, refg, nm in ((dg, gr.grad, "dgamma"), (db, ber.grad, "dbeta"), (dbi, br.grad, "dbias")):
        out = torch.zeros(cols, device=DEV)
        K.colsum_finish(part, out, False)
        _close(out, refg, dtype, scale=math.sqrt(rows), what=nm)


def test_layernorm_dropout_mask_is_the_same_in_both_dtypes(K):
    rows, cols, p = 1031, 768, 0.1
    gen = torch.Generator().manual_seed(3)
    x = (0.75 + torch.rand(rows, cols, generator=gen))
    gam, bet = torch.ones(cols, device=DEV), torch.zeros(cols, device=DEV)
    seed = torch.tensor([4321, 6], dtype=torch.int64, device=DEV)
    keeps = []
    for dtype in (torch.float32, torch.bfloat16):
        _, z, _, _ = K.ln_fwd(x.to(dtype).to(DEV), None, None, gam, bet, 1e-12, 0, p, seed, 23)
        keeps.append(z != 0)
    assert torch.equal(keeps[0], keeps[1])
