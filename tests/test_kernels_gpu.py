"""Parity of each HIP kernel (through the C ABI) with the CPU oracle / a plain fp32 PyTorch statement
of the same op.  Needs an MI355X: run with `-m gpu`."""
import math
import os

import numpy as np
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


def _tol(dtype):
    # fp32: parity bar of north_star (<= 1e-3 rel); bf16: storage rounding 2^-8 on O(1) values
    return (2e-4, 2e-4) if dtype == torch.float32 else (3e-2, 3e-2)


def _assert_close(a, b, dtype, scale=1.0):
    rt, at = _tol(dtype)
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rt, atol=at * scale), f"max abs err {err}, ref max {b.abs().max().item()}"


# ------------------------------------------------------------------------------------------ LSAP
def test_lsap_batched_bit_exact_vs_scipy_golden(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "lsap_scipy.npz"))
    shape = g["shape"]
    for R in (8, 3):
        sel = np.where((shape[:, 0] == R) & (shape[:, 1] <= R) & (shape[:, 1] >= 0))[0]
        assert len(sel) > 500
        cost = torch.from_numpy(g["cost"][sel][:, :R, :R].copy()).to(DEV)
        ncols = torch.from_numpy(shape[sel, 1].astype(np.int32)).to(DEV)
        rows, cols = K.lsap_batched(cost.contiguous(), ncols)
        rows, cols = rows.cpu().numpy(), cols.cpu().numpy()
        exp_r, exp_c = g["rows"][sel][:, :R], g["cols"][sel][:, :R]
        assert np.array_equal(rows, exp_r)
        assert np.array_equal(cols, exp_c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(8, 457), (3, 158), (8, 564)])
def test_hungarian_per_frame_matches_oracle(K, dtype, R, C):
    from oracle import shg_ref
    gen = torch.Generator().manual_seed(R * 1000 + C)
    B, T = 6, 16
    logits = (torch.randn(B, T * R, C, generator=gen) * 2).to(dtype)
    tgt = torch.zeros(B * T, R, dtype=torch.int64)
    lens = torch.randint(0, R + 1, (B * T,), generator=gen).to(torch.int32)
    lens[0], lens[1] = 0, R
    frames = []
    for n in range(B * T):
        k = int(lens[n])
        ids = torch.randperm(C - 1, generator=gen)[:k] + 1
        if k >= 2 and n % 5 == 0:
            ids[1] = ids[0]                      # duplicated class in a frame -> exact ties
        tgt[n, :k] = ids
        frames.append(ids)
    exp = shg_ref.hungarian_per_frame(logits.float(), frames, clip_len=T)
    oq, ot, grid = K.hungarian_per_frame(logits.view(B * T, R, C).to(DEV), tgt.to(DEV), lens.to(DEV))
    oq, ot, grid = oq.cpu(), ot.cpu(), grid.cpu()
    exp_grid = shg_ref.set_target_grid(frames, exp, B * T, R)
    for n, (qi, ti) in enumerate(exp):
        k = len(qi)
        assert torch.equal(oq[n, :k], qi) and torch.equal(ot[n, :k], ti), n
        assert (oq[n, k:] == -1).all() and (ot[n, k:] == -1).all()
    assert torch.equal(grid, exp_grid)


# ------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weighted_ce_fwd_bwd(K, dtype):
    gen = torch.Generator().manual_seed(3)
    rows, C = 1000, 457
    logits = (torch.randn(rows, C, generator=gen) * 3).to(dtype)
    target = torch.randint(0, C, (rows,), generator=gen)
    target[::3] = 0
    w = torch.ones(C)
    w[0] = 0.1
    ref_in = logits.float().requires_grad_(True)
    ref = F.cross_entropy(ref_in, target, w)
    ref.backward()
    stats, sums = K.weighted_ce_fwd(logits.to(DEV), target.to(DEV), w.to(DEV))
    loss = (sums[0] / sums[1]).cpu()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item()) + 1e-5
    d = K.weighted_ce_bwd(logits.to(DEV), target.to(DEV), w.to(DEV), stats, sums)
    _assert_close(d, ref_in.grad, dtype, scale=1e-3)
    # class_error bookkeeping: matched (non-background) rows and how many are top-1 correct
    nb = target != 0
    correct = (logits.float().argmax(-1) == target) & nb
    assert int(sums[3].item()) == int(nb.sum()) and int(sums[2].item()) == int(correct.sum())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bce_logits(K, dtype):
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(32, 171, generator=gen) * 2).to(dtype)
    y = torch.zeros(32, 171)
    y[torch.arange(32), torch.randint(0, 171, (32,), generator=gen)] = 1
    xr = x.float().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(xr, y) * 171
    ref.backward()
    loss, d = K.bce_logits(x.to(DEV), y.to(DEV))
    assert abs(loss.item() - ref.item()) < 1e-4 * ref.item()
    _assert_close(d, xr.grad, dtype, scale=1e-2)


# ------------------------------------------------------------------------------------------ epilogues
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bias_act_fwd_bwd(K, dtype, act):
    gen = torch.Generator().manual_seed(5 + act)
    rows, cols = 777, 3072
    x = torch.randn(rows, cols, generator=gen).to(dtype)
    b = torch.randn(cols, generator=gen) * 0.1
    dy = torch.randn(rows, cols, generator=gen).to(dtype)
    xr = x.float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    u = xr + br
    ref = u if act == 0 else (F.gelu(u) if act == 1 else F.relu(u))
    ref.backward(dy.float())
    y = K.bias_act_fwd(x.to(DEV), b.to(DEV), act)
    _assert_close(y, ref.detach(), dtype)
    dx, part = K.bias_act_bwd(x.to(DEV), b.to(DEV), dy.to(DEV), act)
    _assert_close(dx, xr.grad, dtype)
    dbias = torch.zeros(cols, device=DEV)
    K.colsum_finish(part, dbias, False)
    _assert_close(dbias, br.grad, dtype, scale=math.sqrt(rows))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols,act,eps,res", [(768, 0, 1e-12, True), (1536, 1, 1e-12, False), (768, 0, 1e-5, True),
                                               (2048, 0, 1e-5, True)])
def test_layernorm_fused_fwd_bwd(K, dtype, cols, act, eps, res):
    if dtype == torch.float32 and cols > 2048:
        pytest.skip("fp32 limit")
    gen = torch.Generator().manual_seed(cols + act)
    rows = 515
    x = torch.randn(rows, cols, generator=gen).to(dtype)
    r = torch.randn(rows, cols, generator=gen).to(dtype) if res else None
    b = torch.randn(cols, generator=gen) * 0.1
    gam = 1 + 0.1 * torch.randn(cols, generator=gen)
    bet = 0.1 * torch.randn(cols, generator=gen)
    dy = torch.randn(rows, cols, generator=gen).to(dtype)
    xr = x.float().requires_grad_(True)
    rr = r.float().requires_grad_(True) if res else None
    br, gr, ber = (t.clone().requires_grad_(True) for t in (b, gam, bet))
    u = xr + br
    if act == 1:
        u = F.gelu(u)
    z = u + rr if res else u
    ref = F.layer_norm(z, (cols,), gr, ber, eps)
    ref.backward(dy.float())
    y, zz, mean, rstd = K.ln_fwd(x.to(DEV), b.to(DEV), r.to(DEV) if res else None, gam.to(DEV), bet.to(DEV), eps, act)
    _assert_close(y, ref.detach(), dtype)
    dx, dres, dg, db, dbi = K.ln_bwd(dy.to(DEV), zz, x.to(DEV), b.to(DEV), gam.to(DEV), mean, rstd, act, want_dres=res)
    _assert_close(dx, xr.grad, dtype)
    if res:
        _assert_close(dres, rr.grad, dtype)
    for part, refg in ((dg, gr.grad), (db, ber.grad), (dbi, br.grad)):
        out = torch.zeros(cols, device=DEV)
        K.colsum_finish(part, out, False)
        _assert_close(out, refg, dtype, scale=math.sqrt(rows))


def test_layernorm_zero_variance_row_eps_1e12(K):
    """an exactly-zero row under eps=1e-12 gives rstd = 1e6 and y = beta."""
    x = torch.zeros(4, 768, device=DEV)
    x[1] = torch.randn(768, device=DEV)
    gam = torch.full((768,), 1.5, device=DEV)
    bet = torch.full((768,), 0.25, device=DEV)
    y, z, mean, rstd = K.ln_fwd(x, None, None, gam, bet, 1e-12)
    assert abs(rstd[0].item() - 1e6) / 1e6 < 1e-3
    assert torch.allclose(y[0], bet)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_is_replayed_by_backward_and_has_the_right_rate(K, dtype):
    rows, cols, p = 512, 768, 0.1
    seed = torch.tensor([1234, 7], dtype=torch.int64, device=DEV)
    x = (torch.rand(rows, cols, device=DEV) + 0.5).to(dtype)
    y = K.bias_act_fwd(x, None, 0, p, seed, 11)
    keep = (y != 0)
    rate = 1 - keep.float().mean().item()
    assert abs(rate - p) < 0.01
    assert torch.allclose(y[keep].float(), (x[keep].float() / (1 - p)), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    dy = torch.ones_like(x)
    dx, _ = K.bias_act_bwd(x, None, dy, 0, p, seed, 11, want_dbias=False)
    assert torch.equal(dx != 0, keep)
    y2 = K.bias_act_fwd(x, None, 0, p, seed, 12)          # another call site -> another mask
    assert not torch.equal(y2 != 0, keep)
    seed2 = seed.clone()
    K.add_i64(seed2[1:], 1)                                # next step -> another mask
    y3 = K.bias_act_fwd(x, None, 0, p, seed2, 11)
    assert not torch.equal(y3 != 0, keep)


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H, mask_kind, mask, scale):
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    qh = q.view(B, Sq, H, 64).transpose(1, 2)
    kh = k.view(B, Sk, H, 64).transpose(1, 2)
    vh = v.view(B, Sk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if mask_kind == 1:
        s = s + mask[:, None, None, :]
    elif mask_kind == 2:
        s = s + mask[None, None]
    p = s.softmax(-1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, H * 64), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Sq,Sk,mask_kind", [(40, 40, 1), (393, 393, 1), (40, 393, 1), (393, 40, 1), (177, 40, 0),
                                              (40, 177, 1), (128, 128, 2), (48, 48, 2), (128, 393, 0), (48, 393, 0)])
def test_attention_fwd_bwd(K, dtype, Sq, Sk, mask_kind):
    from oracle import shg_ref
    gen = torch.Generator().manual_seed(Sq * 1000 + Sk)
    B, H = 2, 3
    # fused-projection layout: q/k/v are column slices of wider buffers (strided views)
    qbuf = torch.randn(B, Sq, 3 * H * 64, generator=gen).to(dtype)
    kvbuf = torch.randn(B, Sk, 2 * H * 64, generator=gen).to(dtype)
    q, k, v = qbuf[:, :, H * 64:2 * H * 64], kvbuf[:, :, :H * 64], kvbuf[:, :, H * 64:]
    do = torch.randn(B, Sq, H * 64, generator=gen).to(dtype)
    mask = None
    if mask_kind == 1:
        m01 = torch.ones(B, Sk)
        m01[0, Sk // 2:] = 0
        m01[1, Sk - 3:] = 0
        mask = (1.0 - m01) * -10000.0
    elif mask_kind == 2:
        per = Sq // 16
        mask = shg_ref.frame_causal_mask(16, per)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref_o, ref_lse = _attn_ref(qr, kr, vr, H, mask_kind, mask, 0.125)
    ref_o.backward(do.float())
    qd, kd, vd = qbuf.to(DEV)[:, :, H * 64:2 * H * 64], kvbuf.to(DEV)[:, :, :H * 64], kvbuf.to(DEV)[:, :, H * 64:]
    md = mask.to(DEV).contiguous() if mask is not None else None
    o, lse = K.attention_fwd(qd, kd, vd, H, mask_kind, md, 0.125)
    _assert_close(o, ref_o.detach(), dtype)
    _assert_close(lse, ref_lse.detach(), dtype)
    dqkv = torch.zeros(B, Sq, 3 * H * 64, dtype=dtype, device=DEV)
    dkv = torch.zeros(B, Sk, 2 * H * 64, dtype=dtype, device=DEV)
    K.attention_bwd(qd, kd, vd, o, do.to(DEV), lse, dqkv[:, :, H * 64:2 * H * 64], dkv[:, :, :H * 64], dkv[:, :, H * 64:],
                    H, mask_kind, md, 0.125)
    _assert_close(dqkv[:, :, H * 64:2 * H * 64], qr.grad, dtype, scale=2.0)
    _assert_close(dkv[:, :, :H * 64], kr.grad, dtype, scale=2.0)
    _assert_close(dkv[:, :, H * 64:], vr.grad, dtype, scale=2.0)
    assert (dqkv[:, :, :H * 64] == 0).all() and (dqkv[:, :, 2 * H * 64:] == 0).all()   # nothing outside the views


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Sq,Sk,mask_kind,p", [(393, 393, 1, 0.1), (128, 128, 2, 0.0), (48, 393, 0, 0.15), (40, 177, 1, 0.0)])
def test_attention_bwd_adds_projection_bias_gradients(K, dtype, Sq, Sk, mask_kind, p):
    """shg_attention_bwd's dbias_q / dbias_k / dbias_v: the column sums of dq / dk / dv over all (batch, position) rows, ADDED to
    what the vectors hold (the bias gradients of the q / k / v projections, mc:375-380; nn.MultiheadAttention.in_proj_bias) -
    including the last partial row blocks (393 = 24 x 16 + 9 rows: the clamped duplicate rows must not be counted)."""
    gen = torch.Generator().manual_seed(Sq + 7 * Sk)
    B, H = 3, 4
    q = torch.randn(B, Sq, H * 64, generator=gen).to(dtype).to(DEV)
    k = torch.randn(B, Sk, H * 64, generator=gen).to(dtype).to(DEV)
    v = torch.randn(B, Sk, H * 64, generator=gen).to(dtype).to(DEV)
    do = torch.randn(B, Sq, H * 64, generator=gen).to(dtype).to(DEV)
    mask = None
    if mask_kind == 1:
        m01 = torch.ones(B, Sk)
        m01[0, Sk // 2:] = 0
        mask = ((1.0 - m01) * -10000.0).to(DEV)
    elif mask_kind == 2:
        from oracle import shg_ref
        mask = shg_ref.frame_causal_mask(16, Sq // 16).to(DEV).contiguous()
    seed = torch.tensor([5, 9], dtype=torch.int64, device=DEV)
    o, lse = K.attention_fwd(q, k, v, H, mask_kind, mask, 0.125, p, seed, 4)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    init = [torch.randn(H * 64, generator=gen).to(DEV) for _ in range(3)]
    db = [t.clone() for t in init]
    K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, mask_kind, mask, 0.125, p, seed, 4, dbias=tuple(db))
    ref = K.attention_bwd(q, k, v, o, do, lse, torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), H, mask_kind, mask, 0.125, p, seed, 4)
    for got, base, grad, rows in ((db[0], init[0], dq, B * Sq), (db[1], init[1], dk, B * Sk), (db[2], init[2], dv, B * Sk)):
        exp = base.double() + grad.double().sum((0, 1))                 # (the kernel sums the fp32 values before their rounding)
        tol = (2e-4 if dtype == torch.float32 else 2e-2) * math.sqrt(rows) * max(1.0, grad.float().abs().max().item())
        assert torch.allclose(got.double(), exp, rtol=2e-3, atol=tol), (got.double() - exp).abs().max().item()


def test_attention_operand_layout_with_integer_data(K):
    """A = I style check with ASYMMETRIC data: exact small integers expose a transposed fragment."""
    B, H, S = 1, 1, 64
    q = torch.zeros(B, S, 64)
    k = torch.zeros(B, S, 64)
    for i in range(S):
        q[0, i, i] = 1.0                      # q_i = e_i  ->  score(i, j) = k_j[i]
        k[0, i, (i * 7 + 3) % 64] = 20.0      # key j has its spike at column (7j+3)%64: asymmetric
    v = torch.arange(S * 64, dtype=torch.float32).view(1, S, 64) / 64.0
    ref_o, _ = _attn_ref(q, k, v, 1, 0, None, 1.0)
    for dtype in (torch.float32, torch.bfloat16):
        o, _ = K.attention_fwd(q.to(DEV).to(dtype), k.to(DEV).to(dtype), v.to(DEV).to(dtype), 1, 0, None, 1.0)
        assert torch.allclose(o.float().cpu(), ref_o, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=0.3 if dtype == torch.bfloat16 else 1e-3)


def test_attention_dropout_consistent_between_fwd_and_bwd(K):
    """With v = identity-like columns the kept set is visible in o; bwd must use the same mask:
    check d(sum o)/dv equals the column sums of the dropped probabilities."""
    B, H, S = 1, 2, 40
    gen = torch.Generator().manual_seed(9)
    q = torch.randn(B, S, H * 64, generator=gen).to(DEV)
    k = torch.randn(B, S, H * 64, generator=gen).to(DEV)
    v = torch.zeros(B, S, H * 64, device=DEV)
    for h in range(H):
        for j in range(S):
            v[0, j, h * 64 + j] = 1.0                 # o[b,i,h*64+j] = P_dropped[i,j]
    seed = torch.tensor([99, 3], dtype=torch.int64, device=DEV)
    p = 0.3
    o, lse = K.attention_fwd(q, k, v, H, 0, None, 0.125, p, seed, 5)
    pd = o.view(B, S, H, 64)[..., :S]                 # [B, i, h, j]
    rate = (pd == 0).float().mean().item()
    assert abs(rate - p) < 0.05
    do = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, H, 0, None, 0.125, p, seed, 5)
    # dv[j, h*64 + d] = sum_i P_dropped[i, j] * do[i, d] = column sum of P_dropped
    exp = pd.sum(1)                                   # [B, h, j]
    got = dv.view(B, S, H, 64)[..., 0].permute(0, 2, 1)
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ optimiser
def test_bertadam_arena_matches_reference_golden(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "bertadam_steps.npz"))
    shapes = [g[f"init{i}"].shape for i in range(3)]
    sizes = [int(np.prod(s)) for s in shapes]
    pad = [((s + 3) // 4) * 4 for s in sizes]           # 16-byte aligned segments
    offs = np.cumsum([0] + pad)
    n = int(offs[-1])
    param = torch.zeros(n, device=DEV)
    for i in range(3):
        param[offs[i]:offs[i] + sizes[i]] = torch.from_numpy(g[f"init{i}"].reshape(-1)).to(DEV)
    m, v = torch.zeros_like(param), torch.zeros_like(param)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for s in range(4):
        grad = torch.zeros(n, device=DEV)
        for i in range(3):
            grad[offs[i]:offs[i] + sizes[i]] = torch.from_numpy(g[f"grad{s}_{i}"].reshape(-1)).to(DEV)
        norm = K.grad_norm(grad)
        assert abs(norm.item() - g["norms"][s]) < 1e-5 * g["norms"][s]
        K.bertadam_arena(param, grad, m, v, shadow, norm, 5.0, 1e-3, 0.1, 20, step)
        for i in range(3):
            got = param[offs[i]:offs[i] + sizes[i]].cpu().numpy().reshape(shapes[i])
            assert np.allclose(got, g[f"after{s}_{i}"], rtol=2e-6, atol=1e-7), (s, i)
        assert torch.equal(shadow, param.to(torch.bfloat16))
    assert step.item() == 4


def test_grad_norm_large(K):
    x = torch.randn(10_000_003, device=DEV)
    buf = torch.empty(10_000_004, device=DEV)[:10_000_003]
    buf.copy_(x)
    n = K.grad_norm(buf)
    assert abs(n.item() - x.double().norm().item()) < 1e-5 * x.double().norm().item()


# ------------------------------------------------------------------------------------------ GEMM / conv
def _gemm_case(K, dtype, akm, bkm, M, N, Kd, accumulate, out_dtype, with_bias):
    gen = torch.Generator().manual_seed(M + 7 * N + 13 * Kd)
    a = torch.randn((M, Kd) if akm else (Kd, M), generator=gen).to(dtype)
    b = torch.randn((N, Kd) if bkm else (Kd, N), generator=gen).to(dtype)
    bias = torch.randn(N, generator=gen) if with_bias else None
    A = a.float() if akm else a.float().t()
    Bm = b.float().t() if bkm else b.float()
    ref = A @ Bm
    if with_bias:
        ref = ref + bias
    out = torch.randn(M, N, generator=gen).to(out_dtype).to(DEV) if accumulate else torch.empty(M, N, dtype=out_dtype, device=DEV)
    if accumulate:
        ref = ref + out.float().cpu()
    K.gemm(a.to(DEV), b.to(DEV), out, bias.to(DEV) if with_bias else None, akm, bkm, accumulate)
    rt = 1e-4 if dtype == torch.float32 else 2e-2
    got = out.float().cpu()
    err = (got - ref).abs().max().item()
    assert torch.allclose(got, ref, rtol=rt, atol=rt * math.sqrt(Kd)), f"max err {err} (ref max {ref.abs().max().item()})"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_layouts_and_tails(K, dtype, akm, bkm):
    # the non-contracted extent of a K-strided operand is its contiguous dimension: keep it 16-byte
    # aligned (the model only ever has hidden sizes there); the contracted extent may be anything
    sizes = [(1280, 768, 768), (784, 3072, 776), (136, 456, 200), (64, 8, 64), (8, 136, 1000)]
    if not akm and not bkm:
        sizes.append((264, 72, 1001))          # wgrad form: the contracted extent (rows) may be ragged
    for (M, N, Kd) in sizes:
        _gemm_case(K, dtype, akm, bkm, M, N, Kd, False, torch.float32, with_bias=(M % 16 == 0))
        if dtype == torch.bfloat16:
            _gemm_case(K, dtype, akm, bkm, M, N, Kd, False, torch.bfloat16, with_bias=True)
    _gemm_case(K, dtype, akm, bkm, 256, 264, 512, True, torch.float32, with_bias=False)


def test_gemm_fragment_layout_with_integer_data(K):
    """Asymmetric small-integer operands: exact in bf16, any swapped fragment shows up as a wrong entry."""
    M, N, Kd = 128, 128, 64
    a = (torch.arange(M * Kd).view(M, Kd) % 7 - 3).float()
    b = (torch.arange(N * Kd).view(N, Kd) % 5 - 2).float()
    b[3, 5] = 9
    ref = a @ b.t()
    for dtype in (torch.float32, torch.bfloat16):
        for akm in (True, False):
            for bkm in (True, False):
                aa = (a if akm else a.t().contiguous()).to(dtype).to(DEV)
                bb = (b if bkm else b.t().contiguous()).to(dtype).to(DEV)
                out = torch.empty(M, N, device=DEV)
                K.gemm(aa, bb, out, None, akm, bkm)
                assert torch.equal(out.cpu(), ref), (dtype, akm, bkm)


def test_gemm_rejects_bad_arguments(K):
    from shg_vqa_amd._lib import ShgError
    a = torch.zeros(16, 12, device=DEV)
    with pytest.raises((ShgError, ValueError)):
        K.gemm(torch.zeros(16, 10, device=DEV), torch.zeros(8, 10, device=DEV), torch.zeros(16, 8, device=DEV))
    with pytest.raises((ShgError, ValueError)):
        K.gemm(a, torch.zeros(8, 16, device=DEV), torch.zeros(16, 8, device=DEV))   # K mismatch


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 16, 7, 7, 128, 64), (1, 12, 7, 7, 64, 72), (3, 6, 5, 4, 64, 8),
                                              (18, 16, 7, 7, 64, 768),      # bf16: forward on the 8-phase kernel (126 tiles)
                                              (4, 20, 7, 7, 256, 768)])     # bf16: weight gradient on the 8-phase kernel
def test_conv3d_k533_fwd_wgrad_vs_torch(K, dtype, B, T, H, W, Cin, Cout):
    if Cout == 768 and dtype == torch.float32:
        pytest.skip("large shapes exist for the bf16-only 8-phase kernel")
    gen = torch.Generator().manual_seed(B * 100 + Cin)
    x = torch.randn(B, Cin, T, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 5, 3, 3, generator=gen) / math.sqrt(Cin * 45)
    bias = torch.randn(Cout, generator=gen) * 0.1
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    xr = xq.clone()
    wr = wq.clone().requires_grad_(True)
    pre = F.conv3d(F.pad(xr, (1, 1, 1, 1)), wr, bias)
    ref = F.gelu(pre)
    x_cl = K.ncdhw_to_padded_cl(x.to(DEV), dtype)
    # layout conversion is exact
    exp_cl = F.pad(xq, (1, 1, 1, 1)).permute(0, 2, 3, 4, 1).contiguous()
    assert torch.equal(x_cl.float().cpu(), exp_cl)
    w_cl = wq.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV)
    y = K.conv3d_k533_fwd(x_cl, w_cl, bias.to(DEV), act=1)
    rt = 2e-4 if dtype == torch.float32 else 3e-2
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    assert torch.allclose(got, ref.detach(), rtol=rt, atol=rt), (got - ref.detach()).abs().max()
    yp = K.conv3d_k533_fwd(x_cl, w_cl, bias.to(DEV), act=1, pad_out=True)
    assert torch.equal(yp[:, :, 1:-1, 1:-1].float().cpu(), y.float().cpu())
    assert (yp[:, :, 0] == 0).all() and (yp[:, :, :, 0] == 0).all() and (yp[:, :, -1] == 0).all()
    dy = torch.randn(ref.shape, generator=gen).to(dtype).float()
    pre.backward(dy)
    dw = torch.zeros(Cout, 5, 3, 3, Cin, device=DEV)
    K.conv3d_k533_wgrad(x_cl, dy.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV), dw)
    exp_dw = wr.grad.permute(0, 2, 3, 4, 1)
    assert torch.allclose(dw.cpu(), exp_dw, rtol=rt, atol=rt * exp_dw.abs().max().item()), (dw.cpu() - exp_dw).abs().max()
    dw2 = dw.clone()
    K.conv3d_k533_wgrad(x_cl, dy.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV), dw2, accumulate=True)
    assert torch.allclose(dw2, 2 * dw, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_wgrad_overwrite_form_on_the_small_kernels(K, dtype):
    """shg_conv3d_k533_wgrad_sumsq where the 8-phase kernel is not used (fp32 parity mode, small problems): same gradient as the
    plain call into a dirty destination, sum of squares by the pass over the finished rows."""
    B, T, H, W, Cin, Cout = 2, 7, 5, 5, 64, 48
    gen = torch.Generator().manual_seed(5)
    x = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV, dtype=dtype)
    x[:, :, 1:-1, 1:-1] = torch.randn(B, T, H, W, Cin, generator=gen).to(DEV).to(dtype)
    dy = torch.randn(B, T - 4, H, W, Cout, generator=gen).to(DEV).to(dtype)
    ref = torch.zeros(Cout, 5, 3, 3, Cin, device=DEV)
    K.conv3d_k533_wgrad(x, dy, ref, accumulate=True)          # the plain scheme: accumulate into a zeroed gradient
    dw = torch.full_like(ref, 11.0)
    ss = torch.full((1,), 2.5, dtype=torch.float64, device=DEV)
    K.conv3d_k533_wgrad_sumsq(x, dy, dw, ss, c0=16, cn=32)
    assert torch.equal(dw[16:], ref[16:]) and bool((dw[:16] == 11.0).all())
    want = 2.5 + (ref[16:].double() ** 2).sum().item()
    assert abs(ss.item() - want) <= 1e-6 * want, (ss.item(), want)


def test_conv3d_forward_streamk_is_exact_on_integer_data(K):
    """Conv forward on the stream-K split of the 8-phase kernel (gemm.hip: StreamK; every tile's K-tiles are split between a
    head workgroup, which owns the tile, and one or two tail workgroups).  Small-integer data is exact in fp32, so a lost,
    doubled or stale partial sum shows; repeated launches check that the flags come back to zero; a second stream gets its
    own partial-sum slots."""
    from shg_vqa_amd import _lib
    B, T, H, W, Cin, Cout = 19, 16, 7, 7, 256, 768          # 44 x 3 tiles: 16-17 per XCD (heads), 15-16 tail workgroups
    gen = torch.Generator().manual_seed(3)
    x_cl = torch.zeros(B, T, H + 2, W + 2, Cin)
    x_cl[:, :, 1:-1, 1:-1] = torch.randint(-2, 3, (B, T, H, W, Cin), generator=gen).float()
    w_cl = torch.randint(-1, 2, (Cout, 5, 3, 3, Cin), generator=gen).float()
    bias = torch.randint(-3, 4, (Cout,), generator=gen).float()
    xd, wd = x_cl.to(DEV), w_cl.to(DEV)
    # exact reference: explicit im2col (windows of the padded channels-last input) x fp32 GEMM on integers
    To = T - 4
    sB, sT, sH, sW, sC = xd.stride()
    win = xd.as_strided((B, To, H, W, 5, 3, 3, Cin), (sB, sT, sH, sW, sT, sH, sW, sC)).reshape(B * To * H * W, 45 * Cin)
    ref = (win @ wd.reshape(Cout, -1).t() + bias.to(DEV)).bfloat16().float().cpu().view(B, To, H, W, Cout)
    del win
    xb, wb, bd = xd.bfloat16(), wd.bfloat16(), bias.to(DEV)
    before = int(_lib.lib().shg_gemm_streamk_launches())
    for it in range(3):
        y = K.conv3d_k533_fwd(xb, wb, bd, act=0)
        torch.cuda.synchronize()
        assert torch.equal(y.float().cpu(), ref), it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y2 = K.conv3d_k533_fwd(xb, wb, bd, act=0)
    y1 = K.conv3d_k533_fwd(xb, wb, bd, act=0)
    torch.cuda.synchronize()
    assert torch.equal(y1.float().cpu(), ref) and torch.equal(y2.float().cpu(), ref)
    assert int(_lib.lib().shg_gemm_streamk_launches()) == before + 5, "the stream-K path was not taken"


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (2, 12, 7, 7, 64, 128)), (torch.bfloat16, (2, 12, 7, 7, 64, 128)),
                                         (torch.bfloat16, (18, 12, 7, 7, 768, 64))])      # 8-phase kernel: 42 x 3 tiles
def test_conv3d_k533_dgrad_vs_torch(K, dtype, shape):
    gen = torch.Generator().manual_seed(21)
    B, T, H, W, Cin, Cout = shape
    x = torch.randn(B, Cin, T, H, W, generator=gen).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 5, 3, 3, generator=gen) / math.sqrt(Cin * 45)).to(dtype).float()
    y = F.conv3d(F.pad(x, (1, 1, 1, 1)), w)
    dy = torch.randn(y.shape, generator=gen).to(dtype).float()
    y.backward(dy)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV)            # [B,To,H,W,Cout]
    dyp = F.pad(dy_cl, (0, 0, 1, 1, 1, 1, 4, 4))
    w_cl = w.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV)
    dx = K.conv3d_k533_dgrad(dyp, w_cl)                                          # [B,T,H,W,Cin]
    got = dx.float().cpu().permute(0, 4, 1, 2, 3)
    rt = 2e-4 if dtype == torch.float32 else 3e-2
    assert torch.allclose(got, x.grad, rtol=rt, atol=rt * x.grad.abs().max().item()), (got - x.grad).abs().max()


def test_gemm_split_k_accumulates_into_running_sum(K):
    """Weight-gradient form with few output tiles: split along K, fp32 atomics into C."""
    gen = torch.Generator().manual_seed(8)
    Mred, N1, N2 = 12576, 768, 256
    dy = torch.randn(Mred, N1, generator=gen).to(torch.bfloat16)
    x = torch.randn(Mred, N2, generator=gen).to(torch.bfloat16)
    c0 = torch.randn(N1, N2, generator=gen)
    ref = c0 + dy.float().t() @ x.float()
    out = c0.clone().to(DEV)
    K.gemm(dy.to(DEV), x.to(DEV), out, None, False, False, accumulate=True)
    assert torch.allclose(out.cpu(), ref, rtol=2e-3, atol=0.5), (out.cpu() - ref).abs().max()


@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_large_tile_configuration(K, akm, bkm):
    """Shapes with >= 128 tiles of 256 x 256 take the 4x4-tile / 8-wave kernel (bf16 only)."""
    _gemm_case(K, torch.bfloat16, akm, bkm, 4104, 3080, 832, False, torch.bfloat16, with_bias=True)
    _gemm_case(K, torch.bfloat16, akm, bkm, 4096, 3072, 776, False, torch.float32, with_bias=False)
    _gemm_case(K, torch.bfloat16, akm, bkm, 3592, 3592, 520, True, torch.float32, with_bias=False)


def test_gemm_large_tile_integer_layout(K):
    M, N, Kd = 3584, 3584, 128
    a = (torch.arange(M * Kd).view(M, Kd) % 7 - 3).float()
    b = (torch.arange(N * Kd).view(N, Kd) % 5 - 2).float()
    b[3, 5] = 9
    ref = a @ b.t()
    for akm in (True, False):
        for bkm in (True, False):
            aa = (a if akm else a.t().contiguous()).to(torch.bfloat16).to(DEV)
            bb = (b if bkm else b.t().contiguous()).to(torch.bfloat16).to(DEV)
            out = torch.empty(M, N, device=DEV)
            K.gemm(aa, bb, out, None, akm, bkm)
            assert torch.equal(out.cpu(), ref), (akm, bkm)


# ------------------------------------------------------------------------------------------ fused epilogues
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_act_epilogue_and_pre_activation(K, dtype, act):
    gen = torch.Generator().manual_seed(5 + act)
    for (M, N, Kd) in [(1280, 3072, 768), (136, 1536, 768), (72, 64, 200)]:
        a = torch.randn(M, Kd, generator=gen).to(dtype)
        b = (torch.randn(N, Kd, generator=gen) / math.sqrt(Kd)).to(dtype)
        bias = torch.randn(N, generator=gen)
        pre_ref = a.float() @ b.float().t() + bias
        ref = pre_ref if act == 0 else (F.gelu(pre_ref) if act == 1 else F.relu(pre_ref))
        out = torch.empty(M, N, dtype=dtype, device=DEV)
        pre = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm_act(a.to(DEV), b.to(DEV), out, bias.to(DEV), act, pre)
        _assert_close(out, ref, dtype)
        _assert_close(pre, pre_ref, dtype)
        out2 = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm_act(a.to(DEV), b.to(DEV), out2, bias.to(DEV), act, None)
        assert torch.equal(out2, out)


def test_gemm_accumulates_into_bf16_output(K):
    """dx = dres + dy W: the sum is formed in fp32 and rounded once."""
    gen = torch.Generator().manual_seed(11)
    for (M, N, Kd, bkm) in [(1280, 768, 3072, False), (4096, 768, 768, False), (520, 72, 136, True)]:
        a = torch.randn(M, Kd, generator=gen).bfloat16()
        b = (torch.randn((N, Kd) if bkm else (Kd, N), generator=gen) / math.sqrt(Kd)).bfloat16()
        c0 = torch.randn(M, N, generator=gen).bfloat16()
        ref = c0.float() + a.float() @ (b.float().t() if bkm else b.float())
        c = c0.to(DEV)
        K.gemm(a.to(DEV), b.to(DEV), c, None, True, bkm, accumulate=True)
        assert torch.allclose(c.float().cpu(), ref, rtol=1e-2, atol=1e-2), (c.float().cpu() - ref).abs().max()


def test_colsum_single_launch_and_two_stage_paths(K):
    """K.colsum: bias gradient of an nn.Linear.  Accumulating sums of 16-byte-loadable matrices take the single-launch atomic
    kernel (shg_colsum_accumulate), everything else the two-stage partial / finish kernels; both against float64."""
    gen = torch.Generator().manual_seed(5)
    cases = [(12576, 768, torch.bfloat16, None), (4096, 2304, torch.bfloat16, 768), (1280, 3072, torch.bfloat16, None),
             (37, 1536, torch.bfloat16, None), (300, 456, torch.float32, None), (130, 171, torch.float32, None),
             (64, 4096, torch.bfloat16, None), (200, 2304, torch.float32, None), (96, 2048, torch.float32, None)]
    for rows, cols, dt, sl in cases:
        x = torch.randn(rows, cols, generator=gen).to(dt).to(DEV)
        view = x[:, sl:2 * sl] if sl else x                     # a column slice: rows `cols` elements apart
        ref = view.double().sum(0).cpu()
        for accumulate in (True, False):
            out0 = torch.randn(view.shape[1], generator=gen).to(DEV)
            out = out0.clone()
            K.colsum(view, out, accumulate)
            want = ref + (out0.double().cpu() if accumulate else 0.0)
            assert torch.allclose(out.double().cpu(), want, rtol=1e-5, atol=2e-3 * math.sqrt(rows)), (rows, cols, dt, accumulate)


def test_colsum_finish_multi(K):
    gen = torch.Generator().manual_seed(3)
    for (npart, cols, n) in [(1572, 768, 3), (16, 1536, 2), (300, 72, 4), (1, 768, 1)]:
        parts = [torch.randn(npart, cols, generator=gen).to(DEV) for _ in range(n)]
        outs = [torch.randn(cols, generator=gen).to(DEV) for _ in range(n)]
        refs = [o.double().cpu() + p.double().cpu().sum(0) for o, p in zip(outs, parts)]
        K.colsum_finish_multi(parts, outs)
        for o, r in zip(outs, refs):
            assert torch.allclose(o.double().cpu(), r, rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------------------------------ 8-phase 256 x 256 kernel
def test_gemm_8phase_kernel_shapes_and_epilogues(K):
    """bf16, both operands contraction-contiguous, K a multiple of 64 and >= 120 tiles of 256 x 256 take the
    8-phase kernel (gemm.hip: gemm8_kernel): odd / even numbers of K-tiles, ragged M and N, every epilogue."""
    for (M, N, Kd) in [(4096, 3072, 128), (4104, 3080, 192), (3592, 3592, 832), (12576, 2304, 768), (6200, 4040, 64 * 7)]:
        _gemm_case(K, torch.bfloat16, True, True, M, N, Kd, False, torch.bfloat16, with_bias=True)
        _gemm_case(K, torch.bfloat16, True, True, M, N, Kd, False, torch.float32, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, True, 4096, 4096, 512, True, torch.float32, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, True, 4096, 4096, 512, True, torch.bfloat16, with_bias=False)
    # input-gradient form (B contraction-strided, read with the transposing LDS read)
    for (M, N, Kd) in [(4096, 3072, 128), (12576, 768, 3072), (6200, 4040, 64 * 7)]:
        _gemm_case(K, torch.bfloat16, True, False, M, N, Kd, False, torch.bfloat16, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, False, 12576, 768, 2304, True, torch.bfloat16, with_bias=False)
    gen = torch.Generator().manual_seed(21)
    M, N, Kd = 12576, 3072, 768
    a = torch.randn(M, Kd, generator=gen).bfloat16()
    b = (torch.randn(N, Kd, generator=gen) / math.sqrt(Kd)).bfloat16()
    bias = torch.randn(N, generator=gen)
    pre_ref = a.float() @ b.float().t() + bias
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    K.gemm_act(a.to(DEV), b.to(DEV), out, bias.to(DEV), 1, pre)
    _assert_close(out, F.gelu(pre_ref), torch.bfloat16)
    _assert_close(pre, pre_ref, torch.bfloat16)


def test_gemm_8phase_kernel_is_exact_and_race_free_on_integer_data(K):
    """Small-integer operands are exact in bf16 / fp32 accumulation: any stale or early LDS read (the kernel keeps
    three half-tiles of direct-to-LDS prefetch in flight across its barriers) shows up as a wrong entry.
    Repeated, with a long contraction, under memory load from a concurrent copy."""
    M, N, Kd = 4096, 4096, 4096
    gen = torch.Generator().manual_seed(9)
    a = torch.randint(-3, 4, (M, Kd), generator=gen).float()
    b = torch.randint(-2, 3, (N, Kd), generator=gen).float()
    ref = (a.to(DEV) @ b.to(DEV).t()).cpu()              # fp32 on exact integers: exact
    aa, bb = a.bfloat16().to(DEV), b.bfloat16().to(DEV)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    bt = b.t().contiguous().bfloat16().to(DEV)           # [K, N]: the contraction-strided layout of the same B
    for it in range(6):
        out = torch.empty(M, N, device=DEV)
        with torch.cuda.stream(side):
            big.copy_(big.flip(0)) if it % 2 else big.zero_()
        if it < 3:
            K.gemm(aa, bb, out, None, True, True)
        else:
            K.gemm(aa, bt, out, None, True, False)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), it
    # asymmetric layout check on a small K
    M, N, Kd = 3584, 3584, 128
    a = (torch.arange(M * Kd).view(M, Kd) % 7 - 3).float()
    b = (torch.arange(N * Kd).view(N, Kd) % 5 - 2).float()
    b[3, 5] = 9
    out = torch.empty(M, N, device=DEV)
    K.gemm(a.bfloat16().to(DEV), b.bfloat16().to(DEV), out, None, True, True)
    assert torch.equal(out.cpu(), a @ b.t())


def test_gemm_8phase_192_row_tile(K):
    """The 192 x 256 tile of the 8-phase kernel (gemm.hip: TM = 192; chosen for the 12 576-row problems with 768 / 1 536 output
    columns): forced through the "gemm8_tile_m" switch on shapes with ragged last row tiles (rows % 192 = 96 / 48 / 8 / 0),
    both operand forms, every epilogue - against the same call on the 256-row tile (bit-identical: the contraction order of a
    tile does not depend on its height) and against fp32 references; plus exact small-integer data over 6 launches with an
    odd and an even number of K-tiles (stale or early LDS reads of the asymmetric half-tile staging would show)."""
    from shg_vqa_amd import _lib
    gen = torch.Generator().manual_seed(192)

    def both(fn):
        outs = []
        for tm in (256, 192):
            _lib.set_tuning("gemm8_tile_m", tm)
            try:
                outs.append(fn())
            finally:
                _lib.set_tuning("gemm8_tile_m", 0)
        return outs

    for (M, N, Kd, bkm) in [(12576, 768, 768, True), (12576, 1536, 768, True), (12576, 768, 3072, False), (12576, 768, 2304, False),
                            (4104, 3080, 192, True), (3848, 2048, 448, False), (6144, 2304, 128, True)]:
        a = torch.randn(M, Kd, generator=gen).bfloat16().to(DEV)
        b = (torch.randn((N, Kd) if bkm else (Kd, N), generator=gen) / math.sqrt(Kd)).bfloat16().to(DEV)
        bias = torch.randn(N, generator=gen).to(DEV)
        ref = a.float() @ (b.float().t() if bkm else b.float())
        # plain bf16 / fp32 outputs with bias
        for odt in (torch.bfloat16, torch.float32):
            o256, o192 = both(lambda: K.gemm(a, b, torch.empty(M, N, dtype=odt, device=DEV), bias, True, bkm))
            assert torch.equal(o256, o192), (M, N, Kd, bkm, odt)
            _assert_close(o192, ref + bias, torch.bfloat16, scale=2.0)
        # C += A.B on a bf16 residual gradient
        c0 = torch.randn(M, N, generator=gen).bfloat16().to(DEV)
        o256, o192 = both(lambda: K.gemm(a, b, c0.clone(), None, True, bkm, accumulate=True))
        assert torch.equal(o256, o192)
        _assert_close(o192, ref + c0.float(), torch.bfloat16, scale=2.0)
        if bkm:
            # forward epilogue: bias + GELU with the pre-activation saved
            def fwd():
                out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                K.gemm_act(a, b, out, bias, 1, pre)
                return torch.stack([out, pre])
            o256, o192 = both(fwd)
            assert torch.equal(o256, o192)
            _assert_close(o192[0], F.gelu(ref + bias), torch.bfloat16, scale=2.0)
        else:
            # input-gradient epilogue: activation backward + bias-gradient column sums
            pre = torch.randn(M, N, generator=gen).bfloat16().to(DEV)

            def bwd():
                dx = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                db = torch.zeros(N, device=DEV)
                K.gemm_dact(a, b, dx, pre, db, 1)
                return dx, db
            (d256, b256), (d192, b192) = both(bwd)
            assert torch.equal(d256, d192)
            p32 = pre.float().requires_grad_(True)
            F.gelu(p32).backward(ref)
            _assert_close(d192, p32.grad, torch.bfloat16, scale=2.0)
            assert torch.allclose(b192.double().cpu(), d192.double().cpu().sum(0), rtol=1e-4, atol=2e-3 * math.sqrt(M))
    # exact integers, odd and even K-tile counts, repeated
    for Kd in (64 * 7, 4096):
        M, N = 12576, 768
        a = torch.randint(-3, 4, (M, Kd), generator=gen).float()
        b = torch.randint(-2, 3, (N, Kd), generator=gen).float()
        ref = (a.to(DEV) @ b.to(DEV).t()).cpu()
        aa, bb, bt = a.bfloat16().to(DEV), b.bfloat16().to(DEV), b.t().contiguous().bfloat16().to(DEV)
        _lib.set_tuning("gemm8_tile_m", 192)
        try:
            for it in range(6):
                out = torch.empty(M, N, device=DEV)
                K.gemm(aa, bb if it % 2 == 0 else bt, out, None, True, it % 2 == 0)
                assert torch.equal(out.cpu(), ref), (Kd, it)
        finally:
            _lib.set_tuning("gemm8_tile_m", 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [1, 2])
def test_gemm_dact_fuses_activation_backward_and_bias_gradient(K, dtype, act):
    gen = torch.Generator().manual_seed(31 + act)
    for (M, N, Kd) in [(1280, 3072, 768), (12576, 3072, 768), (200, 264, 72)]:
        if dtype == torch.float32 and M > 5000:
            continue
        dy = torch.randn(M, Kd, generator=gen).to(dtype)
        w = (torch.randn(Kd, N, generator=gen) / math.sqrt(Kd)).to(dtype)
        pre = torch.randn(M, N, generator=gen).to(dtype)
        p32 = pre.float().requires_grad_(True)
        (F.gelu(p32) if act == 1 else F.relu(p32)).backward(dy.float() @ w.float())
        ref = p32.grad
        dx = torch.empty(M, N, dtype=dtype, device=DEV)
        db0 = torch.randn(N, generator=gen)
        db = db0.clone().to(DEV)
        K.gemm_dact(dy.to(DEV), w.to(DEV), dx, pre.to(DEV), db, act)
        _assert_close(dx, ref, dtype)
        exp_db = db0.double() + dx.double().cpu().sum(0)           # the sums are taken over the stored values
        assert torch.allclose(db.double().cpu(), exp_db, rtol=1e-4, atol=2e-3 * math.sqrt(M)), (db.double().cpu() - exp_db).abs().max()
        dx2 = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm_dact(dy.to(DEV), w.to(DEV), dx2, pre.to(DEV), None, act)
        assert torch.equal(dx2, dx)


def test_gemm_act_can_save_the_activation_derivative_for_gemm_dact(K):
    """SHG_ACT_SAVE_GRAD / SHG_ACT_SAVED_GRAD (include/shg_vqa.h): the forward epilogue stores gelu'(u) in `pre`, the input-gradient
    epilogue multiplies by it - against the derivative recomputed from the stored pre-activation (the default pair) and against
    torch's erf GELU in fp64; same outputs `h` either way."""
    gen = torch.Generator().manual_seed(77)
    for (M, N, Kd) in [(12576, 3072, 768), (1280, 3072, 768), (200, 264, 72)]:
        x = torch.randn(M, Kd, generator=gen).bfloat16().to(DEV)
        w1 = (torch.randn(N, Kd, generator=gen) / math.sqrt(Kd)).bfloat16().to(DEV)
        b1 = torch.randn(N, generator=gen).to(DEV)
        dy = torch.randn(M, 96, generator=gen).bfloat16().to(DEV)
        w2 = (torch.randn(96, N, generator=gen) / 10).bfloat16().to(DEV)
        h0, pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        h1, der = torch.empty_like(h0), torch.empty_like(h0)
        K.gemm_act(x, w1, h0, b1, 1, pre)
        K.gemm_act(x, w1, h1, b1, 1 | 0x100, der)
        assert torch.equal(h0, h1)
        u = (x.double() @ w1.double().t() + b1.double())
        u.requires_grad_(True)
        F.gelu(u).sum().backward()
        assert torch.allclose(der.double(), u.grad, rtol=2 ** -7, atol=2e-3), (der.double() - u.grad).abs().max().item()
        dx0, dx1 = torch.empty_like(h0), torch.empty_like(h0)
        db0, db1 = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
        K.gemm_dact(dy, w2, dx0, pre, db0, 1)
        K.gemm_dact(dy, w2, dx1, der, db1, 3)
        ref = (dy.double() @ w2.double()) * u.grad
        _assert_close(dx0, ref, torch.bfloat16, scale=2.0)
        _assert_close(dx1, ref, torch.bfloat16, scale=2.0)
        assert torch.allclose(db1.double().cpu(), dx1.double().cpu().sum(0), rtol=1e-4, atol=2e-3 * math.sqrt(M))


def test_fast_gelu_of_the_bf16_path_over_its_whole_range(K):
    """bf16 kernels evaluate erf by a rational approximation (|err| <= 1.5e-7): forward and gradient against
    torch's erf GELU on a dense grid including both tails."""
    x = torch.linspace(-12.0, 12.0, 64 * 1024).view(64, 1024)
    xb = x.bfloat16()
    y = K.bias_act_fwd(xb.to(DEV), None, 1)
    ref = F.gelu(xb.double())                                   # fp32 erf loses the negative tail: 1 + erf(x) rounds to 0 / 2^-24
    assert torch.allclose(y.double().cpu(), ref, rtol=2 ** -7, atol=2e-6), (y.double().cpu() - ref).abs().max()
    ones = torch.ones_like(xb)
    dx, _ = K.bias_act_bwd(xb.to(DEV), None, ones.to(DEV), 1, want_dbias=False)
    x64 = xb.double().requires_grad_(True)
    F.gelu(x64).sum().backward()
    assert torch.allclose(dx.double().cpu(), x64.grad, rtol=2 ** -7, atol=2e-6), (dx.double().cpu() - x64.grad).abs().max()


# ------------------------------------------------------------------------------------------ per-clip matcher
@pytest.mark.parametrize("tag", ["rel", "act", "ties"])
def test_hungarian_per_clip_bit_exact_vs_reference_matcher_golden(K, golden_dir, tag):
    """matcher.py:82-104 (loss_hg_per_frame=False): one 128 x n / 48 x n problem per sample; golden indices come from
    the REAL reference matcher (oracle/gen_golden.py matcher_clip), including n = 0, n = Q, duplicate labels and
    logits with few distinct values (exact ties in the cost matrix)."""
    g = np.load(os.path.join(golden_dir, "matcher_clip.npz"))
    logits = torch.from_numpy(g[tag + "_logits"]).to(DEV)
    tgt = torch.from_numpy(g[tag + "_tgt"]).clamp(min=0).to(DEV)
    lens = torch.from_numpy(g[tag + "_len"]).to(DEV)
    oq, ot, grid = K.hungarian_per_frame(logits, tgt, lens)
    assert np.array_equal(oq.cpu().numpy(), g[tag + "_q"]), tag
    assert np.array_equal(ot.cpu().numpy(), g[tag + "_t"]), tag
    # target grid: background except matched query <- its label
    exp = torch.zeros_like(grid.cpu())
    for b in range(oq.shape[0]):
        n = int(g[tag + "_len"][b])
        exp[b, torch.from_numpy(g[tag + "_q"][b, :n])] = torch.from_numpy(g[tag + "_tgt"][b])[torch.from_numpy(g[tag + "_t"][b, :n])]
    assert torch.equal(grid.cpu(), exp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hungarian_per_clip_matches_oracle_on_model_sized_problems(K, dtype):
    from oracle import shg_ref
    gen = torch.Generator().manual_seed(12)
    for (B, Q, C) in [(8, 128, 457), (8, 48, 158), (3, 24, 30)]:
        logits = (torch.randn(B, Q, C, generator=gen) * 2).to(dtype)
        lens = torch.randint(0, Q + 1, (B,), generator=gen)
        labels = [torch.randint(1, C, (int(n),), generator=gen) for n in lens]
        ref = shg_ref.hungarian_per_frame(logits.float(), labels, clip_len=1)
        tgt = torch.zeros(B, Q, dtype=torch.int64)
        for b, l in enumerate(labels):
            tgt[b, :len(l)] = l
        oq, ot, _ = K.hungarian_per_frame(logits.to(DEV), tgt.to(DEV), lens.to(torch.int32).to(DEV))
        for b, (qi, ti) in enumerate(ref):
            n = int(lens[b])
            assert torch.equal(oq[b, :n].cpu(), qi) and torch.equal(ot[b, :n].cpu(), ti), (dtype, B, Q, b)
            assert (oq[b, n:] == -1).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_dropout_uses_the_masks_of_the_standalone_kernels(K, dtype):
    """Linear + ReLU + dropout as one kernel (transformer.py:230) and its backward in the dgrad epilogue draw the same
    counter-based masks as shg_bias_act_fwd / bwd: fused and unfused paths must agree element for element."""
    gen = torch.Generator().manual_seed(41)
    seed = torch.tensor([77, 3], dtype=torch.int64, device=DEV)
    for (M, N, Kd) in [(4096, 2048, 768), (200, 264, 72)]:
        x = torch.randn(M, Kd, generator=gen).to(dtype).to(DEV)
        w = (torch.randn(N, Kd, generator=gen) / math.sqrt(Kd)).to(dtype).to(DEV)
        bias = torch.randn(N, generator=gen).to(DEV)
        p, sid = 0.15, 9
        # forward: fused vs GEMM (+ bias, pre-activation out) followed by the stand-alone activation + dropout kernel
        out = torch.empty(M, N, dtype=dtype, device=DEV)
        pre = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm_act(x, w, out, bias, 2, pre, p, seed, sid)
        ref = K.bias_act_fwd(pre, None, 2, p, seed, sid)
        if dtype == torch.float32:
            assert torch.equal(out, ref)
        else:                                  # the fused epilogue rounds once (fp32 accumulator -> bf16), the two-pass path twice
            _assert_close(out, ref, dtype)
            assert torch.equal(out == 0, ref == 0)
        kept = (out != 0).float().mean().item() / max((pre > 0).float().mean().item(), 1e-6)
        assert abs(kept - (1 - p)) < 0.02, kept
        # backward: fused dgrad epilogue vs GEMM followed by the stand-alone backward kernel
        dy = torch.randn(M, 96, generator=gen).to(dtype).to(DEV)
        w2 = (torch.randn(96, N, generator=gen) / 10).to(dtype).to(DEV)
        dx = torch.empty(M, N, dtype=dtype, device=DEV)
        db = torch.zeros(N, device=DEV)
        K.gemm_dact(dy, w2, dx, pre, db, 2, p, seed, sid)
        dh = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm(dy, w2, dh, None, True, False)
        dref, _ = K.bias_act_bwd(pre, None, dh, 2, p, seed, sid, want_dbias=False)
        _assert_close(dx, dref, dtype)
        assert torch.equal(dx == 0, dref == 0)            # identical masks (and ReLU gates)
        assert torch.allclose(db.double().cpu(), dx.double().cpu().sum(0), rtol=1e-4, atol=2e-3 * math.sqrt(M))


def test_gemm_small_problem_kernel(K):
    """bf16 NT problems of at most 256 tiles of 128 x 128 with K % 64 == 0 take the 8-wave small-problem kernel
    (gemm.hip: gemm4_kernel): every K-tile count from 2 up (prologue / tail paths), ragged M and N, every epilogue,
    and an integer-exact race screen (a K-tile of direct-to-LDS prefetch stays in flight across each barrier)."""
    for nk in (2, 3, 4, 5, 12, 33):
        _gemm_case(K, torch.bfloat16, True, True, 1000, 776, 64 * nk, False, torch.bfloat16, with_bias=True)
    for (M, N, Kd) in [(4096, 768, 768), (1536, 2048, 768), (1280, 3072, 768), (4096, 1536, 768), (72, 64, 128)]:
        _gemm_case(K, torch.bfloat16, True, True, M, N, Kd, False, torch.bfloat16, with_bias=True)
        _gemm_case(K, torch.bfloat16, True, True, M, N, Kd, False, torch.float32, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, True, 1280, 768, 512, True, torch.float32, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, True, 1280, 768, 512, True, torch.bfloat16, with_bias=False)
    # input-gradient form: B contraction-strided (transposing LDS reads)
    for (M, N, Kd) in [(4096, 768, 768), (1536, 768, 2048), (1000, 776, 192), (1280, 768, 3072)]:
        _gemm_case(K, torch.bfloat16, True, False, M, N, Kd, False, torch.bfloat16, with_bias=False)
    _gemm_case(K, torch.bfloat16, True, False, 1280, 768, 2304, True, torch.bfloat16, with_bias=False)
    M, N, Kd = 2048, 1536, 4096
    gen = torch.Generator().manual_seed(19)
    a = torch.randint(-3, 4, (M, Kd), generator=gen).float()
    b = torch.randint(-2, 3, (N, Kd), generator=gen).float()
    ref = (a.to(DEV) @ b.to(DEV).t()).cpu()
    aa, bb = a.bfloat16().to(DEV), b.bfloat16().to(DEV)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    for it in range(5):
        out = torch.empty(M, N, device=DEV)
        with torch.cuda.stream(side):
            big.copy_(big.flip(0)) if it % 2 else big.zero_()
        if it < 3:
            K.gemm(aa, bb, out, None, True, True)
        else:
            K.gemm(aa, bb.t().contiguous(), out, None, True, False)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), it
    # layout check with asymmetric integers (a swapped fragment or stage shows up as a wrong entry)
    M, N, Kd = 384, 256, 192
    a = (torch.arange(M * Kd).view(M, Kd) % 7 - 3).float()
    b = (torch.arange(N * Kd).view(N, Kd) % 5 - 2).float()
    b[3, 5] = 9
    out = torch.empty(M, N, device=DEV)
    K.gemm(a.bfloat16().to(DEV), b.bfloat16().to(DEV), out, None, True, True)
    assert torch.equal(out.cpu(), a @ b.t())
    K.gemm(a.bfloat16().to(DEV), b.t().contiguous().bfloat16().to(DEV), out, None, True, False)
    assert torch.equal(out.cpu(), a @ b.t())


# ------------------------------------------------------------------------------------------ per-frame matcher, 10k frames
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", ["rel", "act"])
def test_hungarian_per_frame_bit_exact_vs_reference_matcher_golden_10k_frames(K, golden_dir, tag, dtype):
    """matcher.py:66-80 (--LossHGPerFrame): 10 240 frames of raw logits per head; the golden indices come from the REAL
    reference matcher (oracle/gen_golden.py matcher_frames).  The logits are multiples of 1/64 below 4 in magnitude, i.e. exact
    in bf16 as well: both storage types must reproduce the reference's indices bit for bit."""
    g = np.load(os.path.join(golden_dir, "matcher_frames.npz"))
    k = g[tag + "_logits_x64"]
    n, per, c = k.shape
    assert n >= 10000
    logits = (torch.from_numpy(k.astype(np.float32)) / 64.0).to(dtype)
    assert torch.equal(logits.float() * 64.0, torch.from_numpy(k.astype(np.float32)))
    tgt = torch.from_numpy(g[tag + "_tgt"].astype(np.int64)).to(DEV)
    lens = torch.from_numpy(g[tag + "_len"].astype(np.int32)).to(DEV)
    oq, ot, grid = K.hungarian_per_frame(logits.to(DEV), tgt, lens)
    gq, gt = g[tag + "_q"].astype(np.int64), g[tag + "_t"].astype(np.int64)
    bad = np.nonzero((oq.cpu().numpy() != gq).any(1) | (ot.cpu().numpy() != gt).any(1))[0]
    assert bad.size == 0, (tag, dtype, bad[:10], oq[bad[:1]].cpu(), gq[bad[:1]])
    exp = torch.zeros(n, per, dtype=torch.int64)
    tg = torch.from_numpy(g[tag + "_tgt"].astype(np.int64))
    for f in range(n):
        m = int(g[tag + "_len"][f])
        exp[f, torch.from_numpy(gq[f, :m])] = tg[f][torch.from_numpy(gt[f, :m])]
    assert torch.equal(grid.cpu(), exp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", ["rel", "act"])
def test_hungarian_per_frame_bit_exact_vs_reference_matcher_golden_at_model_class_widths(K, golden_dir, tag, dtype):
    """matcher.py:62-80 at the widths the model calls it with (8 x 457 and 3 x 158 per frame: the register-resident <= 512-class
    path of the kernel): 2 560 frames per head, indices from the REAL reference matcher (oracle/gen_golden.py
    matcher_frames_wide; the logits - multiples of 1/64, exact in bf16 - are regenerated from the stored seed and CRC-checked)."""
    from test_oracle_golden import _wide_frames
    g = np.load(os.path.join(golden_dir, "matcher_frames_wide.npz"))
    k, B, T, per, C = _wide_frames(g, tag)
    n = B * T
    logits = (torch.from_numpy(k.astype(np.float32)) / 64.0).to(dtype).view(n, per, C)
    tgt = torch.from_numpy(g[tag + "_tgt"].astype(np.int64)).to(DEV)
    lens = torch.from_numpy(g[tag + "_len"].astype(np.int32)).to(DEV)
    oq, ot, grid = K.hungarian_per_frame(logits.to(DEV).contiguous(), tgt, lens)
    gq, gt = g[tag + "_q"].astype(np.int64), g[tag + "_t"].astype(np.int64)
    bad = np.nonzero((oq.cpu().numpy() != gq).any(1) | (ot.cpu().numpy() != gt).any(1))[0]
    assert bad.size == 0, (tag, dtype, bad[:10], oq[bad[:1]].cpu(), gq[bad[:1]])


# ------------------------------------------------------------------------------------------ convolutions at the benchmark's shape
def _im2col(xp, T_out, H, W):
    """Rows of the implicit GEMM: windows (5,3,3) of a zero-bordered channels-last tensor [B, T, H+2, W+2, C] -> [B*T_out*H*W, 45*C]."""
    B, _, _, _, C = xp.shape
    sB, sT, sH, sW, sC = xp.stride()
    return xp.as_strided((B, T_out, H, W, 5, 3, 3, C), (sB, sT, sH, sW, sT, sH, sW, sC)).reshape(B * T_out * H * W, 45 * C)


@pytest.mark.parametrize("name,Cin,T", [("conv1", 2048, 16), ("conv2", 768, 12)])
def test_conv_forward_at_bench_shape_streamk_integer_exact_and_bf16_rows(K, name, Cin, T):
    """The two convolutions exactly as bench.py runs them (B = 32: conv1 2048 -> 768 = 222 tiles x 1440 K-tiles, conv2 768 -> 768 =
    147 tiles x 540 K-tiles, both on the stream-K split): (i) small-integer data, every output element equal to an fp32 im2col
    GEMM; (ii) random bf16 data, 4 096 sampled output rows against the fp32 im2col GEMM of the same bf16 operands."""
    from shg_vqa_amd import _lib
    B, H, W, Cout = 32, 7, 7, 768
    To = T - 4
    gen = torch.Generator().manual_seed(11 + Cin)
    before = int(_lib.lib().shg_gemm_streamk_launches())
    # (i) integers
    xi = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV)
    xi[:, :, 1:-1, 1:-1] = torch.randint(-2, 3, (B, T, H, W, Cin), generator=gen).float().to(DEV)
    wi = torch.randint(-1, 2, (Cout, 5, 3, 3, Cin), generator=gen).float().to(DEV)
    bi = torch.randint(-3, 4, (Cout,), generator=gen).float().to(DEV)
    ref = (_im2col(xi, To, H, W) @ wi.reshape(Cout, -1).t() + bi).bfloat16()
    y = K.conv3d_k533_fwd(xi.bfloat16(), wi.bfloat16(), bi, act=0)
    torch.cuda.synchronize()
    assert torch.equal(y.view(-1, Cout), ref), (y.view(-1, Cout).float() - ref.float()).abs().max()
    # position-major rows: tiles drop the taps that read only the zero border, the stream-K launch runs its WEIGHTED plan
    # ("conv_k_order" bit 5); dense output back in standard order through the row table - the same integers
    inv = K.conv_row_table_inv(B, T, H, W, DEV)
    for sw in (62, 30):
        _lib.set_tuning("conv_k_order", sw)
        try:
            y_pm = K.conv3d_k533_fwd(xi.bfloat16(), wi.bfloat16(), bi, act=0, order=1, y_rows=inv)
            torch.cuda.synchronize()
            assert torch.equal(y_pm.view(-1, Cout), ref), (sw, (y_pm.view(-1, Cout).float() - ref.float()).abs().max())
        finally:
            _lib.set_tuning("conv_k_order", 126)
    before += 2
    # ... the same without the stream-K launch (one tile per workgroup, each with its own K-tile list)
    _lib.set_tuning("streamk", 0)
    try:
        y_pm = K.conv3d_k533_fwd(xi.bfloat16(), wi.bfloat16(), bi, act=0, order=1, y_rows=inv)
        assert torch.equal(y_pm.view(-1, Cout), ref)
    finally:
        _lib.set_tuning("streamk", 1)
    # ... and at another batch size (24: 288 / 192 rows per position, other unions of taps per tile, another weighted plan)
    B2 = 24
    inv2 = K.conv_row_table_inv(B2, T, H, W, DEV)
    y_pm = K.conv3d_k533_fwd(xi[:B2].bfloat16(), wi.bfloat16(), bi, act=0, order=1, y_rows=inv2)
    assert torch.equal(y_pm.view(-1, Cout), ref[:B2 * To * H * W])
    before += 1 if B2 * To * H * W // 256 * 3 >= 128 else 0
    del ref, xi, wi
    # (ii) random data, sampled rows, GELU epilogue + pre-activation output as in the step
    xr = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV, dtype=torch.bfloat16)
    xr[:, :, 1:-1, 1:-1] = torch.randn(B, T, H, W, Cin, generator=gen).to(DEV).bfloat16()
    wr = (torch.randn(Cout, 5, 3, 3, Cin, generator=gen) / math.sqrt(45 * Cin)).to(DEV).bfloat16()
    br = (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
    yg, pre = K.conv3d_k533_fwd(xr, wr, br, act=1, want_pre=True)
    rows = torch.randperm(B * To * H * W, generator=gen)[:4096].to(DEV)
    ref_pre = _im2col(xr.float(), To, H, W)[rows] @ wr.float().reshape(Cout, -1).t() + br
    err_pre = (pre.view(-1, Cout)[rows].float() - ref_pre).abs().max().item()
    err_act = (yg.view(-1, Cout)[rows].float() - F.gelu(ref_pre)).abs().max().item()
    scale = ref_pre.abs().max().item()
    print("%s fwd B=32: max |pre - ref| = %.3e, max |gelu - ref| = %.3e (|ref| max %.3f)" % (name, err_pre, err_act, scale))
    assert err_pre <= 2 ** -8 * scale + 1e-3 and err_act <= 2 ** -8 * scale + 1e-3      # one bf16 rounding of an fp32 sum
    assert int(_lib.lib().shg_gemm_streamk_launches()) == before + 2, "the stream-K path was not taken"
    # the step's form: standard-order forward whose pre-activation rows go through the position-major table (stream-K epilogue)
    tbl = K.conv_row_table(B, T, H, W, DEV)
    yg2, pre2 = K.conv3d_k533_fwd(xr, wr, br, act=1, want_pre=True, pre_rows=tbl)
    assert torch.equal(yg2, yg) and torch.equal(pre2.view(-1, Cout)[tbl.long()], pre.view(-1, Cout))


@pytest.mark.parametrize("name,Cin,T", [("conv1", 2048, 16), ("conv2", 768, 12)])
def test_conv_wgrad_at_bench_shape_integer_exact_and_bf16(K, name, Cin, T):
    """Weight gradients of both convolutions at B = 32 (conv1: 3 x 360 tiles of 256 x 256, K = 18 816 rows): integer data equal
    to dY^T . im2col(X) in fp32; random bf16 data against the same fp32 product, 64 sampled output channels."""
    from shg_vqa_amd import _lib
    B, H, W, Cout = 32, 7, 7, 768
    To = T - 4
    gen = torch.Generator().manual_seed(23 + Cin)
    xi = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV)
    xi[:, :, 1:-1, 1:-1] = torch.randint(-2, 3, (B, T, H, W, Cin), generator=gen).float().to(DEV)
    dyi = torch.randint(-2, 3, (B, To, H, W, Cout), generator=gen).float().to(DEV)
    ref = dyi.view(-1, Cout).t() @ _im2col(xi, To, H, W)
    dw = torch.zeros(Cout, 5, 3, 3, Cin, device=DEV)
    K.conv3d_k533_wgrad(xi.bfloat16(), dyi.bfloat16(), dw)
    assert torch.equal(dw.view(Cout, -1), ref), (dw.view(Cout, -1) - ref).abs().max()
    K.conv3d_k533_wgrad(xi.bfloat16(), dyi.bfloat16(), dw, accumulate=True)
    assert torch.equal(dw.view(Cout, -1), 2 * ref)
    # output-channel slices (the data-parallel step issues conv1's gradient as 2/3 + 1/3, ops._VisualConvTokens.backward)
    dw.zero_()
    K.conv3d_k533_wgrad(xi.bfloat16(), dyi.bfloat16(), dw, c0=512, cn=256)
    assert torch.equal(dw.view(Cout, -1)[512:], ref[512:]) and not dw[:512].any()
    K.conv3d_k533_wgrad(xi.bfloat16(), dyi.bfloat16(), dw, c0=0, cn=512)
    assert torch.equal(dw.view(Cout, -1), ref)
    # overwrite form (the step's single-writer path: no zeroed destination, the sum of squares for clip_grad_norm_ on the way; the
    # launch over the whole rounds zeroes the remainder's column blocks, which the split launch then adds into)
    ss = torch.zeros(1, dtype=torch.float64, device=DEV)
    want = (ref.double() ** 2).sum().item()
    dw.fill_(7.0)
    K.conv3d_k533_wgrad_sumsq(xi.bfloat16(), dyi.bfloat16(), dw, ss)
    assert torch.equal(dw.view(Cout, -1), ref), (dw.view(Cout, -1) - ref).abs().max()
    assert abs(ss.item() - want) <= 1e-6 * want, (ss.item(), want)
    dw.fill_(-3.0)
    K.conv3d_k533_wgrad_sumsq(xi.bfloat16(), dyi.bfloat16(), dw, ss, c0=512, cn=256)
    K.conv3d_k533_wgrad_sumsq(xi.bfloat16(), dyi.bfloat16(), dw, ss, c0=0, cn=512)
    assert torch.equal(dw.view(Cout, -1), ref)
    assert abs(ss.item() - 2 * want) <= 2e-6 * want, (ss.item(), 2 * want)
    # position-major rows (shg_conv3d_k533_wgrad_ex, row_order 1): B To = 384 / 256 rows = whole K-tiles per spatial position, every
    # tile skips the positions where its tap reads the zero border - same integers
    tbl = K.conv_row_table(B, T, H, W, DEV).long()
    dy_pm = torch.empty(B * To * H * W, Cout, device=DEV)
    dy_pm[tbl] = dyi.view(-1, Cout)
    dy_pm = dy_pm.view(B, To, H, W, Cout).bfloat16()
    for skip in (30, 14, 6):                                # with and without the skipping (bit 3 of conv_k_order)
        _lib.set_tuning("conv_k_order", skip)
        try:
            dw.fill_(5.0)
            K.conv3d_k533_wgrad(xi.bfloat16(), dy_pm, dw, order=1)
            assert torch.equal(dw.view(Cout, -1), ref), (skip, (dw.view(Cout, -1) - ref).abs().max())
            K.conv3d_k533_wgrad(xi.bfloat16(), dy_pm, dw, accumulate=True, order=1)
            assert torch.equal(dw.view(Cout, -1), 2 * ref), skip
            ss.zero_()
            dw.fill_(-1.0)
            K.conv3d_k533_wgrad_sumsq(xi.bfloat16(), dy_pm, dw, ss, order=1)
            assert torch.equal(dw.view(Cout, -1), ref), skip
            assert abs(ss.item() - want) <= 1e-6 * want, (skip, ss.item(), want)
            # the data-parallel step's form: accumulating output-channel slices (2/3 + 1/3) in the same row order
            dw.zero_()
            K.conv3d_k533_wgrad(xi.bfloat16(), dy_pm, dw, accumulate=True, c0=0, cn=512, order=1)
            K.conv3d_k533_wgrad(xi.bfloat16(), dy_pm, dw, accumulate=True, c0=512, cn=256, order=1)
            assert torch.equal(dw.view(Cout, -1), ref), skip
        finally:
            _lib.set_tuning("conv_k_order", 126)
    del ref, xi, dyi, dy_pm
    xr = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV, dtype=torch.bfloat16)
    xr[:, :, 1:-1, 1:-1] = torch.randn(B, T, H, W, Cin, generator=gen).to(DEV).bfloat16()
    dyr = torch.randn(B, To, H, W, Cout, generator=gen).to(DEV).bfloat16()
    K.conv3d_k533_wgrad(xr, dyr, dw)
    ch = torch.randperm(Cout, generator=gen)[:64].to(DEV)
    ref = dyr.view(-1, Cout)[:, ch].float().t() @ _im2col(xr.float(), To, H, W)
    err = (dw.view(Cout, -1)[ch] - ref).abs().max().item()
    print("%s wgrad B=32: max |dW - ref| = %.3e (|ref| max %.1f)" % (name, err, ref.abs().max().item()))
    assert err <= 2e-4 * ref.abs().max().item() + 1e-3          # fp32 accumulation order only


@pytest.mark.parametrize("M,N,k,n_seg", [(12576, 768, 1536, 5), (300, 96, 128, 3)])
def test_gemm_over_k_segments(K, M, N, k, n_seg):
    """shg_gemm_kseg: sum over segments of A_s . B_s in one launch (the decoders' gradient w.r.t. their memory: 8-phase kernel at
    12 576 x 768, five segments of 1 536) and its segment-by-segment fallback, against fp32 matmuls of the same bf16 operands; padded
    strides between the segments as in the executor's scratch / the parameter arena."""
    gen = torch.Generator().manual_seed(M + k)
    a = torch.randn(n_seg, M + 3, k, generator=gen).to(DEV).bfloat16()[:, :M]                  # (segment stride (M + 3) k)
    b = (torch.randn(n_seg, k + 16, N, generator=gen) / math.sqrt(k * n_seg)).to(DEV).bfloat16()[:, :k]
    ref = sum(a[s].float() @ b[s].float() for s in range(n_seg))
    out = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
    K.gemm_kseg(a, b, out)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2 ** -7 * ref.abs().max().item() + 1e-3, err
    base = torch.randn(M, N, generator=gen).to(DEV).bfloat16()
    out2 = base.clone()
    K.gemm_kseg(a, b, out2, accumulate=True)
    err2 = (out2.float() - (ref + base.float())).abs().max().item()
    assert err2 <= 2 ** -6 * (ref.abs().max().item() + base.abs().max().item()) + 1e-3, err2


def test_bias_act_bwd_with_a_row_table(K):
    """shg_bias_act_bwd_rows: result row r reads x / writes dx at row x_rows[r]; dy (a grouped view), the scattered second output and
    the bias-gradient partial sums stay indexed by r."""
    B, S, C = 3, 40, 768
    gen = torch.Generator().manual_seed(9)
    rows = B * S
    perm = torch.randperm(rows, generator=gen).to(DEV)
    x = torch.randn(rows, C, generator=gen).to(DEV).bfloat16()
    dy = torch.randn(B, S + 1, C, generator=gen).to(DEV).bfloat16()          # one extra leading row per group, skipped
    tbl2 = torch.randperm(rows + 7, generator=gen)[:rows].to(DEV).int()
    buf0 = torch.zeros(rows + 7, C, device=DEV, dtype=torch.bfloat16)
    buf1 = torch.zeros_like(buf0)
    dx0, p0 = K.bias_act_bwd(x, None, dy, 1, want_dbias=True, dy_groups=(S, S + 1, 1), out2=(buf0, tbl2))
    x_pm = torch.empty_like(x)
    x_pm[perm] = x                                                           # row r of x lives at row perm[r]
    dx1, p1 = K.bias_act_bwd(x_pm, None, dy, 1, want_dbias=True, dy_groups=(S, S + 1, 1), out2=(buf1, tbl2), x_rows=perm.int())
    assert torch.equal(dx1[perm], dx0) and torch.equal(buf0, buf1) and torch.equal(p0, p1)


def test_conv_row_order_position_major_forward_and_input_gradient(K):
    """Row order 1 of the convolution GEMMs (include/shg_vqa.h): the forward's dense pre-activation comes out in position-major
    rows (same values, the padded output is a layout and does not change), shg_conv3d_k533_dgrad_rows scatters its rows by a table."""
    B, T, H, W, Cin, Cout = 4, 9, 7, 7, 256, 128
    To = T - 4
    gen = torch.Generator().manual_seed(77)
    x = torch.zeros(B, T, H + 2, W + 2, Cin, device=DEV, dtype=torch.bfloat16)
    x[:, :, 1:-1, 1:-1] = torch.randn(B, T, H, W, Cin, generator=gen).to(DEV).bfloat16()
    w = (torch.randn(Cout, 5, 3, 3, Cin, generator=gen) * 0.05).to(DEV).bfloat16()
    b = torch.randn(Cout, generator=gen).to(DEV)
    tbl = K.conv_row_table(B, T, H, W, DEV).long()
    m = torch.arange(B * To * H * W, device=DEV)
    w_, h_, t_, b_ = m % W, (m // W) % H, (m // (W * H)) % To, m // (W * H * To)
    assert torch.equal(tbl, ((h_ * W + w_) * B + b_) * To + t_)
    y0, pre0 = K.conv3d_k533_fwd(x, w, b, 1, pad_out=True, want_pre=True)
    y1, pre1 = K.conv3d_k533_fwd(x, w, b, 1, pad_out=True, want_pre=True, order=1)
    assert torch.equal(y0, y1)
    assert torch.equal(pre1.view(-1, Cout)[tbl], pre0.view(-1, Cout))
    # ... and a forward in standard order that writes only its pre-activation through the table (what the step does)
    y2, pre2 = K.conv3d_k533_fwd(x, w, b, 1, pad_out=True, want_pre=True, pre_rows=K.conv_row_table(B, T, H, W, DEV))
    assert torch.equal(y0, y2) and torch.equal(pre2, pre1)
    d0 = K.conv3d_k533_fwd(x, w, b, 1, pad_out=False)
    d1 = K.conv3d_k533_fwd(x, w, b, 1, pad_out=False, order=1)
    assert torch.equal(d1.view(-1, Cout)[tbl], d0.view(-1, Cout))
    # input gradient of a following conv whose output grid is this one: rows scattered into position-major order
    dyp = torch.zeros(B, To + 4, H + 2, W + 2, Cout, device=DEV, dtype=torch.bfloat16)
    dyp[:, 4:To, 1:-1, 1:-1] = torch.randn(B, To - 4, H, W, Cout, generator=gen).to(DEV).bfloat16()
    w2 = (torch.randn(Cout, 5, 3, 3, Cout, generator=gen) * 0.05).to(DEV).bfloat16()
    tbl2 = K.conv_row_table(B, To + 4, H, W, DEV)          # the grid of dx: [B, To, H, W]
    dx0 = K.conv3d_k533_dgrad(dyp, w2)
    dx1 = K.conv3d_k533_dgrad(dyp, w2, out_rows=tbl2)
    assert torch.equal(dx1.view(-1, Cout)[tbl2.long()], dx0.view(-1, Cout))


def test_conv2_dgrad_at_bench_shape_integer_exact_and_bf16(K):
    """Input gradient of the second convolution at B = 32 (the only conv input gradient of the step; 147 tiles): integer data equal
    to im2col(dY padded) x flipped weights in fp32; random bf16 data, 4 096 sampled rows."""
    B, To, H, W, C = 32, 8, 7, 7, 768
    gen = torch.Generator().manual_seed(37)

    def flipped(w_cl):          # [(tap', co), ci] = W[co, 44 - tap', ci]
        return w_cl.reshape(C, 45, C).flip(1).permute(1, 0, 2).reshape(45 * C, C)

    dyi = torch.randint(-2, 3, (B, To, H, W, C), generator=gen).float().to(DEV)
    wi = torch.randint(-1, 2, (C, 5, 3, 3, C), generator=gen).float().to(DEV)
    dyp = F.pad(dyi, (0, 0, 1, 1, 1, 1, 4, 4))
    ref = (_im2col(dyp, To + 4, H, W) @ flipped(wi)).bfloat16()
    dx = K.conv3d_k533_dgrad(dyp.bfloat16(), wi.bfloat16())
    assert torch.equal(dx.view(-1, C), ref), (dx.view(-1, C).float() - ref.float()).abs().max()
    # frame-major rows (row order 2): tiles keep only the temporal taps that read data frames of the padded gradient, stream-K with the
    # weighted plan ("conv_k_order" bit 6); rows back in standard order through the table - the same integers
    from shg_vqa_amd import _lib
    inv = K.conv_row_table_inv(B, To + 8, H, W, DEV, order=2)
    before = int(_lib.lib().shg_gemm_streamk_launches())
    for sw in (126, 62):
        _lib.set_tuning("conv_k_order", sw)
        try:
            dx2 = K.conv3d_k533_dgrad(dyp.bfloat16(), wi.bfloat16(), out_rows=inv, order=2)
            assert torch.equal(dx2.view(-1, C), ref), (sw, (dx2.view(-1, C).float() - ref.float()).abs().max())
        finally:
            _lib.set_tuning("conv_k_order", 126)
    assert int(_lib.lib().shg_gemm_streamk_launches()) == before + 1, "the weighted stream-K path was not taken"
    del ref
    dyr = F.pad(torch.randn(B, To, H, W, C, generator=gen).to(DEV).bfloat16(), (0, 0, 1, 1, 1, 1, 4, 4))
    wr = (torch.randn(C, 5, 3, 3, C, generator=gen) / math.sqrt(45 * C)).to(DEV).bfloat16()
    dx = K.conv3d_k533_dgrad(dyr, wr)
    rows = torch.randperm(B * (To + 4) * H * W, generator=gen)[:4096].to(DEV)
    ref = _im2col(dyr.float(), To + 4, H, W)[rows] @ flipped(wr.float())
    err = (dx.view(-1, C)[rows].float() - ref).abs().max().item()
    print("conv2 dgrad B=32: max |dx - ref| = %.3e (|ref| max %.3f)" % (err, ref.abs().max().item()))
    assert err <= 2 ** -8 * ref.abs().max().item() + 1e-3


def test_matcher_forward_rejects_class_ids_outside_the_logits():
    """matcher.py:74: out_prob[:, tgt_ids] raises IndexError for a label >= num_classes; the kernel itself clamps the read."""
    from shg_vqa_amd.matcher import HungarianMatcher
    m = HungarianMatcher(cost_class=1, loss_hg_per_frame=True, clip_len=2)
    logits = torch.randn(1, 8, 5, device=DEV)
    ok = [{"labels": [torch.tensor([1, 4]), torch.tensor([], dtype=torch.int64)]}]
    out = m({"pred_logits": logits}, ok)
    assert len(out) == 2 and len(out[0][0]) == 2 and len(out[1][0]) == 0
    with pytest.raises(IndexError):
        m({"pred_logits": logits}, [{"labels": [torch.tensor([1, 5]), torch.tensor([2])]}])
    with pytest.raises(ValueError):
        m({"pred_logits": logits}, [{"labels": [torch.tensor([1, 2, 3, 4, 1]), torch.tensor([2])]}])


def test_wgrad_group_is_exact_on_integer_data_and_handles_mixed_problem_lists(K):
    """shg_wgrad_group: the weight gradients of several layers in one grid of 256 x 256 tiles (the eight of a decoder layer here,
    with their real shapes), ragged row counts falling back to the plain GEMM, accumulation into a running fp32 sum; small
    integers make every sum exact, so a tile computed twice, never, or with another problem's operands shows."""
    gen = torch.Generator().manual_seed(5)

    def prob(rows, n_out, n_in, ld_extra=0):
        dyb = torch.randint(-2, 3, (rows, n_out + ld_extra), generator=gen).float().to(DEV).bfloat16()
        xb = torch.randint(-2, 3, (rows, n_in), generator=gen).float().to(DEV).bfloat16()
        dy = dyb[:, :n_out]
        gw = torch.randint(-3, 4, (n_out, n_in), generator=gen).float().to(DEV)
        return dy, xb, gw

    shapes = [(4096, 1536, 768, 768), (4096, 768, 768, 0), (4096, 768, 768, 0), (4096, 768, 768, 0), (12576, 1536, 768, 0),
              (4096, 768, 768, 0), (4096, 2048, 768, 0), (4096, 768, 2048, 0), (1536, 768, 768, 0), (1280, 3072, 768, 0),
              (5664, 768, 768, 0), (128, 264, 72, 0)]
    probs = [prob(*s_) for s_ in shapes]
    ref = [gw + dy.float().t() @ x.float() for dy, x, gw in probs]
    K.wgrad_group(probs)
    torch.cuda.synchronize()
    for i, ((dy, x, gw), r) in enumerate(zip(probs, ref)):
        assert torch.equal(gw, r), (i, shapes[i], (gw - r).abs().max().item())
    # a small group: split along the rows, partial tiles meet through fp32 atomics (integers: still exact)
    small = [prob(4096, 768, 768), prob(4096, 1536, 768)]
    ref = [gw + dy.float().t() @ x.float() for dy, x, gw in small]
    K.wgrad_group(small)
    for (dy, x, gw), r in zip(small, ref):
        assert torch.equal(gw, r)
    # a weight that is applied twice (the shared cross layer) has two problems adding into ONE gradient inside the grid
    a, b = prob(5664, 3072, 768), prob(5664, 3072, 768)
    c = prob(5664, 768, 768)
    ref = a[2] + a[0].float().t() @ a[1].float() + b[0].float().t() @ b[1].float()
    K.wgrad_group([a, c, (b[0], b[1], a[2])])
    assert torch.equal(a[2], ref), (a[2] - ref).abs().max()
    # a queue of three relation layers (3 x 108 tiles): launches of at most 256 workgroups, so problems are cut between launches -
    # on 8-tile boundaries (the 36-tile feed-forward weights) and at odd tile counts (27-tile fused q/k/v, 9-tile out-projection)
    layers = []
    for _ in range(3):
        layers += [prob(12576, 2304, 768), prob(12576, 768, 768), prob(12576, 3072, 768), prob(12576, 768, 3072)]
    ref = [gw + dy.float().t() @ x.float() for dy, x, gw in layers]
    K.wgrad_group(layers)
    torch.cuda.synchronize()
    for i, ((dy, x, gw), r) in enumerate(zip(layers, ref)):
        assert torch.equal(gw, r), (i, (gw - r).abs().max().item())
    # fp32 operands: the fallback path
    f32 = [(dy.float(), x.float(), gw.clone()) for dy, x, gw in probs[:3]]
    ref = [gw + dy.t() @ x for dy, x, gw in f32]
    K.wgrad_group(f32)
    for (dy, x, gw), r in zip(f32, ref):
        assert torch.equal(gw, r)


def test_loss_combine_matches_torch_autograd():
    """shg_loss_combine_fwd / bwd (agqaHGQA.py:344-378): total = bce * scale + rel[0] / rel[1] + act[0] / act[1], the reported
    scalars, and the gradients torch's autograd gives for the same expression."""
    from shg_vqa_amd import ops
    gen = torch.Generator().manual_seed(3)
    for scale in (1.0, 0.25):
        rel = (torch.rand(4, generator=gen) * 50 + 1).to(DEV).requires_grad_(True)
        act = (torch.rand(4, generator=gen) * 50 + 1).to(DEV).requires_grad_(True)
        with torch.no_grad():
            act[3] = 0.0                                     # no matched slot: class error 100 - 100 * x / max(0, 1)
        bce = (torch.rand(1, generator=gen) * 100).to(DEV).requires_grad_(True)
        total, diag = ops.combine_losses(rel, act, bce, scale)
        (total * 3.0).backward()
        got = [t.grad.clone() for t in (rel, act, bce)]
        for t in (rel, act, bce):
            t.grad = None
        ref = bce.sum() * scale + rel[0] / rel[1] + act[0] / act[1]
        (ref * 3.0).backward()
        assert torch.allclose(total, ref, rtol=1e-6)
        for a, b in zip(got, (rel.grad, act.grad, bce.grad)):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (a, b)
        exp = torch.stack([bce.detach().sum(), rel[0] / rel[1], act[0] / act[1], 100.0 - 100.0 * rel[2] / rel[3].clamp(min=1),
                           100.0 - 100.0 * act[2] / act[3].clamp(min=1)]).detach()
        assert torch.allclose(diag, exp, rtol=1e-6)
