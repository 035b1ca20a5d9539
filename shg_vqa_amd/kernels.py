"""Thin, shape-checked wrappers: torch tensors (device memory, current stream) -> C ABI calls.

Each function validates shapes / dtypes / contiguity on the host BEFORE the launch (a kernel that
faults can take the whole GPU host down) and then hands raw pointers to libshgvqa.so.
torch is used for allocation and stream plumbing only.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F32, MASK_FULL, MASK_KEY, MASK_NONE  # noqa: F401


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("expected float32 or bfloat16 tensor, got %s" % t.dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream (raw getter: ~10x cheaper than torch.cuda.current_stream())."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _need(cond, msg):
    if not cond:
        raise ValueError(msg)


def _dev(*ts):
    for t in ts:
        if t is not None:
            _need(t.is_cuda, "tensor must live on the GPU (the HIP path has no CPU fallback)")


# ------------------------------------------------------------------------------------------------
def hungarian_per_frame(logits, tgt, tgt_len, background=0, want_grid=True):
    """logits [N, R, C]; tgt [N, R] int64; tgt_len [N] int32 -> (query_idx, target_idx, grid) [N, R] int64."""
    _dev(logits, tgt, tgt_len)
    _need(logits.dim() == 3 and logits.is_contiguous(), "logits must be contiguous [N, R, C]")
    n, r, c = logits.shape
    _need(1 <= r <= 128, "queries per problem must be in [1, 128]")
    _need(tgt.shape == (n, r) and tgt.dtype == torch.int64 and tgt.is_contiguous(), "tgt must be int64 [N, R]")
    _need(tgt_len.shape == (n,) and tgt_len.dtype == torch.int32 and tgt_len.is_contiguous(), "tgt_len must be int32 [N]")
    oq = torch.empty((n, r), dtype=torch.int64, device=logits.device)
    ot = torch.empty_like(oq)
    grid = torch.empty_like(oq) if want_grid else None
    _lib.call("shg_hungarian_per_frame", logits.data_ptr(), _dt(logits), n, r, c, tgt.data_ptr(), tgt_len.data_ptr(),
              int(background), oq.data_ptr(), ot.data_ptr(), _p(grid), _stream())
    return oq, ot, grid


def lsap_batched(cost, n_cols):
    """cost [N, R, Cmax] fp32; n_cols [N] int32 -> (rows, cols) [N, Cmax] int64, -1 padded."""
    _dev(cost, n_cols)
    _need(cost.dim() == 3 and cost.dtype == torch.float32 and cost.is_contiguous(), "cost must be fp32 [N,R,C]")
    n, r, cm = cost.shape
    _need(1 <= cm <= r <= 8, "need 1 <= Cmax <= R <= 8")
    _need(n_cols.shape == (n,) and n_cols.dtype == torch.int32, "n_cols must be int32 [N]")
    orow = torch.empty((n, cm), dtype=torch.int64, device=cost.device)
    ocol = torch.empty_like(orow)
    _lib.call("shg_lsap_batched", cost.data_ptr(), n, r, cm, n_cols.data_ptr(), orow.data_ptr(), ocol.data_ptr(), _stream())
    return orow, ocol


# ------------------------------------------------------------------------------------------------
def weighted_ce_fwd(logits, target, class_weight, background=0):
    """logits [rows, C] contiguous; target [rows] int64 -> (row_stats [4, rows], sums [4])."""
    _dev(logits, target, class_weight)
    _need(logits.dim() == 2 and logits.is_contiguous(), "logits must be contiguous [rows, C]")
    rows, c = logits.shape
    _need(target.shape == (rows,) and target.dtype == torch.int64 and target.is_contiguous(), "target int64 [rows]")
    _need(class_weight.shape == (c,) and class_weight.dtype == torch.float32, "class_weight fp32 [C]")
    stats = torch.empty((4, rows), dtype=torch.float32, device=logits.device)
    sums = torch.empty(4, dtype=torch.float32, device=logits.device)
    _lib.call("shg_weighted_ce_fwd", logits.data_ptr(), _dt(logits), rows, c, target.data_ptr(), class_weight.data_ptr(),
              int(background), stats.data_ptr(), sums.data_ptr(), _stream())
    return stats, sums


def pad8(n):
    return (n + 7) // 8 * 8


def weighted_ce_bwd(logits, target, class_weight, stats, sums, gscale=None, padded=False):
    """-> dlogits [rows, C] (or [rows, pad8(C)] with zeroed pad columns when padded)."""
    _dev(logits, target, class_weight, stats, sums, gscale)
    rows, c = logits.shape
    _need(stats.shape == (4, rows) and sums.numel() >= 2, "stats/sums from weighted_ce_fwd required")
    ldd = pad8(c) if padded else c
    d = torch.empty((rows, ldd), dtype=logits.dtype, device=logits.device)
    _lib.call("shg_weighted_ce_bwd", logits.data_ptr(), _dt(logits), rows, c, target.data_ptr(), class_weight.data_ptr(),
              stats.data_ptr(), sums.data_ptr(), _p(gscale), d.data_ptr(), ldd, _stream())
    return d


def bce_logits(logits, target, gscale=None, want_grad=True, padded=False):
    """-> (loss [1] = C * mean BCE, dlogits or None; [rows, pad8(C)] when padded)."""
    _dev(logits, target, gscale)
    _need(logits.dim() == 2 and logits.is_contiguous(), "logits must be contiguous [rows, C]")
    _need(target.shape == logits.shape and target.dtype == torch.float32 and target.is_contiguous(), "target fp32 [rows, C]")
    rows, c = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    ldd = pad8(c) if padded else c
    d = torch.empty((rows, ldd), dtype=logits.dtype, device=logits.device) if want_grad else None
    _lib.call("shg_bce_logits_fwd_bwd", logits.data_ptr(), _dt(logits), rows, c, target.data_ptr(), _p(gscale),
              loss.data_ptr(), _p(d), ldd, _stream())
    return loss, d


def loss_combine_fwd(rel_sums, act_sums, bce, bce_scale):
    """-> (total [1], diag [5] = bce, rel CE, act CE, rel class error %, act class error %)."""
    _dev(rel_sums, act_sums, bce)
    for t in (rel_sums, act_sums):
        _need(t.dtype == torch.float32 and t.numel() == 4 and t.is_contiguous(), "loss sums must be contiguous fp32 [4]")
    _need(bce.dtype == torch.float32 and bce.numel() == 1, "bce must be fp32 [1]")
    total = torch.empty(1, dtype=torch.float32, device=bce.device)
    diag = torch.empty(5, dtype=torch.float32, device=bce.device)
    _lib.call("shg_loss_combine_fwd", rel_sums.data_ptr(), act_sums.data_ptr(), bce.data_ptr(), float(bce_scale), total.data_ptr(),
              diag.data_ptr(), _stream())
    return total, diag


def loss_combine_bwd(d_total, rel_sums, act_sums, bce_scale):
    """-> (d_rel_sums [4], d_act_sums [4], d_bce [1])."""
    _dev(d_total, rel_sums, act_sums)
    out = torch.empty(9, dtype=torch.float32, device=rel_sums.device)
    if d_total is not None:
        _need(d_total.dtype == torch.float32 and d_total.numel() == 1, "d_total must be fp32 [1]")
    _lib.call("shg_loss_combine_bwd", _p(d_total), rel_sums.data_ptr(), act_sums.data_ptr(), float(bce_scale), out.data_ptr(),
              out[4:].data_ptr(), out[8:].data_ptr(), _stream())
    return out[:4], out[4:8], out[8:]


# ------------------------------------------------------------------------------------------------
def _rows_cols(x):
    _need(x.is_contiguous() and x.dim() >= 2, "activation must be contiguous with >= 2 dims")
    return x.numel() // x.shape[-1], x.shape[-1]


def _f32vec(t, n, name):
    if t is not None:
        _need(t.dtype == torch.float32 and t.numel() == n and t.is_contiguous(), "%s must be contiguous fp32 [%d]" % (name, n))


def colsum_partials(rows):
    return _lib.lib().shg_colsum_partials(int(rows))


def colsum(x2d, out, accumulate):
    """out[c] (+)= sum_r x2d[r, c]; x2d may be a row-strided view."""
    _dev(x2d, out)
    _need(x2d.dim() == 2 and x2d.stride(1) == 1, "x must be 2-D with contiguous inner dim")
    rows, cols = x2d.shape
    vec = 16 // x2d.element_size()
    if accumulate and cols % vec == 0 and x2d.stride(0) % vec == 0 and x2d.data_ptr() % 16 == 0 and cols <= 512 * vec:
        _need(out.numel() == cols and out.dtype == torch.float32 and out.is_contiguous(), "out must be fp32 [cols]")
        _lib.call("shg_colsum_accumulate", x2d.data_ptr(), _dt(x2d), rows, cols, x2d.stride(0), out.data_ptr(), _stream())
        return
    npart = colsum_partials(rows)
    part = torch.empty((npart, cols), dtype=torch.float32, device=x2d.device)
    _lib.call("shg_colsum_partial", x2d.data_ptr(), _dt(x2d), rows, cols, x2d.stride(0), part.data_ptr(), npart, _stream())
    colsum_finish(part, out, accumulate)


def colsum_finish_multi(partials, outs):
    """outs[i] += column sums of partials[i] ([n_partials, cols] each, same shape), one launch for up to 4 pairs."""
    n = len(partials)
    _need(1 <= n <= 4 and len(outs) == n, "1..4 (partial, out) pairs")
    n_partials, cols = partials[0].shape
    arr = ctypes.c_void_p * n
    pa, oa = arr(*[t.data_ptr() for t in partials]), arr(*[t.data_ptr() for t in outs])     # host arrays, alive over the call
    _lib.call("shg_colsum_finish_multi", ctypes.addressof(pa), ctypes.addressof(oa), n, n_partials, cols, _stream())


def colsum_finish(partial, out, accumulate):
    n_partials, cols = partial.shape
    _need(out.numel() == cols and out.dtype == torch.float32 and out.is_contiguous(), "out must be fp32 [cols]")
    _lib.call("shg_colsum_finish", partial.data_ptr(), n_partials, cols, out.data_ptr(), 1 if accumulate else 0, _stream())


def bias_act_fwd(x, bias, act, p_drop=0.0, seed_state=None, stream_id=0):
    _dev(x, bias, seed_state)
    rows, cols = _rows_cols(x)
    _f32vec(bias, cols, "bias")
    _need(p_drop == 0.0 or seed_state is not None, "dropout needs seed_state")
    y = torch.empty_like(x)
    _lib.call("shg_bias_act_fwd", x.data_ptr(), _p(bias), y.data_ptr(), _dt(x), rows, cols, act, float(p_drop),
              _p(seed_state), int(stream_id), _stream())
    return y


def bias_act_bwd(x, bias, dy, act, p_drop=0.0, seed_state=None, stream_id=0, want_dbias=True, dy_groups=None, out2=None, x_rows=None):
    """-> (dx, dbias_partial [n_partials, cols] or None).
    dy_groups = (rows_per_group, group_stride, row_offset): dy is a contiguous [groups * group_stride, cols] tensor of which every
    group contributes rows row_offset .. row_offset + rows_per_group (shg_bias_act_bwd_view); out2 = (buffer, int32 row table):
    a second copy of dx, row r at buffer row table[r]."""
    _dev(x, bias, dy, seed_state)
    rows, cols = _rows_cols(x)
    if dy_groups is None:
        _need(dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous(), "dy must match x")
        rpg = gs = off = 0
    else:
        rpg, gs, off = (int(v) for v in dy_groups)
        _need(dy.dtype == x.dtype and dy.is_contiguous() and dy.shape[-1] == cols and rpg > 0 and rows % rpg == 0 and gs >= rpg + off >= rpg
              and dy.numel() == rows // rpg * gs * cols, "dy view does not fit")
    dx = torch.empty_like(x)
    npart = colsum_partials(rows)
    part = torch.empty((npart, cols), dtype=torch.float32, device=x.device) if want_dbias else None
    buf2 = tbl2 = None
    if out2 is not None:
        buf2, tbl2 = out2
        _dev(buf2, tbl2)
        _need(buf2.dtype == x.dtype and buf2.is_contiguous() and buf2.shape[-1] == cols, "out2 buffer must be contiguous [.., cols] of x's dtype")
        _need(tbl2.dtype == torch.int32 and tbl2.numel() == rows and tbl2.is_contiguous(), "out2 row table must be int32 [rows]")
    if x_rows is not None:                         # result row r <-> row x_rows[r] of x and dx (shg_bias_act_bwd_rows)
        _dev(x_rows)
        _need(x_rows.dtype == torch.int32 and x_rows.numel() == rows and x_rows.is_contiguous() and p_drop == 0.0, "x_rows: int32 [rows], no dropout")
    _lib.call("shg_bias_act_bwd_rows", x.data_ptr(), _p(bias), dy.data_ptr(), dx.data_ptr(), _p(part), npart, _dt(x), rows,
              cols, act, float(p_drop), _p(seed_state), int(stream_id), rpg, gs, off, _p(buf2), _p(tbl2), _p(x_rows), _stream())
    return dx, part


def ln_fwd(x, bias, residual, gamma, beta, eps, act=ACT_NONE, p_drop=0.0, seed_state=None, stream_id=0, save_z=True):
    """y = LN(dropout(act(x + bias)) + residual).  -> (y, z or None, mean, rstd)."""
    _dev(x, bias, residual, gamma, beta, seed_state)
    rows, cols = _rows_cols(x)
    _f32vec(bias, cols, "bias")
    _f32vec(gamma, cols, "gamma")
    _f32vec(beta, cols, "beta")
    if residual is not None:
        _need(residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous(), "residual must match x")
    _need(p_drop == 0.0 or seed_state is not None, "dropout needs seed_state")
    y = torch.empty_like(x)
    z = torch.empty_like(x) if save_z else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.call("shg_bias_act_drop_res_ln_fwd", x.data_ptr(), _p(bias), _p(residual), gamma.data_ptr(), beta.data_ptr(),
              y.data_ptr(), _p(z), mean.data_ptr(), rstd.data_ptr(), _dt(x), rows, cols, act, float(eps), float(p_drop),
              _p(seed_state), int(stream_id), _stream())
    return y, z, mean, rstd


def ln_bwd(dy, z, x, bias, gamma, mean, rstd, act=ACT_NONE, p_drop=0.0, seed_state=None, stream_id=0,
           want_dx=True, want_dres=True, want_dbias=True):
    """-> (dx, dres, dgamma_partial, dbeta_partial, dbias_partial)."""
    _dev(dy, z, x, bias, gamma, mean, rstd, seed_state)
    rows, cols = _rows_cols(dy)
    _need(z.shape == dy.shape and z.dtype == dy.dtype and z.is_contiguous(), "z must match dy")
    if act != ACT_NONE:
        _need(x is not None and x.shape == dy.shape and x.dtype == dy.dtype, "x required when act != NONE")
    _need(mean.numel() == rows and rstd.numel() == rows, "mean/rstd must have one entry per row")
    npart = colsum_partials(rows)
    dx = torch.empty_like(dy) if want_dx else None
    dres = torch.empty_like(dy) if want_dres else None
    parts = torch.empty((3 if want_dbias else 2, npart, cols), dtype=torch.float32, device=dy.device)
    dg, db = parts[0], parts[1]
    dbi = parts[2] if want_dbias else None
    _lib.call("shg_bias_act_drop_res_ln_bwd", dy.data_ptr(), z.data_ptr(), _p(x), _p(bias), gamma.data_ptr(),
              mean.data_ptr(), rstd.data_ptr(), _p(dx), _p(dres), dg.data_ptr(), db.data_ptr(), _p(dbi), npart, _dt(dy),
              rows, cols, act, float(p_drop), _p(seed_state), int(stream_id), _stream())
    return dx, dres, dg, db, dbi


# ------------------------------------------------------------------------------------------------
def _attn_view(t, name):
    """t: [B, S, H*64]-like view with unit stride over the last dim.  -> (B, S, bstride, sstride)."""
    _need(t.dim() == 3 and t.stride(2) == 1, "%s must be [B, S, H*64] with a contiguous last dim" % name)
    return t.shape[0], t.shape[1], t.stride(0), t.stride(1)


def _mask_args(mask_kind, mask, b, sq, sk):
    if mask_kind == MASK_NONE:
        return None
    _need(mask is not None and mask.dtype == torch.float32 and mask.is_contiguous(), "mask must be contiguous fp32")
    if mask_kind == MASK_KEY:
        _need(mask.numel() == b * sk, "key mask must have B*Sk elements")
    else:
        _need(mask.numel() == sq * sk, "full mask must have Sq*Sk elements")
    return mask


def attention_keep_mask(b, heads, sq, sk, device):
    """Buffer for the forward's dropout keep decisions (lane masks, csrc/attention.hip): int64 words, 128-byte aligned."""
    n = _lib.lib().shg_attention_keep_mask_bytes(b, heads, sq, sk) // 8
    buf = torch.empty(n + 16, dtype=torch.int64, device=device)
    off = (-buf.data_ptr() // 8) % 16
    return buf[off:off + n]


def attention_fwd(q, k, v, heads, mask_kind=MASK_NONE, mask=None, scale=0.125, p_drop=0.0, seed_state=None, stream_id=0, keep_mask=None):
    """q [B,Sq,H*64], k/v [B,Sk,H*64] (views into fused projections allowed) -> (o [B,Sq,H*64], lse [B,H,Sq]).
    With dropout the keep decisions are written to `keep_mask` (attention_keep_mask(...)); when none is passed one is allocated
    and travels with the returned lse tensor (attribute _shg_keep), where attention_bwd finds it."""
    _dev(q, k, v, mask, seed_state)
    b, sq, qb, qs = _attn_view(q, "q")
    b2, sk, kb, ks = _attn_view(k, "k")
    b3, sk2, vb, vs = _attn_view(v, "v")
    _need(b == b2 == b3 and sk == sk2, "batch / key length mismatch")
    _need(q.shape[2] == heads * 64 and k.shape[2] == heads * 64 and v.shape[2] == heads * 64, "head dim must be 64")
    _need(q.dtype == k.dtype == v.dtype, "q/k/v dtype mismatch")
    mask = _mask_args(mask_kind, mask, b, sq, sk)
    _need(p_drop == 0.0 or seed_state is not None, "dropout needs seed_state")
    o = torch.empty((b, sq, heads * 64), dtype=q.dtype, device=q.device)
    lse = torch.empty((b, heads, sq), dtype=torch.float32, device=q.device)
    if p_drop > 0.0 and keep_mask is None:
        keep_mask = attention_keep_mask(b, heads, sq, sk, q.device)
    if keep_mask is not None:
        _dev(keep_mask)
        _need(keep_mask.dtype == torch.int64 and keep_mask.is_contiguous()
              and keep_mask.numel() * 8 >= _lib.lib().shg_attention_keep_mask_bytes(b, heads, sq, sk), "keep_mask too small")
        lse._shg_keep = keep_mask
    _lib.call("shg_attention_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), _dt(q), b,
              heads, sq, sk, qb, qs, kb, ks, vb, vs, mask_kind, _p(mask), float(scale), float(p_drop), _p(seed_state),
              int(stream_id), _p(keep_mask), _stream())
    return o, lse


def attention_bwd(q, k, v, o, d_o, lse, dq, dk, dv, heads, mask_kind=MASK_NONE, mask=None, scale=0.125, p_drop=0.0,
                  seed_state=None, stream_id=0, keep_mask=None, dbias=(None, None, None)):
    """Writes dq/dk/dv (views with the same layout rules as q/k/v).  With dropout: keep_mask = the buffer the forward call
    filled (default: the one attention_fwd attached to `lse`).  dbias = (dbias_q, dbias_k, dbias_v): fp32 [H*64] vectors (or
    None) that receive += the column sums of dq / dk / dv."""
    if p_drop > 0.0 and keep_mask is None:
        keep_mask = getattr(lse, "_shg_keep", None)
        _need(keep_mask is not None, "attention_bwd with dropout needs the forward call's keep_mask")
    _dev(q, k, v, o, d_o, lse, dq, dk, dv, mask, seed_state)
    b, sq, qb, qs = _attn_view(q, "q")
    _, sk, kb, ks = _attn_view(k, "k")
    _, _, vb, vs = _attn_view(v, "v")
    _need(o.is_contiguous() and d_o.is_contiguous() and o.shape == (b, sq, heads * 64) and d_o.shape == o.shape,
          "o and d_o must be contiguous [B,Sq,H*64]")
    _need(o.dtype == q.dtype and d_o.dtype == q.dtype, "o/d_o dtype mismatch")
    _need(lse.shape == (b, heads, sq) and lse.dtype == torch.float32 and lse.is_contiguous(), "lse [B,H,Sq] fp32")
    _, sq2, dqb, dqs = _attn_view(dq, "dq")
    _, sk2, dkb, dks = _attn_view(dk, "dk")
    _, sk3, dvb, dvs = _attn_view(dv, "dv")
    _need(sq2 == sq and sk2 == sk and sk3 == sk and dq.dtype == q.dtype and dk.dtype == q.dtype and dv.dtype == q.dtype,
          "gradient buffers must match q/k/v")
    mask = _mask_args(mask_kind, mask, b, sq, sk)
    for t in dbias:
        if t is not None:
            _dev(t)
            _need(t.dtype == torch.float32 and t.numel() == heads * 64 and t.is_contiguous(), "dbias vectors must be contiguous fp32 [H*64]")
    delta = torch.empty((b, heads, sq), dtype=torch.float32, device=q.device)
    _lib.call("shg_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
              delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _dt(q), b, heads, sq, sk, qb, qs, kb, ks,
              vb, vs, dqb, dqs, dkb, dks, dvb, dvs, mask_kind, _p(mask), float(scale), float(p_drop), _p(seed_state),
              int(stream_id), _p(keep_mask), _p(dbias[0]), _p(dbias[1]), _p(dbias[2]), _stream())


# ------------------------------------------------------------------------------------------------
def gemm_act(a, b, out, bias, act, pre=None, p_drop=0.0, seed_state=None, stream_id=0):
    """out = dropout(act(a . b^T + bias)) in one kernel (a [M,K], b [N,K]); pre [M,N] receives a . b^T + bias."""
    _dev(a, b, out, bias, pre, seed_state)
    m, k = a.shape
    n = b.shape[0]
    _need(a.dtype == b.dtype and b.shape[1] == k and out.shape == (m, n) and out.stride(1) == 1 and a.stride(1) == 1
          and b.stride(1) == 1, "gemm_act: shape / layout mismatch")
    if pre is not None:
        _need(pre.shape == (m, n) and pre.is_contiguous() and pre.dtype == out.dtype, "pre must be contiguous [M, N] of out's dtype")
    _f32vec(bias, n, "bias")
    _need(p_drop == 0.0 or seed_state is not None, "dropout needs seed_state")
    _lib.call("shg_gemm_act", a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), _dt(a), _dt(out), m, n, k, a.stride(0),
              b.stride(0), out.stride(0), 1, 1, int(act), _p(pre), float(p_drop), _p(seed_state), int(stream_id), _stream())
    return out


def gemm_dact(dy, w, dx, pre, dbias, act, p_drop=0.0, seed_state=None, stream_id=0):
    """dx = dropout_mask(dy . w) * act'(pre), dbias += column sums of dx.  dy [M,K]; w [K,N] (a forward weight
    [out, in]); pre / dx [M,N]; dbias fp32 [N] or None.  The mask is the one gemm_act applied in the forward."""
    _dev(dy, w, dx, pre, dbias, seed_state)
    m, k = dy.shape
    n = w.shape[1]
    _need(w.shape[0] == k and dx.shape == (m, n) and pre.shape == (m, n), "gemm_dact: shape mismatch")
    _need(dy.dtype == w.dtype == dx.dtype == pre.dtype, "gemm_dact: one dtype for all operands")
    _need(dy.stride(1) == 1 and w.stride(1) == 1 and dx.stride(1) == 1 and pre.is_contiguous(), "gemm_dact: layout")
    _f32vec(dbias, n, "dbias")
    _need(p_drop == 0.0 or seed_state is not None, "dropout needs seed_state")
    _lib.call("shg_gemm_dact", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), pre.data_ptr(), _p(dbias), _dt(dy), m, n, k,
              dy.stride(0), w.stride(0), dx.stride(0), int(act), float(p_drop), _p(seed_state), int(stream_id), _stream())
    return dx


def gemm(a, b, out, bias=None, a_kmajor=True, b_kmajor=True, accumulate=False):
    """out[M,N] (+)= A . B (+ bias).  a: [M,K] if a_kmajor else [K,M]; b: [N,K] if b_kmajor else [K,N].
    2-D tensors with unit inner stride (row stride = leading dimension)."""
    _dev(a, b, out, bias)
    for t, nm in ((a, "a"), (b, "b"), (out, "out")):
        _need(t.dim() == 2 and t.stride(1) == 1, "%s must be 2-D with a contiguous inner dimension" % nm)
    _need(a.dtype == b.dtype, "a/b dtype mismatch")
    m, k = (a.shape if a_kmajor else (a.shape[1], a.shape[0]))
    n, k2 = (b.shape if b_kmajor else (b.shape[1], b.shape[0]))
    _need(k == k2, "contraction size mismatch: %d vs %d" % (k, k2))
    _need(out.shape == (m, n), "out must be [%d, %d]" % (m, n))
    _f32vec(bias, n, "bias")
    _lib.call("shg_gemm", a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), _dt(a), _dt(out), m, n, k, a.stride(0),
              b.stride(0), out.stride(0), 1 if a_kmajor else 0, 1 if b_kmajor else 0, 1 if accumulate else 0, _stream())
    return out


def tokens_assemble(tok, cls, pos, out=None):
    """tok [B, T, C] (compute dtype), cls fp32 [C] (any shape with C elements), pos fp32 [>= T + 1, C] -> [B, T + 1, C]."""
    _dev(tok, cls, pos, out)
    B, T, C = tok.shape
    _need(tok.is_contiguous() and cls.dtype == torch.float32 and cls.numel() == C and cls.is_contiguous(), "tok contiguous, cls fp32 [C]")
    _need(pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[-1] == C and pos.numel() >= (T + 1) * C, "pos fp32 [>= T + 1, C]")
    if out is None:
        out = torch.empty((B, T + 1, C), dtype=tok.dtype, device=tok.device)
    _lib.call("shg_tokens_assemble", tok.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), _dt(tok), B, T + 1, C, _stream())
    return out


def wgrad_group(problems):
    """problems: list of (dy [rows, n_out], x [rows, n_in], gw fp32 [n_out, n_in]); gw += dy^T x for each, grouped launches
    where the shapes allow (shg_wgrad_group)."""
    n = len(problems)
    arr = (_lib.WgradProblemT * max(n, 1))()
    for e, (dy, x, gw) in zip(arr, problems):
        _dev(dy, x, gw)
        _need(dy.dim() == 2 and x.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[0] == x.shape[0] and
              dy.dtype == x.dtype, "wgrad_group: dy [rows, n_out] and x [rows, n_in] of one dtype")
        _need(gw.dtype == torch.float32 and gw.is_contiguous() and tuple(gw.shape) == (dy.shape[1], x.shape[1]), "gw fp32 [n_out, n_in]")
        e.dy, e.x, e.gw = dy.data_ptr(), x.data_ptr(), gw.data_ptr()
        e.rows, e.n_out, e.n_in, e.ldy, e.ldx = dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0)
    if n:
        import ctypes
        _lib.call("shg_wgrad_group", ctypes.addressof(arr), n, _dt(problems[0][0]), _stream())


_conv_ws = {}


def conv_workspace(B, T, H, W, device, order=0):
    """Gather tables of the (5,3,3) conv for one input shape and row order (0: standard, 1: position-major, include/shg_vqa.h;
    built once, cached)."""
    key = (B, T, H, W, str(device), int(order))
    ws = _conv_ws.get(key)
    if ws is None:
        nbytes = _lib.lib().shg_conv3d_k533_workspace_bytes_ex(B, T, H, W, int(order))
        _need(nbytes > 0, "bad conv shape / row order")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.call("shg_conv3d_k533_prepare_ex", ws.data_ptr(), B, T, H, W, int(order), _stream())
        _conv_ws[key] = ws
    return ws


def conv_row_table(B, T, H, W, device, order=1):
    """int32 [B (T-4) H W]: row (in `order`: 1 position-major, 2 frame-major) of every standard row (the third table of that workspace)."""
    ws = conv_workspace(B, T, H, W, device, order)
    seg = ws.numel() // 4
    return ws[2 * seg:3 * seg].view(torch.int32)[:B * (T - 4) * H * W]


def conv_row_table_inv(B, T, H, W, device, order=1):
    """int32 [B (T-4) H W]: standard row of every row of `order` (the fourth table)."""
    ws = conv_workspace(B, T, H, W, device, order)
    seg = ws.numel() // 4
    return ws[3 * seg:].view(torch.int32)[:B * (T - 4) * H * W]


_streamk_ws = {}


def streamk_workspace(device):
    """Partial-sum slots + flags of the stream-K conv forward for torch's CURRENT stream on `device` (one workspace per
    (device, stream): launches on one stream are ordered, two streams may overlap).  Allocated and zeroed on first use;
    None while the stream is being captured into a graph and has none yet (an allocation is illegal there)."""
    st = _stream()
    key = (torch.device(device).index or 0, st)
    ws = _streamk_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        ws = torch.empty(_lib.lib().shg_streamk_workspace_bytes(), dtype=torch.uint8, device=device)
        _lib.call("shg_streamk_workspace_init", ws.data_ptr(), st)
        _streamk_ws[key] = ws
    return ws


def conv3d_k533_fwd(x_cl, w_cl, bias, act=ACT_NONE, pad_out=False, out=None, want_pre=False, pre_out=None, order=0, pre_rows=None,
                    y_rows=None):
    """x_cl [B,T,H+2,W+2,Cin] (zero border); w_cl [Cout,5,3,3,Cin]; -> y [B,T-4,H,W,Cout]
    (or written into the interior of a zero-bordered [B,T-4,H+2,W+2,Cout] buffer when pad_out)."""
    _dev(x_cl, w_cl, bias, out)
    _need(x_cl.dim() == 5 and x_cl.is_contiguous() and w_cl.dim() == 5 and w_cl.is_contiguous(), "channels-last 5-D tensors")
    B, T, Hp, Wp, cin = x_cl.shape
    H, W = Hp - 2, Wp - 2
    cout = w_cl.shape[0]
    _need(tuple(w_cl.shape) == (cout, 5, 3, 3, cin) and w_cl.dtype == x_cl.dtype, "weight must be [Cout,5,3,3,Cin] of x's dtype")
    _f32vec(bias, cout, "bias")
    ws = conv_workspace(B, T, H, W, x_cl.device, order)     # (order 1: the dense outputs - pre, out without pad_out - have position-major rows)
    if out is None:
        shape = (B, T - 4, Hp, Wp, cout) if pad_out else (B, T - 4, H, W, cout)
        out = (torch.zeros if pad_out else torch.empty)(shape, dtype=x_cl.dtype, device=x_cl.device)
    _need(out.is_contiguous() and out.dtype == x_cl.dtype, "out must be contiguous and of x's dtype")
    pre = None
    if want_pre:
        pre = pre_out if pre_out is not None else torch.empty((B, T - 4, H, W, cout), dtype=x_cl.dtype, device=x_cl.device)
        _need(pre.is_contiguous() and pre.dtype == x_cl.dtype and pre.numel() == B * (T - 4) * H * W * cout, "bad pre_out")
    sk = streamk_workspace(x_cl.device) if x_cl.dtype == torch.bfloat16 else None
    if pre_rows is not None:                      # row m of pre goes to row pre_rows[m] (e.g. conv_row_table: a standard-order forward, pre position-major)
        _dev(pre_rows)
        _need(pre_rows.dtype == torch.int32 and pre_rows.is_contiguous() and pre_rows.numel() == B * (T - 4) * H * W, "pre_rows must be int32 [rows]")
    if y_rows is not None:                        # dense output row m goes to row y_rows[m]
        _dev(y_rows)
        _need(not pad_out and y_rows.dtype == torch.int32 and y_rows.is_contiguous() and y_rows.numel() == B * (T - 4) * H * W,
              "y_rows: int32 [rows], dense output only")
    _lib.call("shg_conv3d_k533_fwd_rows", x_cl.data_ptr(), w_cl.data_ptr(), _p(bias), out.data_ptr(), _dt(x_cl), B, T, H, W, cin,
              cout, act, 1 if pad_out else 0, _p(pre), _p(pre_rows), _p(y_rows), int(order), ws.data_ptr(), _p(sk), _stream())
    return (out, pre) if want_pre else out


def conv3d_k533_wgrad(x_cl, dy, dw, accumulate=False, c0=0, cn=None, order=0):
    """dw [Cout,5,3,3,Cin] fp32 (+)= sum over positions of dy [B,T-4,H,W,Cout] x gathered x_cl; c0 / cn: only the output
    channels [c0, c0 + cn) (rows of dw); order: row order of dy's B (T-4) H W rows (1 = position-major)."""
    _dev(x_cl, dy, dw)
    B, T, Hp, Wp, cin = x_cl.shape
    H, W = Hp - 2, Wp - 2
    cout = dy.shape[-1]
    _need(x_cl.is_contiguous() and dy.is_contiguous() and tuple(dy.shape) == (B, T - 4, H, W, cout) and dy.dtype == x_cl.dtype,
          "dy must be contiguous [B,T-4,H,W,Cout] of x's dtype")
    _need(tuple(dw.shape) == (cout, 5, 3, 3, cin) and dw.dtype == torch.float32 and dw.is_contiguous(), "dw fp32 [Cout,5,3,3,Cin]")
    ws = conv_workspace(B, T, H, W, x_cl.device, order)
    if order:
        _lib.call("shg_conv3d_k533_wgrad_ex", x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), _dt(x_cl), B, T, H, W, cin, cout,
                  int(c0), int(cout - c0 if cn is None else cn), 1 if accumulate else 0, None, int(order), ws.data_ptr(), _stream())
    elif c0 == 0 and (cn is None or cn == cout):
        _lib.call("shg_conv3d_k533_wgrad", x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), _dt(x_cl), B, T, H, W, cin, cout,
                  1 if accumulate else 0, ws.data_ptr(), _stream())
    else:
        _lib.call("shg_conv3d_k533_wgrad_slice", x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), _dt(x_cl), B, T, H, W, cin, cout,
                  int(c0), int(cn), 1 if accumulate else 0, ws.data_ptr(), _stream())
    return dw


def conv3d_k533_wgrad_sumsq(x_cl, dy, dw, sumsq, c0=0, cn=None, order=0):
    """Overwrite form: dw rows [c0, c0 + cn) = the weight gradient (no zeroed destination needed), sumsq[0] (float64 [1]) += the sum
    of their squares (shg_conv3d_k533_wgrad_sumsq)."""
    _dev(x_cl, dy, dw, sumsq)
    B, T, Hp, Wp, cin = x_cl.shape
    H, W = Hp - 2, Wp - 2
    cout = dy.shape[-1]
    _need(x_cl.is_contiguous() and dy.is_contiguous() and tuple(dy.shape) == (B, T - 4, H, W, cout) and dy.dtype == x_cl.dtype,
          "dy must be contiguous [B,T-4,H,W,Cout] of x's dtype")
    _need(tuple(dw.shape) == (cout, 5, 3, 3, cin) and dw.dtype == torch.float32 and dw.is_contiguous(), "dw fp32 [Cout,5,3,3,Cin]")
    _need(sumsq.dtype == torch.float64 and sumsq.numel() == 1, "sumsq must be float64 [1]")
    ws = conv_workspace(B, T, H, W, x_cl.device, order)
    if order:
        _lib.call("shg_conv3d_k533_wgrad_ex", x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), _dt(x_cl), B, T, H, W, cin, cout,
                  int(c0), int(cout - c0 if cn is None else cn), 0, sumsq.data_ptr(), int(order), ws.data_ptr(), _stream())
    else:
        _lib.call("shg_conv3d_k533_wgrad_sumsq", x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), _dt(x_cl), B, T, H, W, cin, cout,
                  int(c0), int(cout - c0 if cn is None else cn), sumsq.data_ptr(), ws.data_ptr(), _stream())
    return dw


def conv3d_k533_dgrad(dy_padded, w_cl, out_rows=None, order=0):
    """dy_padded [B,To+8,H+2,W+2,Cout] (dy zero-padded by 4 in T, 1 in H/W); w_cl [Cout,5,3,3,Cin] -> dx [B,To+4,H,W,Cin];
    out_rows (int32 [B (To+4) H W], e.g. conv_row_table of the layer below): row m of dx goes to row out_rows[m]."""
    _dev(dy_padded, w_cl, out_rows)
    _need(dy_padded.dim() == 5 and dy_padded.is_contiguous() and w_cl.is_contiguous(), "contiguous channels-last tensors")
    B, Tp, Hp, Wp, cout = dy_padded.shape
    H, W = Hp - 2, Wp - 2
    cin = w_cl.shape[4]
    _need(tuple(w_cl.shape) == (cout, 5, 3, 3, cin) and w_cl.dtype == dy_padded.dtype, "weight must be [Cout,5,3,3,Cin] of dy's dtype")
    ws = conv_workspace(B, Tp, H, W, dy_padded.device, order)         # (order 2: frame-major rows - out_rows then maps THOSE rows)
    sk = streamk_workspace(dy_padded.device) if (order == 2 and dy_padded.dtype == torch.bfloat16) else None
    dx = torch.empty((B, Tp - 4, H, W, cin), dtype=dy_padded.dtype, device=dy_padded.device)
    if out_rows is not None:
        _need(out_rows.dtype == torch.int32 and out_rows.is_contiguous() and out_rows.numel() == B * (Tp - 4) * H * W, "out_rows must be int32 [rows of dx]")
    _lib.call("shg_conv3d_k533_dgrad_rows", dy_padded.data_ptr(), w_cl.data_ptr(), dx.data_ptr(), _dt(dy_padded), B, Tp, H, W, cin,
              cout, _p(out_rows), int(order), ws.data_ptr(), _p(sk), _stream())
    return dx


def ncdhw_to_padded_cl(x, dtype, out=None):
    """[B,C,T,H,W] fp32 -> [B,T,H+2,W+2,C] `dtype`, zero border (`out`: a zero-bordered buffer to reuse)."""
    _dev(x, out)
    _need(x.dim() == 5 and x.dtype == torch.float32 and x.is_contiguous(), "x must be contiguous fp32 NCDHW")
    B, C, T, H, W = x.shape
    y = out if out is not None else torch.zeros((B, T, H + 2, W + 2, C), dtype=dtype, device=x.device)
    _need(tuple(y.shape) == (B, T, H + 2, W + 2, C) and y.dtype == dtype and y.is_contiguous(), "bad out buffer")
    _lib.call("shg_ncdhw_to_padded_cl", x.data_ptr(), y.data_ptr(), _dt(y), B, C, T, H, W, _stream())
    return y


# ------------------------------------------------------------------------------------------------
def grad_norm(flat_grad, partial=None):
    """L2 norm of a flat fp32 arena -> fp32 [1] (device)."""
    _dev(flat_grad)
    _need(flat_grad.dtype == torch.float32 and flat_grad.is_contiguous() and flat_grad.dim() == 1, "flat fp32 arena")
    npart = 1024
    if partial is None:
        partial = torch.empty(npart, dtype=torch.float64, device=flat_grad.device)
    out = torch.empty(1, dtype=torch.float32, device=flat_grad.device)
    _lib.call("shg_sumsq", flat_grad.data_ptr(), flat_grad.numel(), partial.data_ptr(), npart, out.data_ptr(), _stream())
    return out


def gemm_kseg(a_segs, b_segs, out, accumulate=False):
    """out [M, N] (+)= sum_s a_segs[s] [M, k] @ b_segs[s] [k, N] as one launch (shg_gemm_kseg).  a_segs / b_segs: 3-D tensors [n_seg, M, k] /
    [n_seg, k, N] (any stride between the segments, rows contiguous)."""
    _dev(a_segs, b_segs, out)
    n, M, k = a_segs.shape
    N = b_segs.shape[2]
    _need(b_segs.shape[0] == n and b_segs.shape[1] == k and tuple(out.shape) == (M, N), "shapes do not match")
    _need(a_segs.stride(2) == 1 and b_segs.stride(2) == 1 and out.is_contiguous() and a_segs.dtype == b_segs.dtype == out.dtype, "row-contiguous operands of one dtype")
    _lib.call("shg_gemm_kseg", a_segs.data_ptr(), b_segs.data_ptr(), out.data_ptr(), _dt(out), M, N, k, n, a_segs.stride(1), b_segs.stride(1), N,
              a_segs.stride(0), b_segs.stride(0), 1 if accumulate else 0, _stream())
    return out


def grad_norm_ranges(flat_grad, ranges, extra=None):
    """L2 norm over the element ranges [(lo, hi), ..] of a flat fp32 arena (lo, hi multiples of 4) plus sqrt-under-the-root of
    extra[0] (float64 [1], sums other kernels accumulated; reset to 0 by the call) -> fp32 [1]."""
    _dev(flat_grad, extra)
    _need(flat_grad.dtype == torch.float32 and flat_grad.is_contiguous() and flat_grad.dim() == 1, "flat fp32 arena")
    ranges = [(int(a), int(b)) for a, b in ranges if b > a]
    _need(all(a % 4 == 0 for a, _ in ranges), "range starts must be multiples of 4 elements")
    _need(extra is None or (extra.dtype == torch.float64 and extra.numel() == 1), "extra must be float64 [1]")
    total = sum(b - a for a, b in ranges)
    partial = torch.empty(1024 + len(ranges), dtype=torch.float64, device=flat_grad.device)
    out = torch.empty(1, dtype=torch.float32, device=flat_grad.device)
    used = 0
    for a, b in ranges:
        npart = max(1, min(1024 - used - (len(ranges) - 1), int(round(1024.0 * (b - a) / max(total, 1)))))
        _lib.call("shg_sumsq_partial", flat_grad[a:b].data_ptr(), b - a, partial[used:].data_ptr(), npart, _stream())
        used += npart
    if used == 0:
        partial[:1].zero_()
        used = 1
    _lib.call("shg_sumsq_final", partial.data_ptr(), used, _p(extra), out.data_ptr(), _stream())
    return out


def bertadam_arena(param, grad, m, v, shadow, grad_norm_t, max_norm, lr, warmup, t_total, step_state, b1=0.9, b2=0.999,
                   eps=1e-6, weight_decay=0.01, bump_step=True, zero_grad=False):
    _dev(param, grad, m, v, shadow, grad_norm_t, step_state)
    n = param.numel()
    for t, nm in ((param, "param"), (grad, "grad"), (m, "m"), (v, "v")):
        _need(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, "%s must be a flat fp32 arena" % nm)
    if shadow is not None:
        _need(shadow.dtype == torch.bfloat16 and shadow.numel() == n and shadow.is_contiguous(), "shadow bf16 arena")
    _need(step_state.dtype == torch.int64 and step_state.numel() >= 1, "step_state int64")
    _lib.call("shg_bertadam_arena", param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), _p(shadow), n,
              _p(grad_norm_t), float(max_norm), float(lr), float(warmup), int(t_total), float(b1), float(b2), float(eps),
              float(weight_decay), step_state.data_ptr(), (1 if bump_step else 0) | (2 if zero_grad else 0), _stream())


def add_i64(t, delta):
    _dev(t)
    _need(t.dtype == torch.int64 or t.dtype == torch.uint64, "int64 counter")
    _lib.call("shg_add_i64", t.data_ptr(), int(delta), _stream())


def cast_f32(src, dst):
    _dev(src, dst)
    _need(src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst.numel() == src.numel(),
          "cast: contiguous fp32 source and same-size destination")
    _lib.call("shg_cast_f32", src.data_ptr(), dst.data_ptr(), _dt(dst), src.numel(), _stream())
