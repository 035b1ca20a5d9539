"""Differentiable ops of the HIP path (torch.autograd.Function wrappers over the C ABI kernels).

Activations flow through autograd; PARAMETER gradients do not: every backward accumulates its
weight / bias / LayerNorm gradients straight into the fp32 gradient arena (engine.py) with
accumulate-GEMMs and column-sum kernels and returns None for the parameter inputs.  That removes
the AccumulateGrad nodes, makes shared weights (the x-layers, mc:1247-1249) work by construction,
and lets the data-parallel reducer know exactly when a gradient slice is final.

mc = AGQA/src/lxrt/modeling_capsbert.py of the reference.
"""
import ctypes
import os

import torch

from . import _lib
from . import kernels as K
from .engine import engine

ACT_NONE, ACT_GELU, ACT_RELU = K.ACT_NONE, K.ACT_GELU, K.ACT_RELU


def _cdt():
    return engine().compute_dtype


def _drop_args(p):
    """(p, seed_state, stream_id) for a call site; p = 0 outside training."""
    E = engine()
    if p <= 0.0 or not E.training:
        return 0.0, None, 0
    return float(p), E.seed_state, E.next_stream_id()


def _acc_vec(partial, param):
    """grad(param) += column sums held in `partial` [n_partials, cols]."""
    K.colsum_finish(partial, param._shg_grad.view(-1), True)
    engine().grad_written(param)


def _rows2d(t):
    return t.reshape(-1, t.shape[-1])


def _padded_rows(dy2):
    """A [M, N] gradient whose rows can be read in 16-byte chunks (pads N to a multiple of 8 with
    zeros when it is not already backed by such a buffer)."""
    m, n = dy2.shape
    np8 = K.pad8(n)
    if dy2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and dy2.stride(0) >= np8 and dy2.data_ptr() % 16 == 0:
        return dy2
    buf = torch.zeros((m, np8), dtype=dy2.dtype, device=dy2.device)
    buf[:, :n].copy_(dy2)
    return buf[:, :n]


class _NoBranch:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def deferred_branch():
    """Context that continues the engine's deferred branch (if any) - see Engine.defer_x_layers."""
    b = engine().deferred_branch
    return b.extend() if b is not None else _NoBranch()


def join_deferred_branch(*outs):
    E = engine()
    if E.deferred_branch is not None:
        E.deferred_branch.main = torch.cuda.current_stream()
        E.deferred_branch.join(*outs)
        E.deferred_branch = None


class Branch:
    """Runs an independent sub-graph of the model on a side stream (with-block), e.g. the action decoder
    beside the relation decoder.  Autograd replays each backward node on the stream of its forward, so
    the backward of the branch overlaps as well.  `inputs`: tensors produced on the main stream that the
    branch reads; call `.publish(*outs)` on the tensors the main stream consumes after `.join()`."""

    def __init__(self, index, *inputs, wait=True):
        self.side = engine().aux_stream(index)
        # a branch asked for from INSIDE the same side stream (the cross layers' language <- vision half when the x-layers
        # already run deferred on that stream) runs inline: forking a stream from itself is an event the stream records and
        # then waits for - harmless eagerly, but a self-edge in a stream capture
        if self.side is not None and self.side == torch.cuda.current_stream():
            self.side = None
        self.main = torch.cuda.current_stream() if self.side is not None else None
        self.inputs = inputs
        self.wait = wait          # False: everything the block reads was produced on the side stream itself (or long ago)

    def __enter__(self):
        if self.side is not None:
            # (under hipGraph capture every entry forks from the capturing stream: a launch on a stream that has
            # not joined the capture would not be part of the graph)
            if self.wait or torch.cuda.is_current_stream_capturing():
                self.side.wait_stream(self.main)
            for t in self.inputs:
                if t is not None:
                    t.record_stream(self.side)
            engine().note_fork(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.side is not None:
            self.ctx.__exit__(*a)
        return False

    def extend(self, *inputs):
        """Re-enters the branch later for more work that also reads `inputs` from the main stream
        (use as `with branch.extend(x):`)."""
        if self.side is not None:
            self.main = torch.cuda.current_stream()
            self.inputs = inputs
            self.wait = True
        return self

    def join(self, *outs):
        """Main stream waits for the branch; `outs` are branch results the main stream will read."""
        if self.side is not None:
            self.main.wait_stream(self.side)
            for t in outs:
                if t is not None:
                    t.record_stream(self.main)


class _WgradStream:
    """Context that runs weight-gradient work on the engine's side stream.

    dW = dY^T X is off the critical path of backward (nothing downstream in backward reads it), while
    the input-gradient chain is a sequence of small, latency-bound launches.  Issuing the weight
    gradients on a second stream lets the GPU overlap them with that chain (as branches of the captured
    hipGraph, or as concurrent eager launches).  Ordering: the side stream first waits for everything the
    main stream has enqueued so far (dY and X are complete there); consumers of the gradient arena
    (norm / optimiser / all-reduce) wait for the side stream (Engine.join_side_streams)."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.E = engine()
        self.side = self.E.wgrad_stream()

    def __enter__(self):
        if self.side is None:
            return self
        self.side.wait_stream(torch.cuda.current_stream())
        self.E.note_fork(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.side is None:
            return False
        self.ctx.__exit__(*a)
        for t in self.tensors:                 # keep the allocator from recycling them under the side stream
            if t is not None:
                t.record_stream(self.side)
        return False


def _wgrad(dy2, x2, weight, bias):
    """Accumulates dW += dy^T x and db += colsum(dy) into the gradient arena (on the side stream).

    With Engine.wgrad_batch > 1 the work is queued and issued `wgrad_batch` layers at a time
    (flush_wgrads): one fork event per batch instead of one per layer keeps the dependency graph coarse
    (a captured hipGraph only runs coarse branches concurrently) and saves host time in eager mode."""
    E = engine()
    if weight._shg_grad is None:
        return
    if E.wgrad_batch > 1 and E.wgrad_stream() is not None:
        E.deferred_wgrads.append((torch.cuda.current_stream(), dy2, x2, weight, bias))
        if len(E.deferred_wgrads) >= E.wgrad_batch:
            flush_wgrads()
        return
    with _WgradStream(dy2, x2):
        K.gemm(dy2, x2, weight._shg_grad, None, False, False, accumulate=True)
        if bias is not None:
            K.colsum(dy2, bias._shg_grad.view(-1), True)
    E.grad_written(weight)
    if bias is not None:
        E.grad_written(bias)


def flush_wgrads():
    """Issues the queued weight gradients on the side stream, behind everything their producer streams
    have enqueued so far."""
    E = engine()
    E.flush_native_wgrads()
    q, E.deferred_wgrads = E.deferred_wgrads, []
    if not q:
        return
    side = E.wgrad_stream()
    E.note_fork(side)
    seen = []
    for st, *_ in q:
        if all(st != o for o in seen):
            seen.append(st)
            side.wait_stream(st)
    with torch.cuda.stream(side):
        for _, dy2, x2, weight, bias in q:
            K.gemm(dy2, x2, weight._shg_grad, None, False, False, accumulate=True)
            if bias is not None:
                K.colsum(dy2, bias._shg_grad.view(-1), True)
    for _, dy2, x2, weight, bias in q:
        dy2.record_stream(side)
        x2.record_stream(side)
        E.grad_written(weight)
        if bias is not None:
            E.grad_written(bias)


# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = x W^T (+ b in the GEMM epilogue).  nn.Linear of mc:373-375, :427, :466, :481."""

    @staticmethod
    def forward(ctx, x, anchor, weight, bias):
        E = engine()
        w = E.operand(weight)
        x2 = _rows2d(x)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        n = w.shape[0]
        out = torch.empty((x2.shape[0], n), dtype=x2.dtype, device=x2.device)
        K.gemm(x2, w, out, None if bias is None else bias._shg_store, True, True)
        ctx.save_for_backward(x2)
        ctx.weight, ctx.bias = weight, bias
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        E = engine()
        weight, bias = ctx.weight, ctx.bias
        w = E.operand(weight)
        dy2 = _padded_rows(_rows2d(dy))
        _wgrad(dy2, x2, weight, bias)          # first: it only needs dy, and runs beside the dgrad chain
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x2.shape, dtype=x2.dtype, device=x2.device)
            K.gemm(dy2, w, dx, None, True, False)
            dx = dx.view(ctx.xshape)
        return dx, None, None, None


def linear(x, weight, bias=None):
    """`weight` / `bias`: nn.Parameter or ParamSlice."""
    return _Linear.apply(x, _anchor(weight), weight, bias)


# ------------------------------------------------------------------------------------------------
class _BiasAct(torch.autograd.Function):
    """y = dropout(act(x + b)).  mc:472-475 (GELU), transformer.py:230 (ReLU + dropout)."""

    @staticmethod
    def forward(ctx, x, anchor, bias, act, p):
        p, seed, sid = _drop_args(p)
        x = x.contiguous()
        y = K.bias_act_fwd(x, None if bias is None else bias._shg_store, act, p, seed, sid)
        ctx.save_for_backward(x)
        ctx.bias, ctx.act, ctx.drop = bias, act, (p, seed, sid)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        p, seed, sid = ctx.drop
        bias = ctx.bias
        want = bias is not None and bias._shg_grad is not None
        dx, part = K.bias_act_bwd(x, None if bias is None else bias._shg_store, dy.contiguous(), ctx.act, p, seed, sid,
                                  want_dbias=want)
        if want:
            _acc_vec(part, bias)
        return dx, None, None, None, None


def bias_act(x, bias, act, p_drop=0.0):
    return _BiasAct.apply(x, None if bias is None else _anchor(bias), bias, act, p_drop)


def dropout(x, p_drop):
    if p_drop <= 0.0 or not engine().training:
        return x
    return _BiasAct.apply(x, None, None, ACT_NONE, p_drop)


# ------------------------------------------------------------------------------------------------
class _BiasResLN(torch.autograd.Function):
    """y = LayerNorm(dropout(act(x + b)) + residual) * gamma + beta.
    BertAttOutput/BertOutput (mc:431-435, :485-489), decoder norms (transformer.py:220-232),
    head GELU+LayerNorm (agqa_model.py:105-110), embedding LayerNorms (mc:322, :353)."""

    @staticmethod
    def forward(ctx, x, anchor, bias, residual, gamma, beta, eps, act, p):
        p, seed, sid = _drop_args(p)
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        y, z, mean, rstd = K.ln_fwd(x, None if bias is None else bias._shg_store, res, gamma._shg_store,
                                    beta._shg_store, eps, act, p, seed, sid, save_z=True)
        ctx.save_for_backward(z, mean, rstd, x if act != ACT_NONE else None)
        ctx.params = (bias, gamma, beta)
        ctx.cfg = (act, p, seed, sid, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, x = ctx.saved_tensors
        bias, gamma, beta = ctx.params
        act, p, seed, sid, has_res = ctx.cfg
        want_dbias = bias is not None and bias._shg_grad is not None
        want_dx = ctx.needs_input_grad[0]
        want_dres = has_res and ctx.needs_input_grad[3]
        dx, dres, dg, db, dbi = K.ln_bwd(dy.contiguous(), z, x, None if bias is None else bias._shg_store,
                                         gamma._shg_store, mean, rstd, act, p, seed, sid,
                                         want_dx=want_dx, want_dres=want_dres, want_dbias=want_dbias)
        if gamma._shg_grad is not None:
            _acc_vec(dg, gamma)
            _acc_vec(db, beta)
        if want_dbias:
            _acc_vec(dbi, bias)
        return (dx if want_dx else None), None, None, dres, None, None, None, None, None


def bias_res_layernorm(x, bias, residual, gamma, beta, eps, act=ACT_NONE, p_drop=0.0):
    return _BiasResLN.apply(x, _anchor(gamma), bias, residual, gamma, beta, eps, act, p_drop)


# ------------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    """softmax(scale * Q K^T + mask) V with dropout on the probabilities (mc:394-421;
    transformer.py:219-229).  q/k/v may be column slices of fused projection outputs."""

    @staticmethod
    def forward(ctx, q, k, v, heads, mask_kind, mask, scale, p):
        p, seed, sid = _drop_args(p)
        o, lse = K.attention_fwd(q, k, v, heads, mask_kind, mask, scale, p, seed, sid)
        ctx.keep = getattr(lse, "_shg_keep", None)          # dropout lane masks written by the forward kernel
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (heads, mask_kind, mask, scale, p, seed, sid)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        heads, mask_kind, mask, scale, p, seed, sid = ctx.cfg
        dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        dk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
        dv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
        K.attention_bwd(q, k, v, o, d_o.contiguous(), lse, dq, dk, dv, heads, mask_kind, mask, scale, p, seed, sid, keep_mask=ctx.keep)
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, heads, mask_kind=K.MASK_NONE, mask=None, scale=0.125, p_drop=0.0):
    return _Attention.apply(q, k, v, heads, mask_kind, mask, scale, p_drop)


# ------------------------------------------------------------------------------------------------
class _EmbedSum(torch.autograd.Function):
    """sum_i table_i[ids_i] with nn.Embedding(padding_idx=0) gradient semantics (mc:332-334): rows
    selected by id 0 receive no gradient.  `whole` tables are added in full (HGEmbeddings uses
    word_embeddings.weight directly, mc:319, so its row 0 does train)."""

    @staticmethod
    def forward(ctx, dummy, ids_list, tables, whole):
        out = None
        for ids, tab in zip(ids_list, tables):
            e = tab._shg_store[ids]                      # [B,S,H] for [B,S] ids, [S,H] for [S] ids
            out = e if out is None else out + e
        if whole is not None:
            out = out + whole._shg_store.unsqueeze(0)
        ctx.ids_list, ctx.tables, ctx.whole = ids_list, tables, whole
        return out.to(_cdt())

    @staticmethod
    def backward(ctx, dy):
        g = dy.float()                                   # [B, S, H]
        E = engine()
        for ids, tab in zip(ctx.ids_list, ctx.tables):
            if tab._shg_grad is None:
                continue
            src = g.reshape(-1, g.shape[-1]) if ids.dim() == 2 else g.sum(0)
            idx = ids.reshape(-1)
            keep = (idx != 0).unsqueeze(1)
            tab._shg_grad.index_add_(0, idx, src * keep)
            E.grad_written(tab)
        if ctx.whole is not None and ctx.whole._shg_grad is not None:
            ctx.whole._shg_grad.add_(g.sum(0))
            E.grad_written(ctx.whole)
        return None, None, None, None


def embed_sum(ids_list, tables, whole=None):
    """ids broadcastable to [B, S]; returns [B, S, H] in the compute dtype."""
    dummy = tables[0]          # a Parameter input so that the output takes part in autograd
    return _EmbedSum.apply(dummy, ids_list, tables, whole)


# ------------------------------------------------------------------------------------------------
def conv1_forward(feat, w1, b1, bufs=None):
    """First half of VisualFeatEncoder's conv stack (no gradient flows into the features):
    NCDHW fp32 -> channels-last zero-bordered -> conv(5,3,3) 2048->768 + bias, GELU written into the
    zero-bordered input buffer of the second conv, pre-activation kept for backward.
    `bufs` (x_cl, y1p, pre1): persistent buffers to write into (hipGraph-friendly); allocated when None.
    This is the dominant kernel of the step; bench.py times it with events on this stream."""
    E = engine()
    cdt = E.compute_dtype
    cin = w1.shape[1]
    # features straight from the cache (feature_cache.py) are already channels-last: [B, T, H, W, C]
    channels_last = feat.shape[-1] == cin and feat.shape[1] != cin
    if channels_last:
        B, T, H, W, C = feat.shape
    else:
        B, C, T, H, W = feat.shape
    if bufs is None:
        # the two zero-bordered buffers (170 MB + 47 MB at B = 32) are kept across steps: the kernels only ever write their
        # interiors, so the borders stay zero and a step does not pay for zero-filling them again.  A step's backward has
        # read them before the next step's forward overwrites them (same stream); torch.no_grad() passes share them too.
        key = (B, T, H, W, C, w1.shape[0], cdt, feat.device)
        cache = E.__dict__.setdefault("_conv_bufs", {})
        bufs = cache.get(key)
        # (a forward whose backward has not run yet still owns them: a second forward in between gets fresh buffers)
        busy = getattr(E, "_conv_bufs_busy", False)
        if busy:
            bufs = None
        if bufs is None or torch.cuda.is_current_stream_capturing():
            bufs = (torch.zeros((B, T, H + 2, W + 2, C), dtype=cdt, device=feat.device),
                    torch.zeros((B, T - 4, H + 2, W + 2, w1.shape[0]), dtype=cdt, device=feat.device),
                    torch.empty((B, T - 4, H, W, w1.shape[0]), dtype=cdt, device=feat.device))
            if not torch.cuda.is_current_stream_capturing() and not busy:
                cache.clear()                      # one shape at a time (a new batch size replaces the old buffers)
                cache[key] = bufs
        if torch.is_grad_enabled() and cache.get(key) is bufs:
            E._conv_bufs_busy = True
    x_cl, y1p, pre1 = bufs
    if channels_last:
        x_cl[:, :, 1:-1, 1:-1].copy_(feat)                 # the zero border stays as allocated
    else:
        K.ncdhw_to_padded_cl(feat.float().contiguous(), cdt, out=x_cl)
    K.conv_workspace(B, T, H, W, feat.device, E.conv1_row_order)
    evs = getattr(E, "kernel_events", None)
    if evs is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # (E.conv1_row_order = 1: the BACKWARD works on position-major rows - the weight gradient then skips the zero-border positions
    #  of every tap.  The forward keeps the standard row order (neighbouring rows share input lines: 3-5 % faster) and writes the
    #  pre-activation it saves for backward through the row table; y1p is a layout and does not depend on the order)
    #  E.conv_fwd_pm: the forward itself in position-major rows, where a tile leaves out the taps that read only the zero border
    #  (the stream-K launch's weighted plan, "conv_k_order" bit 5))
    if E.conv1_row_order and (E.conv_fwd_pm & 1):
        K.conv3d_k533_fwd(x_cl, E.operand(w1), b1._shg_store, ACT_GELU, pad_out=True, out=y1p, want_pre=True, pre_out=pre1, order=1)
    else:
        rows = K.conv_row_table(B, T, H, W, feat.device) if E.conv1_row_order else None
        K.conv3d_k533_fwd(x_cl, E.operand(w1), b1._shg_store, ACT_GELU, pad_out=True, out=y1p, want_pre=True, pre_out=pre1, pre_rows=rows)
    if evs is not None:
        e1.record()
        evs.append((e0, e1))
    E.wait_params_ready()              # conv1's own weight / bias were updated on this stream; everything else follows here
    return x_cl, y1p, pre1


def _padded_grad_buffer(E, B, To, H, W, C, dtype, device):
    """(buffer [B, To + 8, H + 2, W + 2, C] with a zero border, int32 row table [B * To * H * W]): where conv2's output gradient
    lives for the input-gradient convolution (dy padded by 4 in T and 1 in H / W, include/shg_vqa.h).  One per shape, reused every
    step: only the interior is ever written, and the convolution that reads it is enqueued before the next step's writer."""
    key = (B, To, H, W, C, dtype, str(device))
    cache = E.__dict__.setdefault("_d2p_bufs", {})
    hit = cache.get(key)
    if hit is None:
        cache.clear()
        buf = torch.zeros((B, To + 8, H + 2, W + 2, C), dtype=dtype, device=device)
        m = torch.arange(B * To * H * W, device=device)
        w_, h_, t_, b_ = m % W, (m // W) % H, (m // (W * H)) % To, m // (W * H * To)
        rows = (((b_ * (To + 8) + t_ + 4) * (H + 2) + h_ + 1) * (W + 2) + w_ + 1).to(torch.int32).contiguous()
        hit = cache[key] = (buf, rows)
    return hit


def _event_pair(E, attr):
    """bench.py's live kernel timing: when the engine carries a list under `attr`, an event is recorded on the current stream
    now and another one by _event_done (both around launches on THIS stream)."""
    evs = getattr(E, attr, None)
    if evs is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return evs, e0, e1


def _event_done(tm):
    if tm is not None:
        tm[2].record()
        tm[0].append((tm[1], tm[2]))


class _VisualConvTokens(torch.autograd.Function):
    """Second conv + token assembly of VisualFeatEncoder (mc:991-996, :1037-1073) and the backward of
    BOTH convolutions: conv(5,3,3) 768->768 + bias + GELU -> tokens [B, 392, C] in (t,h,w) order,
    cls token + learned positions.  Inputs x_cl / y1p / pre1 come from conv1_forward."""

    @staticmethod
    def forward(ctx, x_cl, y1p, pre1, w1, b1, w2, b2, cls_token, pe):
        E = engine()
        cdt = E.compute_dtype
        # (as for conv1: forward in standard row order, the saved pre-activation in the position-major rows its backward works on)
        ctx.order2 = E.conv1_row_order
        shp = (y1p.shape[0], y1p.shape[1], y1p.shape[2] - 2, y1p.shape[3] - 2, y1p.device)
        if ctx.order2 and (E.conv_fwd_pm & 2):         # position-major forward (tap skipping), the token rows back in sequence order
            y2, pre2 = K.conv3d_k533_fwd(y1p, E.operand(w2), b2._shg_store, ACT_GELU, pad_out=False, want_pre=True, order=1,
                                         y_rows=K.conv_row_table_inv(*shp))
        else:
            rows2 = K.conv_row_table(*shp) if ctx.order2 else None
            y2, pre2 = K.conv3d_k533_fwd(y1p, E.operand(w2), b2._shg_store, ACT_GELU, pad_out=False, want_pre=True, pre_rows=rows2)
        B = x_cl.shape[0]
        C = y2.shape[-1]
        out = K.tokens_assemble(y2.view(B, -1, C), cls_token._shg_store.view(-1), pe._shg_store)     # cls + positions, one kernel
        ctx.save_for_backward(x_cl, y1p, pre1, pre2)
        ctx.params = (w1, b1, w2, b2, cls_token, pe)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x_cl, y1p, pre1, pre2 = ctx.saved_tensors
        w1, b1, w2, b2, cls_token, pe = ctx.params
        E = engine()
        # the relation layers' last weight gradients are still queued for a grouped launch: issue them now, so that they drain
        # beside the convolutions' backward instead of behind it (everything below fills the chip and ends the step)
        E.flush_native_wgrads()
        E._conv_bufs_busy = False                  # (conv1_forward: the persistent buffers are free again after this backward)
        B, n_tok, C = d_out.shape
        # position / cls-token gradients (sums over the batch) only feed the optimiser: on the weight-gradient stream,
        # not in front of the convolutions' backward (the batch reduction alone is ~0.4 ms of the main chain otherwise)
        with _WgradStream(d_out):
            g32 = d_out.float()
            if pe._shg_grad is not None:
                pe._shg_grad[:n_tok].add_(g32.sum(0))
            if cls_token._shg_grad is not None:
                cls_token._shg_grad.view(-1).add_(g32[:, 0].sum(0))
            del g32
        if pe._shg_grad is not None:
            E.grad_written(pe)
        if cls_token._shg_grad is not None:
            E.grad_written(cls_token)
        # conv2: GELU', bias grad, weight grad, input grad.  One kernel reads the token gradients without their cls rows (no
        # contiguous copy) and writes d2 twice: dense (the weight gradient's operand) and into the zero-bordered layout the
        # input-gradient convolution gathers from (no F.pad pass) - a persistent buffer whose border stays zero.
        To, H, W = pre2.shape[1], pre2.shape[2], pre2.shape[3]
        fused = d_out.is_contiguous() and not torch.cuda.is_current_stream_capturing() and os.environ.get("SHG_CONV_BWD_FUSED", "1") != "0"
        order2 = ctx.order2
        tbl2 = K.conv_row_table(B, To + 4, H, W, d_out.device) if order2 else None
        if fused:
            # (order2: pre2 and the dense result d2 - the weight gradient's operand - in position-major rows; the token gradients
            #  and the padded copy are indexed by the sequence order as before)
            d2p, rows2 = _padded_grad_buffer(E, B, To, H, W, C, pre2.dtype, d_out.device)
            d2, part = K.bias_act_bwd(pre2, None, d_out, ACT_GELU, want_dbias=True, dy_groups=(n_tok - 1, n_tok, 1), out2=(d2p, rows2),
                                      x_rows=tbl2)
        else:
            if order2:                                     # (rare path: back to sequence order)
                pre2 = pre2.view(-1, C)[tbl2.long()].view(pre2.shape)
                order2 = 0
            d_tok = d_out[:, 1:].contiguous().view(pre2.shape)
            d2, part = K.bias_act_bwd(pre2, None, d_tok, ACT_GELU, want_dbias=True)
        _acc_vec(part, b2)
        # input gradient of conv2 = the forward gather over dy padded by (4 in T, 1 in H/W); the kernel reads
        # the weight flipped / transposed in place.  It is issued BEFORE conv2's weight gradient: both fill the chip,
        # and only the input gradient is on the critical path (-> d1 -> conv1's weight gradient, the last kernel
        # of backward), so the side stream's conv2 wgrad waits for it instead of sharing the CUs with it.
        if not fused:
            d2p = torch.nn.functional.pad(d2, (0, 0, 1, 1, 1, 1, 4, 4))
        order1 = E.conv1_row_order
        shp1 = (B, x_cl.shape[1], x_cl.shape[2] - 2, x_cl.shape[3] - 2, x_cl.device)
        rows1 = K.conv_row_table(*shp1) if order1 else None
        if E.conv_dgrad_tm and d2p.dtype == torch.bfloat16:
            # frame-major GEMM rows (a tile keeps only the temporal taps that read data frames of the padded gradient); its rows go
            # straight to where conv1's backward wants them: (frame-major row -> standard row) then (standard -> position-major)
            key = ("dgrad_rows",) + shp1[:4] + (order1,)
            cache = E.__dict__.setdefault("_row_tables", {})
            comp = cache.get(key)
            if comp is None:
                inv_t = K.conv_row_table_inv(*shp1, order=2)
                comp = (rows1[inv_t.long()] if rows1 is not None else inv_t).contiguous()
                cache[key] = comp
            d_y1 = K.conv3d_k533_dgrad(d2p, E.operand(w2), out_rows=comp, order=2)
        else:
            d_y1 = K.conv3d_k533_dgrad(d2p, E.operand(w2), out_rows=rows1)       # in the row order of pre1 / conv1's weight gradient
        # The conv weight gradients are the last kernels of backward and fill the chip.  They stay on THIS stream:
        # behind the weight-gradient stream's backlog of small split-K GEMMs (it runs ~2 ms late at this point)
        # they would start only when that has drained; here the backlog drains beside them instead.
        inline = os.environ.get("SHG_CONV_WGRAD_INLINE", "1") != "0"
        d1, part1 = K.bias_act_bwd(pre1, None, d_y1, ACT_GELU, want_dbias=True)
        _acc_vec(part1, b1)
        if inline and E.grad_ready_hook is not None and os.environ.get("SHG_CONV_WGRAD_SPLIT", "1") != "0":
            # data parallel: conv1's gradient is a quarter of the step's all-reduce bytes and the LAST kernel of backward.
            # conv2's weight gradient goes first (its exchange runs under conv1's), and conv1's leaves as two launches over
            # output channels - 2/3 (720 tiles of 256 x 256 = 3 rounds on 256 CUs) and 1/3 (360 tiles = 2 rounds: the same
            # five rounds as one launch of 1 080) - so that the exchange of the first part runs under the second.
            K.conv3d_k533_wgrad(y1p, d2, w2._shg_grad, accumulate=True, order=order2)
            E.grad_written(w2)
            cout, per = w1._shg_grad.shape[0], w1._shg_grad[0].numel()
            cut = (cout // 256) * 2 // 3 * 256
            parts = [(0, cut), (cut, cout - cut)] if 0 < cut < cout else [(0, cout)]
            tm = _event_pair(E, "kernel_events_wgrad")
            for c0, cn in parts:
                K.conv3d_k533_wgrad(x_cl, d1, w1._shg_grad, accumulate=True, c0=c0, cn=cn, order=order1)
                E.grad_written(w1, c0 * per, cn * per)
            _event_done(tm)
        elif inline:
            tm = _event_pair(E, "kernel_events_wgrad")
            _conv_wgrad(E, x_cl, d1, w1, order1)
            _event_done(tm)
            E.grad_written(w1)
            _conv_wgrad(E, y1p, d2, w2, order2)
            E.grad_written(w2)
        else:
            with _WgradStream(y1p, d2):
                _conv_wgrad(E, y1p, d2, w2, order2)
            E.grad_written(w2)
            with _WgradStream(x_cl, d1):
                _conv_wgrad(E, x_cl, d1, w1, order1)
            E.grad_written(w1)
        return None, None, None, None, None, None, None, None, None


def _conv_wgrad(E, x, d, w, order=0):
    """The convolution's weight gradient: its only writer in a step SETS it and adds its share of the gradient norm as it goes
    (Engine.claim_overwrite); otherwise the usual accumulation.  order: row order of d (and of x's gather table)."""
    if E.claim_overwrite(w):
        K.conv3d_k533_wgrad_sumsq(x, d, w._shg_grad, E.norm_scalar(), order=order)
    else:
        K.conv3d_k533_wgrad(x, d, w._shg_grad, accumulate=True, order=order)


def visual_conv_tokens(feat, w1, b1, w2, b2, cls_token, pe):
    """feat: (B,2048,16,7,7) features, or None when the trainer already ran conv1_forward for this step
    (engine().conv1_cache) - that lets the rest of the step live in a captured hipGraph while the
    dominant kernel is launched and timed eagerly."""
    E = engine()
    cache = getattr(E, "conv1_cache", None)
    if cache is not None:
        x_cl, y1p, pre1 = cache
    else:
        x_cl, y1p, pre1 = conv1_forward(feat, w1, b1)
    return _VisualConvTokens.apply(x_cl, y1p, pre1, w1, b1, w2, b2, cls_token, pe)


# ------------------------------------------------------------------------------------------------
class _AddParamRows(torch.autograd.Function):
    """x [B, S, H] + rows [S, H] built from small parameters (type tokens / cls tokens of
    CrossEncoder, mc:1168-1181).  `builder` maps the parameters to the [S, H] fp32 addend; its
    gradient goes back through `scatter` (a function dsum [S,H] -> None that accumulates)."""

    @staticmethod
    def forward(ctx, x, anchor, addend, scatter):
        ctx.scatter = scatter
        return (x.float() + addend.unsqueeze(0)).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        ctx.scatter(dy.float().sum(0))
        return dy, None, None, None


def add_param_rows(x, anchor, addend, scatter):
    return _AddParamRows.apply(x, anchor, addend, scatter)


class _PrependParamRow(torch.autograd.Function):
    """cat([row.expand(B,1,H), x], dim=1) for a [1,1,H] parameter row (cls tokens, mc:1183-1184)."""

    @staticmethod
    def forward(ctx, x, row_param):
        B, S, H = x.shape
        out = torch.empty((B, S + 1, H), dtype=x.dtype, device=x.device)
        out[:, 0] = row_param._shg_store.view(1, H).to(x.dtype)
        out[:, 1:] = x
        ctx.row_param = row_param
        return out

    @staticmethod
    def backward(ctx, dy):
        rp = ctx.row_param
        if rp._shg_grad is not None:
            rp._shg_grad.view(-1).add_(dy[:, 0].float().sum(0))
            engine().grad_written(rp)
        return dy[:, 1:], None


def prepend_param_row(x, row_param):
    return _PrependParamRow.apply(x, row_param)


# ------------------------------------------------------------------------------------------------
class _SetLoss(torch.autograd.Function):
    """Hungarian matching + class-weighted CE over all query slots, fused on the device
    (matcher.py:62-80 + agqaHGQA.py:203-229).  Returns (loss_sums [4], grid, query_idx, target_idx);
    loss = sums[0] / sums[1] is formed by the caller so that data-parallel runs can all-reduce the
    numerator and denominator first."""

    @staticmethod
    def forward(ctx, logits, tgt, tgt_len, class_weight, per_frame):
        B, Q, C = logits.shape
        n_frames = B * (Q // per_frame)
        lg = logits.contiguous().view(n_frames, per_frame, C)
        oq, ot, grid = K.hungarian_per_frame(lg, tgt.view(n_frames, per_frame), tgt_len.view(n_frames))
        rows = lg.view(-1, C)
        stats, sums = K.weighted_ce_fwd(rows, grid.view(-1), class_weight)
        ctx.save_for_backward(rows, grid, stats, sums, class_weight)
        ctx.shape = (B, Q, C)
        ctx.mark_non_differentiable(grid, oq, ot)
        return sums, grid, oq, ot

    @staticmethod
    def backward(ctx, d_sums, *_):
        rows, grid, stats, sums, cw = ctx.saved_tensors
        B, Q, C = ctx.shape
        # upstream gradient arrives on sums[0] (the numerator); the caller divides by sums[1]:
        # d loss / d numerator = 1 / sums[1] is already folded into the kernel, so pass d_sums[0]*sums[1]
        gscale = (d_sums[0] * sums[1]).reshape(1).float().contiguous()
        d = K.weighted_ce_bwd(rows, grid.view(-1), cw, stats, sums, gscale, padded=True)
        return d.view(B, Q, -1)[:, :, :C], None, None, None, None


def set_loss(logits, tgt, tgt_len, class_weight, per_frame):
    return _SetLoss.apply(logits, tgt, tgt_len, class_weight, per_frame)


class _BCELoss(torch.autograd.Function):
    """BCEWithLogitsLoss(mean) * n_classes (agqaHGQA.py:344-345)."""

    @staticmethod
    def forward(ctx, logits, target):
        lg = logits.contiguous()
        one = torch.ones(1, dtype=torch.float32, device=lg.device)
        loss, d = K.bce_logits(lg, target.contiguous(), one, want_grad=True, padded=True)
        ctx.save_for_backward(d)
        ctx.c = lg.shape[1]
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        out = d if g is None else (d.float() * g.float()).to(d.dtype)
        return out[:, :ctx.c], None


def bce_with_logits_times_c(logits, target):
    return _BCELoss.apply(logits, target)


class _CombineLosses(torch.autograd.Function):
    """total = bce * scale + rel_sums[0] / rel_sums[1] + act_sums[0] / act_sums[1] (agqaHGQA.py:344-378) and the reported
    scalars, one kernel forward and one backward instead of ~40 single-thread torch kernels between the forward pass and the
    first kernel of backward."""

    @staticmethod
    def forward(ctx, rel_sums, act_sums, bce, scale):
        rs, as_, b = rel_sums.contiguous(), act_sums.contiguous(), bce.reshape(1).contiguous()
        total, diag = K.loss_combine_fwd(rs, as_, b, scale)
        ctx.save_for_backward(rs, as_)
        ctx.scale = float(scale)
        ctx.bce_shape = bce.shape
        ctx.mark_non_differentiable(diag)
        return total.view(()), diag

    @staticmethod
    def backward(ctx, g, _):
        rs, as_ = ctx.saved_tensors
        d_rel, d_act, d_bce = K.loss_combine_bwd(None if g is None else g.reshape(1).float().contiguous(), rs, as_, ctx.scale)
        return d_rel, d_act, d_bce.view(ctx.bce_shape), None


def combine_losses(rel_sums, act_sums, bce, scale=1.0):
    """-> (total [], diag [5] = bce, rel CE, act CE, rel class error %, act class error %)."""
    return _CombineLosses.apply(rel_sums, act_sums, bce, scale)


# ------------------------------------------------------------------------------------------------
class ParamSlice:
    """Rows r0:r1 of a parameter (e.g. the q/k/v blocks of nn.MultiheadAttention.in_proj_weight,
    transformer.py:192-193) presented to the ops like a parameter of its own."""

    def __init__(self, base, r0, r1):
        self.base, self.r0, self.r1 = base, r0, r1
        self.shape = (r1 - r0,) + tuple(base.shape[1:])

    @property
    def _row(self):
        n = 1
        for s in self.base.shape[1:]:
            n *= s
        return n

    @property
    def _shg_store(self):
        return self.base._shg_store[self.r0:self.r1]

    @property
    def _shg_shadow(self):
        return self.base._shg_shadow[self.r0:self.r1]

    @property
    def _shg_grad(self):
        g = self.base._shg_grad
        return None if g is None else g[self.r0:self.r1]

    @property
    def _shg_off(self):
        return self.base._shg_off + self.r0 * self._row

    @property
    def _shg_numel(self):
        return (self.r1 - self.r0) * self._row


class ParamConcat:
    """Several parameters that Engine.adopt laid out back-to-back (groups=...), seen as one operand:
    e.g. BertAttention's query / key / value weights as a single [3*768, 768] matrix (mc:373-375)."""

    def __init__(self, params):
        self.params = list(params)
        self.base = self.params[0]
        off = self.base._shg_off
        for p in self.params:
            if p._shg_off != off:
                raise RuntimeError("ParamConcat: parameters are not contiguous in the arena (Engine.adopt groups=)")
            off += p._shg_numel
        rows = sum(p.shape[0] for p in self.params)
        self.shape = (rows,) + tuple(self.base.shape[1:])
        self._shg_off = self.base._shg_off
        self._shg_numel = off - self.base._shg_off

    def _view(self, arena):
        return arena[self._shg_off:self._shg_off + self._shg_numel].view(self.shape)

    @property
    def _shg_store(self):
        return self._view(engine().param_arena)

    @property
    def _shg_shadow(self):
        return self._view(engine().shadow_arena)

    @property
    def _shg_grad(self):
        return None if self.base._shg_grad is None else self._view(engine().grad_arena)


def _anchor(p):
    return p.base if isinstance(p, (ParamSlice, ParamConcat)) else p


class _SelfAttnQKV(torch.autograd.Function):
    """Fused-projection self-attention: ONE GEMM (N = 3*768) for Q/K/V, attention on strided views of its
    output, and in backward the attention kernels write dQ/dK/dV straight into one [.., 3*768] buffer
    that feeds one dgrad and one wgrad GEMM (mc:384-421 with mc:373-375 merged)."""

    @staticmethod
    def forward(ctx, x, anchor, w_qkv, b_qkv, heads, mask_kind, mask, scale, p):
        E = engine()
        p, seed, sid = _drop_args(p)
        B, S, H = x.shape
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        qkv = torch.empty((B * S, 3 * H), dtype=x2.dtype, device=x2.device)
        K.gemm(x2, E.operand(w_qkv), qkv, b_qkv._shg_store.view(-1), True, True)
        q3 = qkv.view(B, S, 3 * H)
        o, lse = K.attention_fwd(q3[:, :, :H], q3[:, :, H:2 * H], q3[:, :, 2 * H:], heads, mask_kind, mask, scale, p, seed, sid)
        ctx.keep = getattr(lse, "_shg_keep", None)          # dropout lane masks written by the forward kernel
        ctx.save_for_backward(x2, qkv, o, lse)
        ctx.cfg = (w_qkv, b_qkv, heads, mask_kind, mask, scale, p, seed, sid, (B, S, H))
        return o

    @staticmethod
    def backward(ctx, d_o):
        x2, qkv, o, lse = ctx.saved_tensors
        w_qkv, b_qkv, heads, mask_kind, mask, scale, p, seed, sid, (B, S, H) = ctx.cfg
        E = engine()
        q3 = qkv.view(B, S, 3 * H)
        dqkv = torch.empty_like(qkv)
        d3 = dqkv.view(B, S, 3 * H)
        K.attention_bwd(q3[:, :, :H], q3[:, :, H:2 * H], q3[:, :, 2 * H:], o, d_o.contiguous(), lse,
                        d3[:, :, :H], d3[:, :, H:2 * H], d3[:, :, 2 * H:], heads, mask_kind, mask, scale, p, seed, sid, keep_mask=ctx.keep)
        _wgrad(dqkv, x2, w_qkv, b_qkv)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            K.gemm(dqkv, E.operand(w_qkv), dx, None, True, False)
            dx = dx.view(B, S, H)
        return dx, None, None, None, None, None, None, None, None


class _CrossAttnQKV(torch.autograd.Function):
    """Cross-attention with Q from `h` (one GEMM) and K/V from `c` (one GEMM, N = 2*768)."""

    @staticmethod
    def forward(ctx, h, c, anchor, w_q, b_q, w_kv, b_kv, heads, mask_kind, mask, scale, p):
        E = engine()
        p, seed, sid = _drop_args(p)
        B, Sq, H = h.shape
        Sk = c.shape[1]
        h2 = h.reshape(-1, H)
        c2 = c.reshape(-1, H)
        h2 = h2 if h2.is_contiguous() else h2.contiguous()
        c2 = c2 if c2.is_contiguous() else c2.contiguous()
        q = torch.empty((B * Sq, H), dtype=h2.dtype, device=h2.device)
        kv = torch.empty((B * Sk, 2 * H), dtype=h2.dtype, device=h2.device)
        K.gemm(h2, E.operand(w_q), q, b_q._shg_store.view(-1), True, True)
        K.gemm(c2, E.operand(w_kv), kv, b_kv._shg_store.view(-1), True, True)
        kv3 = kv.view(B, Sk, 2 * H)
        o, lse = K.attention_fwd(q.view(B, Sq, H), kv3[:, :, :H], kv3[:, :, H:], heads, mask_kind, mask, scale, p, seed, sid)
        ctx.keep = getattr(lse, "_shg_keep", None)          # dropout lane masks written by the forward kernel
        ctx.save_for_backward(h2, c2, q, kv, o, lse)
        ctx.cfg = (w_q, b_q, w_kv, b_kv, heads, mask_kind, mask, scale, p, seed, sid, (B, Sq, Sk, H))
        return o

    @staticmethod
    def backward(ctx, d_o):
        h2, c2, q, kv, o, lse = ctx.saved_tensors
        w_q, b_q, w_kv, b_kv, heads, mask_kind, mask, scale, p, seed, sid, (B, Sq, Sk, H) = ctx.cfg
        E = engine()
        kv3 = kv.view(B, Sk, 2 * H)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dkv3 = dkv.view(B, Sk, 2 * H)
        K.attention_bwd(q.view(B, Sq, H), kv3[:, :, :H], kv3[:, :, H:], o, d_o.contiguous(), lse, dq.view(B, Sq, H),
                        dkv3[:, :, :H], dkv3[:, :, H:], heads, mask_kind, mask, scale, p, seed, sid, keep_mask=ctx.keep)
        _wgrad(dq, h2, w_q, b_q)
        _wgrad(dkv, c2, w_kv, b_kv)
        dh = dc = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h2)
            K.gemm(dq, E.operand(w_q), dh, None, True, False)
            dh = dh.view(B, Sq, H)
        if ctx.needs_input_grad[1]:
            dc = torch.empty_like(c2)
            K.gemm(dkv, E.operand(w_kv), dc, None, True, False)
            dc = dc.view(B, Sk, H)
        return dh, dc, None, None, None, None, None, None, None, None, None, None


def self_attention_qkv(x, w_qkv, b_qkv, heads, mask_kind, mask, scale, p_drop):
    return _SelfAttnQKV.apply(x, _anchor(w_qkv), w_qkv, b_qkv, heads, mask_kind, mask, scale, p_drop)


def cross_attention_qkv(h, c, w_q, b_q, w_kv, b_kv, heads, mask_kind, mask, scale, p_drop):
    return _CrossAttnQKV.apply(h, c, _anchor(w_q), w_q, b_q, w_kv, b_kv, heads, mask_kind, mask, scale, p_drop)


# ------------------------------------------------------------------------------------------------
# Whole sub-layers as ONE autograd node AND one C-ABI call each (csrc/executor.hip): shg_attn_sublayer_*,
# shg_ffn_sublayer_*, shg_decoder_* enqueue the 6-16 kernels of a sub-layer (80 / 200 for a five-layer decoder) from C++,
# so Python pays for ~100 calls per step instead of ~1 150 launches.  What the sequences do: (i) the residual gradient and
# the projection's input gradient are summed by the dgrad GEMM's accumulate epilogue, (ii) a LayerNorm's gamma / beta /
# bias column sums finish in one launch, (iii) Linear + GELU / ReLU (+ dropout) is one kernel forward and backward,
# (iv) weight gradients go to the side stream behind an event (dW only feeds the optimiser).
# ------------------------------------------------------------------------------------------------
_ADDR = ctypes.addressof


def _c2(t):
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.is_contiguous() else t2.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _fill_linear(L, w, b):
    E = engine()
    L.w = _ptr(E.operand(w)) if w is not None else None
    L.bias = _ptr(b._shg_store) if b is not None else None
    L.gw = _ptr(w._shg_grad) if w is not None else None
    L.gb = _ptr(b._shg_grad) if b is not None else None


def _fill_norm(N, gamma, beta, eps):
    N.gamma, N.beta = gamma._shg_store.data_ptr(), beta._shg_store.data_ptr()
    N.g_gamma, N.g_beta = _ptr(gamma._shg_grad), _ptr(beta._shg_grad)
    N.eps = eps


def _trainable(*ps):
    return [p for p in ps if p is not None and p._shg_grad is not None]


def _bytes(n, dev):
    return torch.empty(n, dtype=torch.uint8, device=dev)


class FFNParams:
    """Parameter handles + constants of a position-wise feed-forward sub-layer (shg_ffn_sublayer_t)."""

    def __init__(self, w1, b1, w2, b2, gamma, beta, eps, act, p_inner, p_out):
        self.w1, self.b1, self.w2, self.b2, self.gamma, self.beta = w1, b1, w2, b2, gamma, beta
        self.eps, self.act, self.p_inner, self.p_out = eps, act, p_inner, p_out
        self._c = None

    def fill(self, L):
        L.act, L.p_inner, L.p_out = self.act, self.p_inner, self.p_out
        _fill_linear(L.l1, self.w1, self.b1)
        _fill_linear(L.l2, self.w2, self.b2)
        _fill_norm(L.ln, self.gamma, self.beta, self.eps)

    def cstruct(self):
        if self._c is None:
            self._c = _lib.FfnSublayerT()
            self.fill(self._c)
            self.trainable = _trainable(self.w1, self.b1, self.w2, self.b2, self.gamma, self.beta)
        return self._c


_MODES = {"self": 0, "cross": 1, "dec_self": 2, "dec_cross": 3}


class AttnParams:
    """Parameter handles + constants of an attention sub-layer (shg_attn_sublayer_t).
    mode: 'self'      q, k, v = W_in x                      (one GEMM, N = 3H; BertSelfattLayer mc:450-460)
          'cross'     q = W_q x ; k, v = W_kv mem           (BertCrossattLayer mc:438-447)
          'dec_self'  q, k = W_qk (x + pos) ; v = W_v x     (transformer.py:216-219)
          'dec_cross' q = W_q (x + pos) ; k, v = W_kv mem   (transformer.py:222-226)
    w_a / b_a: rows of the in-projection applied to the first source, w_b / b_b: rows applied to the second."""

    def __init__(self, mode, w_a, b_a, w_b, b_b, w_o, b_o, gamma, beta, eps, heads, scale, p_attn, p_out):
        self.mode, self.w_a, self.b_a, self.w_b, self.b_b, self.w_o, self.b_o = mode, w_a, b_a, w_b, b_b, w_o, b_o
        self.gamma, self.beta, self.eps, self.heads, self.scale, self.p_attn, self.p_out = gamma, beta, eps, heads, scale, p_attn, p_out
        self._c = None

    def fill(self, L):
        L.mode, L.heads, L.mask_kind, L.mask = _MODES[self.mode], self.heads, K.MASK_NONE, None
        L.scale, L.p_attn, L.p_out = self.scale, self.p_attn, self.p_out
        _fill_linear(L.a, self.w_a, self.b_a)
        _fill_linear(L.b, self.w_b, self.b_b)
        _fill_linear(L.o, self.w_o, self.b_o)
        _fill_norm(L.ln, self.gamma, self.beta, self.eps)

    def cstruct(self):
        if self._c is None:
            self._c = _lib.AttnSublayerT()
            self.fill(self._c)
            self.trainable = _trainable(self.w_a, self.b_a, self.w_b, self.b_b, self.w_o, self.b_o, self.gamma, self.beta)
        return self._c


class DecoderParams:
    """The layer table of a TransformerDecoder (shg_decoder_layer_t[n]): per layer (AttnParams dec_self, AttnParams
    dec_cross, FFNParams)."""

    def __init__(self, layers):
        self.layers = layers
        self._c = None

    def cstruct(self):
        if self._c is None:
            arr = (_lib.DecoderLayerT * len(self.layers))()
            tr = []
            for L, (ps, pc, pf) in zip(arr, self.layers):
                ps.fill(L.self_attn)
                pc.fill(L.cross_attn)
                pf.fill(L.ffn)
                tr += _trainable(ps.w_a, ps.b_a, ps.w_b, ps.b_b, ps.w_o, ps.b_o, ps.gamma, ps.beta,
                                 pc.w_a, pc.b_a, pc.w_b, pc.b_b, pc.w_o, pc.b_o, pc.gamma, pc.beta,
                                 pf.w1, pf.b1, pf.w2, pf.b2, pf.gamma, pf.beta)
            self.trainable = tr
            self.heads = self.layers[0][0].heads
            self.ffn_dim = self.layers[0][2].w1.shape[0]
            self._c = arr
        return self._c


class _FFNSublayer(torch.autograd.Function):
    """y = LayerNorm(x + drop_out(W2 drop_in(act(W1 x + b1)) + b2)):
    BertIntermediate + BertOutput (mc:463-489; GELU, no inner dropout) and the decoder's
    linear1 / ReLU / dropout / linear2 / dropout3 / norm3 (transformer.py:230-232)."""

    @staticmethod
    def forward(ctx, x, anchor, P):
        E = engine()
        L = P.cstruct()
        x2 = _c2(x)
        rows, H = x2.shape
        F = P.w1.shape[0]
        dtc = K._dt(x2)
        sid = E.take_stream_ids(2)
        y = torch.empty_like(x2)
        saved = _bytes(_lib.lib().shg_ffn_sublayer_saved_bytes(dtc, rows, H, F), x2.device)
        _lib.call("shg_ffn_sublayer_fwd", _ADDR(L), E.run_addr(dtc), rows, H, F, x2.data_ptr(), y.data_ptr(), None, None,
                  saved.data_ptr(), sid)
        ctx.save_for_backward(x2, saved)
        ctx.P, ctx.cfg = P, (sid, F, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, saved = ctx.saved_tensors
        P = ctx.P
        sid, F, shape = ctx.cfg
        E = engine()
        rows, H = x2.shape
        dtc = K._dt(x2)
        dy2 = _c2(dy)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        scratch = _bytes(_lib.lib().shg_ffn_sublayer_scratch_bytes(dtc, rows, H, F), x2.device)
        _lib.call("shg_ffn_sublayer_bwd", _ADDR(P.cstruct()), E.run_addr(dtc), rows, H, F, x2.data_ptr(), saved.data_ptr(),
                  dy2.data_ptr(), _ptr(dx), scratch.data_ptr(), sid)
        E.after_backward_call((x2, saved, scratch), P.trainable)
        return (dx.view(shape) if dx is not None else None), None, None


def ffn_sublayer(x, P):
    return _FFNSublayer.apply(x, _anchor(P.gamma), P)


class _AttnSublayer(torch.autograd.Function):
    """y = LayerNorm(x + drop_out(W_o attention(...) + b_o)) for the 'self' and 'cross' layouts of AttnParams
    (the decoder layouts run inside shg_decoder_*, see decoder_stack)."""

    @staticmethod
    def forward(ctx, x, mem, anchor, P, mask_kind, mask):
        E = engine()
        L = P.cstruct()
        if P.mode not in ("self", "cross"):
            raise ValueError("attn_sublayer: decoder layouts go through decoder_stack")
        B, Sq, H = x.shape
        x2 = _c2(x)
        mem2 = None
        Sk = Sq
        if P.mode == "cross":
            Sk = mem.shape[1]
            mem2 = _c2(mem)
        dtc = K._dt(x2)
        mask = K._mask_args(mask_kind, mask, B, Sq, Sk)
        L.mask_kind, L.mask = mask_kind, _ptr(mask)
        sid = E.take_stream_ids(2)
        y = torch.empty_like(x2)
        saved = _bytes(_lib.lib().shg_attn_sublayer_saved_bytes(L.mode, dtc, B, Sq, Sk, P.heads), x2.device)
        _lib.call("shg_attn_sublayer_fwd", _ADDR(L), E.run_addr(dtc), B, Sq, Sk, x2.data_ptr(), None, _ptr(mem2), y.data_ptr(),
                  None, None, saved.data_ptr(), sid)
        ctx.save_for_backward(x2, mem2, saved)
        ctx.P, ctx.cfg = P, (sid, mask_kind, mask, (B, Sq, Sk, H))
        return y.view(B, Sq, H)

    @staticmethod
    def backward(ctx, dy):
        x2, mem2, saved = ctx.saved_tensors
        P = ctx.P
        sid, mask_kind, mask, (B, Sq, Sk, H) = ctx.cfg
        E = engine()
        L = P.cstruct()
        L.mask_kind, L.mask = mask_kind, _ptr(mask)
        dtc = K._dt(x2)
        dy2 = _c2(dy)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dmem = torch.empty_like(mem2) if (mem2 is not None and ctx.needs_input_grad[1]) else None
        scratch = _bytes(_lib.lib().shg_attn_sublayer_scratch_bytes(L.mode, dtc, B, Sq, Sk, P.heads), x2.device)
        _lib.call("shg_attn_sublayer_bwd", _ADDR(L), E.run_addr(dtc), B, Sq, Sk, x2.data_ptr(), None, _ptr(mem2), saved.data_ptr(),
                  dy2.data_ptr(), _ptr(dx), None, _ptr(dmem), 0, scratch.data_ptr(), sid)
        E.after_backward_call((x2, mem2, saved, scratch), P.trainable)
        return (dx.view(B, Sq, H) if dx is not None else None), (dmem.view(B, Sk, H) if dmem is not None else None), None, None, None, None


def attn_sublayer(x, mem, P, mask_kind=K.MASK_NONE, mask=None):
    return _AttnSublayer.apply(x, mem, _anchor(P.gamma), P, mask_kind, mask)


class _DecoderStack(torch.autograd.Function):
    """All layers of a TransformerDecoder (transformer.py:86-124 over forward_post layers :212-233) as ONE node:
    out = decoder(tgt, memory, query_pos, tgt_mask); tgt None = the zeros of agqa_model.py:234."""

    @staticmethod
    def forward(ctx, memory, qpos, tgt, anchor, D, tgt_mask):
        E = engine()
        arr = D.cstruct()
        n = len(arr)
        B, Q, H = qpos.shape
        S = memory.shape[1]
        mem2, qp2 = _c2(memory), _c2(qpos)
        tgt2 = _c2(tgt) if tgt is not None else None
        dtc = K._dt(mem2)
        kind = K.MASK_FULL if tgt_mask is not None else K.MASK_NONE
        tgt_mask = K._mask_args(kind, tgt_mask, B, Q, Q)
        for L in arr:
            L.self_attn.mask_kind, L.self_attn.mask = kind, _ptr(tgt_mask)
        sid = E.take_stream_ids(6 * n)
        out = torch.empty_like(qp2)
        saved = _bytes(_lib.lib().shg_decoder_saved_bytes(n, dtc, B, Q, S, D.heads, D.ffn_dim), mem2.device)
        _lib.call("shg_decoder_fwd", _ADDR(arr), n, E.run_addr(dtc), B, Q, S, D.ffn_dim, _ptr(tgt2), mem2.data_ptr(), qp2.data_ptr(),
                  out.data_ptr(), saved.data_ptr(), sid)
        ctx.save_for_backward(mem2, qp2, tgt2, saved)
        ctx.D, ctx.cfg = D, (sid, kind, tgt_mask, (B, Q, S, H))
        return out.view(B, Q, H)

    @staticmethod
    def backward(ctx, d_out):
        mem2, qp2, tgt2, saved = ctx.saved_tensors
        D = ctx.D
        sid, kind, tgt_mask, (B, Q, S, H) = ctx.cfg
        E = engine()
        arr = D.cstruct()
        n = len(arr)
        for L in arr:
            L.self_attn.mask_kind, L.self_attn.mask = kind, _ptr(tgt_mask)
        dtc = K._dt(mem2)
        d2 = _c2(d_out)
        dmem = torch.empty_like(mem2) if ctx.needs_input_grad[0] else None
        dpos = torch.empty_like(qp2) if ctx.needs_input_grad[1] else None
        dtgt = torch.empty_like(tgt2) if (tgt2 is not None and ctx.needs_input_grad[2]) else None
        scratch = _bytes(_lib.lib().shg_decoder_scratch_bytes(n, dtc, B, Q, S, D.heads, D.ffn_dim), mem2.device)
        _lib.call("shg_decoder_bwd", _ADDR(arr), n, E.run_addr(dtc), B, Q, S, D.ffn_dim, _ptr(tgt2), mem2.data_ptr(), qp2.data_ptr(),
                  saved.data_ptr(), d2.data_ptr(), _ptr(dtgt), _ptr(dpos), _ptr(dmem), scratch.data_ptr(), sid)
        E.after_backward_call((mem2, qp2, tgt2, saved, scratch), D.trainable)
        return ((dmem.view(B, S, H) if dmem is not None else None), (dpos.view(B, Q, H) if dpos is not None else None),
                (dtgt.view(B, Q, H) if dtgt is not None else None), None, None, None)


def decoder_stack(memory, query_pos, D, tgt_mask=None, tgt=None):
    """memory [B,S,H], query_pos [B,Q,H], tgt [B,Q,H] or None (zeros); D: DecoderParams."""
    return _DecoderStack.apply(memory, query_pos, tgt, _anchor(D.layers[0][2].gamma), D, tgt_mask)
