"""DETR-style transformer decoder of the situation hyper-graph set decoder
(AGQA/src/lxrt/transformer.py:86-233), on the HIP ops.  Parameter names follow
torch.nn.MultiheadAttention / the reference so that checkpoints load unchanged:
{self_attn,multihead_attn}.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias},
linear{1,2}.*, norm{1,2,3}.*.
"""
import copy

import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .ops import ParamSlice


class MultiheadAttention(nn.Module):
    """Parameter layout of torch.nn.MultiheadAttention (packed in-projection)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        if embed_dim % num_heads or embed_dim // num_heads != 64:
            raise ValueError("the HIP attention kernel is specialised for head size 64")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def rows(self, r0, r1):
        return ParamSlice(self.in_proj_weight, r0, r1), ParamSlice(self.in_proj_bias, r0, r1)

    def attend(self, q_in, k_in, v_in, full_mask, same_qk):
        """Batch-first [B,S,E] inputs; returns the heads' output BEFORE out_proj (the caller fuses
        out_proj's bias, the dropout, the residual and the LayerNorm into one epilogue)."""
        E = self.embed_dim
        if same_qk:                       # self-attention: q and k share their input -> one GEMM, N = 2E
            w, b = self.rows(0, 2 * E)
            qk = ops.linear(q_in, w, b)
            q, k = qk[..., :E], qk[..., E:]
            wv, bv = self.rows(2 * E, 3 * E)
            v = ops.linear(v_in, wv, bv)
        else:                             # cross-attention: k and v share the memory -> one GEMM, N = 2E
            wq, bq = self.rows(0, E)
            q = ops.linear(q_in, wq, bq)
            wkv, bkv = self.rows(E, 3 * E)
            kv = ops.linear(k_in, wkv, bkv)
            k, v = kv[..., :E], kv[..., E:]
        kind = K.MASK_FULL if full_mask is not None else K.MASK_NONE
        return ops.attention(q, k, v, self.num_heads, kind, full_mask, 0.125, self.dropout)


class TransformerDecoderLayer(nn.Module):
    """transformer.py:187-270 (post-norm branch; activation ReLU)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before or activation != "relu":
            raise NotImplementedError("the reference model uses the post-norm ReLU layer (agqa_model.py:98)")
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def _params(self):
        if getattr(self, "_sp", None) is None:
            E, p = self.self_attn.embed_dim, self.dropout.p
            sa, ca = self.self_attn, self.multihead_attn
            w_qk, b_qk = sa.rows(0, 2 * E)
            w_v, b_v = sa.rows(2 * E, 3 * E)
            w_q, b_q = ca.rows(0, E)
            w_kv, b_kv = ca.rows(E, 3 * E)
            self._sp = (
                ops.AttnParams("dec_self", w_qk, b_qk, w_v, b_v, sa.out_proj.weight, sa.out_proj.bias, self.norm1.weight,
                               self.norm1.bias, self.norm1.eps, sa.num_heads, 0.125, sa.dropout, self.dropout1.p),
                ops.AttnParams("dec_cross", w_q, b_q, w_kv, b_kv, ca.out_proj.weight, ca.out_proj.bias, self.norm2.weight,
                               self.norm2.bias, self.norm2.eps, ca.num_heads, 0.125, ca.dropout, self.dropout2.p),
                ops.FFNParams(self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                              self.norm3.weight, self.norm3.bias, self.norm3.eps, ops.ACT_RELU, p, self.dropout3.p))
        return self._sp

    def _stack(self):
        if getattr(self, "_dp", None) is None:
            self._dp = ops.DecoderParams([self._params()])
        return self._dp

    def forward_bf(self, tgt, memory, query_pos, tgt_mask):
        """Batch-first forward_post (transformer.py:212-233): the three fused sub-layers in one executor call."""
        return ops.decoder_stack(memory, query_pos, self._stack(), tgt_mask, tgt)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        """Sequence-first signature of the reference (tgt [Q,B,E], memory [S,B,E])."""
        if memory_mask is not None or tgt_key_padding_mask is not None or memory_key_padding_mask is not None or pos is not None:
            raise NotImplementedError("the reference never passes these masks (agqa_model.py:236)")
        qp = query_pos.transpose(0, 1) if query_pos is not None else torch.zeros_like(tgt).transpose(0, 1)
        out = self.forward_bf(tgt.transpose(0, 1).contiguous(), memory.transpose(0, 1).contiguous(), qp.contiguous(),
                              _mask_f32(tgt_mask))
        return out.transpose(0, 1)


def _mask_f32(m):
    return None if m is None else m.float().contiguous()


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerDecoder(nn.Module):
    """transformer.py:86-124."""

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        if norm is not None or return_intermediate:
            raise NotImplementedError("the reference builds the decoder without a final norm (agqa_model.py:99)")
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = None
        self.return_intermediate = False

    def _stack(self):
        if getattr(self, "_dp", None) is None:
            self._dp = ops.DecoderParams([layer._params() for layer in self.layers])
        return self._dp

    def forward_bf(self, tgt, memory, query_pos, tgt_mask):
        """All layers in ONE executor call (shg_decoder_fwd); tgt None = zeros (agqa_model.py:234)."""
        return ops.decoder_stack(memory, query_pos, self._stack(), tgt_mask, tgt)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        """Sequence-first in, returns output.unsqueeze(0) = [1,Q,B,E] like the reference."""
        qp = query_pos.transpose(0, 1).contiguous() if query_pos is not None else torch.zeros_like(tgt).transpose(0, 1)
        out = self.forward_bf(tgt.transpose(0, 1).contiguous(), memory.transpose(0, 1).contiguous(), qp, _mask_f32(tgt_mask))
        return out.transpose(0, 1).unsqueeze(0)
