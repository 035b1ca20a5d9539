"""Host-side modules of the LXRT stack with the reference's class names, constructor arguments,
forward() signatures and state_dict keys (AGQA/src/lxrt/modeling_capsbert.py, cited as mc:LINE).
nn.Linear / nn.LayerNorm / nn.Embedding objects are used only as parameter containers (so that
names, shapes and initialisation match the reference); the arithmetic is the HIP path of ops.py.
"""
import math

import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .engine import engine


class VisualConfig:
    """mc:144-196 (only what the no-capsule path uses)."""

    def __init__(self, l_layers=12, x_layers=5, r_layers=0, hw=7):
        self.l_layers, self.x_layers, self.r_layers = l_layers, x_layers, r_layers
        self.visual_feat_dim = 2048
        self.visual_pos_dim = 4
        self.hw = hw
        self.t = 8
        self.max_spatial_pos_emb = self.t * hw * hw


VISUAL_CONFIG = VisualConfig()


class BertConfig:
    """mc:206-260 - bert-base-uncased defaults."""

    def __init__(self, vocab_size_or_config_json_file=30522, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02, visualization=True):
        self.vocab_size = vocab_size_or_config_json_file
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.visualization = visualization


BertLayerNorm = nn.LayerNorm


class GeLU(nn.Module):
    """Parameter-free placeholder that keeps the Sequential indices of the MLP heads (0, 2, 3)."""

    def forward(self, x):
        return ops.bias_act(x, None, ops.ACT_GELU)


def key_mask_2d(add_mask):
    """Reference masks are additive [B,1,1,Sk] tensors; the attention kernel wants fp32 [B,Sk]."""
    if add_mask is None:
        return K.MASK_NONE, None
    return K.MASK_KEY, add_mask.reshape(add_mask.shape[0], -1).float().contiguous()


def mlp_head(seq, x):
    """nn.Sequential(Linear, GeLU, LayerNorm(1e-12), Linear) of agqa_model.py:105-110 / :135-140:
    the first bias, the GELU and the LayerNorm run as one fused epilogue."""
    h = ops.linear(x, seq[0].weight, None)
    h = ops.bias_res_layernorm(h, seq[0].bias, None, seq[2].weight, seq[2].bias, seq[2].eps, ops.ACT_GELU)
    return ops.linear(h, seq[3].weight, seq[3].bias)


class BertEmbeddings(nn.Module):
    """mc:327-355."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=0)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size, padding_idx=0)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None):
        pos = torch.arange(input_ids.size(1), dtype=torch.long, device=input_ids.device)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        e = ops.embed_sum([input_ids, pos, token_type_ids],
                          [self.word_embeddings.weight, self.position_embeddings.weight,
                           self.token_type_embeddings.weight])
        e = ops.bias_res_layernorm(e, None, None, self.LayerNorm.weight, self.LayerNorm.bias, 1e-12)
        return ops.dropout(e, self.dropout.p)


class HGEmbeddings(nn.Module):
    """mc:299-325: every query embedding + frame-id type embedding -> LayerNorm -> dropout."""

    def __init__(self, num_queries, type_vocab_size, hidden_size, hidden_dropout_prob=0.1, gt_hg=False):
        super().__init__()
        self.gt_hg = gt_hg
        self.word_embeddings = nn.Embedding(num_queries, hidden_size, padding_idx=0)
        self.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size, padding_idx=0)
        self.LayerNorm = BertLayerNorm(hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(hidden_dropout_prob)

    def forward(self, token_type_ids=None, token_ids=None):
        if self.gt_hg and token_ids is not None:
            e = ops.embed_sum([token_ids, token_type_ids], [self.word_embeddings.weight, self.token_type_embeddings.weight])
        else:
            e = ops.embed_sum([token_type_ids], [self.token_type_embeddings.weight], whole=self.word_embeddings.weight)
        e = ops.bias_res_layernorm(e, None, None, self.LayerNorm.weight, self.LayerNorm.bias, 1e-12)
        return ops.dropout(e, self.dropout.p)


class BertAttention(nn.Module):
    """mc:358-421.  The probabilities are never materialised, so the visualisation dictionary of the
    reference (attn / queries / keys) is not produced: the second return value is None."""

    def __init__(self, config, ctx_dim=None):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        if self.attention_head_size != 64:
            raise ValueError("the HIP attention kernel is specialised for head size 64")
        self.all_head_size = config.hidden_size
        ctx_dim = config.hidden_size if ctx_dim is None else ctx_dim
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(ctx_dim, self.all_head_size)
        self.value = nn.Linear(ctx_dim, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def fusion_groups(self, prefix):
        """Parameter groups Engine.adopt lays out back-to-back so that Q/K/V are one GEMM operand."""
        return [[prefix + n + ".weight" for n in ("query", "key", "value")],
                [prefix + n + ".bias" for n in ("query", "key", "value")]]

    def _fused(self):
        if getattr(self, "_fz", None) is None:
            P = ops.ParamConcat
            self._fz = (P([self.query.weight, self.key.weight, self.value.weight]),
                        P([self.query.bias, self.key.bias, self.value.bias]),
                        P([self.key.weight, self.value.weight]), P([self.key.bias, self.value.bias]))
        return self._fz

    def forward(self, hidden_states, context, attention_mask=None):
        kind, mask = key_mask_2d(attention_mask)
        scale = 1.0 / math.sqrt(self.attention_head_size)
        w_qkv, b_qkv, w_kv, b_kv = self._fused()
        if context is hidden_states:
            o = ops.self_attention_qkv(hidden_states, w_qkv, b_qkv, self.num_attention_heads, kind, mask, scale,
                                       self.dropout.p)
        else:
            o = ops.cross_attention_qkv(hidden_states, context, self.query.weight, self.query.bias, w_kv, b_kv,
                                        self.num_attention_heads, kind, mask, scale, self.dropout.p)
        return o, None


class BertAttOutput(nn.Module):
    """mc:424-435."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, None)
        return ops.bias_res_layernorm(h, self.dense.bias, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                      1e-12, ops.ACT_NONE, self.dropout.p)


class BertCrossattLayer(nn.Module):
    """mc:438-447."""

    def __init__(self, config):
        super().__init__()
        self.att = BertAttention(config)
        self.output = BertAttOutput(config)

    def _params(self):
        if getattr(self, "_ap", None) is None:
            a, out = self.att, self.output
            _, _, w_kv, b_kv = a._fused()
            self._ap = ops.AttnParams("cross", a.query.weight, a.query.bias, w_kv, b_kv, out.dense.weight, out.dense.bias,
                                      out.LayerNorm.weight, out.LayerNorm.bias, 1e-12, a.num_attention_heads,
                                      1.0 / math.sqrt(a.attention_head_size), a.dropout.p, out.dropout.p)
        return self._ap

    def forward(self, input_tensor, ctx_tensor, ctx_att_mask=None):
        """Attention + output projection + residual LayerNorm as one fused sub-layer (ops.attn_sublayer)."""
        kind, mask = key_mask_2d(ctx_att_mask)
        return ops.attn_sublayer(input_tensor, ctx_tensor, self._params(), kind, mask), None


class BertSelfattLayer(nn.Module):
    """mc:450-460."""

    def __init__(self, config):
        super().__init__()
        self.self = BertAttention(config)
        self.output = BertAttOutput(config)

    def _params(self):
        if getattr(self, "_ap", None) is None:
            a, out = self.self, self.output
            w_qkv, b_qkv, _, _ = a._fused()
            self._ap = ops.AttnParams("self", w_qkv, b_qkv, None, None, out.dense.weight, out.dense.bias,
                                      out.LayerNorm.weight, out.LayerNorm.bias, 1e-12, a.num_attention_heads,
                                      1.0 / math.sqrt(a.attention_head_size), a.dropout.p, out.dropout.p)
        return self._ap

    def forward(self, input_tensor, attention_mask):
        """Attention + output projection + residual LayerNorm as one fused sub-layer (ops.attn_sublayer)."""
        kind, mask = key_mask_2d(attention_mask)
        return ops.attn_sublayer(input_tensor, None, self._params(), kind, mask), None


class BertIntermediate(nn.Module):
    """mc:463-475."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return ops.bias_act(ops.linear(hidden_states, self.dense.weight, None), self.dense.bias, ops.ACT_GELU)


class BertOutput(nn.Module):
    """mc:478-489."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, None)
        return ops.bias_res_layernorm(h, self.dense.bias, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                      1e-12, ops.ACT_NONE, self.dropout.p)


def ffn_params(owner, inter, out, slot="_fp"):
    """ops.FFNParams of a BertIntermediate + BertOutput pair, cached on `owner`."""
    P = getattr(owner, slot, None)
    if P is None:
        P = ops.FFNParams(inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias, out.LayerNorm.weight,
                          out.LayerNorm.bias, 1e-12, ops.ACT_GELU, 0.0, out.dropout.p)
        object.__setattr__(owner, slot, P)
    return P


class BertLayer(nn.Module):
    """mc:492-503."""

    def __init__(self, config):
        super().__init__()
        self.attention = BertSelfattLayer(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask):
        a, probs = self.attention(hidden_states, attention_mask)
        return ops.ffn_sublayer(a, ffn_params(self, self.intermediate, self.output)), probs


class CrossLayer(nn.Module):
    """mc:624-677: ONE attention module serves both directions, then separate FFNs."""

    def __init__(self, config):
        super().__init__()
        self.visual_attention = BertCrossattLayer(config)
        self.lang_inter = BertIntermediate(config)
        self.lang_output = BertOutput(config)
        self.visn_inter = BertIntermediate(config)
        self.visn_output = BertOutput(config)
        self.visualization = config.visualization

    def cross_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        la, pl = self.visual_attention(lang_input, visn_input, ctx_att_mask=visn_attention_mask)
        va, pv = self.visual_attention(visn_input, lang_input, ctx_att_mask=lang_attention_mask)
        return la, va, pl, pv

    def output_fc(self, lang_input, visn_input):
        return (ops.ffn_sublayer(lang_input, ffn_params(self, self.lang_inter, self.lang_output, "_fl")),
                ops.ffn_sublayer(visn_input, ffn_params(self, self.visn_inter, self.visn_output, "_fv")))

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask, last=None):
        # The two directions only exchange their INPUTS: language <- vision (40 query tokens per sample: a chain of
        # ~10 us kernels) runs on a side stream beside vision <- language, attention and feed-forward alike.
        side = ops.Branch(2, lang_feats, visn_feats, lang_attention_mask, visn_attention_mask)
        with side:
            la, pl = self.visual_attention(lang_feats, visn_feats, ctx_att_mask=visn_attention_mask)
            lo = ops.ffn_sublayer(la, ffn_params(self, self.lang_inter, self.lang_output, "_fl"))
        va, pv = self.visual_attention(visn_feats, lang_feats, ctx_att_mask=lang_attention_mask)
        vo = ops.ffn_sublayer(va, ffn_params(self, self.visn_inter, self.visn_output, "_fv"))
        side.join(lo)
        probs = {"attn_prob_l": [], "attn_prob_v": [], "attn_prob_xl": pl, "attn_prob_xv": pv, "attn_prob_vl": []}
        return lo, vo, probs


class _NotOnThePath(nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError(
            "%s exists only so that checkpoints keep the reference's keys; the hot path is --crossAttnType cross"
            % type(self).__name__)


class SelfCrossLayer(_NotOnThePath):
    """Parameter layout of mc:679-753 (not selected by --crossAttnType cross)."""

    def __init__(self, config):
        super().__init__()
        self.cross_att = BertSelfattLayer(config)
        self.vl_inter = BertIntermediate(config)
        self.vl_output = BertOutput(config)


class CrossAndSelfLayer(_NotOnThePath):
    """Parameter layout of mc:756-830 (not selected by --crossAttnType cross)."""

    def __init__(self, config):
        super().__init__()
        self.visual_attention = BertCrossattLayer(config)
        self.self_att_layer = BertSelfattLayer(config)
        self.vl_inter = BertIntermediate(config)
        self.vl_output = BertOutput(config)


class LearnedPositionalEncoding(nn.Module):
    """lxrt/PositionalEncoding.py:25-41."""

    def __init__(self, max_position_embeddings, embedding_dim, seq_length):
        super().__init__()
        self.pe = nn.Embedding(max_position_embeddings, embedding_dim)
        self.seq_length = seq_length
        self.register_buffer("position_ids", torch.arange(max_position_embeddings).expand((1, -1)))


class VisualFeatEncoder(nn.Module):
    """mc:966-1073, no_caps branch.  `conv` keeps the reference's Sequential indices (1 and 4 hold
    the Conv3d weights); box_fc / pos_layer_norm are constructed but unused, as in the reference."""

    def __init__(self, config, norm_inputs=False, patches=False, no_caps=True):
        super().__init__()
        if not no_caps or patches:
            raise NotImplementedError("only the --noCaps video-feature path is on the hot path")
        self.no_caps = True
        self.conv = nn.Sequential(nn.ZeroPad2d(1), nn.Conv3d(2048, config.hidden_size, kernel_size=(5, 3, 3)), GeLU(),
                                  nn.ZeroPad2d(1), nn.Conv3d(config.hidden_size, config.hidden_size, kernel_size=(5, 3, 3)),
                                  GeLU())
        self.cls_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.caps_dim = config.hidden_size
        self.seq_length = VISUAL_CONFIG.max_spatial_pos_emb + 1
        self.position_encoding = LearnedPositionalEncoding(self.seq_length, self.caps_dim, self.seq_length)
        self.box_fc = nn.Linear(VISUAL_CONFIG.visual_pos_dim, config.hidden_size)
        self.pos_layer_norm = BertLayerNorm(self.caps_dim, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, visn_input):
        feats, _boxes = visn_input
        c1, c2 = self.conv[1], self.conv[4]
        x = ops.visual_conv_tokens(feats, c1.weight, c1.bias, c2.weight, c2.bias, self.cls_token,
                                   self.position_encoding.pe.weight)
        return ops.dropout(x, self.dropout.p), -1


class BertPooler(_NotOnThePath):
    """mc:1505-1517 (parameter holder: the 'self' / 'cross_self' / 'old' poolers are never used)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)


class BertPooler2(nn.Module):
    """mc:1519-1533: tanh(W [h1[:,0] ; h2[:,0]])."""

    def __init__(self, config):
        super().__init__()
        self.dense2 = nn.Linear(config.hidden_size * 2, config.hidden_size)

    def forward(self, hidden_states1, hidden_states2):
        first = torch.cat([hidden_states1[:, 0], hidden_states2[:, 0]], dim=-1)
        y = ops.linear(first, self.dense2.weight, self.dense2.bias)
        return torch.tanh(y.float()).to(y.dtype)


def _cross_layer_dict(config):
    return nn.ModuleDict({"cross": CrossLayer(config), "self": SelfCrossLayer(config),
                          "cross_self": CrossAndSelfLayer(config), "old": CrossLayer(config)})


def _pooler_dict(config):
    return nn.ModuleDict({"cross": BertPooler2(config), "self": BertPooler(config), "cross_self": BertPooler(config),
                          "no_cross": BertPooler2(config), "old": BertPooler(config)})


class _InitMixin:
    def init_bert_weights(self, module):
        """mc:1640-1651."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()


class CrossEncoder(nn.Module, _InitMixin):
    """mc:1075-1215: hyper-graph tokens (+ act/rel type tokens, + cls) x question tokens."""

    def __init__(self, config, cross_attn_type="cross", mask_features=False, num_max_act=3, num_max_rel=8,
                 add_action=False, add_relation=False):
        super().__init__()
        if cross_attn_type != "cross":
            raise NotImplementedError("only --crossAttnType cross is on the hot path")
        self.config = config
        self.cross_attn_type = cross_attn_type
        self.num_max_act, self.num_max_rel = num_max_act, num_max_rel
        self.act_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.rel_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.cross_attn_layer = _cross_layer_dict(config)
        self.num_x_layers = VISUAL_CONFIG.x_layers
        self.x_layers = nn.ModuleList([self.cross_attn_layer[cross_attn_type] for _ in range(self.num_x_layers)])
        self.pooler_dict = _pooler_dict(config)
        self.pooler = self.pooler_dict[cross_attn_type]
        self.apply(self.init_bert_weights)

    def _type_rows(self):
        a, r = self.num_max_act, self.num_max_rel
        rows = torch.cat([self.act_token._shg_store.view(1, -1).expand(a, -1),
                          self.rel_token._shg_store.view(1, -1).expand(r, -1)], dim=0)
        return rows

    def _scatter_type_grad(self, dsum):
        a = self.num_max_act
        per = self.num_max_act + self.num_max_rel
        d = dsum.view(-1, per, dsum.shape[-1]).sum(0)
        if self.act_token._shg_grad is not None:
            self.act_token._shg_grad.view(-1).add_(d[:a].sum(0))
            self.rel_token._shg_grad.view(-1).add_(d[a:].sum(0))
            engine().grad_written(self.act_token)
            engine().grad_written(self.rel_token)

    def forward(self, lang_feats, lang_attention_mask, hg_feats, hg_attention_mask=None,
                output_all_attention_masks=False):
        B, S, D = hg_feats.shape
        per = self.num_max_act + self.num_max_rel
        T = S // per
        addend = self._type_rows().repeat(T, 1)                       # [S, D] fp32
        hg = ops.add_param_rows(hg_feats, self.act_token, addend, self._scatter_type_grad)
        hg = ops.prepend_param_row(hg, self.cls_token)
        ext = None
        if hg_attention_mask is not None:
            m = torch.cat([torch.ones(B, 1, device=hg_feats.device), hg_attention_mask.reshape(B, -1).float()], dim=1)
            ext = ((1.0 - m) * -10000.0)[:, None, None, :]
        for layer in self.x_layers:
            lang_feats, hg, _ = layer(lang_feats, lang_attention_mask, hg, ext)
        pooled = self.pooler(hg, lang_feats)
        return pooled, ([], [], [], [], [])


class NoCapsEncoder(nn.Module):
    """mc:1218-1302."""

    def __init__(self, config, shared_weights=False, cross_attn=False, cross_attn_type="cross", no_caps=True,
                 mask_features=False):
        super().__init__()
        if cross_attn_type != "cross" or mask_features:
            raise NotImplementedError("only --crossAttnType cross without feature masking is on the hot path")
        self.no_caps = no_caps
        self.visn_fc = VisualFeatEncoder(config, no_caps=no_caps)
        self.cross_attn_layer = _cross_layer_dict(config)
        self.num_l_layers, self.num_x_layers, self.num_r_layers = (VISUAL_CONFIG.l_layers, VISUAL_CONFIG.x_layers,
                                                                   VISUAL_CONFIG.r_layers)
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(self.num_l_layers)])
        self.x_layers = nn.ModuleList([self.cross_attn_layer[cross_attn_type] for _ in range(self.num_x_layers)])
        self.r_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_r_layers)])

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask=None,
                output_all_attention_masks=False):
        # The language layers (tiny launches) run on a side stream beside the conv stack + relation layers.  The fork
        # point is here (the side stream waits for what the main stream holds NOW), but their kernels are issued AFTER the
        # relation layers': autograd replays nodes in reverse creation order, and the language backward - ready as soon
        # as the hyper-graph encoder's backward is done - must not queue behind the conv stack's backward, where it would
        # crawl beside the chip-filling conv weight gradients and hold up the optimiser (measured: ~1 ms per step).
        # lang_feats may be a zero-argument callable producing the embeddings (NoCapsModel.forward), issued there too.
        branch = ops.Branch(2, lang_attention_mask)
        with branch:
            pass
        visn_feats, _ = self.visn_fc(visn_feats)
        for layer in self.r_layers:
            visn_feats, _ = layer(visn_feats, visn_attention_mask)
        visn_out = visn_feats
        branch.wait = False
        with branch:
            if callable(lang_feats):
                lang_feats = lang_feats()
            for layer in self.layer:
                lang_feats, _ = layer(lang_feats, lang_attention_mask)
        lang_out = lang_feats
        E = engine()
        if E.defer_x_layers and branch.side is not None and E.deferred_branch is None:
            # --taskHGQA: nothing on the loss path reads the x-layers' output; they continue on the language
            # stream (which first waits for the relation layers' output) and the main stream goes on to the decoders
            with branch.extend(visn_feats):
                for layer in self.x_layers:
                    lang_feats, visn_feats, _ = layer(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
            E.deferred_branch = branch
        else:
            branch.join(lang_feats)
            for layer in self.x_layers:
                lang_feats, visn_feats, _ = layer(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
        return lang_feats, visn_feats, ([], [], [], [], [], (lang_out, lang_attention_mask, visn_out, visn_attention_mask))


class BertPreTrainedModel(nn.Module, _InitMixin):
    """mc:1625-1699.  `from_pretrained` builds from BertConfig defaults: there is no network here and
    the training entry points always re-initialise (--fromScratch) or load a checkpoint."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, BertConfig):
            raise ValueError("Parameter config in `%s(config)` should be an instance of class `BertConfig`."
                             % type(self).__name__)
        self.config = config

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, state_dict=None, cache_dir=None, from_tf=False,
                        *inputs, **kwargs):
        model = cls(BertConfig(30522), *inputs, **kwargs)
        if state_dict is not None:
            model.load_state_dict(state_dict, strict=False)
        return model


def additive_mask(mask01, like_ids):
    """mc:1826-1834: (1 - m) * -10000 as [B,1,1,S] (the 0/1 mask goes through the ids' integer dtype first)."""
    ext = mask01.unsqueeze(1).unsqueeze(2).to(dtype=like_ids.dtype)
    return (1.0 - ext) * -10000.0


class NoCapsModel(BertPreTrainedModel):
    """mc:1787-1857."""

    def __init__(self, config, shared_weights=False, cross_attn=False, cross_attn_type="cross", no_caps=True):
        super().__init__(config)
        self.cross_attn_type = cross_attn_type
        self.embeddings = BertEmbeddings(config)
        self.encoder = NoCapsEncoder(config, cross_attn_type=cross_attn_type, no_caps=no_caps)
        self.pooler_dict = _pooler_dict(config)
        self.pooler = self.pooler_dict[cross_attn_type]
        self.apply(self.init_bert_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, visual_feats=None,
                visual_attention_mask=None, output_all_attention_masks=False):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        ext = additive_mask(attention_mask, input_ids)
        vmask = visual_feats[1]                      # the "boxes" slot carries the visual 0/1 mask, mc:1836
        ext_v = additive_mask(vmask, input_ids) if vmask is not None else None
        # the question embeddings only feed the language layers: the encoder issues them on that side stream (no
        # parameter is read on the main stream before conv1, see BertAdam.step)
        lang, visn, probs = self.encoder(lambda: self.embeddings(input_ids, token_type_ids), ext, visn_feats=visual_feats,
                                         visn_attention_mask=ext_v)
        with ops.deferred_branch():                  # (the x-layers' stream, when they were deferred)
            pooled = self.pooler(visn, lang)
        return (lang, visn), pooled, probs


class LXRTFeatureExtraction(BertPreTrainedModel):
    """mc:2128-2192."""

    def __init__(self, config, mode="lxr", skip_connection=False, shared_weights=False, cross_attn=False,
                 cross_attn_type="cross", freeze_weights=False, patches=False, vit_init=False, start_index=0,
                 no_caps=True, margin=0.1):
        super().__init__(config)
        if not no_caps:
            raise NotImplementedError("only --noCaps is on the hot path")
        self.bert = NoCapsModel(config, cross_attn_type=cross_attn_type, no_caps=no_caps)
        self.mode = mode
        self.apply(self.init_bert_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, visual_feats=None,
                visual_attention_mask=None, output_all_attention_masks=False):
        feat_seq, pooled, probs = self.bert(input_ids, token_type_ids, attention_mask, visual_feats=visual_feats,
                                            visual_attention_mask=visual_attention_mask,
                                            output_all_attention_masks=output_all_attention_masks)
        if self.mode == "x":
            return pooled, probs
        if "x" in self.mode and ("l" in self.mode or "r" in self.mode):
            return feat_seq, pooled, probs
        return feat_seq, probs


class BertNoCapsEncoder(nn.Module):
    """mc:2200-2232 (question-only model): only `layer` runs; `r_layers` holds parameters."""

    def __init__(self, config, no_caps=True, **_):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(VISUAL_CONFIG.l_layers)])
        self.r_layers = nn.ModuleList([BertLayer(config) for _ in range(VISUAL_CONFIG.r_layers)])

    def forward(self, lang_feats, lang_attention_mask, visn_feats=None, visn_attention_mask=None,
                output_all_attention_masks=False):
        for layer in self.layer:
            lang_feats, _ = layer(lang_feats, lang_attention_mask)
        return lang_feats, lang_attention_mask


class BertNoCapsModel(BertPreTrainedModel):
    """mc:2310-2344."""

    def __init__(self, config, shared_weights=False, cross_attn=False, cross_attn_type="self", no_caps=True):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertNoCapsEncoder(config, no_caps=no_caps)
        self.apply(self.init_bert_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        ext = additive_mask(attention_mask, input_ids)
        lang, probs = self.encoder(self.embeddings(input_ids, token_type_ids), ext, visn_feats=None)
        return lang, lang[:, 0], probs


class BertFeatureExtraction(BertPreTrainedModel):
    """mc:2417-2467."""

    def __init__(self, config, mode="lxr", no_caps=True, **_):
        super().__init__(config)
        self.bert = BertNoCapsModel(config, no_caps=no_caps)
        self.mode = mode
        self.apply(self.init_bert_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, output_all_attention_masks=False):
        return self.bert(input_ids, token_type_ids, attention_mask)
