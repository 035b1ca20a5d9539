"""CPU oracle for the SHG-VQA hot path (TEST INFRASTRUCTURE - never imported by the product).

A restatement, in plain fp32 PyTorch on the CPU, of the arithmetic the reference performs on
the path named by BASELINE.json: LXRT language / relation / cross stack, the situation
hyper-graph set decoder, the answer heads, the per-frame Hungarian matcher, the set loss and
the BertAdam update.  It is written as pure functions over a flat ``{name: tensor}`` parameter
dictionary whose names are the reference's ``named_parameters()`` names, so the same
name-derived weights (oracle/detweights.py) can be put into the reference, this oracle and
the HIP product.

PINNING: oracle/gen_golden.py imports the real reference in the build container
(oracle/ref_harness.py), runs it on seeded inputs with those weights and commits the
outputs under tests/golden/; tests/test_oracle_golden.py checks this file against them.
The LSAP solver (scipy, a third-party dependency of the reference, requirements.txt:82) is
restated in oracle/lsap.c / lsap_py() below and pinned by golden vectors produced with the
SciPy installed in the build container.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference citations use ``mc`` = AGQA/src/lxrt/modeling_capsbert.py.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

HID = 768
HEADS = 12
DH = 64
TEXT_LEN = 40
VIS_TOKENS = 393


class Cfg:
    """Shape/flag bundle (param.py:82-137 defaults for the AGQA HGQA configuration)."""

    def __init__(self, llayers=5, xlayers=2, rlayers=5, dlayers=5, num_answers=171,
                 rel_classes=457, act_classes=158, num_rel=8, num_act=3, num_situations=16,
                 task="hgqa", use_hg_mask=False):
        self.llayers, self.xlayers, self.rlayers, self.dlayers = llayers, xlayers, rlayers, dlayers
        self.num_answers = num_answers
        self.rel_classes, self.act_classes = rel_classes, act_classes
        self.num_rel, self.num_act, self.num_situations = num_rel, num_act, num_situations
        self.task = task
        self.use_hg_mask = use_hg_mask

    @property
    def rel_queries(self):
        return self.num_rel * self.num_situations

    @property
    def act_queries(self):
        return self.num_act * self.num_situations


# --------------------------------------------------------------------------------------
# parameter specification (names and shapes of the tensors that take part in the path)
# --------------------------------------------------------------------------------------
def _bert_layer_spec(pre):
    s = []
    for n in ("query", "key", "value"):
        s += [(f"{pre}.attention.self.{n}.weight", (HID, HID)), (f"{pre}.attention.self.{n}.bias", (HID,))]
    s += [(f"{pre}.attention.output.dense.weight", (HID, HID)), (f"{pre}.attention.output.dense.bias", (HID,)),
          (f"{pre}.attention.output.LayerNorm.weight", (HID,)), (f"{pre}.attention.output.LayerNorm.bias", (HID,)),
          (f"{pre}.intermediate.dense.weight", (4 * HID, HID)), (f"{pre}.intermediate.dense.bias", (4 * HID,)),
          (f"{pre}.output.dense.weight", (HID, 4 * HID)), (f"{pre}.output.dense.bias", (HID,)),
          (f"{pre}.output.LayerNorm.weight", (HID,)), (f"{pre}.output.LayerNorm.bias", (HID,))]
    return s


def _cross_layer_spec(pre):
    s = []
    for n in ("query", "key", "value"):
        s += [(f"{pre}.visual_attention.att.{n}.weight", (HID, HID)), (f"{pre}.visual_attention.att.{n}.bias", (HID,))]
    s += [(f"{pre}.visual_attention.output.dense.weight", (HID, HID)),
          (f"{pre}.visual_attention.output.dense.bias", (HID,)),
          (f"{pre}.visual_attention.output.LayerNorm.weight", (HID,)),
          (f"{pre}.visual_attention.output.LayerNorm.bias", (HID,))]
    for side in ("lang", "visn"):
        s += [(f"{pre}.{side}_inter.dense.weight", (4 * HID, HID)), (f"{pre}.{side}_inter.dense.bias", (4 * HID,)),
              (f"{pre}.{side}_output.dense.weight", (HID, 4 * HID)), (f"{pre}.{side}_output.dense.bias", (HID,)),
              (f"{pre}.{side}_output.LayerNorm.weight", (HID,)), (f"{pre}.{side}_output.LayerNorm.bias", (HID,))]
    return s


def _decoder_layer_spec(pre, ffn=2048):
    s = []
    for a in ("self_attn", "multihead_attn"):
        s += [(f"{pre}.{a}.in_proj_weight", (3 * HID, HID)), (f"{pre}.{a}.in_proj_bias", (3 * HID,)),
              (f"{pre}.{a}.out_proj.weight", (HID, HID)), (f"{pre}.{a}.out_proj.bias", (HID,))]
    s += [(f"{pre}.linear1.weight", (ffn, HID)), (f"{pre}.linear1.bias", (ffn,)),
          (f"{pre}.linear2.weight", (HID, ffn)), (f"{pre}.linear2.bias", (HID,))]
    for n in ("norm1", "norm2", "norm3"):
        s += [(f"{pre}.{n}.weight", (HID,)), (f"{pre}.{n}.bias", (HID,))]
    return s


def _head_spec(pre, n_out):
    return [(f"{pre}.0.weight", (2 * HID, HID)), (f"{pre}.0.bias", (2 * HID,)),
            (f"{pre}.2.weight", (2 * HID,)), (f"{pre}.2.bias", (2 * HID,)),
            (f"{pre}.3.weight", (n_out, 2 * HID)), (f"{pre}.3.bias", (n_out,))]


def _text_encoder_spec(bert, cfg):
    emb = f"{bert}.embeddings"
    s = [(f"{emb}.word_embeddings.weight", (30522, HID)), (f"{emb}.position_embeddings.weight", (512, HID)),
         (f"{emb}.token_type_embeddings.weight", (2, HID)),
         (f"{emb}.LayerNorm.weight", (HID,)), (f"{emb}.LayerNorm.bias", (HID,))]
    for i in range(cfg.llayers):
        s += _bert_layer_spec(f"{bert}.encoder.layer.{i}")
    return s


def param_spec(cfg):
    """(name, shape) of every parameter the path *uses* (a subset of the reference's 576)."""
    if cfg.task == "q":
        return _text_encoder_spec("bert_encoder.model.bert", cfg) + _head_spec("logit_fc", cfg.num_answers)
    bert = "lxrt_encoder.model.bert"
    s = _text_encoder_spec(bert, cfg)
    vf = f"{bert}.encoder.visn_fc"
    s += [(f"{vf}.cls_token", (1, 1, HID)),
          (f"{vf}.conv.1.weight", (HID, 2048, 5, 3, 3)), (f"{vf}.conv.1.bias", (HID,)),
          (f"{vf}.conv.4.weight", (HID, HID, 5, 3, 3)), (f"{vf}.conv.4.bias", (HID,)),
          (f"{vf}.position_encoding.pe.weight", (VIS_TOKENS, HID))]
    for i in range(cfg.rlayers):
        s += _bert_layer_spec(f"{bert}.encoder.r_layers.{i}")
    s += _cross_layer_spec(f"{bert}.encoder.cross_attn_layer.cross")
    s += [(f"{bert}.pooler_dict.cross.dense2.weight", (HID, 2 * HID)), (f"{bert}.pooler_dict.cross.dense2.bias", (HID,))]
    if cfg.task == "hgqa":
        s += [("hgq_encoder.act_token", (1, 1, HID)), ("hgq_encoder.rel_token", (1, 1, HID)),
              ("hgq_encoder.cls_token", (1, 1, HID))]
        s += _cross_layer_spec("hgq_encoder.cross_attn_layer.cross")
        s += [("hgq_encoder.pooler_dict.cross.dense2.weight", (HID, 2 * HID)),
              ("hgq_encoder.pooler_dict.cross.dense2.bias", (HID,))]
        for nm, nq in (("relation_query_embed", cfg.rel_queries), ("action_query_embed", cfg.act_queries)):
            s += [(f"{nm}.word_embeddings.weight", (nq, HID)), (f"{nm}.token_type_embeddings.weight", (16, HID)),
                  (f"{nm}.LayerNorm.weight", (HID,)), (f"{nm}.LayerNorm.bias", (HID,))]
        for dec in ("rel_decoder", "action_decoder"):
            for i in range(cfg.dlayers):
                s += _decoder_layer_spec(f"{dec}.layers.{i}")
        s += _head_spec("class_embed", cfg.rel_classes) + _head_spec("action_embed", cfg.act_classes)
    s += _head_spec("logit_fc", cfg.num_answers)
    return s


def det_params(cfg, base_seed=2024, requires_grad=False):
    from . import detweights
    w = detweights.fill(param_spec(cfg), base_seed)
    return {k: torch.from_numpy(v).requires_grad_(requires_grad) for k, v in w.items()}


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def erf_gelu(x):
    """mc:127-133 - exact erf form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def dense(p, pre, x):
    return F.linear(x, p[pre + ".weight"], p[pre + ".bias"])


def lnorm(p, pre, x, eps):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], eps)


def _drop(x, rate, train):
    return F.dropout(x, rate, True) if (train and rate > 0) else x


def _heads(x):
    b, s, _ = x.shape
    return x.view(b, s, HEADS, DH).transpose(1, 2)


def bert_attention(p, pre, hidden, context, add_mask, train=False):
    """mc:384-421.  add_mask: additive (B,1,1,Sk) or None."""
    q, k, v = (_heads(dense(p, f"{pre}.{n}", t)) for n, t in
               (("query", hidden), ("key", context), ("value", context)))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(DH)
    if add_mask is not None:
        scores = scores + add_mask
    probs = _drop(torch.softmax(scores, dim=-1), 0.1, train)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(hidden.shape[0], hidden.shape[1], HID)
    return ctx


def residual_out(p, pre, y, residual, train=False):
    """BertAttOutput / BertOutput, mc:431-435, :485-489."""
    return lnorm(p, f"{pre}.LayerNorm", _drop(dense(p, f"{pre}.dense", y), 0.1, train) + residual, 1e-12)


def bert_layer(p, pre, x, add_mask, train=False):
    """mc:499-503."""
    a = bert_attention(p, f"{pre}.attention.self", x, x, add_mask, train)
    a = residual_out(p, f"{pre}.attention.output", a, x, train)
    h = erf_gelu(dense(p, f"{pre}.intermediate.dense", a))
    return residual_out(p, f"{pre}.output", h, a, train)


def cross_layer(p, pre, lang, lang_mask, visn, visn_mask, train=False):
    """mc:658-677: one attention module serves both directions, then two FFNs."""
    att = f"{pre}.visual_attention"
    la = residual_out(p, f"{att}.output", bert_attention(p, f"{att}.att", lang, visn, visn_mask, train), lang, train)
    va = residual_out(p, f"{att}.output", bert_attention(p, f"{att}.att", visn, lang, lang_mask, train), visn, train)
    lo = residual_out(p, f"{pre}.lang_output", erf_gelu(dense(p, f"{pre}.lang_inter.dense", la)), la, train)
    vo = residual_out(p, f"{pre}.visn_output", erf_gelu(dense(p, f"{pre}.visn_inter.dense", va)), va, train)
    return lo, vo


def pooler2(p, pre, h1, h2):
    """mc:1525-1533."""
    return torch.tanh(dense(p, f"{pre}.dense2", torch.cat([h1[:, 0], h2[:, 0]], dim=-1)))


def text_embeddings(p, pre, ids, seg, train=False):
    """mc:341-355."""
    pos = torch.arange(ids.shape[1])
    # all three tables are nn.Embedding(padding_idx=0) (mc:332-334): row 0 receives no gradient
    e = F.embedding(ids, p[f"{pre}.word_embeddings.weight"], padding_idx=0) \
        + F.embedding(pos, p[f"{pre}.position_embeddings.weight"], padding_idx=0)[None] \
        + F.embedding(seg, p[f"{pre}.token_type_embeddings.weight"], padding_idx=0)
    return _drop(lnorm(p, f"{pre}.LayerNorm", e, 1e-12), 0.1, train)


def visual_tokens(p, pre, feat, train=False):
    """mc:1037-1073 (no_caps branch): two (5,3,3) convs with spatial zero padding 1, erf-GELU,
    tokens in (t,h,w) order, cls token first, learned positions."""
    x = F.conv3d(F.pad(feat.float(), (1, 1, 1, 1)), p[f"{pre}.conv.1.weight"], p[f"{pre}.conv.1.bias"])
    x = erf_gelu(x)
    x = F.conv3d(F.pad(x, (1, 1, 1, 1)), p[f"{pre}.conv.4.weight"], p[f"{pre}.conv.4.bias"])
    x = erf_gelu(x)
    b, c = x.shape[:2]
    tok = x.permute(0, 2, 3, 4, 1).reshape(b, -1, c)
    tok = torch.cat([p[f"{pre}.cls_token"].expand(b, -1, -1), tok], dim=1)
    tok = tok + p[f"{pre}.position_encoding.pe.weight"][None, : tok.shape[1]]
    return _drop(tok, 0.1, train)


def additive_mask(m01):
    """mc:1826-1842: (1 - m) * -10000, broadcast (B,1,1,S)."""
    return (1.0 - m01.float())[:, None, None, :] * -10000.0


def lxrt_forward(p, cfg, input_ids, input_mask, segment_ids, feat, pos, train=False, bert="lxrt_encoder.model.bert"):
    """NoCapsModel.forward + NoCapsEncoder.forward (mc:1814-1857, :1254-1302)."""
    lmask = additive_mask(input_mask)
    vmask = additive_mask(pos) if pos is not None else None
    lang = text_embeddings(p, f"{bert}.embeddings", input_ids, segment_ids, train)
    visn = visual_tokens(p, f"{bert}.encoder.visn_fc", feat, train)
    for i in range(cfg.llayers):
        lang = bert_layer(p, f"{bert}.encoder.layer.{i}", lang, lmask, train)
    lang_pre_x = lang
    for i in range(cfg.rlayers):
        visn = bert_layer(p, f"{bert}.encoder.r_layers.{i}", visn, vmask, train)
    memory = visn
    for _ in range(cfg.xlayers):                      # the same CrossLayer object n times, mc:1247-1249
        lang, visn = cross_layer(p, f"{bert}.encoder.cross_attn_layer.cross", lang, lmask, visn, vmask, train)
    pooled = pooler2(p, f"{bert}.pooler_dict.cross", visn, lang)
    return dict(lang=lang, visn=visn, pooled=pooled, lang_pre_x=lang_pre_x, lang_mask=lmask, memory=memory)


def text_only_forward(p, cfg, input_ids, input_mask, segment_ids, train=False, bert="bert_encoder.model.bert"):
    """BertNoCapsModel.forward (mc:2324-2344): embeddings + llayers BertLayers, output = token 0."""
    lmask = additive_mask(input_mask)
    lang = text_embeddings(p, f"{bert}.embeddings", input_ids, segment_ids, train)
    for i in range(cfg.llayers):
        lang = bert_layer(p, f"{bert}.encoder.layer.{i}", lang, lmask, train)
    return lang[:, 0]


def hg_query_embeddings(p, pre, seg_ids, rate, train=False):
    """HGEmbeddings.forward, mc:313-325: every query row + frame-id type embedding -> LN."""
    # the query table is used whole (its row 0 trains); the frame-id table has padding_idx=0 (mc:305-306)
    e = p[f"{pre}.word_embeddings.weight"][None] \
        + F.embedding(seg_ids, p[f"{pre}.token_type_embeddings.weight"], padding_idx=0)
    return _drop(lnorm(p, f"{pre}.LayerNorm", e, 1e-12), rate, train)


def frame_causal_mask(num_situations, per_frame):
    """entry.py:114-121: -inf where the key's frame is later than the query's frame."""
    fr = torch.arange(num_situations).repeat_interleave(per_frame)
    m = torch.zeros(fr.numel(), fr.numel())
    m[fr[None, :] > fr[:, None]] = float("-inf")
    return m


def mha(p, pre, q_in, k_in, v_in, attn_mask, rate, train=False):
    """torch.nn.MultiheadAttention as used by transformer.py:192-193 (batch-first here)."""
    w, b = p[f"{pre}.in_proj_weight"], p[f"{pre}.in_proj_bias"]
    q = _heads(F.linear(q_in, w[:HID], b[:HID])) * (1.0 / math.sqrt(DH))
    k = _heads(F.linear(k_in, w[HID:2 * HID], b[HID:2 * HID]))
    v = _heads(F.linear(v_in, w[2 * HID:], b[2 * HID:]))
    s = torch.matmul(q, k.transpose(-1, -2))
    if attn_mask is not None:
        s = s + attn_mask
    a = _drop(torch.softmax(s, dim=-1), rate, train)
    o = torch.matmul(a, v).transpose(1, 2).reshape(q_in.shape[0], q_in.shape[1], HID)
    return dense(p, f"{pre}.out_proj", o)


def decoder_layer(p, pre, tgt, memory, query_pos, tgt_mask, rate=0.15, train=False):
    """transformer.py:212-233 (post-norm, ReLU FFN, LayerNorm eps 1e-5)."""
    qk = tgt + query_pos
    tgt = lnorm(p, f"{pre}.norm1", tgt + _drop(mha(p, f"{pre}.self_attn", qk, qk, tgt, tgt_mask, rate, train), rate, train), 1e-5)
    x = mha(p, f"{pre}.multihead_attn", tgt + query_pos, memory, memory, None, rate, train)
    tgt = lnorm(p, f"{pre}.norm2", tgt + _drop(x, rate, train), 1e-5)
    h = _drop(torch.relu(dense(p, f"{pre}.linear1", tgt)), rate, train)
    return lnorm(p, f"{pre}.norm3", tgt + _drop(dense(p, f"{pre}.linear2", h), rate, train), 1e-5)


def mlp_head(p, pre, x):
    """agqa_model.py:105-110: Linear -> erf-GELU -> LayerNorm(1e-12) -> Linear."""
    return dense(p, f"{pre}.3", lnorm(p, f"{pre}.2", erf_gelu(dense(p, f"{pre}.0", x)), 1e-12))


def hg_decode(p, cfg, memory, rel_seg, act_seg, train=False):
    """The "HGDecoder" block, agqa_model.py:220-260."""
    b = memory.shape[0]
    out = {}
    for tag, dec, emb, head, seg, per, rate in (
            ("rel", "rel_decoder", "relation_query_embed", "class_embed", rel_seg, cfg.num_rel, 0.1),
            ("act", "action_decoder", "action_query_embed", "action_embed", act_seg, cfg.num_act, 0.15)):
        qpos = hg_query_embeddings(p, emb, seg, rate, train)
        mask = frame_causal_mask(cfg.num_situations, per)
        x = torch.zeros_like(qpos)
        for i in range(cfg.dlayers):
            x = decoder_layer(p, f"{dec}.layers.{i}", x, memory, qpos, mask, 0.15, train)
        out[tag + "_out"] = x
        out[tag + "_preds"] = mlp_head(p, head, x)
    t = cfg.num_situations
    out["hg_in"] = torch.cat([out["act_out"].view(b, t, cfg.num_act, HID),
                              out["rel_out"].view(b, t, cfg.num_rel, HID)], dim=2).view(b, -1, HID)
    return out


def hg_cross_encoder(p, cfg, lang, lang_mask, hg_in, hg_mask01=None, train=False, pre="hgq_encoder"):
    """CrossEncoder.forward, mc:1152-1215."""
    b = hg_in.shape[0]
    types = torch.cat([p[f"{pre}.act_token"].expand(b, cfg.num_act, -1),
                       p[f"{pre}.rel_token"].expand(b, cfg.num_rel, -1)], dim=1)
    hg = (hg_in.view(b, cfg.num_situations, -1, HID) + types[:, None]).view(b, -1, HID)
    hg = torch.cat([p[f"{pre}.cls_token"].expand(b, -1, -1), hg], dim=1)
    hmask = None
    if hg_mask01 is not None:
        hmask = additive_mask(torch.cat([torch.ones(b, 1), hg_mask01.view(b, -1).float()], dim=1))
    for _ in range(cfg.xlayers):
        lang, hg = cross_layer(p, f"{pre}.cross_attn_layer.cross", lang, lang_mask, hg, hmask, train)
    return pooler2(p, f"{pre}.pooler_dict.cross", hg, lang)


def agqa_forward(p, cfg, batch, train=False):
    """AGQAModel.forward, agqa_model.py:166-269.  batch: dict of CPU tensors."""
    if cfg.task == "q":
        x = text_only_forward(p, cfg, batch["input_ids"], batch["input_mask"], batch["segment_ids"], train)
        return dict(logit=mlp_head(p, "logit_fc", x))
    enc = lxrt_forward(p, cfg, batch["input_ids"], batch["input_mask"], batch["segment_ids"],
                       batch["feat"], batch["pos"], train)
    res = dict(logit=mlp_head(p, "logit_fc", enc["pooled"]), memory=enc["memory"], lang_pre_x=enc["lang_pre_x"],
               pooled=enc["pooled"])
    if cfg.task == "vqa":
        return res
    dec = hg_decode(p, cfg, enc["memory"], batch["rel_segment_ids"], batch["act_segment_ids"], train)
    hg_mask = batch.get("hg_mask") if cfg.use_hg_mask else None
    x = hg_cross_encoder(p, cfg, enc["lang_pre_x"], enc["lang_mask"], dec["hg_in"], hg_mask, train)
    res.update(rel_preds=dec["rel_preds"], act_preds=dec["act_preds"], hg_in=dec["hg_in"],
               hg_pooled=x, hg_logit=mlp_head(p, "logit_fc", x))
    return res


# --------------------------------------------------------------------------------------
# linear sum assignment (scipy.optimize.linear_sum_assignment restated)
# --------------------------------------------------------------------------------------
def lsap_py(cost):
    """Pure-Python shortest-augmenting-path LSAP with SciPy's scan order and tie rules.
    cost: 2-D float array.  Returns (row_idx, col_idx) int64 arrays sorted by row."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    flipped = nc < nr
    if flipped:
        cost = cost.T.copy()
        nr, nc = nc, nr
    u = np.zeros(nr)
    v = np.zeros(nc)
    col4row = -np.ones(nr, np.int64)
    row4col = -np.ones(nc, np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        path = -np.ones(nc, np.int64)
        in_sr = np.zeros(nr, bool)
        in_sc = np.zeros(nc, bool)
        remaining = [nc - 1 - t for t in range(nc)]
        min_val, i, sink = 0.0, cur, -1
        while sink == -1:
            in_sr[i] = True
            best, lowest = -1, np.inf
            for it, j in enumerate(remaining):
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j], spc[j] = i, r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest, best = spc[j], it
            min_val = lowest
            j = remaining[best]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            in_sc[j] = True
            remaining[best] = remaining[-1]
            remaining.pop()
        u[cur] += min_val
        for r_ in range(nr):
            if in_sr[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for c_ in range(nc):
            if in_sc[c_]:
                v[c_] -= min_val - spc[c_]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if flipped:
        order = np.argsort(col4row, kind="stable")
        return col4row[order].astype(np.int64), order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), col4row.astype(np.int64)


_LSAP_LIB = None


def _lsap_lib():
    """Builds (gcc) and loads oracle/lsap.c."""
    global _LSAP_LIB
    if _LSAP_LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "liblsap_oracle.so")
        src = os.path.join(here, "lsap.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        lib = ctypes.CDLL(so)
        lib.lsap_solve.restype = ctypes.c_int
        lib.lsap_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _LSAP_LIB = lib
    return _LSAP_LIB


def lsap_c(cost):
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.zeros(max(n, 1), np.int64)
    cols = np.zeros(max(n, 1), np.int64)
    got = _lsap_lib().lsap_solve(cost.ctypes.data, nr, nc, rows.ctypes.data, cols.ctypes.data)
    assert got == n, (got, n)
    return rows[:n].copy(), cols[:n].copy()


def hungarian_per_frame(pred_logits, frame_targets, clip_len=16, solver=None):
    """HungarianMatcher.forward, per-frame branch (matcher.py:62-80).
    pred_logits (B,Q,C); frame_targets: list of B*clip_len int64 tensors (class ids of each frame).
    Returns a list of B*clip_len (query_idx, target_idx) int64 tensor pairs."""
    solver = solver or lsap_c
    b, q, c = pred_logits.shape
    per = q // clip_len
    prob = pred_logits.detach().reshape(b * clip_len, per, c).float().softmax(-1)
    out = []
    for n, tg in enumerate(frame_targets):
        cost = -prob[n][:, tg.long()]                            # fp32 (per, n_tgt)
        r, cidx = solver(cost.double().numpy())
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
    return out


def set_target_grid(frame_targets, indices, n_frames, per, background=0):
    """agqaHGQA.py:215-220: grid of class ids, background except matched (frame, query) slots."""
    grid = torch.full((n_frames, per), background, dtype=torch.int64)
    for n, (tg, (qi, ti)) in enumerate(zip(frame_targets, indices)):
        grid[n, qi] = tg.long()[ti]
    return grid


def set_loss(pred_logits, frame_targets, indices, class_weight, clip_len=16):
    """AGQA.loss_labels (agqaHGQA.py:203-229): weighted CE over every slot + class_error."""
    b, q, c = pred_logits.shape
    per = q // clip_len
    logits = pred_logits.reshape(b * clip_len, per, c)
    grid = set_target_grid(frame_targets, indices, b * clip_len, per)
    loss = F.cross_entropy(logits.transpose(1, 2), grid, class_weight)
    matched = torch.cat([torch.stack([torch.full_like(qi, n), qi]) for n, (qi, _) in enumerate(indices)], dim=1)
    if matched.shape[1]:
        top1 = logits[matched[0], matched[1]].argmax(-1)
        err = 100.0 - 100.0 * (top1 == grid[matched[0], matched[1]]).float().mean()
    else:
        err = torch.tensor(100.0)
    return loss, err, grid


def hgqa_losses(out, batch, cfg):
    """agqaHGQA.py:344-378: BCE*C + CE_w(rel) + CE_w(act)."""
    n_ans = out["hg_logit"].shape[1]
    bce = F.binary_cross_entropy_with_logits(out["hg_logit"], batch["target"]) * n_ans
    w_rel = torch.ones(cfg.rel_classes)
    w_rel[0] = 0.1
    w_act = torch.ones(cfg.act_classes)
    w_act[0] = 0.1
    rel_idx = hungarian_per_frame(out["rel_preds"], batch["rel_targets"], cfg.num_situations)
    act_idx = hungarian_per_frame(out["act_preds"], batch["act_targets"], cfg.num_situations)
    rel_ce, rel_err, rel_grid = set_loss(out["rel_preds"], batch["rel_targets"], rel_idx, w_rel, cfg.num_situations)
    act_ce, act_err, act_grid = set_loss(out["act_preds"], batch["act_targets"], act_idx, w_act, cfg.num_situations)
    return dict(total=bce + rel_ce + act_ce, bce=bce, rel_ce=rel_ce, act_ce=act_ce, rel_err=rel_err,
                act_err=act_err, rel_idx=rel_idx, act_idx=act_idx, rel_grid=rel_grid, act_grid=act_grid)


# --------------------------------------------------------------------------------------
# optimiser (optimization.py:101-180) and gradient clipping (agqaHGQA.py:391)
# --------------------------------------------------------------------------------------
def warmup_linear(x, warmup):
    return x / warmup if x < warmup else max((x - 1.0) / (warmup - 1.0), 0.0)


def clip_grad_norm(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_: coef = min(max_norm / (total + 1e-6), 1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def bertadam_step(params, grads, state, lr, step, t_total, warmup=0.1, b1=0.9, b2=0.999, eps=1e-6, wd=0.01):
    """One BertAdam update for every tensor; `step` = number of updates already applied."""
    lr_t = lr * warmup_linear(step / t_total, warmup) if t_total != -1 else lr
    for k, w in params.items():
        g = grads.get(k)
        if g is None:
            continue
        m, v = state.setdefault(k, (torch.zeros_like(w), torch.zeros_like(w)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = m / (v.sqrt() + eps) + wd * w
        w.sub_(lr_t * upd)
    return lr_t


def train_step(p, cfg, batch, state, lr, step, t_total, train=False):
    """One full optimiser step of agqaHGQA.py:262-392 on the oracle; p requires grad."""
    out = agqa_forward(p, cfg, batch, train)
    if cfg.task == "hgqa":
        losses = hgqa_losses(out, batch, cfg)
        total = losses["total"]
    else:
        total = F.binary_cross_entropy_with_logits(out["logit"], batch["target"]) * out["logit"].shape[1]
        losses = dict(total=total)
    names = [k for k in p if p[k].requires_grad]
    gl = torch.autograd.grad(total, [p[k] for k in names], allow_unused=True)
    grads = {k: g for k, g in zip(names, gl) if g is not None}
    norm = clip_grad_norm(list(grads.values()), 5.0)
    with torch.no_grad():
        bertadam_step(p, grads, state, lr, step, t_total)
    return out, losses, grads, norm


# --------------------------------------------------------------------------------------
# synthetic AGQA-shaped batches
# --------------------------------------------------------------------------------------
def synthetic_batch(bsz, cfg, seed=1234, with_feat=True):
    g = torch.Generator().manual_seed(seed)
    t = cfg.num_situations
    ids = torch.zeros(bsz, TEXT_LEN, dtype=torch.int64)
    mask = torch.zeros(bsz, TEXT_LEN, dtype=torch.int64)
    for i in range(bsz):
        n = int(torch.randint(8, 31, (1,), generator=g))
        ids[i, :n] = torch.randint(1000, 30522, (n,), generator=g)
        ids[i, 0], ids[i, n - 1] = 101, 102
        mask[i, :n] = 1
    batch = dict(input_ids=ids, input_mask=mask, segment_ids=torch.zeros_like(ids))
    if with_feat:
        batch["feat"] = torch.randn(bsz, 2048, 16, 7, 7, generator=g)
        batch["pos"] = torch.ones(bsz, VIS_TOKENS, dtype=torch.float64)

    def ragged(per, n_cls):
        tri = torch.zeros(bsz, t, per, dtype=torch.int64)
        lens = torch.randint(0, per + 1, (bsz, t), generator=g)
        for i in range(bsz):
            for f in range(t):
                n = int(lens[i, f])
                tri[i, f, :n] = torch.randperm(n_cls - 1, generator=g)[:n] + 1
        return tri, lens

    rel, rel_len = ragged(cfg.num_rel, cfg.rel_classes)
    act, act_len = ragged(cfg.num_act, cfg.act_classes)
    batch.update(rel_triplets=rel, lengths=rel_len, act_tokens=act, act_lengths=act_len)
    batch["rel_segment_ids"] = torch.arange(t).repeat_interleave(cfg.num_rel)[None].expand(bsz, -1).contiguous()
    batch["act_segment_ids"] = torch.arange(t).repeat_interleave(cfg.num_act)[None].expand(bsz, -1).contiguous()
    batch["rel_targets"] = [rel[i, f, : int(rel_len[i, f])] for i in range(bsz) for f in range(t)]
    batch["act_targets"] = [act[i, f, : int(act_len[i, f])] for i in range(bsz) for f in range(t)]
    batch["hg_mask"] = torch.cat([(act > 0), (rel > 0)], dim=2).float()
    tgt = torch.zeros(bsz, cfg.num_answers)
    tgt[torch.arange(bsz), torch.randint(0, cfg.num_answers, (bsz,), generator=g)] = 1.0
    batch["target"] = tgt
    return batch
