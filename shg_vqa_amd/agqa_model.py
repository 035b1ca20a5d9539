"""AGQAModel and the situation hyper-graph decoder block with the reference's interface
(AGQA/src/tasks/agqa_model.py:17-269)."""
import torch
import torch.nn as nn

from . import modeling as M
from . import ops
from .engine import engine
from .entry import BertTextEncoder, LXRTEncoder, rel_target_mask_device
from .transformer import TransformerDecoder, TransformerDecoderLayer

MAX_STAR_LENGTH = 40


def _head(hid, n_out):
    return nn.Sequential(nn.Linear(hid, hid * 2), M.GeLU(), M.BertLayerNorm(hid * 2, eps=1e-12), nn.Linear(hid * 2, n_out))


def _init_bert_weights(module):
    """agqa_model.py:152-163 (std hard-coded to 0.02)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, M.BertLayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class HGDecoder(nn.Module):
    """The block the task statement calls "HGDecoder" (agqa_model.py:220-260): query embeddings ->
    DETR decoder over the visual memory -> class heads, once for relations and once for actions,
    then the hyper-graph token sequence [B, T*(num_act+num_rel), 768] for the cross encoder.

    forward(memory [B,393,768], rel_segment_ids [B,128], act_segment_ids [B,48])
        -> rel_preds [B,128,n_rel+1], act_preds [B,48,n_act+1], hg_in [B,176,768]
    """

    def __init__(self, hid_dim, num_queries, act_queries, num_classes, num_actions, num_situations=16, num_rel=8,
                 num_act=3, dlayers=5, emb_drop_rate=0.15, decoder_drop_rate=0.15, gt_hg=False):
        super().__init__()
        self.hid_dim, self.num_situations, self.num_rel, self.num_act = hid_dim, num_situations, num_rel, num_act
        self.relation_query_embed = M.HGEmbeddings(num_queries=num_queries, type_vocab_size=16, hidden_size=hid_dim, gt_hg=gt_hg)
        self.action_query_embed = M.HGEmbeddings(num_queries=act_queries, type_vocab_size=16, hidden_size=hid_dim,
                                                 hidden_dropout_prob=emb_drop_rate, gt_hg=gt_hg)
        layer = TransformerDecoderLayer(d_model=hid_dim, nhead=12, dropout=decoder_drop_rate)
        self.rel_decoder = TransformerDecoder(layer, num_layers=dlayers)
        self.class_embed = _head(hid_dim, num_classes + 1)
        self.action_decoder = TransformerDecoder(layer, num_layers=dlayers)
        self.action_embed = _head(hid_dim, num_actions + 1)
        self.rel_decoder.apply(_init_bert_weights)
        self.action_decoder.apply(_init_bert_weights)

    def forward(self, memory, rel_segment_ids, act_segment_ids):
        B = memory.shape[0]
        dev = memory.device

        def decode(emb, dec, seg, per):
            qpos = emb(seg)
            mask = rel_target_mask_device(self.num_situations, per, dev)
            return dec.forward_bf(None, memory, qpos, mask)          # tgt = zeros (agqa_model.py:234)

        # the two decoders only share `memory`: the action decoder runs on a side stream beside the
        # relation decoder (both are chains of small launches that leave most of the GPU idle)
        branch = ops.Branch(1, memory, act_segment_ids)
        with branch:
            act_out = decode(self.action_query_embed, self.action_decoder, act_segment_ids, self.num_act)
        rel_out = decode(self.relation_query_embed, self.rel_decoder, rel_segment_ids, self.num_rel)
        branch.join(act_out)
        # Both prediction heads (and later the two set losses, agqa_hgqa.forward_losses) stay on the side stream:
        # the main chain goes straight on to the hyper-graph cross encoder, which only needs the decoder outputs.
        # In backward autograd replays them there as well, beside the cross encoder's backward.
        heads = ops.Branch(1, rel_out)
        with heads:
            act_preds = M.mlp_head(self.action_embed, act_out)
            rel_preds = M.mlp_head(self.class_embed, rel_out)
        if getattr(self, "defer_heads_join", False):    # set by AGQAModel.forward, which joins before it returns
            self.heads_branch = heads
        else:
            self.heads_branch = None
            heads.join(rel_preds, act_preds)
        T = self.num_situations
        hg_in = torch.cat([act_out.view(B, T, -1, self.hid_dim), rel_out.view(B, T, -1, self.hid_dim)], dim=2)
        return rel_preds, act_preds, hg_in.view(B, -1, self.hid_dim)


class _Passthrough(nn.Module):
    """Stands where the frozen video backbone sits (video_encoder.py:7-51): the hot path starts from
    precomputed (B,2048,16,7,7) features, so encode() is the identity."""

    def encode(self, x):
        return x


class AGQAModel(nn.Module):
    def __init__(self, num_answers, num_queries=128, num_classes=456, num_actions=156, model_name="", args=None):
        super().__init__()
        if args is None:
            from .param import hgqa_args
            args = hgqa_args()
        self.args = args
        object.__setattr__(self, "vid_encoder", _Passthrough())     # not registered: no parameters, no keys
        self.max_seq_length = MAX_STAR_LENGTH
        self.num_queries = num_queries
        self.act_queries = args.num_situations * args.num_act
        if args.task_q:
            self.bert_encoder = BertTextEncoder(args, max_seq_length=MAX_STAR_LENGTH, mode="lxr")
            hid = self.bert_encoder.dim
        elif args.task_vqa:
            self.lxrt_encoder = LXRTEncoder(args, max_seq_length=MAX_STAR_LENGTH)
            hid = self.lxrt_encoder.dim
        elif args.task_hgqa:
            self.lxrt_encoder = LXRTEncoder(args, max_seq_length=MAX_STAR_LENGTH, mode="lxr")
            config = self.lxrt_encoder.model.config
            self.hgq_encoder = M.CrossEncoder(config=config, cross_attn_type=args.cross_attn_type,
                                              num_max_act=args.num_act, num_max_rel=args.num_rel)
            hid = self.lxrt_encoder.dim
            hgd = HGDecoder(hid, self.num_queries, self.act_queries, num_classes, num_actions, args.num_situations,
                            args.num_rel, args.num_act, args.dlayers, args.emb_drop_rate, args.decoder_drop_rate)
            # the reference keeps these six modules directly on the model (state_dict keys!)
            self.relation_query_embed = hgd.relation_query_embed
            self.action_query_embed = hgd.action_query_embed
            self.rel_decoder = hgd.rel_decoder
            self.class_embed = hgd.class_embed
            self.action_decoder = hgd.action_decoder
            self.action_embed = hgd.action_embed
            object.__setattr__(self, "hg_decoder", hgd)
        else:
            raise NotImplementedError("task must be one of --taskQ / --taskVQA / --taskHGQA")
        self.hid_dim = hid
        self.logit_fc = _head(hid, num_answers)
        self.logit_fc.apply(_init_bert_weights)

    # ------------------------------------------------------------------ arena / mode plumbing
    def active_parameter_names(self):
        """Names (named_parameters) of the tensors that receive gradients for the configured task
        (under --taskHGQA the x-layers and their pooler get none)."""
        a = self.args
        names = []
        for n, _ in self.named_parameters():
            if ".cross_attn_layer.self." in n or ".cross_attn_layer.cross_self." in n or ".cross_attn_layer.old." in n:
                continue
            if ".pooler_dict." in n and ".pooler_dict.cross." not in n:
                continue
            if ".box_fc." in n or ".pos_layer_norm." in n:
                continue
            if a.task_hgqa and n.startswith("lxrt_encoder.") and (".cross_attn_layer." in n or ".pooler_dict." in n):
                continue
            if a.task_q and ".r_layers." in n:
                continue
            names.append(n)
        return set(names)

    def to_engine(self, compute_dtype=None):
        """Moves the parameters into the HBM arenas (engine.py).  Call once after construction/loading."""
        E = engine()
        if compute_dtype is not None:
            E.compute_dtype = compute_dtype
        groups = []
        for name, mod in self.named_modules():
            if isinstance(mod, M.BertAttention):
                groups += mod.fusion_groups(name + ".")
            for slot in ("_fz", "_ap", "_fp", "_fl", "_fv", "_sp", "_dp"):      # cached operand handles refer to arena offsets
                if slot in mod.__dict__:
                    mod.__dict__[slot] = None
        # shared modules are registered under several names: keep the groups whose names are canonical
        canon = {n for n, _ in self.named_parameters()}
        groups = [g for g in groups if all(n in canon for n in g)]
        E.adopt(self, self.active_parameter_names(), groups)
        # the step's first consumer of parameters (BertAdam.step keeps their update on the main stream)
        E.first_params = None
        if not self.args.task_q:
            c1 = self.lxrt_encoder.model.bert.encoder.visn_fc.conv[1]
            E.first_params = [c1.weight, c1.bias]
        return self

    def train(self, mode=True):
        super().train(mode)
        engine().training = bool(mode)
        return self

    # ------------------------------------------------------------------ forward (agqa_model.py:166-269)
    def forward(self, feat, pos, input_ids, input_masks, segment_ids, rel_segment_ids=None, rel_tgt_mask=None,
                act_segment_ids=None, act_tgt_mask=None, hg_mask=None, rel_tgt_ids=None, act_tgt_ids=None):
        a = self.args
        if a.task_q:
            engine().wait_params_ready()
            feats, x, attn = self.bert_encoder((input_ids, input_masks, segment_ids))
            return M.mlp_head(self.logit_fc, x), attn
        feat = self.vid_encoder.encode(feat)
        E = engine()
        # x-layers / pooler / answer head feed only `logit`; with the pre-cross-attention features going to the
        # decoders (the default) nothing on the HGQA loss path waits for them
        E.defer_x_layers = bool(a.task_hgqa and not a.after_cross_attn_feats)
        try:
            feats, x, attn = self.lxrt_encoder((input_ids, input_masks, segment_ids), (feat, pos))
        finally:
            E.defer_x_layers = False
        with ops.deferred_branch():
            logit = M.mlp_head(self.logit_fc, x)
        if a.task_vqa:
            return logit, attn
        if a.after_cross_attn_feats:
            lang_feats, memory = feats[0], feats[1]
            lang_mask = M.additive_mask(input_masks, input_ids)
        else:
            lang_feats, lang_mask, memory, _ = attn[-1]
        # rel_tgt_mask / act_tgt_mask arguments are ignored like in the reference (agqa_model.py:220, :241)
        self.hg_decoder.defer_heads_join = True
        try:
            rel_preds, act_preds, hg_in = self.hg_decoder(memory, rel_segment_ids, act_segment_ids)
        finally:
            self.hg_decoder.defer_heads_join = False
        ops.join_deferred_branch(lang_feats, logit)      # the language stream: question features + the deferred x-layers
        B = memory.shape[0]
        hgm = hg_mask.view(B, -1) if (a.use_hg_mask and hg_mask is not None) else None
        x, attn = self.hgq_encoder(lang_feats, lang_mask, hg_in, hgm)
        hg_logit = M.mlp_head(self.logit_fc, x)
        # the heads finished long ago (the cross encoder above is ~10x their work): this only orders later readers
        hb, self.hg_decoder.heads_branch = self.hg_decoder.heads_branch, None
        if hb is not None:
            hb.main = torch.cuda.current_stream()
            hb.join(rel_preds, act_preds)
        return logit, rel_preds, act_preds, hg_logit, attn
