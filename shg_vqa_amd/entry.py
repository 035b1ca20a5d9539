"""Encoder facades and host-side feature converters with the reference's interface
(AGQA/src/lxrt/entry.py:38-445)."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import modeling as M


class InputFeatures:
    """entry.py:28-35."""

    def __init__(self, input_ids, input_mask, segment_ids, targets=None):
        self.input_ids, self.input_mask, self.segment_ids, self.targets = input_ids, input_mask, segment_ids, targets


class HashTokenizer:
    """Stand-in for BertTokenizer when no vocabulary file is available offline (the reference
    downloads bert-base-uncased's vocab, tokenization.py:26-34): lower-cases, splits on non-alphanumerics
    and hashes each token into the non-special id range.  Deterministic; for synthetic runs only."""

    def tokenize(self, text):
        out, cur = [], ""
        for ch in text.lower():
            if ch.isalnum():
                cur += ch
            else:
                if cur:
                    out.append(cur)
                cur = ""
                if not ch.isspace():
                    out.append(ch)
        if cur:
            out.append(cur)
        return out

    def convert_tokens_to_ids(self, tokens):
        ids = []
        for t in tokens:
            if t == "[CLS]":
                ids.append(101)
            elif t == "[SEP]":
                ids.append(102)
            else:
                h = 0
                for ch in t:
                    h = (h * 131 + ord(ch)) % 29522
                ids.append(1000 + h)
        return ids


def convert_sents_to_features(sents, max_seq_length, tokenizer):
    """entry.py:38-73: [CLS] tokens [SEP], zero padded to max_seq_length."""
    feats = []
    for sent in sents:
        toks = tokenizer.tokenize(sent.strip())[: max_seq_length - 2]
        toks = ["[CLS]"] + toks + ["[SEP]"]
        ids = tokenizer.convert_tokens_to_ids(toks)
        pad = [0] * (max_seq_length - len(ids))
        feats.append(InputFeatures(ids + pad, [1] * len(ids) + pad, [0] * max_seq_length))
    return feats


def convert_relations_to_features(rel_trplts_tokens, num_rel=8, num_situations=16, lengths=[], loss_hg_per_frame=False):
    """entry.py:76-97: flattened ids, frame-id segment ids and the ragged per-frame targets."""
    feats = []
    seg = np.repeat(np.arange(num_situations), num_rel)
    lengths = lengths.tolist() if hasattr(lengths, "tolist") else lengths
    for i, trip in enumerate(rel_trplts_tokens):
        tg = [trip[j, : lengths[i][j]] for j in range(num_situations)]
        if not loss_hg_per_frame:
            tg = [int(x) for fr in tg for x in fr]
        feats.append(InputFeatures(np.array(trip.reshape(-1)), None, seg.copy(), tg))
    return feats


def convert_relations_to_features_test(rel_trplts_tokens, num_rel=8, num_situations=16, lengths=[], bsize=8):
    """entry.py:99-112."""
    seg = np.repeat(np.arange(num_situations), num_rel)
    return [InputFeatures(None, None, seg.copy(), None) for _ in range(bsize)]


_SEGMENT_IDS = {}


def frame_segment_ids(batch, num_situations, per_frame, device):
    """Vectorised form of the segment ids above: [B, num_situations * per_frame] int64 on `device`.  A constant of the shape:
    built once per (shape, device) - three tiny kernels per call otherwise, at the very start of every step (read-only)."""
    key = (int(batch), int(num_situations), int(per_frame), str(device))
    capturing = torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()
    t = None if capturing else _SEGMENT_IDS.get(key)            # (a captured step builds its own: graph-private memory)
    if t is None:
        t = torch.arange(num_situations, device=device).repeat_interleave(per_frame).unsqueeze(0).expand(batch, -1).contiguous()
        if not capturing:
            _SEGMENT_IDS[key] = t
    return t


def clip_targets_device(triplets, lengths):
    """Per-clip matching targets (convert_relations_to_features with loss_hg_per_frame=False, entry.py:87-89) built on
    the device: the valid class ids of all frames, in frame order, left-packed.
    triplets [B, T, per] int64, lengths [B, T] -> (tgt [B, T*per] int64, tgt_len [B] int32)."""
    B, T, per = triplets.shape
    valid = (torch.arange(per, device=triplets.device).view(1, 1, per) < lengths.view(B, T, 1)).view(B, -1)
    flat = triplets.reshape(B, -1)
    pos = valid.cumsum(1) - 1
    out = torch.zeros((B, T * per + 1), dtype=torch.int64, device=triplets.device)
    out.scatter_(1, torch.where(valid, pos, torch.full_like(pos, T * per)), flat)     # invalid slots land in the spare column
    return out[:, :T * per].contiguous(), valid.sum(1).to(torch.int32)


_MASK_CACHE = {}


def generate_rel_target_mask(num_situations, num_rel):
    """entry.py:114-121: additive block-causal mask (numpy, -inf above the frame diagonal)."""
    fr = np.repeat(np.arange(num_situations), num_rel)
    m = np.zeros((fr.size, fr.size), dtype=np.float32)
    m[fr[None, :] > fr[:, None]] = -np.inf
    return m


def rel_target_mask_device(num_situations, num_rel, device):
    """The same mask, built once per (shape, device) and kept in HBM (the reference rebuilds it on the
    host and copies it on every forward, agqa_model.py:220, :241)."""
    key = (num_situations, num_rel, str(device))
    if key not in _MASK_CACHE:
        _MASK_CACHE[key] = torch.from_numpy(generate_rel_target_mask(num_situations, num_rel)).to(device)
    return _MASK_CACHE[key]


def set_visual_config(args):
    """entry.py:124-136."""
    M.VISUAL_CONFIG.l_layers = args.llayers
    M.VISUAL_CONFIG.x_layers = args.xlayers
    M.VISUAL_CONFIG.r_layers = args.rlayers


class _EncoderBase(nn.Module):
    @property
    def dim(self):
        return 768

    def save(self, path):
        torch.save(self.model.state_dict(), os.path.join("%s_LXRT.pth" % path))

    def load(self, path):
        """entry.py:207-238: strips `module.` / `lxrt_encoder.model.` prefixes, loads non-strictly."""
        sd = torch.load("%s_LXRT.pth" % path, map_location="cpu")
        new = {}
        for k, v in sd.items():
            k2 = k[len("module."):] if k.startswith("module.") else k
            new[k2] = v
            if k.startswith("lxrt_encoder.model."):
                new[k[len("lxrt_encoder.model."):]] = v
        self.model.load_state_dict(new, strict=False)


class LXRTEncoder(_EncoderBase):
    """entry.py:139-238.  mode 'x' -> (None, pooled, attn), mode 'lxr' -> ((lang, visn), pooled, attn)."""

    def __init__(self, args, max_seq_length, mode="x"):
        super().__init__()
        self.max_seq_length = max_seq_length
        set_visual_config(args)
        self.args = args
        self.mode = mode
        self.tokenizer = HashTokenizer()
        cross = getattr(args, "cross_attn_type", "old")
        self.model = M.LXRTFeatureExtraction.from_pretrained("bert-base-uncased", mode=mode, cross_attn_type=cross,
                                                             no_caps=args.no_caps)
        if args.from_scratch:
            self.model.apply(self.model.init_bert_weights)

    def forward(self, sents, feats, visual_attention_mask=None):
        input_ids, input_mask, segment_ids = sents[0], sents[1], sents[2]
        out_attn = getattr(self.args, "output_attention", False)
        if self.mode == "lxr":
            feat, output, attn = self.model(input_ids, segment_ids, input_mask, visual_feats=feats,
                                            visual_attention_mask=visual_attention_mask,
                                            output_all_attention_masks=out_attn)
        else:
            feat = None
            output, attn = self.model(input_ids, segment_ids, input_mask, visual_feats=feats,
                                      visual_attention_mask=visual_attention_mask,
                                      output_all_attention_masks=out_attn)
        return feat, output, attn


class BertTextEncoder(_EncoderBase):
    """entry.py:248-303 (question-only model)."""

    def __init__(self, args, max_seq_length, mode="x"):
        super().__init__()
        self.max_seq_length = max_seq_length
        set_visual_config(args)
        self.args = args
        self.mode = mode
        self.tokenizer = HashTokenizer()
        self.model = M.BertFeatureExtraction.from_pretrained("bert-base-uncased", mode=mode, no_caps=args.no_caps)
        if args.from_scratch:
            self.model.apply(self.model.init_bert_weights)

    def forward(self, sents):
        input_ids, input_mask, segment_ids = sents[0], sents[1], sents[2]
        return self.model(input_ids, segment_ids, input_mask)
