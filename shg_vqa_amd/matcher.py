"""HungarianMatcher with the reference's interface (AGQA/src/lxrt/matcher.py:14-104), solved on
the GPU by shg_hungarian_per_frame: no cost matrix is materialised, nothing is copied to the host
and there is no scipy call on the product path."""
import torch
from torch import nn

from . import kernels as K


def pad_frame_targets(targets, per_frame, device):
    """list (B) of {"labels": [clip_len tensors]} -> (tgt [B*clip_len, per_frame] int64, len int32).
    A frame with MORE labels than query slots raises: SciPy would solve that rectangular problem (every query matched,
    surplus labels dropped), but the dataset never produces it (agqa_data.py truncates to num_rel / num_act per frame) and the
    per-frame kernel keeps one lane per label of at most per_frame labels."""
    flat = [t for d in targets for t in d["labels"]]
    tgt = torch.zeros((len(flat), per_frame), dtype=torch.int64)
    lens = torch.zeros(len(flat), dtype=torch.int32)
    for i, t in enumerate(flat):
        n = int(t.numel())
        if n > per_frame:
            raise ValueError("a frame has %d targets but only %d queries" % (n, per_frame))
        if n:
            tgt[i, :n] = t.detach().to("cpu", torch.int64).reshape(-1)
        lens[i] = n
    return tgt.to(device), lens.to(device)


def pad_clip_targets(targets, num_queries, device):
    """list (B) of {"labels": 1-D tensor of the clip's class ids} -> (tgt [B, num_queries] int64, len int32)."""
    tgt = torch.zeros((len(targets), num_queries), dtype=torch.int64)
    lens = torch.zeros(len(targets), dtype=torch.int32)
    for i, d in enumerate(targets):
        t = d["labels"].detach().to("cpu", torch.int64).reshape(-1)
        n = int(t.numel())
        if n > num_queries:
            raise ValueError("a clip has %d targets but only %d queries" % (n, num_queries))
        tgt[i, :n] = t
        lens[i] = n
    return tgt.to(device), lens.to(device)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, loss_hg_per_frame: bool = False, clip_len: int = 16):
        super().__init__()
        assert cost_class != 0, "cost cant be 0"
        self.cost_class = cost_class
        self.loss_hg_per_frame = loss_hg_per_frame
        self.clip_len = clip_len

    @torch.no_grad()
    def match_padded(self, pred_logits, tgt, tgt_len):
        """Device-resident fast path: pred_logits [B,Q,C]; tgt [B*clip_len, Q/clip_len] int64 (class ids,
        first tgt_len valid); tgt_len int32.  -> (query_idx, target_idx, grid) int64 [B*clip_len, per]."""
        b, q, c = pred_logits.shape
        # a positive cost_class scales all costs alike and cannot change the assignment
        if self.cost_class < 0:
            raise NotImplementedError("negative cost_class")
        if not self.loss_hg_per_frame:
            # per-clip branch (matcher.py:82-104): ONE problem per sample, num_queries x labels-of-the-clip;
            # tgt [B, Q] / tgt_len [B]; solved by the wave-cooperative kernel (up to 128 queries)
            return K.hungarian_per_frame(pred_logits.contiguous(), tgt, tgt_len)
        per = q // self.clip_len
        return K.hungarian_per_frame(pred_logits.contiguous().view(b * self.clip_len, per, c), tgt, tgt_len)

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference signature: returns a list of (index_i, index_j) int64 CPU tensors, one per frame
        (--LossHGPerFrame) or one per sample (per-clip matching)."""
        logits = outputs["pred_logits"]
        if not self.loss_hg_per_frame:
            tgt, lens = pad_clip_targets(targets, logits.shape[1], logits.device)
        else:
            per = logits.shape[1] // self.clip_len
            tgt, lens = pad_frame_targets(targets, per, logits.device)
        # the reference's out_prob[:, tgt_ids] (matcher.py:74, :91) raises IndexError for a class id outside the logits;
        # this entry point is off the hot path, so the check may synchronise (the kernel itself clamps the read)
        if tgt.numel() and (int(tgt.max()) >= logits.shape[-1] or int(tgt.min()) < 0):
            raise IndexError("target class id outside [0, %d)" % logits.shape[-1])
        oq, ot, _ = self.match_padded(logits, tgt, lens)
        oq, ot, lens = oq.cpu(), ot.cpu(), lens.cpu()
        return [(oq[i, : int(lens[i])].clone(), ot[i, : int(lens[i])].clone()) for i in range(oq.shape[0])]


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class)
