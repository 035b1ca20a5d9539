"""Precomputed video-feature cache and its loader: the step BEFORE the hot path.

The reference runs the frozen slow_r50 backbone inside every training step (video_encoder.py:43-51,
agqa_model.py:197) and hands the conv stack a (B, 2048, 16, 7, 7) fp32 NCDHW tensor.  With frozen
weights those features are a pure function of the clip, so they are computed once and cached in the
layout and precision the first convolution consumes:

    <name>.feat   raw little-endian bf16, [N, T=16, H=7, W=7, C=2048]  (channels-last, 3.2 MB per clip
                  instead of 6.4 MB fp32; no NCDHW -> channels-last transpose left for the GPU)
    <name>.json   {"n", "shape": [T, H, W, C], "dtype": "bf16", "layout": "NTHWC", "ids": [...]}

`FeatureCache` memory-maps the file (nothing is read until a clip is touched; the OS page cache is the
only host copy).  `PrefetchLoader` gathers each batch into one of two pinned staging buffers and copies it
to HBM on its own stream while the previous batch trains: the consumer waits on an event, never on the
host.  The device tensor is what `ops.conv1_forward` accepts directly (channels-last bf16).
"""
import json
import os

import numpy as np
import torch

_LAYOUT = "NTHWC"


def _bf16_bits(x32):
    """fp32 tensor -> uint16 numpy array holding the round-to-nearest-even bf16 bit patterns."""
    return x32.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def write_feature_cache(prefix, feats, ids=None):
    """feats: iterable of (C, T, H, W) fp32 tensors (one clip each, the backbone's NCDHW output without
    the batch dimension).  Streams them to <prefix>.feat / <prefix>.json; returns the number of clips."""
    n, shape, all_ids = 0, None, []
    with open(prefix + ".feat", "wb") as f:
        for i, x in enumerate(feats):
            if x.dim() != 4:
                raise ValueError("each clip must be (C, T, H, W)")
            cl = x.permute(1, 2, 3, 0)                       # (T, H, W, C)
            if shape is None:
                shape = list(cl.shape)
            elif list(cl.shape) != shape:
                raise ValueError("clip %d has shape %s, expected %s" % (i, list(cl.shape), shape))
            f.write(_bf16_bits(cl.float()).tobytes())
            all_ids.append(ids[i] if ids is not None else i)
            n += 1
    with open(prefix + ".json", "w") as f:
        json.dump({"n": n, "shape": shape, "dtype": "bf16", "layout": _LAYOUT, "ids": all_ids}, f)
    return n


class FeatureCache:
    """Read side: clip i as a (T, H, W, C) bf16 tensor (a copy of the mapped pages)."""

    def __init__(self, prefix):
        with open(prefix + ".json") as f:
            meta = json.load(f)
        if meta.get("dtype") != "bf16" or meta.get("layout") != _LAYOUT:
            raise ValueError("unsupported feature cache: %s" % {k: meta.get(k) for k in ("dtype", "layout")})
        self.n, self.shape, self.ids = int(meta["n"]), tuple(meta["shape"]), meta["ids"]
        self.index = {v: i for i, v in enumerate(self.ids)}
        per = int(np.prod(self.shape))
        want = self.n * per * 2
        have = os.path.getsize(prefix + ".feat")
        if have != want:
            raise ValueError("%s.feat holds %d bytes, the header describes %d" % (prefix, have, want))
        self._map = np.memmap(prefix + ".feat", dtype=np.uint16, mode="r", shape=(self.n,) + self.shape)

    def __len__(self):
        return self.n

    def bits(self, i):
        return self._map[i]

    def __getitem__(self, i):
        return torch.from_numpy(np.array(self._map[i]).view(np.int16)).view(torch.bfloat16)

    def gather_into(self, out_u16, indices):
        """Copies the clips `indices` into the numpy uint16 view of a (pinned) staging buffer."""
        for j, i in enumerate(indices):
            out_u16[j] = self._map[i]


class PrefetchLoader:
    """Iterates device batches [B, T, H, W, C] bf16 over `batches` (lists of clip indices), one batch ahead.

    Two pinned staging buffers; the host->device copy of batch k+1 is issued on a side stream before batch k
    is handed out.  A staging buffer is only refilled after the copy that read it has completed (event), and a
    device buffer is only overwritten after the consumer's stream has passed the point where it was handed the
    NEXT batch (the consumer's work on the old one is ordered before that by stream order)."""

    def __init__(self, cache, batches, device="cuda", depth=2):
        self.cache, self.batches, self.device = cache, [list(b) for b in batches], torch.device(device)
        self.depth = depth
        bmax = max((len(b) for b in self.batches), default=0)
        shape = (bmax,) + cache.shape
        cuda = self.device.type == "cuda"
        self._pinned = [torch.empty(shape, dtype=torch.bfloat16, pin_memory=cuda) for _ in range(depth)]
        self._views = [p.view(torch.int16).numpy().view(np.uint16) for p in self._pinned]
        self._dev = [torch.empty(shape, dtype=torch.bfloat16, device=self.device) for _ in range(depth)] if cuda else None
        self._stream = torch.cuda.Stream(device=self.device) if cuda else None
        self._copied = [None] * depth          # event: H2D copy out of staging buffer s finished
        self._released = [None] * depth        # event: the consumer is done with device buffer s

    def _issue(self, k):
        s = k % self.depth
        idx = self.batches[k]
        if self._copied[s] is not None:
            self._copied[s].synchronize()                      # the staging buffer is free again
        self.cache.gather_into(self._views[s], idx)
        if self._dev is None:
            return self._pinned[s][:len(idx)].clone()
        with torch.cuda.stream(self._stream):
            if self._released[s] is not None:
                self._stream.wait_event(self._released[s])     # consumer finished with this device buffer
            self._dev[s][:len(idx)].copy_(self._pinned[s][:len(idx)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._copied[s] = ev
        return self._dev[s][:len(idx)]

    def __iter__(self):
        if not self.batches:
            return
        nxt = self._issue(0)
        for k in range(len(self.batches)):
            cur, s = nxt, k % self.depth
            if k + 1 < len(self.batches):
                nxt = self._issue(k + 1)
            if self._dev is not None:
                torch.cuda.current_stream().wait_event(self._copied[s])
            yield cur
            if self._dev is not None:                           # everything the consumer enqueued so far used `cur`
                rel = torch.cuda.Event()
                rel.record(torch.cuda.current_stream())
                self._released[s] = rel
