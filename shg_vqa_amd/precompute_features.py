"""Fills the video-feature cache of feature_cache.py: the frozen backbone run ONCE per clip instead of inside every
training step (the reference calls `self.vid_encoder.encode(video)` under no_grad in every forward, agqa_model.py:197 with
VideoBackbone.encode, video_encoder.py:28-37).

    python -m shg_vqa_amd.precompute_features --backbone pkg.module:factory --clips DIR --out PREFIX [--batch 8] [--device cuda]
    python -m shg_vqa_amd.precompute_features --features DIR --out PREFIX

--backbone names a zero-argument factory returning the frozen network: anything with `.encode(x)` (the reference's
VideoBackbone) or a plain callable, mapping clips [B, 3, T, H, W] to features [B, C, T', H', W'] (slow_r50: 2048 x 16 x 7 x 7).
DIR holds one `<clip id>.pt` / `.npy` per clip: transformed frames (3, T, H, W) for --clips, backbone outputs (C, T', H', W')
for --features (e.g. dumped once from the reference pipeline).  The pretrained weights themselves are not part of this
repository (torch.hub needs the network); the tool only assumes the factory can build them.
"""
import argparse
import importlib
import os

import numpy as np
import torch

from .feature_cache import write_feature_cache


def _load(path):
    if path.endswith(".npy"):
        return torch.from_numpy(np.load(path))
    return torch.load(path, map_location="cpu")


def list_clips(directory):
    """[(clip id, path)] sorted by id: every .pt / .npy file of the directory."""
    out = []
    for f in sorted(os.listdir(directory)):
        stem, ext = os.path.splitext(f)
        if ext in (".pt", ".npy"):
            out.append((stem, os.path.join(directory, f)))
    return out


def encode_clips(backbone, clips, batch_size=8, device="cpu"):
    """Generator of per-clip features (C, T', H', W') fp32 on the host: `clips` is an iterable of (3, T, H, W) tensors,
    run through the frozen backbone in batches under no_grad (eval mode if it is a module)."""
    enc = backbone.encode if hasattr(backbone, "encode") else backbone
    if isinstance(backbone, torch.nn.Module):
        backbone.eval().to(device)
    batch = []

    def flush():
        with torch.no_grad():
            y = enc(torch.stack(batch).to(device))
        for row in y.float().cpu():
            yield row

    for c in clips:
        batch.append(c.float())
        if len(batch) == batch_size:
            yield from flush()
            batch = []
    if batch:
        yield from flush()


def precompute(backbone, clip_dir, prefix, batch_size=8, device="cpu"):
    """clip_dir/<id>.pt (frames) -> <prefix>.feat / .json; returns the number of clips."""
    items = list_clips(clip_dir)
    feats = encode_clips(backbone, (_load(p) for _, p in items), batch_size, device)
    return write_feature_cache(prefix, feats, ids=[i for i, _ in items])


def convert(feature_dir, prefix):
    """feature_dir/<id>.pt (backbone outputs, (C, T', H', W')) -> <prefix>.feat / .json."""
    items = list_clips(feature_dir)
    return write_feature_cache(prefix, (_load(p).float() for _, p in items), ids=[i for i, _ in items])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--backbone", help="module:factory returning the frozen backbone")
    ap.add_argument("--clips", help="directory of <id>.pt / .npy frame tensors (3, T, H, W)")
    ap.add_argument("--features", help="directory of <id>.pt / .npy backbone outputs (C, T, H, W)")
    ap.add_argument("--out", required=True, help="cache prefix (<out>.feat, <out>.json)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args(argv)
    if (a.features is None) == (a.clips is None):
        ap.error("give exactly one of --clips (with --backbone) and --features")
    if a.features is not None:
        n = convert(a.features, a.out)
    else:
        if not a.backbone or ":" not in a.backbone:
            ap.error("--clips needs --backbone module:factory")
        mod, attr = a.backbone.split(":", 1)
        n = precompute(getattr(importlib.import_module(mod), attr)(), a.clips, a.out, a.batch, a.device)
    print("wrote %d clips to %s.feat" % (n, a.out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
