"""Reference import harness (TEST INFRASTRUCTURE, runs only in the build container).

Imports the reference's Python hot path from /root/reference/AGQA on CPU so that
golden vectors can be generated (oracle/gen_golden.py).  Nothing here is used by the
product path and nothing here travels to the GPU box in a usable form: /root/reference
does not exist there, so `load()` raises.

What it does:
  * third-party modules that are absent offline and never touch the path's arithmetic
    (timm, boto3, cv2, torchvision, pytorchvideo ...) become empty placeholder modules;
  * every loader that would fetch from the network is rebound to a function that raises;
  * `from_pretrained` builds the model locally from BertConfig defaults (the reference
    re-initialises all weights under --fromScratch anyway, entry.py:170-172);
  * the frozen video backbone is replaced by a pass-through, so `feat` is the
    (B,2048,16,7,7) slow_r50-shaped tensor the hot path starts from.
"""
import os
import sys
import types

REF_ROOT = "/root/reference/AGQA"

HGQA_ARGV = ["ref", "--llayers", "5", "--xlayers", "2", "--rlayers", "5", "--noCaps",
             "--crossAttnType", "cross", "--batchSize", "2", "--taskHGQA", "--fromScratch",
             "--LossHGPerFrame", "--backbone", "slow_r50", "--optim", "bert", "--lr", "1e-5"]


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


def _placeholder(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _blocked(*a, **k):
    raise RuntimeError("network fetch attempted inside the oracle harness")


def load(argv=None):
    """Returns a namespace with the reference modules (mc, entry, matcher, transformer,
    optimization, agqa_model, agqa_hgqa)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    for n in ["timm", "boto3", "botocore", "ffmpeg", "h5py", "ipyplot", "matplotlib"]:
        _placeholder(n)
    _placeholder("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
    _placeholder("matplotlib.pyplot")
    cv2 = _placeholder("cv2")
    cv2.cv2 = cv2
    tv = _placeholder("torchvision")
    tv.models = _placeholder("torchvision.models", resnext101_32x8d=None)
    tv.transforms = _placeholder("torchvision.transforms", Compose=_Inert, Lambda=_Inert, Resize=_Inert)
    _placeholder("pytorchvideo")
    _placeholder("pytorchvideo.transforms", **{k: _Inert for k in [
        "ApplyTransformToKey", "ShortSideScale", "UniformTemporalSubsample", "UniformCropVideo",
        "Normalize", "AugMix", "RandAugment", "Permute"]})
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.argv = list(argv or HGQA_ARGV)

    import torch
    import torch.nn as nn
    import src.lxrt.file_utils as fu
    fu.cached_path = fu.http_get = fu.get_from_cache = _blocked
    import src.lxrt.modeling_capsbert as mc
    import src.lxrt.tokenization as tok
    tok.cached_path = mc.cached_path = _blocked
    torch.hub.load = _blocked
    mc.BertPreTrainedModel.from_pretrained = classmethod(
        lambda cls, name, *a, **kw: cls(mc.BertConfig(30522), *a, **kw))
    tok.BertTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: None)
    import src.video_encoder as ve

    class PassThroughBackbone(nn.Module):
        def __init__(self, name):
            super().__init__()

        def encode(self, x):
            return x

    ve.VideoBackbone = PassThroughBackbone
    import src.tasks.agqa_model as am
    am.VideoBackbone = PassThroughBackbone
    import src.tasks.agqaHGQA as hg
    import src.lxrt.entry as entry
    import src.lxrt.matcher as matcher
    import src.lxrt.transformer as transformer
    import src.lxrt.optimization as optimization
    return types.SimpleNamespace(mc=mc, entry=entry, matcher=matcher, transformer=transformer,
                                 optimization=optimization, agqa_model=am, agqa_hgqa=hg, torch=torch)
