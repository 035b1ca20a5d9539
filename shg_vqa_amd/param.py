"""Flags of the training scripts, with the reference's names and defaults (AGQA/src/param.py:33-201).
Unlike the reference nothing is parsed at import: call parse_args(argv)."""
import argparse


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--train", default="train")
    p.add_argument("--valid", default="valid")
    p.add_argument("--test", default=None)
    p.add_argument("--batchSize", dest="batch_size", type=int, default=32)
    p.add_argument("--optim", default="bert")
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--dropout", type=float, default=0.3)
    p.add_argument("--seed", type=int, default=9595)
    p.add_argument("--output", type=str, default="snap/agqa")
    p.add_argument("--tqdm", action="store_const", default=False, const=True)
    p.add_argument("--load", type=str, default=None)
    p.add_argument("--loadLXMERT", dest="load_lxmert", type=str, default=None)
    p.add_argument("--fromScratch", dest="from_scratch", action="store_const", default=False, const=True)
    p.add_argument("--logFreq", dest="log_freq", type=int, default=50)
    p.add_argument("--llayers", default=5, type=int)
    p.add_argument("--xlayers", default=2, type=int)
    p.add_argument("--rlayers", default=5, type=int)
    p.add_argument("--dlayers", default=5, type=int)
    p.add_argument("--crossAttnType", dest="cross_attn_type", default="cross", type=str)
    p.add_argument("--noCaps", dest="no_caps", action="store_const", default=False, const=True)
    p.add_argument("--numRel", dest="num_rel", default=8, type=int)
    p.add_argument("--numAct", dest="num_act", default=3, type=int)
    p.add_argument("--numSituations", dest="num_situations", default=16, type=int)
    p.add_argument("--clipLEN", dest="CLIP_LEN", default=16, type=int)
    p.add_argument("--embDropRate", dest="emb_drop_rate", default=0.15, type=float)
    p.add_argument("--decoderDropRate", dest="decoder_drop_rate", default=0.15, type=float)
    p.add_argument("--taskQ", dest="task_q", action="store_const", default=False, const=True)
    p.add_argument("--taskVQA", dest="task_vqa", action="store_const", default=False, const=True)
    p.add_argument("--taskHGQA", dest="task_hgqa", action="store_const", default=False, const=True)
    p.add_argument("--taskVHGA", dest="task_vhga", action="store_const", default=False, const=True)
    p.add_argument("--gtHG", dest="gt_hg", action="store_const", default=False, const=True)
    p.add_argument("--useHGMask", dest="use_hg_mask", action="store_const", default=False, const=True)
    p.add_argument("--LossHGPerFrame", dest="loss_hg_per_frame", action="store_const", default=False, const=True)
    p.add_argument("--linearCls", dest="linear_cls", action="store_const", default=False, const=True)
    p.add_argument("--afterCrossAttnFeats", dest="after_cross_attn_feats", action="store_const", default=False, const=True)
    p.add_argument("--outputAttn", dest="output_attention", action="store_const", default=False, const=True)
    p.add_argument("--backbone", type=str, default="slow_r50")
    p.add_argument("--multiGPU", action="store_const", default=False, const=True)
    p.add_argument("--numWorkers", dest="num_workers", default=8, type=int)
    p.add_argument("--novelComp", dest="novel_comp", action="store_const", default=False, const=True)
    p.add_argument("--indirectRef", dest="indirect_ref", action="store_const", default=False, const=True)
    p.add_argument("--compSteps", dest="comp_steps", action="store_const", default=False, const=True)
    # additions of this build
    p.add_argument("--computeDtype", dest="compute_dtype", default="bf16", choices=["bf16", "fp32"])
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    for k in ("skip_connection", "shared_weights", "cross_attn", "freeze_weights", "patches", "vit_init"):
        setattr(args, k, False)
    args.margin, args.start_index = 0.1, 7
    return args


def hgqa_args(**over):
    """The headline configuration of BASELINE.json: 5/2/5 layers, --taskHGQA --LossHGPerFrame."""
    a = parse_args(["--noCaps", "--crossAttnType", "cross", "--taskHGQA", "--fromScratch", "--LossHGPerFrame"])
    for k, v in over.items():
        setattr(a, k, v)
    return a
