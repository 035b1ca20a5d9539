"""Runs only bench.attention_stack (the encoder's language / relation / cross layers, forward + backward) so that a
rocprofv3 --kernel-trace of this script is the kernel list of that sub-benchmark (ITERS + 2 warm-up passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.engine import reset_engine
from shg_vqa_amd.param import hgqa_args

dev = torch.device("cuda", 0)
reset_engine(compute_dtype=torch.bfloat16, device=dev)
args = hgqa_args(compute_dtype="bf16", batch_size=32)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000)
torch.cuda.synchronize()
print("MARK setup done", flush=True)
print(bench.attention_stack(tr, 32, iters=int(os.environ.get("ITERS", "6"))), flush=True)
