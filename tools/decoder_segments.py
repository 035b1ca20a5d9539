"""Profiler-free timing of the relation decoder's sub-layers on the main stream (forward), steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shg_vqa_amd import ops
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.engine import engine, reset_engine
from shg_vqa_amd.param import hgqa_args

dev = torch.device("cuda", 0)
reset_engine(compute_dtype=torch.bfloat16, device=dev)
args = hgqa_args(compute_dtype="bf16", batch_size=32)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000)
batches = bench.synthetic_device_batches(2, 32, 1234, dev)
marks = []
main = torch.cuda.current_stream()


def mark(name):
    if torch.cuda.current_stream() != main:
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, ev))


orig_attn, orig_ffn = ops.attn_sublayer, ops.ffn_sublayer


def attn(x, pos, mem, P, *a, **k):
    y = orig_attn(x, pos, mem, P, *a, **k)
    mark("attn " + P.mode + " Sq=%d" % x.shape[1])
    return y


def ffn(x, P):
    y = orig_ffn(x, P)
    mark("ffn rows=%d F=%d" % (x.shape[0] * x.shape[1], P.w1.shape[0]))
    return y


ops.attn_sublayer, ops.ffn_sublayer = attn, ffn
N = 8
allm = []
for it in range(N):
    marks.clear()
    mark("start")
    tr.train_step(batches[it % 2])
    allm.append(list(marks))
torch.cuda.synchronize()
import collections
acc = collections.OrderedDict()
for ms in allm[2:]:
    prev = ms[0][1]
    for i, (name, ev) in enumerate(ms[1:]):
        key = (i, name)
        acc.setdefault(key, 0.0)
        acc[key] += prev.elapsed_time(ev) / (N - 2)
        prev = ev
for (i, name), t in acc.items():
    print("%3d %-34s %8.3f ms" % (i, name, t))
