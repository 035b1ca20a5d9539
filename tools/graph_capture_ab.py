"""Whole-step hipGraph capture vs eager steps under the stream switches (SHG_OVERLAP_BRANCHES / SHG_OVERLAP_WGRAD):
prints the losses of the test's step sequence for both execution modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_model_gpu as T
from shg_vqa_amd.engine import engine
from oracle import shg_ref
from shg_vqa_amd.transformer import MultiheadAttention

cfg = shg_ref.Cfg()
for graphed in (False, True):
    tr = T._build(torch.bfloat16)
    batches = [T._device_batch(shg_ref.synthetic_batch(2, cfg, seed=90 + i)) for i in range(3)]
    for m in tr.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, MultiheadAttention):
            m.dropout = 0.0
    o = tr.train_step(batches[0])
    print("first", {k: round(float(o[k]), 4) for k in ("total", "bce", "rel_ce", "act_ce", "grad_norm")}, flush=True)
    if graphed:
        tr.capture(batches[0])
        out = []
        for b in batches:
            o = tr.train_step_graphed(b)
            torch.cuda.synchronize()
            print("   ", {k: round(float(o[k]), 4) for k in ("total", "bce", "rel_ce", "act_ce", "grad_norm")}, flush=True)
            out.append(float(o["total"]))
        out.append(float(tr.train_step(batches[0])["total"]))
    else:
        for _ in range(2):
            o = tr.train_step(batches[0])
            print("warm ", {k: round(float(o[k]), 4) for k in ("total", "bce", "rel_ce", "act_ce", "grad_norm")}, flush=True)
        out = []
        for b in batches + batches[:1]:
            o = tr.train_step(b)
            torch.cuda.synchronize()
            print("   ", {k: round(float(o[k]), 4) for k in ("total", "bce", "rel_ce", "act_ce", "grad_norm")}, flush=True)
            out.append(float(o["total"]))
    torch.cuda.synchronize()
    print("graphed" if graphed else "eager  ", ["%.4f" % x for x in out], flush=True)
