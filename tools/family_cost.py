"""What a kernel family costs IN the training step (beside the other streams' kernels), measured by issuing every launch of the
family twice ("repeat_family" tuning switch: all families listed are idempotent, the results of the step do not change) and
timing interleaved blocks of steps in one process: added ms per step next to the family's solo kernel time per step.

    python tools/family_cost.py [steps per block] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shg_vqa_amd import _lib
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.engine import engine, reset_engine
from shg_vqa_amd.param import hgqa_args

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
reset_engine(compute_dtype=torch.bfloat16, device=dev, seed=9595)
args = hgqa_args(compute_dtype="bf16", batch_size=32, lr=1e-5)
torch.manual_seed(9595)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000)
batches = bench.synthetic_device_batches(4, 32, 1234, dev)
for i in range(3):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
fams = [(0, "baseline"), (1, "attention forward"), (2, "attention backward"), (4, "LayerNorm forward"), (8, "LayerNorm backward"),
        (16, "GEMMs >= 120 tiles of 256 x 256 (non-accumulating)"), (32, "smaller GEMMs (non-accumulating)"), (64, "convolution forward"),
        (128, "grouped weight gradients"), (256, "convolution weight gradients"), (512, "convolution input gradient"),
        (1024, "GEMM + activation backward (dact)"), (2048, "accumulating bf16 GEMMs (C += A.B)"), (4096, "bias column sums")]
res = {m: [] for m, _ in fams}
for r in range(rounds):
    for m, _ in fams:
        _lib.set_tuning("repeat_family", m)
        tr.train_step(batches[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train_step(batches[i % 4])
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / steps * 1e3)
_lib.set_tuning("repeat_family", 0)
base = sum(res[0]) / len(res[0])
print("baseline %.3f ms/step (%s)" % (base, ", ".join("%.2f" % x for x in res[0])))
for m, name in fams[1:]:
    t = sum(res[m]) / len(res[m])
    print("%-52s +%.3f ms/step when issued twice   (%s)" % (name, t - base, ", ".join("%.2f" % x for x in res[m])))
