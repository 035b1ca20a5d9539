"""Interleaved A/B of tuning switches on the full training step in ONE process (same box, same clocks): for every setting
`name=value[,name=value...]` given on the command line, blocks of steps are timed in rotation and the mean ms/step printed next to
the baseline (all switches at their defaults).

    python tools/step_ab.py gemm8_tile_m=0 epilogue_side=0 "attn_nb_dq=2,attn_nb=1" py:wgrad_flush_tiles=200 [--steps 10] [--rounds 3]
(`py:<name>` sets an attribute of the host-side engine instead of a library switch; `step:overlap_update=1` passes the keyword to
AGQA.train_step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shg_vqa_amd import _lib
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.engine import engine, reset_engine
from shg_vqa_amd.param import hgqa_args

argv = sys.argv[1:]
steps = int(argv[argv.index("--steps") + 1]) if "--steps" in argv else 10
rounds = int(argv[argv.index("--rounds") + 1]) if "--rounds" in argv else 3
settings = [a for i, a in enumerate(argv) if "=" in a]
dev = torch.device("cuda", 0)
reset_engine(compute_dtype=torch.bfloat16, device=dev, seed=9595)
args = hgqa_args(compute_dtype="bf16", batch_size=32, lr=1e-5)
torch.manual_seed(9595)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000)
batches = bench.synthetic_device_batches(4, 32, 1234, dev)
if os.environ.get("SHG_MAIN_PRIO"):                       # the whole step on a stream of this priority (with SHG_STREAM_PRIO)
    print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
    _main = torch.cuda.Stream(device=dev, priority=int(os.environ["SHG_MAIN_PRIO"]))
    _main.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(_main)
for i in range(3):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
cases = [("baseline", {})] + [(s, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(","))) for s in settings]
E = engine()


step_kw = {}


def _get(k):                      # "py:<attr>" = an attribute of the engine (host-side switch), else a library tuning switch
    if k.startswith("step:"):
        return 0
    return getattr(E, k[3:]) if k.startswith("py:") else _lib.get_tuning(k)


def _set(k, v):
    if k.startswith("step:"):
        step_kw[k[5:]] = bool(v)
    elif k.startswith("py:"):
        setattr(E, k[3:], v)
    else:
        _lib.set_tuning(k, v)


defaults = {k: _get(k) for _, d in cases for k in d}
res = {n: [] for n, _ in cases}
for r in range(rounds):
    for name, d in cases:
        for k, v in defaults.items():
            _set(k, d.get(k, v))
        tr.train_step(batches[0], **step_kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train_step(batches[i % 4], **step_kw)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
for k, v in defaults.items():
    _set(k, v)
base = sum(res["baseline"]) / rounds
for name, _ in cases:
    t = sum(res[name]) / rounds
    print("%-40s %.3f ms/step  (%+.3f)   %s" % (name, t, t - base, ", ".join("%.2f" % x for x in res[name])))
