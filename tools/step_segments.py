"""Where does the step's wall time go on the MAIN stream?  Events at segment boundaries (forward: module hooks;
backward: gradient hooks on the tensors between the segments), read after the step; no profiler, so the host runs at
full speed and side-stream overlap is as in the benchmark."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.engine import engine, reset_engine
from shg_vqa_amd.param import hgqa_args

dev = torch.device("cuda", 0)
reset_engine(compute_dtype=torch.bfloat16, device=dev)
args = hgqa_args(compute_dtype="bf16", batch_size=32)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
red = None
if os.environ.get("SEG_DDP"):                      # forced single-rank collectives through the gradient reducer (RCCL)
    import torch.distributed as dist
    from shg_vqa_amd.ddp import GradReducer
    os.dup2(2, 1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
    red = GradReducer(engine().grad_arena, force_collectives=os.environ["SEG_DDP"] != "hooks")
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000, world=red)
batches = bench.synthetic_device_batches(2, 32, 1234, dev)
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()                       # on the current stream of the calling thread
    marks.append((name, ev, time.perf_counter()))


enc = model.lxrt_encoder.model.bert.encoder


def fwd_hook(name):
    def h(mod, inp, out):
        mark("fwd " + name)
    return h


def bwd_mark(t, name):
    if torch.is_tensor(t) and t.requires_grad:
        t.register_hook(lambda g: (mark("bwd " + name), None)[1])


enc.visn_fc.register_forward_hook(lambda m, i, o: (mark("fwd conv stack"), bwd_mark(o[0], "r-layers done (grad of conv tokens)"))[0])
enc.r_layers[-1].register_forward_hook(lambda m, i, o: (mark("fwd r-layers"), bwd_mark(o[0], "decoders+hg done (grad of memory)"))[0])
model.rel_decoder.register_forward_hook(lambda m, i, o: (mark("fwd rel decoder"), bwd_mark(o, "hg encoder bwd done (grad of rel decoder out)"))[0]) if False else None
orig_dec = model.rel_decoder.forward_bf


def dec_bf(*a, **k):
    out = orig_dec(*a, **k)
    mark("fwd rel decoder")
    bwd_mark(out, "hg encoder + heads bwd done (grad of rel decoder output)")
    return out


model.rel_decoder.forward_bf = dec_bf
model.hgq_encoder.register_forward_hook(fwd_hook("hg cross encoder"))

for i in range(4):
    tr.train_step(batches[i % 2])
torch.cuda.synchronize()
tot = {}
N = 8
all_marks = []
for it in range(N):                 # back-to-back, as in bench.py: the host runs ahead of the GPU across steps
    marks.clear()
    mark("step start")
    out = tr.train_step(batches[it % 2])
    mark("step end (optimizer issued)")
    all_marks.append(list(marks))
torch.cuda.synchronize()
# host column: when the host ISSUED the boundary, on the clock of the step's first event (all steps are timed against the first
# timed step's start on both clocks).  device - host = how far the host runs ahead there; where that lead is ~0 the device waits
# for the host.
ev0, h0 = all_marks[2][0][1], all_marks[2][0][2]
for k, ms in enumerate(all_marks[2:]):
    t0, hh0 = ms[0][1], ms[0][2]
    seq = sorted(((t0.elapsed_time(ev), name, (h - hh0) * 1e3, ev0.elapsed_time(ev) - (h - h0) * 1e3) for name, ev, h in ms[1:]))
    prev = 0.0
    for t, name, h, lead in seq:
        tot.setdefault(name, [0.0, 0.0, 0.0, 1e9])
        tot[name][0] += t / (N - 2)
        tot[name][1] += (t - prev) / (N - 2)
        tot[name][2] += h / (N - 2)
        tot[name][3] = min(tot[name][3], lead)
        prev = t
print("%-62s %9s %9s %12s %16s" % ("boundary (main stream)", "at ms", "segment", "issued at ms", "min device lag"))
for name, (t, d, h, lead) in sorted(tot.items(), key=lambda kv: kv[1][0]):
    print("%-62s %9.2f %9.2f %12.2f %16.2f" % (name, t, d, h, lead))
