"""How long does Python need to ISSUE one eager training step (no device sync inside)?"""
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
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000)
batches = bench.synthetic_device_batches(2, 32, 1234, dev)
for i in range(3):
    tr.train_step(batches[i % 2])
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    tr.train_step(batches[trial % 2])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("issue %.1f ms, drain %.1f ms, total %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0)), flush=True)

if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        tr.train_step(batches[i % 2])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)

st0 = torch.cuda.memory_stats()
for i in range(4):
    tr.train_step(batches[i % 2])
torch.cuda.synchronize()
st1 = torch.cuda.memory_stats()
print("reserved GB %.2f -> %.2f ; segment allocs %d -> %d ; peak allocated GB %.2f" % (
    st0["reserved_bytes.all.current"] / 2**30, st1["reserved_bytes.all.current"] / 2**30,
    st0["segment.all.allocated"], st1["segment.all.allocated"], st1["allocated_bytes.all.peak"] / 2**30), flush=True)

for i in range(12):
    tr.train_step(batches[i % 2])
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print("step", i, "segments", st["segment.all.allocated"], "reserved GB %.2f" % (st["reserved_bytes.all.current"] / 2**30), flush=True)
