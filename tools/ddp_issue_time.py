"""Host issue time of a training step with the gradient reducer's collectives forced on one rank (RCCL):
does dist.all_reduce block the host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.dup2(2, 1)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
import bench
from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
from shg_vqa_amd.agqa_model import AGQAModel
from shg_vqa_amd.ddp import GradReducer
from shg_vqa_amd.engine import engine, reset_engine
from shg_vqa_amd.param import hgqa_args
reset_engine(compute_dtype=torch.bfloat16, device=dev)
args = hgqa_args(compute_dtype="bf16", batch_size=32)
model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
model.to_engine(torch.bfloat16)
red = GradReducer(engine().grad_arena, force_collectives=True)
tr = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000, world=red)
batches = bench.synthetic_device_batches(2, 32, 1234, dev)
for i in range(4):
    tr.train_step(batches[i % 2])
torch.cuda.synchronize()
print("buckets", len(red.bounds), file=sys.stderr)
for trial in range(3):
    t0 = time.perf_counter(); tr.train_step(batches[trial % 2]); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("issue %.1f ms, drain %.1f ms, total %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0)), file=sys.stderr, flush=True)
# cost of one collective call on an idle GPU
v = engine().grad_arena[:1 << 24]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(v)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("20 x all_reduce(64 MB): host %.2f ms, gpu drain %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), file=sys.stderr)
dist.destroy_process_group()
