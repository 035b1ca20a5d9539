import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "20", "--warmup", "5"]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
if mode == "lazy":
    dist.init_process_group("nccl")
elif mode == "eager":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
elif mode == "eager_used":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
elif mode == "eager_after":
    # bind the engine's four streams to hardware queues BEFORE the communicator creates its own
    from shg_vqa_amd.engine import engine, reset_engine
    E = reset_engine(compute_dtype=torch.bfloat16, device=torch.device("cuda", 0))
    for st in (torch.cuda.current_stream(), E.wgrad_stream(), E.aux_stream(1), E.aux_stream(2)):
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
    import shg_vqa_amd.engine as EM
    EM.reset_engine = lambda **kw: E                     # bench.py would create a new engine (new streams) otherwise
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
elif mode == "gloo":
    dist.init_process_group("gloo")
import runpy
runpy.run_path("bench.py", run_name="__main__")
