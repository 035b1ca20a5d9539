"""Two data-parallel ranks on ONE MI355X (gloo transport, both ranks on cuda:0): the gradients after the
bucketed all-reduce must equal the gradients of a single process on the concatenated batch - the same
semantics DataParallel gives the reference (loss normalised over the global batch)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(world=None):
    from oracle import detweights
    from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd.engine import engine, reset_engine
    from shg_vqa_amd.param import hgqa_args
    reset_engine(compute_dtype=torch.float32)
    args = hgqa_args(compute_dtype="fp32")
    model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
    model.to_engine(torch.float32)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.data.copy_(torch.from_numpy(detweights.tensor_for(name, tuple(prm.shape))))
    engine().refresh_shadows()
    red = None
    if world:
        from shg_vqa_amd.ddp import GradReducer
        red = GradReducer(engine().grad_arena, bucket_bytes=32 << 20)
    return AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=4), [None] * 10, None), model=model, t_total=100, world=red)


def _batch(lo, hi):
    from oracle import shg_ref
    cfg = shg_ref.Cfg()
    b = shg_ref.synthetic_batch(4, cfg, seed=77)
    out = {}
    for k, v in b.items():
        if torch.is_tensor(v):
            out[k] = v[lo:hi].contiguous().cuda()
    out["pos"] = out["pos"].float()
    out["lengths"] = out["lengths"].to(torch.int32)
    out["act_lengths"] = out["act_lengths"].to(torch.int32)
    return out


def _grads_after_backward(tr, b):
    from shg_vqa_amd.engine import engine
    E = engine()
    E.begin_step()
    tr.optim.zero_grad()
    E.training = False
    if tr.world is not None:
        tr.world.begin_step()
    out = tr.forward_losses(b)
    out["total"].backward()
    if tr.world is not None:
        tr.world.finish()
    E.join_side_streams()                      # (also issues weight gradients still queued for a grouped launch)
    torch.cuda.synchronize()
    return E.grad_arena.clone(), float(out["rel_ce"]), float(out["act_ce"])


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tr = _build(world=world)
        g, rel, act = _grads_after_backward(tr, _batch(2 * rank, 2 * rank + 2))
        # second step exercises the overlapped launch path (write counts learned in step 1)
        g2, _, _ = _grads_after_backward(tr, _batch(2 * rank, 2 * rank + 2))
        assert torch.allclose(g, g2, rtol=1e-4, atol=1e-6), "overlapped step differs from the learning step"
        assert len(tr.world.launch_order) == len(tr.world.bounds)
        if rank == 0:
            dist.barrier()
            ref_tr = _build(world=None)
            ref, rrel, ract = _grads_after_backward(ref_tr, _batch(0, 4))
            err = (g - ref).abs().max().item()
            scale = ref.abs().max().item()
            ok = err <= 2e-3 * scale and abs(rel - rrel) < 1e-4 * abs(rrel) and abs(act - ract) < 1e-4 * abs(ract)
            q.put((rank, "ok" if ok else "MISMATCH err=%g scale=%g rel %g/%g" % (err, scale, rel, rrel)))
        else:
            dist.barrier()
            q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()[-1500:]))


def test_two_ranks_match_single_process_on_the_global_batch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bucket_collective_waits_for_the_main_stream_when_the_last_write_came_from_a_branch_stream(monkeypatch):
    """A bucket that mixes parameters written on the main stream with parameters written on a branch stream (autograd runs a
    backward node on the stream of its forward): when the LAST write is reported from the branch stream, the collective must
    still be ordered behind the earlier main-stream write.  The collective is replaced by an in-place doubling (what a two-rank
    sum of equal gradients does), the main-stream write is delayed by a long spin kernel: without the ordering the doubling
    would run first and the late write would land un-doubled on top."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd import ddp
    from shg_vqa_amd.engine import reset_engine
    reset_engine(compute_dtype=torch.float32)
    n = 1 << 16
    arena = torch.zeros(n, device="cuda")
    red = ddp.GradReducer(arena, bucket_bytes=4 * n, force_collectives=True, param_spans=[(0, n // 2), (n // 2, n // 2)])
    assert len(red.bounds) == 1
    monkeypatch.setattr(ddp.dist, "all_reduce", lambda view, op=None, async_op=False: view.mul_(2))
    aux = torch.cuda.Stream()
    for step in range(3):
        arena.zero_()
        torch.cuda.synchronize()
        red.begin_step()
        if step:
            torch.cuda._sleep(200_000_000)                  # ~0.1 s on the main stream before its write
        arena[: n // 2].add_(1.0)
        red.on_grad(0, n // 2)
        with torch.cuda.stream(aux):                        # the branch stream: independent of the main stream's work
            arena[n // 2:].add_(3.0)
            red.on_grad(n // 2, n // 2)                     # last expected write -> launches the bucket from here
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(arena[: n // 2], torch.full((n // 2,), 2.0, device="cuda")), (step, arena[:4])
        assert torch.equal(arena[n // 2:], torch.full((n // 2,), 6.0, device="cuda")), (step, arena[-4:])


def _nccl_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from shg_vqa_amd.ddp import GradReducer
        from shg_vqa_amd.engine import engine
        ref_tr = _build(world=None)
        ref, rrel, ract = _grads_after_backward(ref_tr, _batch(0, 2))
        tr = _build(world=None)
        red = GradReducer(engine().grad_arena, bucket_bytes=32 << 20, force_collectives=True)
        tr.world = red
        engine().grad_ready_hook = red.on_grad
        g1, rel, act = _grads_after_backward(tr, _batch(0, 2))          # learning step: every bucket reduced at finish()
        g2, _, _ = _grads_after_backward(tr, _batch(0, 2))              # overlapped step: buckets go out during backward
        early = sum(1 for i, b in enumerate(red.launch_order) if b != len(red.bounds) - 1 - i)
        seen = torch.ones(1, device="cuda")
        dist.all_reduce(seen)
        ok = (torch.allclose(g1, ref, rtol=1e-4, atol=1e-6 * ref.abs().max().item()) and
              torch.allclose(g2, ref, rtol=1e-4, atol=1e-6 * ref.abs().max().item()) and abs(rel - rrel) < 1e-5 * abs(rrel)
              and len(red.launch_order) == len(red.bounds) and int(seen.item()) == 1)
        q.put("ok buckets=%d out_of_order=%d" % (len(red.bounds), early) if ok else
              "MISMATCH %g %g" % ((g1 - ref).abs().max().item(), (g2 - ref).abs().max().item()))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put("FAIL " + traceback.format_exc()[-1500:])


def test_grad_reducer_through_rccl_single_rank_equals_the_plain_step():
    """The reducer's RCCL path (backend "nccl", one rank, collectives forced): the comm stream, the per-bucket waits on every
    writer stream and the in-arena all-reduces run exactly as with N ranks; a one-rank SUM is the identity, so the gradients
    must equal the step without a reducer - in the learning step and in the overlapped step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res.startswith("ok"), res


def test_bench_world_2_branch_runs_under_torch_distributed_run_on_one_gpu():
    """The N > 1 branch of bench.py (rendezvous from the environment, per-rank seeds, the gradient reducer, barrier + MAX-over-ranks
    timing, the all-gather of rank ids, ONE JSON line from rank 0) launched exactly as the driver launches it - two ranks of
    `python -m torch.distributed.run ... bench.py --gpus 2` - but with `--backend gloo` so that both ranks can share this
    box's single GPU (RCCL needs one device per rank)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--backend", "gloo", "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["ranks_seen"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["backend"] == "gloo"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 0.02 * d["value"]
    assert d["roofline"]["launches_timed"] == 2 and len(d["roofline_rows"]) == 2
