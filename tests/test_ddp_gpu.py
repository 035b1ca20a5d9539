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
