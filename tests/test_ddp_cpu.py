"""world_size-2 gloo tests (CPU) of the data-parallel pieces: bucketed all-reduce of the gradient
arena driven by per-slice readiness, and the globally normalised loss sums."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from shg_vqa_amd.ddp import GradReducer
        n = 1000
        arena = torch.zeros(n)
        red = GradReducer(arena, bucket_bytes=4 * 256, overlap=True)       # 4 buckets of 256 floats
        assert len(red.bounds) == 4
        slices = [(900, 100), (600, 300), (600, 300), (256, 344), (0, 256)]   # (offset, numel); one slice written twice
        for step in range(3):
            red.begin_step()
            arena.zero_()
            for off, k in slices:
                arena[off:off + k] += (rank + 1) * (step + 1)
                red.on_grad(off, k)
            red.finish()
            exp = torch.zeros(n)
            for off, k in slices:
                exp[off:off + k] += 3.0 * (step + 1)                          # ranks contribute 1x and 2x
            assert torch.equal(arena, exp), (rank, step)
            if step == 0:
                assert red.launch_order == [3, 2, 1, 0]                       # learning step: all at finish()
            else:
                # buckets go out as soon as their last write lands (the slice [256,600) completes
                # bucket 1 and bucket 2 in that order), identically on every rank
                assert red.launch_order == [3, 1, 2, 0], red.launch_order
        # globally normalised weighted loss: every rank ends with the gradient of the global loss
        sums = torch.tensor([2.0 + rank, 4.0 + 2 * rank, 1.0, 2.0], requires_grad=True)
        gs = red.global_loss_sums(sums * 1.0)
        loss = gs[0] / gs[1]
        loss.backward()
        assert abs(loss.item() - 5.0 / 10.0) < 1e-6
        assert abs(sums.grad[0].item() - 1.0 / 10.0) < 1e-7
        assert red.bce_scale() == 0.5
        # both set losses in one collective
        a = torch.tensor([1.0 + rank, 2.0, 0.0, 1.0], requires_grad=True)
        b = torch.tensor([3.0, 5.0 + rank, 1.0, 1.0], requires_grad=True)
        ga, gb = red.global_loss_sums2(a * 1.0, b * 1.0)
        (ga[0] / ga[1] + gb[0] / gb[1]).backward()
        assert torch.allclose(ga.detach(), torch.tensor([3.0, 4.0, 0.0, 2.0])) and torch.allclose(gb.detach(), torch.tensor([6.0, 11.0, 2.0, 2.0]))
        assert abs(a.grad[0].item() - 0.25) < 1e-7 and abs(b.grad[0].item() - 1.0 / 11.0) < 1e-7
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, "FAIL %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_global_loss_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_reducer_is_a_no_op():
    from shg_vqa_amd.ddp import GradReducer
    arena = torch.arange(10.0)
    r = GradReducer(arena, bucket_bytes=16)
    r.begin_step()
    r.on_grad(0, 10)
    r.finish()
    assert torch.equal(arena, torch.arange(10.0))
    s = torch.tensor([1.0, 2.0, 0.0, 0.0])
    assert r.global_loss_sums(s) is s


def test_bucket_bounds_follow_parameter_bounds():
    """A large tensor gets buckets of its own (conv1's 283 MB weight gradient must not share a bucket with the embeddings,
    which are written at the other end of backward); small tensors are packed up to the bucket size; the cover is exact."""
    from shg_vqa_amd.ddp import param_aligned_bounds
    spans = [(0, 100), (104, 50), (160, 1000), (1160, 8), (1168, 300), (1472, 40)]      # (offset, numel), 8-aligned offsets
    n = 1512
    b = param_aligned_bounds(n, 256, spans)
    assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    big = [x for x in b if x[0] >= 160 and x[1] <= 1160]
    assert big[0][0] == 160 and big[-1][1] == 1160 and len(big) == 4 and all(e - s <= 256 for s, e in big)
    assert (0, 160) in b                                   # the two small tensors in front of it
    assert (1160, 1468) in b and b[-1] == (1468, n)        # the 300-element tensor closes a bucket at its own end
    assert param_aligned_bounds(1000, 256, None) == [(0, 256), (256, 512), (512, 768), (768, 1000)]


def test_reducer_raises_when_a_step_writes_differently_from_the_learning_step():
    from shg_vqa_amd.ddp import GradReducer
    arena = torch.zeros(64)
    r = GradReducer(arena, bucket_bytes=4 * 32)
    for _ in range(2):
        r.begin_step()
        r.on_grad(0, 32)
        r.on_grad(32, 32)
        r.finish()
    r.begin_step()
    r.on_grad(0, 32)
    with pytest.raises(RuntimeError, match="written"):
        r.on_grad(0, 32)                                   # a second write the learning step did not have
    r.begin_step()
    r.on_grad(0, 32)
    with pytest.raises(RuntimeError, match="written"):
        r.finish()                                         # bucket 1 was never written


def test_default_wire_format_is_fp32_except_for_two_rccl_ranks():
    """ddp.default_wire_dtype: bf16 on the wire only where one xGMI link carries the whole exchange (two ranks, RCCL)."""
    from shg_vqa_amd import ddp
    assert ddp.default_wire_dtype(1) is None and ddp.default_wire_dtype(8) is None
    assert ddp.default_wire_dtype(2) is None          # no process group here (and gloo never switches)
