"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm) over xGMI.  The reference only has single-process nn.DataParallel (agqaHGQA.py:124-129).

Design for MI355X / xGMI:
  * gradients already live in ONE contiguous fp32 arena (engine.py), so a bucket is a slice of it -
    no flatten / unflatten copies, and buckets can be large (default 64 MB: few, large collectives;
    xGMI is point-to-point, so per-collective latency matters more than on a switch);
  * every backward op reports the slice it has just finished (Engine.grad_written); a bucket whose
    expected number of writes has arrived is all-reduced immediately on a side stream, overlapping
    the remaining backward GEMMs.  The expected counts are learned during the first step (shared
    weights such as the twice-applied x-layer write twice);
  * the weighted set losses are normalised by the GLOBAL sum of class weights (as DataParallel does,
    which computes the loss on the gathered batch): the two loss sums are all-reduced before the
    division, and the BCE term is scaled by 1/world, so the all-reduce is a plain SUM of gradients
    and needs no extra averaging pass.
"""
import torch
import torch.distributed as dist


class _AllReduceSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sums):
        out = sums.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out

    @staticmethod
    def backward(ctx, g):
        # the loss is identical on every rank, so is g: d(global sum)/d(local sum) = 1
        return g


class GradReducer:
    def __init__(self, grad_arena, bucket_bytes=64 << 20, overlap=True):
        self.arena = grad_arena
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        n = grad_arena.numel()
        per = max(1, bucket_bytes // 4)
        self.bounds = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.expected = None
        self.counts = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.overlap = overlap
        self.use_streams = grad_arena.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.use_streams else None
        self.launch_order = []

    def extra_streams(self):
        try:
            from .engine import engine
            return engine().side_streams()
        except Exception:
            return []

    # ------------------------------------------------------------------ hooks
    def begin_step(self):
        self.counts = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.launch_order = []

    def _buckets_of(self, off, numel):
        per = self.bounds[0][1] - self.bounds[0][0]
        return range(off // per, min(len(self.bounds) - 1, (off + max(numel, 1) - 1) // per) + 1)

    def on_grad(self, off, numel):
        """Engine.grad_ready_hook: the gradient slice [off, off+numel) has just been written."""
        for b in self._buckets_of(off, numel):
            self.counts[b] += 1
            if (self.overlap and self.expected is not None and not self.launched[b]
                    and self.counts[b] == self.expected[b]):
                self._launch(b)

    def _launch(self, b):
        s, e = self.bounds[b]
        view = self.arena[s:e]
        self.launched[b] = True
        self.launch_order.append(b)
        if self.world == 1:
            return
        if self.use_streams:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s_ in self.extra_streams():               # weight gradients are written on a side stream
                self.comm_stream.wait_stream(s_)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """After backward: reduce whatever is left, then make the compute stream wait for the collectives."""
        for b in reversed(range(len(self.bounds))):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        if self.use_streams and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.expected is None:
            self.expected = list(self.counts)

    # ------------------------------------------------------------------ losses
    def global_loss_sums(self, sums):
        """[sum w*nll, sum w, correct, matched] summed over all ranks (differentiable in sums[0])."""
        if self.world == 1:
            return sums
        return _AllReduceSums.apply(sums)

    def global_loss_sums2(self, a, b):
        """Both set losses' sums in ONE collective (it sits on the critical path between forward and backward)."""
        if self.world == 1:
            return a, b
        both = _AllReduceSums.apply(torch.cat([a, b]))
        return both[:a.numel()], both[a.numel():]

    def bce_scale(self):
        return 1.0 / self.world
