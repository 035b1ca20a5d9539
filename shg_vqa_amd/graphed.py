"""hipGraph capture of a host-bound SEGMENT of the model (forward and backward), replayed inside the eager step.

The decoders are chains of ~80 (forward) / ~200 (backward) small dependent kernels each; issuing one costs the host
~15 us (4 us of it the bare launch), more than most of them run, and the host has three other streams to feed in the
same window: those segments are host-bound.  A `GraphedSegment` captures fn(*inputs) and its backward into two
hipGraphs over static buffers (the scheme of torch.cuda.make_graphed_callables, reduced to what this code needs) and
exposes them as ONE autograd node: a step then costs the host two replays instead of ~280 launches.

What stays outside the graphs:
  * weight gradients (ops._wgrad): during the backward capture they are only RECORDED (the tensors are static buffers
    of the graph's pool); after every backward replay they are issued eagerly on the weight-gradient stream, beside the
    rest of backward, as before (inside the graph they would lengthen the serial chain);
  * the data-parallel reducer's bookkeeping: the parameters whose gradients the captured kernels write are recorded at
    capture time and reported (Engine.grad_written) after each replay.
Counter-based dropout needs nothing: seed and step live in device memory, call-site ids are constants of the capture.
"""
import torch

from .engine import engine


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, *inputs):
        for dst, src in zip(seg.static_in, inputs):
            if dst is not src:
                dst.copy_(src)
        seg.fwd_graph.replay()
        ctx.seg = seg
        return seg.static_out.detach()

    @staticmethod
    def backward(ctx, grad_out):
        seg = ctx.seg
        seg.static_gout.copy_(grad_out)
        seg.bwd_graph.replay()
        seg.after_backward()
        return (None,) + tuple(g.detach() if g is not None else None for g in seg.static_gin)


class GraphedSegment:
    def __init__(self, fn, sample_inputs, warmup=2):
        """fn(*inputs) -> one tensor.  sample_inputs: tensors of the final shapes / dtypes (those with requires_grad get
        gradients).  Must be called where fn would run (current stream, engine in training mode), before any real
        gradient of this step has been written: the warm-up and capture passes leave garbage in the gradient arena,
        which is zeroed again at the end."""
        E = engine()
        self.E = E
        self.static_in = [t.detach().clone().requires_grad_(t.requires_grad) for t in sample_inputs]
        hook, E.grad_ready_hook = E.grad_ready_hook, None
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):                                   # eager passes: lazy initialisation, allocator warm-up
                E.begin_capture_pass(None)
                out = fn(*self.static_in)
                g = torch.autograd.grad(out, [t for t in self.static_in if t.requires_grad], torch.ones_like(out), allow_unused=True)
                del out, g
            E.join_side_streams()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        overlap, E.overlap_wgrad = E.overlap_wgrad, False             # no forks inside the captures
        try:
            E.begin_capture_pass(None)
            with torch.cuda.graph(self.fwd_graph, pool=pool):
                self.static_out = fn(*self.static_in)
            self.static_gout = torch.zeros_like(self.static_out)
            self.wgrads, self.written = [], []
            E.begin_capture_pass(self)
            with torch.cuda.graph(self.bwd_graph, pool=pool):
                need = [t for t in self.static_in if t.requires_grad]
                grads = torch.autograd.grad(self.static_out, need, self.static_gout, allow_unused=True)
            it = iter(grads)
            self.static_gin = [next(it) if t.requires_grad else None for t in self.static_in]
        finally:
            E.begin_capture_pass(None)
            E.overlap_wgrad = overlap
            E.grad_ready_hook = hook
        torch.cuda.synchronize()
        E.grad_arena.zero_()                                           # warm-up / capture passes accumulated into it
        E.grad_dirty = False

    # called by ops._wgrad / Engine.grad_written while the backward is being captured
    def record_wgrad(self, dy2, x2, weight, bias):
        self.wgrads.append((dy2, x2, weight, bias))

    def record_written(self, p):
        self.written.append(p)

    def after_backward(self):
        from . import ops
        E = self.E
        for dy2, x2, weight, bias in self.wgrads:
            ops._wgrad(dy2, x2, weight, bias)
        for p in self.written:
            E.grad_written(p)

    def __call__(self, *inputs):
        return _Replay.apply(self, *inputs)
