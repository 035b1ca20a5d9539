"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm) over xGMI.  The reference only has single-process nn.DataParallel (agqaHGQA.py:124-129).

Design for MI355X / xGMI:
  * gradients already live in ONE contiguous fp32 arena (engine.py), so a bucket is a slice of it -
    no flatten / unflatten copies, and buckets can be large (default 64 MB: few, large collectives;
    xGMI is point-to-point, so per-collective latency matters more than on a switch).  Bucket bounds
    follow PARAMETER bounds: a large tensor (conv1's 283 MB weight gradient) is cut into buckets of its
    own, so its exchange starts the moment its weight-gradient kernel has been enqueued and does not
    wait for neighbours in the arena that are written much later (the embeddings); with a reducer attached the conv stack's
    backward issues conv2's weight gradient first and conv1's as two launches over output channels (2/3 + 1/3: the same number
    of tile rounds as one launch), so that only the last third of the largest gradient is exchanged behind backward;
  * every backward op reports the slice it has just finished (Engine.grad_written); a bucket whose
    expected number of writes has arrived is all-reduced immediately on a side stream, overlapping
    the remaining backward GEMMs.  The expected counts are learned during the first step (shared
    weights such as the twice-applied x-layer write twice) and CHECKED on every later step: a step
    that writes a bucket more or less often than the learning step raises instead of silently
    reducing a half-written slice;
  * the collective of a bucket is issued on the weight-gradient stream (no fifth stream: the four streams of a step own
    the four hardware queues, see init_process_group), behind one event per
    stream that REPORTED a write into the bucket (autograd runs a backward node on the stream of its forward,
    so the last write of a bucket may come from a branch stream while an earlier one is still in flight on
    main); the executor's flush has already ordered the weight-gradient stream behind the producers of the
    queued weight gradients.  Cross-stream events are kept few on purpose: a separate communication stream
    ordered behind all four streams per bucket cost 3.5 ms per step on a single rank;
  * the weighted set losses are normalised by the GLOBAL sum of class weights (as DataParallel does,
    which computes the loss on the gathered batch): the two loss sums are all-reduced before the
    division, and the BCE term is scaled by 1/world, so the all-reduce is a plain SUM of gradients
    and needs no extra averaging pass;
  * optional bf16 wire format (grad_dtype=torch.bfloat16): a bucket is cast into a bf16 staging
    buffer, reduced, and cast back - half the bytes on the links (0.58 GB instead of 1.16 GB per step)
    at bf16 summation accuracy.  fp32 (= the reference's arithmetic) unless asked for, except with two ranks
    (default_wire_dtype);
  * what a slow collective does to the weight gradients queued behind it on W: the executor's flush issues ALL grouped
    weight-gradient launches of a flush first and only then reports their parameters (Engine.flush_native_wgrads), so a
    bucket's collective sits behind the last group of ITS flush and delays only the groups of LATER flushes - never the
    input-gradient chain on the main stream, which does not wait for W before the gradient norm.  W then carries ~3 ms of
    grouped launches plus the collectives (6.7 ms at 8 ranks, 7.6 ms at 4, 7.6 ms bf16 at 2) inside an ~11 ms backward
    window.  A dedicated communication stream would free W but costs a fifth hardware queue (measured: +3.5-4 ms per step on
    one rank), so it is not used.
"""
import bisect

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


def init_process_group(device, backend="nccl", **kw):
    """torch.distributed.init_process_group for the data-parallel step: binds the engine's four streams to the hardware
    queues FIRST (Engine.bind_streams), then creates the communicator eagerly on `device`.  The other order costs ~2.7 ms
    per step on MI355X even with an idle communicator (its streams take hardware-queue slots, two of the step's streams
    end up sharing one)."""
    from .engine import engine
    engine().bind_streams()
    if backend == "nccl":
        kw.setdefault("device_id", torch.device(device))
    dist.init_process_group(backend, **kw)


def default_wire_dtype(world=None):
    """Wire format of the gradient exchange when none is asked for: fp32 (the reference's arithmetic), except with TWO ranks -
    one xGMI link carries the whole 1.156 GB exchange there (15 ms per step in fp32 against an 11 ms backward window, DESIGN.md
    section 5), so two ranks default to bf16 on the wire (7.6 ms, hidden)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    rccl = dist.is_initialized() and dist.get_backend() == "nccl"        # (the gloo rehearsal path stays fp32)
    return torch.bfloat16 if (world == 2 and rccl) else None


def param_aligned_bounds(n, per, spans=None):
    """Cuts [0, n) into buckets of about `per` elements whose bounds are parameter bounds (spans: sorted
    (offset, numel) of the parameters); a parameter of at least two buckets' size gets buckets of its own."""
    if not spans:
        return [(s, min(n, s + per)) for s in range(0, n, per)]
    bounds, start = [], 0
    for off, numel in spans:
        end = min(n, off + numel)
        if numel >= 2 * per:
            if off > start:
                bounds.append((start, off))
            k = max(1, (numel + per - 1) // per)
            step = (numel + k - 1) // k
            step = (step + 7) // 8 * 8
            for s in range(off, end, step):
                bounds.append((s, min(end, s + step)))
            start = end
        elif end - start >= per:
            bounds.append((start, end))
            start = end
    if start < n:
        bounds.append((start, n))
    return bounds


class GradReducer:
    def __init__(self, grad_arena, bucket_bytes=64 << 20, overlap=True, param_spans=None, grad_dtype=None,
                 force_collectives=False):
        self.arena = grad_arena
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.force = bool(force_collectives)          # issue the collectives even with one rank (tests / bench --force-ddp)
        n = grad_arena.numel()
        per = max(8, bucket_bytes // 4)
        if param_spans is None:
            param_spans = self._engine_spans(n)
        self.bounds = param_aligned_bounds(n, per, param_spans)
        self.starts = [s for s, _ in self.bounds]
        self.expected = None
        self.counts = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.overlap = overlap
        self.use_streams = grad_arena.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.use_streams else None
        self.main_stream = None
        self.launch_order = []
        self.grad_dtype = grad_dtype
        self.staging = {}
        self.writers = [dict() for _ in self.bounds]      # per bucket: streams that reported a write this step
        self.in_burst, self.burst_events = False, {}

    @staticmethod
    def _engine_spans(n):
        try:
            from .engine import engine
            E = engine()
            if E.grad_arena is None or E.grad_arena.numel() != n or E.model is None:
                return None
            spans = sorted((p._shg_off, p._shg_numel) for p in E.model.parameters() if getattr(p, "_shg_grad", None) is not None)
            return spans
        except Exception:
            return None

    def extra_streams(self):
        try:
            from .engine import engine
            return engine().side_streams()
        except Exception:
            return []

    def active(self):
        return self.world > 1 or self.force

    # ------------------------------------------------------------------ hooks
    def begin_step(self):
        self.counts = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.launch_order = []
        self.writers = [dict() for _ in self.bounds]
        self.main_stream = torch.cuda.current_stream() if self.use_streams else None

    def _buckets_of(self, off, numel):
        lo = bisect.bisect_right(self.starts, off) - 1
        hi = bisect.bisect_right(self.starts, off + max(numel, 1) - 1) - 1
        return range(max(lo, 0), min(hi, len(self.bounds) - 1) + 1)

    def on_grad(self, off, numel):
        """Engine.grad_ready_hook: the gradient slice [off, off+numel) has just been written."""
        cur = torch.cuda.current_stream() if self.use_streams else None
        for b in self._buckets_of(off, numel):
            self.counts[b] += 1
            if cur is not None:
                self.writers[b][cur.cuda_stream] = cur
            if self.expected is not None:
                if self.counts[b] > self.expected[b]:
                    raise RuntimeError("GradReducer: bucket %d [%d, %d) was written %d times, the learning step wrote it %d times "
                                       "(the step's set of gradient writes changed: build a new reducer)"
                                       % (b, self.bounds[b][0], self.bounds[b][1], self.counts[b], self.expected[b]))
                if self.overlap and not self.launched[b] and self.counts[b] == self.expected[b]:
                    self._launch(b)

    def _launch(self, b):
        s, e = self.bounds[b]
        view = self.arena[s:e]
        self.launched[b] = True
        self.launch_order.append(b)
        if not self.active():
            return
        if self.use_streams:
            # The collective is issued on the weight-gradient stream W (most of a bucket is written there, and the executor's
            # flush has already ordered W behind every stream that produced a queued operand; W has ~2 ms of work per step, so
            # it has room for the communication), behind one event per stream that REPORTED a write into this bucket (bias /
            # LayerNorm / embedding gradients written on the chain or a branch stream).  Cross-stream events are not free here:
            # ordering a separate communication stream behind all four streams for every bucket (4 x 18 event records per step)
            # cost 3.5 ms per step on ONE rank with no data moved - the decoders' backward segment doubled
            # (tools/step_segments.py, SEG_DDP=1) - most of that turned out to be the hardware-queue collision described at
            # init_process_group above, but the records stay limited to the streams that matter and are shared within a burst.
            W = self._wgrad_stream()
            for st in self.writers[b].values():
                if st != W:
                    ev = self.burst_events.get(st.cuda_stream) if self.in_burst else None
                    if ev is None:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        if self.in_burst:
                            self.burst_events[st.cuda_stream] = ev
                    W.wait_event(ev)
            with torch.cuda.stream(W):
                self._reduce(view, b)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def _reduce(self, view, b):
        """Runs on the weight-gradient stream, in stream order.  bf16 wire: cast - reduce - cast back."""
        if self.grad_dtype is None or self.grad_dtype == view.dtype:
            # synchronous in STREAM terms (the weight-gradient stream waits for the collective, the host does not): with
            # async_op=True handles the same one-rank step took 27.4 instead of 23.5 ms (c10d keeps polling the open works)
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
            return
        buf = self.staging.get(b)
        if buf is None:
            buf = self.staging[b] = torch.empty(view.numel(), dtype=self.grad_dtype, device=view.device)
        buf.copy_(view)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        view.copy_(buf)

    def _wgrad_stream(self):
        try:
            from .engine import engine
            W = engine().wgrad_stream()
        except Exception:
            W = None
        return W if W is not None else self.comm_stream

    def burst(self, on):
        """Engine.flush_native_wgrads brackets its notification loop with burst(True) / burst(False): no kernel is enqueued
        between the notifications of one flush, so the buckets launched inside it share one event per writer stream."""
        self.in_burst = bool(on)
        self.burst_events = {}

    def finish(self):
        """After backward: reduce whatever is left, then make the compute stream wait for the collectives."""
        if self.use_streams:
            try:                                     # weight gradients still queued for a grouped launch report on issue
                from .ops import flush_wgrads
                flush_wgrads()
            except ImportError:
                pass
        if self.expected is not None and self.counts != self.expected:
            bad = [b for b in range(len(self.bounds)) if self.counts[b] != self.expected[b]]
            raise RuntimeError("GradReducer: buckets %s were written %s times, the learning step wrote them %s times"
                               % (bad[:8], [self.counts[b] for b in bad[:8]], [self.expected[b] for b in bad[:8]]))
        for b in reversed(range(len(self.bounds))):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            if hasattr(h, "wait"):
                h.wait()                             # (GPU tensors: the CURRENT stream waits for the collective, not the host)
        if self.use_streams and self.active():
            torch.cuda.current_stream().wait_stream(self._wgrad_stream())
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
