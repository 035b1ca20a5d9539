"""Runtime context of the HIP path: compute dtype, dropout seed state, and the flat HBM arenas.

MI355X-first memory layout: every parameter of the model lives in ONE contiguous fp32 arena
(parameters that receive gradients first), with a same-shaped gradient arena, two optimiser-state
arenas and a bf16 "shadow" arena holding the matrix-core operand copy of every weight at the same
element offset.  Consequences:
  * zeroing gradients is one memset, the global gradient norm one reduction, clip + BertAdam + the
    bf16 re-cast ONE streaming kernel over 289 M elements (reference: a python loop over 576 tensors,
    lxrt/optimization.py:114-173);
  * data-parallel gradient exchange works on large contiguous slices of the gradient arena - no
    flatten/unflatten copies (shg_vqa_amd/ddp.py);
  * the two (5,3,3) convolution weights are stored channels-last ([Cout,5,3,3,Cin]) - the layout the
    implicit-GEMM kernel reads - and exposed under the reference's [Cout,Cin,5,3,3] shape as a
    permuted view, so checkpoints keep the reference's keys and shapes.
nn.Parameters become views into the arenas; `p.grad` is a persistent view of the gradient arena.
"""
import ctypes
import os

import torch

_ENGINE = None


def engine():
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = Engine()
    return _ENGINE


def reset_engine(**kw):
    global _ENGINE
    _ENGINE = Engine(**kw)
    return _ENGINE


def _round_up(n, m):
    return (n + m - 1) // m * m


class Engine:
    def __init__(self, compute_dtype=torch.bfloat16, seed=9595, device="cuda"):
        self.compute_dtype = compute_dtype
        self.device = torch.device(device)
        self.seed = seed
        self._seed_state = None
        self._stream_id = 0
        self.param_arena = self.grad_arena = self.m_arena = self.v_arena = self.shadow_arena = None
        self.n_active = 0
        self.n_total = 0
        self.step_state = None
        self.training = False            # dropout on/off; set by AGQAModel.train()/eval()
        self.grad_ready_hook = None      # set by ddp: called with (offset, numel) after a gradient write
        self.model = None
        # SHG_OVERLAP_WGRAD / SHG_OVERLAP_BRANCHES = 0 turn the side streams off (measurement switches)
        self.overlap_wgrad = os.environ.get("SHG_OVERLAP_WGRAD", "1") != "0"      # weight gradients on a side stream (ops._WgradStream)
        self._wgrad_stream = None
        self.wgrad_batch = int(os.environ.get("SHG_WGRAD_BATCH", "1"))   # layers per fork of the weight-gradient stream
        self.deferred_wgrads = []
        self.overlap_branches = os.environ.get("SHG_OVERLAP_BRANCHES", "1") != "0"   # independent sub-graphs (action decoder, language layers) on side streams
        self._aux_streams = {}
        # bit i: branch stream i may be used (measurement switch; AGQA.capture keeps only stream 1 while capturing)
        self.branch_mask = int(os.environ.get("SHG_BRANCH_MASK", "255"))
        # --taskHGQA: the cross-modality x-layers, their pooler and the answer head only produce `logit`, which
        # the HGQA loss never reads (agqaHGQA.py:344-345 uses hg_logit).  They stay on the language branch's
        # stream, off the critical path, until AGQAModel.forward joins it (ops.Branch.reenter / join)
        self.defer_x_layers = False
        self.deferred_branch = None
        self.grad_dirty = False           # gradients written since the arena was last zeroed
        self.unjoined = set()             # side streams that received work since the last join_side_streams() (fork / join ledger)
        # weight gradients of the executor calls are queued and go out as grouped launches once ~a chip's worth of 256 x 256
        # tiles is pending (SHG_WGRAD_DEFER=0: every weight gradient is its own launch, issued at once)
        self.defer_wgrads = os.environ.get("SHG_WGRAD_DEFER", "1") != "0"
        self.kv_ahead = int(os.environ.get("SHG_KV_AHEAD", "0"))
        # (priority of the branch streams, priority of the weight-gradient stream): SHG_STREAM_PRIO="-1,0" runs the dependent chains
        # above the weight-gradient backlog (whose 300-us workgroups otherwise hold every CU while a chain kernel waits)
        self.stream_priority = tuple(int(v) for v in os.environ.get("SHG_STREAM_PRIO", "0,0").split(","))
        # queued weight-gradient tiles (256 x 256) from which a grouped launch goes out: two relation layers are 216 tiles = ONE round
        # of the 256 CUs; with 224 (round 2) the queue went out at three layers = 324 tiles = two rounds, the second a quarter
        # full (tools/step_ab.py: 100 / 150 / 180 / 200 / 215 / 224 / 256 / 430 / 512 tiles -> -0.29 / -0.57 / -0.52 / -0.56 / -0.61 / 0 /
        # -0.16 / -0.19 / +0.12 ms per step)
        self.wgrad_flush_tiles = int(os.environ.get("SHG_WGRAD_FLUSH_TILES", "200"))
        self.pending_keep, self.pending_params = [], []
        self._exec = None                 # shg_exec_t* of the sub-layer executor (event ring for the weight-gradient stream)
        self._run = None                  # persistent shg_run_t handed to every executor call
        self.params_ready_event = None
        # set per call by AGQA.train_step(overlap_update=True): the caller then waits (wait_params_ready / a device
        # synchronisation / the next train_step) before it reads parameters on another stream
        self.lazy_adam = False
        self.first_params = None          # the parameters the step reads first (conv1's weight, bias): updated on the main stream
        self.kernel_events = None        # bench.py: list collecting (start, end) events of the dominant kernel
        self.pending_clip = None
        # Gradients with exactly ONE writer per step that is able to SET them and to add their sum of squares to norm_extra as it
        # goes (the two convolutions' weights, 58 % of all parameters: shg_conv3d_k533_wgrad_sumsq): the norm's pass skips them
        # and the optimiser does not zero them.  overwritten: {arena offset: elements} set that way in the current step;
        # unzeroed: the same of the last optimiser pass - every one of them is either overwritten again or zeroed before
        # anything accumulates into it or reads it (claim_overwrite / settle_stale_grads).
        self.fused_conv_norm = os.environ.get("SHG_CONV_NORM_FUSED", "1") != "0"
        # row order of the first convolution's GEMM rows (ops.conv1_forward and its backward): 1 = position-major, with which the
        # weight gradient skips the zero-border positions of every tap (include/shg_vqa.h, shg_conv3d_k533_wgrad_ex)
        self.conv1_row_order = int(os.environ.get("SHG_CONV1_ROW_ORDER", "1"))
        # the conv FORWARDS in position-major rows too: tiles leave out the taps that read only the zero border, the stream-K launch
        # balances their different lengths with its weighted plan ("conv_k_order" bit 5)
        # conv2's input gradient in frame-major rows: tiles leave out the temporal taps that read the four padding frames
        # ("conv_k_order" bit 6, weighted stream-K plan)
        self.conv_dgrad_tm = int(os.environ.get("SHG_CONV_DGRAD_TM", "0"))
        self.conv_fwd_pm = int(os.environ.get("SHG_CONV_FWD_PM", "2"))     # bit 0: conv1 (off: -0.13 ms per step, but 5.8 instead of 2.6 GB of L2 misses per launch - its tiles stop walking K together), bit 1: conv2
        self.norm_extra = None
        self.overwritten, self.unzeroed = {}, {}
        self.overwrite_poisoned = False

    # ------------------------------------------------------------------ dropout plumbing
    @property
    def seed_state(self):
        if self._seed_state is None:
            self._seed_state = torch.tensor([self.seed, 0], dtype=torch.int64, device=self.device)
        return self._seed_state

    def begin_step(self):
        """Resets the per-step call-site counter (the device-side step counter is advanced by the
        optimiser so that hipGraph replays see fresh masks)."""
        self._stream_id = 0
        self.deferred_branch = None

    def next_stream_id(self):
        self._stream_id += 1
        return self._stream_id

    def take_stream_ids(self, n):
        """First of n consecutive dropout call-site ids (the sub-layer executor numbers its call sites itself)."""
        first = self._stream_id + 1
        self._stream_id += n
        return first

    # ------------------------------------------------------------------ sub-layer executor context
    def run_addr(self, dtype_code):
        """Address of the shg_run_t for an executor call issued NOW: torch's current stream, the weight-gradient side
        stream (if enabled), dropout on / off."""
        from . import _lib
        from .kernels import _stream
        R = self._run
        if R is None:
            R = self._run = _lib.RunT()
            h = _lib.lib().shg_exec_create(64)
            if not h:
                raise _lib.ShgError("shg_exec_create failed: %s" % _lib.lib().shg_last_error_string().decode())
            self._exec = h
            R.exec = h
        R.dtype = dtype_code
        R.training = 1 if self.training else 0
        R.stream = _stream()
        side = self.wgrad_stream()
        R.wgrad_stream = side.cuda_stream if side is not None else None
        R.defer_wgrad = 1 if (side is not None and self.defer_wgrads) else 0
        # decoder key / value projections of the encoder memory ahead of the chain, on the weight-gradient stream (idle in a
        # forward pass).  SHG_KV_AHEAD: 0 off (default), 1 only for a decoder issued on the main stream, 2 for every decoder.
        # Measured: -0.5 ms per step when the branches run inline (SHG_BRANCH_MASK=0: 27.76 -> 27.25 ms), nothing with the two
        # branch streams on (23.1-23.3 either way): beside three busy streams the step is bound by the chip's throughput, not
        # by the length of the decoder chain
        R.kv_ahead = 1 if (side is not None and (self.kv_ahead == 2 or (self.kv_ahead == 1 and not self._on_branch_stream()))) else 0
        if side is not None:
            self.unjoined.add(side.cuda_stream)
        R.seed_state = self.seed_state.data_ptr()
        return ctypes.addressof(R)

    def after_backward_call(self, keep, params):
        """Bookkeeping after an executor backward call.  keep: buffers its weight-gradient launches read (they may still be
        queued: shg_run_t.defer_wgrad - hold them until the flush; the launches already issued read them on the side
        stream: record it); params: parameters whose gradient slices the call finished (or queued)."""
        side = self.wgrad_stream()
        if side is not None:
            for t in keep:
                if t is not None:
                    t.record_stream(side)
        self.grad_dirty = True
        if self._run is not None and self._run.defer_wgrad:
            self.pending_keep.append(keep)
            self.pending_params.extend(params)
            from . import _lib
            if _lib.lib().shg_exec_pending_tiles(self._exec) >= self.wgrad_flush_tiles:
                self.flush_native_wgrads()
        elif self.grad_ready_hook is not None:
            for p in params:
                self.grad_written(p)

    def flush_native_wgrads(self):
        """Issues the queued weight gradients (grouped launches on the side stream) and tells the reducer."""
        if self._exec is None:
            return
        from . import _lib
        if self.pending_keep or _lib.lib().shg_exec_pending_tiles(self._exec) > 0:
            side = self.wgrad_stream()
            try:                                   # (the library empties its queues on every exit path, so may we)
                _lib.call("shg_exec_flush_wgrads", self._exec, self._run.dtype, side.cuda_stream if side is not None else None)
            finally:
                self.pending_keep.clear()
            if self.grad_ready_hook is not None:
                burst = getattr(getattr(self.grad_ready_hook, "__self__", None), "burst", None)
                if burst is not None:
                    burst(True)                    # (ddp.GradReducer: one event per writer stream for the whole flush)
                try:
                    for p in self.pending_params:
                        self.grad_written(p)
                finally:
                    if burst is not None:
                        burst(False)
            self.pending_params.clear()

    def note_fork(self, stream):
        """Fork / join ledger: `stream` has received work that the step's origin stream has not waited for yet.
        join_side_streams() clears it; AGQA.capture() refuses to end a capture while it is non-empty."""
        self.unjoined.add(stream.cuda_stream)

    def __del__(self):
        try:
            if self._exec:
                from . import _lib
                _lib.lib().shg_exec_destroy(self._exec)
                self._exec = None
        except Exception:
            pass

    # ------------------------------------------------------------------ arenas
    def adopt(self, model, active_names, groups=()):
        """Moves every parameter of `model` into the arenas.  active_names: set of parameter names
        (as in named_parameters()) that receive gradients for the configured task.  groups: lists of
        parameter names to lay out back-to-back (e.g. query/key/value weights -> one [2304,768] operand)."""
        named = list(model.named_parameters())
        order = {n: i for i, (n, _) in enumerate(named)}
        for grp in groups:                                   # pull a group's members right behind its first one
            base = order[grp[0]]
            for j, n in enumerate(grp[1:], 1):
                order[n] = base + j * 1e-3
        named.sort(key=lambda np_: order[np_[0]])
        act = [(n, p) for n, p in named if n in active_names]
        ina = [(n, p) for n, p in named if n not in active_names]
        missing = set(active_names) - {n for n, _ in named}
        if missing:
            raise KeyError("active parameter names not in the model: %s" % sorted(missing)[:5])
        offs, off = {}, 0
        for n, p in act:
            offs[n] = off
            off = _round_up(off + p.numel(), 8)          # 8 elements: 16-byte aligned bf16 shadows
        self.n_active = off
        for n, p in ina:
            offs[n] = off
            off = _round_up(off + p.numel(), 8)
        self.n_total = off
        dev = self.device
        self.param_arena = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.m_arena = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.v_arena = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.shadow_arena = torch.zeros(self.n_total, dtype=torch.bfloat16, device=dev)
        self.step_state = torch.zeros(1, dtype=torch.int64, device=dev)
        self.offsets = offs
        for n, p in act + ina:
            o, k = offs[n], p.numel()
            store_shape, perm = _storage_layout(p)
            src = p.detach().to(dev, torch.float32)
            if perm is not None:
                src = src.permute(perm)                    # reference layout -> storage layout
            self.param_arena[o:o + k].view(store_shape).copy_(src)
            inv = _inverse(perm)
            p.data = _view(self.param_arena, o, k, store_shape, inv)
            p._shg_off, p._shg_numel = o, k
            p._shg_store = self.param_arena[o:o + k].view(store_shape)
            p._shg_shadow = self.shadow_arena[o:o + k].view(store_shape)
            if n in active_names:
                p.grad = _view(self.grad_arena, o, k, store_shape, inv)
                p._shg_grad = self.grad_arena[o:o + k].view(store_shape)
            else:
                p.grad = None
                p._shg_grad = None
        self.refresh_shadows()
        self.model = model
        return self

    def refresh_shadows(self):
        """bf16 operand copies of every weight (after loading a checkpoint / initialisation)."""
        from . import kernels as K
        self.wait_params_ready()
        K.cast_f32(self.param_arena, self.shadow_arena)

    def operand(self, p):
        """The tensor a matrix-core kernel should read for parameter p (storage layout)."""
        return p._shg_shadow if self.compute_dtype == torch.bfloat16 else p._shg_store

    # ------------------------------------------------------------------ side stream for weight gradients
    def wgrad_stream(self):
        if not self.overlap_wgrad or self.device.type != "cuda":
            return None
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=self.stream_priority[1])
        return self._wgrad_stream

    def _on_branch_stream(self):
        cur = torch.cuda.current_stream().cuda_stream
        return any(s.cuda_stream == cur for s in self._aux_streams.values())

    def aux_stream(self, i):
        """Side stream for an independent branch of the model (None: run it inline)."""
        if not self.overlap_branches or self.device.type != "cuda" or i < 0:
            return None
        if not (self.branch_mask >> i) & 1:
            return None
        if i not in self._aux_streams:
            self._aux_streams[i] = torch.cuda.Stream(device=self.device, priority=self.stream_priority[0])
        return self._aux_streams[i]

    def bind_streams(self):
        """Creates the weight-gradient stream and the two branch streams and runs one tiny kernel on each (and on the current
        stream), so that the four streams of a step own the process's four hardware queues.  Call it BEFORE anything else
        creates streams on the device - in particular before torch.distributed creates the RCCL communicator (whose own
        streams otherwise take queue slots first: two of the step's streams then share a hardware queue and serialise -
        measured 26.4 vs 23.6 ms per step with an idle one-rank communicator, tools/pg_probe.py)."""
        if self.device.type != "cuda":
            return self
        streams = [torch.cuda.current_stream(self.device), self.wgrad_stream(), self.aux_stream(1), self.aux_stream(2)]
        for st in streams:
            if st is not None:
                with torch.cuda.stream(st):
                    torch.zeros(8, device=self.device).add_(1)
        torch.cuda.synchronize(self.device)
        return self

    def side_streams(self):
        """Every side stream this engine has created (idle ones cost one event each to wait for)."""
        return [s for s in (self._wgrad_stream,) if s is not None] + list(self._aux_streams.values())

    def join_side_streams(self):
        """Makes the current stream wait for all weight-gradient work issued so far."""
        if self.deferred_wgrads:
            from . import ops
            ops.flush_wgrads()
        self.flush_native_wgrads()
        cur = torch.cuda.current_stream()
        joined = set()
        for s in self.side_streams():
            cur.wait_stream(s)
            joined.add(s.cuda_stream)
        self.unjoined -= joined

    def zero_grad(self):
        """The optimiser pass leaves the gradient arena zeroed (BertAdam.step): the sweep is only needed when
        something has written gradients since (grad_dirty)."""
        if self.grad_arena is not None and self.grad_dirty:
            self.wait_params_ready()
            self.grad_arena.zero_()
            self.grad_dirty = False
            self.overwritten, self.unzeroed, self.overwrite_poisoned = {}, {}, False
            if self.norm_extra is not None:
                self.norm_extra.zero_()

    # ------------------------------------------------------------------ single-writer gradients (fused norm, no zeroing)
    def norm_scalar(self):
        if self.norm_extra is None:
            self.norm_extra = torch.zeros(1, dtype=torch.float64, device=self.device)
        return self.norm_extra

    def claim_overwrite(self, p):
        """True: the caller SETS p's gradient now (and adds its sum of squares to norm_scalar()).  False: it accumulates as
        usual - into a gradient that is zero or holds this step's earlier contribution."""
        off = p._shg_off
        ok = (self.fused_conv_norm and self.grad_ready_hook is None and self.device.type == "cuda" and off % 4 == 0
              and p._shg_numel % 4 == 0 and off not in self.overwritten and not torch.cuda.is_current_stream_capturing())
        if ok:
            self.overwritten[off] = p._shg_numel
            self.unzeroed.pop(off, None)
            return True
        if off in self.overwritten:
            self.overwrite_poisoned = True          # a second writer in this step: the fused sum no longer describes the gradient
        if self.unzeroed.pop(off, None) is not None:
            p._shg_grad.zero_()                     # left unzeroed by the last optimiser pass
        return False

    def settle_stale_grads(self):
        """Gradients the last optimiser pass left unzeroed and this step has not overwritten hold LAST step's values: zero them."""
        for off, n in list(self.unzeroed.items()):
            if off not in self.overwritten:
                self.grad_arena[off:off + n].zero_()
        self.unzeroed = {}

    def wait_params_ready(self):
        """BertAdam.step may leave the update of everything but the first convolution's weight running on a side
        stream (it overlaps the next step's conv1, which is compute-bound while the update is HBM-bound).  Whoever
        reads parameters, gradients or the step counters on the current stream calls this first."""
        ev, self.params_ready_event = self.params_ready_event, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def grad_written(self, p, first=0, count=None):
        """The gradient of parameter p - or its elements [first, first + count) - has just been enqueued."""
        self.grad_dirty = True
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(p._shg_off + first, p._shg_numel - first if count is None else count)


def _storage_layout(p):
    """(storage shape, permutation reference->storage) ; Conv3d weights are stored channels-last."""
    if p.dim() == 5:
        co, ci, kt, kh, kw = p.shape
        return (co, kt, kh, kw, ci), (0, 2, 3, 4, 1)
    return tuple(p.shape), None


def _inverse(perm):
    if perm is None:
        return None
    inv = [0] * len(perm)
    for i, j in enumerate(perm):
        inv[j] = i
    return tuple(inv)


def _view(arena, off, numel, store_shape, inv_perm):
    v = arena[off:off + numel].view(store_shape)
    return v.permute(inv_perm) if inv_perm is not None else v
