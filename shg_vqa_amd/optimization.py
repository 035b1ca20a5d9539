"""BertAdam with the reference's constructor (AGQA/src/lxrt/optimization.py:64-180) running as one
fused kernel over the parameter arena: global-norm clip + Adam moments + decoupled weight decay +
warm-up/linear schedule + bf16 shadow refresh (shg_bertadam_arena)."""
import torch

from . import kernels as K
from .engine import engine


def _round8(n):
    return (n + 7) // 8 * 8


def warmup_linear(x, warmup=0.002):
    """optimization.py:38-43."""
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0)


def clip_grad_norm_(parameters, max_norm):
    """Drop-in for nn.utils.clip_grad_norm_(model.parameters(), max_norm) (agqaHGQA.py:391): computes
    the global L2 norm of the gradient arena on the device and hands the clip to the next
    BertAdam.step(), which folds the scaling into its update kernel.  Returns the norm (fp32 [1], device)."""
    E = engine()
    E.join_side_streams()                       # weight gradients may still be in flight on the side stream
    E.settle_stale_grads()
    if E.overwritten and not E.overwrite_poisoned:
        # the single-writer gradients (conv weights) have added their sums of squares to E.norm_extra already: the pass reads the rest
        ranges, at = [], 0
        for off, n in sorted(E.overwritten.items()):
            ranges.append((at, off))
            at = off + n
        ranges.append((at, E.grad_arena.numel()))
        norm = K.grad_norm_ranges(E.grad_arena, ranges, E.norm_scalar())
    else:
        if E.overwritten:
            E.norm_scalar().zero_()
        norm = K.grad_norm(E.grad_arena)
    E.pending_clip = (norm, float(max_norm))
    return E.pending_clip[0]


class BertAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-5, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if schedule != "warmup_linear":
            raise NotImplementedError("the training loop uses warmup_linear (agqaHGQA.py:147-155)")
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)

    def get_lr(self):
        g = self.param_groups[0]
        step = int(engine().step_state.item())
        if g["t_total"] != -1:
            return [g["lr"] * warmup_linear(step / g["t_total"], g["warmup"])]
        return [g["lr"]]

    def zero_grad(self, set_to_none=True):
        engine().zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        E = engine()
        if E.param_arena is None:
            raise RuntimeError("BertAdam needs the parameters to live in the engine arenas (Engine.adopt)")
        g = self.param_groups[0]
        E.join_side_streams()
        E.wait_params_ready()
        E.settle_stale_grads()
        if E.overwritten and getattr(E, "pending_clip", None) is None:
            E.norm_scalar().zero_()             # (a step without clip_grad_norm_: nobody consumed the fused sums)
        norm, max_norm = getattr(E, "pending_clip", None) or (None, 0.0)
        E.pending_clip = None
        n = E.n_active
        # gradients their single writer SETS in every step (Engine.claim_overwrite) are not zeroed: 4 bytes per parameter less
        keep = sorted(E.overwritten.items())
        E.unzeroed, E.overwritten, E.overwrite_poisoned = dict(keep), {}, False

        def update(lo, hi, bump):
            if hi <= lo:
                return
            segs, at = [], lo
            for off, cnt in keep:
                a, b = max(off, lo), min(off + cnt, hi)
                if b > a:
                    if a > at:
                        segs.append((at, a, True))
                    segs.append((a, b, False))
                    at = b
            if hi > at:
                segs.append((at, hi, True))
            for i, (a, b, zero) in enumerate(segs):
                # warmup < 0 means "no warm-up" in the reference (schedule still applies); map to the kernel's contract
                K.bertadam_arena(E.param_arena[a:b], E.grad_arena[a:b], E.m_arena[a:b], E.v_arena[a:b],
                                 E.shadow_arena[a:b], norm, max_norm, g["lr"], g["warmup"], g["t_total"], E.step_state,
                                 g["b1"], g["b2"], g["e"], g["weight_decay"], bump_step=bump and i == len(segs) - 1, zero_grad=zero)

        first = E.first_params
        side = None
        if E.lazy_adam and first and not torch.cuda.is_current_stream_capturing():
            lo = min(p._shg_off for p in first)
            hi = min(n, _round8(max(p._shg_off + p._shg_numel for p in first)))
            if hi - lo <= sum(_round8(p._shg_numel) for p in first) and all(p._shg_grad is not None for p in first):
                side = E.aux_stream(2)                      # (the parameters must sit back to back in the arena)
        if side is None:
            update(0, n, True)
            K.add_i64(E.seed_state[1:], 1)    # next step -> fresh dropout masks (also under graph replay)
        else:
            # The step's first consumer of parameters is conv1 (2 ms, bound by the matrix cores): its weight and bias are
            # updated here, everything else on a side stream, where the HBM-bound sweep overlaps the next step's conv1.
            main = torch.cuda.current_stream()
            update(lo, hi, False)             # alone on the chip first: it gates the next step's conv1
            done_main = torch.cuda.Event()
            done_main.record(main)
            if norm is not None:
                norm.record_stream(side)      # read by the side stream's kernels after this function has dropped it
            with torch.cuda.stream(side):
                side.wait_event(done_main)    # (also orders it behind the gradient norm and the joined weight gradients,
                update(0, lo, False)          #  and every update kernel reads the step counter before it is bumped)
                update(hi, n, True)
                K.add_i64(E.seed_state[1:], 1)
                ev = torch.cuda.Event()
                ev.record(side)
            E.params_ready_event = ev
        E.grad_dirty = False
        return None
