"""BertAdam with the reference's constructor (AGQA/src/lxrt/optimization.py:64-180) running as one
fused kernel over the parameter arena: global-norm clip + Adam moments + decoupled weight decay +
warm-up/linear schedule + bf16 shadow refresh (shg_bertadam_arena)."""
import torch

from . import kernels as K
from .engine import engine


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
    E.pending_clip = (K.grad_norm(E.grad_arena), float(max_norm))
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
        norm, max_norm = getattr(E, "pending_clip", None) or (None, 0.0)
        E.pending_clip = None
        n = E.n_active
        shadow = E.shadow_arena[:n]
        # warmup < 0 means "no warm-up" in the reference (schedule still applies); map to the kernel's contract
        K.bertadam_arena(E.param_arena[:n], E.grad_arena, E.m_arena, E.v_arena, shadow, norm, max_norm, g["lr"],
                         g["warmup"], g["t_total"], E.step_state, g["b1"], g["b2"], g["e"], g["weight_decay"],
                         bump_step=True)
        K.add_i64(E.seed_state[1:], 1)        # next step -> fresh dropout masks (also under graph replay)
        return None
