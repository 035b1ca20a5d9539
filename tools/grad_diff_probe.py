"""Per-parameter gradient difference between two settings of SHG_WGRAD_GROUP on the B=2 golden batch (bf16)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
if len(sys.argv) > 1:
    import test_model_gpu as T
    from shg_vqa_amd.engine import engine
    g = np.load(os.path.join(os.path.dirname(T.__file__), "golden", "agqa_hgqa_b2.npz"))
    tr = T._build(torch.bfloat16)
    cfg, batch = T._oracle_batch("hgqa", g)
    b = T._device_batch(batch)
    tr.model.eval()
    engine().begin_step(); engine().zero_grad()
    out = tr.forward_losses(b)
    out["total"].backward()
    engine().join_side_streams()
    torch.cuda.synchronize()
    torch.save({n: p.grad.detach().float().cpu().clone() for n, p in tr.model.named_parameters() if p.grad is not None}, sys.argv[1])
else:
    for mode in ("1", "3"):
        subprocess.check_call([sys.executable, __file__, "/tmp/g%s.pt" % mode], env=dict(os.environ, SHG_WGRAD_GROUP=mode))
    a, b = torch.load("/tmp/g1.pt"), torch.load("/tmp/g3.pt")
    rows = []
    for n in a:
        d = (a[n] - b[n]).norm().item(); s = a[n].norm().item()
        rows.append((d / (s + 1e-12), d, s, n, tuple(a[n].shape)))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("%.3e  diff %.3e  norm %.3e  %s %s" % r)
