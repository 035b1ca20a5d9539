import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
dev = "cuda"
gen = torch.Generator().manual_seed(1)
for rows in (786, 354, 256, 786 + 64, 130, 191, 12576):
    probs = []
    for (no, ni) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (1536, 768)]:
        dy = torch.randint(-2, 3, (rows, no), generator=gen).float().to(dev).bfloat16()
        x = torch.randint(-2, 3, (rows, ni), generator=gen).float().to(dev).bfloat16()
        gw = torch.zeros(no, ni, device=dev)
        probs.append((dy, x, gw))
    ref = [dy.float().t() @ x.float() for dy, x, gw in probs]
    K.wgrad_group(probs)
    torch.cuda.synchronize()
    print(rows, [float((gw - r).abs().max()) for (dy, x, gw), r in zip(probs, ref)], flush=True)
