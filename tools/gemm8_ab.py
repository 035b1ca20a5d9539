"""A/B of the 8-phase kernel against the 2-stage 256^2 kernel (SHG_GEMM8=0/1 must be set before the library loads,
so this script re-runs itself in child processes) on forward-GEMM and conv-forward shapes of the step."""
import os, subprocess, sys
if len(sys.argv) == 1:
    for mode in ("0", "1"):
        env = dict(os.environ, SHG_GEMM8=mode)
        print("SHG_GEMM8=" + mode, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shg_vqa_amd import kernels as K
from tools.gemm_shapes import bench
dev = "cuda"
for (M, N, Kd) in [(8192, 8192, 8192), (4096, 4096, 4096), (12576, 768, 768), (12576, 2304, 768), (12576, 3072, 768), (12576, 768, 3072),
                   (12576, 1536, 768), (18816, 768, 92160), (12544, 768, 34560)]:
    x = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(x, w, y, None, True, True), iters=5 if Kd > 10000 else 20)
    print("  gemm NT M=%6d N=%5d K=%6d  %8.1f us %6.0f TF" % (M, N, Kd, t, 2.0 * M * N * Kd / t / 1e6), flush=True)
    del x, w, y
B = 32
x_cl = torch.randn(B, 16, 9, 9, 2048, device=dev).bfloat16()
w1 = (torch.randn(768, 5, 3, 3, 2048, device=dev) * 0.01).bfloat16()
b1 = torch.zeros(768, device=dev)
y1 = torch.zeros(B, 12, 9, 9, 768, device=dev, dtype=torch.bfloat16)
pre = torch.empty(B, 12, 7, 7, 768, device=dev, dtype=torch.bfloat16)
t = bench(lambda: K.conv3d_k533_fwd(x_cl, w1, b1, 1, pad_out=True, out=y1, want_pre=True, pre_out=pre), iters=5)
print("  conv1 fwd B=32            %8.1f us %6.0f TF" % (t, 2.0 * B * 12 * 49 * 768 * 45 * 2048 / t / 1e6), flush=True)
w2 = (torch.randn(768, 5, 3, 3, 768, device=dev) * 0.01).bfloat16()
t = bench(lambda: K.conv3d_k533_fwd(y1, w2, b1, 1, pad_out=False, want_pre=True), iters=5)
print("  conv2 fwd B=32            %8.1f us %6.0f TF" % (t, 2.0 * B * 8 * 49 * 768 * 45 * 768 / t / 1e6), flush=True)

for (M, N, Kd) in [(12576, 768, 3072), (12576, 768, 2304), (12576, 3072, 768), (12576, 768, 768)]:
    dy = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(Kd, N, device=dev).bfloat16()
    dx = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(dy, w, dx, None, True, False))
    print("  dgrad NN M=%6d N=%5d K=%6d  %8.1f us %6.0f TF" % (M, N, Kd, t, 2.0 * M * N * Kd / t / 1e6), flush=True)
d1 = torch.randn(B, 12, 7, 7, 768, device=dev).bfloat16()
dw1 = torch.zeros(768, 5, 3, 3, 2048, device=dev)
t = bench(lambda: K.conv3d_k533_wgrad(x_cl, d1, dw1, accumulate=True), iters=5)
print("  conv1 wgrad B=32          %8.1f us %6.0f TF" % (t, 2.0 * B * 12 * 49 * 768 * 45 * 2048 / t / 1e6), flush=True)
d2 = torch.randn(B, 8, 7, 7, 768, device=dev).bfloat16()
dw2 = torch.zeros(768, 5, 3, 3, 768, device=dev)
t = bench(lambda: K.conv3d_k533_wgrad(y1, d2, dw2, accumulate=True), iters=5)
print("  conv2 wgrad B=32          %8.1f us %6.0f TF" % (t, 2.0 * B * 8 * 49 * 768 * 45 * 768 / t / 1e6), flush=True)
d2p = torch.nn.functional.pad(d2, (0, 0, 1, 1, 1, 1, 4, 4))
t = bench(lambda: K.conv3d_k533_dgrad(d2p, w2), iters=5)
print("  conv2 dgrad B=32          %8.1f us %6.0f TF" % (t, 2.0 * B * 12 * 49 * 768 * 45 * 768 / t / 1e6), flush=True)
