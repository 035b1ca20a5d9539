"""Does a captured hipGraph run fork/join branches concurrently?  Two chains of small-grid GEMM-ish kernels
(each uses a fraction of the CUs) on two streams, eager vs captured."""
import os, sys, time, torch
dev = torch.device('cuda', 0)
n_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 40
a = [torch.randn(512, 2048, device=dev, dtype=torch.bfloat16) for _ in range(2)]
w = [torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16) * 0.01 for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def chain(i):
    x = a[i]
    for _ in range(n_chain):
        x = x @ w[i]
    return x

def both():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    o1 = chain(0)
    with torch.cuda.stream(s2):
        o2 = chain(1)
    cur.wait_stream(s2)
    return o1, o2

def serial():
    return chain(0), chain(1)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print('eager serial   %.3f ms' % timeit(serial))
print('eager 2-stream %.3f ms' % timeit(both))
for name, fn in (('serial', serial), ('2-stream', both)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s1):
            fn()
    print('graph %-9s %.3f ms' % (name, timeit(g.replay)))
