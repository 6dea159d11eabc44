"""What a few-us kernel costs INSIDE a replayed hipGraph on this box (un-profiled): N dependent tiny elementwise launches captured
once, replayed; per-kernel time = replay time / N.  python tools/ubench/graph_small_kernels.py"""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
for n in (50, 200, 800):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("graph of %4d tiny kernels: %.1f us per replay = %.2f us per kernel" % (n, dt * 1e6, dt * 1e6 / n))
