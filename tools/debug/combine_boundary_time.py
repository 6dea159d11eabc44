"""combine_kernel: the first layer's boundary form against the dense form, on the headline shape (465 k rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ultra_torchdrug_amd import functional as UF

dev = torch.device("cuda:0")
n_node, nq = 14541, 32
gen = torch.Generator(device=dev).manual_seed(0)
lin, norm = torch.nn.Linear(128, 64).to(dev), torch.nn.LayerNorm(64).to(dev)
node = torch.randint(0, n_node, (nq,), device=dev, generator=gen).to(torch.int32)
value = torch.randn(nq, 64, device=dev, generator=gen)
dense = torch.zeros(n_node, nq, 64, device=dev)
dense[node.long(), torch.arange(nq, device=dev)] = value
other = torch.randn(n_node, nq, 64, device=dev, generator=gen)


def timed(name, make_update, fn, reps=20):
    ts = []
    for _ in range(reps):
        u = make_update()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(u); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("%-60s median %.1f us  min %.1f us" % (name, ts[len(ts) // 2], ts[0]))


args = (lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, True, True)
with torch.no_grad():
    rnd = lambda: torch.randn(n_node, nq, 64, device=dev, generator=gen)
    zer = lambda: torch.zeros(n_node, nq, 64, device=dev)
    timed("dense form, random update, random input", rnd, lambda u: UF.combine_forward(other, u, *args, reuse_update=True))
    timed("dense form, random update, boundary as dense input", rnd, lambda u: UF.combine_forward(dense, u, *args, reuse_update=True))
    timed("dense form, zero update, boundary as dense input", zer, lambda u: UF.combine_forward(dense, u, *args, reuse_update=True))
    timed("boundary form, random update", rnd, lambda u: UF.combine_forward(None, u, *args, reuse_update=True, input_boundary=(node, value)))
    timed("boundary form, zero update", zer, lambda u: UF.combine_forward(None, u, *args, reuse_update=True, input_boundary=(node, value)))
    a = UF.combine_forward(dense, other.clone(), *args)
    b = UF.combine_forward(None, other.clone(), *args, input_boundary=(node, value))
    print("forms agree:", torch.equal(a, b))
