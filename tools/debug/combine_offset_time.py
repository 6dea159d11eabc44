"""combine_kernel (465 k rows, in place over `update`) against the byte offset between its two operand arrays modulo the
allocation alignment: are the two / three streams of one wave fighting for the same HBM channels or banks?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ultra_torchdrug_amd import functional as UF

dev = torch.device("cuda:0")
n_node, nq = 14541, 32
rows = n_node * nq
numel = rows * 64
gen = torch.Generator(device=dev).manual_seed(0)
lin, norm = torch.nn.Linear(128, 64).to(dev), torch.nn.LayerNorm(64).to(dev)
args = (lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, True, True)
pool = torch.empty(3 * numel + (64 << 20), dtype=torch.float32, device=dev)      # one allocation: offsets are exact
base = (-pool.data_ptr() // 4) % (1 << 19)                                       # first 2 MiB-aligned element
x = pool[base:base + numel].view(n_node, nq, 64).normal_(generator=gen)


def timed(delta_bytes, reps=15):
    start = base + numel + (32 << 20) // 4 + delta_bytes // 4                    # 2 MiB-aligned + delta
    start -= (start * 4 + pool.data_ptr() - delta_bytes) % (2 << 20) // 4
    u = pool[start:start + numel].view(n_node, nq, 64)
    ts = []
    with torch.no_grad():
        for _ in range(reps):
            u.normal_(generator=gen)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); UF.combine_forward(x, u, *args, reuse_update=True); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    off = (u.data_ptr() - x.data_ptr()) % (2 << 20)
    print("update at x + k * 2 MiB + %8d B: median %.1f us  min %.1f us" % (off, ts[len(ts) // 2], ts[0]))


for d in (0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096 + 256):
    timed(d)
