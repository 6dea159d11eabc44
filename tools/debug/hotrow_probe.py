"""Timing probe: how fast does the headline forward kernel run when a fraction of its gathers hit ONE row (cache hits)?
(results are wrong on purpose: the node field of a fraction of the packed words is rewritten)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ultra_torchdrug_amd as U
from ultra_torchdrug_amd import functional as UF
from ultra_torchdrug_amd.data import synthetic_kg
dev = torch.device("cuda:0")
g = synthetic_kg("S-fb15k237", device=dev).undirected(add_inverse=True)
csr = U.RelCSR.from_edge_list(g.edge_list, g.edge_weight, g.num_node, g.num_relation)
plan = csr.fwd
F = 2048
gen = torch.Generator(device="cpu").manual_seed(0)
relation = torch.randn(g.num_relation, F, generator=gen).to(dev)
x = torch.randn(g.num_node, F, generator=gen).to(dev)
shift = plan.packed_src_shift
orig = plan.packed.clone()
E = csr.n_edges
def timeit():
    for _ in range(5): UF.rspmm_forward(csr, relation, x, "add", "mul")
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); UF.rspmm_forward(csr, relation, x, "add", "mul"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
print("shift", shift, "baseline us", timeit())
low = (1 << shift) - 1
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    mask = torch.rand(E, device=dev) < frac
    w = orig.clone()
    w64 = w[:E].long() & 0xffffffff
    new = torch.where(mask, (w64 & low) | (7 << shift), w64)
    new = torch.where(new >= 2 ** 31, new - 2 ** 32, new).to(torch.int32)
    plan.packed[:E] = new
    print("fraction of gathers at one row %.2f: %.1f us" % (frac, timeit()))
plan.packed.copy_(orig)
