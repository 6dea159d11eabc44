import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from test_reference_definition_gpu import _graph, reference_rspmm
from ultra_torchdrug_amd import RelCSR, functional as UF, _lib
dev = torch.device("cuda:0")
for name, weights, sum, mul in [("S-fb15k237", False, "add", "add"), ("S-wn18rr", False, "add", "add"), ("S-wn18rr", False, "max", "mul")]:
    dst, src, rel, w, n, n_rel = _graph(name, weights)
    F = 128
    gen = torch.Generator(device=dev).manual_seed(11)
    relation = torch.randn(n_rel, F, device=dev, generator=gen).requires_grad_()
    x = torch.randn(n, F, device=dev, generator=gen).requires_grad_()
    grad = torch.randn(n, F, device=dev, generator=gen)
    csr = RelCSR(dst, src, rel, w, n, n, n_rel)
    want = reference_rspmm(dst, src, rel, w, relation, x, n, sum, mul)
    has = (torch.bincount(dst, minlength=n) > 0).unsqueeze(-1)
    want.backward(grad * has)
    wd_rel, wd_x = relation.grad.clone(), x.grad.clone()
    lib = _lib.load()
    for knob in (0, 4, 1):
        lib.ultra_rspmm_force_general_path(knob)
        relation.grad = x.grad = None
        out = UF.generalized_rspmm(csr, relation, x, sum=sum, mul=mul)
        out.backward(grad)
        d_rel, d_x = relation.grad.clone(), x.grad.clone()
        e_rel = (d_rel - wd_rel).abs(); e_x = (d_x - wd_x).abs()
        cnt = torch.bincount(rel, minlength=n_rel)
        r_bad = e_rel.max(dim=1).values.argmax().item()
        print(name, sum, mul, "knob", knob, "fwd maxdiff %.3g" % (out - want)[has.expand_as(out)].abs().max().item(),
              "d_rel maxdiff %.3g at rel %d (edges %d, |want| %.3g)" % (e_rel.max().item(), r_bad, cnt[r_bad].item(), wd_rel[r_bad].abs().max().item()),
              "d_x maxdiff %.3g (|want| max %.3g)" % (e_x.max().item(), wd_x.abs().max().item()))
        bad_rows = (e_rel.max(dim=1).values > 1e-3 * (1 + wd_rel.abs().max(dim=1).values)).nonzero().flatten()
        print("   relations off by >1e-3 rel:", bad_rows[:20].tolist(), "counts", cnt[bad_rows[:20]].tolist())
        xr = (e_x.max(dim=1).values > 1e-3 * (1 + wd_x.abs().max(dim=1).values)).nonzero().flatten()
        print("   x rows off:", xr[:10].tolist(), "out-degree", torch.bincount(src, minlength=n)[xr[:10]].tolist())
    lib.ultra_rspmm_force_general_path(0)
