"""Where do the min / max gradients of the HIP operator and of the ATen definition differ at F = 1 024 on S-wn18rr?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import test_reference_definition_gpu as T
from ultra_torchdrug_amd import RelCSR, functional as UF

name, F, sum_, mul = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
dev = torch.device("cuda:0")
dst, src, rel, w, n, n_rel = T._graph(name, False)
gen = torch.Generator(device=dev).manual_seed(11)
relation = torch.randn(n_rel, F, device=dev, generator=gen).requires_grad_()
x = torch.randn(n, F, device=dev, generator=gen).requires_grad_()
grad = torch.randn(n, F, device=dev, generator=gen)
csr = RelCSR(dst, src, rel, w, n, n, n_rel)
out = UF.generalized_rspmm(csr, relation, x, sum=sum_, mul=mul)
out.backward(grad)
d_rel, d_x = relation.grad.clone(), x.grad.clone()
relation.grad = x.grad = None
want = T.reference_rspmm(dst, src, rel, w, relation, x, n, sum_, mul)
has = (torch.bincount(dst, minlength=n) > 0).unsqueeze(-1)
want.backward(grad * has)
wd_rel, wd_x = relation.grad.clone(), x.grad.clone()
print("forward equal on rows with edges:", bool(torch.equal(out[has.expand_as(out)], want[has.expand_as(out)])))
diff = (d_x - wd_x).abs()
print("d_x: max diff %.3g at %s; entries > 1e-4: %d of %d" % (diff.max().item(), divmod(int(diff.argmax()), F), int((diff > 1e-4).sum()), diff.numel()))
bad = (diff > 1e-4).nonzero()
cols = torch.unique(bad[:, 1])
print("bad columns:", cols[:40].tolist(), "count", len(cols), " bad column tiles (64):", torch.unique(cols // 64).tolist())
rows = torch.unique(bad[:, 0])
print("bad rows:", len(rows), rows[:20].tolist())
with torch.no_grad():
    msg = relation[rel] * x[src] if mul == "mul" else relation[rel] + x[src]
    for r, c in bad[:6].tolist():
        e = (src == r).nonzero().flatten()
        sel_hip = msg[e, c] == out[dst[e], c]
        sel_ref = msg[e, c] == want[dst[e], c]
        dm = (relation[rel[e], c] if mul == "mul" else torch.ones_like(msg[e, c]))
        brute_hip = (grad[dst[e], c] * dm * sel_hip).sum().item()
        brute_ref = (grad[dst[e], c] * dm * sel_ref).sum().item()
        print("  (row %d, col %d): out-edges %d, hip %.6g, aten %.6g, brute(hip mask) %.6g, brute(ref mask) %.6g, selected %d"
              % (r, c, len(e), d_x[r, c].item(), wd_x[r, c].item(), brute_hip, brute_ref, int(sel_ref.sum())))
d2 = (d_rel - wd_rel).abs()
print("d_rel: max diff %.3g, scale %.3g" % (d2.max().item(), wd_rel.abs().max().item()))
# second launch: run-to-run equality of the HIP gradient
relation.grad = x.grad = None
out2 = UF.generalized_rspmm(csr, relation, x, sum=sum_, mul=mul)
out2.backward(grad)
print("HIP d_x run-to-run equal:", bool(torch.equal(x.grad, d_x)), " d_rel:", bool(torch.equal(relation.grad, d_rel)))
# ---- the test's own bound, and ties
with torch.no_grad():
    hit = msg == want[dst]
    tied = torch.zeros(n, F, device=dev).index_add_(0, dst, hit.float()) > 1.5
    tied_edge = hit & tied[dst]
    print("tied outputs:", int(tied.sum()), "tied edges:", int(tied_edge.sum()))
    ok_x = torch.zeros(n, F, device=dev).index_add_(0, src, tied_edge.float()) < 0.5
    g_abs = grad.abs() * has
    rel_of = relation.abs() if mul == "mul" else torch.ones_like(relation)
    s_x = torch.zeros_like(x).index_add_(0, src, g_abs[dst] * rel_of[rel])
    n_x = torch.bincount(src, minlength=n)
    bound = 32 * n_x.clamp(min=1).float().sqrt().unsqueeze(-1) * 2.0 ** -24 * s_x + 1e-6
    viol = (diff > bound)
    print("bound violators:", int(viol.sum()), " of which not masked as ties:", int((viol & ok_x).sum()))
    for r, c in (viol & ok_x).nonzero()[:8].tolist():
        e = (src == r).nonzero().flatten()
        sel = msg[e, c] == want[dst[e], c]
        print("   (row %d, col %d): n_x %d, s_x %.4g, bound %.3g, diff %.3g, hip %.6g aten %.6g, selected %d, dst of selected %s, tied there %s"
              % (r, c, int(n_x[r]), s_x[r, c].item(), bound[r, c].item(), diff[r, c].item(), d_x[r, c].item(), wd_x[r, c].item(),
                 int(sel.sum()), dst[e][sel][:5].tolist(), tied[dst[e][sel], c][:5].tolist()))
        for d_ in dst[e][sel][:2].tolist():
            into = (dst == d_).nonzero().flatten()
            vals = msg[into, c]
            print("      dst %d: in-edges %d, out(hip) %.9g want %.9g, messages equal to want: %d, sorted head %s"
                  % (d_, len(into), out[d_, c].item(), want[d_, c].item(), int((vals == want[d_, c]).sum()),
                     torch.sort(vals, descending=(sum_ == "max")).values[:3].tolist()))
