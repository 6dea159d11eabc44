"""One entity layer on S-stress (config 5), fused (csrc/layer_fused.hip) or as the two launches it replaces, event-timed.

    [ULTRA_RSPMM_LIB=variant.so] python tools/layer_bench.py [--queries 2] [--reps 6] [--form fused|split|both]

Prints one line per form.  Made to run under tools/pmc_layer_fused.sh (FETCH_SIZE / WRITE_SIZE per launch: the A/B that shows
the `update` tensor never reaching memory on the fused path).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=2)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--form", default="both")
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--triples", type=int, default=50_000_000)
    ap.add_argument("--relations", type=int, default=500)
    args = ap.parse_args()
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    U.require_library()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1024)
    h = torch.randint(0, args.nodes, (args.triples,), device=dev, generator=gen)
    t = torch.randint(0, args.nodes, (args.triples,), device=dev, generator=gen)
    r = torch.randint(0, args.relations, (args.triples,), device=dev, generator=gen)
    csr = U.RelCSR(torch.cat([t, h]), torch.cat([h, t]), torch.cat([r, r + args.relations]), None, args.nodes, args.nodes,
                   2 * args.relations)
    del h, t, r
    N, Q, R = args.nodes, args.queries, 2 * args.relations
    F = 64 * Q
    x = torch.randn(N, Q, 64, device=dev, generator=gen)
    relation = torch.randn(R, F, device=dev, generator=gen)
    w = torch.randn(64, 128, device=dev, generator=gen) * 0.1
    b = torch.randn(64, device=dev, generator=gen) * 0.1
    g = 1 + 0.1 * torch.randn(64, device=dev, generator=gen)
    beta = 0.1 * torch.randn(64, device=dev, generator=gen)
    boundary = (torch.randint(0, N, (Q,), device=dev, generator=gen).to(torch.int32), torch.randn(Q, 64, device=dev, generator=gen))

    def fused():
        return UF.layer_forward(csr, relation, x, boundary, w, b, g, beta, 1e-5, True, True)

    def split():
        update = UF.rspmm_forward(csr, relation, x.flatten(1), "add", "mul", boundary=boundary).view(N, Q, 64)
        return UF.combine_forward(x, update, w, b, g, beta, 1e-5, True, True, reuse_update=True)

    E = csr.n_edges
    algo = E * (4 * F + 12) + 4 * N * F + 4 * R * F + 4 * (N + 1)
    results = {}
    for name, fn in (("fused", fused), ("split", split)):
        if args.form not in ("both", name):
            continue
        out = fn()
        assert out is not None, "the fused entry declined"
        del out
        torch.cuda.synchronize()
        times = []
        for _ in range(args.reps):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            z.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(z))
            del out
        ms = float(np.median(times))
        # a layer's algorithmic bytes: the rspmm's + the epilogue's own input rows in and output rows out (fused) -- the two
        # launches move the update rows out and in again on top of that
        layer = algo + (2 if name == "fused" else 4) * N * F * 4 - 4 * N * F
        results[name] = ms
        print("S-stress layer %s Q=%d F=%d: median %.3f ms (min %.3f)  %.1f GB algorithmic = %.2f TB/s = %.3f of 8 TB/s"
              % (name, Q, F, ms, min(times), layer / 1e9, layer / ms / 1e9, layer / ms / 1e9 / 8.0))
    if len(results) == 2:
        print("fused / split = %.3f" % (results["fused"] / results["split"]))


if __name__ == "__main__":
    main()
