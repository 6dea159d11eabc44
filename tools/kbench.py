"""Kernel micro-benchmark: one rspmm forward (add, mul) on a benchmark-shaped graph, HIP-event timed.

    ULTRA_RSPMM_LIB=/path/to/variant.so python tools/kbench.py [--workload S-fb15k237] [--batch 16] [--reps 50]

Used to A/B kernel variants (built with different -D flags) in one gpurun call; prints one line per run.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S-fb15k237")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--piece", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--knob", type=int, default=0, help="ultra_rspmm_force_general_path bits (include/ultra_rspmm.h)")
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--boundary", action="store_true", help="forward with the fused `+ boundary` epilogue (add_rows), as inside a layer")
    ap.add_argument("--relgraph", action="store_true", help="the workload's RELATION graph (2R nodes, 4 edge types) instead of the entity graph")
    ap.add_argument("--weights", action="store_true", help="per-edge weights (0 for 1 %% of the edges, 1 elsewhere): the training step's edge removal")
    ap.add_argument("--removed", action="store_true", help="64 edges (and their inverses) removed as a training step removes them: zero weights + marked "
                    "words (RelCSR.with_removed_edges); with --knob 128 the weighted kernels run on the same plans")
    ap.add_argument("--hot", action="store_true", help="plans with the LDS hot-row cache (kernel VAR 4)")
    ap.add_argument("--reserve", type=int, default=0, help="ultra_rspmm_reserve_cus: size the persistent grids for this many compute units fewer")
    ap.add_argument("--combine", action="store_true", help="time the fused layer epilogue (forward, or fwd+bwd with --backward)")
    args = ap.parse_args()
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import _lib, relcsr, functional as UF
    from ultra_torchdrug_amd.data import synthetic_kg
    lib = U.require_library()
    lib.ultra_rspmm_force_general_path((1 if args.general else 0) | args.knob)
    if args.reserve:
        lib.ultra_rspmm_reserve_cus(args.reserve)
    dev = torch.device("cuda:0")
    g = synthetic_kg(args.workload, device=dev)
    if args.relgraph:
        from ultra_torchdrug_amd.rel_model import construct_relation_graph
        g = construct_relation_graph(g)       # adds the inverse relations itself
    else:
        g = g.undirected(add_inverse=True)
    opts = dict(piece_len=args.piece, chunk_edges=args.chunk, hot_cache=args.hot)      # None: chosen from the edge count
    csr = U.RelCSR.from_edge_list(g.edge_list, g.edge_weight, g.num_node, g.num_relation, **opts)
    if args.weights:
        wgen = torch.Generator(device="cpu").manual_seed(1)
        csr = csr.with_edge_weights((torch.rand(g.edge_list.shape[0], generator=wgen) > 0.01).float().to(dev))
    if args.removed:
        pick = torch.randperm(csr.n_edges, generator=torch.Generator(device="cpu").manual_seed(2))[:256].to(dev)
        pick = pick[csr.rel_id[pick] < g.num_relation // 2][:64]
        csr = csr.with_removed_edges(csr.src[pick], csr.dst[pick], csr.rel_id[pick], g.num_relation // 2)
    F = args.batch * 64
    gen = torch.Generator(device="cpu").manual_seed(0)
    relation = torch.randn(g.num_relation, F, generator=gen).to(dev)
    x = torch.randn(g.num_node, F, generator=gen).to(dev)
    grad = torch.randn(g.num_node, F, generator=gen).to(dev)

    bsparse = (torch.randint(0, g.num_node, (args.batch,), generator=gen).to(dev).to(torch.int32),
               torch.randn(args.batch, 64, generator=gen).to(dev))
    if args.combine:
        lin, norm = torch.nn.Linear(128, 64).to(dev), torch.nn.LayerNorm(64).to(dev)
        xi = x.view(g.num_node, args.batch, 64)
        up = grad.view(g.num_node, args.batch, 64)
        if args.backward:
            xi = xi.clone().requires_grad_()

    def run():
        if args.combine:
            if args.backward:
                out = UF.combine(xi, up, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, True, True)
                return torch.autograd.grad(out, [xi, lin.weight], grad_outputs=up)
            return UF.combine_forward(xi, up, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, True, True)
        if args.backward:
            return UF.rspmm_backward(csr, relation, x, None, grad, "add", "mul")
        if args.boundary:     # the layer's fused `+ boundary` epilogue, sparse form
            return UF.rspmm_forward(csr, relation, x, "add", "mul", boundary=bsparse)
        return UF.rspmm_forward(csr, relation, x, "add", "mul")

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3)
    times = np.array(times)
    E = csr.n_edges
    algo = E * (4 * F + 12) + 4 * g.num_node * F + 4 * g.num_relation * F
    if args.hot:
        print("hot rows cached:", csr.fwd.n_hot)
    print("%s lib=%s %s B=%d E=%d chunks=%d pieces=%d: median %.1f us  min %.1f us  (%.2f TB/s algorithmic, %.2e edge-msgs/s)"
          % (args.workload, os.path.basename(_lib.LIB_PATH), "bwd" if args.backward else "fwd", args.batch, E,
             csr.fwd.chunks.shape[0], csr.fwd.n_pieces, np.median(times), times.min(), algo / np.median(times) / 1e6,
             E * args.batch / (np.median(times) * 1e-6)))


if __name__ == "__main__":
    main()
