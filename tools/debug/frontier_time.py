"""First-layer frontier kernel on REAL evaluation batches of S-fb15k237 (heads and tails of seeded test triples: Zipf hubs),
against the uniform random boundary nodes bench.py's `first_layer_frontier_kernel_ms` uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch


def main():
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.data import synthetic_triples, DEFAULT_SEED
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    dev = torch.device("cuda:0")
    U.require_library().ultra_rspmm_force_general_path(int(os.environ.get("ULTRA_KNOB", "0")))
    n_node, n_fact, n_rel = 14541, 272115, 237
    triples, _, _ = synthetic_triples((n_node, n_fact + 2048, n_rel), DEFAULT_SEED)
    mask = np.zeros(len(triples), dtype=bool); mask[:n_fact] = True
    task = build_ultra(n_rel, full_batch_eval=True)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n_node, num_relation=n_rel), torch.from_numpy(mask))
    task.to(dev).eval()
    und = task.model._undirected(task.fact_graph)
    csr = und.relcsr
    src_ptr, _ = csr.frontier_index
    deg = (src_ptr[1:] - src_ptr[:-1]).cpu().numpy()
    test = torch.from_numpy(triples[n_fact:]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    F = 32 * 64
    rel = torch.randn(csr.shape[2], F, device=dev, generator=gen)
    def timed(nodes):
        b = (nodes.to(torch.int32), torch.randn(len(nodes), 64, device=dev, generator=gen))
        for _ in range(2): UF.rspmm_frontier(csr, rel, b)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): UF.rspmm_frontier(csr, rel, b)
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / 5 * 1e3
    for i in range(4):
        batch = test[16 * i:16 * i + 16]
        nodes = torch.cat([batch[:, 0], batch[:, 1]])
        d = deg[nodes.cpu().numpy()]
        print("test batch %d: out-degrees max %d sum %d -> %.1f us (fill + kernel)" % (i, d.max(), d.sum(), timed(nodes)))
    nodes = torch.randint(0, n_node, (32,), device=dev, generator=gen)
    d = deg[nodes.cpu().numpy()]
    print("uniform nodes: out-degrees max %d sum %d -> %.1f us" % (d.max(), d.sum(), timed(nodes)))
    top = torch.from_numpy(np.argsort(-deg)[:32].copy()).to(dev)
    print("32 largest hubs: out-degrees max %d sum %d -> %.1f us" % (deg[top.cpu().numpy()].max(), deg[top.cpu().numpy()].sum(), timed(top)))


if __name__ == "__main__":
    main()
