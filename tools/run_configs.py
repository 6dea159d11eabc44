"""Runs the BASELINE.md section-3 workloads on one MI355X and prints a markdown table (kernel-level numbers).

    python tools/run_configs.py [--with-stress] > profiles/rNN_configs.md

Per workload: rspmm forward (add, mul) and, where the config trains, forward+backward; HIP-event timed, median of
20 after 5 warm-ups; algorithmic bytes per SURVEY.md 8d; the CPU port (oracle row loop, OpenMP) on the same graph.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

CONFIGS = [
    # name, workload, B, backward?
    ("CoDExSmall-shaped inference (config 2)", "S-codexs", 16, False),
    ("WN18RR-shaped fine-tuning (config 3)", "S-wn18rr", 16, True),
    ("FB15k237-shaped inference (headline)", "S-fb15k237", 16, False),
    ("FB15k237-shaped pre-training batch (config 4)", "S-fb15k237", 64, True),
    ("CoDExMedium-shaped pre-training batch (config 4)", "S-codexm", 64, True),
]


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return float(np.median(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-stress", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.data import synthetic_kg
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    cfgs = list(CONFIGS)
    if args.with_stress:
        cfgs.append(("Synthetic 10M nodes / 100M edges (config 5)", "S-stress", 1, False))
    threads = len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    print("| config | N | E | R | B | fwd us | fwd edge-msgs/s | fwd algorithmic GB/s (fraction of the bounding peak: 34.5 TB/s XCD-L2 when the slice of the gathered matrix one 64-column tile touches, N x 256 B, is cache-resident (<= 32 MB: an XCD's L2 plus its share of the Infinity Cache; the kernels walk the graph tile by tile), 8 TB/s HBM otherwise) | bwd us (d_input + d_relation) | CPU port fwd edge-msgs/s (%d cores) |" % threads)
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, wl, B, bwd in cfgs:
        g = synthetic_kg(wl, device=dev).undirected(add_inverse=True)
        csr = g.relcsr
        N, E, R, F = g.num_node, csr.n_edges, g.num_relation, 64 * B
        gen = torch.Generator(device=dev).manual_seed(0)
        rel = torch.randn(R, F, device=dev, generator=gen)
        x = torch.randn(N, F, device=dev, generator=gen)
        t_f = timeit(lambda: UF.rspmm_forward(csr, rel, x, "add", "mul"))
        algo = E * (4 * F + 12) + 4 * N * F + 4 * R * F + 4 * (N + 1)
        t_b = None
        if bwd:
            grad = torch.randn(N, F, device=dev, generator=gen)
            t_b = timeit(lambda: UF.rspmm_backward(csr, rel, x, None, grad, "add", "mul"))
        cpu = ""
        if not args.no_cpu and E <= 2_000_000:
            Fc = min(F, 1024)
            co = O.coalesce_csr(csr.dst.cpu().numpy(), csr.src.cpu().numpy(), csr.rel_id.cpu().numpy(), None, N, N, R)
            rc, xc = rel[:, :Fc].cpu().numpy(), x[:, :Fc].cpu().numpy()
            O.rspmm_forward(co, rc, xc)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); O.rspmm_forward(co, rc, xc); ts.append(time.perf_counter() - t0)
            cpu = "%.2e" % (E * (Fc // 64) / float(np.median(ts)))
        print("| %s | %d | %d | %d | %d | %.1f | %.2e | %.0f (%.2f) | %s | %s |" % (
            name, N, E, R, B, t_f, E * B / (t_f * 1e-6), algo / t_f / 1e3, algo / t_f / 1e3 / (34500 if N * 256 <= 32e6 else 8000),
            "%.1f" % t_b if t_b else "-", cpu))
        del g, csr, rel, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
