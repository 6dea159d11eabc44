"""Config 5 (S-stress: 10 M nodes / 100 M edges / 1 k relations, 64d, B = 1) forward, generated on the device.

    python tools/stress_bench.py [--reps 10] [--knob 0|8] [--batch 1]

--knob 8: the chunked kernel (packed_kernel VAR 2) instead of one row per group (rowgroup_kernel).  Prints one line;
made to run under tools/pmc.sh.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--knob", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--triples", type=int, default=50_000_000)
    ap.add_argument("--relations", type=int, default=500)
    args = ap.parse_args()
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    lib = U.require_library()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1024)
    h = torch.randint(0, args.nodes, (args.triples,), device=dev, generator=gen)
    t = torch.randint(0, args.nodes, (args.triples,), device=dev, generator=gen)
    r = torch.randint(0, args.relations, (args.triples,), device=dev, generator=gen)
    csr = U.RelCSR(torch.cat([t, h]), torch.cat([h, t]), torch.cat([r, r + args.relations]), None, args.nodes,
                   args.nodes, 2 * args.relations)
    del h, t, r
    F, R = 64 * args.batch, 2 * args.relations
    x = torch.randn(args.nodes, F, device=dev, generator=gen)
    relation = torch.randn(R, F, device=dev, generator=gen)
    lib.ultra_rspmm_force_general_path(args.knob)
    for _ in range(3):
        UF.rspmm_forward(csr, relation, x, "add", "mul")
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        UF.rspmm_forward(csr, relation, x, "add", "mul")
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    E = csr.n_edges
    algo = E * (4 * F + 12) + 4 * args.nodes * F + 4 * R * F + 4 * (args.nodes + 1)
    ms = float(np.median(times))
    print("S-stress knob=%d B=%d E=%d: median %.3f ms min %.3f ms  %.2f TB/s algorithmic = %.3f of 8 TB/s"
          % (args.knob, args.batch, E, ms, min(times), algo / ms / 1e9, algo / ms / 1e9 / 8.0))


if __name__ == "__main__":
    main()
