"""Dense form vs edge walk on relation-graph-shaped adjacencies (n nodes x n nodes x 4 types), by density.

    python tools/dense_crossover.py [--nodes 474] [--queries 16] [--out gpurun_out/dense_crossover.json]

For every density the same RelCSR (built with the dense form forced on) runs its forward / d_input / d_relation once through the
matrix-core kernels (csrc/relgraph_dense.hip) and once through the edge-list kernels (knob bit 6), timed with device events over
repeated launches.  The product's switch (relcsr.DENSE_MIN_DENSITY) is set from this table.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(lib, events, fn, n=20, warm=3):
    """Average kernel time (us): HIP events recorded by the library around the plan's main kernel, on its stream (bench.py's
    hook) -- host-side launch cost stays out of it."""
    from bench import timed_kernel
    for _ in range(warm):
        fn()
    ms, _ = timed_kernel(lib, events, fn, n)
    return 1e3 * ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, nargs="+", default=[474, 102, 84, 22])
    ap.add_argument("--queries", type=int, nargs="+", default=[16, 64])
    ap.add_argument("--density", type=float, nargs="+", default=[1.0, 0.5, 0.25, 0.12, 0.06, 0.03])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ["ULTRA_DENSE_MIN_DENSITY"] = "0"          # every graph here carries the dense form; the knob picks the path
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    from bench import HipEvents
    lib = U.require_library()
    events = HipEvents(lib)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(7)
    rows = []
    for n in args.nodes:
        full = torch.cartesian_prod(torch.arange(n), torch.arange(n), torch.arange(4)).to(dev)
        for density in args.density:
            keep = torch.rand(full.shape[0], device=dev, generator=gen) < density if density < 1.0 else torch.ones(full.shape[0], dtype=torch.bool, device=dev)
            e = full[keep]
            if e.shape[0] == 0:
                continue
            csr = U.RelCSR(e[:, 0], e[:, 1], e[:, 2], None, n, n, 4)
            assert csr.dense_form
            for q in args.queries:
                F = 64 * q
                x = torch.randn(n, F, device=dev, generator=gen)
                g = torch.randn(n, F, device=dev, generator=gen)
                rel = torch.randn(4, F, device=dev, generator=gen)
                row = {"nodes": n, "density": density, "edges": int(e.shape[0]), "queries": q}
                for name, knob in (("dense", 0), ("edges", 64)):
                    lib.ultra_rspmm_force_general_path(knob)
                    try:
                        row[name + "_fwd_us"] = timed(lib, events, lambda: UF.rspmm_forward(csr, rel, x, "add", "mul"))
                        row[name + "_dinput_us"] = timed(lib, events, lambda: UF.rspmm_backward(csr, rel, x, None, g, "add", "mul", need_relation=False))
                        row[name + "_drelation_us"] = timed(lib, events, lambda: UF.rspmm_backward(csr, rel, x, None, g, "add", "mul", need_input=False))
                    finally:
                        lib.ultra_rspmm_force_general_path(0)
                rows.append(row)
                print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
