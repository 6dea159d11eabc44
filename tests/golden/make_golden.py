"""Regenerates tests/golden/rspmm_seeded.npz: seeded inputs + expected outputs of the rspmm hot path.

The reference itself cannot produce vectors here (torchdrug / torch_scatter are not installed and cannot be:
SURVEY.md 8c), so these vectors come from the build's CPU oracle (oracle/rspmm_oracle.c, "parity unpinned") and
serve as a drift detector: the oracle, and on the GPU box the HIP library, must keep reproducing them bit for bit.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from graphs import random_graph   # noqa: E402
from oracle import oracle as O    # noqa: E402

PIECE = 128


def main():
    n, r, F = 96, 5, 72
    g = random_graph(seed=20240, n_node=n, n_edge=1500, n_rel=r, skew=True, unique=True, weights=True, hub_row=9,
                     hub_edges=400, isolated=6)
    rng = np.random.default_rng(77)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr = O.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    out = dict(dst=g["dst"], src=g["src"], rel=g["rel"], w=g["w"], relation=relation, x=x, grad=grad,
               n_node=np.int64(n), n_rel=np.int64(r), piece=np.int64(PIECE))
    for s in ("add", "min", "max"):
        for m in ("mul", "add"):
            fwd = O.rspmm_forward(csr, relation, x, s, m, piece=PIECE)
            d_rel, d_x = O.rspmm_backward(csr, relation, x, fwd, grad, s, m, piece=PIECE)
            out["fwd_%s_%s" % (s, m)] = fwd
            out["drel_%s_%s" % (s, m)] = d_rel
            out["dx_%s_%s" % (s, m)] = d_x
    np.savez_compressed(os.path.join(HERE, "rspmm_seeded.npz"), **out)
    print("wrote", os.path.join(HERE, "rspmm_seeded.npz"))


if __name__ == "__main__":
    main()
