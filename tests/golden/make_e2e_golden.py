"""Regenerates tests/golden/e2e_codexs.json: the END-TO-END fixture of SURVEY.md 8c(v) -- int64 filtered ranks and a checksum of
the scores of a seeded random-init 6 x 64d Ultra on the seeded S-codexs graph (N = 2 034, 32 888 fact triples, 42 relations),
one batch of 16 held-out triples = 32 queries over all entities (/root/reference/ultra/task.py:228-277,307-315).

Computed on the CPU with the oracle behind every operator (tests/oracle_ops.py), twice:
  * ``reference_order``: the rspmm sums strictly sequential per row (oracle ``piece = 0``: the reference's order);
  * ``kernel_order``:    the HIP library's documented order (split rows in pieces; the dense relation-graph form) -- the HIP path
                         must reproduce these scores bit for bit, so their SHA-256 is part of the fixture.
The reference itself cannot produce vectors here (torchdrug / torch_scatter absent, checkpoints are missing blobs: SURVEY 8c),
so this is a drift detector for BOTH sides at once: live HIP-vs-oracle equality cannot see a change that moves the two together.

    python tests/golden/make_e2e_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

SHAPE, SEED, N_TEST, BATCH = "S-codexs", 1024, 64, 16


def build_task():
    """The seeded task and its first batch of held-out triples (the construction the GPU tests and bench.py use)."""
    from ultra_torchdrug_amd.data import SHAPES, synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    n_node, n_fact, n_rel = SHAPES[SHAPE]
    triples, _, _ = synthetic_triples((n_node, n_fact + N_TEST, n_rel), SEED)
    fact_mask = np.zeros(len(triples), dtype=bool)
    fact_mask[:n_fact] = True
    torch.manual_seed(SEED)
    task = build_ultra(n_rel)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n_node, num_relation=n_rel), torch.from_numpy(fact_mask))
    return task.eval(), torch.from_numpy(triples[n_fact:n_fact + BATCH])


def digest(pred):
    """SHA-256 of the fp32 scores' bytes with -0.0 folded into +0.0 (equal as numbers)."""
    a = np.ascontiguousarray(pred.detach().cpu().numpy(), dtype=np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


def summary(pred, ranks):
    p = pred.detach().cpu().double()
    return {"ranks": ranks.cpu().tolist(), "scores_sha256": digest(pred), "scores_sum": float(p.sum()),
            "scores_abs_sum": float(p.abs().sum()), "scores_max": float(p.max()), "scores_min": float(p.min())}


def oracle_side(piece):
    from oracle_ops import oracle_rspmm
    task, batch = build_task()
    with torch.no_grad(), oracle_rspmm(piece):
        pred = task.predict(batch)
        ranks = task.get_ranking(pred, task.target(batch))
    return pred, ranks


def main():
    out = {"shape": SHAPE, "seed": SEED, "batch": BATCH, "queries": 2 * BATCH,
           "what": "filtered ranks (task.py:307-315) and score checksums of a seeded random-init 6 x 64d Ultra, oracle path on the CPU"}
    for name, piece in (("reference_order", 0), ("kernel_order", None)):
        pred, ranks = oracle_side(piece)
        out[name] = summary(pred, ranks)
        out["score_shape"] = list(pred.shape)
    with open(os.path.join(HERE, "e2e_codexs.json"), "w") as f:
        json.dump(out, f, indent=1)
    same = out["reference_order"]["ranks"] == out["kernel_order"]["ranks"]
    print("wrote e2e_codexs.json; ranks of the two orders identical:", same)


if __name__ == "__main__":
    main()
