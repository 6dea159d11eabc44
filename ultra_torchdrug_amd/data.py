"""Knowledge-graph inputs for the hot path: seeded synthetic graphs of the benchmark shapes and a triple reader.

The reference downloads its datasets at run time (``/root/reference/ultra/dataset.py:165,179,450-460``) and
takes FB15k237 / WN18RR from torchdrug; there is no network here, so benchmarks and tests use synthetic graphs of
the same sizes.  A ``(h, r, t)``-per-line reader of the same shape as ``dataset.py:69-96`` lets a
real split be used when one is present on disk.
"""
import os

import numpy as np
import torch

from .graph import Graph

# name: (num_node, train triples, base relations)   -- sizes from SURVEY.md 8a / 8d
SHAPES = {
    "S-codexs": (2034, 32888, 42),
    "S-wn18rr": (40943, 86835, 11),
    "S-fb15k237": (14541, 272115, 237),
    "S-codexm": (17050, 185584, 51),
    "S-stress": (10_000_000, 50_000_000, 500),
    "S-tiny": (300, 2000, 6),
}
DEFAULT_SEED = 1024     # the reference's default seed (ultra/util.py:77)


def _zipf_ranks(rng, n, size, alpha):
    """Ranks 0..n-1 with P(k) ~ 1/(k+1)^alpha (alpha <= 0: uniform)."""
    if alpha <= 0:
        return rng.integers(0, n, size)
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** alpha
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, rng.random(size), side="right"), n - 1)


def synthetic_triples(name_or_shape, seed=DEFAULT_SEED, alpha=None):
    """``(h, t, r)`` int64 triples ``(T, 3)`` in the reference's column order (``ultra/task.py:123``).
    Heads, tails and relations ~ Zipf(alpha) (alpha=1 for the KG shapes, uniform for S-stress); duplicates and
    drawn until exactly the requested number of DISTINCT triples exists, so E = 2 * triples after inverses."""
    n_node, n_triple, n_rel = SHAPES[name_or_shape] if isinstance(name_or_shape, str) else name_or_shape
    if alpha is None:
        alpha = 0.0 if name_or_shape == "S-stress" else 1.0
    rng = np.random.default_rng(seed)
    dedup = n_node * n_node * n_rel < 2 ** 62
    triples = np.zeros((0, 3), dtype=np.int64)
    want = n_triple
    for _ in range(64):                    # oversample until n_triple DISTINCT triples exist (hubs collide often)
        m = int((want - len(triples)) * 1.3) + 16
        new = np.stack([_zipf_ranks(rng, n_node, m, alpha), _zipf_ranks(rng, n_node, m, alpha),
                        _zipf_ranks(rng, n_rel, m, alpha)], axis=1).astype(np.int64)
        triples = np.concatenate([triples, new])
        if dedup:
            key = (triples[:, 0] * n_node + triples[:, 1]) * n_rel + triples[:, 2]
            _, first = np.unique(key, return_index=True)
            triples = triples[np.sort(first)]
        if len(triples) >= want:
            break
    triples = triples[:want]
    # hubs get arbitrary ids
    node_perm, rel_perm = rng.permutation(n_node), rng.permutation(n_rel)
    triples = np.stack([node_perm[triples[:, 0]], node_perm[triples[:, 1]], rel_perm[triples[:, 2]]], axis=1)
    return triples, n_node, n_rel


def synthetic_kg(name_or_shape="S-fb15k237", seed=DEFAULT_SEED, device="cpu", alpha=None):
    """Fact graph (edge_list rows = (h, t, r), as torchdrug KG datasets store them) of a benchmark shape."""
    triples, n_node, n_rel = synthetic_triples(name_or_shape, seed, alpha)
    return Graph(torch.from_numpy(triples).to(device), num_node=n_node, num_relation=n_rel)


def stress_task(device, n_node=None, n_triple=None, n_rel=None, seed=DEFAULT_SEED):
    """BASELINE config 5 as a TASK: the shipped 6 x 64d architecture (seeded random init) over S-stress built ON THE DEVICE --
    uniform heads / tails / relations (duplicates, about 1 in 10^4, are merged by the plans) -- with every triple
    a fact.  10 M nodes / 50 M triples / 500 relations by default: 100 M edges and 1 000 relations as rspmm sees them.
    Returns ``(task in eval mode, generator)``; the generator continues the seeded stream (test batches)."""
    from .task import build_ultra
    d_node, d_triple, d_rel = SHAPES["S-stress"]
    n_node, n_triple, n_rel = int(n_node or d_node), int(n_triple or d_triple), int(n_rel or d_rel)
    gen = torch.Generator(device=device).manual_seed(seed)
    cols = [torch.randint(0, n, (n_triple,), device=device, generator=gen) for n in (n_node, n_node, n_rel)]
    triples = torch.stack(cols, dim=1)
    del cols
    torch.manual_seed(seed)
    task = build_ultra(n_rel).to(device).eval()
    task.preprocess(Graph(triples, None, n_node, n_rel))
    return task, gen


def load_triples(path, entity_vocab=None, relation_vocab=None):
    """Read ``h<TAB>r<TAB>t`` lines (the layout of ``ultra/dataset.py:69-96``) into ``(h, t, r)`` ids."""
    entity_vocab = {} if entity_vocab is None else entity_vocab
    relation_vocab = {} if relation_vocab is None else relation_vocab
    rows = []
    with open(os.path.expanduser(path)) as fin:
        for line in fin:
            parts = line.split()
            if len(parts) != 3:
                continue
            h, r, t = parts
            rows.append((entity_vocab.setdefault(h, len(entity_vocab)), entity_vocab.setdefault(t, len(entity_vocab)),
                         relation_vocab.setdefault(r, len(relation_vocab))))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 3), entity_vocab, relation_vocab


SPLIT_FILES = ("train.txt", "valid.txt", "test.txt")


def load_split_dir(path):
    """A transductive dataset directory in the reference's layout (``ultra/dataset.py:33-66``): ``train.txt``, ``valid.txt``
    and ``test.txt`` of ``h<TAB>r<TAB>t`` lines, ONE entity / relation vocabulary built in that order (``build_vocab`` reads
    the three files one after the other with shared dictionaries).  Returns ``(triples (T, 3) int64 rows of (h, t, r) = train +
    valid + test concatenated, counts [n_train, n_valid, n_test], num_node, num_relation)`` -- what ``self.triplets`` /
    ``self.num_samples`` hold there."""
    path = os.path.expanduser(path)
    entity_vocab, relation_vocab = {}, {}
    parts = []
    for name in SPLIT_FILES:
        file = os.path.join(path, name)
        if not os.path.isfile(file):
            raise FileNotFoundError("%s: a transductive split directory holds %s" % (file, ", ".join(SPLIT_FILES)))
        rows, entity_vocab, relation_vocab = load_triples(file, entity_vocab, relation_vocab)
        parts.append(rows)
    return np.concatenate(parts), [len(p) for p in parts], len(entity_vocab), len(relation_vocab)


def task_from_split_dir(path, checkpoint=None, device="cpu", **task_kwargs):
    """The shipped 6 x 64d Ultra over a real split directory: the train triples carry messages, valid + test belong to the
    filter graph (``ultra/task.py:31-63``), weights from ``checkpoint`` (``td_ultra_3g.pth`` / ``td_ultra_4g.pth`` layout,
    ``ultra/util.py:233-276``) or seeded random init.  Returns ``(task in eval mode on device, splits)`` with
    ``splits = {"train" | "valid" | "test": (n, 3) tensors}``."""
    from .checkpoint import load_checkpoint
    from .task import build_ultra
    triples, counts, n_node, n_rel = load_split_dir(path)
    triples = torch.from_numpy(triples)
    fact_mask = torch.zeros(len(triples), dtype=torch.bool)
    fact_mask[:counts[0]] = True
    torch.manual_seed(DEFAULT_SEED)
    task = build_ultra(n_rel, **task_kwargs)
    missing = unexpected = None
    if checkpoint is not None:
        missing, unexpected = load_checkpoint(task, checkpoint, map_location="cpu")
    task.preprocess(Graph(triples, num_node=n_node, num_relation=n_rel), fact_mask)
    task.to(device).eval()
    bounds = np.cumsum([0] + counts)
    splits = {name.split(".")[0]: triples[bounds[i]:bounds[i + 1]] for i, name in enumerate(SPLIT_FILES)}
    task.checkpoint_keys = (missing, unexpected)
    return task, splits
