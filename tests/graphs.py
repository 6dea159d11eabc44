"""Seeded synthetic relational graphs shared by the tests (numpy only)."""
import numpy as np


def zipf_choice(rng, n, size, alpha=1.0):
    """Indices 0..n-1 with P(k) ~ 1/(k+1)^alpha, randomly relabelled so that hubs are not the low ids."""
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** alpha
    p /= p.sum()
    perm = rng.permutation(n)
    return perm[rng.choice(n, size=size, p=p)]


def random_graph(seed, n_node, n_edge, n_rel, skew=False, unique=False, weights=False, hub_row=None, hub_edges=0,
                 isolated=0):
    """Returns dict(dst, src, rel, w) int64/float32 arrays.  `isolated` trailing nodes get no in-edges."""
    rng = np.random.default_rng(seed)
    live = max(n_node - isolated, 1)
    if skew:
        dst = zipf_choice(rng, live, n_edge)
        src = zipf_choice(rng, n_node, n_edge)
        rel = zipf_choice(rng, n_rel, n_edge)
    else:
        dst = rng.integers(0, live, n_edge)
        src = rng.integers(0, n_node, n_edge)
        rel = rng.integers(0, n_rel, n_edge)
    if hub_row is not None and hub_edges:
        dst[:hub_edges] = hub_row
    if unique and n_edge:
        key = (dst.astype(np.int64) * n_node + src) * n_rel + rel
        _, first = np.unique(key, return_index=True)
        first.sort()
        dst, src, rel = dst[first], src[first], rel[first]
    w = rng.uniform(0.25, 2.0, dst.shape[0]).astype(np.float32) if weights else None
    return dict(dst=dst.astype(np.int64), src=src.astype(np.int64), rel=rel.astype(np.int64), w=w)


def kg_graph(seed, n_node, n_triple, n_base_rel, alpha=1.0):
    """SURVEY.md 8d generator: Zipf heads/tails/relations, inverse edges (t, h, r + n_base_rel) appended.
    Returns dst/src/rel as rspmm sees them (destination = tail), E = 2 * n_triple, R = 2 * n_base_rel."""
    rng = np.random.default_rng(seed)
    h = zipf_choice(rng, n_node, n_triple, alpha)
    t = zipf_choice(rng, n_node, n_triple, alpha)
    r = zipf_choice(rng, n_base_rel, n_triple, alpha)
    src = np.concatenate([h, t])
    dst = np.concatenate([t, h])
    rel = np.concatenate([r, r + n_base_rel])
    return dict(dst=dst.astype(np.int64), src=src.astype(np.int64), rel=rel.astype(np.int64), w=None)
