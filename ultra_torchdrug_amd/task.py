"""Knowledge-graph-completion task: the caller loop that defines "results" for the hot path.

Mirrors ``/root/reference/ultra/task.py`` -- transductive (``:197-351``), inductive (``:525-634``: per-split graphs)
and multi-graph pre-training (``:637-890``: one graph context per dataset, batches carry a graph id):

* ``KnowledgeGraphCompletionBase`` ``:22-195``  -- fact-graph split, filter masks (``:65-100``), strict negatives
  (``:102-118``), BCE + self-adversarial loss (``:160-195``);
* ``KnowledgeGraphCompletionAdapted`` ``:197-351`` -- relation-graph preparation (``:215-226``), ``predict``
  (``:228-277``), ``get_ranking`` (``:307-315``), ``evaluate`` (``:317-351``).

Differences, all outside the arithmetic: no torchdrug ``Registry``/``Engine``; graphs are plain attributes
(not module buffers) so ``state_dict()`` holds exactly the ``model.*`` and ``rel_models.*`` tensors a reference
checkpoint holds after ``util.clean_save`` (``ultra/util.py:278-325``); scores and masks stay on the device and
only the int64 ranks leave it (the reference moves ``(B, 2, N)`` scores and masks to the host per batch,
``task.py:263,295``).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import backend, layer
from .graph import Graph


def variadic_sample(candidates, sizes, num_sample):
    """``torchdrug.layers.functional.variadic_sample``: ``num_sample`` draws (with replacement) from each of the
    variable-length runs of ``candidates`` (``task.py:108,113``)."""
    rand = torch.rand(len(sizes), num_sample, device=sizes.device)
    index = (rand * sizes.unsqueeze(-1)).long()
    index = index + (sizes.cumsum(0) - sizes).unsqueeze(-1)
    return candidates[index]


class KnowledgeGraphCompletion(nn.Module):
    """``tasks.KnowledgeGraphCompletionAdapted`` with the configuration surface of
    ``config/transductive/inference.yaml:8-39``."""

    def __init__(self, model, rel_models, criterion="bce",
                 metric=("mr", "mrr", "hits@1", "hits@3", "hits@10", "mrr-tail", "hits@1-tail", "hits@10-tail"),
                 num_negative=128, margin=6, adversarial_temperature=0, strict_negative=True, filtered_ranking=True,
                 fact_ratio=None, sample_weight=False, metric_per_rel=False, full_batch_eval=False):
        super().__init__()
        assert strict_negative                                   # task.py:27
        self._relation_cache = {}                                # graph context -> per relation model (R, 2R, 64) tables
        self.model = model
        self.rel_models = rel_models
        self.criterion = {criterion: 1} if isinstance(criterion, str) else dict(criterion)
        self.metric = tuple(metric)
        self.num_negative = num_negative
        self.margin = margin
        self.adversarial_temperature = adversarial_temperature
        self.strict_negative = strict_negative
        self.filtered_ranking = filtered_ranking
        self.fact_ratio = fact_ratio
        self.sample_weight = sample_weight
        self.metric_per_rel = metric_per_rel
        self.full_batch_eval = full_batch_eval
        # full-batch evaluation scores tails and heads in ONE Bellman-Ford over 2B queries instead of two over B
        # (task.py:249-259 issues two model calls).  Queries are independent columns of every kernel on the path, so
        # every score is bit-identical to the two-call form (tests/test_model_gpu.py); False = the literal two calls.
        self.fuse_sides = True
        self.contexts = {}
        self.split = None
        if fact_ratio is not None and not 0 < float(fact_ratio) <= 1:
            raise ValueError("fact_ratio must be in (0, 1], got %r" % (fact_ratio,))

    @property
    def device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ graphs (task.py:31-63, 215-226, 539-581, 655-672)
    # A context = the graphs one call works on: `graph` (all known triples, used to filter the ranking),
    # `fact_graph` (edges that carry messages) and the relation graph(s) built from the fact graph.  The transductive
    # task has one ("default"); the inductive task has "train" / "valid" / "test" (task.py:539-581: separate entity
    # sets, shared relation vocabulary); multi-graph pre-training has one per dataset (task.py:655-672).  They are
    # plain attributes, not module buffers, so state_dict() holds exactly the tensors a reference checkpoint holds
    # after util.clean_save.
    def add_context(self, name, graph, fact_graph=None, fact_mask=None, train_triples=None):
        """``train_triples`` (``(T, 3)`` rows of (h, t, r); default: the fact edges): the training set the
        ``sample_weight`` degree tables are counted over (task.py:50-57)."""
        if fact_graph is None:
            fact_graph = graph if fact_mask is None else graph.edge_mask(fact_mask)
        if fact_graph.num_node != graph.num_node:
            # scores are indexed by the fact graph's entities, filters by the graph's: they must be one entity set
            # (task.py:31-63, 539-581 -- every split of a dataset shares its vocabulary)
            raise ValueError("context `%s`: the filter graph has %d nodes, the fact graph %d"
                             % (name, graph.num_node, fact_graph.num_node))
        ctx = {"graph": graph, "fact_graph": fact_graph,
               "rel_graphs": [rel_model.construct_relation_graph(fact_graph) for rel_model in self.rel_models]}
        if self.sample_weight:
            ctx["degree"] = self._degree_tables(fact_graph.edge_list if train_triples is None else train_triples,
                                                fact_graph.num_relation)
        self.contexts[str(name)] = ctx
        if self.split is None:
            self.split = str(name)
        return ctx

    def use(self, name):
        """Select the context subsequent calls work on (the reference's ``self.split``, task.py:590-592)."""
        if str(name) not in self.contexts:
            raise KeyError("unknown graph context `%s` (have %s)" % (name, sorted(self.contexts)))
        self.split = str(name)
        return self

    def _ctx(self, key):
        if self.split is None:
            raise RuntimeError("call preprocess() / add_context() first")
        return self.contexts[self.split][key]

    graph = property(lambda self: self._ctx("graph"))
    fact_graph = property(lambda self: self._ctx("fact_graph"))
    rel_graphs = property(lambda self: self._ctx("rel_graphs"))
    num_entity = property(lambda self: self._ctx("fact_graph").num_node)
    num_relation = property(lambda self: self._ctx("fact_graph").num_relation)

    def preprocess(self, graph, fact_mask=None, generator=None):
        """Transductive setup (task.py:31-63, 215-226): ``graph`` holds all triples (train+valid+test) with (h, t, r)
        rows; ``fact_mask`` selects the training edges, which message passing may use.  With ``fact_ratio`` only that
        fraction of them (a random subset) stays in the fact graph and the REST becomes the training set
        (task.py:42-47); ``self.train_index`` lists the training triples' rows of ``graph.edge_list`` either way."""
        self.contexts, self.split = {}, None
        if fact_mask is None:
            fact_mask = torch.ones(graph.num_edge, dtype=torch.bool, device=graph.device)
        fact_mask = torch.as_tensor(fact_mask, dtype=torch.bool, device=graph.device).clone()
        train_index = fact_mask.nonzero().flatten()
        if self.fact_ratio:
            length = int(len(train_index) * self.fact_ratio)
            rest = torch.randperm(len(train_index), generator=generator)[length:].to(graph.device)
            train_index = train_index[rest]
            fact_mask[train_index] = False
        self.train_index = train_index
        self.add_context("default", graph, fact_mask=fact_mask, train_triples=graph.edge_list[train_index])
        return self

    @staticmethod
    def _degree_tables(triples, num_relation):
        """task.py:50-57 -- how often each (h, r) and each (t, r) occurs in the training set -- as sorted pair keys
        with counts (the reference's dense ``(num_entity, num_relation)`` tables do not scale to big graphs)."""
        h, t, r = triples[:, 0], triples[:, 1], triples[:, 2]
        n_rel = max(int(num_relation), 1)
        return {"n_rel": n_rel,
                "hr": torch.unique(h * n_rel + r, return_counts=True),
                "tr": torch.unique(t * n_rel + r, return_counts=True)}

    def _pair_degree(self, side, node, rel):
        table = self._ctx("degree")
        keys, counts = table[side]
        keys, counts = keys.to(node.device), counts.to(node.device)
        want = node * table["n_rel"] + rel
        if keys.numel() == 0:
            return torch.zeros_like(want)
        pos = torch.searchsorted(keys, want).clamp(max=keys.numel() - 1)
        return torch.where(keys[pos] == want, counts[pos], torch.zeros_like(want))

    def preprocess_inductive(self, train_graph, valid_graph, test_graph, graph=None, inductive_graph=None):
        """Inductive setup (task.py:539-581, target :435-450): messages travel on the split's own graph; rankings are
        filtered against ``graph`` (train/valid) and ``inductive_graph`` (test)."""
        self.contexts, self.split = {}, None
        self.add_context("train", graph if graph is not None else train_graph, fact_graph=train_graph)
        self.add_context("valid", graph if graph is not None else valid_graph, fact_graph=valid_graph)
        self.add_context("test", inductive_graph if inductive_graph is not None else test_graph, fact_graph=test_graph)
        return self.use("train")

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        device = self.device
        for ctx in self.contexts.values():
            same = ctx["fact_graph"] is ctx["graph"]
            ctx["graph"] = ctx["graph"].to(device)
            ctx["fact_graph"] = ctx["graph"] if same else ctx["fact_graph"].to(device)
            ctx["rel_graphs"] = [g.to(device) for g in ctx["rel_graphs"]]
        return self

    # ------------------------------------------------------------------ filter masks (task.py:65-100)
    def _calculate_mask(self, graph, anchor_index, pos_r_index, anchor_col):
        """Boolean ``(B, N)``: False at every entity that completes (anchor, r, ?) / (?, r, anchor) in ``graph``."""
        any = -torch.ones_like(anchor_index)
        cols = [anchor_index, any, pos_r_index] if anchor_col == 0 else [any, anchor_index, pos_r_index]
        pattern = torch.stack(cols, dim=-1)
        edge_index, num_truth = graph.match(pattern)
        truth_index = graph.edge_list[edge_index, 1 - anchor_col]
        pos_index = torch.repeat_interleave(num_truth)
        mask = torch.ones(len(pattern), graph.num_node, dtype=torch.bool, device=graph.device)
        mask[pos_index, truth_index] = 0
        return mask

    def _filter_pairs(self, graph, anchor_index, pos_r_index, anchor_col):
        """The same truths as ``_calculate_mask`` as ``(pattern row, entity)`` pairs (duplicates possible)."""
        any = -torch.ones_like(anchor_index)
        cols = [anchor_index, any, pos_r_index] if anchor_col == 0 else [any, anchor_index, pos_r_index]
        edge_index, num_truth = graph.match(torch.stack(cols, dim=-1))
        return torch.repeat_interleave(num_truth), graph.edge_list[edge_index, 1 - anchor_col]

    def target_lists(self, batch):
        """``target`` without the dense ``(B, 2, N)`` mask: per ranking row ``2 b + side`` (side 0 = tail, 1 = head) the
        sorted DISTINCT entities the mask would clear, as CSR lists ``(filt_ptr int32 (2B + 1,), filt_node int32)``,
        plus the positives ``(B, 2)``."""
        batch = self._select(batch)
        pos_h_index, pos_t_index, pos_r_index = batch.t()
        n, rows = self.graph.num_node, 2 * len(batch)
        t_row, t_node = self._filter_pairs(self.graph, pos_h_index, pos_r_index, 0)
        h_row, h_node = self._filter_pairs(self.graph, pos_t_index, pos_r_index, 1)
        key = torch.unique(torch.cat([(2 * t_row) * n + t_node, (2 * h_row + 1) * n + h_node]))      # sorted, distinct
        row = torch.div(key, n, rounding_mode="floor")
        filt_ptr = torch.zeros(rows + 1, dtype=torch.long, device=batch.device)
        torch.cumsum(torch.bincount(row, minlength=rows), 0, out=filt_ptr[1:])
        return (filt_ptr.to(torch.int32), (key - row * n).to(torch.int32)), torch.stack([pos_t_index, pos_h_index], dim=1)

    def _calculate_t_mask(self, graph, pos_h_index, pos_r_index):
        return self._calculate_mask(graph, pos_h_index, pos_r_index, 0)

    def _calculate_h_mask(self, graph, pos_t_index, pos_r_index):
        return self._calculate_mask(graph, pos_t_index, pos_r_index, 1)

    @torch.no_grad()
    def _strict_negative(self, pos_h_index, pos_t_index, pos_r_index):
        """task.py:102-118: first half of the batch corrupts tails, second half heads; negatives are non-edges.
        On the device the candidates are never materialised: the ``k``-th surviving entity is found by binary search
        in the fact graph's sorted completion keys (``csrc/sampler.inc``) -- the same entity the reference's
        ``mask.nonzero()`` + ``variadic_sample`` returns for the same uniform numbers, with no ``(B / 2, N)`` mask, no
        ``nonzero`` and no host synchronisation."""
        static = getattr(self, "_static_negative", None)
        if static is not None:          # a caller that feeds pre-drawn negatives through a static buffer
            return static
        half = len(pos_h_index) // 2
        ops = backend.get()
        if ops.accepts(pos_h_index):
            fact = self.fact_graph
            n, r = fact.num_node, max(fact.num_relation, 1)
            # one draw per half, in the reference's order (variadic_sample draws `torch.rand(rows, num_sample)`)
            rand_t = torch.rand(half, self.num_negative, device=pos_h_index.device)
            neg_t = ops.strict_negatives(fact.completion_keys(0), pos_h_index[:half], pos_r_index[:half], r, n, rand_t)
            rand_h = torch.rand(len(pos_h_index) - half, self.num_negative, device=pos_h_index.device)
            neg_h = ops.strict_negatives(fact.completion_keys(1), pos_t_index[half:], pos_r_index[half:], r, n, rand_h)
            return torch.cat([neg_t, neg_h])
        t_mask = self._calculate_t_mask(self.fact_graph, pos_h_index[:half], pos_r_index[:half])
        neg_t = variadic_sample(t_mask.nonzero()[:, 1], t_mask.sum(dim=-1), self.num_negative)
        h_mask = self._calculate_h_mask(self.fact_graph, pos_t_index[half:], pos_r_index[half:])
        neg_h = variadic_sample(h_mask.nonzero()[:, 1], h_mask.sum(dim=-1), self.num_negative)
        return torch.cat([neg_t, neg_h])

    # ------------------------------------------------------------------ predict (task.py:228-277)
    def relation_representations(self, pos_r_index, all_loss=None, metric=None):
        cache = self._relation_cache.get(self.split) if self._relation_cache else None
        if cache is not None and all_loss is None and not self.training and not torch.is_grad_enabled():
            return [table[pos_r_index] for table in cache]
        return [rel_model(rel_graph, None, pos_r_index, all_loss=all_loss, metric=metric)["node_feature"]
                for rel_model, rel_graph in zip(self.rel_models, self.rel_graphs)]

    @torch.no_grad()
    def cache_relation_representations(self, batch_size=16):
        """Inference only, opt-in.  The relation representations of a query depend on the relation graph, the weights
        and the query's relation alone (ultra/rel_model.py:351-378: the boundary is one row of ones at ``r``), and every
        query of a batch is computed in its own columns -- so the ``(2R, 64)`` table of each of the R relations is
        computed ONCE here (R / batch_size passes of the relation stack) and ``predict`` then picks the rows of its
        batch: the same bits as recomputing them per batch, as the reference does (task.py:238-240), without a sixth
        of an evaluation batch's time on an FB15k237-sized vocabulary.  Dropped by ``train()``,
        ``load_state_dict()`` and :meth:`clear_relation_cache`; one table per graph context."""
        if self.training:
            raise RuntimeError("cache_relation_representations: evaluation only (call eval() first)")
        self._relation_cache.pop(self.split, None)
        device = next(self.parameters()).device
        n_rel = self.fact_graph.num_relation
        parts = [self.relation_representations(torch.arange(i, min(i + batch_size, n_rel), device=device))
                 for i in range(0, n_rel, batch_size)]
        self._relation_cache[self.split] = [torch.cat([p[m] for p in parts]) for m in range(len(self.rel_models))]
        return self

    def clear_relation_cache(self):
        self._relation_cache.clear()
        return self

    def train(self, mode=True):
        if mode:
            self._relation_cache.clear()          # the weights are about to change
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._relation_cache.clear()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):                       # .to(device) / .float() / ...
        self._relation_cache.clear()
        return super()._apply(fn, *args, **kwargs)

    def _select(self, batch):
        """Multi-graph pre-training batches are ``(triples, graph_id)`` (task.py:722-731): switch to that graph."""
        if isinstance(batch, (tuple, list)):
            batch, graph_id = batch
            self.use(graph_id)
        return batch

    def predict(self, batch, all_loss=None, metric=None):
        batch = self._select(batch)
        pos_h_index, pos_t_index, pos_r_index = batch.t()
        batch_size = len(batch)
        rel_inputs = self.relation_representations(pos_r_index, all_loss, metric)
        # (a phased backward resumes from these; kept only while engine.GraphedTrainStep._capture_phased asks for them)
        self.last_relation_inputs = rel_inputs if (all_loss is not None and getattr(self, "record_cuts", False)) else None

        if all_loss is None and self.full_batch_eval and self.fuse_sides:         # evaluation, both sides at once
            # rows 0..B-1: (h, r, ?);  rows B..2B-1: (?, r, t) in tail form = (t, r + R, ?)  (model.py:76-83)
            if len(rel_inputs) == 1 and hasattr(self.model, "score_both_sides"):
                pred = self.model.score_both_sides(self.fact_graph, rel_inputs[0], batch)      # the fused sequence, if it applies
                if pred is not None:
                    return pred.view(2, batch_size, self.num_entity).transpose(0, 1).contiguous()    # (B, 2, N)
            rel2 = [torch.cat([r, r]) for r in rel_inputs]
            pred = self.model.score_all_entities(self.fact_graph, rel2, torch.cat([pos_h_index, pos_t_index]),
                                                 torch.cat([pos_r_index, pos_r_index + self.fact_graph.num_relation]))
            if pred is not None:
                return pred.view(2, batch_size, self.num_entity).transpose(0, 1).contiguous()    # (B, 2, N)
            all_row = torch.arange(self.num_entity, device=batch.device).unsqueeze(0).expand(batch_size, -1)
            pos_h, pos_t = (x.unsqueeze(-1).expand(-1, self.num_entity) for x in (pos_h_index, pos_t_index))
            h_index = torch.cat([pos_h, all_row])           # rows B..2B-1 are head-corrupted: the model flips them
            t_index = torch.cat([all_row, pos_t])           # to tail form with r + R (model.py:76-83)
            r_index = pos_r_index.repeat(2).unsqueeze(-1).expand(-1, self.num_entity)
            pred = self.model(self.fact_graph, [torch.cat([r, r]) for r in rel_inputs], h_index, t_index, r_index,
                              all_entities=True)
            return pred.view(2, batch_size, self.num_entity).transpose(0, 1).contiguous()    # (B, 2, N)

        if all_loss is None:                                                     # evaluation: all entities
            all_index = torch.arange(self.num_entity, device=batch.device)
            num_negative = self.num_entity if self.full_batch_eval else self.num_negative
            t_preds, h_preds = [], []
            for neg_index in all_index.split(num_negative):
                r_index = pos_r_index.unsqueeze(-1).expand(-1, len(neg_index))
                h_index, t_index = torch.meshgrid(pos_h_index, neg_index, indexing="ij")
                t_preds.append(self.model(self.fact_graph, rel_inputs, h_index, t_index, r_index,
                                          all_entities=len(neg_index) == self.num_entity))
            for neg_index in all_index.split(num_negative):
                r_index = pos_r_index.unsqueeze(-1).expand(-1, len(neg_index))
                t_index, h_index = torch.meshgrid(pos_t_index, neg_index, indexing="ij")
                h_preds.append(self.model(self.fact_graph, rel_inputs, h_index, t_index, r_index,
                                          all_entities=len(neg_index) == self.num_entity))
            return torch.stack([torch.cat(t_preds, dim=-1), torch.cat(h_preds, dim=-1)], dim=1)   # (B, 2, N)

        h_index, t_index, r_index = self.training_indices(batch)
        return self.model(self.fact_graph, rel_inputs, h_index, t_index, r_index, all_loss=all_loss, metric=metric)

    def training_indices(self, batch):
        """task.py:264-274: ``(B, 1 + num_negative)`` index grids, column 0 = the positive triple, the rest strict
        negatives (tails corrupted in the first half of the batch, heads in the second)."""
        pos_h_index, pos_t_index, pos_r_index = batch.t()
        batch_size = len(batch)
        neg_index = self._strict_negative(pos_h_index, pos_t_index, pos_r_index)
        self.last_negatives = neg_index
        h_index = pos_h_index.unsqueeze(-1).repeat(1, self.num_negative + 1)
        t_index = pos_t_index.unsqueeze(-1).repeat(1, self.num_negative + 1)
        r_index = pos_r_index.unsqueeze(-1).repeat(1, self.num_negative + 1)
        t_index[:batch_size // 2, 1:] = neg_index[:batch_size // 2]
        h_index[batch_size // 2:, 1:] = neg_index[batch_size // 2:]
        return h_index, t_index, r_index

    def target(self, batch):
        """task.py:279-295: filter masks over the FULL graph and the true tail / head of each triple."""
        batch = self._select(batch)
        pos_h_index, pos_t_index, pos_r_index = batch.t()
        t_mask = self._calculate_t_mask(self.graph, pos_h_index, pos_r_index)
        h_mask = self._calculate_h_mask(self.graph, pos_t_index, pos_r_index)
        return torch.stack([t_mask, h_mask], dim=1), torch.stack([pos_t_index, pos_h_index], dim=1)

    def get_ranking(self, pred, target):
        """task.py:307-315: ``sum((pos_pred <= pred) & mask, -1) + 1`` -> int64 ``(B, 2)``."""
        mask, target = target
        pos_pred = pred.gather(-1, target.unsqueeze(-1))
        if self.filtered_ranking:
            return torch.sum((pos_pred <= pred) & mask, dim=-1) + 1
        return torch.sum(pos_pred <= pred, dim=-1) + 1

    @torch.no_grad()
    def rank_batch(self, batch, pred=None):
        """Scores, filters and ranks stay on the device; only ``(B, 2)`` int64 ranks are returned.  On the device the
        filter is a pair of CSR lists and the count runs in ``libultra_rspmm`` (``ultra_filtered_rank``); the dense
        ``(B, 2, N)`` masks of ``target`` / ``get_ranking`` (task.py:279-315) remain the CPU / cross-check path."""
        if pred is None:        # (a caller that replays `predict` as a hipGraph passes its scores in)
            pred = self.predict(batch)
        ops = backend.get()
        if ops.accepts(pred):
            batch = self._select(batch)
            pos_h_index, pos_t_index, pos_r_index = batch.t()
            graph = self.graph
            n_rel = max(graph.num_relation, 1)
            keys = (graph.completion_keys(0), graph.completion_keys(1)) if self.filtered_ranking else (None, None)
            # the filter of each side is a range of the graph's sorted completion keys, found inside the kernel:
            # no (B, N) mask, no per-batch list building, no host synchronisation (capturable with predict)
            t_rank = ops.filtered_rank_keys(pred[:, 0], pos_t_index, keys[0], pos_h_index, pos_r_index, n_rel, graph.num_node)
            h_rank = ops.filtered_rank_keys(pred[:, 1], pos_h_index, keys[1], pos_t_index, pos_r_index, n_rel, graph.num_node)
            return torch.stack([t_rank, h_rank], dim=1)
        return self.get_ranking(pred, self.target(batch))

    def evaluate(self, ranking, rel=None):
        """task.py:317-351 on an int64 ``(n, 2)`` ranking tensor (column 0 = tail, 1 = head).  With
        ``metric_per_rel`` and ``rel`` (``(n,)`` relation of every ranked triple) every undirected metric is also
        reported per relation (task.py:290-292,512-517: tails under ``r``, heads under ``r + num_relation``)."""
        metric = {}
        if self.metric_per_rel and rel is None:
            raise ValueError("metric_per_rel needs the relation of every ranked triple: evaluate(ranking, rel)")
        for name in self.metric:
            _ranking, _name = ranking, name
            if "-" in name:
                _name, direction = name.split("-")
                if direction not in ("head", "tail"):
                    raise ValueError("Unknown direction `%s`" % direction)
                _ranking = ranking.select(1, 1 if direction == "head" else 0)
            if _name == "mr":
                score = _ranking.float().mean()
            elif _name == "mrr":
                score = (1 / _ranking.float()).mean()
            elif _name.startswith("hits@"):
                score = (_ranking <= int(_name[5:])).float().mean()
            else:
                raise ValueError("Unknown metric `%s`" % name)
            metric[name] = score
            if self.metric_per_rel and "-" not in name:
                n_rel = self.num_relation
                rel2 = torch.stack([rel, rel + n_rel], dim=1).reshape(-1).to(ranking.device)
                value = ranking.reshape(-1).float()
                value = value if _name == "mr" else (1 / value if _name == "mrr" else (value <= int(_name[5:])).float())
                total = torch.zeros(2 * n_rel, device=ranking.device).index_add_(0, rel2, value)
                count = torch.zeros(2 * n_rel, device=ranking.device).index_add_(0, rel2, torch.ones_like(value))
                for ridx in range(2 * n_rel):
                    metric["%s_rel_%d" % (name, ridx)] = total[ridx] / count[ridx].clamp(min=1)
        return metric

    # ------------------------------------------------------------------ training loss (task.py:160-195)
    def forward(self, batch, all_loss=None, metric=None):
        batch = self._select(batch)
        all_loss = torch.zeros((), dtype=torch.float32, device=batch.device)
        metric = {}
        pred = self.predict(batch, all_loss, metric)
        first = True                                        # (nothing on the path adds to `all_loss` before the criteria)
        pos_h_index, pos_t_index, pos_r_index = batch.t()
        names = {"bce": "binary cross entropy", "ce": "cross entropy", "ranking": "ranking loss"}
        for criterion, weight in self.criterion.items():
            if criterion == "bce" and backend.get().accepts(pred) and pred.dtype == torch.float32 and pred.dim() == 2:
                loss = backend.get().bce_adversarial_loss(pred, self.adversarial_temperature)     # task.py:169-180, one launch
            elif criterion == "bce":                                                 # task.py:169-180
                target = torch.zeros_like(pred)
                target[:, 0] = 1
                loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none")
                neg_weight = torch.ones_like(pred)
                if self.adversarial_temperature > 0:
                    with torch.no_grad():
                        neg_weight[:, 1:] = F.softmax(pred[:, 1:] / self.adversarial_temperature, dim=-1)
                else:
                    neg_weight[:, 1:] = 1 / self.num_negative
                loss = (loss * neg_weight).sum(dim=-1) / neg_weight.sum(dim=-1)
            elif criterion == "ce":                                                  # task.py:698-700
                loss = F.cross_entropy(pred, torch.zeros(len(pred), dtype=torch.long, device=pred.device),
                                       reduction="none")
            elif criterion == "ranking":                                             # task.py:701-705
                loss = F.margin_ranking_loss(pred[:, :1], pred[:, 1:], torch.ones_like(pred[:, 1:]),
                                             margin=self.margin)
            else:
                raise ValueError("Unknown criterion `%s`" % criterion)
            if self.sample_weight:                                                   # task.py:184-187
                degree = self._pair_degree("hr", pos_h_index, pos_r_index) * self._pair_degree("tr", pos_t_index, pos_r_index)
                sample_weight = 1 / degree.float().sqrt()
                loss = (loss * sample_weight).sum() / sample_weight.sum()
            loss = loss.mean()
            metric[names[criterion]] = loss
            # (`0 + loss * 1` of the shipped configuration without its two launches forward and one backward: the same value)
            term = loss if weight == 1 else loss * weight
            all_loss = term if first else all_loss + term
            first = False
        return all_loss, metric


def build_ultra(num_relation, input_dim=64, hidden_dims=(64,) * 6, rel_hidden=64, rel_layers=6, **task_kwargs):
    """The shipped architecture: ``config/transductive/inference.yaml:8-39`` (6 x 64d, distmult, sum, shortcut,
    layer norm, projected relations; relation model 6 x 64d)."""
    from .model import TransferNBFNet
    from .rel_model import RelationModelList
    model = TransferNBFNet(input_dim=input_dim, hidden_dims=list(hidden_dims), num_relation=num_relation,
                           message_func="distmult", aggregate_func="sum", short_cut=True, layer_norm=True,
                           project=True, mod=True, remove_one_hop=False)
    rel_models = RelationModelList(num_rel_models=1, num_relation=2 * num_relation,
                                   rel_model=dict(class_str="RelNBFNet", input_dim=input_dim, input_type="ones",
                                                  num_layers=rel_layers, hidden=rel_hidden))
    defaults = dict(criterion="bce", num_negative=128, strict_negative=True, adversarial_temperature=1.0,
                    sample_weight=False, full_batch_eval=True)
    defaults.update(task_kwargs)
    return KnowledgeGraphCompletion(model, rel_models, **defaults)
