"""Query-sharded multi-GPU driver for the hot path (one process per GPU, ``torch.distributed``).

The path shards over QUERIES only: every rank holds the whole graph and the 0.78 MB of weights and runs
Bellman-Ford for its own batch elements; there is no exchange inside rspmm.  What crosses ranks:

* evaluation -- one ``all_gather`` of the int64 ``(n_local, 2)`` rankings at the end (the reference gathers the
  full ``(n, 2, N)`` score tensors through gloo: ``/root/reference/ultra/engine.py:146-151``);
* training -- ONE all-reduce of a flat fp32 gradient buffer per step (the reference: DDP with
  ``find_unused_parameters=True``, ``ultra/engine.py:55-60``; parameters that never receive a gradient --
  ``model.dist_embed``, the relation model's ``mlp`` -- are simply absent from the buffer), plus one packed
  all-reduce for the metric scalars (``ultra/engine.py:90``).

Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` on CPU (tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (set by ``torch.distributed.run``)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return get_rank(), get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    return get_rank(), get_world_size()


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_indices(n, rank=None, world=None):
    """Strided shard ``rank, rank + world, ...`` of ``range(n)`` (DistributedSampler order, no padding)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    return torch.arange(rank, n, world)


def gather_variable(local):
    """All-gather tensors whose first dimension differs per rank; returns them re-interleaved in the strided
    order of :func:`shard_indices`, i.e. in the original global order."""
    world = get_world_size()
    if world == 1:
        return local
    n_local = torch.tensor([local.shape[0]], dtype=torch.long, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    total = sum(sizes)
    out = torch.zeros((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, (part, size) in enumerate(zip(parts, sizes)):
        out[r:total:world][:size] = part[:size]
    return out


class GraphedPredict:
    """``task.predict`` for a fixed batch size as ONE hipGraph: an evaluation batch is ~250 small launches
    (18 rspmm + epilogues + relation projections + score MLP) whose host-side issue cost exceeds their GPU time
    on small graphs; replaying a captured graph removes it.  The reference's two per-call index asserts
    (``ultra/model.py:174-175``) are host syncs and cannot be captured; ``predict`` builds those index grids
    itself (``ultra/task.py:249-259``) so they hold by construction and are switched off for the capture."""

    def __init__(self, task, example_batch, warmup=3):
        assert example_batch.is_cuda and not task.training
        self.task = task
        self.static_batch = example_batch.clone()
        model = task.model
        model.check_indices = False
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):             # builds plans, sets kernel attributes, warms the allocator
                    task.predict(self.static_batch)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the RCCL watchdog polls events) must not abort the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
                self.static_pred = task.predict(self.static_batch)
        finally:
            model.check_indices = True

    def __call__(self, batch):
        """Scores ``(B, 2, N)``; the returned tensor is overwritten by the next call."""
        if batch.shape != self.static_batch.shape:
            with torch.no_grad():
                return self.task.predict(batch)     # ragged last batch: eager path
        self.static_batch.copy_(batch)
        self.graph.replay()
        return self.static_pred


class GraphedTrainStep:
    """Forward + backward of one fine-tuning step (``ultra/engine.py:62-92``) for a fixed batch size as ONE hipGraph.
    A step is ~800 launches, most of them tiny, and their host-side issue cost exceeds the GPU time of the kernels.
    What has data-dependent shapes stays eager and feeds static buffers: the strict negatives (``nonzero`` over the
    filter masks, task.py:102-118) and the mask of the batch's own edges (``graph.match``, model.py:57-74).  The
    gradient all-reduce and the optimizer step follow the replay eagerly (``train_step`` semantics)."""

    def __init__(self, task, optimizer, example_batch, warmup=3):
        assert example_batch.is_cuda and task.training
        self.task, self.optimizer = task, optimizer
        model = task.model
        self.static_batch = example_batch.clone()
        self.static_neg, self.static_keep = self._eager_inputs(self.static_batch)
        self.static_neg, self.static_keep = self.static_neg.clone(), self.static_keep.clone()
        model.check_indices = False
        task._static_negative, model._static_keep = self.static_neg, self.static_keep
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):             # plans, kernel attributes, workspaces, allocator
                    optimizer.zero_grad(set_to_none=True)
                    loss, _ = task(self.static_batch)
                    loss.backward()
            torch.cuda.current_stream().wait_stream(side)
            optimizer.zero_grad(set_to_none=True)   # the captured backward allocates the gradients in the graph's pool
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
                self.static_loss, self.static_metric = task(self.static_batch)
                self.static_loss.backward()
        finally:
            model.check_indices = True
            task._static_negative, model._static_keep = None, None

    def _eager_inputs(self, batch):
        task, model = self.task, self.task.model
        task._static_negative, model._static_keep = None, None
        h_index, t_index, r_index = task.training_indices(batch)
        keep = model.easy_edge_mask(task.fact_graph, h_index, t_index, r_index)
        half = len(batch) // 2
        neg = torch.cat([t_index[:half, 1:], h_index[half:, 1:]])
        return neg, keep

    def __call__(self, batch):
        """One step on ``batch`` (same shape as the example): returns ``(loss, metrics averaged over ranks)``."""
        assert batch.shape == self.static_batch.shape
        neg, keep = self._eager_inputs(batch)
        self.static_batch.copy_(batch)
        self.static_neg.copy_(neg)
        self.static_keep.copy_(keep)
        self.graph.replay()
        allreduce_gradients(self.task)
        self.optimizer.step()
        return self.static_loss.detach(), reduce_metrics(self.static_metric)


@torch.no_grad()
def evaluate(task, triples, batch_size=16, graphed=None):
    """Filtered ranking of ``triples`` ((n, 3) rows of (h, t, r)) sharded over ranks; every rank returns the
    metrics of the WHOLE set.  Only int64 ranks cross ranks.  ``graphed`` (default: on a GPU, when the shard holds at
    least two full batches): ``predict`` is captured once and replayed as a hipGraph for every full batch."""
    device = task.device
    mine = shard_indices(len(triples))
    local = triples[mine].to(device)
    if graphed is None:
        graphed = device.type == "cuda" and len(local) >= 2 * batch_size and not task.training
    replay = GraphedPredict(task, local[:batch_size]) if graphed and len(local) >= batch_size else None
    ranks = []
    for i in range(0, len(local), batch_size):
        batch = local[i:i + batch_size]
        pred = replay(batch) if replay is not None and len(batch) == batch_size else None
        ranks.append(task.rank_batch(batch, pred=pred))
    ranks = torch.cat(ranks) if ranks else torch.zeros(0, 2, dtype=torch.long, device=device)
    ranking = gather_variable(ranks)
    return task.evaluate(ranking), ranking


def sample_edges_from_graph(task, batch_size, generator=None):
    """Multi-graph pre-training batch (``ultra/engine.py:23-34``): draw a graph with probability proportional to its
    number of fact edges, then ``batch_size`` distinct fact edges of it.  Returns ``(triples, graph_id)``.  Every rank
    draws independently (ranks are seeded ``seed + rank``, ``script/run_full.py:102-107``), so ranks may train on
    different graphs in the same step; the gradient all-reduce is also the straggler barrier."""
    names = sorted(task.contexts)
    sizes = torch.tensor([task.contexts[n]["fact_graph"].num_edge for n in names], dtype=torch.float)
    graph_id = names[int(torch.multinomial(sizes / sizes.sum(), 1, generator=generator))]
    fact = task.contexts[graph_id]["fact_graph"]
    pick = torch.randperm(fact.num_edge, generator=generator)[:batch_size]
    return fact.edge_list[pick.to(fact.device)], graph_id


def allreduce_gradients(module, average=True):
    """One flat all-reduce over every parameter that has a gradient on this step.  Which parameters have one is
    a static property of the architecture, so all ranks build the same buffer."""
    world = get_world_size()
    params = [p for p in module.parameters() if p.grad is not None]
    if world == 1 or not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat)
    if average:
        flat /= world
    offset = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[offset:offset + n].view_as(p.grad))
        offset += n
    return flat.numel()


def reduce_metrics(metric):
    """Mean over ranks of a dict of 0-d tensors, in one packed all-reduce (``ultra/engine.py:90``)."""
    world = get_world_size()
    if world == 1 or not metric:
        return metric
    keys = sorted(metric)
    packed = torch.stack([metric[k].detach().float().reshape(()) for k in keys])
    dist.all_reduce(packed)
    packed /= world
    return {k: packed[i] for i, k in enumerate(keys)}


def train_step(task, optimizer, batch):
    """One fine-tuning step (``ultra/engine.py:62-92`` / torchdrug ``Engine.train``): forward, backward,
    gradient all-reduce, optimizer step.  Returns (loss, metrics averaged over ranks)."""
    loss, metric = task(batch)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    allreduce_gradients(task)
    optimizer.step()
    return loss.detach(), reduce_metrics(metric)
